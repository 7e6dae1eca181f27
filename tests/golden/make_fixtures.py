#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/.

The reference (scvae/scvae, TensorFlow 1.15 / TFP 0.7) cannot be imported in
the build container and ships no fixtures of its own, so
these vectors come from (i) independent closed-form implementations
(scipy.stats / scipy.special) and (ii) the fp64 oracle in oracle/ ("parity
unpinned": they pin the oracle against regressions and give the GPU tests
fixed inputs/outputs; they are NOT reference outputs).  Also records the first
rows and a checksum of the reference's deterministic ``development`` data set
generator as restated in scvae_amd/data/synthetic.py.

    python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np
import scipy.stats as st
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import models as om  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def likelihood_kat():
    rng = np.random.RandomState(7)
    n = 64
    t = np.concatenate([np.zeros(8), rng.poisson(4.0, n - 10), [300., 5000.]])
    a_p = rng.normal(0, 2.0, n)
    a_r = rng.normal(0, 2.0, n)
    a_pi = rng.normal(0, 2.0, n)
    a_l = rng.normal(0, 2.0, n)
    p = 1 / (1 + np.exp(-a_p))
    pi = 1 / (1 + np.exp(-a_pi))
    nb = st.nbinom.logpmf(t, np.exp(a_r), 1 - p)
    po = st.poisson.logpmf(t, np.exp(a_l))
    with np.errstate(divide="ignore"):
        zinb = np.where(t > 0, np.log1p(-pi) + nb,
                        np.logaddexp(np.log(pi), np.log1p(-pi) + nb))
        zip_ = np.where(t > 0, np.log1p(-pi) + po,
                        np.logaddexp(np.log(pi), np.log1p(-pi) + po))
    np.savez(os.path.join(HERE, "likelihood_kat.npz"), t=t, a_p=a_p, a_r=a_r,
             a_pi=a_pi, a_l=a_l, poisson=po, negative_binomial=nb,
             zero_inflated_poisson=zip_,
             zero_inflated_negative_binomial=zinb)


def _flat(d):
    return {k.replace("/", "__"): v.numpy() for k, v in d.items()}


def step_fixture(kind, name, likelihood, F, L, H, B, K=1, S=1, bn=True,
                 seed=0):
    cfg = om.ModelConfig(feature_size=F, latent_size=L, hidden_sizes=H,
                         likelihood=likelihood, minibatch_normalisation=bn,
                         n_clusters=K, n_iw=S, n_mc=1)
    shapes = (om.vae_parameter_shapes(cfg) if kind == "vae"
              else om.gmvae_parameter_shapes(cfg))
    params = om.init_parameters(shapes, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    for k in params:
        if not k.endswith("weights"):
            params[k] = torch.randn(params[k].shape, generator=g,
                                    dtype=torch.float64) * 0.1
    moving = om.init_moving_statistics(shapes)
    rng = np.random.RandomState(seed)
    lam = rng.gamma(0.5, 3.0, size=(1, F))
    x = rng.poisson(lam, size=(B, F)) * (rng.rand(B, F) > 0.6)
    x = torch.from_numpy(x.astype(np.float64))
    eps_shape = (S, B, L) if kind == "vae" else (K, S, B, L)
    eps = torch.from_numpy(rng.standard_normal(eps_shape))
    state = om.adam_state(params)
    step = om.vae_train_step if kind == "vae" else om.gmvae_train_step
    new_params, new_moving, out, grads = step(
        cfg, dict(params), moving, state, x, x, eps, 1e-3, warm_up_weight=0.5)
    arrays = {"x": x.numpy(), "eps": eps.numpy()}
    arrays.update({"param__" + k: v for k, v in _flat(params).items()})
    arrays.update({"grad__" + k: v for k, v in _flat(grads).items()})
    arrays.update({"new_param__" + k: v for k, v in _flat(new_params).items()})
    arrays.update({"new_moving__" + k: v
                   for k, v in _flat(new_moving).items()})
    for key in ("lower_bound", "lower_bound_weighted", "reconstruction_error",
                "kl_divergence", "kl_divergence_z", "kl_divergence_y",
                "log_p_x_given_z", "q_z_mean", "z_mean", "q_y_logits"):
        if key in out:
            arrays["out__" + key] = out[key].numpy()
    meta = dict(kind=kind, likelihood=likelihood, F=F, L=L, H=list(H), B=B,
                K=K, S=S, bn=bn, learning_rate=1e-3, warm_up_weight=0.5)
    arrays["meta"] = np.array(repr(meta))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)



if __name__ == "__main__":
    likelihood_kat()
    step_fixture("vae", "vae_step_nb", "negative binomial", 41, 5, (12, 10),
                 23)
    step_fixture("vae", "vae_step_zinb_iw", "zero-inflated negative binomial",
                 33, 4, (9,), 11, S=3, seed=1)
    step_fixture("vae", "vae_step_poisson_nobn", "poisson", 100, 2, (100,),
                 100, bn=False, seed=2)
    step_fixture("gmvae", "gmvae_step_nb", "negative binomial", 37, 4,
                 (10, 8), 13, K=3, seed=3)
    step_fixture("gmvae", "gmvae_step_zip", "zero-inflated poisson", 29, 3,
                 (8,), 9, K=2, S=2, seed=4)
    print("fixtures written to", HERE)
