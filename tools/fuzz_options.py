#!/usr/bin/env python3
"""Random sweep over the graph options against the fp64 oracle (not part of the test suite: run
on a GPU box to hunt for option interactions).  Usage: python tools/fuzz_options.py [cases] [seed0]
(FUZZ_B_RANGE="lo,hi": cells per step, default 3,30)"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))   # the tests' shared helpers (_parity)
from oracle import models as om  # noqa: E402

spec = importlib.util.spec_from_file_location(
    "dropout_helpers", os.path.join(ROOT, "tests", "test_gpu_dropout.py"))
helpers = importlib.util.module_from_spec(spec)
spec.loader.exec_module(helpers)

# cells per step: FUZZ_B_RANGE="130,700" moves the sweep onto the kernels of large minibatches (the
# producer / consumer head kernel, the tile chain)
B_RANGE = tuple(int(v) for v in os.environ.get("FUZZ_B_RANGE", "3,30").split(","))

COUNT = ["poisson", "negative binomial", "zero-inflated poisson",
         "zero-inflated negative binomial"]


def case(seed):
    rng = np.random.default_rng(seed)
    gm = bool(rng.integers(0, 2))
    likelihood = str(rng.choice(COUNT + ["constrained poisson", "bernoulli"]))
    k_max = 0
    if likelihood in ("poisson", "negative binomial") and rng.random() < 0.3:
        k_max = int(rng.integers(1, 4))
    n_hidden = int(rng.integers(1, 3))
    c = dict(
        gm=gm, likelihood=likelihood, k_max=k_max,
        F=int(rng.integers(4, 120)), L=int(rng.integers(1, 7)),
        H=tuple(int(2 * rng.integers(2, 16)) for _ in range(n_hidden)),
        B=int(rng.integers(*B_RANGE)), K=int(rng.integers(2, 4)) if gm else 1,
        n_iw=int(rng.integers(1, 3)), n_mc=int(rng.integers(1, 3)),
        bn=bool(rng.integers(0, 2)), warm_up=float(rng.choice([1.0, 0.4])),
        extra=int(rng.choice([0, 0, 2])),
        keeps=tuple(float(rng.choice([0.0, 0.0, 0.6, 0.9]))
                    for _ in range(4 if gm else 3)),
    )
    if gm:
        c["prior"] = str(rng.choice(["uniform", "learn", "custom"]))
        c["free_nats"] = float(rng.choice([0.0, 0.6]))
        c["n_mc"] = 1
    else:
        c["latent"] = str(rng.choice(["gaussian", "gaussian",
                                      "unit-variance gaussian"]))
        c["analytical"] = bool(rng.integers(0, 2))
        c["inference"] = str(rng.choice(["MLP", "MLP", "LFM"]))
        c["generative"] = str(rng.choice(["MLP", "MLP", "LFM"]))
        if c["generative"] == "LFM":
            c["extra"] = 0
    return c


def run(c, seed, device):
    from scvae_amd.engine import Engine
    rng = np.random.default_rng(10_000 + seed)
    gm = c["gm"]
    F, L, H, B, K = c["F"], c["L"], c["H"], c["B"], c["K"]
    S = c["n_iw"] * c["n_mc"]
    kwargs = dict(batch_norm=c["bn"], device=device, seed=seed,
                  decoder_extra=c["extra"], k_max=c["k_max"],
                  dropout_keep_probabilities=c["keeps"])
    cfg_kwargs = dict(feature_size=F, latent_size=L, hidden_sizes=H,
                      likelihood=c["likelihood"], minibatch_normalisation=c["bn"],
                      n_iw=c["n_iw"], n_mc=c["n_mc"], k_max=c["k_max"],
                      decoder_extra_size=c["extra"])
    if gm:
        probabilities = None
        if c["prior"] == "custom":
            probabilities = rng.dirichlet(np.ones(K)).tolist()
        kwargs.update(model_type="GMVAE", n_clusters=K,
                      free_nats_proportion=c["free_nats"],
                      prior_probabilities_method=c["prior"],
                      prior_probabilities=probabilities)
        cfg_kwargs.update(n_clusters=K, free_nats_proportion=c["free_nats"],
                          prior_probabilities_method=c["prior"],
                          prior_probabilities=tuple(probabilities or ()))
    else:
        kwargs.update(latent_distribution=c["latent"],
                      analytical_kl_term=c["analytical"],
                      inference_architecture=c["inference"],
                      generative_architecture=c["generative"])
        cfg_kwargs.update(latent_distribution=c["latent"],
                          analytical_kl_term=c["analytical"],
                          inference_architecture=c["inference"],
                          generative_architecture=c["generative"])
    eng = Engine(F, L, H, c["likelihood"], **kwargs)
    g = torch.Generator().manual_seed(seed)
    for name, p in eng.named_parameters().items():
        if not name.endswith("weights") and name != "Y/P/LOGITS":
            p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    cfg = om.ModelConfig(**cfg_kwargs)
    params = {k: v.detach().cpu().double()
              for k, v in eng.named_parameters().items()}
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    x = (rng.poisson(2.0, (B, F)) * (rng.random((B, F)) < 0.5)).astype(np.float64)
    x[:, 0] += 1
    x = torch.from_numpy(x)
    t = (x > 0.5).double() if c["likelihood"] == "bernoulli" else x
    count_sum = x.sum(dim=1)
    eps = torch.from_numpy(rng.standard_normal(
        (K, S, B, L) if gm else (S, B, L)))
    extra = torch.from_numpy(rng.random((B, c["extra"]))) if c["extra"] else None
    masks = None
    if any(c["keeps"]):
        if gm:
            masks = helpers._gmvae_masks(eng, cfg, B, S, c["keeps"], c["k_max"])
        else:
            masks = helpers._vae_masks(eng, cfg, B, S * B, c["keeps"], c["k_max"])
    step_kwargs = {}
    oracle_kwargs = {}
    if c["likelihood"] == "constrained poisson":
        step_kwargs["count_sum"] = count_sum.float().to(device)
        oracle_kwargs["count_sum"] = count_sum
    if extra is not None:
        step_kwargs["decoder_extra"] = extra.float().to(device)
        oracle_kwargs["decoder_extra"] = extra
    if masks is not None:
        step_kwargs["dropout_seed"] = helpers.SEED
        oracle_kwargs["dropout"] = masks
    sc = eng.step(x.float().to(device), t.float().to(device),
                  eps=eps.float().to(device), training=True, n_iw=c["n_iw"],
                  n_mc=c["n_mc"], warm_up_weight=c["warm_up"],
                  **step_kwargs).cpu().numpy()
    torch.cuda.synchronize()
    forward = om.gmvae_forward if gm else om.vae_forward
    out, grads = om.gradients(
        lambda p: forward(cfg, p, moving, x, t, eps, True, c["warm_up"], {},
                          **oracle_kwargs), params)
    problems = []

    def check(got, want, rtol, what):
        got = np.asarray(got, dtype=np.float64)
        want = np.asarray(want, dtype=np.float64)
        scale = max(np.abs(want).max(), 1e-30)
        err = np.abs(got - want).max() / scale
        if not err <= rtol and np.abs(got - want).max() > 1e-9:
            where = ""
            if what.endswith("BATCH_NORM/beta"):
                # (a ReLU kink -- fp32 and fp64 put a normalised activation within an ulp of zero on
                #  different sides -- shows as ONE unit of the topmost affected layer)
                units = np.nonzero(np.abs(got - want) > rtol * scale)[0].tolist()
                where = "  units " + str(units[:8]) + (" ..." if len(units) > 8 else "")
            problems.append("{}: {:.2e} of {:.2e}{}".format(what, err, scale, where))
    if gm and c["free_nats"]:
        # the free-nats gate is a comparison: near the threshold fp32 and fp64 may disagree
        thr = c["free_nats"] * float(np.log(K))
        if abs(float(out["kl_divergence_y"]) - thr) < 2e-3 * thr:
            return []
    check(sc[0], out["lower_bound"], 2e-4, "lower_bound")
    check(sc[1], out["lower_bound_weighted"], 2e-4, "lower_bound_weighted")
    for name, gr in eng.named_gradients().items():
        if c["bn"] and name.endswith("DENSE/biases") and (
                "ENCODER" in name or "DECODER" in name or "LAYER_" in name):
            continue
        got, want = gr.cpu(), grads[name]
        if (gm and c["bn"] and name == "Z/Q/ENCODER/LAYER_1/DENSE/weights"
                and not c["keeps"][1]):
            got, want = got[:F], want[:F]
        check(got, want, 3e-3 if (gm or c["n_iw"] > 1) else 5e-4, "grad " + name)
    # ---- evaluation step (is_training = False) with the reconstruction statistics ----
    moving = {k: v.detach().cpu().double()
              for k, v in eng.named_moving_statistics().items()}
    outs = {k: torch.zeros(B, F, device=device) for k in (
        "p_x_mean", "p_x_stddev", "stddev_of_p_x_given_z_mean")}
    step_kwargs.pop("dropout_seed", None)
    oracle_kwargs.pop("dropout", None)
    sc = eng.step(x.float().to(device), t.float().to(device),
                  eps=eps.float().to(device), training=False, n_iw=c["n_iw"],
                  n_mc=c["n_mc"], warm_up_weight=c["warm_up"], outputs=outs,
                  **step_kwargs).cpu().numpy()
    torch.cuda.synchronize()
    out = forward(cfg, params, moving, x, t, eps, False, c["warm_up"],
                  evaluation_statistics=True, **oracle_kwargs)
    check(sc[0], out["lower_bound"], 2e-4, "lower_bound (evaluation)")
    check(outs["p_x_mean"].cpu(), out["p_x_mean"], 5e-4, "p_x_mean")
    check(outs["p_x_stddev"].cpu(), out["p_x_stddev"], 5e-4, "p_x_stddev")
    return problems


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    device = torch.device("cuda:0")
    failures = 0
    for seed in range(seed0, seed0 + cases):
        c = case(seed)
        try:
            problems = run(c, seed, device)
        except Exception as error:   # report and go on
            problems = ["exception: {!r}".format(error)]
        if problems:
            failures += 1
            print("seed", seed, c)
            for problem in problems[:6]:
                print("    ", problem)
    print("{} of {} cases differ".format(failures, cases))


if __name__ == "__main__":
    main()
