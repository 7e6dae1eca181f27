#!/usr/bin/env python3
"""Random sweep over the model classes' constructor options: train two epochs, evaluate, check
that everything stays finite (host-side plumbing of option combinations; run on a GPU box).
Usage: python tools/fuzz_models.py [cases] [seed0]"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from scvae_amd.data import DataSet
    from scvae_amd.models import (
        GaussianMixtureVariationalAutoencoder, VariationalAutoencoder)
    from scvae_amd.models.utilities import load_learning_curves
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    failures = 0
    for seed in range(seed0, seed0 + cases):
        rng = np.random.default_rng(seed)
        n, F = int(rng.integers(60, 140)), int(rng.integers(12, 60))
        x = rng.poisson(1.5, size=(n, F)).astype(np.float32)
        x[:, 0] += 1
        labels = np.array(["c%d" % k for k in rng.integers(0, 3, size=n)])
        batches = rng.integers(0, 2, size=n)
        data = DataSet("toy", values=x, labels=labels,
                       example_names=np.arange(n).astype(str),
                       feature_names=np.arange(F).astype(str),
                       batch_indices=batches, kind="training")
        gm = bool(rng.integers(0, 2))
        likelihood = str(rng.choice([
            "poisson", "negative binomial", "zero-inflated poisson",
            "zero-inflated negative binomial", "constrained poisson",
            "bernoulli"]))
        options = dict(
            feature_size=F, latent_size=int(rng.integers(1, 5)),
            hidden_sizes=[int(2 * rng.integers(2, 10))
                          for _ in range(rng.integers(1, 3))],
            reconstruction_distribution=likelihood,
            minibatch_normalisation=bool(rng.integers(0, 2)),
            number_of_warm_up_epochs=int(rng.choice([0, 2])),
            dropout_keep_probabilities=[
                float(rng.choice([1.0, 0.8])) for _ in range(4 if gm else 3)],
            count_sum=bool(rng.integers(0, 2)),
            number_of_importance_samples=int(rng.integers(1, 3)),
            kl_weight=float(rng.choice([1.0, 0.5])))
        if likelihood in ("poisson", "negative binomial") and rng.random() < 0.3:
            options["number_of_reconstruction_classes"] = int(rng.integers(1, 4))
        if rng.random() < 0.3:
            options.update(batch_correction=True, number_of_batches=2)
        if gm:
            cls = GaussianMixtureVariationalAutoencoder
            options.update(
                number_of_latent_clusters=int(rng.integers(2, 4)),
                prior_probabilities_method=str(rng.choice(["uniform", "learn"])),
                proportion_of_free_nats_for_y_kl_divergence=float(
                    rng.choice([0.0, 0.5])),
                latent_distribution=str(rng.choice([
                    "gaussian mixture", "legacy gaussian mixture"])))
        else:
            cls = VariationalAutoencoder
            options.update(
                number_of_monte_carlo_samples=int(rng.integers(1, 3)),
                latent_distribution=str(rng.choice([
                    "gaussian", "unit-variance gaussian"])),
                inference_architecture=str(rng.choice(["MLP", "MLP", "LFM"])),
                generative_architecture="MLP")
            if rng.random() < 0.3:
                options["analytical_kl_term"] = False
        try:
            with tempfile.TemporaryDirectory() as directory:
                model = cls(log_directory=directory, **options)
                model.train(data, data, number_of_epochs=2,
                            minibatch_size=int(rng.integers(7, 40)),
                            learning_rate=1e-3)
                curves = load_learning_curves(model)
                for kind in ("training", "validation"):
                    assert np.isfinite(curves[kind]["lower_bound"]).all(), kind
                outputs = model.evaluate(data, minibatch_size=25)
                reconstructed = outputs[1]
                assert np.isfinite(np.asarray(reconstructed.values)).all()
        except Exception as error:
            failures += 1
            print("seed", seed, cls.__name__, options)
            print("    ", repr(error))
    print("{} of {} cases failed".format(failures, cases))


if __name__ == "__main__":
    main()
