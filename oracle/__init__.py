"""CPU oracle for the scVAE VAE/GMVAE hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in float64 on the CPU, the arithmetic that the
reference (scvae/scvae v2.1.4, TensorFlow 1.15 + TFP 0.7) performs on the
path named by BASELINE.json: graph build + train step + evaluate step of
``scvae/models/{variational_autoencoder,gaussian_mixture_variational_autoencoder}.py``
and the log-prob ops of ``scvae/distributions/``.

PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
for this path, and neither TensorFlow 1.15 nor TFP 0.7 can be imported in
the build container, so the oracle cannot be checked
against reference outputs.  It is pinned instead against independent
closed-form implementations (``scipy.stats``, ``scipy.special``,
``torch.distributions``) in ``tests/test_oracle_kat.py``.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product (``scvae_amd``) never does.
"""
