"""scVAE for AMD Instinct MI355X: the VAE / GMVAE training and evaluation
path of scvae/scvae behind the reference's Python surface
(``scvae.models``, ``scvae.distributions``, ``scvae train`` / ``evaluate``),
executed by hand-written gfx950 HIP kernels (``libscvae_hip.so``)."""

__version__ = "0.1.0"
__title__ = "scvae"
__description__ = ("Model single-cell transcript counts using deep learning "
                   "(MI355X-native training/evaluation path).")
