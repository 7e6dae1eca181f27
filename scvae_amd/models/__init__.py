"""Model classes of the scVAE drop-in (``scvae/models/__init__.py:19-36``)."""

from scvae_amd.models.variational_autoencoder import VariationalAutoencoder
from scvae_amd.models.gaussian_mixture_variational_autoencoder import (
    GaussianMixtureVariationalAutoencoder)

__all__ = ["VariationalAutoencoder", "GaussianMixtureVariationalAutoencoder"]
