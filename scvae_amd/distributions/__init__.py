from scvae_amd.distributions.utilities import (
    DISTRIBUTIONS, GAUSSIAN_MIXTURE_DISTRIBUTIONS, LATENT_DISTRIBUTIONS,
    CountDistribution, parse_distribution)

__all__ = [
    "DISTRIBUTIONS", "LATENT_DISTRIBUTIONS", "GAUSSIAN_MIXTURE_DISTRIBUTIONS",
    "CountDistribution", "parse_distribution",
]
