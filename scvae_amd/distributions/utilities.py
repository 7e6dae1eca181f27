"""Likelihood / latent plug-in registry -- the surface of
``scvae/distributions/utilities.py:30-389`` that the model classes look up:

``DISTRIBUTIONS[name] = {"parameters": {pname: {"support": [lo, hi],
"activation function": f, ...}}, "class": callable(theta) -> distribution}``

with ``.log_prob(x)``, ``.mean()``, ``.variance()`` on the returned object.
The count likelihoods of the BASELINE configs (poisson, negative binomial and
their zero-inflated forms) are executed by the HIP kernels of
``libscvae_hip.so``; inside the fused training step the same kernels are driven
directly from the head pre-activations (``scvae_plan_step``), this registry is
the element-wise plug-in view of them.  The Gaussian entries describe the
latent distributions (``gaussian`` for the VAE posterior/prior, ``softplus
gaussian`` for the GMVAE) whose arithmetic is fused into the latent kernels.
"""

import ctypes
import math

import numpy
import torch

from scvae_amd import _lib
from scvae_amd.utilities import normalise_string

FLOAT32_TINY = float(numpy.finfo(numpy.float32).tiny)
_F32_MIN_HALF = float(numpy.finfo(numpy.float32).min / 2)
_F32_MAX_HALF = float(numpy.finfo(numpy.float32).max / 2)


class Activation:
    """Named activation function applied to a head's linear output."""

    def __init__(self, name, function):
        self.name = name
        self._function = function

    def __call__(self, x):
        return self._function(x)

    def __repr__(self):
        return "<activation {}>".format(self.name)


identity = Activation("identity", lambda x: x)
sigmoid = Activation("sigmoid", torch.sigmoid)
softplus = Activation("softplus", torch.nn.functional.softplus)
softmax = Activation("softmax", lambda x: torch.softmax(x, dim=-1))


def _stream(tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(tensor.device).cuda_stream)


class CountDistribution:
    """Distribution object returned by ``DISTRIBUTIONS[name]["class"](theta)``.

    ``theta`` maps parameter names to device tensors holding the activated,
    clipped parameters (as in ``va:2466-2489``).  They are mapped back to the
    kernels' pre-activation form exactly as TFP does for ``probs``:
    ``logits = log(p) - log1p(-p)`` (log-parameters are passed through)."""

    def __init__(self, name, theta):
        self.name = name
        self.kind, self.parameter_names = _lib.LIKELIHOOD_KINDS[name]
        missing = [p for p in self.parameter_names if p not in theta]
        if missing:
            raise KeyError("Missing parameters for {}: {}".format(
                name, ", ".join(missing)))
        self.theta = theta
        pre = []
        for pname in self.parameter_names:
            value = theta[pname].to(torch.float32).contiguous()
            if pname in ("p", "pi"):
                value = torch.log(value) - torch.log1p(-value)
            pre.append(value.contiguous())
        shape = torch.broadcast_shapes(*[v.shape for v in pre])
        self._pre = [v.expand(shape).contiguous() for v in pre]
        self.batch_shape = shape
        self._lib = _lib.load()

    def _run(self, x, want_log_prob, want_moments):
        pre = self._pre
        if not pre[0].is_cuda:
            raise _lib.HipLibraryError(
                "Distribution parameters must live on the GPU: the likelihood "
                "kernels have no CPU path.")
        n = pre[0].numel()
        pointers = (ctypes.c_void_p * len(pre))(*[v.data_ptr() for v in pre])
        log_prob = mean = variance = None
        t = None
        if want_log_prob:
            t = torch.as_tensor(x, dtype=torch.float32,
                                device=pre[0].device).expand(
                                    self.batch_shape).contiguous()
            log_prob = torch.empty(self.batch_shape, dtype=torch.float32,
                                   device=pre[0].device)
        if want_moments:
            mean = torch.empty(self.batch_shape, dtype=torch.float32,
                               device=pre[0].device)
            variance = torch.empty_like(mean)
        _lib.check(self._lib.scvae_likelihood_elementwise(
            self.kind,
            ctypes.c_void_p(t.data_ptr()) if t is not None else None,
            pointers,
            ctypes.c_void_p(log_prob.data_ptr()) if want_log_prob else None,
            ctypes.c_void_p(mean.data_ptr()) if want_moments else None,
            ctypes.c_void_p(variance.data_ptr()) if want_moments else None,
            n, _stream(pre[0])), "scvae_likelihood_elementwise")
        return log_prob, mean, variance

    def log_prob(self, x):
        return self._run(x, True, False)[0]

    def prob(self, x):
        return torch.exp(self.log_prob(x))

    def mean(self):
        return self._run(None, False, True)[1]

    def variance(self):
        return self._run(None, False, True)[2]

    def stddev(self):
        return torch.sqrt(self.variance())


def _count(name):
    return lambda theta: CountDistribution(name, theta)


class NormalDistribution:
    """Latent Normal(loc, scale) (log_prob / mean / stddev / sample)."""

    def __init__(self, loc, scale):
        self.loc, self.scale = loc, scale

    def mean(self):
        return self.loc

    def stddev(self):
        return self.scale

    def variance(self):
        return self.scale ** 2

    def log_prob(self, z):
        return (-0.5 * ((z - self.loc) / self.scale) ** 2
                - torch.log(self.scale) - 0.5 * math.log(2 * math.pi))

    def sample(self, sample_shape=()):
        shape = tuple(sample_shape) + tuple(self.loc.shape)
        return self.loc + self.scale * torch.randn(
            shape, device=self.loc.device, dtype=self.loc.dtype)


DISTRIBUTIONS = {
    "gaussian": {
        "parameters": {
            "mu": {
                "support": [_F32_MIN_HALF, _F32_MAX_HALF],
                "activation function": identity,
                "initial value": torch.zeros
            },
            "log_sigma": {
                "support": [-3, 3],
                "activation function": identity,
                "initial value": torch.zeros
            }
        },
        "class": lambda theta: NormalDistribution(
            loc=theta["mu"], scale=torch.exp(theta["log_sigma"]))
    },
    "softplus gaussian": {
        "parameters": {
            "mean": {
                "support": [_F32_MIN_HALF, _F32_MAX_HALF],
                "activation function": identity,
                "initial value": torch.zeros
            },
            "softplus_scale": {
                "support": [_F32_MIN_HALF, _F32_MAX_HALF],
                "activation function": identity,
                "initial value": torch.zeros
            }
        },
        "class": lambda theta: NormalDistribution(
            loc=theta["mean"],
            scale=torch.sqrt(torch.nn.functional.softplus(
                theta["softplus_scale"])))
    },
    "categorical": {
        "parameters": {
            "logits": {
                "support": [-numpy.inf, numpy.inf],
                "activation function": identity
            }
        },
        "class": lambda theta: torch.distributions.Categorical(
            logits=theta["logits"])
    },
    "poisson": {
        "parameters": {
            "log_lambda": {
                "support": [-10, 10],
                "activation function": identity
            }
        },
        "class": _count("poisson")
    },
    "zero-inflated poisson": {
        "parameters": {
            "pi": {
                "support": [0, 1],
                "activation function": sigmoid
            },
            "log_lambda": {
                "support": [-10, 10],
                "activation function": identity
            }
        },
        "class": _count("zero-inflated poisson")
    },
    "negative binomial": {
        "parameters": {
            "p": {
                "support": [0, 1],
                "activation function": sigmoid
            },
            "log_r": {
                "support": [-10, 10],
                "activation function": identity
            }
        },
        "class": _count("negative binomial")
    },
    "zero-inflated negative binomial": {
        "parameters": {
            "pi": {
                "support": [0, 1],
                "activation function": sigmoid
            },
            "p": {
                "support": [0, 1],
                "activation function": sigmoid
            },
            "log_r": {
                "support": [-10, 10],
                "activation function": identity
            }
        },
        "class": _count("zero-inflated negative binomial")
    }
}
DISTRIBUTIONS["modified gaussian"] = DISTRIBUTIONS["softplus gaussian"]


class ConstrainedPoisson:
    """``Poisson(rate=theta["lambda"] * N)`` (du:218-228): ``lambda`` is the
    softmax over the genes the model applied, ``N`` the count sums of the
    cells, broadcast against it.  (Element-wise torch arithmetic on the
    caller's device; inside a training step the softmax, the likelihood and
    its gradient are one row kernel, csrc/elementwise.hip.)"""

    def __init__(self, theta, N):
        self.rate = theta["lambda"] * N

    def log_prob(self, x):
        x = torch.as_tensor(x, dtype=self.rate.dtype, device=self.rate.device)
        return torch.xlogy(x, self.rate) - torch.lgamma(x + 1) - self.rate

    def prob(self, x):
        return torch.exp(self.log_prob(x))

    def mean(self):
        return self.rate

    def variance(self):
        return self.rate

    def stddev(self):
        return torch.sqrt(self.rate)


DISTRIBUTIONS["bernoulli"] = {
    "parameters": {
        "logits": {
            "support": [-numpy.inf, numpy.inf],
            "activation function": identity
        }
    },
    "class": _count("bernoulli")
}
DISTRIBUTIONS["constrained poisson"] = {
    "parameters": {
        "lambda": {
            "support": [0, 1],
            "activation function": lambda x: torch.softmax(x, dim=-1)
        }
    },
    "class": lambda theta, N: ConstrainedPoisson(theta, N)
}

#: reference likelihoods that this build does not provide kernels for
UNSUPPORTED_DISTRIBUTIONS = (
    "multivariate gaussian", "gaussian mixture", "log-normal",
    "exponentially_modified_gaussian", "gamma", "lomax")

LATENT_DISTRIBUTIONS = {
    "gaussian": {
        "prior": {
            "name": "gaussian",
            "parameters": {"mu": 0.0, "log_sigma": 0.0}
        },
        "posterior": {
            "name": "gaussian",
            "parameters": {}
        }
    },
    "unit-variance gaussian": {
        "prior": {
            "name": "gaussian",
            "parameters": {"mu": 0.0, "log_sigma": 0.0}
        },
        "posterior": {
            "name": "gaussian",
            "parameters": {"log_sigma": 0.0}
        }
    }
}

GAUSSIAN_MIXTURE_DISTRIBUTIONS = {
    "gaussian mixture": {
        "z prior": "softplus gaussian",
        "z posterior": "softplus gaussian"
    },
    "legacy gaussian mixture": {
        "z prior": "modified gaussian",
        "z posterior": "modified gaussian"
    }
}


def parse_distribution(distribution, model_type=None):
    """Resolve a user-supplied name against a registry
    (``scvae/distributions/utilities.py:356-389``)."""
    distribution = normalise_string(distribution)
    if model_type is None:
        kind = "reconstruction"
        distributions = DISTRIBUTIONS
    elif isinstance(model_type, str):
        kind = "latent"
        if model_type == "VAE":
            distributions = LATENT_DISTRIBUTIONS
        elif model_type == "GMVAE":
            distributions = GAUSSIAN_MIXTURE_DISTRIBUTIONS
        else:
            raise ValueError("Model type not found.")
    else:
        raise TypeError("`model_type` should be a string.")

    parsed = None
    for name in distributions:
        if normalise_string(name) == distribution:
            parsed = name
    if parsed is None:
        raise ValueError(
            "{} distribution `{}` not supported{}.".format(
                kind.capitalize(), distribution,
                " for {}".format(model_type) if model_type else ""))
    return parsed
