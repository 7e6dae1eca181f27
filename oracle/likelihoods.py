"""Oracle (TEST INFRASTRUCTURE): count likelihoods of the scVAE decoder.

PARITY UNPINNED (see ``oracle/__init__.py``).

Each likelihood takes the *pre-activations* of its ``X_TILDE/<PARAM>`` heads
(reference: ``scvae/models/variational_autoencoder.py:2466-2489``), applies
the activation function and the support clip of the ``DISTRIBUTIONS``
registry (``scvae/distributions/utilities.py:206-216`` poisson, ``:247-264``
zero-inflated poisson, ``:266-281`` negative binomial, ``:283-305``
zero-inflated negative binomial) and evaluates ``log_prob``, ``mean`` and
``variance`` the way TFP 0.7 does for the distribution class named there:

* ``tfp.distributions.Poisson(rate=exp(log_lambda))``:
  ``x*log(rate) - lgamma(x+1) - rate``; mean = variance = rate.
* ``tfp.distributions.NegativeBinomial(total_count=exp(log_r), probs=p)``:
  ``logits = log(p) - log1p(-p)``;
  ``r*log_sigmoid(-logits) + x*log_sigmoid(logits)
  + lgamma(r+x) - lgamma(1+x) - lgamma(r)``;
  mean ``r*exp(logits)``, variance ``mean/sigmoid(-logits)``.
* ``ZeroInflated(dist, pi)`` (``scvae/distributions/zero_inflated.py:180-199``).

The arithmetic is written in the algebraically equal *stable* form
(``log p = -softplus(-a)``, ``log(1-p) = -softplus(a)``, ``logaddexp`` for the
zero branch).  The reference's float32 graph differs from this only where it
is itself non-finite: ``sigmoid(a) == 1.0f`` for ``a > 16.6`` makes its
``log1p(-p)`` ``-inf`` (the upper clip ``1 - float32.tiny == 1`` is a no-op,
SURVEY.md appendix A.10).

All functions are dtype-generic torch code: float64 for parity tests and
fixtures, float32 for the ``cpu_baseline`` leg of ``bench.py``.
"""

import math

import torch
import torch.nn.functional as F

FLOAT32_TINY = 1.1754943508222875e-38
#: logit below which sigmoid(a) < float32.tiny, i.e. the lower clip is active
LOGIT_OF_TINY = math.log(FLOAT32_TINY)  # -87.33654475...

#: head order as in the DISTRIBUTIONS registry (dict iteration order)
LIKELIHOOD_PARAMETERS = {
    "poisson": ("log_lambda",),
    "negative binomial": ("p", "log_r"),
    "zero-inflated poisson": ("pi", "log_lambda"),
    "zero-inflated negative binomial": ("pi", "p", "log_r"),
    # du:218-228: activation softmax over the genes, rate = lambda * N with N
    # the count sum of the cell (va:2400-2405, 2490-2496)
    "constrained poisson": ("lambda",),
    # du:194-204: tfp.distributions.Bernoulli(logits=...), binarised targets
    "bernoulli": ("logits",),
}


#: the count likelihoods of the fused decoder kernels (their data term
#: -lgamma(1 + t) is added by the caller once per cell)
ELEMENTWISE_LIKELIHOODS = (
    "poisson", "negative binomial", "zero-inflated poisson",
    "zero-inflated negative binomial")


def _clip_log(a):
    """identity activation clipped to the support [-10, 10]."""
    return torch.clamp(a, -10.0, 10.0)


def _log_sigmoid_pair(a):
    """(log p, log(1-p)) for p = clip(sigmoid(a), tiny, 1)."""
    a = torch.clamp(a, min=LOGIT_OF_TINY)
    return F.logsigmoid(a), F.logsigmoid(-a), a


def poisson_log_prob(t, log_lambda_pre):
    ll = _clip_log(log_lambda_pre)
    return t * ll - torch.lgamma(t + 1.0) - torch.exp(ll)


def negative_binomial_log_prob(t, p_pre, log_r_pre):
    log_p, log_1mp, _ = _log_sigmoid_pair(p_pre)
    r = torch.exp(_clip_log(log_r_pre))
    return (r * log_1mp + t * log_p
            + torch.lgamma(r + t) - torch.lgamma(1.0 + t) - torch.lgamma(r))


def _zero_inflate(t, pi_pre, base_log_prob, base_log_prob_at_zero):
    log_pi, log_1mpi, _ = _log_sigmoid_pair(pi_pre)
    y_0 = torch.logaddexp(log_pi, log_1mpi + base_log_prob_at_zero)
    y_1 = log_1mpi + base_log_prob
    return torch.where(t > 0, y_1, y_0)


def zero_inflated_poisson_log_prob(t, pi_pre, log_lambda_pre):
    # zero_inflated.py:197 evaluates dist.prob(x) at the observed x and only
    # *uses* it where x <= 0, i.e. at x == 0 for count data.
    base = poisson_log_prob(t, log_lambda_pre)
    return _zero_inflate(t, pi_pre, base, base)


def zero_inflated_negative_binomial_log_prob(t, pi_pre, p_pre, log_r_pre):
    base = negative_binomial_log_prob(t, p_pre, log_r_pre)
    return _zero_inflate(t, pi_pre, base, base)


def constrained_poisson_rate(lambda_pre, count_sum):
    """``clip(softmax(pre), tiny, 1 - tiny) * N``; ``count_sum``: [rows, 1]."""
    lam = torch.softmax(lambda_pre, dim=-1)
    lam = torch.clamp(lam, min=FLOAT32_TINY, max=1.0)
    return lam * count_sum


def constrained_poisson_log_prob(t, lambda_pre, count_sum):
    """``tfp.distributions.Poisson(rate=lambda * N).log_prob(t)``."""
    rate = constrained_poisson_rate(lambda_pre, count_sum)
    return torch.xlogy(t, rate) - torch.lgamma(t + 1.0) - rate


def log_prob(name, t, pre, count_sum=None):
    """``pre``: tuple of head pre-activations in registry order."""
    if name == "constrained poisson":
        return constrained_poisson_log_prob(t, pre[0], count_sum)
    if name == "bernoulli":   # -sigmoid_cross_entropy_with_logits
        return t * F.logsigmoid(pre[0]) + (1.0 - t) * F.logsigmoid(-pre[0])
    if name == "poisson":
        return poisson_log_prob(t, *pre)
    if name == "negative binomial":
        return negative_binomial_log_prob(t, *pre)
    if name == "zero-inflated poisson":
        return zero_inflated_poisson_log_prob(t, *pre)
    if name == "zero-inflated negative binomial":
        return zero_inflated_negative_binomial_log_prob(t, *pre)
    raise ValueError(name)


def categorised_log_prob(name, t, pre, logits, k_max):
    """``Categorised(dist, cat)`` of distributions/categorised.py:255-263:
    counts 0..k_max-1 are classes of a categorical, class k_max stands for
    "k_max or more" and hands the excess ``t - k_max`` to the count
    distribution.  ``logits``: [..., F, k_max + 1]."""
    log_pi = torch.log_softmax(logits, dim=-1)
    classes = torch.clamp(t, 0, k_max).to(torch.int64)
    cat = torch.gather(log_pi, -1, classes.unsqueeze(-1)).squeeze(-1)
    tail = t >= k_max
    # evaluated only where it counts (tf.where also masks the other branch)
    excess = torch.where(tail, t - k_max, torch.zeros_like(t))
    return cat + torch.where(tail, log_prob(name, excess, pre),
                             torch.zeros_like(t))


def categorised_mean_variance(name, pre, logits, k_max):
    """categorised.py:210-253."""
    pi = torch.softmax(logits, dim=-1)
    ks = torch.arange(k_max, dtype=logits.dtype)
    cat_mean = (pi[..., :k_max] * ks).sum(dim=-1)
    cat_second = (pi[..., :k_max] * ks * ks).sum(dim=-1)
    m, v = mean_variance(name, pre)
    tail = pi[..., k_max]
    mean = cat_mean + tail * (m + k_max)
    second = cat_second + tail * (2 * k_max * m + v + m * m + k_max ** 2)
    return mean, second - mean * mean


def mean_variance(name, pre, count_sum=None):
    """E[x|z], Var[x|z] of the decoder distribution (TFP semantics)."""
    if name == "constrained poisson":
        rate = constrained_poisson_rate(pre[0], count_sum)
        return rate, rate
    if name == "bernoulli":
        p = torch.sigmoid(pre[0])
        return p, p * (1.0 - p)
    if name == "poisson":
        lam = torch.exp(_clip_log(pre[0]))
        return lam, lam
    if name == "negative binomial":
        _, log_1mp, a = _log_sigmoid_pair(pre[0])
        r = torch.exp(_clip_log(pre[1]))
        mean = r * torch.exp(a)
        return mean, mean * torch.exp(-log_1mp)
    if name == "zero-inflated poisson":
        log_pi, log_1mpi, _ = _log_sigmoid_pair(pre[0])
        m, v = mean_variance("poisson", pre[1:])
    elif name == "zero-inflated negative binomial":
        log_pi, log_1mpi, _ = _log_sigmoid_pair(pre[0])
        m, v = mean_variance("negative binomial", pre[1:])
    else:
        raise ValueError(name)
    one_minus_pi = torch.exp(log_1mpi)
    zi_mean = one_minus_pi * m
    zi_var = one_minus_pi * (v + m * m) - zi_mean * zi_mean
    return zi_mean, zi_var
