"""Comparison helpers of the parity tests.

BASELINE.json's bar: ELBO and per-cell reconstruction log-likelihood within
1e-4 *relative* in fp32.  ``close_elementwise`` holds every element to
``|got - want| <= rtol * |want| + atol`` (a cell whose |ll| is far below the
batch maximum is held to its own magnitude, not to the maximum's);
``close_maxnorm`` is the per-tensor bound used for gradients and post-Adam
weights, where single elements are differences of much larger terms (the bound
is relative to the tensor's largest magnitude).
"""
import numpy as np

# per-cell sum_F log p(t|z): values of order 1e2 .. 1e4; the absolute term is one
# fp32 ulp of a value of order 1e4
LL_RTOL, LL_ATOL = 1e-4, 1e-3
# scalars of the ELBO (lower_bound, reconstruction_error, kl_divergence...)
ELBO_RTOL = 1e-4


def _arrays(got, want):
    if hasattr(got, "detach"):
        got = got.detach().cpu().numpy()
    if hasattr(want, "detach"):
        want = want.detach().cpu().numpy()
    return (np.asarray(got, dtype=np.float64),
            np.asarray(want, dtype=np.float64))


def close_elementwise(got, want, rtol=LL_RTOL, atol=0.0, what=""):
    got, want = _arrays(got, want)
    assert got.shape == want.shape, "{}: shape {} vs {}".format(
        what, got.shape, want.shape)
    excess = np.abs(got - want) - (rtol * np.abs(want) + atol)
    if excess.size and excess.max() > 0:
        i = np.unravel_index(np.argmax(excess), excess.shape)
        raise AssertionError(
            "{}: element {} got {!r} want {!r} (|diff| {:.3e} > {:.1e}*|want| "
            "+ {:.1e})".format(what, i, got[i], want[i],
                               abs(got[i] - want[i]), rtol, atol))


def close_scalar(got, want, rtol=ELBO_RTOL, atol=0.0, what=""):
    close_elementwise(np.asarray(float(got)), np.asarray(float(want)),
                      rtol=rtol, atol=atol, what=what)


def close_maxnorm(got, want, rtol, what=""):
    got, want = _arrays(got, want)
    assert got.shape == want.shape, "{}: shape {} vs {}".format(
        what, got.shape, want.shape)
    scale = max(np.abs(want).max(), 1e-30) if want.size else 1.0
    err = np.abs(got - want).max() / scale if want.size else 0.0
    assert err <= rtol, "{}: max err {:.3e} of scale {:.3e}".format(
        what, err, scale)
