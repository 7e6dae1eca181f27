"""ctypes binding of ``libscvae_hip.so`` (C ABI declared in ``include/scvae_hip.h``).

The library is the only compute path of this package: loading fails loudly
when it has not been built (``python __graft_entry__.py`` or
``scvae_amd/csrc/build.sh``), and there is no CPU fallback.
"""

import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int32, c_int64,
                    c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
#: (SCVAE_HIP_LIBRARY: an A/B build of the same library, tools/ab_variants.sh)
LIBRARY_PATH = os.environ.get("SCVAE_HIP_LIBRARY") or os.path.join(
    _HERE, "csrc", "libscvae_hip.so")

MAX_HIDDEN = 8
NAME_MAX = 96

POISSON, NB, ZIP, ZINB, CONSTRAINED_POISSON, BERNOULLI = 0, 1, 2, 3, 4, 5
#: flags of scvae_decoder_fused's ``train`` argument: the arithmetic of that call
HEADS_FP32, HEADS_BF16X9, HEADS_DD_ATOMICS, HEADS_BF16X6 = 0x100, 0x200, 0x400, 0x800
HEAD_ARITH_FLAGS = {"fp32": HEADS_FP32, "bf16x9": HEADS_BF16X9, "bf16x6": HEADS_BF16X6}
MODEL_VAE, MODEL_GMVAE = 0, 1

#: registry name -> (kind, head parameter names in registry order)
LIKELIHOOD_KINDS = {
    "poisson": (POISSON, ("log_lambda",)),
    "negative binomial": (NB, ("p", "log_r")),
    "zero-inflated poisson": (ZIP, ("pi", "log_lambda")),
    "zero-inflated negative binomial": (ZINB, ("pi", "p", "log_r")),
    "constrained poisson": (CONSTRAINED_POISSON, ("lambda",)),
    "bernoulli": (BERNOULLI, ("logits",)),
}


class ModelConfig(Structure):
    _fields_ = [
        ("model_type", c_int32),
        ("feature_size", c_int32),
        ("latent_size", c_int32),
        ("n_hidden", c_int32),
        ("hidden", c_int32 * MAX_HIDDEN),
        ("likelihood", c_int32),
        ("batch_norm", c_int32),
        ("n_clusters", c_int32),
        ("kl_weight", c_float),
        ("free_nats_proportion", c_float),
        ("k_max", c_int32),
        ("prior_mode", c_int32),
        ("linear_factor", c_int32),
        ("decoder_extra", c_int32),
        ("latent_mode", c_int32),
        ("dropout_keep", c_float * 4),
    ]


class StepArgs(Structure):
    _fields_ = [
        ("x", c_void_p),
        ("t", c_void_p),
        ("row_const", c_void_p),
        ("eps", c_void_p),
        ("cells", c_int64),
        ("global_cells", c_int64),
        ("n_iw", c_int32),
        ("n_mc", c_int32),
        ("training", c_int32),
        ("deterministic_z", c_int32),
        ("warm_up_weight", c_float),
        ("scalars", c_void_p),
        ("log_p_x_given_z", c_void_p),
        ("q_z_mean", c_void_p),
        ("kl_neurons", c_void_p),
        ("q_y_logits", c_void_p),
        ("p_x_mean", c_void_p),
        ("p_x_stddev", c_void_p),
        ("stddev_of_p_x_given_z_mean", c_void_p),
        ("cluster_stats", c_void_p),
        ("decoder_extra", c_void_p),
        ("dropout_seed", c_uint64),
        ("count_sum", c_void_p),
        ("row_offset", c_int64),
        ("x_counts", c_int32),
        ("counts_u16", c_void_p),
        ("counts_ld", c_int64),
        ("count_tiles", c_void_p),
        ("side", c_void_p),
    ]


class CountTilesStruct(Structure):
    """``scvae_count_tiles``: a minibatch as tile-indexed non-zeros."""
    _fields_ = [
        ("entries", c_void_p),
        ("tile_ptr", c_void_p),
        ("block_ptr", c_void_p),
        ("capacity", c_int64),
        ("status", c_void_p),
    ]


class SideWork(Structure):
    """``scvae_side_work``: the optimiser update of a training step and the
    fetch / noise of the next minibatch, carried by the step."""
    _fields_ = [
        ("adam_m", c_void_p),
        ("adam_v", c_void_p),
        ("adam_grad_scale", c_float),
        ("adam_lr_t", c_float),
        ("adam_beta1", c_float),
        ("adam_beta2", c_float),
        ("adam_epsilon", c_float),
        ("fetch_as_u16", c_int32),
        ("fetch_indptr", c_void_p),
        ("fetch_indices", c_void_p),
        ("fetch_values", c_void_p),
        ("fetch_rows", c_void_p),
        ("fetch_n", c_int64),
        ("fetch_features", c_int64),
        ("fetch_out", c_void_p),
        ("fetch_ld", c_int64),
        ("fetch_row_values", c_void_p),
        ("fetch_row_values_out", c_void_p),
        ("fetch_tiles", c_void_p),
        ("noise_out", c_void_p),
        ("noise_blocks", c_int64),
        ("noise_block_rows", c_int64),
        ("noise_cols", c_int64),
        ("noise_block_stride", c_int64),
        ("noise_row_offset", c_int64),
        ("noise_seed", c_uint64),
        ("noise_stream_id", c_uint64),
    ]


SYNC_FN = ctypes.CFUNCTYPE(c_int32, c_void_p, c_void_p, c_int64, c_int32,
                           c_int64)

#: every symbol declared in include/scvae_hip.h: name -> (restype, argtypes)
SIGNATURES = {
    "scvae_last_error": (c_char_p, []),
    "scvae_version": (c_int32, []),
    "scvae_plan_create": (c_int32, [POINTER(ModelConfig), POINTER(c_void_p)]),
    "scvae_plan_destroy": (None, [c_void_p]),
    "scvae_plan_param_count": (c_int64, [c_void_p]),
    "scvae_plan_param_floats": (c_int64, [c_void_p]),
    "scvae_plan_moving_floats": (c_int64, [c_void_p]),
    "scvae_plan_param_info": (c_int32, [
        c_void_p, c_int64, c_char_p, POINTER(c_int64), POINTER(c_int64),
        POINTER(c_int64)]),
    "scvae_plan_moving_count": (c_int64, [c_void_p]),
    "scvae_plan_moving_info": (c_int32, [
        c_void_p, c_int64, c_char_p, POINTER(c_int64), POINTER(c_int64)]),
    "scvae_plan_workspace_bytes": (c_int64, [c_void_p, c_int64, c_int64]),
    "scvae_plan_bind": (c_int32, [
        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
        c_int64]),
    "scvae_plan_set_sync": (c_int32, [c_void_p, SYNC_FN, c_void_p]),
    "scvae_plan_probe_stages": (c_int32, [c_void_p, c_int32]),
    "scvae_plan_probe_stages_us": (c_int32, [c_void_p, POINTER(c_float), c_int32]),
    "scvae_plan_set_fused": (c_int32, [c_void_p, c_int32]),
    "scvae_plan_set_head_arith": (c_int32, [c_void_p, c_int32]),
    "scvae_plan_head_arith": (c_int32, [c_void_p]),
    "scvae_plan_fused_categorised": (c_int32, [c_void_p]),
    "scvae_plan_set_tile_resident": (c_int32, [c_void_p, c_int32]),
    "scvae_plan_uses_tile_resident": (c_int32, [c_void_p, c_int64, c_int32]),
    "scvae_plan_set_dd_atomics": (c_int32, [c_void_p, c_int32]),
    "scvae_plan_dd_atomics": (c_int32, [c_void_p]),
    "scvae_default_dd_atomics": (c_int32, []),
    "scvae_plan_set_count_gemm": (c_int32, [c_void_p, c_int32]),
    "scvae_plan_set_bn_one_launch": (c_int32, [c_void_p, c_int32]),
    "scvae_plan_set_mid_chain": (c_int32, [c_void_p, c_int32]),
    "scvae_plan_set_tile_chain": (c_int32, [c_void_p, c_int32]),
    "scvae_plan_uses_tile_chain": (c_int32, [c_void_p, c_int64, c_int32]),
    "scvae_plan_probe_heads": (c_int32, [c_void_p, c_int32]),
    "scvae_plan_probe_heads_ms": (c_int32, [c_void_p, c_void_p, c_int32]),
    "scvae_count_gemm": (c_int32, [
        c_int32, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64,
        c_int64, c_void_p, c_int32, c_void_p, c_int64, c_void_p, c_int64,
        c_void_p]),
    "scvae_plan_accepts_counts_u16": (c_int32, [c_void_p, c_int64, c_int32]),
    "scvae_count_gemm_u16": (c_int32, [
        c_int32, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64,
        c_int64, c_void_p, c_int32, c_void_p, c_int64, c_void_p, c_int64,
        c_void_p]),
    "scvae_csr_minibatch": (c_int32, [
        c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
        c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "scvae_csr_densify_u16": (c_int32, [
        c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
        c_int64, c_void_p]),
    "scvae_count_gemm_workspace_bytes": (c_int64, [c_int32, c_int64, c_int64,
                                                   c_int64]),
    "scvae_check_counts": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p]),
    "scvae_plan_step": (c_int32, [c_void_p, POINTER(StepArgs), c_void_p]),
    "scvae_adam_clip_step": (c_int32, [
        c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float,
        c_float, c_float, c_float, c_void_p]),
    "scvae_gemm": (c_int32, [
        c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
        c_int64, c_int64, c_int64, c_int64, c_int64, c_int32, c_int32,
        c_void_p, c_int64, c_void_p]),
    "scvae_gemm_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "scvae_loglik_fwd": (c_int32, [
        c_int32, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_int64,
        c_int64, c_int64, c_void_p]),
    "scvae_loglik_bwd": (c_int32, [
        c_int32, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p,
        c_int64, c_int64, c_int64, c_void_p]),
    "scvae_decoder_fused_workspace_bytes": (c_int64, [c_int64, c_int64,
                                                      c_int64]),
    "scvae_decoder_fused_variant": (c_int32, [c_int32, c_int64]),
    "scvae_default_head_arith": (c_int32, []),
    "scvae_decoder_train_kernel": (c_int32, [c_int32, c_int64, c_int32]),
    "scvae_decoder_train_kernel_name": (c_int32, [c_int32, c_int64, c_int64, c_int32, c_int32,
                                                   c_char_p, c_int64]),
    "scvae_plan_prior_offset": (c_int64, [c_void_p]),
    "scvae_plan_decode": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p,
                                    c_void_p]),
    "scvae_decoder_fused": (c_int32, [
        c_int32, c_int32, c_void_p, c_int64, c_int64, POINTER(c_void_p),
        POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_int64,
        c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
        c_void_p]),
    "scvae_decoder_fused_u16": (c_int32, [
        c_int32, c_int32, c_void_p, c_int64, c_int64, POINTER(c_void_p),
        POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), c_int64,
        c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
        c_void_p, c_void_p]),
    "scvae_likelihood_elementwise": (c_int32, [
        c_int32, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p,
        c_int64, c_void_p]),
    "scvae_gauss_latent_fwd": (c_int32, [
        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
        c_int64, c_int64, c_int32, c_void_p]),
    "scvae_csr_densify": (c_int32, [
        c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p,
        c_void_p]),
    "scvae_count_tiles_padded": (c_int64, [c_int64]),
    "scvae_csr_row_entries": (c_int32, [
        c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "scvae_csr_count_tiles": (c_int32, [
        c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
        POINTER(CountTilesStruct), c_void_p]),
    "scvae_count_gemm_tiles": (c_int32, [
        c_int32, POINTER(CountTilesStruct), c_void_p, c_int64, c_int64,
        c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int32, c_void_p,
        c_int64, c_void_p, c_int64, c_void_p]),
    "scvae_csr_row_lgamma1p": (c_int32, [
        c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "scvae_gather_rows": (c_int32, [
        c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "scvae_philox_normal": (c_int32, [
        c_void_p, c_int64, c_int64, c_int64, c_uint64, c_uint64, c_void_p]),
    "scvae_philox_normal_blocks": (c_int32, [
        c_void_p, c_int64, c_int64, c_int64, c_int64, c_int64, c_uint64,
        c_uint64, c_void_p]),
    "scvae_dropout_apply": (c_int32, [
        c_void_p, c_void_p, c_int64, c_int64, c_float, c_uint64, c_int32,
        c_int32, c_void_p]),
    "scvae_bn_workspace_floats": (c_int64, [c_int64]),
    "scvae_bn_stats": (c_int32, [
        c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p,
        c_void_p]),
    "scvae_bn_apply_relu_fwd": (c_int32, [
        c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
        c_int64, c_int64, c_int32, c_void_p]),
    "scvae_bn_apply_relu_bwd": (c_int32, [
        c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
        c_void_p, c_int64, c_int64, c_int32, c_void_p, c_int64, c_void_p,
        c_void_p, c_void_p]),
    "scvae_softplus_gaussian_logprob_pair_fwd": (c_int32, [
        c_void_p] * 10 + [c_int64] * 4 + [c_void_p]),
    "scvae_softplus_gaussian_logprob_pair_bwd": (c_int32, [
        c_void_p] * 12 + [c_int64] * 4 + [c_void_p]),
    "scvae_categorical_entropy_kl_fwd": (c_int32, [
        c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "scvae_categorical_entropy_kl_bwd": (c_int32, [
        c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int64,
        c_void_p, c_void_p]),
    "scvae_iw_logmeanexp": (c_int32, [
        c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int64, c_float,
        c_float, c_void_p, c_void_p, c_void_p]),
    "scvae_pxmean_stats": (c_int32, [
        c_int32, POINTER(c_void_p), c_int64, c_int64, c_int64, c_void_p,
        c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "scvae_bn_merge": (c_int32, [
        c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBRARY_PATH):
        raise HipLibraryError(
            "{} has not been built; run `python __graft_entry__.py` or "
            "scvae_amd/csrc/build.sh (there is no CPU fallback).".format(
                LIBRARY_PATH))
    lib = ctypes.CDLL(LIBRARY_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        message = load().scvae_last_error().decode("utf-8", "replace")
        raise HipLibraryError("{} failed ({}): {}".format(what, rc, message))
