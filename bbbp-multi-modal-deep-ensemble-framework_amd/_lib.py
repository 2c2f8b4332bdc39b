"""ctypes binding of libbbbp_hip.so (the C ABI declared in include/bbbp_hip.h).

The product path has NO fallback: if the library is missing, or a call fails, a RuntimeError is
raised.  Nothing here imports torch; device pointers arrive as integers.
"""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_uint8, c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BBBP_LIB", os.path.join(_HERE, "libbbbp_hip.so"))    # BBBP_LIB: A/B another build
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "bbbp_hip.h")

_lib = None


class MixedDesc(Structure):
    """bbbp_mixed_desc (include/bbbp_hip.h)."""
    _fields_ = [("batch", c_int), ("fingerprint_size", c_int), ("nhead", c_int), ("num_layers", c_int),
                ("dim_feedforward", c_int), ("training", c_int), ("dropout_p", c_float), ("seed", c_uint64),
                ("need_input_grad", c_int), ("fusion", c_int), ("inference", c_int),
                ("world", c_int), ("rank", c_int), ("collective", c_void_p), ("collective_ctx", c_void_p)]


# bbbp_collective_fn (include/bbbp_hip.h): ctx, op, what, layer, send_off, recv_off, count, stream
CollectiveFn = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, c_int, c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, c_void_p)


class GemmDesc(Structure):
    """bbbp_gemm_desc (include/bbbp_hip.h)."""
    _fields_ = [("transA", c_int), ("transB", c_int), ("M", c_int), ("N", c_int), ("K", c_int), ("alpha", c_float),
                ("A", c_void_p), ("lda", c_int), ("B", c_void_p), ("ldb", c_int), ("C", c_void_p), ("ldc", c_int),
                ("bias", c_void_p), ("residual", c_void_p), ("ldr", c_int), ("act", c_int),
                ("gate", c_void_p), ("ldg", c_int), ("gate_scale", c_float), ("batch", c_int),
                ("strideA", c_long), ("strideB", c_long), ("strideC", c_long), ("strideR", c_long), ("strideG", c_long),
                ("gate_after_residual", c_int), ("asum", c_void_p), ("drop_p", c_float), ("drop_seed", c_uint64)]


class MlpModel(Structure):
    """bbbp_mlp_model (include/bbbp_hip.h)."""
    _fields_ = [("n_layers", c_int), ("units", c_int * 5), ("activation", c_int), ("batch_size", c_int), ("n_train", c_int),
                ("n_iter_no_change", c_int), ("max_iter", c_int),
                ("lr_init", ctypes.c_double), ("alpha", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double),
                ("eps", ctypes.c_double), ("tol", ctypes.c_double),
                ("params", c_void_p), ("adam_m", c_void_p), ("adam_v", c_void_p), ("grads", c_void_p),
                ("act", c_void_p), ("delta", c_void_p), ("order", c_void_p), ("loss_curve", c_void_p),
                ("t", c_long), ("best_loss", ctypes.c_double), ("no_improve", c_int), ("n_iter", c_int), ("done", c_int)]


_FP = c_void_p          # device float*
_PP = POINTER(c_void_p)  # host array of device pointers

_SIGNATURES = {
    "bbbp_abi_version": (c_int, []),
    "bbbp_last_error": (c_char_p, []),
    "bbbp_gemm_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "bbbp_gemm_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, _FP, c_int, _FP, c_int, _FP, c_int,
                              _FP, _FP, c_int, c_int, c_int, c_long, c_long, c_long, c_long, c_void_p, c_size_t]),
    "bbbp_gemm_folds_asum": (c_int, [c_int, c_int, c_int, c_int]),
    "bbbp_gemm_f32_grouped": (c_int, [c_void_p, POINTER(GemmDesc), c_int, c_void_p, c_size_t]),
    "bbbp_mixed_backward_wait_bucket": (c_int, [c_void_p, c_int]),
    "bbbp_mixed_backward_wait_released": (c_int, [c_void_p, c_int]),
    "bbbp_set_release_events": (c_int, [c_int]),
    "bbbp_mixed_bucket_param": (c_int, [POINTER(MixedDesc), c_int]),
    "bbbp_mixed_bucket_range": (c_int, [POINTER(MixedDesc), c_int, POINTER(c_int), POINTER(c_int)]),
    "bbbp_mixed_debug_ffn_gate": (c_int, [c_void_p, POINTER(MixedDesc), c_void_p, c_int, c_void_p]),
    "bbbp_mixed_debug_pool_mask": (c_int, [c_void_p, POINTER(MixedDesc), c_void_p, c_int, c_void_p]),
    "bbbp_set_graphs": (c_int, [c_int]),
    "bbbp_set_fused_head_bwd": (c_int, [c_int]),
    "bbbp_set_fused_encoder": (c_int, [c_int]),
    "bbbp_set_fold_outproj": (c_int, [c_int]),
    "bbbp_set_flash_attention": (c_int, [c_int]),
    "bbbp_set_gemm_split_bf16": (c_int, [c_int]),
    "bbbp_set_gemm_fold_reduce": (c_int, [c_int]),
    "bbbp_gemm_split_bf16_phases": (c_int, [POINTER(c_uint64)]),
    "bbbp_gbt_predict": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float,
                         c_void_p, c_void_p]),
    "bbbp_oblivious_predict": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                               c_double, c_double, c_void_p, c_void_p]),
    "bbbp_linear_layernorm_supported": (c_int, [c_int, c_int, c_int]),
    "bbbp_linear_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_uint64]),
    "bbbp_layernorm_linear_supported": (c_int, [c_int, c_int, c_int]),
    "bbbp_layernorm_linear_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                                  c_uint64, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int]),
    "bbbp_forest_groups": (c_int, [c_int]),
    "bbbp_forest_predict": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_void_p]),
    "bbbp_mlp_train_epochs": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int]),
    "bbbp_mlp_predict_proba": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "bbbp_graph_stats": (c_int, [POINTER(c_long), POINTER(c_long)]),
    "bbbp_conv_last_clock": (c_int, [POINTER(c_uint64), POINTER(c_uint64)]),
    "bbbp_set_conv_winograd": (c_int, [c_int]),
    "bbbp_get_conv_winograd": (c_int, []),
    "bbbp_conv_winograd_phases": (c_int, [POINTER(c_uint64)]),
    "bbbp_conv_b3_phases": (c_int, [POINTER(c_uint64)]),
    "bbbp_conv3x3_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "bbbp_conv3x3_relu_pool_fwd": (c_int, [c_void_p, _FP, _FP, _FP, _FP, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                           c_void_p, c_size_t]),
    "bbbp_conv3x3_relu_pool_bwd_data": (c_int, [c_void_p, _FP, c_void_p, _FP, _FP, c_int, c_int, c_int, c_int, c_int,
                                                c_void_p, c_size_t]),
    "bbbp_conv3x3_relu_pool_bwd_weight": (c_int, [c_void_p, _FP, _FP, c_void_p, _FP, _FP, c_int, c_int, c_int, c_int,
                                                  c_int, c_void_p, c_size_t]),
    "bbbp_layernorm_fwd": (c_int, [c_void_p, _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_float, c_float, c_uint64]),
    "bbbp_layernorm_bwd": (c_int, [c_void_p, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_float, c_uint64]),
    "bbbp_softmax_fwd": (c_int, [c_void_p, _FP, _FP, c_long, c_int, c_float, c_uint64]),
    "bbbp_softmax_bwd": (c_int, [c_void_p, _FP, _FP, c_long, c_int, c_float, c_uint64]),
    "bbbp_dropout": (c_int, [c_void_p, _FP, _FP, c_long, c_float, c_uint64]),
    "bbbp_batchnorm1d_fwd": (c_int, [c_void_p, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_float, c_float, c_int]),
    "bbbp_batchnorm1d_bwd": (c_int, [c_void_p, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int]),
    "bbbp_batchnorm1d_bwd_relu": (c_int, [c_void_p, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_int]),
    "bbbp_column_moments": (c_int, [c_void_p, _FP, _FP, _FP, _FP, c_int, c_int]),
    "bbbp_batchnorm1d_bwd_apply": (c_int, [c_void_p, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP, c_int, c_int, c_long]),
    "bbbp_bias_act_bwd": (c_int, [c_void_p, _FP, c_int, _FP, c_int, _FP, c_int, c_int, c_int, c_float]),
    "bbbp_fusion_combine_fwd": (c_int, [c_void_p, _FP, _FP, _PP, _PP, _FP, _FP, c_int, c_int, c_int, c_int]),
    "bbbp_fusion_combine_bwd": (c_int, [c_void_p, _FP, _FP, _FP, _FP, _PP, _FP, _FP, _FP, c_int, c_int, c_int, c_int]),
    "bbbp_mse": (c_int, [c_void_p, _FP, _FP, _FP, _FP, c_int, c_float]),
    "bbbp_adamw_step": (c_int, [c_void_p, _FP, _FP, _FP, _FP, c_long, c_double, c_double, c_double, c_double, c_double, c_int,
                                c_double]),
    "bbbp_adamw_step_multi": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_int, c_double, c_double, c_double, c_double, c_double, c_int, c_double, c_void_p]),
    "bbbp_adamw_hyper_store": (c_int, [c_void_p, c_void_p, c_double, c_double, c_double, c_double, c_double, c_int, c_double]),
    "bbbp_set_seed_base": (c_int, [c_void_p]),
    "bbbp_mlp_profile": (c_int, [c_int, c_void_p]),
    "bbbp_mlp_profile_groups": (c_int, [c_void_p, c_int]),
    "bbbp_set_conv_wgrad_beside_encoder": (c_int, [c_int]),
    "bbbp_set_conv2_fwd_pipe": (c_int, [c_int]),
    "bbbp_adamw_step_deferred": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_long, c_long, c_double, c_double, c_double, c_double, c_double, c_int, c_double]),
    "bbbp_param_sync": (c_int, [c_void_p]),
    "bbbp_param_stream": (c_void_p, []),
    "bbbp_scale": (c_int, [c_void_p, _FP, c_long, c_float]),
    "bbbp_resize_bilinear_totensor": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, _FP, c_void_p, c_void_p, c_int, c_void_p,
                                              c_void_p, c_int, c_int, c_int, c_int, c_int, c_int]),
    "bbbp_standardize_chunk": (c_int, [c_void_p, c_void_p, _FP, _FP, _FP, c_void_p, c_void_p, c_int, c_int, c_int]),
    "bbbp_set_partition": (c_int, [c_int, c_size_t]),
    "bbbp_set_comm_cus": (c_int, [c_int]),
    "bbbp_set_ln_absorb": (c_int, [c_int]),
    "bbbp_set_overlap": (c_int, [c_int]),
    "bbbp_profile_enable": (c_int, [c_int]),
    "bbbp_profile_select": (c_int, [ctypes.c_uint]),
    "bbbp_profile_num_sections": (c_int, []),
    "bbbp_profile_section_name": (c_char_p, [c_int]),
    "bbbp_profile_collect": (c_int, [POINTER(c_float), POINTER(c_int)]),
    "bbbp_profile_timeline": (c_int, [POINTER(c_int), POINTER(c_float), POINTER(c_float), c_int]),
    "bbbp_mixed_num_params": (c_int, [POINTER(MixedDesc)]),
    "bbbp_mixed_workspace_bytes": (c_size_t, [POINTER(MixedDesc)]),
    "bbbp_mixed_forward": (c_int, [c_void_p, POINTER(MixedDesc), _PP, _PP, _FP, _FP, _FP, c_void_p, c_size_t]),
    "bbbp_mixed_backward": (c_int, [c_void_p, POINTER(MixedDesc), _PP, _PP, _FP, _FP, _FP, c_void_p, c_size_t]),
}


def declared_symbols():
    """Every function name declared in include/bbbp_hip.h (used by the ABI test)."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bbbp_[a-z0-9_]+)\s*\(", text)))


def lib() -> ctypes.CDLL:
    """Load the HIP library, or fail loudly (there is no CPU fallback on the product path)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The BBBP hot path has no CPU fallback.")
    l = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(l, name)       # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = l
    return l


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().bbbp_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def ptr_array(ptrs):
    arr = (c_void_p * len(ptrs))(*ptrs)
    return arr
