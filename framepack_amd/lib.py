"""ctypes binding of libframepack_hip.so (C ABI: include/framepack_hip.h).

The library is built in-tree by `make -C framepack_amd/csrc` (or `__graft_entry__.build()`).
There is NO fallback: if the shared object is missing or a symbol is absent, loading raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# FRAMEPACK_HIP_LIB: a differently built copy of the same library (timing-only probe builds, tools/attn_skip_probe.sh)
LIB_PATH = os.environ.get("FRAMEPACK_HIP_LIB") or os.path.join(_HERE, "libframepack_hip.so")

P, I, L, F = c_void_p, c_int, c_int64, c_float

# name -> argtypes, exactly the declarations of include/framepack_hip.h
SIGNATURES = {
    "fp_abi_version": [],
    "fp_attn_workspace_bytes": [],
    "fp_attn_fwd_bf16": [P, P, P, P, L, L, L, L, P, I, I, I, I, I, F, P, L, P],
    "fp_attn_static_scratch_bytes": [],
    "fp_attn_fwd_bf16_static": [P, P, P, P, L, L, L, L, P, I, I, I, I, I, F, P],
    "fp_attn_fp8_workspace_bytes": [I, I],
    "fp_attn_fwd_fp8": [P, P, P, P, L, L, L, L, I, I, I, F, P, L, P],
    "fp_attn_fp8_varlen_workspace_bytes": [I, I, I, I],
    "fp_attn_fwd_fp8_varlen": [P, P, P, P, L, L, L, L, P, I, I, I, I, I, F, P, L, P],
    "fp_qk_rmsnorm_rope_bf16": [P, P, L, I, I, I, P, P, P, P, P, I, F, P],
    "fp_qk_rmsnorm_rope_pack_bf16": [P, P, P, L, I, I, I, P, P, P, P, P, I, F, P, I, I, P],
    "fp_layernorm_modulate_bf16": [P, L, P, P, P, L, I, I, F, P],
    "fp_layernorm_affine_bf16": [P, L, P, P, P, L, I, I, F, P],
    "fp_gate_residual_bf16": [P, L, P, L, P, I, I, P],
    "fp_silu_gemv_bf16": [P, P, P, P, I, I, I, P],
    "fp_rel_l1_acc_floats": [],
    "fp_rel_l1_bf16": [P, P, L, P, P],
    "fp_gemm_bf16": [P, L, P, I, P, L, P, P, L, I, I, I, I, I, P, L, P, P],
    "fp_gemm_bf16_f32out": [P, L, P, L, P, P, L, I, I, I, P],
    "fp_gemm_bf16_rows2": [P, L, P, P, L, P, P, P, L, I, I, I, I, I, I, P, L, P, P, P],
    "fp_gemm_bf16_qknorm": [P, L, P, L, P, P, L, I, I, I, I, I, P, P, P, P, P, I, I, I, I, I, F, P],
    "fp_gemm_f16": [P, L, P, I, P, L, P, P, L, I, I, I, I, I, P, L, P, P],
    "fp_debug_gemm_variant": [I],
    "fp_debug_attn_variant": [I],
    "fp_quantize_rows_fp8": [P, L, I, P, L, I, P, L, P, I, P],
    "fp_layernorm_modulate_fp8": [P, L, P, P, P, L, P, I, I, F, P],
    "fp_gemm_fp8_bf16": [P, L, P, P, L, P, P, P, L, I, I, I, I, I, P, L, P, P],
    "fp_conv3d_causal_f16": [P, P, P, P, I, I, I, I, I, I, I, I, P, P],
    "fp_conv3d_causal_strided_f16": [P, P, P, P, I, I, I, I, I, I, I, I, P],
    "fp_groupnorm_workspace_floats": [I],
    "fp_groupnorm_stats_f16": [P, L, I, I, F, P, P, P],
    "fp_groupnorm_apply_f16": [P, P, L, I, I, P, P, P, I, P],
    "fp_gemm_f16_f32out": [P, L, P, L, P, L, I, I, I, P],
    "fp_softmax_framecausal_f32_f16": [P, I, I, L, I, I, I, F, P],
}

_lib = None


class FramepackHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the library once; raise loudly if it is not there (no CPU path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FramepackHipError(
            f"{LIB_PATH} not found: build it with `make -C framepack_amd/csrc` "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = c_int64 if name.endswith("_bytes") else c_int
    _lib = lib
    return lib


def check(err: int, what: str) -> None:
    if err == 0:
        return
    if err == 1:
        raise ValueError(f"{what}: argument rejected by libframepack_hip (hipErrorInvalidValue)")
    raise FramepackHipError(f"{what}: HIP error {err}")
