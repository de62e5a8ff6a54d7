"""ctypes binding of liblgd_hip.so (C ABI: include/lgd_hip.h).

This is the binding a maintainer of the reference would add: plain pointers and sizes, no torch
types across the boundary.  The library is mandatory — there is no CPU or PyTorch fallback for the
kernels; a missing / stale .so raises at import of the first op.
"""
import ctypes as C
import os

# PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so).  It must be in the process
# BEFORE liblgd_hip.so is dlopen'ed so that the library's libamdhip64.so.7 dependency resolves to
# that same runtime (one HIP runtime per process: device pointers and streams are shared with torch).
import torch  # noqa: F401  (load order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblgd_hip.so")
ABI_VERSION = 10

c_void_p, c_int, c_i64, c_float = C.c_void_p, C.c_int, C.c_int64, C.c_float


class LgdGemmDesc(C.Structure):
    _fields_ = [
        ("a0", c_void_p), ("a1", c_void_p),
        ("lda0", c_i64), ("lda1", c_i64),
        ("c0", C.c_int32), ("c1", C.c_int32),
        ("taps", C.c_int32),
        ("hin", C.c_int32), ("win", C.c_int32), ("hout", C.c_int32), ("wout", C.c_int32),
        ("stride", C.c_int32), ("ups", C.c_int32),
        ("w", c_void_p), ("ldw", c_i64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("nb_o", C.c_int32), ("nb_i", C.c_int32),
        ("a_bs_o", c_i64), ("a_bs_i", c_i64), ("w_bs_o", c_i64), ("w_bs_i", c_i64),
        ("c_bs_o", c_i64), ("c_bs_i", c_i64), ("r_bs_o", c_i64), ("r_bs_i", c_i64),
        ("bias", c_void_p), ("bias2", c_void_p),
        ("res", c_void_p), ("ldr", c_i64),
        ("alpha", c_float), ("epi", C.c_int32),
        ("c", c_void_p), ("ldc", c_i64),
        ("splits", C.c_int32), ("ws", c_void_p),
        ("tile", C.c_int32),
        ("cnt", c_void_p),
        ("rowstat", c_void_p), ("colsum", c_void_p),
    ]


EPI_GEGLU, EPI_OUT_F32, EPI_RES_F32, EPI_ROWNORM = 1, 2, 4, 8

# name -> argtypes (every function returns int; last argument is the hipStream_t)
_P, _I, _L, _F = c_void_p, c_int, c_i64, c_float
SIGNATURES = {
    "lgd_abi_version": [],
    "lgd_set_option": [C.c_char_p, _I],
    "lgd_gemm_f16": [C.POINTER(LgdGemmDesc), _P],
    "lgd_conv_in_f16": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "lgd_conv_out_f16": [_P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "lgd_groupnorm_f16": [_P, _P, _I, _I, _I, _I, _I, _F, _P, _P, _I, _P, _P, _I, _P, _P],
    "lgd_groupnorm_bwd_f16": [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P],
    "lgd_layernorm_f16": [_P, _L, _P, _L, _I, _I, _F, _P, _P, _P, _I, _L, _L, _P],
    "lgd_layernorm_bwd_f16": [_P, _L, _P, _L, _P, _L, _I, _I, _P, _P, _I, _L, _L, _L, _I, _P],
    "lgd_attn_fwd_f16": [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _I, _I, _I, _I, _I, _F, _P],
    "lgd_attn_bwd_f16": [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _P,
                         _P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _F, _P],
    "lgd_attn_bwd_keys_f16": [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _P,
                              _P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _I, _I, _F, _P],
    "lgd_cross_attn_fwd_f16": [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _I, _I,
                               _I, _I, _I, _I, _I, _F, _P],
    "lgd_cross_attn_bwd_f16": [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _P, _L, _L,
                               _I, _I, _I, _I, _I, _F, _P],
    "lgd_geglu_bwd_f16": [_P, _P, _P, _L, _I, _P],
    "lgd_geglu_fwd_f16": [_P, _P, _L, _I, _P],
    "lgd_add_f16": [_P, _P, _P, _L, _P],
    "lgd_softmax_rows_f16": [_P, _P, _L, _I, _F, _P],
    "lgd_scale_f16": [_P, _P, _F, _L, _P],
    "lgd_upsample2x_bwd_f16": [_P, _P, _I, _I, _I, _I, _P],
    "lgd_cfg_ddim_step_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "lgd_cfg_multistep_step_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "lgd_axpy_f32": [_P, _P, _P, _P, _I, _P, _L, _L, _P],
    "lgd_select_row_f32": [_P, _P, _P, _I, _P],
    "lgd_scale_rows_f32": [_P, _P, _P, _P, _I, _I, _L, _I, _P],
    "lgd_attn_causal_fwd_f16": [_P, _L, _L, _P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _I, _I, _I, _F, _P],
    "lgd_quick_gelu_f16": [_P, _P, _L, _P],
    "lgd_act_f16": [_P, _P, _L, _I, _P],
    "lgd_nchw_to_nhwc8_f16": [_P, _P, _I, _I, _I, _P],
    "lgd_sam_relpos_qkv_f16": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    "lgd_sam_window_merge_f16": [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "lgd_ca_energy_f32": [_P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P, _P],
    "lgd_boxdiff_energy_f32": [_P, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, _P, _P],
}

_lib = None


def load():
    """Loads the shared library once; raises RuntimeError if it is missing or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
            "(there is no CPU/PyTorch fallback for the HIP kernels)")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"liblgd_hip.so does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = c_int
    if lib.lgd_abi_version() != ABI_VERSION:
        raise RuntimeError("liblgd_hip.so ABI version mismatch; rebuild")
    # LGD_OPTIONS="gn_fused=0,attn_w4=2": kernel-variant switches (lgd_set_option) for A/B runs of unmodified commands
    for item in filter(None, os.environ.get("LGD_OPTIONS", "").split(",")):
        name, _, value = item.partition("=")
        check(lib.lgd_set_option(name.strip().encode(), int(value)), f"lgd_set_option({name})")
    _lib = lib
    return lib


_ERR = {-1: "invalid argument", -2: "kernel launch failed", -3: "unsupported shape"}


def check(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name} failed: {_ERR.get(rc, rc)}")
