"""ctypes binding of libivit_hip.so (the C ABI declared in include/ivit_hip.h).

The product path has NO fallback: if the shared library is missing or a call
returns a non-zero status this module raises.  ``build()`` (re)compiles the
library in-tree with hipcc for gfx950; it is what ``__graft_entry__.build()``
calls.  The library is git-ignored but travels to the GPU box with the tree.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libivit_hip.so")
CSRC = os.path.join(_HERE, "csrc")

i64, i32, u32, f32, vp = C.c_int64, C.c_int32, C.c_uint32, C.c_float, C.c_void_p
ci = C.c_int

# name -> argtypes, exactly the prototypes of include/ivit_hip.h
SIGNATURES = {
    "ivit_quantize_input_f32_i8": [vp, vp, i64, f32, vp],
    "ivit_quantize_patchify_f32_i8": [vp, vp, ci, ci, ci, ci, f32, vp],
    "ivit_gemm_i8_requant": [vp, i64, vp, i64, vp, vp, vp, vp, i64, ci, ci, ci, vp],
    "ivit_gemm_i8_requant_residual": [vp, i64, vp, i64, vp, vp, vp, vp, i64, u32, i32, u32, i32, vp, i64, ci, ci, ci, vp],
    "ivit_attention_fused_i8_ex": [vp, vp, ci, ci, ci, ci, u32, i32, f32, u32, i32, ci, vp],
    "ivit_layernorm_i8_ex": [vp, i64, ci, ci, vp, vp, vp, vp, vp, i64, ci, vp],
    "ivit_layernorm_i8_compat": [vp, i64, ci, ci, vp, vp, vp, vp, vp, vp, vp, i64, ci, vp],
    "ivit_attention_fused_i8_compat": [vp, vp, ci, ci, ci, ci, u32, i32, f32, u32, i32, vp, ci, vp],
    "ivit_attention_fused_i8_compat_band": [vp, vp, ci, ci, ci, ci, u32, i32, f32, u32, i32, vp, vp, ci, ci, vp],
    "ivit_attention_fused_i8_wide": [vp, vp, ci, ci, ci, ci, u32, i32, f32, u32, i32, vp, vp, ci, ci, ci, vp],
    "ivit_shiftgelu_build_lut_ex": [f32, u32, i32, vp, vp, vp],
    "ivit_shiftgelu_lut_i8_ex": [vp, i64, ci, ci, vp, vp, i64, ci, vp],
    "ivit_pack_weight_frags_i8": [vp, i64, ci, ci, vp, vp],
    "ivit_pack_weight_frags16_i8": [vp, i64, ci, ci, vp, vp],
    "ivit_tile_operand_i8": [vp, i64, i64, ci, vp, vp],
    "ivit_untile_operand_i8": [vp, i64, ci, vp, i64, vp],
    "ivit_gemm_i8_requant_ex": [vp, i64, vp, i64, vp, vp, vp, vp, i64, ci, ci, ci, ci, vp],
    "ivit_gemm_i8_requant_residual_ex": [vp, i64, vp, i64, vp, vp, vp, vp, i64, u32, i32, u32, i32, vp, i64, ci, ci, ci, ci, vp],
    "ivit_gemm_i8_requant_qkv_ex": [vp, i64, vp, i64, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, vp],
    "ivit_gemm_i8_requant_residual_i16": [vp, i64, vp, i64, vp, vp, vp, vp, i64, u32, i32, u32, i32, vp, i64, ci, ci, ci, vp],
    "ivit_gemm_i8_requant_lut_ex": [vp, i64, vp, i64, vp, vp, vp, vp, vp, i64, ci, ci, ci, ci, vp],
    "ivit_gemm_i8_requant_i16": [vp, i64, vp, i64, vp, vp, vp, vp, i64, ci, ci, ci, vp],
    "ivit_gemm_i8_requant_residual_i16_ex": [vp, i64, vp, i64, vp, vp, vp, vp, i64, u32, i32, u32, i32, vp, i64, ci, ci, ci, ci, vp],
    "ivit_gemm_i8_requant_i16_residual_i16_ex": [vp, i64, vp, i64, vp, vp, vp, vp, i64, u32, i32, u32, i32, vp, i64, ci, ci, ci, ci, vp],
    "ivit_gemm_i8_requant_qkv": [vp, i64, vp, i64, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, vp],
    "ivit_gemm_i8_i32": [vp, i64, vp, i64, vp, vp, i64, ci, ci, ci, vp],
    "ivit_attention_fused_i8": [vp, vp, ci, ci, ci, ci, u32, i32, f32, u32, i32, vp],
    "ivit_layernorm_i8": [vp, i64, ci, ci, vp, vp, vp, vp, vp, i64, vp],
    "ivit_layernorm_i32_f32": [vp, i64, ci, ci, vp, vp, vp, i64, vp],
    "ivit_layernorm_f32_f32": [vp, i64, ci, ci, vp, ci, vp, vp, vp, i64, vp],
    "ivit_layernorm_f32_f32_ex": [vp, i64, ci, ci, vp, ci, vp, vp, vp, i64, ci, vp],
    "ivit_shiftmax_f32_i8": [vp, i64, ci, ci, f32, vp, i64, vp],
    "ivit_shiftmax_f32_i16": [vp, i64, ci, ci, f32, ci, vp, i64, vp],
    "ivit_bgemm_pv_i16_i8": [vp, vp, vp, ci, ci, ci, ci, vp],
    "ivit_bgemm_pv_i32_i8": [vp, vp, vp, ci, ci, ci, ci, i64, vp],
    "ivit_quantize_input_f32_i32": [vp, vp, i64, f32, ci, vp],
    "ivit_shiftgelu_i8": [vp, i64, ci, ci, f32, u32, i32, vp, i64, vp],
    "ivit_shiftgelu_i8_i32": [vp, i64, ci, ci, f32, vp, i64, vp],
    "ivit_shiftgelu_build_lut": [f32, u32, i32, vp, vp],
    "ivit_shiftgelu_lut_i8": [vp, i64, ci, ci, vp, vp, i64, vp],
    "ivit_shiftmax_i8": [vp, i64, ci, ci, f32, vp, i64, vp],
    "ivit_requant_i32": [vp, i64, ci, vp, vp, ci, vp, vp, vp, ci, ci, vp, vp],
    "ivit_residual_requant_i8": [vp, u32, i32, vp, u32, i32, vp, i64, vp],
    "ivit_embed_assemble_i8": [vp, vp, vp, u32, i32, vp, ci, ci, ci, vp],
    "ivit_embed_assemble_i16": [vp, vp, vp, u32, i32, vp, ci, ci, ci, vp],
    "ivit_head_argmax": [vp, vp, ci, ci, vp, vp, vp],
    "ivit_bgemm_qk_i8": [vp, vp, vp, ci, ci, ci, ci, vp],
    "ivit_bgemm_pv_i8": [vp, vp, vp, ci, ci, ci, ci, vp],
    "ivit_f32_to_i32": [vp, i64, ci, vp, ci, ci, vp, vp],
    "ivit_i32_to_f32": [vp, i64, ci, vp, ci, vp, vp],
    "ivit_narrow_i32_i8": [vp, vp, i64, vp, vp],
    # Swin (include/ivit_hip.h, second half)
    "ivit_quantize_patchify_ld_f32_i8": [vp, vp, i64, ci, ci, ci, ci, f32, vp],
    "ivit_quantize_patchify_u8_i8": [vp, vp, i64, ci, ci, ci, ci, vp, vp],
    "ivit_minmax_f32": [vp, i64, vp, vp],
    "ivit_shiftmax_i32_i8": [vp, i64, ci, ci, f32, vp, i64, vp],
    "ivit_requant_i8_i16": [vp, u32, i32, vp, i64, vp],
    "ivit_residual_requant_i16": [vp, ci, vp, vp, u32, i32, vp, u32, i32, vp, i64, ci, ci, ci, ci, ci, vp],
    "ivit_layernorm_i16_i8": [vp, ci, ci, vp, vp, vp, vp, vp, i64, ci, ci, ci, ci, vp],
    "ivit_layernorm_i16_i8_compat": [vp, ci, ci, f32, ci, vp, vp, vp, vp, vp, i64, ci, ci, ci, ci, vp],
    "ivit_patch_merge_i16": [vp, vp, ci, ci, ci, ci, vp],
    "ivit_avgpool_requant_i8": [vp, vp, ci, ci, ci, u32, i32, vp],
    # I-BERT operator family (include/ivit_hip.h, last section)
    "ivit_ibert_gelu_i32": [vp, i64, f32, f32, f32, vp, vp],
    "ivit_ibert_softmax_i32": [vp, i64, ci, ci, f32, f32, f32, f32, f32, u32, i32, ci, vp, i64, vp, vp],
    "ivit_ibert_layernorm_i32_f32": [vp, i64, ci, ci, vp, vp, f32, vp, i64, vp],
    "ivit_ibert_gelu_build_lut": [f32, f32, f32, f32, f32, u32, i32, vp, vp],
    "ivit_ibert_softmax_build_table": [f32, f32, f32, f32, f32, f32, u32, i32, vp, vp],
    "ivit_attention_fused_i8_ibert": [vp, vp, ci, ci, ci, ci, u32, i32, u32, i32, vp, vp, ci, ci, vp],
    "ivit_attention_fused_i8_ibert_wide": [vp, vp, ci, ci, ci, ci, u32, i32, u32, i32, vp, vp, ci, ci, ci, vp],
    "ivit_ibert_layernorm_i8": [vp, i64, ci, ci, f32, vp, vp, f32, vp, vp, vp, i64, ci, vp],
    "ivit_ibert_layernorm_i16_i8": [vp, i64, ci, ci, f32, vp, vp, f32, vp, vp, vp, i64, vp],
    "ivit_ibert_layernorm_i16_i8_ex": [vp, i64, ci, ci, f32, vp, vp, f32, vp, vp, vp, i64, ci, vp],
    "ivit_ibert_gelu_f32_f32": [vp, i64, f32, f32, f32, f32, f32, vp, vp],
    "ivit_ibert_softmax_f32_f32": [vp, i64, ci, ci, f32, f32, f32, f32, f32, f32, u32, i32, ci, vp, i64, vp, vp],
    "ivit_ibert_layernorm_f32_f32": [vp, i64, ci, ci, vp, ci, vp, vp, f32, vp, i64, vp],
    "ivit_window_attention_i8": [vp, vp, i64, vp, vp, ci, ci, ci, ci, ci, ci, u32, i32, u32, i32, f32, u32, i32, vp],
    "ivit_window_attention_i8_compat": [vp, vp, i64, vp, vp, ci, ci, ci, ci, ci, ci, u32, i32, u32, i32, f32, u32, i32, vp, vp, vp],
    "ivit_window_attention_i8_band": [vp, vp, i64, vp, vp, ci, ci, ci, ci, ci, u32, i32, u32, i32, f32, u32, i32, vp, ci, ci, ci, ci, ci, ci, vp],
    "ivit_window_attention_i8_unwindow": [vp, vp, i64, vp, vp, ci, ci, ci, ci, ci, ci, u32, i32, u32, i32, f32, u32, i32, vp, vp,
                                          ci, ci, ci, ci, vp],
}


# include/ivit_hip_debug.h: exported by libivit_hip_lab.so only (tests / scripts; the product library is stateless)
LAB_SIGNATURES = {
    "ivit_debug_force_small_gemm": [ci],
    "ivit_debug_set_gemm_flags": [ci],
    "ivit_debug_set_gemm_flags2": [ci],
    "ivit_debug_ln_wave_per_row": [ci],
    "ivit_debug_ln_ablate": [ci],
    "ivit_debug_ln_stream_cfg": [ci],
    "ivit_debug_ln_stamp_buffer": [vp],
    "ivit_gemm_i8_requant_gelu_ex": [vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, i64, ci, ci, ci, ci, vp],
    "ivit_gemm_gelu_workspace_bytes": [ci, vp],
    "ivit_debug_set_stamp_buffer": [vp],
}
LAB_PATH = os.path.join(_HERE, "libivit_hip_lab.so")


class IvitError(RuntimeError):
    pass


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip -> libivit_hip.so + libivit_hip_lab.so (hipcc --offload-arch=gfx950).  Always runs `make`: it is
    incremental, and it also runs the register / scratch check of the GEMM kernels (csrc/Makefile, target `check`)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else []) + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None
_lab = None
_use_lab = os.environ.get("IVIT_USE_LAB_LIBRARY") == "1"     # scripts/: measure with the lab build's knobs


def _load(path, signatures):
    if not os.path.exists(path):
        raise IvitError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(there is no CPU / PyTorch fallback for the integer ViT path)")
    L = C.CDLL(path)
    L.ivit_version.restype = ci
    L.ivit_last_error_string.restype = C.c_char_p
    for name, args in signatures.items():
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = ci
    return L


def lib():
    """Load the product library (never builds implicitly; never falls back)."""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, SIGNATURES)
    return _lib


def lab():
    """libivit_hip_lab.so: the same C ABI compiled with the measurement hooks of include/ivit_hip_debug.h."""
    global _lab
    if _lab is None:
        _lab = _load(LAB_PATH, {**SIGNATURES, **LAB_SIGNATURES})
    return _lab


class lab_session:
    """`with _lib.lab_session():` -- every `call()` inside goes to the lab build (tests of kernel forms, A/B scripts).
    On exit the lab build's knobs are reset."""

    def __enter__(self):
        global _use_lab
        self.prev, _use_lab = _use_lab, True
        return lab()

    def __exit__(self, *exc):
        global _use_lab
        L = lab()
        L.ivit_debug_force_small_gemm(0)
        L.ivit_debug_set_gemm_flags(0)
        L.ivit_debug_set_gemm_flags2(0)
        L.ivit_debug_ln_wave_per_row(0)
        L.ivit_debug_ln_ablate(0)
        L.ivit_debug_ln_stream_cfg(0)
        L.ivit_debug_ln_stamp_buffer(None)
        _use_lab = self.prev
        return False


def call(name: str, *args):
    L = lab() if _use_lab else lib()
    rc = getattr(L, name)(*args)
    if rc != 0:
        raise IvitError(f"{name} failed ({rc}): {L.ivit_last_error_string().decode()}")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())
