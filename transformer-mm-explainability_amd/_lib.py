"""ctypes binding of ``libmmx_hip.so`` (C-ABI declared in ``include/mmx_relevancy.h``).

There is NO fallback: if the library is missing or a kernel cannot run, callers get an exception.
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# MMX_LIB_PATH: load another build of the SAME library (the host-AddressSanitizer build csrc/asan/libmmx_hip_asan.so in
# tests/test_abi.py); never a fallback -- the file must exist and pass the ABI-version check like the default one
LIB_PATH = os.environ.get("MMX_LIB_PATH") or os.path.join(_HERE, "csrc", "libmmx_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mmx_relevancy.h")

ABI_VERSION = 2                 # == MMX_ABI_VERSION of include/mmx_relevancy.h (2: the round-5 schedule signature)
MMX_F32, MMX_F16, MMX_BF16 = 0, 1, 2
MMX_ENOTSUP = -95               # shape / view outside what the kernels support (include/mmx_relevancy.h)
MMX_ATTN_IO_BF16 = 0x200        # backward: bf16 dO in, bf16 dq / dk / dv out (with MMX_ATTN_MMA_BF16)
MMX_ATTN_MMA_BF16 = 0x100       # OR-ed into slab_dtype of the attention *_ex entry points (bf16 matrix cores)
MM_NORMALIZE, MM_SELF_IN_RULE10, MM_NAN_TO_ZERO = 1, 2, 4
CHAIN_CAUSAL = 1                # mmx_relevancy_self_chain_flags: probabilities of a causally masked tower (zeros above the diagonal)
SCALE_Q_FIRST, SCALE_SCORES = 0, 1
LRP_VALUES, LRP_SCORES = 1, 2        # phases of mmx_attn_relprop_phase
MAX_LAYERS = 48

_vp, _i, _i64, _sz, _f, _u = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float, C.c_uint
_vpp = C.POINTER(C.c_void_p)

_PROTOTYPES = {
    "mmx_abi_version": (_i, []),
    "mmx_last_error": (C.c_char_p, []),
    "mmx_set_option": (_i, [C.c_char_p, _i]),
    "mmx_avg_heads": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mmx_avg_heads_ex": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i64, _vp]),
    "mmx_self_chain_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "mmx_relevancy_self_chain": (_i, [_vpp, _vpp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "mmx_relevancy_self_chain_ex": (_i, [_vpp, _vpp, _i, _i, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "mmx_relevancy_self_chain_flags": (_i, [_vpp, _vpp, _i, _i, _i, _i, _i, _i64, _vp, _vp, _vp, _vp, _i, _u, _vp, _sz, _vp]),
    "mmx_relevancy_self_chain_half": (_i, [_vpp, _vpp, _i, _i, _i, _i, _i, _i64, _vp, _vp, _sz, _vp]),
    "mmx_bmm_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i64, _i64, _i64, _i, _vp]),
    "mmx_handle_residual": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "mmx_mm_rules_workspace_bytes": (_sz, [_i, _i]),
    "mmx_mm_attention_rules": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _u, _vp, _vp, _sz, _vp]),
    "mmx_lxmert_schedule_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "mmx_lxmert_schedule": (_i, [_vpp, _vpp, _i, _vpp, _vpp, _i] + [_vpp] * 8 + [_i, _i, _i, _i, _i, _u, _vp,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mmx_lrp_workspace_bytes": (_sz, []),
    "mmx_lrp_split_signs": (_i, [_vp, _vp, _i64, _i, _vp]),
    "mmx_lrp_safe_divide": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "mmx_lrp_linear_combine": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _i64, _vp, _vp]),
    "mmx_lrp_add_relprop": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i64, _vp, _vp]),
    "mmx_lrp_clone_relprop": (_i, [_vpp, _i, _vp, _vp, _i64, _vp]),
    "mmx_lrp_mha_rescale": (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp]),
    "mmx_heatmap_bilinear_minmax": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "mmx_otsu_masks": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "mmx_rollout_workspace_bytes": (_sz, [_i, _i]),
    "mmx_rollout_chain": (_i, [_vpp, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "mmx_attn_capture_fwd": (_i, [_vp, _vp, _vp] + [_i64] * 9 + [_vp, _i64, _i64, _vp, _vp, _i64, _i64, _i64,
                                  _i, _i, _i, _i, _i, _f, _i, _vp]),
    "mmx_attn_capture_fwd_ex": (_i, [_vp, _vp, _vp] + [_i64] * 9 + [_vp, _i64, _i64, _vp, _i, _vp, _i64, _i64, _i64,
                                     _i, _i, _i, _i, _i, _f, _i, _vp]),
    "mmx_attn_capture_bwd_ex": (_i, [_vp, _vp, _vp] + [_i64] * 9 + [_vp, _i64, _i, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64,
                                     _vp, _vp, _vp, _vp]
                                + [_i64] * 9 + [_i, _i, _i, _i, _i, _f, _i, _i, _vp, _sz, _vp]),
    "mmx_attn_capture_bwd_rowrel": (_i, [_vp, _vp, _vp] + [_i64] * 9 + [_vp, _i64, _i, _vp, _i64, _i64, _i64, _vp, _i64, _i64,
                                         _i64, _vp, _vp, _vp, _vp]
                                    + [_i64] * 9 + [_i, _i, _i, _i, _i, _f, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "mmx_attn_capture_bwd_rowrel_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "mmx_attn_capture_bwd_workspace_bytes": (_sz, [_i, _i, _i]),
    "mmx_attn_capture_bwd": (_i, [_vp, _vp, _vp] + [_i64] * 9 + [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _vp, _vp]
                             + [_i64] * 9 + [_i, _i, _i, _i, _i, _f, _i, _i, _vp, _sz, _vp]),
    "mmx_attn_relprop": (_i, [_vp] * 5 + [_i64] * 15 + [_vp] * 5 + [_i64] * 9 + [_i, _i, _i, _i, _i, _f, _i, _vp]),
    "mmx_attn_relprop_phase": (_i, [_vp] * 5 + [_i64] * 15 + [_vp] * 5 + [_i64] * 9 + [_i, _i, _i, _i, _i, _f, _i, _vp, _i, _vp]),
    "mmx_detr_decoder_rows_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "mmx_detr_decoder_rows": (_i, [_vpp] * 4 + [_i] * 5 + [_i64, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "mmx_linear_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mmx_rows_to_dense": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "mmx_rows_add": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "mmx_chain_matvec": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mmx_chain_vecmat_workspace_bytes": (_sz, [_i, _i]),
    "mmx_chain_vecmat": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _sz, _vp]),
    "mmx_avg_heads_vecmat_workspace_bytes": (_sz, [_i, _i]),
    "mmx_avg_heads_vecmat": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i64, _vp, _sz, _vp]),
    "mmx_quick_gelu_fwd": (_i, [_vp, _vp, _i64, _vp]),
    "mmx_quick_gelu_fwd_bf16": (_i, [_vp, _vp, _i64, _vp]),
    "mmx_quick_gelu_bwd": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "mmx_quick_gelu_bwd_bcast": (_i, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "mmx_quick_gelu_bwd_bcast_bf16": (_i, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "mmx_layernorm_bwd_add_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    "mmx_layernorm_bwd_add": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    "mmx_add_layernorm_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _vp]),
    "mmx_add_layernorm_fwd_ex": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _i, _vp]),
    "mmx_event_create": (_i, [_vpp]),
    "mmx_event_destroy": (_i, [_vp]),
    "mmx_event_record": (_i, [_vp, _vp]),
    "mmx_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(C.c_float)]),
}


class MMXError(RuntimeError):
    pass


_lib = None


def header_symbols():
    """Function names declared in include/mmx_relevancy.h."""
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(mmx_[a-z0-9_]+)\s*\(", text)))


def lib():
    """Load (once) and return the ctypes handle.  Raises MMXError if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MMXError(
                "libmmx_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C transformer-mm-explainability_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.mmx_abi_version() != ABI_VERSION:
            raise MMXError("libmmx_hip.so ABI version %d != %d (stale build? `make -C transformer-mm-explainability_amd/csrc`)"
                           % (handle.mmx_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().mmx_last_error().decode("utf-8", "replace")
        raise MMXError("%s failed (rc=%d): %s" % (what, rc, msg))


def ptr_table(ptrs):
    """HOST array of device pointers for the `const void* const*` parameters."""
    arr = (C.c_void_p * len(ptrs))(*ptrs)
    return C.cast(arr, _vpp), arr  # keep `arr` alive while the call runs
