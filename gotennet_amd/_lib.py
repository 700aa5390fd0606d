"""ctypes binding of libgotennet_hip.so (the C ABI in include/gotennet_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  Loading
fails loudly when the shared object is missing (run ``python gotennet_amd/build.py``
or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
#: GN_LIB_PATH: a tuning variant of the library built by tools/variants.py (same ABI); default = the product library
LIB_PATH = os.environ.get("GN_LIB_PATH") or os.path.join(_PKG, "libgotennet_hip.so")

GN_ERR_BAD_ARG = 10001
ABI_VERSION = 8
LMAX_SLICED = 0x100      # GN_LMAX_SLICED: OR-ed into the lmax argument of the message / HTR entry points
LMAX_MEAN, LMAX_MAX = 0x200, 0x400      # GN_LMAX_MEAN / GN_LMAX_MAX: the reference's aggr = "mean" / "max" (message entries)

_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_long

# symbol -> argtypes (mirrors include/gotennet_hip.h one for one)
class GemmDesc(C.Structure):
    """gn_gemm_desc of include/gotennet_hip.h (one problem of gn_gemm_group)."""
    _fields_ = [("A", _P), ("lda", _I), ("W", _P), ("bias", _P), ("C", _P), ("ldc", _I),
                ("M", _I), ("N", _I), ("K", _I), ("act_lo", _I), ("act_hi", _I),
                ("row_cnt", _I), ("row_gstride", _I), ("row_goff", _I),
                ("res", _P), ("gate", _P), ("gate_mode", _I), ("pre_out", _P),
                ("pro_mode", _I), ("pro_lo", _I), ("pro_hi", _I), ("a_pre", _P), ("ldp", _I),
                ("a_gate", _P), ("ldg", _I), ("A2", _P), ("A3", _P), ("a_seg", _I), ("act_kind", _I)]


SIGNATURES = {
    "gn_abi_version": [C.POINTER(C.c_char_p)],
    "gn_build_csr": [_P, _I, _I, _P, _P, _P, _P],
    "gn_check_edges": [_P, _I, _I, _P, _P],
    "gn_build_csc": [_P, _P, _I, _I, _P, _P, _P, _P, _P],
    "gn_molecule_ptr": [_P, _I, _I, _P, _P],
    "gn_cosine_cutoff": [_P, _I, _F, _P, _P],
    "gn_out_degree": [_P, _I, _P, _P],
    "gn_edge_geometry": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _F, _P, _P, _P, _P],
    "gn_node_init": [_P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P],
    "gn_edge_init": [_P, _P, _P, _P, _I, _I, _I, _P, _P],
    "gn_layernorm_silu": [_P, _P, _P, _F, _I, _I, _P, _I, _P],
    "gn_layernorm": [_P, _P, _P, _F, _I, _I, _P, _P],
    "gn_tensor_norm": [_P, _P, _F, _I, _I, _I, _P, _P],
    "gn_gemm": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "gn_attn_softmax": [_P, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _P, _I, _P],
    "gn_message_aggregate": [_P, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "gn_eqff_fused_supported": [_I, _I, _I],
    "gn_eqff_fused_forward": [_P, _P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P],
    "gn_eqff_fused_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P, _I, _P],
    "gn_htr_edge": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "gn_eqff_context": [_P, _P, _F, _I, _I, _I, _P, _P],
    "gn_eqff_update": [_P, _P, _I, _I, _I, _P, _P, _P],
    "gn_gemm_ex": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _I, _I, _I, _P, _I, _P, _I, _I, _P],
    "gn_gemm_group": [_P, _I, _P],
    "gn_edge_vectors": [_P, _P, _P, _I, _P, _P, _P],
    "gn_gemm_split": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _I, _I, _I, _P, _I, _P, _I, _I, _P],
    "gn_split_bf16x3": [_P, _I, _I, _P, _P],
    "gn_split_bf16x3_size": [_I, _I],
    "gn_gemm_group_split": [_P, _I, _P],
    "gn_gemm_f16x2": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P, _I, _I, _I, _P, _I, _P, _I, _I, _P],
    "gn_split_f16x2": [_P, _I, _I, _P, _P],
    "gn_split_f16x2_size": [_I, _I],
    "gn_gemm_group_f16x2": [_P, _I, _P],
    "gn_htr_backward": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P],
    "gn_message_backward": [_P, _P, _I, _P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                            _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, C.c_long, _I, _I, _I, _I, _I, _I, _I, _P],
    "gn_message_backward_groups": [_I, _I, _I, _I],
    "gn_eqff_backward_a": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "gn_eqff_backward_b": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "gn_edge_init_backward": [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P],
    "gn_node_init_backward": [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _P, _P, _P],
    "gn_layernorm_silu_backward": [_P, _P, _P, _F, _P, _I, _I, _P, _I, _P],
    "gn_layernorm_backward": [_P, _P, _F, _P, _I, _I, _P, _P],
    "gn_tensor_norm_backward": [_P, _P, _P, _F, _I, _I, _I, _P, _P],
    "gn_gate": [_P, _I, _L, _P, _P],
    "gn_gate_backward": [_P, _P, _I, _L, _P, _P],
    "gn_edge_gate_backward": [_P, _P, _I, _P, _L, _P, _P, _P],
    "gn_edge_geometry_backward": [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _F, _P, _I, _P, _I, _P, _P, _P, _P],
    "gn_pos_scatter": [_P, _P, _P, _P, _P, _P, _I, _F, _P, _P],
    "gn_head_energy": [_P, _P, _F, _F, _F, _F, _P, _P, _P, _I, _I, _P, _P, _I, _P, _I, _P],
    "gn_head_grad": [_P, _P, _F, _P, _I, _I, _P, _I, _P],
    "gn_geb_context": [_P, _I, _I, _P, _I, _I, _I, _P, _I, _P],
    "gn_geb_gate": [_P, _I, _I, _I, _P, _I, _I, _I, _I, _P, _I, _P, _I, _P],
    "gn_dipole_reduce": [_P, _I, _P, _I, _P, _P, _I, _F, _F, _I, _I, _P, _P, _P],
    "gn_ese_reduce": [_P, _P, _P, _P, _I, _P, _I, _P, _P],
    "gn_radius_count": [_P, _P, _I, _F, _I, _P, _P],
    "gn_radius_fill": [_P, _P, _I, _F, _I, _P, C.c_int64, _P, _P, _P, _P],
}

_lib = None


class GotenNetHipError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GotenNetHipError(
            f"{LIB_PATH} is missing: the HIP extension has not been built "
            "(python gotennet_amd/build.py). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = C.c_long if name in ("gn_split_bf16x3_size", "gn_split_f16x2_size") else C.c_int
    if lib.gn_abi_version(None) != ABI_VERSION:
        raise GotenNetHipError("libgotennet_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


#: optional per-launch timer (bench/profiling only): an object with ``want(name, args) -> tag or None``
#: and a list ``events``; matching launches are bracketed by HIP events on the current stream.
TIMER = None


def call(name: str, *args) -> None:
    t = TIMER
    tag = t.want(name, args) if t is not None else None
    if tag is not None:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(load(), name)(*args)
        e1.record()
        t.events.append((tag, e0, e1))
    else:
        rc = getattr(load(), name)(*args)
    if rc != 0:
        what = "unsupported shape/flag combination" if rc == GN_ERR_BAD_ARG else f"hipError_t {rc}"
        raise GotenNetHipError(f"{name} failed: {what}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
