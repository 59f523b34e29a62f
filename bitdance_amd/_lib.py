"""ctypes binding of libbitdance_hip.so (C ABI in include/bitdance_hip.h).

There is no CPU fallback: if the library is missing or a call fails this raises, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

# torch ships its own libamdhip64.so; it must be loaded BEFORE our library so that both resolve to the same HIP
# runtime instance (loading ours first binds /opt/rocm's copy and torch's device pointers become foreign to it).
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
# BD_HIP_LIB: a measurement build of the SAME sources (tools/launch_anatomy.py loads libbitdance_hip_stamp.so); never a fallback
LIB_PATH = os.environ.get("BD_HIP_LIB") or os.path.join(HERE, "libbitdance_hip.so")

_lib = None


class BitDanceHipError(RuntimeError):
    pass


class BitDanceUnsupported(BitDanceHipError):
    """A shape / mode the native kernels do not cover, refused on the host BEFORE anything was launched (the C ABI's
    BD_ERR_UNSUPPORTED = -22, or a host-side validation in the wrappers): the only error a caller may answer by taking another
    path.  Launch and runtime failures are plain BitDanceHipError and must propagate."""


BD_ERR_UNSUPPORTED = -22


_PROTOS = {
    "bd_version": (C.c_int, []),
    "bd_last_error": (C.c_char_p, []),
    "bd_pack_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bd_set_weight_layout": (C.c_int, [C.c_int]),
    "bd_set_gemm_option": (C.c_int, [C.c_char_p, C.c_int]),
    "bd_pack_weight_swiglu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "bd_rows_to_frag": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bd_gemm_partial": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bd_gemm_bf16": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.c_void_p]),
    "bd_gemm_f32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p]),
    "bd_pack_weight8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bd_pack_weight8_swiglu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "bd_pack_weight8k": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bd_pack_weight8k_swiglu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "bd_gemm_w8a8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bd_quant_rows8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bd_gemm_w8": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bd_gemm_swiglu": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bd_gemm_swiglu_splitk": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p]),
    "bd_ctx_create": (C.c_void_p, []),
    "bd_ctx_destroy": (None, [C.c_void_p]),
    "bd_ctx_set_int": (C.c_int, [C.c_void_p, C.c_char_p, C.c_longlong]),
    "bd_ctx_set_float": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "bd_ctx_set_ptr": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p]),
    "bd_ctx_set_comm": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bd_ctx_set_tp": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "bd_ctx_finalize": (C.c_int, [C.c_void_p]),
    "bd_ctx_ws_count": (C.c_int, [C.c_void_p]),
    "bd_ctx_ws_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "bd_ctx_ws_bytes": (C.c_longlong, [C.c_void_p, C.c_int]),
    "bd_ctx_bind": (C.c_int, [C.c_void_p]),
    "bd_head_set_schedule": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_float]),
    "bd_head_set_cfg": (C.c_int, [C.c_void_p, C.c_float]),
    "bd_head_sample": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bd_head_cond": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bd_head_eval": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "bd_projector": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bd_llm_step": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bd_graph_capture": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "bd_graph_launch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "bd_step_reset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "bd_prof_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "bd_prof_count": (C.c_int, [C.c_void_p]),
    "bd_prof_get": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_double)]),
    "bd_gfq_indices": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bd_gfq_codes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bd_probe_read": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_void_p, C.c_void_p]),
    "bd_comm_create": (C.c_void_p, [C.c_int, C.c_int, C.c_longlong]),
    "bd_comm_create2": (C.c_void_p, [C.c_int, C.c_int, C.c_longlong, C.c_longlong]),
    "bd_comm_create3": (C.c_void_p, [C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_longlong]),
    "bd_comm_ipc_handles3": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bd_comm_open_peer3": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "bd_comm_set_peer_ptrs3": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bd_comm_local_hbuf": (C.c_void_p, [C.c_void_p]),
    "bd_comm_hbuf_bytes": (C.c_longlong, [C.c_void_p]),
    "bd_comm_sp_selftest": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "bd_comm_set_loopback": (C.c_int, [C.c_void_p]),
    "bd_comm_prepushed": (C.c_longlong, [C.c_void_p]),
    "bd_comm_gather_ptr": (C.c_void_p, [C.c_void_p]),
    "bd_comm_gather_bytes": (C.c_longlong, [C.c_void_p]),
    "bd_comm_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "bd_comm_destroy": (None, [C.c_void_p]),
    "bd_comm_ipc_handles": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bd_comm_open_peer": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "bd_comm_set_peer_ptrs": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "bd_comm_local_data": (C.c_void_p, [C.c_void_p]),
    "bd_comm_local_flags": (C.c_void_p, [C.c_void_p]),
    "bd_comm_set_rccl": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "bd_comm_set_timeout": (C.c_int, [C.c_void_p, C.c_double]),
    "bd_comm_set_fences": (C.c_int, [C.c_void_p, C.c_int]),
    "bd_comm_mark_prepushed": (C.c_int, [C.c_void_p]),
    "bd_comm_reset": (C.c_int, [C.c_void_p]),
    "bd_comm_error": (C.c_int, [C.c_void_p]),
    "bd_comm_exchanges": (C.c_longlong, [C.c_void_p]),
    "bd_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_longlong)]),
    "bd_comm_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                  C.POINTER(C.c_int), C.c_void_p]),
    "bd_comm_copy_out": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]),
    "bd_conv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                          C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bd_conv_strided": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bd_gn_stats": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "bd_gn_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                              C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bd_tokens_to_padded": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "bd_gemm_config": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS)


def lib() -> C.CDLL:
    """Load (once) and return the native library; raise if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BitDanceHipError(
                f"{LIB_PATH} not found: build it with `python -m bitdance_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path.")
        try:
            l = C.CDLL(LIB_PATH)
        except OSError as e:  # missing libamdhip64 etc.
            raise BitDanceHipError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().bd_last_error()
        raise (BitDanceUnsupported if rc == BD_ERR_UNSUPPORTED else BitDanceHipError)(
            f"{what or 'libbitdance_hip'} failed ({rc}): {msg.decode() if msg else ''}")
