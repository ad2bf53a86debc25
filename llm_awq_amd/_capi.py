"""ctypes view of the C ABI (include/awq_cdna4.h) -- used by the host-side format tools, by
`llm_awq_amd.ops` and by the parity tests (which must call through the C ABI).

There is NO fallback: if libawq_cdna4.so is missing this module raises at first use.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (AWQ_CDNA4_LIB: another build of the same library, e.g. an AWQ_PROBES=1 build kept beside the product one for timing experiments)
LIB_PATH = os.environ.get("AWQ_CDNA4_LIB") or os.path.join(_HERE, "lib", "libawq_cdna4.so")

AWQ_F16, AWQ_BF16 = 0, 1
AWQ_ERR_WORKSPACE = -7  # include/awq_cdna4.h
_lib = None

_vp, _i, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
SIGNATURES = {
    "awq_abi_version": (_i, []),
    "awq_status_string": (ctypes.c_char_p, [_i]),
    "awq_last_hip_error": (ctypes.c_char_p, []),
    "awq_w4a16_gemv": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "awq_w4a16_gemm_workspace_bytes": (_sz, [_i, _i, _i]),
    "awq_w4a16_gemm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "awq_w4a16_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "awq_unpack_v2": (_i, [_vp, _vp, _i, _i, _vp]),
    "awq_dequant_v2": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "awq_pack_v2": (_i, [_vp, _vp, _i, _i, _vp]),
    "awq_repack_v1_to_v2": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "awq_repack_v2_to_cdna4": (_i, [_vp, _vp, _i, _i, _vp]),
    "awq_repack_cdna4_to_v2": (_i, [_vp, _vp, _i, _i, _vp]),
    "awq_unpack_cdna4": (_i, [_vp, _vp, _i, _i, _vp]),
    "awq_dequant_cdna4": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "awq_pack_sz_cdna4": (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    "awq_pack_szh_cdna4": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "awq_w4a16_decode_cdna4": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "awq_w4a16_gemv_cdna4": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "awq_w4a16_mlp_gate_up_cdna4": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "awq_w4a16_mlp_gate_up_forward_cdna4": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "awq_w4a16_mlp_gate_up_forward_cdna4_workspace_bytes": (_sz, [_i, _i, _i]),
    "awq_w4a16_mlp_gate_up_forward_cdna4_ws": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "awq_w4a16_rmsnorm_forward_cdna4": (_i, [_vp, _vp, ctypes.c_float, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "awq_rmsnorm": (_i, [_vp, _vp, ctypes.c_float, _vp, _i, _i, _i, _vp]),
    "awq_w4a16_forward_cdna4_workspace_bytes": (_sz, [_i, _i, _i]),
    "awq_midm_init": (_i, []),
    "awq_w4a16_gemm_cdna4_plan": (_i, [_i, _i, _i, _vp, _vp]),
    "awq_w4a16_gemm_cdna4_pair_plan": (_i, [_i, _i, _i]),
    "awq_w4a16_gemm_cdna4_pair_lost": (_i, [_vp]),
    "awq_w4a16_gemm_cdna4_narrow_kernel": (_i, [_i, _i, _i, _i, _i, _i]),
    "awq_w4a16_decode_cdna4_plan": (_i, [_i, _i, _i, _i, _vp]),
    "awq_w4a16_gemm_cdna4": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "awq_w4a16_forward_cdna4": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "awq_w4a16_forward_cdna4_szh": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "awq_w4a16_moe_gemm": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "awq_w4a16_moe_forward_cdna4": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "awq_w4a16_moe_forward_cdna4_szh": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "awq_silu_mul": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "awq_w4a16_moe_mlp_gate_up_cdna4": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "awq_w4a16_moe_mlp_gate_up_cdna4_szh": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "awq_pack_w3": (_i, [_vp, _vp, _i, _i, _vp]),
    "awq_unpack_w3": (_i, [_vp, _vp, _i, _i, _vp]),
    "awq_dequant_w3": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "awq_w3a16_forward_workspace_bytes": (_sz, [_i, _i, _i]),
    "awq_w3a16_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "awq_oneshot_buffer_bytes": (_sz, [_i, _i]),
    "awq_oneshot_alloc": (_i, [ctypes.POINTER(_vp), _i, _i]),
    "awq_oneshot_free": (_i, [_vp]),
    "awq_oneshot_ipc_export": (_i, [_vp, _vp]),
    "awq_oneshot_ipc_open": (_i, [_vp, ctypes.POINTER(_vp)]),
    "awq_oneshot_ipc_close": (_i, [_vp]),
    "awq_oneshot_allreduce": (_i, [ctypes.POINTER(_vp), _vp, _vp, _i, _i, _i, _i, ctypes.c_uint, _i, _vp, _vp]),
    "awq_oneshot_allreduce_selftest": (_i, [ctypes.POINTER(_vp), _vp, _vp, _i, _i, _i, ctypes.c_uint, _i, _vp, _vp]),
    "awq_oneshot_allreduce_f32": (_i, [ctypes.POINTER(_vp), _vp, _vp, _i, _vp, _i, _i, _i, _i, ctypes.c_uint, _i, _vp, _vp]),
    "awq_oneshot_allreduce_f32_selftest": (_i, [ctypes.POINTER(_vp), _vp, _vp, _i, _vp, _i, _i, _i, ctypes.c_uint, _i, _vp, _vp]),
    "awq_oneshot_set_spin_limit": (_i, [ctypes.c_uint]),
    "awq_w4a16_partial_cdna4": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "awq_w3a16_partial": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "awq_w3a16_mlp_gate_up_forward_workspace_bytes": (_sz, [_i, _i, _i]),
    "awq_w3a16_mlp_gate_up_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "awq_round_bias_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "awq_tune_set": (_i, [ctypes.c_char_p, _i]),
}


class AwqNativeError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AwqNativeError(
                f"{LIB_PATH} is missing: the MI355X HIP library has not been built "
                "(run `python -m llm_awq_amd.build`). There is no CPU/PyTorch fallback for the hot path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.awq_abi_version() != 1:
            raise AwqNativeError("libawq_cdna4.so ABI version mismatch")
        _lib = L
    return _lib


def check(status: int) -> None:
    if status != 0:
        L = lib()
        msg = L.awq_status_string(status).decode()
        if status == -8:
            msg += " " + L.awq_last_hip_error().decode()
        raise AwqNativeError(f"awq_cdna4 error {status}: {msg}")


def tune(**knobs) -> None:
    """awq_tune_set for each knob (tests, experiments and benchmarks only: opts this process in with AWQ_TUNING=1)."""
    os.environ["AWQ_TUNING"] = "1"
    for k, v in knobs.items():
        check(lib().awq_tune_set(k.encode(), int(v)))
