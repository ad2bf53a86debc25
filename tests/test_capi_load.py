"""not-gpu: libawq_cdna4.so loads and exports every symbol include/awq_cdna4.h declares; argument
validation (no kernel launch needed) returns the documented codes."""
import ctypes
import os
import re

from llm_awq_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "awq_cdna4.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(awq_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 11
    L = ctypes.CDLL(_capi.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n
        assert n in _capi.SIGNATURES, f"{n} missing from the ctypes signature table"
    assert _capi.lib().awq_abi_version() == 1


def test_argument_validation_without_gpu():
    L = _capi.lib()
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    ok = (p16, p16, p16, p16, p16)
    assert L.awq_w4a16_gemv(*ok, 1, 64, 256, 64, 0, None) == -2   # group size
    assert L.awq_w4a16_gemv(*ok, 0, 64, 256, 128, 0, None) == -1  # batch
    assert L.awq_w4a16_gemv(*ok, 17, 64, 256, 128, 0, None) == -1
    assert L.awq_w4a16_gemv(*ok, 1, 64, 256, 128, 7, None) == -3  # dtype
    assert L.awq_w4a16_gemv(*ok, 1, 60, 256, 128, 0, None) == -4  # n % 8
    assert L.awq_w4a16_gemv(*ok, 1, 64, 200, 128, 0, None) == -4  # k % 128
    assert L.awq_w4a16_gemv(p16 + 2, p16, p16, p16, p16, 1, 64, 256, 128, 0, None) == -5
    assert L.awq_w4a16_gemv(None, p16, p16, p16, p16, 1, 64, 256, 128, 0, None) == -6
    assert L.awq_w4a16_gemm(*ok, 32, 64, 256, 64, 1, None, 0, None) == -2
    assert L.awq_unpack_v2(p16, p16, 6, 64, None) == -4
    assert b"batch size" in L.awq_status_string(-1) and b"group size" in L.awq_status_string(-2)
    assert L.awq_w4a16_gemm_workspace_bytes(2048, 4096, 4096) >= 0


def test_engine_module_exports():
    import llm_awq_amd
    eng = llm_awq_amd.load_engine()
    for name in ("gemv_forward_cuda_new", "gemm_forward_cuda_new"):
        assert callable(getattr(eng, name))
    import sys
    llm_awq_amd.install_as_awq_inference_engine()
    assert sys.modules["awq_inference_engine"] is eng
