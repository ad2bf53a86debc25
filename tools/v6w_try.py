"""GPU experiment (AWQ_PROBES=1 build): the v6 prefill loop on the PLANNED 32-row interleave "cdna4w" (csrc/awq_gemm_v6w.hip; layout:
oracle/awq_oracle.py::pack_cdna4w, pinned by tests/test_cdna4w_layout.py) against the product path on the same weights.
  1. correctness: small shapes against the oracle forward (tests/helpers.check_forward), then full shapes against the product GEMM;
  2. time per call at M = 2048 / 4096 on the Llama-3-8B shapes (the product path next to it).
The cdna4w buffer is packed on the HOST by the oracle (this is an experiment: there is no device repacker yet).
usage: AWQ_PROBES=1 python -c "import llm_awq_amd.build as b; b.build_all(force=True)";  AWQ_TUNING=1 python tools/v6w_try.py [quick]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops  # noqa: E402
from oracle import awq_oracle as O  # noqa: E402  (experiment script: the oracle packs the planned layout and checks the result)


def timeit(fn, it=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


def pack_cdna4w_torch(q: torch.Tensor) -> torch.Tensor:
    """oracle.cdna4w_position evaluated with torch on the device (the numpy oracle takes ~15 s per 4096 x 4096 on the host: too slow to hold
    a GPU box for); checked against the oracle on a small matrix in main()."""
    N, K = q.shape
    dev = q.device
    n = torch.arange(N, device=dev, dtype=torch.int64)[:, None]
    k = torch.arange(K, device=dev, dtype=torch.int64)[None, :]
    npair, c = n // 32, n % 32
    nq, j = c // 4, c % 4
    kt, kk = k // 64, k % 64
    a, r16 = kk // 16, kk % 16
    kb, e = r16 // 8, r16 % 8
    th, rr = e // 4, e % 4
    lane = 32 * kb + 4 * nq + rr
    i = 2 * th + (j >> 1)
    p = (i + 4 * (j & 1)).expand(N, K)
    word = (((npair * (K // 64) + kt) * 64 + lane) * 4 + a).expand(N, K)
    val = (q.to(torch.int64) & 0xF) << (4 * p)
    out = torch.zeros(N * K // 8, dtype=torch.int64, device=dev)
    out.index_put_((word.reshape(-1),), val.reshape(-1), accumulate=True)   # disjoint bit fields: the sum is the OR
    out = torch.where(out >= 2 ** 31, out - 2 ** 32, out).to(torch.int32)
    return out.view(torch.int16).reshape(N // 4, K)


def main():
    L = _capi.lib()
    probe = getattr(L, "awq_probe_gemm_cdna4w", None)
    if probe is None:
        raise SystemExit("this library was not built with AWQ_PROBES=1")
    probe.restype = ctypes.c_int
    probe.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
    from tests.helpers import check_forward, make_case
    quick = len(sys.argv) > 1

    def run(x, qww, szp, bias, N, K):
        out = torch.empty(x.shape[0], N, dtype=x.dtype, device="cuda")
        _capi.check(probe(x.data_ptr(), qww.data_ptr(), szp.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(),
                          x.shape[0], N, K, 0 if x.dtype == torch.float16 else 1, torch.cuda.current_stream().cuda_stream))
        return out

    qs = torch.randint(0, 16, (96, 384), dtype=torch.uint8, device="cuda")
    assert np.array_equal(pack_cdna4w_torch(qs).cpu().numpy(), O.pack_cdna4w(qs.cpu().numpy())), "device-side packer != oracle"
    # the HIP repacker v2 -> cdna4w (awq_probe_repack_v2_to_cdna4w) against both
    rep = getattr(L, "awq_probe_repack_v2_to_cdna4w")
    rep.restype = ctypes.c_int
    rep.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    for (N, K) in ((96, 384), (4096, 4096)):
        qv = torch.randint(0, 16, (N, K), dtype=torch.uint8, device="cuda")
        v2 = ops.pack_v2(qv)
        dst = torch.empty_like(v2)
        _capi.check(rep(v2.data_ptr(), dst.data_ptr(), N, K, torch.cuda.current_stream().cuda_stream))
        assert torch.equal(dst, pack_cdna4w_torch(qv)), ("HIP repacker != packer", N, K)
    print("repack_v2_to_cdna4w ok", flush=True)

    # ---- 1. small shapes against the oracle ----
    for dtype in (torch.bfloat16, torch.float16):
        for (N, K, M) in ((256, 128, 256), (512, 1024, 300), (1280, 512, 777)):
            c = make_case(N, K, dtype, seed=N + K + M, M=M, bias=True)
            qww = pack_cdna4w_torch(torch.from_numpy(c["q"]).cuda())
            szp = ops.pack_sz_cdna4(c["scales"].cuda(), c["scaled_zeros"].cuda(), K)
            for b in (None, c["bias"]):
                y = run(c["x"].cuda(), qww, szp, b.cuda() if b is not None else None, N, K)
                check_forward(y.cpu(), c["x"], c["q"], c["scales"], c["scaled_zeros"], dtype, bias=b)
            print("oracle ok", dtype, (N, K, M), flush=True)

    # ---- 1b. decode on the same weights at slab granularity (awq_probe_decode_cdna4w) against the oracle and the product decode ----
    dec = getattr(L, "awq_probe_decode_cdna4w")
    dec.restype = ctypes.c_int
    dec.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 6 + [ctypes.c_void_p]

    def run_dec(x, qww, szh, bias, N, K, waves, ring):
        out = torch.empty(x.shape[0], N, dtype=x.dtype, device="cuda")
        _capi.check(dec(x.data_ptr(), qww.data_ptr(), szh.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(),
                        x.shape[0], N, K, 0 if x.dtype == torch.float16 else 1, waves, ring, torch.cuda.current_stream().cuda_stream))
        return out

    for dtype in (torch.bfloat16, torch.float16):
        for (N, K, waves, ring) in ((64, 1280, 8, 2), (256, 4096, 8, 2), (64, 14336, 16, 1), (2048, 4096, 4, 7), (96, 11008, 4, 4)):
            c = make_case(N, K, dtype, seed=N + K, M=8, bias=True)
            qww = pack_cdna4w_torch(torch.from_numpy(c["q"]).cuda())
            szh, exact = ops.pack_szh_cdna4(c["scales"].cuda(), c["scaled_zeros"].cuda(), K)
            assert exact
            for M in (1, 3, 8):
                xx = c["x"][:M].contiguous()
                for b in (None, c["bias"]):
                    y = run_dec(xx.cuda(), qww, szh, b.cuda() if b is not None else None, N, K, waves, ring)
                    check_forward(y.cpu(), xx, c["q"], c["scales"], c["scaled_zeros"], dtype, bias=b)
            print("decode oracle ok", dtype, (N, K, waves, ring), flush=True)
    from tools.gemvc_sweep import time_graph
    from llm_awq_amd import synth as _synth
    print("# decode M = 1, graph over rotating copies: shape  product_us  cdna4w_us")
    for (name, K, N, waves, ring) in (("o", 4096, 4096, 8, 2), ("qkv", 4096, 6144, 8, 2), ("down", 14336, 4096, 16, 1)):
        copies = []
        for ci in range(max(10, min(40, (700 << 20) // (N * K // 2)))):
            w = _synth.random_wq(K, N, dtype=torch.bfloat16, seed=ci, keep_q=True)
            szh, exact = ops.pack_szh_cdna4(w["scales"], w["scaled_zeros"], K)
            copies.append(dict(c4=ops.repack_v2_to_cdna4(w["qweight"]), cw=pack_cdna4w_torch(w["q"]), szh=szh))
            del w
        x1 = torch.randn(1, K, device="cuda").bfloat16()
        o1 = torch.empty(1, N, device="cuda", dtype=torch.bfloat16)
        st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
        t_prod = time_graph(lambda c: _capi.check(L.awq_w4a16_decode_cdna4(x1.data_ptr(), c["c4"].data_ptr(), c["szh"].data_ptr(), None, o1.data_ptr(), 1, N, K, 128, 1, 0, st())), copies)
        t_new = time_graph(lambda c: _capi.check(dec(x1.data_ptr(), c["cw"].data_ptr(), c["szh"].data_ptr(), None, o1.data_ptr(), 1, N, K, 1, waves, ring, st())), copies)
        print(f"{name:6s} {K:6d} {N:6d}  {t_prod:7.2f} {t_new:7.2f}", flush=True)
        del copies
        torch.cuda.empty_cache()

    # ---- 2. Llama-3-8B shapes: equality with the product path + time ----
    from llm_awq_amd import synth
    print("# shape K N M  product_us  v6w_us  TF_product TF_v6w  identical_fraction")
    for (name, K, N) in (("o", 4096, 4096), ("qkv", 4096, 6144), ("down", 14336, 4096)) + (() if quick else (("gate+up", 4096, 28672),)):
        w = synth.random_wq(K, N, dtype=torch.bfloat16, seed=K + N, keep_q=True)
        qww = pack_cdna4w_torch(w["q"])
        c4 = ops.repack_v2_to_cdna4(w["qweight"])
        szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
        for M in (2048, 4096):
            x = torch.randn(M, K, device="cuda").bfloat16()
            y0 = ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], None, szp)
            y1 = run(x, qww, szp, None, N, K)
            same = (y0 == y1).float().mean().item()
            rel = ((y0.float() - y1.float()).norm() / y0.float().norm()).item()
            t0 = timeit(lambda: ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], None, szp))
            t1 = timeit(lambda: run(x, qww, szp, None, N, K))
            tf = lambda us: 2.0 * M * N * K / us / 1e6  # noqa: E731
            print(f"{name:8s} {K:6d} {N:6d} {M:5d}  {t0:9.1f} {t1:9.1f}  {tf(t0):7.1f} {tf(t1):7.1f}  {same:.5f} rel {rel:.2e}", flush=True)


if __name__ == "__main__":
    main()
