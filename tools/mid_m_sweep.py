"""GPU experiment: the product forward (awq_w4a16_forward_cdna4, whatever kernel it routes to) per Llama-3-8B layer shape for
row counts between decode and prefill, over rotating weight copies (> the 256 MB Infinity Cache), against the two floors of the
shape: streaming its algorithmic bytes at 8 TB/s and its flops at 2.5 PFLOP/s.   python tools/mid_m_sweep.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402

SHAPES = [(4096, 6144, "qkv"), (4096, 4096, "o"), (4096, 28672, "gate+up"), (14336, 4096, "down")]
if os.environ.get("MID_SHAPES"):
    SHAPES = [s for s in SHAPES if s[2] in os.environ["MID_SHAPES"].split(",")]


def graph_time(fn, items, reps=3):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for it in items[:2]:
            fn(it)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for it in items:
                fn(it)
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            g.replay()
            e1.record(s)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
    return best * 1e3 / len(items)


def main():
    L = _capi.lib()
    Ms = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128, 255, 512, 1024, 2048]
    if os.environ.get("MID_TUNE"):  # knobs as key=value,key=value (AWQ_TUNING=1)
        _capi.tune(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ["MID_TUNE"].split(",")})
        print("# knobs:", os.environ["MID_TUNE"])
    dtype = torch.bfloat16
    print(f"{'shape':>8} {'K':>6} {'N':>6} {'M':>5} {'us':>9} {'TFLOP/s':>8} {'GB/s':>8} {'stream floor us':>16} {'mfma floor us':>14} {'x floor':>8}")
    for (K, N, name) in SHAPES:
        R = max(4, min(12, (320 << 20) // (N * K // 2) + 1))
        copies = []
        for i in range(R):
            w = synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False)
            copies.append(dict(qw=ops.repack_v2_to_cdna4(w["qweight"]), s=w["scales"], z=w["scaled_zeros"],
                               szp=ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)))
            del w
        for M in Ms:
            x = torch.randn(M, K, device="cuda").to(dtype)
            out = torch.empty(M, N, device="cuda", dtype=dtype)
            wsb = L.awq_w4a16_forward_cdna4_workspace_bytes(M, N, K)
            ws = torch.empty(max(wsb, 16) // 4, dtype=torch.float32, device="cuda")

            def fn(c):
                _capi.check(L.awq_w4a16_forward_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["s"].data_ptr(), c["z"].data_ptr(), c["szp"].data_ptr(), None,
                                                      out.data_ptr(), M, N, K, 128, 1, ws.data_ptr() if wsb else None, wsb,
                                                      torch.cuda.current_stream().cuda_stream))

            us = graph_time(fn, copies)
            by = N * K // 2 + 4 * (K // 128) * N + 2 * M * K + 2 * M * N
            fl = 2.0 * M * N * K
            f_stream, f_mfma = by / 8e12 * 1e6, fl / 2.5e15 * 1e6
            print(f"{name:>8} {K:6d} {N:6d} {M:5d} {us:9.1f} {fl / us / 1e6:8.1f} {by / us / 1e3:8.1f} {f_stream:16.1f} {f_mfma:14.1f} {us / max(f_stream, f_mfma):8.2f}",
                  flush=True)
        del copies
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
