"""Prompts / decode batches of 9..255 rows: the skinny kernel against the prefill GEMM's single masked row tile
(+ split-K), timed from a hipGraph of 20 calls.   python tools/small_m_sweep.py > gpurun_out/small_m_sweep.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from llm_awq_amd import ops, synth

SHAPES = [(4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096), (8192, 8192), (8192, 1024), (11008, 4096)]
MS = (32, 64, 80, 96, 128, 160, 192, 224, 255)


def graph_time(fn, calls=20, iters=10):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(calls):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        for _ in range(iters):
            g.replay()
        b.record(s)
        torch.cuda.synchronize()
    return a.elapsed_time(b) / (iters * calls) * 1e3


def main():
    L = ops._capi.lib()
    print(f"{'K':>6} {'N':>6} {'M':>4} {'skinny us':>10} {'gemm us':>8} {'auto':>6} {'ratio':>6}")
    for K, N in SHAPES:
        w = synth.random_wq(K, N, dtype=torch.bfloat16, seed=1, keep_q=False)
        c4 = ops.repack_v2_to_cdna4(w["qweight"])
        szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
        for M in MS:
            x = torch.randn(M, K, device="cuda").bfloat16()
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            res = []
            for knob in (0, 2, 1):
                ops._capi.tune(gemm_small_m=knob)
                wsb = L.awq_w4a16_forward_cdna4_workspace_bytes(M, N, K)
                ws = torch.empty(max(wsb, 16) // 4, dtype=torch.float32, device="cuda")

                def run():
                    ops._capi.check(L.awq_w4a16_forward_cdna4(x.data_ptr(), c4.data_ptr(), w["scales"].data_ptr(), w["scaled_zeros"].data_ptr(),
                                                              szp.data_ptr(), None, out.data_ptr(), M, N, K, 128, 1,
                                                              ws.data_ptr() if wsb else None, wsb, torch.cuda.current_stream().cuda_stream))
                res.append(graph_time(run))
            ops._capi.tune(gemm_small_m=1)
            auto = "gemm" if abs(res[2] - res[1]) < abs(res[2] - res[0]) else "skinny"
            print(f"{K:6d} {N:6d} {M:4d} {res[0]:10.1f} {res[1]:8.1f} {auto:>6} {res[0] / res[1]:6.2f}", flush=True)


if __name__ == "__main__":
    main()
