"""Prefill GEMM with few output tiles: split-K (awq_gemm_v4n.hip) against the unsplit kernel, HIP-event timed.
   python tools/splitk_sweep.py > gpurun_out/splitk_sweep.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from llm_awq_amd import ops, synth

SHAPES = [(4096, 4096), (14336, 4096), (4096, 6144), (4096, 14336), (8192, 8192), (8192, 1024), (11008, 4096)]
MS = (256, 512, 768, 1024, 1536)


def time_it(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    L = ops._capi.lib()
    forced = (2, 3, 4, 6, 8, 12, 16)
    print(f"{'K':>6} {'N':>6} {'M':>5} {'unsplit':>8} {'auto':>7} {'ws MiB':>7} | " + " ".join(f"ks={d:<4d}" for d in forced) + " | TF unsplit -> auto")
    for K, N in SHAPES:
        w = synth.random_wq(K, N, dtype=torch.bfloat16, seed=1, keep_q=False)
        c4 = ops.repack_v2_to_cdna4(w["qweight"])
        szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
        for M in MS:
            x = torch.randn(M, K, device="cuda").bfloat16()
            out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            st = torch.cuda.current_stream().cuda_stream

            def timed(knob):
                ops._capi.tune(gemm_splitk=knob)
                wsb = L.awq_w4a16_forward_cdna4_workspace_bytes(M, N, K)
                if knob and not wsb:
                    return float("nan"), 0
                ws = torch.empty(max(wsb, 16) // 4, dtype=torch.float32, device="cuda")

                def run():
                    ops._capi.check(L.awq_w4a16_forward_cdna4(x.data_ptr(), c4.data_ptr(), w["scales"].data_ptr(),
                                                              w["scaled_zeros"].data_ptr(), szp.data_ptr(), None, out.data_ptr(), M, N, K,
                                                              128, 1, ws.data_ptr() if wsb else None, wsb, st))
                return time_it(run), wsb
            t0, _ = timed(0)
            t1, wsb = timed(1)
            tf = [timed(d)[0] for d in forced]
            ops._capi.tune(gemm_splitk=1)
            fl = 2.0 * M * N * K
            print(f"{K:6d} {N:6d} {M:5d} {t0:8.1f} {t1:7.1f} {wsb / 2**20:7.1f} | " + " ".join(f"{t:7.1f}" for t in tf) +
                  f" | {fl / t0 / 1e6:7.1f} -> {fl / (t1 if t1 == t1 else t0) / 1e6:7.1f}")


if __name__ == "__main__":
    main()
