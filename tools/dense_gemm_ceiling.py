"""GPU experiment: what does the vendor's dense bf16 GEMM (torch.matmul -> hipBLASLt) reach on the SAME shapes on this box?
A dense GEMM has no dequantisation work at all and is tuned by the vendor: its TFLOP/s under the chip's power-limited clock is the
practical ceiling the W4A16 prefill kernels can be compared with (the 2.5 PFLOP/s nominal peak assumes 2.4 GHz, which a chip
running its matrix pipes on random data does not sustain).  usage: python tools/dense_gemm_ceiling.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import ops, synth  # noqa: E402


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


def main():
    dtype = torch.bfloat16
    for M in (2048, 4096):
        tot_d = tot_q = fl = 0.0
        for (K, N) in [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]:
            w = synth.random_wq(K, N, dtype=dtype, seed=1, keep_q=False)
            c4 = ops.repack_v2_to_cdna4(w["qweight"])
            szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
            W = ops.dequant_cdna4(c4, w["scales"], w["scaled_zeros"])  # [N, K] bf16: the same weights, dense
            x = torch.randn(M, K, device="cuda").to(dtype)
            us_d = timeit(lambda: torch.matmul(x, W.t()))
            us_q = timeit(lambda: ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], None, szp))
            f = 2.0 * M * N * K
            tot_d += us_d
            tot_q += us_q
            fl += f
            print(f"M={M:5d} K={K:6d} N={N:6d}  dense bf16 (hipBLASLt) {us_d:8.1f} us {f / us_d / 1e6:7.1f} TF ({f / us_d / 1e6 / 25:4.1f}%)   "
                  f"W4A16 {us_q:8.1f} us {f / us_q / 1e6:7.1f} TF ({f / us_q / 1e6 / 25:4.1f}%)   W4A16 / dense = {us_d / us_q:.3f}", flush=True)
        print(f"layer total M={M}: dense {tot_d:.1f} us = {fl / tot_d / 1e6 / 25:.1f}% of 2.5 PF; W4A16 {tot_q:.1f} us = {fl / tot_q / 1e6 / 25:.1f}%; ratio {tot_d / tot_q:.3f}")


if __name__ == "__main__":
    main()
