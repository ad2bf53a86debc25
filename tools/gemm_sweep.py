"""GPU experiment: prefill GEMM TFLOP/s per Llama-3-8B layer shape and M, per kernel variant."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, nargs="+", default=[64, 256, 1024, 2048, 4096])
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--variants", type=int, nargs="+", default=[1, 2])
    args = ap.parse_args()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    print(f"device {torch.cuda.get_device_name(0)} dtype {args.dtype}")
    for (K, N) in [(4096, 4096), (4096, 6144), (4096, 14336), (14336, 4096)]:
        w = synth.random_wq(K, N, dtype=dtype, seed=1, keep_q=False)
        for M in args.m:
            x = torch.randn(M, K, device="cuda").to(dtype)
            for v in args.variants:
                _capi.tune(gemm_variant=v)
                for _ in range(3):
                    ops.gemm(x, w["qweight"], w["scales"], w["scaled_zeros"])
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                it = 10
                e0.record()
                for _ in range(it):
                    ops.gemm(x, w["qweight"], w["scales"], w["scaled_zeros"])
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / it
                tf = 2.0 * M * N * K / us / 1e6
                print(f"K={K:6d} N={N:6d} M={M:5d} variant={v}  {us:9.1f} us  {tf:7.1f} TFLOP/s  {tf/25:5.1f}% of 2.5PF", flush=True)
    _capi.tune(gemm_variant=0)


if __name__ == "__main__":
    main()
