"""GPU: the drop-in entry points (awq_inference_engine.gemv_forward_cuda_new / gemm_forward_cuda_new on raw reference-layout
buffers) for fp16 and bf16, with the lazy cdna4 cache on (default) and off (reference-layout kernels)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import synth  # noqa: E402
from llm_awq_amd.qmodule import load_engine  # noqa: E402


def timeit(fn, items, reps=3):
    for it in items[:2]:
        fn(it)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in items:
            fn(it)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / len(items))
    return best


def main():
    eng = load_engine()
    shapes = [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]
    for dtype in (torch.bfloat16, torch.float16):
        for cache in (True, False):
            eng.cdna4_cache_clear()
            eng.cdna4_cache_enable(cache)
            for (K, N) in shapes:
                R = max(4, min(12, (400 << 20) // (N * K // 2)))
                ws = [synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False) for i in range(R)]
                x1 = torch.randn(1, K, device="cuda").to(dtype)
                us = timeit(lambda w: eng.gemv_forward_cuda_new(x1, w["qweight"], w["scales"], w["scaled_zeros"], 1, N, K, 128), ws)
                by = N * K // 2 + 4 * (K // 128) * N
                line = f"{str(dtype)[6:]:9s} cache={int(cache)} K={K:6d} N={N:6d}  decode M=1 {us:7.2f} us {by / us / 1e3:7.1f} GB/s"
                for M in (64, 2048, 4096):
                    xm = torch.randn(M, K, device="cuda").to(dtype)
                    us = timeit(lambda w: eng.gemm_forward_cuda_new(xm, w["qweight"], w["scales"], w["scaled_zeros"]), ws[:3])
                    line += f" | M={M} {us:8.1f} us {2.0 * M * N * K / us / 1e6:7.1f} TF"
                print(line, flush=True)
                del ws
                torch.cuda.empty_cache()
    eng.cdna4_cache_enable(True)


if __name__ == "__main__":
    main()
