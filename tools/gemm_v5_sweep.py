"""GPU experiment: prefill GEMM v5 (weights streamed into registers per wave, awq_gemm_v5.hip) against the shipped v4 / v4n plan:
same inputs, outputs compared, time per Llama-3-8B layer shape and M.  usage: python tools/gemm_v5_sweep.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402


def timeit(fn, it=8):
    for _ in range(2):
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
    Ms = [int(a) for a in sys.argv[1:]] or [2048, 4096, 1024, 3072, 8192]
    dtype = torch.bfloat16
    tot = {}
    for (K, N) in [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]:
        w = synth.random_wq(K, N, dtype=dtype, seed=1, keep_q=False)
        c4 = ops.repack_v2_to_cdna4(w["qweight"])
        szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
        for M in Ms:
            x = torch.randn(M, K, device="cuda").to(dtype)
            res = {}
            fn = lambda: ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], None, szp)
            for rep in range(3):  # interleaved repeats, best of three: the box's clock drifts between launches
                for tag, knob in (("v4 plan", 0), ("v5 256-row", 2), ("v5 128-row", 3)):
                    _capi.tune(gemm_v5=knob)
                    y = fn()
                    us = timeit(fn)
                    if tag not in res or us < res[tag][0]:
                        res[tag] = (us, y)
            for tag in res:
                tot[(M, tag)] = tot.get((M, tag), 0.0) + res[tag][0]
            _capi.tune(gemm_v5=0)
            ref = res["v4 plan"][1].float()
            line = f"K={K:6d} N={N:6d} M={M:5d}"
            for tag, (us, y) in res.items():
                tf = 2.0 * M * N * K / us / 1e6
                rel = ((y.float() - ref).norm() / ref.norm()).item()
                same = (y.float() == ref).float().mean().item()
                line += f" | {tag}: {us:8.1f} us {tf:7.1f} TF ({tf / 25:4.1f}%) rel {rel:.1e} same {same:.4f}"
            print(line, flush=True)
    for M in Ms:
        fl = sum(2.0 * M * K * N for (K, N) in [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)])
        print(f"layer total M={M}: " + "  ".join(f"{tag}: {tot[(M, tag)]:.1f} us = {fl / tot[(M, tag)] / 1e6 / 25:.1f}% of 2.5 PF" for tag in ("v4 plan", "v5 256-row", "v5 128-row")) + f"   best-of per shape: n/a")


if __name__ == "__main__":
    main()
