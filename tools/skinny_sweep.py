"""GPU experiment: forward_cdna4 latency across M for the Llama-3-8B shapes (decode fast path, skinny kernel, 128x128, GEMM v3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import ops, synth  # noqa: E402

for (K, N) in [(4096, 4096), (4096, 6144), (4096, 28672), (14336, 4096)]:
    w = synth.random_wq(K, N, dtype=torch.bfloat16, seed=1, keep_q=False)
    c4 = ops.repack_v2_to_cdna4(w["qweight"])
    szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
    for M in (1, 8, 9, 16, 32, 48, 64, 65, 128, 256):
        x = torch.randn(M, K, device="cuda").bfloat16()
        for _ in range(3):
            ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], None, szp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 20
        e0.record()
        for _ in range(it):
            ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], None, szp)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / it
        print(f"K={K:6d} N={N:6d} M={M:4d}  {us:8.1f} us  (weights L2/MALL-warm: same buffer every call)", flush=True)

# reference (v2) layout, fp16: skinny_v2_kernel (auto) against the 128 x 128 tile kernel (gemm_variant=1)
from llm_awq_amd import _capi  # noqa: E402

print("== v2 layout, fp16: ops.forward (auto = skinny_v2 for 9..255) vs the 128x128 tile kernel ==")
for (K, N) in [(4096, 4096), (4096, 28672), (14336, 4096)]:
    w = synth.random_wq(K, N, dtype=torch.float16, seed=1, keep_q=False)
    for M in (9, 16, 32, 64, 100, 128, 255):
        x = torch.randn(M, K, device="cuda").half()
        res = []
        for variant in (0, 1):
            _capi.tune(gemm_variant=variant)
            for _ in range(3):
                ops.gemm(x, w["qweight"], w["scales"], w["scaled_zeros"])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            it = 20
            e0.record()
            for _ in range(it):
                ops.gemm(x, w["qweight"], w["scales"], w["scaled_zeros"])
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / it)
        _capi.tune(gemm_variant=0)
        print(f"K={K:6d} N={N:6d} M={M:4d}  skinny_v2 {res[0]:8.1f} us   128x128 {res[1]:8.1f} us", flush=True)
