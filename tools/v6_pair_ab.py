"""GPU A/B: K split over pairs of 256 x 256 blocks inside one launch (gemm_cdna4_v6_pair_kernel, knob gemm_v6_pair) against the 256 x 128 blocks it replaces,
on the shapes whose 256-wide tiles fill at most half the chip (down_proj of Llama-3-8B at 1536 .. 2048 rows).  Correctness first (against the unsplit
kernels: same products, another fp32 association), then us per call over rotating weight copies, alternating.  usage: AWQ_TUNING=1 python tools/v6_pair_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402
from tools.gemvc_sweep import time_graph  # noqa: E402


def main():
    dtype = torch.bfloat16
    shapes = [("down", 14336, 4096), ("o", 4096, 4096)] if os.environ.get("PAIR_O") else [("down", 14336, 4096)]
    if os.environ.get("PAIR_O"):
        _capi.tune(gemm_v6_pair_min_nit=32)
    rows = [int(a) for a in sys.argv[1:]] or [2048, 1792, 1536]
    for (name, K, N) in shapes:
        copies = []
        for i in range(6):
            w = synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False)
            copies.append(dict(qw=ops.repack_v2_to_cdna4(w["qweight"]), s=w["scales"], z=w["scaled_zeros"], szp=ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)))
            del w
        for M in rows:
            x = torch.randn(M, K, device="cuda").to(dtype)
            bias = (torch.randn(N, device="cuda") * 0.02).to(dtype)
            c = copies[0]
            _capi.tune(gemm_v6_pair=0)
            ref = ops.gemm_cdna4(x, c["qw"], c["s"], c["z"], None, c["szp"])
            refb = ops.gemm_cdna4(x, c["qw"], c["s"], c["z"], bias, c["szp"])
            _capi.tune(gemm_v6_pair=1)
            for lead in (0, 1, 3):
                _capi.tune(gemm_v6_pair_lead=lead)
                for rep in range(3):
                    y = ops.gemm_cdna4(x, c["qw"], c["s"], c["z"], None, c["szp"])
                    yb = ops.gemm_cdna4(x, c["qw"], c["s"], c["z"], bias, c["szp"])
                    torch.cuda.synchronize()
                    for (a, b) in ((y, ref), (yb, refb)):
                        rel = ((a.float() - b.float()).norm() / b.float().norm()).item()
                        same = (a == b).float().mean().item()
                        assert rel < 1e-3 and same > 0.9 and bool(torch.isfinite(a.float()).all()), (M, lead, rep, rel, same)
            print(f"{name} M={M}: pair == unsplit within fp32 re-association (rel {rel:.2e}, identical {same:.4f})", flush=True)
            flops = 2.0 * M * N * K
            for rnd in range(2):
                res = []
                for (pair, lead) in ((0, 0), (1, 0), (1, 1), (1, 2), (1, 4)):
                    _capi.tune(gemm_v6_pair=pair, gemm_v6_pair_lead=lead)
                    us = time_graph(lambda cc: ops.gemm_cdna4(x, cc["qw"], cc["s"], cc["z"], None, cc["szp"]), copies, reps=4)
                    res.append(f"{'pair lead ' + str(lead) if pair else '256x128   '} {us:7.1f} us {flops / us / 1e6:7.1f} TF")
                print(f"{name} M={M}  " + "  |  ".join(res), flush=True)
    _capi.tune(gemm_v6_pair=1, gemm_v6_pair_lead=1)


if __name__ == "__main__":
    main()
