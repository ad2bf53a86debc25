"""GPU A/B of the prefill launches whose 256-wide tiles fill at most half the chip (Llama-3-8B at M rows: down_proj 14336 -> 4096, o_proj 4096 -> 4096, and the
gate/up launch 4096 -> 2 x 14336 whose last 128 column tiles follow three full rounds): 256 x 128 blocks (knob gemm_v6_pair = 0) against the SYMMETRIC block
pairs of round 5 (each block half of K and half of the tile's rows; gemm_v6_pair_min_nit 64 = down_proj only, 32 = the K = 4096 launches too), with the
T-typed sz_packed and with the layer's sz_half side buffer (f16-mantissa dequant), qkv (192-wide blocks) beside them.  Correctness against the unsplit
kernels first, then us per launch over rotating weight copies, alternating.  usage: AWQ_TUNING=1 python tools/v6_pair_ab.py [rows ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402
from llm_awq_amd.fused_mlp import interleave_gate_up  # noqa: E402
from tools.gemvc_sweep import time_graph  # noqa: E402


def make(K, N, seed, dtype, fused=False):
    if fused:
        g = synth.random_wq(K, N // 2, dtype=dtype, seed=seed, keep_q=False)
        u = synth.random_wq(K, N // 2, dtype=dtype, seed=seed + 100, keep_q=False)
        q, s, z = interleave_gate_up(g["qweight"], u["qweight"], g["scales"], u["scales"], g["scaled_zeros"], u["scaled_zeros"])
    else:
        w = synth.random_wq(K, N, dtype=dtype, seed=seed, keep_q=False)
        q, s, z = w["qweight"], w["scales"], w["scaled_zeros"]
    szh, exact = ops.pack_szh_cdna4(s, z, K)
    assert exact
    return dict(qw=ops.repack_v2_to_cdna4(q), s=s, z=z, szp=ops.pack_sz_cdna4(s, z, K), szh=szh)


def main():
    dtype = torch.bfloat16
    rows = [int(a) for a in sys.argv[1:]] or [2048]
    shapes = [("down", 14336, 4096, False), ("o", 4096, 4096, False), ("gate_up", 4096, 28672, True), ("qkv", 4096, 6144, False)]
    total = {}
    for (name, K, N, fused) in shapes:
        copies = [make(K, N, 10 * i + 1, dtype, fused) for i in range(4 if fused else 6)]

        def run(cc, x, szh):
            if fused:
                return ops.mlp_gate_up_forward_cdna4(x, cc["qw"], cc["szp"], cc["szh"] if szh else None)
            return ops.gemm_cdna4(x, cc["qw"], cc["s"], cc["z"], None, cc["szp"], sz_half=cc["szh"] if szh else None)

        for M in rows:
            x = torch.randn(M, K, device="cuda").to(dtype)
            c = copies[0]
            _capi.tune(gemm_v6_pair=0)
            ref = run(c, x, False)
            _capi.tune(gemm_v6_pair=1)
            for nit in (64, 32):
                _capi.tune(gemm_v6_pair_min_nit=nit)
                for szh in (False, True):
                    for rep in range(2):
                        y = run(c, x, szh)
                        torch.cuda.synchronize()
                        rel = ((y.float() - ref.float()).norm() / ref.float().norm()).item()
                        same = (y == ref).float().mean().item()
                        assert rel < 1e-3 and same > 0.95 and bool(torch.isfinite(y.float()).all()), (name, M, nit, szh, rep, rel, same)
            flops = 2.0 * M * N * K
            for rnd in range(2):
                res = []
                for (label, pair, nit, szh) in (("256x128      ", 0, 64, False), ("pair>=64     ", 1, 64, False), ("pair>=32     ", 1, 32, False),
                                                ("pair>=64 szh ", 1, 64, True), ("pair>=32 szh ", 1, 32, True)):
                    _capi.tune(gemm_v6_pair=pair, gemm_v6_pair_min_nit=nit)
                    us = time_graph(lambda cc: run(cc, x, szh), copies, reps=4)
                    res.append(f"{label}{us:7.1f} us {flops / us / 1e6 / 25:5.1f}%")
                    total.setdefault((M, label), []).append(us)
                print(f"{name:8s} M={M}  " + " | ".join(res), flush=True)
        del copies
        torch.cuda.empty_cache()
    _capi.tune(gemm_v6_pair=1, gemm_v6_pair_min_nit=64)
    for M in rows:
        fl = sum(2.0 * M * N * K for (_n, K, N, _f) in shapes)
        print(f"layer M={M} (sum of the four launches, best of the rounds): " +
              " | ".join(f"{lab}{s:8.1f} us {fl / s / 1e6 / 25:5.1f}%" for lab, s in layer_sums(total, M, len(shapes))))

def layer_sums(total, M, nshapes):
    out = []
    for (mm, lab), v in total.items():
        if mm != M:
            continue
        # v = [shape0 rnd0, shape0 rnd1, shape1 rnd0, ...]: best round per shape
        per = [min(v[2 * i: 2 * i + 2]) for i in range(nshapes)]
        out.append((lab, sum(per)))
    return out


if __name__ == "__main__":
    main()
