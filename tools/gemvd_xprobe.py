"""GPU experiment (AWQ_PROBES=1 build, e.g. AWQ_CDNA4_LIB=llm_awq_amd/lib/libawq_cdna4_probes.so): what do the x and scale staging DMAs cost the decode
launches?  Every block of the streaming kernel stages its own x slices (M x 8 KiB per slab at K = 4096) and scale dwords by LDS-DMA beside 32 KiB of
weights; the probes drop them (wrong results, timing only).  Llama-3-8B shapes, one row, graph over rotating weight copies.
usage: AWQ_TUNING=1 AWQ_CDNA4_LIB=... python tools/gemvd_xprobe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402
from tools.gemvc_sweep import time_graph  # noqa: E402


def main():
    L = _capi.lib()
    dtype = torch.bfloat16
    for (name, K, N, epi) in [("gate_up", 4096, 28672, 2), ("down", 14336, 4096, 0), ("qkv", 4096, 6144, 0), ("o", 4096, 4096, 0)]:
        R = max(10, min(40, (900 << 20) // (N * K // 2)))
        copies = []
        for i in range(R):
            w = synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False)
            copies.append(dict(qw=ops.repack_v2_to_cdna4(w["qweight"]), szh=ops.pack_szh_cdna4(w["scales"], w["scaled_zeros"], K)[0]))
            del w
        x = torch.randn(1, K, device="cuda").to(dtype)
        out = torch.empty(1, N // 2 if epi else N, device="cuda", dtype=dtype)

        def fn(c):
            _capi.check(L.awq_w4a16_decode_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["szh"].data_ptr(), None, out.data_ptr(), 1, N, K, 128, 1, epi,
                                                 torch.cuda.current_stream().cuda_stream))

        res = {}
        for rnd in range(2):
            for probe, label in ((0, "full"), (4, "no x DMA"), (8, "no sz DMA"), (12, "no x, no sz"), (1, "no math"), (5, "no math, no x"), (3, "neither (stream, math)"), (15, "nothing")):
                _capi.tune(gemvd_probe=probe)
                res.setdefault(label, []).append(time_graph(fn, copies, reps=4))
        _capi.tune(gemvd_probe=0)
        print(f"{name:8s} K={K:6d} N={N:6d}: " + " | ".join(f"{k} {min(v):6.2f}" for k, v in res.items()), flush=True)
        del copies
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
