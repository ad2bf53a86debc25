"""GPU experiment: batched decode (M = 2..4) on the streaming kernel -- ring depth / wave count forced through the knobs gemvd_d / gemvd_waves
against pick_dma's choice (which sizes the ring so that the SAME number of blocks stays co-resident as at M = 1, i.e. shrinks it to depth 1
once the staged x takes the LDS).  Needs AWQ_TUNING=1.  usage: python tools/decode_m_ring_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402
from tools.gemvc_sweep import time_graph  # noqa: E402


def main():
    L = _capi.lib()
    dtype = torch.bfloat16
    shapes = [("qkv", 4096, 6144, 0), ("gate/up", 4096, 28672, 2), ("down", 14336, 4096, 0)]
    print("# shape M  auto_us | (waves, d): us ...")
    for (name, K, N, epi) in shapes:
        R = max(10, min(40, (700 << 20) // (N * K // 2)))
        copies = []
        for i in range(R):
            w = synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False)
            szh, exact = ops.pack_szh_cdna4(w["scales"], w["scaled_zeros"], K)  # (timing only: gate/up rows taken as an interleaved pair)
            copies.append(dict(qw=ops.repack_v2_to_cdna4(w["qweight"]), szh=szh))
            del w
        for M in (1, 2, 3, 4):
            x = torch.randn(M, K, device="cuda").to(dtype)
            out = torch.empty(M, N // 2 if epi else N, device="cuda", dtype=dtype)

            def fn(c):
                rc = L.awq_w4a16_decode_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["szh"].data_ptr(), None, out.data_ptr(), M, N, K, 128, 1, epi,
                                              torch.cuda.current_stream().cuda_stream)
                if rc != 0:
                    raise RuntimeError(rc)

            _capi.tune(decode_skinny_from=9, gemvd_waves=0, gemvd_d=0)
            res = [f"auto {time_graph(fn, copies):6.2f}"]
            for waves in ((4, 8) if K == 4096 else (8, 16)):
                for d in (1, 2, 4, 7, 8):
                    _capi.tune(gemvd_waves=waves, gemvd_d=d)
                    try:
                        fn(copies[0])
                        torch.cuda.synchronize()
                        res.append(f"({waves},{d}) {time_graph(fn, copies):6.2f}")
                    except RuntimeError:
                        res.append(f"({waves},{d})   --  ")
            print(f"{name:8s} M={M}  " + "  ".join(res), flush=True)
        del copies
        torch.cuda.empty_cache()
    _capi.tune(decode_skinny_from=0, gemvd_waves=0, gemvd_d=0)


if __name__ == "__main__":
    main()
