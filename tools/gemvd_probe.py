"""GPU experiment (build with AWQ_PROBES=1): where does a decode launch's time go?  The LDS-DMA streaming kernel with its timing
probes: full / stream only (no math) / math only (no weight DMA, no waits) / neither (launch + x, scales + reduce + store), per
Llama-3-8B decode shape, graph of launches over rotating weight copies.  usage: python tools/gemvd_probe.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402
from tools.gemvc_sweep import algo_bytes, time_graph  # noqa: E402


def main():
    L = _capi.lib()
    dtype = torch.bfloat16
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    for (K, N, epi) in [(4096, 4096, 0), (4096, 6144, 0), (14336, 4096, 0), (4096, 28672, 1), (4096, 28672, 2)]:
        R = max(10, min(48, (900 << 20) // (N * K // 2)))
        copies = []
        for i in range(R):
            w = synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False)
            copies.append(dict(qw=ops.repack_v2_to_cdna4(w["qweight"]), szh=ops.pack_szh_cdna4(w["scales"], w["scaled_zeros"], K)[0],
                               szp=ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)))
            del w
        x = torch.randn(M, K, device="cuda").to(dtype)
        out = torch.empty(M, N // 2 if epi else N, device="cuda", dtype=dtype)
        ab = algo_bytes(M, K, N) - (M * N if epi else 0)
        for fmt in ("szh", "szp"):
            def fn(c):
                st = torch.cuda.current_stream().cuda_stream
                if fmt == "szh":
                    _capi.check(L.awq_w4a16_decode_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["szh"].data_ptr(), None, out.data_ptr(), M, N, K, 128, 1, epi, st))
                elif epi == 1:
                    _capi.check(L.awq_w4a16_mlp_gate_up_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["szp"].data_ptr(), out.data_ptr(), M, N, K, 128, 1, st))
                else:
                    _capi.check(L.awq_w4a16_forward_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["szp"].data_ptr(), c["szp"].data_ptr(), c["szp"].data_ptr(),
                                                          None, out.data_ptr(), M, N, K, 128, 1, None, 0, st))
            if fmt == "szp" and epi == 2:
                continue
            for cfg in ((0, 0, 0), (8, 4, 1), (8, 2, 1), (16, 4, 1), (16, 2, 1), (8, 4, 2), (16, 2, 2), (4, 8, 1), (4, 8, 2), (4, 4, 4)):
                if fmt == "szp" and cfg != (0, 0, 0):
                    continue
                row = []
                for probe in (0, 1, 2, 3):
                    _capi.tune(gemvd_waves=cfg[0], gemvd_d=cfg[1], gemvd_bpc=cfg[2], gemvd_probe=probe)
                    try:
                        row.append(time_graph(fn, copies))
                    except Exception as e:  # noqa
                        row.append(float("nan"))
                print(f"K={K:6d} N={N:6d} epi={epi} {fmt} waves={cfg[0]:2d} d={cfg[1]} bpc={cfg[2]}  full {row[0]:6.2f}  stream-only {row[1]:6.2f}  math-only {row[2]:6.2f}  "
                      f"neither {row[3]:6.2f} us   ({ab / row[0] / 1e3:6.0f} GB/s full, {ab / row[1] / 1e3:6.0f} stream-only)", flush=True)
        _capi.tune(gemvd_waves=0, gemvd_d=0, gemvd_probe=0, gemvd_bpc=0)
        del copies
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
