"""GPU experiment: block shape of the LDS-DMA streaming decode kernel (awq_gemv_dma.hip) for BATCHED decode, M = 2 .. 4 -- the x staging is
m x K x 2 bytes per block whatever its wave count, so at m > 1 fewer, larger blocks per CU (knobs gemvd_waves / gemvd_want / gemvd_d) trade
co-resident blocks for ring depth.  A graph of launches over rotating weight copies (> the 256 MB Infinity Cache).  Needs AWQ_TUNING=1.
usage: python tools/decode_cfg_sweep.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402
from tools.gemvc_sweep import time_graph  # noqa: E402


def main():
    L = _capi.lib()
    if os.environ.get("EXTRA_TUNE"):  # further knobs for the whole run: "key=value,..."
        _capi.tune(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ["EXTRA_TUNE"].split(",")})
    dtype = torch.bfloat16
    ms = [int(a) for a in sys.argv[1:]] or [1, 2, 4]
    shapes = [("qkv", 4096, 6144, 0), ("o", 4096, 4096, 0), ("gate/up", 4096, 28672, 2), ("down", 14336, 4096, 0),
              ("gu70b", 8192, 57344, 2), ("gu7b", 4096, 22016, 2), ("qkv70b", 8192, 10240, 0), ("down70b", 28672, 8192, 0), ("o70b", 8192, 8192, 0), ("qkv7b", 4096, 12288, 0), ("down7b", 11008, 4096, 0), ("gate", 4096, 14336, 0)]
    if os.environ.get("SHAPES"):
        shapes = [s for s in shapes if s[0] in os.environ["SHAPES"].split(",")]
    cfgs = [(0, 0, 0)] + [(w, 1, d) for w in (4, 8, 16) for d in (1, 2, 4, 8)] + [(0, 0, 0)]  # (want = 1: the ring depth is the forced one unless ONE block exceeds the LDS)
    if os.environ.get("CFGS"):  # "waves:want:d[:ns2],..." ; waves = -1: the skinny kernel; ns2 = 1: two slabs per block on the wide launches (knob gemvd_ns2)
        cfgs = [tuple(int(v) for v in c.split(":")) for c in os.environ["CFGS"].split(",")]
    for (name, K, N, epi) in shapes:
        R = max(10, min(40, (700 << 20) // (N * K // 2)))
        copies = []
        for i in range(R):
            w = synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False)
            if epi == 2:
                from llm_awq_amd.fused_mlp import interleave_gate_up
                h = N // 2
                qi, si, zi = interleave_gate_up(w["qweight"][: h // 4], w["qweight"][h // 4:], w["scales"][:, :h].contiguous(),
                                                w["scales"][:, h:].contiguous(), w["scaled_zeros"][:, :h].contiguous(),
                                                w["scaled_zeros"][:, h:].contiguous())
                szh, exact = ops.pack_szh_cdna4(si, zi, K)
                copies.append(dict(qw=ops.repack_v2_to_cdna4(qi), szh=szh))
            else:
                szh, exact = ops.pack_szh_cdna4(w["scales"], w["scaled_zeros"], K)
                copies.append(dict(qw=ops.repack_v2_to_cdna4(w["qweight"]), szh=szh))
            assert exact
            del w
        for M in ms:
            x = torch.randn(M, K, device="cuda").to(dtype)
            out = torch.empty(M, N // 2 if epi else N, device="cuda", dtype=dtype)

            def fn(c):
                st = torch.cuda.current_stream().cuda_stream
                _capi.check(L.awq_w4a16_decode_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["szh"].data_ptr(), None, out.data_ptr(), M, N, K,
                                                     128, 1, epi, st))

            _capi.tune(decode_skinny_from=9)
            ref = None
            line = []
            for cfg in cfgs:
                (w, want, d), ns2, il = cfg[:3], (cfg[3] if len(cfg) > 3 else 0), (cfg[4] if len(cfg) > 4 else 0)  # il: knob gemvd_il (interleaved K split)
                _capi.tune(decode_skinny_from=1 if w < 0 else 9)
                if ns2:  # (the knob exists in builds of the two-slab experiment only: tools/EXPERIMENTS.md)
                    _capi.tune(gemvd_ns2=ns2)
                _capi.tune(gemvd_waves=max(w, 0), gemvd_want=want, gemvd_d=d, gemvd_il=il)
                out.zero_()
                try:
                    fn(copies[0])
                    torch.cuda.synchronize()
                except Exception as e:  # the shape is not served in this configuration
                    line.append(f"{w}w/d{d}:n/a")
                    continue
                if ref is None:
                    ref = out.clone()
                ok = bool((out == ref).all())
                if (out.float() - ref.float()).abs().max().item() > 0.05 * ref.float().abs().max().item():
                    line.append(f"{w}w/d{d}:WRONG")
                    continue
                us = time_graph(fn, copies)
                line.append(f"{w}w/d{d}{'/ns2' if ns2 else ''}{'/il%d' % il if il else ''}:{us:.2f}{'' if ok else '!'}")
            print(f"{name:8s} M={M}  " + "  ".join(line), flush=True)
        del copies
        torch.cuda.empty_cache()
    _capi.tune(gemvd_waves=0, gemvd_want=0, gemvd_d=0, decode_skinny_from=0, gemvd_il=0)


if __name__ == "__main__":
    main()
