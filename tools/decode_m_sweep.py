"""GPU experiment: batched decode (M = 1..8) through awq_w4a16_decode_cdna4 on the Llama-3-8B shapes -- the LDS-DMA streaming
kernel (awq_gemv_dma.hip: x staged per slab, m x K x 2 bytes by LDS-DMA) against the skinny kernel (awq_skinny_cdna4.hip: x through
registers, shared by the slabs of a block) behind the same entry (knob decode_skinny_from), a graph of launches over rotating weight
copies (> the 256 MB Infinity Cache).  Needs AWQ_TUNING=1.  usage: python tools/decode_m_sweep.py [quick]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402
from tools.gemvc_sweep import algo_bytes, time_graph  # noqa: E402


def main():
    L = _capi.lib()
    dtype = torch.bfloat16
    quick = len(sys.argv) > 1
    shapes = [("qkv", 4096, 6144, 0), ("o", 4096, 4096, 0), ("gate/up", 4096, 28672, 2), ("down", 14336, 4096, 0)]
    ms = (1, 4, 5, 6, 7, 8) if quick else (1, 2, 3, 4, 5, 6, 7, 8)
    total = {(m, v): 0.0 for m in ms for v in ("dma", "skinny", "auto")}
    print("# shape  K  N  M  dma_us  skinny_us  product_us  dma_GB/s  skinny_GB/s  same")
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

            res, outs = {}, {}
            for v, frm in (("dma", 9), ("skinny", 1), ("auto", 0)):  # auto = the product routing (skinny_takes, awq_gemv_dma.hip)
                _capi.tune(decode_skinny_from=frm)
                out.zero_()
                fn(copies[0])
                torch.cuda.synchronize()
                outs[v] = out.clone()
                res[v] = time_graph(fn, copies)
                total[(M, v)] += res[v]
            same = (outs["dma"] == outs["skinny"]).float().mean().item()
            ab = algo_bytes(M, K, N) - (M * N if epi else 0)
            print(f"{name:8s} {K:6d} {N:6d} {M}  {res['dma']:8.2f} {res['skinny']:8.2f} {res['auto']:8.2f}  {ab / res['dma'] / 1e3:8.1f} {ab / res['skinny'] / 1e3:8.1f}  "
                  f"{same:.4f}", flush=True)
        del copies
        torch.cuda.empty_cache()
    _capi.tune(decode_skinny_from=0)
    print("# per layer (qkv + o + gate/up + down), us")
    for M in ms:
        print(f"# M={M}  dma {total[(M, 'dma')]:7.2f}  skinny {total[(M, 'skinny')]:7.2f}  product {total[(M, 'auto')]:7.2f}")


if __name__ == "__main__":
    main()
