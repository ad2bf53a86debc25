"""GPU experiment: LDS-DMA streaming decode kernel (awq_gemv_dma.hip) vs the register-ring kernel (awq_gemv_cdna4.hip) on the
Llama-3-8B decode shapes, M = 1, a graph of launches over rotating weight copies (> the 256 MB Infinity Cache), with a
correctness check of every configuration against the ring kernel's output.  usage: python tools/gemvd_sweep.py [M] [quick]"""
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
    quick = len(sys.argv) > 2
    shapes = [(4096, 4096, 0), (4096, 6144, 0), (14336, 4096, 0), (4096, 28672, 1)]
    for (K, N, fused) in shapes:
        R = max(10, min(48, (900 << 20) // (N * K // 2)))
        copies = []
        for i in range(R):
            w = synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False)
            szh, exact = ops.pack_szh_cdna4(w["scales"], w["scaled_zeros"], K)
            assert exact
            c = dict(qw=ops.repack_v2_to_cdna4(w["qweight"]), s=w["scales"], z=w["scaled_zeros"],
                     szp=ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K), szh=szh)
            if fused:
                from llm_awq_amd.fused_mlp import interleave_gate_up
                h = N // 2
                qi, si, zi = interleave_gate_up(w["qweight"][: h // 4], w["qweight"][h // 4:], w["scales"][:, :h].contiguous(),
                                                w["scales"][:, h:].contiguous(), w["scaled_zeros"][:, :h].contiguous(),
                                                w["scaled_zeros"][:, h:].contiguous())
                c["qwi"] = ops.repack_v2_to_cdna4(qi)
                c["szhi"] = ops.pack_szh_cdna4(si, zi, K)[0]
            copies.append(c)
            del w
        x = torch.randn(M, K, device="cuda").to(dtype)
        out = torch.empty(M, N // 2 if fused else N, device="cuda", dtype=dtype)

        def fn(c):
            st = torch.cuda.current_stream().cuda_stream
            if fused:
                _capi.check(L.awq_w4a16_mlp_gate_up_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["szp"].data_ptr(), out.data_ptr(),
                                                          M, N, K, 128, 1, st))
            else:
                _capi.check(L.awq_w4a16_forward_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["s"].data_ptr(), c["z"].data_ptr(),
                                                      c["szp"].data_ptr(), None, out.data_ptr(), M, N, K, 128, 1, None, 0, st))

        ab = algo_bytes(M, K, N) - (M * N if fused else 0)
        _capi.tune(gemv_dma=0)
        fn(copies[0])
        torch.cuda.synchronize()
        ref = out.clone()
        us = time_graph(fn, copies)
        print(f"K={K:6d} N={N:6d} M={M} fused={fused} ring kernel                {us:8.2f} us  {ab / us / 1e3:8.1f} GB/s  {ab / us / 1e3 / 80:5.1f}%", flush=True)
        _capi.tune(gemv_dma=1)
        cfgs = [(0, 0)] if quick else [(0, 0), (4, 0), (8, 0), (16, 0), (8, 1), (8, 2), (16, 1), (16, 2), (16, 4), (4, 2), (4, 4)]
        for (waves, d) in cfgs:
            _capi.tune(gemvd_waves=waves, gemvd_d=d)
            out.zero_()
            try:
                fn(copies[0])
                torch.cuda.synchronize()
            except Exception as e:  # noqa
                print("cfg failed", waves, d, e)
                continue
            same = torch.equal(out, ref)
            maxd = (out.float() - ref.float()).abs().max().item()
            us = time_graph(fn, copies)
            print(f"K={K:6d} N={N:6d} M={M} fused={fused} dma waves={waves:2d} d={d}  {us:8.2f} us  {ab / us / 1e3:8.1f} GB/s  "
                  f"{ab / us / 1e3 / 80:5.1f}%  identical_to_ring={same} maxdiff={maxd:.3g}", flush=True)
        _capi.tune(gemvd_waves=0, gemvd_d=0)

        # ---- f16-mantissa dequant (sz_half) through awq_w4a16_decode_cdna4 ----
        def fn_h(c, epi=(1 if fused else 0)):
            st = torch.cuda.current_stream().cuda_stream
            qw, szh = (c["qwi"], c["szhi"]) if epi == 2 else (c["qw"], c["szh"])
            _capi.check(L.awq_w4a16_decode_cdna4(x.data_ptr(), qw.data_ptr(), szh.data_ptr(), None, out.data_ptr(), M, N, K, 128, 1, epi, st))

        hcfgs = [(0, 0)] if quick else [(0, 0), (8, 2), (8, 4), (16, 1), (16, 2), (16, 4), (4, 4), (4, 8)]
        for epi in ((1, 2) if fused else (0,)):
            for (waves, d) in hcfgs:
                _capi.tune(gemvd_waves=waves, gemvd_d=d)
                out.zero_()
                try:
                    fn_h(copies[0], epi)
                    torch.cuda.synchronize()
                except Exception as e:  # noqa
                    print("cfg failed", waves, d, e)
                    continue
                same = (out == ref).float().mean().item()
                us = time_graph(lambda c: fn_h(c, epi), copies)
                print(f"K={K:6d} N={N:6d} M={M} fused={fused} szh epi={epi} waves={waves:2d} d={d}  {us:8.2f} us  {ab / us / 1e3:8.1f} GB/s  "
                      f"{ab / us / 1e3 / 80:5.1f}%  same_as_ring={same:.4f}", flush=True)
        _capi.tune(gemvd_waves=0, gemvd_d=0)
        del copies
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
