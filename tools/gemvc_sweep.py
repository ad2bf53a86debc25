"""GPU experiment: decode fast path (gemv_cdna4_kernel) across (waves, chunk) for the Llama-3-8B decode shapes, M = 1,
graph of launches over rotating weight copies (> the 256 MB Infinity Cache).  usage: python tools/gemvc_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402


def algo_bytes(M, K, N):
    return N * K // 2 + 2 * (K // 128) * N * 2 + M * K * 2 + M * N * 2


def time_graph(fn, copies, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for c in copies[:2]:
            fn(c)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for c in copies:
                fn(c)
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            g.replay()
            e1.record(s)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
    return best * 1e3 / len(copies)


def main():
    L = _capi.lib()
    dtype = torch.bfloat16
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    for (K, N, fused) in [(4096, 4096, 0), (4096, 6144, 0), (14336, 4096, 0), (4096, 28672, 1), (4096, 14336, 0)]:
        R = max(10, min(48, (900 << 20) // (N * K // 2)))
        copies = []
        for i in range(R):
            w = synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False)
            copies.append(dict(qw=ops.repack_v2_to_cdna4(w["qweight"]), s=w["scales"], z=w["scaled_zeros"],
                               szp=ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)))
            del w
        x = torch.randn(M, K, device="cuda").to(dtype)
        out = torch.empty(M, N, device="cuda", dtype=dtype)

        def fn(c):
            st = torch.cuda.current_stream().cuda_stream
            if fused:
                _capi.check(L.awq_w4a16_mlp_gate_up_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["szp"].data_ptr(), out.data_ptr(),
                                                          M, N, K, 128, 1, st))
            else:
                _capi.check(L.awq_w4a16_forward_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["s"].data_ptr(), c["z"].data_ptr(),
                                                      c["szp"].data_ptr(), None, out.data_ptr(), M, N, K, 128, 1, None, 0, st))
        ab = algo_bytes(M, K, N)
        pipe_mode = os.environ.get("GEMVC_PIPE")
        if pipe_mode:
            for (pipe, ps) in ((0, 1), (2, 1), (2, 2), (3, 1)):
                for waves in (4, 8, 16):
                    _capi.tune(gemvc_waves=waves, gemvc_s=0, gemvc_pipe=pipe, gemvc_pipe_s=ps)
                    us = time_graph(fn, copies)
                    print(f"K={K:6d} N={N:6d} M={M} fused={fused} ring={pipe} step={ps} waves={waves:2d}  {us:8.2f} us  {ab / us / 1e3:8.1f} GB/s  "
                          f"{ab / us / 1e3 / 80:5.1f}%", flush=True)
            _capi.tune(gemvc_waves=0, gemvc_s=0, gemvc_pipe=-1, gemvc_pipe_s=0)
            del copies
            torch.cuda.empty_cache()
            continue
        for waves in (0, 4, 8, 16):
            for s_ in ((0,) if waves == 0 else (2, 4, 7, 8)):
                nit = K // 128
                if waves and (waves * s_ < nit // 2 and False):
                    continue
                _capi.tune(gemvc_waves=waves, gemvc_s=s_)
                try:
                    us = time_graph(fn, copies)
                except Exception as e:  # noqa
                    print("cfg failed", waves, s_, e)
                    continue
                print(f"K={K:6d} N={N:6d} M={M} fused={fused} waves={waves:2d} S={s_}  {us:8.2f} us  {ab / us / 1e3:8.1f} GB/s  "
                      f"{ab / us / 1e3 / 80:5.1f}%", flush=True)
        _capi.tune(gemvc_waves=0, gemvc_s=0)
        del copies
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
