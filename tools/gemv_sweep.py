"""GPU experiment: time the decode GEMV (through the C ABI) for Llama-3-8B layer shapes across tuning
knobs, rotating over enough distinct weight copies to defeat the 256 MB Infinity Cache.  Each
measurement = one hipGraph of R back-to-back launches, replayed; HIP events on the capture stream.
usage: python tools/gemv_sweep.py [--m 1] [--quick] [--dtype bf16|f16]"""
import argparse
import itertools
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402

PROBES = os.environ.get("AWQ_PROBES") == "1"  # the library must have been built with AWQ_PROBES=1 too
DEFAULT = dict(gemv_waves=0, gemv_pf=0, gemv_v2fast=1, **(dict(gemv_probe=0, gemv_probe_blocks=2048) if PROBES else {}))


def algo_bytes(M, K, N):
    return N * K // 2 + 2 * (K // 128) * N * 2 + M * K * 2 + M * N * 2


def time_cfg(fn, copies, reps=5):
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
    return best * 1e3 / len(copies)  # us per launch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, nargs="+", default=[1])
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--defaults-only", action="store_true")
    args = ap.parse_args()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    shapes = [(4096, 4096), (4096, 6144), (4096, 14336), (14336, 4096), (4096, 28672)]
    if args.quick:
        shapes = [(4096, 4096), (4096, 14336), (14336, 4096)]
    print(f"device {torch.cuda.get_device_name(0)}  dtype {args.dtype}")
    L = _capi.lib()
    for (K, N) in shapes:
        nbytes = N * K // 2
        R = max(8, min(32, (450 << 20) // nbytes))
        copies = [synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False) for i in range(R)]
        if dtype == torch.bfloat16:
            for c in copies:
                c["qweight_cdna4"] = ops.repack_v2_to_cdna4(c["qweight"])
                c["sz_packed"] = ops.pack_sz_cdna4(c["scales"], c["scaled_zeros"], K)
        for M in args.m:
            x = torch.randn(M, K, device="cuda").to(dtype)
            out = torch.empty(M, N, device="cuda", dtype=dtype)

            dt = 1 if dtype == torch.bfloat16 else 0
            cur = {"layout": 0}

            def fn(c):
                st = torch.cuda.current_stream().cuda_stream
                if cur["layout"] == 2:
                    _capi.check(L.awq_w4a16_forward_cdna4(x.data_ptr(), c["qweight_cdna4"].data_ptr(), c["scales"].data_ptr(),
                                                          c["scaled_zeros"].data_ptr(), c["sz_packed"].data_ptr(), None,
                                                          out.data_ptr(), M, N, K, 128, dt, None, 0, st))
                elif cur["layout"]:
                    _capi.check(L.awq_w4a16_gemv_cdna4(x.data_ptr(), c["qweight_cdna4"].data_ptr(), c["scales"].data_ptr(),
                                                       c["scaled_zeros"].data_ptr(), c["sz_packed"].data_ptr(),
                                                       out.data_ptr(), M, N, K, 128, dt, st))
                else:
                    _capi.check(L.awq_w4a16_gemv(x.data_ptr(), c["qweight"].data_ptr(), c["scales"].data_ptr(),
                                                 c["scaled_zeros"].data_ptr(), out.data_ptr(), M, N, K, 128, dt, st))
            ab = algo_bytes(M, K, N)
            layouts = (0, 1) if dtype == torch.bfloat16 else (0,)
            cfgs = [dict(DEFAULT, layout=l) for l in (layouts + ((2,) if dtype == torch.bfloat16 else ()))]
            cfgs.insert(1, dict(DEFAULT, gemv_v2fast=0, layout=0))  # the older v2-layout kernel
            if not args.defaults_only and PROBES:
                cfgs.append(dict(DEFAULT, gemv_probe=3, layout=0))
                cfgs.append(dict(DEFAULT, gemv_probe=2, gemv_probe_blocks=4096, layout=0))
                for layout in layouts:
                    for probe in (1, 0):
                        for w, pf in itertools.product((4, 8, 16), (2, 4, 8)):
                            if (K // 128) // w < pf and pf > 2:
                                continue
                            cfgs.append(dict(DEFAULT, gemv_waves=w, gemv_pf=pf, gemv_probe=probe, layout=layout))
            for cfg in cfgs:
                cfg = dict(cfg)
                cur["layout"] = cfg.pop("layout")
                _capi.tune(**cfg)
                try:
                    us = time_cfg(fn, copies)
                except Exception as e:  # noqa
                    print("   cfg failed", cfg, e)
                    continue
                print(f"K={K:6d} N={N:6d} M={M:2d} layout={cur['layout']} waves={cfg['gemv_waves']:2d} pf={cfg['gemv_pf']} "
                      f"probe={cfg.get('gemv_probe', 0)} v2fast={cfg['gemv_v2fast']}  {us:8.2f} us  {ab / us / 1e3:8.1f} GB/s  {ab / us / 1e3 / 80:5.1f}%", flush=True)
            _capi.tune(**DEFAULT)
        del copies
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
