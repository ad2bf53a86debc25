"""GPU experiment: the mid-M kernel (csrc/awq_midm_cdna4.hip) per Llama-3-8B layer shape and row count, every block shape (waves x slabs per wave) and
K part count the library compiles, against the round-5 path (knob midm = 0: skinny / masked-tile kernels) -- us per launch over rotating weight copies
(> the 256 MB Infinity Cache) in one graph, and the norm-wise distance of every configuration's result from the round-5 path's.
    python tools/midm_sweep.py [M ...]         MIDM_SHAPES=qkv,o,gate+up,down   MIDM_CFGS=8x1,4x1,4x2   MIDM_KS=0,1,2,4,8,16   MIDM_SZH=1   MIDM_PROBE=0,3,4 (AWQ_PROBES builds)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402

SHAPES = [(4096, 6144, "qkv"), (4096, 4096, "o"), (4096, 28672, "gate+up"), (14336, 4096, "down")]
OTHER = [(8192, 10240, "qkv70b"), (8192, 8192, "o70b"), (8192, 57344, "gate+up70b"), (28672, 8192, "down70b"), (4096, 12288, "qkv7b"), (4096, 22016, "gate+up7b"), (11008, 4096, "down7b")]  # by name only
if os.environ.get("MIDM_SHAPES"):
    SHAPES = [s for s in SHAPES + OTHER if s[2] in os.environ["MIDM_SHAPES"].split(",")]
CFGS = [tuple(int(v) for v in c.split("x")) for c in os.environ.get("MIDM_CFGS", "8x1,4x1,4x2").split(",")]
KSS = [int(v) for v in os.environ.get("MIDM_KS", "0").split(",")]
SZH = os.environ.get("MIDM_SZH", "0") == "1"
PROBES = [int(v) for v in os.environ.get("MIDM_PROBE", "0").split(",")]  # timing probes (wrong results): bit 0 no x traffic, 1 no weight traffic, 2 no LDS reads / math



def graph_time(fn, items, reps=3):
    if os.environ.get("MIDM_EAGER") == "1":  # (under rocprofv3 --pmc: plain launches, one pass)
        for it in items:
            fn(it)
        torch.cuda.synchronize()
        return float("nan")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for it in items[:2]:
            fn(it)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for it in items:
                fn(it)
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
    return best * 1e3 / len(items)


def main():
    L = _capi.lib()
    Ms = [int(a) for a in sys.argv[1:]] or [16, 32, 64, 128]
    dtype = torch.bfloat16
    print(f"{'shape':>8} {'K':>6} {'N':>6} {'M':>5} {'cfg':>14} {'us':>8} {'vs r5':>7} {'GB/s':>8} {'TFLOP/s':>8} {'rel err':>9}")
    for (K, N, name) in SHAPES:
        R = max(4, min(12, (320 << 20) // (N * K // 2) + 1))
        copies = []
        for i in range(R):
            w = synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False)
            szh, exact = ops.pack_szh_cdna4(w["scales"], w["scaled_zeros"], K)
            copies.append(dict(qw=ops.repack_v2_to_cdna4(w["qweight"]), s=w["scales"], z=w["scaled_zeros"],
                               szp=ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K), szh=szh if (exact and SZH) else None))
            del w
        for M in Ms:
            x = torch.randn(M, K, device="cuda").to(dtype)
            out = torch.empty(M, N, device="cuda", dtype=dtype)
            _capi.tune(midm=0)
            wsb = max(L.awq_w4a16_forward_cdna4_workspace_bytes(M, N, K), 32 * 128 * N * 4, 16)  # (the round-5 path's own scratch; every split of this kernel fits 32 parts)
            ws = torch.empty(wsb // 4, dtype=torch.float32, device="cuda")

            def fn(c):
                if c["szh"] is not None:
                    _capi.check(L.awq_w4a16_forward_cdna4_szh(x.data_ptr(), c["qw"].data_ptr(), c["s"].data_ptr(), c["z"].data_ptr(), c["szp"].data_ptr(),
                                                              c["szh"].data_ptr(), None, out.data_ptr(), M, N, K, 128, 1, ws.data_ptr(), wsb,
                                                              torch.cuda.current_stream().cuda_stream))
                else:
                    _capi.check(L.awq_w4a16_forward_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["s"].data_ptr(), c["z"].data_ptr(), c["szp"].data_ptr(), None,
                                                          out.data_ptr(), M, N, K, 128, 1, ws.data_ptr(), wsb, torch.cuda.current_stream().cuda_stream))

            by = N * K // 2 + 4 * (K // 128) * N + 2 * M * K + 2 * M * N
            fl = 2.0 * M * N * K
            _capi.tune(midm=0)
            base = graph_time(fn, copies)
            fn(copies[0])
            ref = out.float().clone()
            print(f"{name:>8} {K:6d} {N:6d} {M:5d} {'round 5':>14} {base:8.1f} {1.0:7.2f} {by / base / 1e3:8.1f} {fl / base / 1e6:8.1f} {0.0:9.1e}", flush=True)
            if os.environ.get("MIDM_PRODUCT") == "1":  # the library's own routing (midm_takes) instead of every row count on this kernel
                _capi.tune(midm=1, midm_min=65, midm_max=128)
            else:
                _capi.tune(midm=1, midm_min=9, midm_max=255)
            for (wv, ns) in CFGS:
                for ks in KSS:
                    for pr in PROBES:
                        _capi.tune(midm_waves=wv, midm_ns=ns, midm_ks=ks, midm_probe=pr)
                        out.zero_()
                        tag = f"{wv}x{ns} ks{ks}" + (f" p{pr}" if pr else "")
                        try:
                            us = graph_time(fn, copies)
                        except Exception as e:  # noqa: BLE001
                            print(f"{name:>8} {K:6d} {N:6d} {M:5d} {tag:>14}  failed: {e}")
                            continue
                        fn(copies[0])
                        err = ((out.float() - ref).norm() / ref.norm()).item()
                        print(f"{name:>8} {K:6d} {N:6d} {M:5d} {tag:>14} {us:8.1f} {us / base:7.2f} {by / us / 1e3:8.1f} {fl / us / 1e6:8.1f} {err:9.1e}", flush=True)
            _capi.tune(midm_waves=0, midm_ns=0, midm_ks=0, midm_probe=0)
        del copies
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
