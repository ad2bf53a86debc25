"""GPU experiment (AWQ_PROBES build of csrc/awq_midm_cdna4.hip): per-block time stamps of one mid-M launch -- where a launch's microseconds go.
    python tools/midm_stamps.py [M] [shape]      shape: qkv | o | gate+up | down"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402

SHAPES = {"qkv": (4096, 6144), "o": (4096, 4096), "gate+up": (4096, 28672), "down": (14336, 4096)}


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    name = sys.argv[2] if len(sys.argv) > 2 else "gate+up"
    K, N = SHAPES[name]
    L = _capi.lib()
    dtype = torch.bfloat16
    copies = []
    for i in range(6):
        w = synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False)
        szh, exact = ops.pack_szh_cdna4(w["scales"], w["scaled_zeros"], K)
        copies.append(dict(qw=ops.repack_v2_to_cdna4(w["qweight"]), s=w["scales"], z=w["scaled_zeros"], szp=ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K), szh=szh))
    x = torch.randn(M, K, device="cuda").to(dtype)
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    _capi.tune(midm=1, midm_min=9, midm_max=255, midm_waves=int(os.environ.get("WAVES", "8")), midm_ns=1, midm_ks=1, midm_probe=int(os.environ.get("PROBE", "0")))
    ws = torch.empty(1 << 20, dtype=torch.float32, device="cuda")

    def fn(c):
        _capi.check(L.awq_w4a16_forward_cdna4_szh(x.data_ptr(), c["qw"].data_ptr(), c["s"].data_ptr(), c["z"].data_ptr(), c["szp"].data_ptr(), c["szh"].data_ptr(),
                                                  None, out.data_ptr(), M, N, K, 128, 1, ws.data_ptr(), ws.numel() * 4, torch.cuda.current_stream().cuda_stream))

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for c in copies[:2]:
            fn(c)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for c in copies:
                fn(c)
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    print(f"{name} M={M}: {e0.elapsed_time(e1) * 1e3 / len(copies):.1f} us per launch (graph of {len(copies)})")
    blocks = (N // 16 + 7) // 8
    buf = np.zeros(blocks * 8, dtype=np.uint64)
    lib = ctypes.CDLL(_capi.lib()._name)
    rc = lib.awq_dev_midm_stamps(buf.ctypes.data_as(ctypes.c_void_p), blocks)
    assert rc == 0, rc
    st = buf.reshape(blocks, 4, 2).astype(np.int64)
    rt, cy = st[:, :, 0], st[:, :, 1]
    t0 = rt[:, 0].min()
    us = (rt - t0) / 100.0  # 100 MHz
    print("per block, us from the first block's start (min / median / max):")
    for j, nm in enumerate(["start", "loop start", "loop end", "end"]):
        print(f"  {nm:>10}: {us[:, j].min():7.2f} {np.median(us[:, j]):7.2f} {us[:, j].max():7.2f}")
    d = np.diff(cy, axis=1)
    dr = np.diff(rt, axis=1) / 100.0
    for j, nm in enumerate(["prologue", "loop", "epilogue"]):
        print(f"  {nm:>10}: cycles min / median / max {d[:, j].min():8d} {int(np.median(d[:, j])):8d} {d[:, j].max():8d}   us median {np.median(dr[:, j]):6.2f}   clock {np.median(d[:, j]) / max(np.median(dr[:, j]), 1e-9) / 1e3:5.2f} GHz")
    print(f"  launch span (first start -> last end): {us[:, 3].max():.2f} us; steps {K // 128}: loop cycles per step median {np.median(d[:, 1]) / (K // 128):.0f}")


if __name__ == "__main__":
    main()
