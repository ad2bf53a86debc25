"""EXPERIMENT (apply tools/experiments/gemv_trace.patch first: it adds a trace pointer + s_memrealtime stamps to the decode kernel):
per-wave timeline of one decode GEMV launch (entry, every chunk, loop end, barrier, end) in 10 ns ticks."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402


def main():
    L = _capi.lib()
    L.awq_debug_set_trace.argtypes = [ctypes.c_void_p]
    L.awq_debug_set_trace.restype = None
    dtype = torch.bfloat16
    for (K, N, fused) in [(4096, 4096, 0), (4096, 6144, 0), (14336, 4096, 0), (4096, 28672, 1)]:
        R = 6
        copies = []
        for i in range(R):
            w = synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False)
            copies.append(dict(qw=ops.repack_v2_to_cdna4(w["qweight"]), s=w["scales"], z=w["scaled_zeros"],
                               szp=ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)))
        x = torch.randn(1, K, device="cuda").to(dtype)
        out = torch.empty(1, N, device="cuda", dtype=dtype)
        trace = torch.zeros(4096 * 16 * 16, dtype=torch.int64, device="cuda")

        def fn(c):
            st = torch.cuda.current_stream().cuda_stream
            if fused:
                _capi.check(L.awq_w4a16_mlp_gate_up_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["szp"].data_ptr(), out.data_ptr(), 1, N, K, 128, 1, st))
            else:
                _capi.check(L.awq_w4a16_forward_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["s"].data_ptr(), c["z"].data_ptr(),
                                                      c["szp"].data_ptr(), None, out.data_ptr(), 1, N, K, 128, 1, None, 0, st))
        for c in copies[:-1]:
            fn(c)  # warm code / flush the cache with other copies
        torch.cuda.synchronize()
        L.awq_debug_set_trace(trace.data_ptr())
        fn(copies[-1])
        torch.cuda.synchronize()
        L.awq_debug_set_trace(None)
        t = trace.cpu().numpy().reshape(-1, 16)
        nw = int((t[:, 0] != 0).sum())
        wpb = {(4096, 4096): 16, (4096, 6144): 8, (14336, 4096): 8, (4096, 28672): 4}[(K, N)]
        blk = np.arange(t.shape[0]) // wpb
        t, blk = t[:nw], blk[:nw]
        z = t[:, 0].min()
        ent = t[:, 0] - z
        end = t[:, 14] - z
        life = t[:, 14] - t[:, 0]
        pc = lambda v, qs: [int(np.percentile(v, q)) for q in qs]
        print(f"  (10 ns ticks) entry pct [0,50,90,99,100] = {pc(ent, (0, 50, 90, 99, 100))}; end pct [1,50,90,100] = {pc(end, (1, 50, 90, 100))}; "
              f"life pct [1,50,90,100] = {pc(life, (1, 50, 90, 100))}")
        first = t[:, 1] - t[:, 0]
        late = ent > np.percentile(ent, 90)
        print(f"  first-chunk latency pct [1,50,90,100] = {pc(first, (1, 50, 90, 100))}; for the 10% latest starters: entry med {int(np.median(ent[late]))} life med {int(np.median(life[late]))}")
        slow = life >= np.percentile(life, 95)
        print(f"  slowest 5% waves: entry med {int(np.median(ent[slow]))}, chunk-done times rel. to own entry (med): {[int(np.median(t[slow, j] - t[slow, 0])) for j in range(1, 5)]}; "
              f"fastest half: {[int(np.median(t[~slow, j] - t[~slow, 0])) for j in range(1, 5)]}")
        # per-CU-slot view is not available; blocks of the slow waves:
        print(f"  blocks of the slowest waves (first 24): {sorted(set(blk[slow].tolist()))[:24]}")
        t0 = t[:, 0].min()
        clk = 100e6  # s_memtime on gfx950 ticks at 100 MHz?  print raw numbers and let the reader judge
        ent = t[:, 0] - t0
        nchunk = int(((t[:, 1:12] != 0).sum(1)).max())
        print(f"\nK={K} N={N} fused={fused}: {t.shape[0]} waves, {nchunk} chunks per wave; all times in counter ticks relative to the first wave entry")
        print(f"  entry            : min {ent.min():7d} med {int(np.median(ent)):7d} max {ent.max():7d}")
        for j in range(1, nchunk + 1):
            col = t[:, j]
            m = col != 0
            d = col[m] - t0
            print(f"  chunk {j} done     : min {d.min():7d} med {int(np.median(d)):7d} max {d.max():7d}")
        for (j, name) in ((12, "loop end (red wr)"), (13, "after barrier   "), (14, "end             ")):
            d = t[:, j] - t0
            print(f"  {name}: min {d.min():7d} med {int(np.median(d)):7d} max {d.max():7d}")
        d1 = t[:, 1] - t[:, 0]
        print(f"  per wave: entry -> first chunk done  med {int(np.median(d1))}  (min {d1.min()}, max {d1.max()})")
        dl = t[:, 12] - t[:, 0]
        print(f"  per wave: entry -> loop end          med {int(np.median(dl))}")
        de = t[:, 14] - t[:, 12]
        print(f"  per wave: loop end -> end            med {int(np.median(de))}  max {de.max()}")


if __name__ == "__main__":
    main()
