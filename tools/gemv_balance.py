"""GPU experiment: does the decode GEMV's time per byte depend on how evenly the slabs divide over the 256 CUs?"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402
from tools.gemvc_sweep import time_graph, algo_bytes  # noqa: E402


def main():
    L = _capi.lib()
    dtype = torch.bfloat16
    M = 1
    for (K, N, fused) in [(4096, 4096, 0), (4096, 6144, 0), (4096, 8192, 0), (4096, 12288, 0), (4096, 14336, 0), (4096, 16384, 0),
                          (4096, 28672, 1), (4096, 32768, 1), (14336, 4096, 0), (14336, 8192, 0)]:
        R = max(6, min(24, (700 << 20) // (N * K // 2)))
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
                _capi.check(L.awq_w4a16_mlp_gate_up_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["szp"].data_ptr(), out.data_ptr(), M, N, K, 128, 1, st))
            else:
                _capi.check(L.awq_w4a16_forward_cdna4(x.data_ptr(), c["qw"].data_ptr(), c["s"].data_ptr(), c["z"].data_ptr(),
                                                      c["szp"].data_ptr(), None, out.data_ptr(), M, N, K, 128, 1, None, 0, st))
        us = time_graph(fn, copies)
        ab = algo_bytes(M, K, N)
        blocks = N // 16 // (2 if fused else 1)
        print(f"K={K:6d} N={N:6d} fused={fused} blocks={blocks:5d} ({blocks / 256:5.2f} per CU)  {us:7.2f} us  {ab / us / 1e3:7.1f} GB/s  "
              f"{(us - 1.6) / (ab / 1e6):.4f} us/MB beyond 1.6 us", flush=True)
        del copies
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
