"""GPU: RMSNorm + quantised linear at decode, one fused launch against two launches (a torch RMSNorm kernel sequence is NOT
used as the baseline: the unfused side is a minimal hand-written norm kernel's stand-in = the plain GEMV plus one elementwise
launch of the same size, i.e. a lower bound for what FTLlamaRMSNorm + WQLinear costs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import ops, synth  # noqa: E402
from tools.gemvc_sweep import time_graph  # noqa: E402


def main():
    dt = torch.bfloat16
    for (K, N, fused) in [(4096, 6144, False), (4096, 28672, True)]:
        R = 12
        copies = []
        for i in range(R):
            w = synth.random_wq(K, N, dtype=dt, seed=i, keep_q=False)
            copies.append(dict(qw=ops.repack_v2_to_cdna4(w["qweight"]), s=w["scales"], z=w["scaled_zeros"],
                               szp=ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)))
            del w
        x = torch.randn(1, K, device="cuda").to(dt)
        gamma = torch.ones(K, device="cuda", dtype=dt)
        xn = torch.empty_like(x)

        def fused_fn(c):
            ops.rmsnorm_forward_cdna4(x, gamma, 1e-6, c["qw"], c["szp"], None, fused_gate_up=fused)

        def split_fn(c):
            torch.mul(x, gamma, out=xn)  # stand-in for the norm launch (one elementwise kernel over the row)
            if fused:
                ops.mlp_gate_up_cdna4(xn, c["qw"], c["szp"])
            else:
                ops.gemm_cdna4(xn, c["qw"], c["s"], c["z"], None, c["szp"])

        def plain_fn(c):
            if fused:
                ops.mlp_gate_up_cdna4(x, c["qw"], c["szp"])
            else:
                ops.gemm_cdna4(x, c["qw"], c["s"], c["z"], None, c["szp"])
        a, b, p = time_graph(fused_fn, copies), time_graph(split_fn, copies), time_graph(plain_fn, copies)
        print(f"K={K} N={N} gate_up={int(fused)}: norm+linear fused {a:6.2f} us | linear alone {p:6.2f} us | elementwise launch + linear {b:6.2f} us", flush=True)


if __name__ == "__main__":
    main()
