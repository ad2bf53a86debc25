"""GPU experiment: the vendor's dense bf16 GEMV (torch.matmul at M = 1 -> hipBLASLt / rocBLAS) on matrices with the SAME BYTE COUNT as the
Llama-3-8B W4A16 decode launches (a bf16 matrix of N x K/4 has the bytes of the int4 N x K one), graph-captured over rotating copies
(> 256 MB).  It streams the same bytes with no dequantisation at all: its GB/s is the practical streaming rate of one launch of that
size on this stack, next to this repository's decode kernel.  usage: python tools/dense_gemv_ceiling.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402
from tools.gemvc_sweep import time_graph  # noqa: E402


def main():
    L = _capi.lib()
    dtype = torch.bfloat16
    for (K, N, epi) in [(4096, 4096, 0), (4096, 6144, 0), (14336, 4096, 0), (4096, 28672, 2)]:
        R = max(10, min(48, (900 << 20) // (N * K // 2)))
        Kd = K // 4  # same bytes as the packed int4 matrix
        dense = [torch.randn(N, Kd, device="cuda").to(dtype) for _ in range(R)]
        xd = torch.randn(Kd, 1, device="cuda").to(dtype)
        us_d = time_graph(lambda w: torch.matmul(w, xd), dense)
        del dense
        copies = []
        for i in range(R):
            w = synth.random_wq(K, N, dtype=dtype, seed=i, keep_q=False)
            s, z, q = w["scales"], w["scaled_zeros"], w["qweight"]
            if epi == 2:
                from llm_awq_amd.fused_mlp import interleave_gate_up
                h = N // 2
                q, s, z = interleave_gate_up(q[: h // 4], q[h // 4:], s[:, :h].contiguous(), s[:, h:].contiguous(), z[:, :h].contiguous(), z[:, h:].contiguous())
            copies.append((ops.repack_v2_to_cdna4(q), ops.pack_szh_cdna4(s, z, K)[0]))
            del w
        x = torch.randn(1, K, device="cuda").to(dtype)
        out = torch.empty(1, N // 2 if epi else N, device="cuda", dtype=dtype)

        def fn(c):
            _capi.check(L.awq_w4a16_decode_cdna4(x.data_ptr(), c[0].data_ptr(), c[1].data_ptr(), None, out.data_ptr(), 1, N, K, 128, 1, epi,
                                                 torch.cuda.current_stream().cuda_stream))
        us_q = time_graph(fn, copies)
        wb = N * K // 2
        print(f"{wb / 1e6:6.1f} MB of weights ({N} x {K} int4 | {N} x {Kd} bf16): dense bf16 GEMV {us_d:6.2f} us = {wb / us_d / 1e3:6.0f} GB/s ({wb / us_d / 1e3 / 80:4.1f}% of 8 TB/s)   "
              f"W4A16 decode kernel {us_q:6.2f} us = {wb / us_q / 1e3:6.0f} GB/s ({wb / us_q / 1e3 / 80:4.1f}%)", flush=True)
        del copies
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
