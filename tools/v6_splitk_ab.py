"""GPU experiment: split-K on the v6 loop's 256 x 128 blocks (knob gemm_v6_splitk) against awq_gemm_v4n.hip's split-K, prompts of 256 .. 1024
rows on the Llama-3-8B shapes: outputs compared (same K ranges, same summation order), time per call.  Needs AWQ_TUNING=1."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


def main():
    L = _capi.lib()
    dtype = torch.bfloat16
    print("# shape K N M  ws_MB  v4n_split_us  v6_split_us  TF_v4n TF_v6  equal")
    for (name, K, N) in (("qkv", 4096, 6144), ("o", 4096, 4096), ("gate+up", 4096, 28672), ("down", 14336, 4096)):
        w = synth.random_wq(K, N, dtype=dtype, seed=K + N, keep_q=False)
        c4 = ops.repack_v2_to_cdna4(w["qweight"])
        szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
        bias = (torch.randn(N, device="cuda") * 0.02).to(dtype)
        for M in (256, 300, 512, 777, 1024):
            x = torch.randn(M, K, device="cuda").to(dtype)
            wsb = L.awq_w4a16_forward_cdna4_workspace_bytes(M, N, K)
            res, ys = {}, {}
            for v in (0, 1):
                _capi.tune(gemm_v6_splitk=v)
                ys[v] = ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], bias, szp)
                res[v] = timeit(lambda: ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], None, szp))
            _capi.tune(gemm_v6_splitk=0)
            tf = lambda us: 2.0 * M * N * K / us / 1e6  # noqa: E731
            same = torch.equal(ys[0], ys[1])
            frac = (ys[0] == ys[1]).float().mean().item()
            print(f"{name:8s} {K:6d} {N:6d} {M:5d} {wsb / 2**20:7.1f}  {res[0]:8.1f} {res[1]:8.1f}  {tf(res[0]):7.1f} {tf(res[1]):7.1f}  {same} {frac:.5f}", flush=True)


if __name__ == "__main__":
    main()
