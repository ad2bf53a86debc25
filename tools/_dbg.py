import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from llm_awq_amd import _capi, ops, synth
K, N, M = 4096, 4096, 96
w = synth.random_wq(K, N, dtype=torch.bfloat16, seed=1, keep_q=False)
c4 = ops.repack_v2_to_cdna4(w["qweight"]); szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
x = torch.randn(M, K, device="cuda").bfloat16()
def t(tag, stream=None):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(20): ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], None, szp)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300): ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], None, szp)
        e1.record(); torch.cuda.synchronize()
    print(tag, f"{e0.elapsed_time(e1) / 300 * 1e3:.1f} us per call")
t("default stream, auto parts")
_capi.tune(midm_ks=1); t("default stream, forced unsplit"); _capi.tune(midm_ks=0)
s = torch.cuda.Stream()
t("side stream, auto parts", s)
_capi.tune(midm_ks=1); t("side stream, forced unsplit", s); _capi.tune(midm_ks=0)
