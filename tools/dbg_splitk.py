import torch
from llm_awq_amd import ops, synth
L = ops._capi.lib()
K, N = 4096, 4096
w = synth.random_wq(K, N, dtype=torch.bfloat16, seed=77, keep_q=False)
c4 = ops.repack_v2_to_cdna4(w["qweight"])
szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
W = ops.dequant_cdna4(c4, w["scales"], w["scaled_zeros"]).float()
def rel(a, b):
    return ((a - b).norm() / b.norm()).item()
for knob in (0, 2):
  ops._capi.tune(skinny_splitk=knob)
  for M in (48, 64):
    for sname in ("null", "torch"):
        x = torch.randn(M, K, device="cuda").bfloat16()
        buf = torch.full((M + 64, N), 7.0, device="cuda", dtype=torch.bfloat16)
        wsb = L.awq_w4a16_forward_cdna4_workspace_bytes(M, N, K)
        ws = torch.full((max(wsb, 16) // 4,), 1000.0, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        st = None if sname == "null" else torch.cuda.current_stream().cuda_stream
        ops._capi.check(L.awq_w4a16_forward_cdna4(x.data_ptr(), c4.data_ptr(), w["scales"].data_ptr(), w["scaled_zeros"].data_ptr(), szp.data_ptr(), None, buf.data_ptr(), M, N, K, 128, 1, ws.data_ptr() if wsb else None, wsb, st))
        torch.cuda.synchronize()
        y = buf[:M].float()
        p0 = x[:, :K // 2].float() @ W[:, :K // 2].t()
        p1 = x[:, K // 2:].float() @ W[:, K // 2:].t()
        print("knob", knob, "M", M, sname, "st", st, "wsb", wsb, "| vs full", round(rel(y, p0 + p1), 4), "vs p0", round(rel(y, p0), 4), "vs p1", round(rel(y, p1), 4), "vs 2p0", round(rel(y, 2 * p0), 4),
              "| ws parts vs p0/p1:", (round(rel(ws[:M * N].view(M, N), p0), 4), round(rel(ws[M * N:2 * M * N].view(M, N), p1), 4)) if wsb else None, flush=True)
