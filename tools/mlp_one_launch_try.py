"""GPU check + timing of QuantLlamaMLP.forward at decode as ONE launch (awq_w4a16_mlp_decode_cdna4, granule hand-over of h) against the two launches
(fused gate/up + SiLU*mul, then down_proj): correctness against the two-launch path and the oracle, replay from a hipGraph (the epoch lives in the state
buffer), and time per MLP over rotating weight copies (> the 256 MB Infinity Cache).  usage: python tools/mlp_one_launch_try.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import ops, synth  # noqa: E402
from llm_awq_amd.fused_mlp import interleave_gate_up  # noqa: E402
from tools.gemvc_sweep import time_graph  # noqa: E402


def build(hidden, ffn, n_out, seed, dtype=torch.bfloat16):
    g = synth.random_wq(hidden, ffn, dtype=dtype, seed=seed, keep_q=False)
    u = synth.random_wq(hidden, ffn, dtype=dtype, seed=seed + 1, keep_q=False)
    d = synth.random_wq(ffn, n_out, dtype=dtype, seed=seed + 2, keep_q=False)
    qi, si, zi = interleave_gate_up(g["qweight"], u["qweight"], g["scales"], u["scales"], g["scaled_zeros"], u["scaled_zeros"])
    gu_szh, e1 = ops.pack_szh_cdna4(si, zi, hidden)
    d_szh, e2 = ops.pack_szh_cdna4(d["scales"], d["scaled_zeros"], ffn)
    assert e1 and e2
    return dict(gu=ops.repack_v2_to_cdna4(qi), gu_szp=ops.pack_sz_cdna4(si, zi, hidden), gu_szh=gu_szh, d=ops.repack_v2_to_cdna4(d["qweight"]), d_szh=d_szh,
                state=ops.mlp_decode_state(1, ffn, "cuda"))


def two(c, x):
    h = ops.mlp_gate_up_forward_cdna4(x, c["gu"], c["gu_szp"], c["gu_szh"])
    return ops.decode_cdna4(h, c["d"], c["d_szh"], None, 0)


def one(c, x):
    return ops.mlp_decode_cdna4(x, c["gu"], c["gu_szh"], c["d"], c["d_szh"], c["state"])


def main():
    for dtype in (torch.bfloat16, torch.float16):
        for (hidden, ffn, n_out) in ((4096, 4096, 256), (4096, 14336, 4096), (4096, 11008, 4096)):
            c = build(hidden, ffn, n_out, 11, dtype)
            for it in range(5):
                x = torch.randn(1, hidden, device="cuda").to(dtype)
                y2, y1 = two(c, x), one(c, x)
                torch.cuda.synchronize()
                diff = (y1 != y2).float().mean().item()
                rel = ((y1.float() - y2.float()).norm() / y2.float().norm()).item()
                assert int(c["state"][2].item()) == 0, "a consumer gave up waiting"
                assert rel < 2e-3 and diff < 0.3, (dtype, hidden, ffn, n_out, it, diff, rel)  # (8-wave gate/up blocks sum K in another order than the 4-wave launch: 1-ulp flips of h)
            assert int(c["state"][0].item()) == 5 and int(c["state"][1].item()) == 0, c["state"][:3]
            # graph replay: three calls captured, replayed with new inputs
            xs = [torch.zeros(1, hidden, device="cuda", dtype=dtype) for _ in range(3)]
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph, stream=side):
                    ys = [one(c, xx) for xx in xs]
                for rep in range(4):
                    for xx in xs:
                        xx.copy_(torch.randn(1, hidden, device="cuda").to(dtype))
                    gph.replay()
                    torch.cuda.synchronize()
                    for xx, yy in zip(xs, ys):
                        ref = two(c, xx)
                        assert ((yy.float() - ref.float()).norm() / ref.float().norm()).item() < 2e-3, ("replay", rep)
            assert int(c["state"][2].item()) == 0
            print("ok", dtype, (hidden, ffn, n_out), "epoch", int(c["state"][0].item()), flush=True)
    # ---- timing, Llama-3-8B MLP, bf16 ----
    copies = [build(4096, 14336, 4096, 100 + 3 * i) for i in range(10)]
    x = torch.randn(1, 4096, device="cuda").bfloat16()
    for rnd in range(3):
        t2 = time_graph(lambda c: two(c, x), copies)
        t1 = time_graph(lambda c: one(c, x), copies)
        print(f"QuantLlamaMLP decode M=1: two launches {t2:6.2f} us   one launch {t1:6.2f} us   ({100 * (t2 - t1) / t2:+.1f} %)", flush=True)
    assert all(int(c["state"][2].item()) == 0 for c in copies)


if __name__ == "__main__":
    main()
