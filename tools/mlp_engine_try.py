"""GPU check + timing of QuantLlamaMLP.forward at decode as ONE persistent launch (awq_w4a16_mlp_decode_cdna4, csrc/awq_mlp_engine.hip) against the two
launches (fused gate/up + SiLU*mul, then down_proj): agreement with the two-launch path, h read back from the granules, graph replay (the epoch lives in the
state buffer), time per MLP over rotating weight copies (> the 256 MB Infinity Cache), and the kernel's own per-phase s_memtime stamps.
usage: python tools/mlp_engine_try.py [quick]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402
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
                state=ops.mlp_decode_state(1, ffn, "cuda"), ffn=ffn)


def two_h(c, x):
    return ops.mlp_gate_up_forward_cdna4(x, c["gu"], c["gu_szp"], c["gu_szh"])


def two(c, x):
    return ops.decode_cdna4(two_h(c, x), c["d"], c["d_szh"], None, 0)


def one(c, x):
    return ops.mlp_decode_cdna4(x, c["gu"], c["gu_szh"], c["d"], c["d_szh"], c["state"])


def granule_h(c, dtype):
    F = c["ffn"]
    g = c["state"][_capi.AWQ_MLP_DECODE_COUNTER_BYTES // 4:][:F].view(F // 2, 2)
    return g[:, 0].contiguous().view(torch.int16).view(dtype).reshape(1, F), g[:, 1]


def check(quick):
    shapes = ((4096, 14336, 4096), (4096, 11008, 4096), (4096, 4096, 4096), (4096, 2048, 4096), (4096, 128, 4096), (4096, 1152, 4096))
    for dtype in (torch.bfloat16,) if quick else (torch.bfloat16, torch.float16):
        for (hidden, ffn, n_out) in shapes:
            c = build(hidden, ffn, n_out, 11, dtype)
            worst = (0.0, 0.0, 0.0)
            for it in range(4):
                x = torch.randn(1, hidden, device="cuda").to(dtype)
                h2 = two_h(c, x)
                y2, y1 = two(c, x), one(c, x)
                torch.cuda.synchronize()
                st = c["state"][:3].tolist()
                assert st[2] == 0, ("a wave gave up waiting", st)
                assert st[0] == it + 1 and st[1] == 0, st
                h1, tags = granule_h(c, dtype)
                assert bool((tags == it + 1).all()), "granule tags"
                dh = (h1 != h2).float().mean().item()
                diff = (y1 != y2).float().mean().item()
                rel = ((y1.float() - y2.float()).norm() / y2.float().norm()).item()
                worst = (max(worst[0], dh), max(worst[1], diff), max(worst[2], rel))
                assert rel < 2e-3 and diff < 0.3 and dh < 0.05, (dtype, hidden, ffn, n_out, it, dh, diff, rel)
            # graph replay: three calls captured, replayed with new inputs
            xs = [torch.zeros(1, hidden, device="cuda", dtype=dtype) for _ in range(3)]
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                gph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gph, stream=side):
                    ys = [one(c, xx) for xx in xs]
                for rep in range(3):
                    for xx in xs:
                        xx.copy_(torch.randn(1, hidden, device="cuda").to(dtype))
                    gph.replay()
                    torch.cuda.synchronize()
                    for xx, yy in zip(xs, ys):
                        ref = two(c, xx)
                        assert ((yy.float() - ref.float()).norm() / ref.float().norm()).item() < 2e-3, ("replay", rep)
            assert int(c["state"][2].item()) == 0
            print(f"ok {dtype} {(hidden, ffn, n_out)} epoch {int(c['state'][0].item())}  worst: h flips {worst[0]:.4f}  y flips {worst[1]:.4f}  rel {worst[2]:.2e}", flush=True)


def stamps(copies, x):
    L = _capi.lib()
    buf = torch.zeros(256 * 16 * 8, dtype=torch.int64, device="cuda")
    for c in copies[:3]:
        one(c, x)
    torch.cuda.synchronize()
    _capi.check(L.awq_w4a16_mlp_decode_cdna4_set_stamps(ctypes.c_void_p(buf.data_ptr())))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    one(copies[3], x)
    e1.record()
    torch.cuda.synchronize()
    _capi.check(L.awq_w4a16_mlp_decode_cdna4_set_stamps(None))
    t = buf.view(256, 16, 8).cpu().double()
    # s_memtime is per XCD (the eight counters are not synchronised): every stamp is taken relative to ITS OWN wave's start; waves of one block share a CU
    d = t - t[:, :, :1]
    blk = t - t[:, :1, :1]          # relative to wave 0 of the same block (same CU, same counter)
    names = ["start", "first tile landed", "gate/up phase done", "h published (wave 0) / past the barrier", "h gathered", "down_proj phase done", "end"]
    print(f"stamps of one launch (s_memtime ticks = shader cycles, relative to the wave's own start; eager launch incl. events {e0.elapsed_time(e1) * 1e3:.1f} us)")
    q = torch.tensor([0.0, 0.1, 0.5, 0.9, 1.0], dtype=torch.float64)
    for p, nm in enumerate(names):
        v = d[:, :, p].flatten()
        qs = torch.quantile(v, q).tolist()
        print(f"  {nm:42s} min {qs[0]:7.0f}  p10 {qs[1]:7.0f}  median {qs[2]:7.0f}  p90 {qs[3]:7.0f}  max {qs[4]:7.0f}")
    print("  by wave index (median over the 256 blocks): start after the block's wave 0 | first tile | gate/up done | past barrier | gathered | down done | end")
    for w in range(16):
        row = [blk[:, w, 0].median().item()] + [d[:, w, p].median().item() for p in range(1, 7)]
        print(f"    wave {w:2d}: " + " ".join(f"{x:8.0f}" for x in row))
    span = d[:, :, 6].max().item()
    return span


def main():
    quick = "quick" in sys.argv[1:]
    check(quick)
    copies = [build(4096, 14336, 4096, 100 + 3 * i) for i in range(10)]
    x = torch.randn(1, 4096, device="cuda").bfloat16()
    ts = []
    for rnd in range(3):
        t2 = time_graph(lambda c: two(c, x), copies)
        t1 = time_graph(lambda c: one(c, x), copies)
        tg = time_graph(lambda c: two_h(c, x), copies)
        ts.append((t2, t1))
        print(f"QuantLlamaMLP decode M=1: two launches {t2:6.2f} us (gate/up launch alone {tg:6.2f})   one persistent launch {t1:6.2f} us   ({100 * (t2 - t1) / t2:+.1f} %)", flush=True)
    assert all(int(c["state"][2].item()) == 0 for c in copies)
    span = stamps(copies, x)
    print(f"(ticks per us, if the span is the graph-replay time of the launch minus ~1.5 us of boundary: {span / max(ts[-1][1] - 1.5, 1):.0f})")
    if os.environ.get("AWQ_PROBES") == "1":  # (an AWQ_PROBES=1 build of the library: AWQ_CDNA4_LIB=... beside the product one)
        for pv, nm in ((1, "no math (stream only)"), (2, "no weight DMA (math + hand-over only)"), (3, "neither")):
            _capi.tune(mlp_engine_probe=pv)
            print(f"probe {nm}: {time_graph(lambda c: one(c, x), copies):6.2f} us", flush=True)
        _capi.tune(mlp_engine_probe=0)


if __name__ == "__main__":
    main()
