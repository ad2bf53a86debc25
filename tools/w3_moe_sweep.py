"""GPU experiment: the two parity-only configurations of BASELINE.json timed on their real shapes.
 * W3A16 (Llama-2-7B shapes), decode M = 1 over rotating weight copies + prefill M = 2048 (expand + GEMM v3)
 * Mixtral-8x7B grouped per-expert GEMM (E = 8, top-2), 2048 tokens."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops  # noqa: E402
from llm_awq_amd.moe import sort_by_expert  # noqa: E402


def graph_time(fn, items, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for it in items[:2]:
            fn(it)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for it in items:
                fn(it)
        g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            g.replay()
            e1.record(s)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
    return best * 1e3 / len(items)


def rand_sz(K, N, dev="cuda"):
    G, gpad = K // 128, ((K // 128 + 7) // 8) * 8
    s = torch.zeros(gpad, N, dtype=torch.bfloat16, device=dev)
    z = torch.zeros(gpad, N, dtype=torch.bfloat16, device=dev)
    s[:G] = ((5.2 + 0.8 * torch.rand(G, N, device=dev)) * 0.02 / 7.0).bfloat16()
    z[:G] = -(s[:G].float() * torch.randint(2, 6, (G, N), device=dev).float()).bfloat16()
    return s, z


def main():
    dt = torch.bfloat16
    only_moe = "moe" in sys.argv[1:]
    print("== W3A16, Llama-2-7B shapes (w3c tiles, 0.375 B / weight) ==")
    for (K, N) in ([] if only_moe else [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]):
        R = max(8, min(40, (900 << 20) // (N * K * 3 // 8)))
        items = []
        for i in range(R):
            q = torch.randint(0, 8, (N, K), dtype=torch.uint8, device="cuda")
            s, z = rand_sz(K, N)
            items.append(dict(qw=ops.pack_w3(q), s=s, z=z, szp=ops.pack_sz_cdna4(s, z, K)))
            del q
        x1 = torch.randn(1, K, device="cuda").to(dt)
        us = graph_time(lambda c: ops.forward_w3(x1, c["qw"], c["s"], c["z"], c["szp"]), items)
        by = N * K * 3 // 8 + 4 * (K // 128) * N + 2 * K + 2 * N
        print(f"K={K:6d} N={N:6d} decode M=1   {us:8.2f} us  {by / us / 1e3:8.1f} GB/s  {by / us / 1e3 / 80:5.1f}% of 8 TB/s", flush=True)
        xm = torch.randn(2048, K, device="cuda").to(dt)
        us = graph_time(lambda c: ops.forward_w3(xm, c["qw"], c["s"], c["z"], c["szp"]), items[:4])
        tf = 2.0 * 2048 * N * K / us / 1e6
        print(f"K={K:6d} N={N:6d} prefill M=2048 {us:8.1f} us  {tf:7.1f} TFLOP/s  {tf / 25:5.1f}% of 2.5 PF", flush=True)
        del items
        torch.cuda.empty_cache()
    print("== Mixtral-8x7B grouped GEMM: E = 8, top-2, 2048 tokens (4096 sorted rows) ==")
    E, H, F, T = 8, 4096, 14336, 2048
    for (K, N, name) in [(H, F, "w1/w3"), (F, H, "w2")]:
        qws, ss, zs = [], [], []
        for e in range(E):
            q = torch.randint(0, 16, (N, K), dtype=torch.uint8, device="cuda")
            qws.append(ops.repack_v2_to_cdna4(ops.pack_v2(q)))
            s, z = rand_sz(K, N)
            ss.append(s)
            zs.append(z)
            del q
        qw, s, z = torch.stack(qws), torch.stack(ss), torch.stack(zs)
        gen = torch.Generator(device="cuda").manual_seed(1234)  # (the same routing in every run: rows per expert are printed below)
        ids = torch.stack([torch.randperm(E, device="cuda", generator=gen)[:2] for _ in range(T)])
        order, off = sort_by_expert(ids, E)
        cnts = (off[1:] - off[:-1]).tolist()
        print(f"# rows per expert {cnts}: 256-row tiles {sum((c + 255) // 256 for c in cnts)} (ideal {2 * T / 256:.0f})")
        xs = torch.randn(2 * T, K, device="cuda").to(dt)
        szp = torch.stack([ops.pack_sz_cdna4(ss[e], zs[e], K) for e in range(E)])
        for label, knob, fn in (("128x128 grouped kernel", 0, lambda _c: ops.moe_forward_cdna4(xs, qw, s, z, szp, off)),
                                ("256x256 grouped v6    ", 1, lambda _c: ops.moe_forward_cdna4(xs, qw, s, z, szp, off))):
            _capi.tune(moe_v6=knob)
            us = graph_time(fn, [0, 1, 2, 3])
            _capi.tune(moe_v6=1)
            tf = 2.0 * 2 * T * N * K / us / 1e6
            print(f"{name:6s} K={K:6d} N={N:6d} rows={2 * T} {label} {us:8.1f} us  {tf:7.1f} TFLOP/s  {tf / 25:5.1f}% of 2.5 PF",
                  flush=True)
        if name == "w1/w3":  # the pair as ONE grouped launch (w1 / w3 interleaved 8 + 8 per expert, SiLU * mul in the tile epilogue)
            from llm_awq_amd.fused_mlp import interleave_gate_up
            qi, si, zi = [], [], []
            for e in range(E):
                q2 = torch.randint(0, 16, (N, K), dtype=torch.uint8, device="cuda")
                s2, z2 = rand_sz(K, N)
                qq, sq, zq = interleave_gate_up(ops.repack_cdna4_to_v2(qw[e]), ops.pack_v2(q2), ss[e], s2, zs[e], z2)
                qi.append(ops.repack_v2_to_cdna4(qq))
                si.append(sq)
                zi.append(zq)
                del q2
            qwi, sI, zI = torch.stack(qi), torch.stack(si), torch.stack(zi)
            szpi = torch.stack([ops.pack_sz_cdna4(si[e], zi[e], K) for e in range(E)])
            del qi
            us_f = graph_time(lambda _c: ops.moe_mlp_gate_up_cdna4(xs, qwi, sI, zI, szpi, off), [0, 1, 2, 3])
            tf = 2.0 * 2 * T * (2 * N) * K / us_f / 1e6
            us_2 = graph_time(lambda _c: ops.silu_mul(ops.moe_forward_cdna4(xs, qw, s, z, szp, off), ops.moe_forward_cdna4(xs, qw, s, z, szp, off)), [0, 1, 2, 3])
            print(f"w1+w3  K={K:6d} N=2x{N:5d} rows={2 * T} FUSED grouped v6 (one launch, SiLU*mul epilogue) {us_f:8.1f} us  {tf:7.1f} TFLOP/s  {tf / 25:5.1f}% of 2.5 PF"
                  f"   | two grouped launches + tail kernel {us_2:8.1f} us ({2.0 * 2 * T * (2 * N) * K / us_2 / 1e6 / 25:5.1f}%)", flush=True)
            del qwi, sI, zI, szpi
        for tokens in (8, 32, 96):  # batched decode: top-2 -> 2 x tokens sorted rows
            idd = torch.stack([torch.randperm(E, device="cuda")[:2] for _ in range(tokens)])
            _o, offd = sort_by_expert(idd, E)
            xd = torch.randn(2 * tokens, K, device="cuda").to(dt)
            res = []
            for knob in (1, 0):
                _capi.tune(moe_v4=knob)
                res.append(graph_time(lambda _c: ops.moe_forward_cdna4(xd, qw, s, z, szp, offd), [0, 1, 2, 3]))
            _capi.tune(moe_v4=1)
            by = E * N * K // 2
            print(f"{name:6s} K={K:6d} N={N:6d} rows={2 * tokens:4d} grouped skinny {res[0]:8.1f} us ({by / res[0] / 1e3:6.0f} GB/s if every expert is hit)"
                  f"   128x128 grouped {res[1]:8.1f} us", flush=True)
        del qw, s, z, qws, ss, zs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
