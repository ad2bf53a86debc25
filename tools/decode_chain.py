"""GPU experiment: decode launches as a CHAIN ordered by in-kernel flags -- consecutive linears on two streams (parallel graph branches),
each launch requests its weight ring at once, waits for its producer's flag, then reads x past the L2 (awq_w4a16_decode_cdna4_chain).
  1. correctness: L layers of o -> gate/up -> down where every launch really consumes its producer's output, chained on two streams (eager and as a replayed
     graph, flags re-zeroed per pass) against the same launches on one stream -- bit for bit;
  2. timing: the bench's decode step (32 x {qkv, o, gate/up, down}, every layer its own weights, static x per width as bench.py) as one graph: plain launches on
     one stream vs the chain on two streams.
    python tools/decode_chain.py [layers]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_awq_amd import _capi, ops, synth  # noqa: E402
from llm_awq_amd.fused_mlp import interleave_gate_up  # noqa: E402

dev = "cuda"
dtype = torch.bfloat16


def native(K, N, w, epi):
    szh, exact = ops.pack_szh_cdna4(w["scales"], w["scaled_zeros"], K)
    assert exact
    return dict(K=K, N=N, qw=ops.repack_v2_to_cdna4(w["qweight"]), szh=szh, epi=epi)


def build_layers(L):
    links = []
    for li in range(L):
        ws = {name: synth.random_wq(K, N, dtype=dtype, device=dev, seed=li * 16 + si, keep_q=False)
              for si, (name, K, N) in enumerate([("qkv", 4096, 6144), ("o", 4096, 4096), ("gate", 4096, 14336), ("up", 4096, 14336), ("down", 14336, 4096)])}
        links.append(native(4096, 6144, ws["qkv"], 0))
        links.append(native(4096, 4096, ws["o"], 0))
        g, u = ws["gate"], ws["up"]
        q, s, z = interleave_gate_up(g["qweight"], u["qweight"], g["scales"], u["scales"], g["scaled_zeros"], u["scaled_zeros"])
        links.append(native(4096, 28672, dict(qweight=q, scales=s, scaled_zeros=z), 2))
        links.append(native(14336, 4096, ws["down"], 0))
    return links


def run_plain(links, xs, outs):
    for i, l in enumerate(links):
        x = xs[i]
        L = _capi.lib()
        _capi.check(L.awq_w4a16_decode_cdna4(x.data_ptr(), l["qw"].data_ptr(), l["szh"].data_ptr(), None, outs[i].data_ptr(), x.numel() // l["K"],
                                             l["N"], l["K"], 128, 1, l["epi"], torch.cuda.current_stream().cuda_stream))


def run_chain(links, xs, outs, states, streams):
    """Three of a layer's four hand-overs are chained (qkv -> o, gate/up -> down, down -> next qkv); o -> gate/up keeps its kernel boundary: a WAITING gate/up
    grid (1792 blocks, four per CU) can fill every CU's LDS before its producer's blocks are dispatched -- nothing orders the dispatch of two graph branches --
    and then nobody makes progress (first version of this script: every replay ran into the bounded wait).  The other consumers leave room by construction
    (one or two blocks per CU).  Streams alternate by layer: A: qkv0 down0 o1 gu1 qkv2 ...; B: o0 gu0 qkv1 down1 o2 ...: at most two launches in flight."""
    cur = torch.cuda.current_stream()
    states.zero_()
    ev = torch.cuda.Event()
    ev.record(cur)
    for st in streams:
        st.wait_event(ev)
    n = len(links)
    A, B = streams[0], streams[1]
    for i, l in enumerate(links):
        layer, kind = divmod(i, 4)
        a, b = (A, B) if layer % 2 == 0 else (B, A)
        st = a if kind in (0, 3) else b
        if kind == 0:
            wait, sig = (states[i - 1] if i > 0 else None), states[i]
        elif kind == 1:
            wait, sig = states[i - 1], None
        elif kind == 2:
            wait, sig = None, states[i]
        else:
            wait, sig = states[i - 1], (states[i] if i + 1 < n else None)
        if os.environ.get("CHAIN_NOWAIT") == "1":
            wait = None  # (timing probe: signals raised, nobody waits -- wrong ordering, the price of the producer side alone)
        if os.environ.get("CHAIN_NOSIG") == "1" and wait is None:
            sig = None
        if os.environ.get("CHAIN_ONE_STREAM") == "1":
            st = A
        with torch.cuda.stream(st):
            if wait is None and sig is None:
                lib = _capi.lib()
                x = xs[i]
                _capi.check(lib.awq_w4a16_decode_cdna4(x.data_ptr(), l["qw"].data_ptr(), l["szh"].data_ptr(), None, outs[i].data_ptr(), x.numel() // l["K"],
                                                       l["N"], l["K"], 128, 1, l["epi"], torch.cuda.current_stream().cuda_stream))
            else:
                cnt = ops.decode_chain_blocks(xs[i - 1].numel() // links[i - 1]["K"], links[i - 1]["N"], links[i - 1]["K"], links[i - 1]["epi"]) if wait is not None else 0
                ops.decode_cdna4_chain(xs[i], l["qw"], l["szh"], None, l["epi"], wait_state=wait, wait_count=cnt, signal_state=sig, out=outs[i])
    for st in streams:
        e2 = torch.cuda.Event()
        e2.record(st)
        cur.wait_event(e2)


def graphed(fn, side):
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            fn()
    return g


def time_graph(g, side, steps=30, warm=60):
    with torch.cuda.stream(side):
        for _ in range(warm):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(steps):
            g.replay()
        e1.record(side)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    M = int(os.environ.get("CHAIN_M", "1"))
    links = build_layers(L)
    n = len(links)
    side = torch.cuda.Stream()
    streams = [torch.cuda.Stream() for _ in range(int(os.environ.get("CHAIN_STREAMS", "2")))]
    states = torch.zeros(n, ops.CHAIN_STATE_WORDS, dtype=torch.int32, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1)

    # ---- 1. correctness on a REAL dependency chain: o -> gate/up -> down per layer (qkv's 6144 outputs feed o's 4096 inputs as their first 4096 columns at M = 1) ----
    if M == 1:
        x0 = (torch.randn(1, 4096, device=dev, generator=gen) * 0.5).to(dtype)
        nchk = min(n, 16)
        sub = links[:nchk]

        def mk():
            outs = [torch.empty(1, l["N"] // 2 if l["epi"] else l["N"], device=dev, dtype=dtype) for l in sub]
            xs = [x0] + [outs[i][:, :sub[i + 1]["K"]] for i in range(nchk - 1)]
            return xs, outs
        xs_a, outs_a = mk()
        xs_b, outs_b = mk()
        with torch.cuda.stream(side):
            run_plain(sub, xs_a, outs_a)
            torch.cuda.synchronize()
            run_chain(sub, xs_b, outs_b, states, streams)
            torch.cuda.synchronize()
        bad = [i for i in range(nchk) if not torch.equal(outs_a[i].view(torch.int16), outs_b[i].view(torch.int16))]
        print(f"real chain of {nchk} links, eager, two streams vs one stream: {'bit-identical' if not bad else 'MISMATCH at links ' + str(bad)};  "
              f"finite {all(torch.isfinite(o.float()).all().item() for o in outs_a)};  timeouts {int(states[:, ops.CHAIN_TIMEOUT_WORD].sum())}")
        gch = graphed(lambda: run_chain(sub, xs_b, outs_b, states, streams), side)
        ok = True
        for rep in range(20):
            for o in outs_b:
                o.fill_(float("nan"))
            with torch.cuda.stream(side):
                gch.replay()
            torch.cuda.synchronize()
            ok = ok and all(torch.equal(outs_a[i].view(torch.int16), outs_b[i].view(torch.int16)) for i in range(nchk))
        print(f"   the same chain as a graph, 20 replays over poisoned outputs: {'bit-identical' if ok else 'MISMATCH'};  timeouts {int(states[:, ops.CHAIN_TIMEOUT_WORD].sum())}")

    # ---- 2. timing, the bench's step ----
    xk = {K: torch.randn(M, K, device=dev, generator=gen).to(dtype) for K in (4096, 14336)}
    xs = [xk[l["K"]] for l in links]
    outs = [torch.empty(M, l["N"] // 2 if l["epi"] else l["N"], device=dev, dtype=dtype) for l in links]
    by = sum(l["N"] * l["K"] // 2 + 4 * (l["K"] // 128) * l["N"] + 2 * M * l["K"] + 2 * M * (l["N"] // 2 if l["epi"] else l["N"]) for l in links)
    g_plain = graphed(lambda: run_plain(links, xs, outs), side)
    ref = [o.clone() for o in outs]
    g_chain = graphed(lambda: run_chain(links, xs, outs, states, streams), side)
    for rnd in range(3):
        tp = time_graph(g_plain, side)
        tc = time_graph(g_chain, side)
        torch.cuda.synchronize()
        same = all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(ref, outs))
        print(f"round {rnd}: M={M} L={L}: plain {tp:.4f} ms/step = {by / tp / 1e6 / 8000:.4f} of 8 TB/s | chain ({len(streams)} streams) {tc:.4f} ms/step = "
              f"{by / tc / 1e6 / 8000:.4f} | chain/plain {tc / tp:.3f} | outputs {'identical' if same else 'DIFFER'} | timeouts {int(states[:, ops.CHAIN_TIMEOUT_WORD].sum())}", flush=True)


if __name__ == "__main__":
    main()
