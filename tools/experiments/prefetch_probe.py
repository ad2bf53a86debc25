"""Does a side-stream weight prefetch into the Infinity Cache (awq_prefetch) shorten a decode pass?
32 distinct Llama-3-8B layers (3.7 GB >> 256 MiB, so every byte still comes from HBM once per pass), 128 launches in one
hipGraph; variants: no prefetch / prefetch D launches ahead on P side streams, with or without the join that makes launch i
wait for its own prefetch.   python tools/prefetch_probe.py > gpurun_out/prefetch_probe.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import llm_awq_amd
from llm_awq_amd import _capi, synth

LAYERS = int(os.environ.get("LAYERS", "32"))
SHAPES = [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)]


def main():
    eng = llm_awq_amd.load_engine()
    L = _capi.lib()
    dev = torch.device("cuda:0")
    dtype = torch.bfloat16
    weights = []
    for li in range(LAYERS):
        for si, (name, K, N) in enumerate(SHAPES):
            w = synth.random_wq(K, N, dtype=dtype, device=dev, seed=li * 16 + si, keep_q=False)
            weights.append((name, K, N, eng.repack_v2_to_cdna4(w["qweight"]), w["scales"], w["scaled_zeros"],
                            eng.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)))
            del w
    xs = {K: torch.randn(1, K, device=dev).to(dtype) for K in (4096, 14336)}
    nbytes = sum(N * K // 2 + N * (K // 128) * 4 for (_n, K, N, *_r) in weights)

    def gemv(i):
        name, K, N, qw, s, sz, szp = weights[i]
        if name == "gate_up":
            return eng.mlp_gate_up_cdna4(xs[K], qw, szp)
        return eng.forward_cdna4(xs[K], qw, s, sz, szp, None)

    def prefetch(i, stream, blocks, with_sz):
        qw, szp = weights[i][3], weights[i][6]
        _capi.check(L.awq_prefetch(qw.data_ptr(), qw.numel() * qw.element_size(), blocks, stream.cuda_stream))
        if with_sz:
            _capi.check(L.awq_prefetch(szp.data_ptr(), szp.numel() * 4, max(blocks // 8, 8), stream.cuda_stream))

    main_s = torch.cuda.Stream(device=dev)

    def build(D, P, blocks, join, with_sz=False, hot=False):
        n = len(weights)
        pfs = [torch.cuda.Stream(device=dev) for _ in range(P)]
        g = torch.cuda.CUDAGraph()
        keep = []
        with torch.cuda.graph(g, stream=main_s):
            if D > 0:
                for j in range(min(D, n)):
                    s = pfs[j % P]
                    s.wait_stream(main_s)
                    with torch.cuda.stream(s):
                        prefetch(j, s, blocks, with_sz)
            for i in range(n):
                if D > 0:
                    if join:
                        main_s.wait_stream(pfs[i % P])
                    j = i + D
                    if j < n:
                        s = pfs[j % P]
                        s.wait_stream(main_s)
                        with torch.cuda.stream(s):
                            prefetch(j, s, blocks, with_sz)
                keep.append(gemv(i % 4 if hot else i))
            for s in pfs:
                main_s.wait_stream(s)
        return g, keep

    def time_graph(g, iters=20):
        with torch.cuda.stream(main_s):
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(main_s)
            for _ in range(iters):
                g.replay()
            b.record(main_s)
            torch.cuda.synchronize()
        return a.elapsed_time(b) / iters

    with torch.cuda.stream(main_s):
        for i in range(4):
            gemv(i)
        torch.cuda.synchronize()
    print(f"{LAYERS} layers, {len(weights)} launches, {nbytes / 1e9:.3f} GB of weights + scales per pass")
    print(f"{'variant':<46} {'ms/pass':>8} {'tok/s (32 layers)':>18} {'GB/s':>8} {'% of 8 TB/s':>11}")

    def report(label, g):
        ms = time_graph(g[0])
        print(f"{label:<46} {ms:8.4f} {1e3 / ms * LAYERS / 32:18.1f} {nbytes / ms / 1e6:8.0f} {nbytes / ms / 1e6 / 80:11.1f}", flush=True)

    report("no prefetch", build(0, 1, 0, False))
    report("no prefetch, one layer's weights (cache-hot)", build(0, 1, 0, False, hot=True))
    for (D, P, blocks, join, wsz) in [(1, 1, 256, True, False), (1, 2, 256, True, False), (2, 2, 256, True, False), (2, 3, 256, True, False),
                                      (3, 3, 256, True, False), (2, 2, 64, True, False), (2, 2, 1024, True, False), (2, 2, 256, True, True),
                                      (1, 1, 256, False, False), (2, 2, 256, False, False), (4, 4, 256, True, True)]:
        try:
            report(f"prefetch D={D} streams={P} blocks={blocks} join={int(join)} sz={int(wsz)}", build(D, P, blocks, join, wsz))
        except Exception as e:  # noqa: BLE001
            print(f"D={D} P={P}: {type(e).__name__}: {e}")
    # the prefetch stream alone: what a pure read of the same bytes costs in this harness
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=main_s):
        for i in range(len(weights)):
            prefetch(i, main_s, 256, True)
    report("prefetch kernels only, one stream (read floor)", (g,))


if __name__ == "__main__":
    main()
