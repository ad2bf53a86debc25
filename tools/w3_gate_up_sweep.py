"""tools/w3_gate_up_sweep.py -- the fused W3 gate/up decode launch (register-ring kernel, EPI 2) and the plain W3 launches of Llama-2-7B per knob set:
graph of 12 rotating weight copies (> 256 MB in total for the big shapes), microseconds per launch.  AWQ_TUNING=1."""
import sys

import torch

import bench_extra
import llm_awq_amd
from llm_awq_amd import _capi, ops

if __name__ == "__main__":
    eng = llm_awq_amd.load_engine()
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    dt = torch.bfloat16
    R = 12
    shapes = {"gate_up": (4096, 22016, 2), "qkv": (4096, 12288, 0), "o": (4096, 4096, 0), "gate": (4096, 11008, 0), "down": (11008, 4096, 0)}
    bufs = {}
    for nm, (K, N, epi) in shapes.items():
        cp = []
        for _ in range(R):
            q = torch.randint(0, 8, (N, K), dtype=torch.uint8, device=dev, generator=gen)
            s, z = bench_extra._rand_sz(K, N, 7, dt, dev, gen)
            cp.append((ops.pack_w3(q), s, z, ops.pack_sz_cdna4(s, z, K)))
        bufs[nm] = cp
    xs = {K: torch.randn(1, K, device=dev, generator=gen).to(dt) for K in (4096, 11008)}
    sets = [[]] + [a.split(",") for a in sys.argv[1:]]
    for kv in sets:
        if kv:
            _capi.tune(**{e.split("=")[0]: int(e.split("=")[1]) for e in kv})
        row = []
        for nm, (K, N, epi) in shapes.items():
            def run():
                return [ops.mlp_gate_up_forward_w3(xs[K], qw, szp) if epi == 2 else ops.forward_w3(xs[K], qw, s, z, szp) for (qw, s, z, szp) in bufs[nm]]
            us = bench_extra._graph_us(run, st, 30, 5) / R
            by = N * K * 3 // 8 + 2 * (K // 128) * N * 2
            row.append(f"{nm} {us:6.2f} us ({by / us / 1e3 / 8000:.3f})")
        print(kv, " | ".join(row), flush=True)
