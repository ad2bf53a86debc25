"""tools/w3_short_prompts.py -- 3-bit layers at 9..64 rows: the skinny kernel on w3c tiles against the masked 256-row tile of the prefill GEMM
(knob w3_skinny_max 64 / 8, AWQ_TUNING=1): microseconds per launch over rotating weight copies in one graph, Llama-2-7B shapes, bf16."""
import torch

import bench_extra
from llm_awq_amd import _capi, ops

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    dt = torch.bfloat16
    R = 8
    shapes = {"qkv": (4096, 12288, 0), "o": (4096, 4096, 0), "gate_up": (4096, 22016, 2), "down": (11008, 4096, 0)}
    bufs = {}
    for nm, (K, N, epi) in shapes.items():
        cp = []
        for _ in range(R):
            q = torch.randint(0, 8, (N, K), dtype=torch.uint8, device=dev, generator=gen)
            s, z = bench_extra._rand_sz(K, N, 7, dt, dev, gen)
            cp.append((ops.pack_w3(q), s, z, ops.pack_sz_cdna4(s, z, K)))
        bufs[nm] = cp
    for M in (9, 16, 32, 48, 64):
        xs = {K: torch.randn(M, K, device=dev, generator=gen).to(dt) for K in (4096, 11008)}
        for knob in (8, 64, 8, 64):
            _capi.tune(w3_skinny_max=knob)
            row = []
            for nm, (K, N, epi) in shapes.items():
                def run():
                    return [ops.mlp_gate_up_forward_w3(xs[K], qw, szp) if epi == 2 else ops.forward_w3(xs[K], qw, s, z, szp) for (qw, s, z, szp) in bufs[nm]]
                row.append(f"{nm} {bench_extra._graph_us(run, st, 20, 3) / R:6.1f}")
            print(f"M={M:3d} w3_skinny_max={knob:2d} us per launch: " + " | ".join(row), flush=True)
        _capi.tune(w3_skinny_max=64)
