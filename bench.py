#!/usr/bin/env python
"""bench.py -- Llama-3-8B W4A16 (g128, bf16 activations) WQLinear hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.

  step      = one decode pass of the hot path: the 160 WQLinear GEMV calls (32 layers x
              {qkv 4096->6144, o 4096->4096, gate 4096->14336, up 4096->14336, down 14336->4096})
              of one Llama-3-8B token, M = 1, every layer with its OWN packed weights (3.7 GB total,
              >> the 256 MB Infinity Cache, so every byte comes from HBM), captured in one hipGraph.
              Attention / norms / lm_head are off-path and excluded.
  value     = decode tokens/s over the whole job (N GPUs = K-sharded tensor parallel, strong scaling).
  roofline  = the dominant kernel (decode GEMV): algorithmic bytes per launch / average launch
              duration (HIP events over the timed region, on the launch stream) vs 8 TB/s.
  prefill   = extra object: the same 160 calls at M = 2048 (GEMM), tok/s and fraction of the 2.5 PFLOP/s
              dense bf16 MFMA peak.
  cpu_baseline = the reference's pure-PyTorch pseudo-quant Linear (awq/quantize/quantizer.py:106-122,
              restated in oracle/awq_oracle.py) timed with F.linear on the host cores, rank 0, N = 1 only,
              on a bounded sample (one decoder block's five linears, M = 1).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA
LAYERS = 32
SHAPES = [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate", 4096, 14336), ("up", 4096, 14336), ("down", 14336, 4096)]


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed PMC pass (profiles/*pmc_traffic.json, produced by
    tools/rocpd_pmc.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this script; FETCH_SIZE
    doubled per MI355X_MICROARCH.md).  Counters cannot be read from inside the timed run, so this is the last
    measured value for the same workload, or None if no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None, None
    try:
        ks = [k for k in json.load(open(files[-1]))["kernels"] if k["kernel"].startswith(kernel_prefix)]
        calls = sum(k["calls"] for k in ks)
        if not calls:
            return None, None
        return int(sum(k["hbm_bytes_per_launch_corrected"] * k["calls"] for k in ks) / calls), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def algo_bytes(M, K, N, esz=2, group=128):
    """packed int4 + scales + scaled_zeros + x + out."""
    return N * K // 2 + 2 * (K // group) * N * esz + M * K * esz + M * N * esz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--prefill-m", type=int, default=4096, help="prefill rows of the headline GEMM figure (the reference quotes TTFT up to 4096 tokens, tinychat/README.md:174-178)")
    ap.add_argument("--prefill-m2", type=int, default=2048, help="second prefill size reported beside it (0 = skip)")
    ap.add_argument("--prefill-m3", type=int, default=512, help="a short prompt (split-K territory) reported beside them (0 = skip)")
    ap.add_argument("--prefill-iters", type=int, default=3)
    ap.add_argument("--no-prefill", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--layers", type=int, default=LAYERS)
    ap.add_argument("--tune", action="append", default=[], help="experiments only: awq_tune_set knobs as key=value (e.g. gemv_probe=2 with --layout v2 turns every decode launch into a linear read of the same weight bytes: the streaming floor of this harness)")
    ap.add_argument("--layout", default="cdna4", choices=["cdna4", "v2"], help="cdna4 = what the rewritten repacker emits (default); v2 = reference checkpoint layout through gemv/gemm_forward_cuda_new")
    ap.add_argument("--unfused-mlp", action="store_true", help="run gate and up as two launches (160 launches per token instead of 128)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"], help="activation / scale dtype: bf16 is BASELINE.json's configuration (default); f16 is the reference's default WQLinear dtype (single-GPU leg only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_tp = os.environ.get("AWQ_BENCH_FORCE_TP") == "1"  # exercise the tensor-parallel leg on one GPU (world size 1)
    if world > 1 or force_tp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import llm_awq_amd
    from llm_awq_amd import synth
    eng = llm_awq_amd.load_engine()
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    if args.tune:
        from llm_awq_amd import _capi
        _capi.tune(**{kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tune})

    if world > 1 or force_tp:
        from llm_awq_amd.parallel import run_tp_bench
        out = run_tp_bench(args, eng, dist, rank, world, dev, SHAPES, algo_bytes)
        if rank == 0:
            print(json.dumps(out))
        dist.destroy_process_group()
        return

    # ---------------- weights: every layer distinct (HBM-resident, 3.7 GB), in the layout the rewritten
    # repacker emits (cdna4 interleave + packed scales); gate and up are stacked along N like tinychat fuses
    # q/k/v (fused_attn.py:566-572) so that decode runs them with the SiLU*mul epilogue in ONE launch ----------------
    L = args.layers
    fused = not args.unfused_mlp
    layer_shapes = ([("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096)] if fused
                    else SHAPES)
    weights = []
    for li in range(L):
        for si, (name, K, N) in enumerate(layer_shapes):
            w = synth.random_wq(K, N, dtype=dtype, device=dev, seed=li * 16 + si, keep_q=False)
            if args.layout == "cdna4":
                qw = eng.repack_v2_to_cdna4(w["qweight"])
                szp = eng.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
            else:
                qw, szp = w["qweight"], None
            weights.append((name, K, N, qw, w["scales"], w["scaled_zeros"], szp))
            del w
    torch.cuda.synchronize()

    def run_pass(xs):
        outs = []
        for (name, K, N, qw, s, sz, szp) in weights:
            x = xs[K]
            m = x.numel() // K
            if szp is not None:
                if name == "gate_up" and m <= 8:
                    outs.append(eng.mlp_gate_up_cdna4(x, qw, szp))      # QuantLlamaMLP: gate, up, silu*mul
                else:
                    outs.append(eng.forward_cdna4(x, qw, s, sz, szp, None))
            elif m < 8:
                outs.append(eng.gemv_forward_cuda_new(x, qw, s, sz, m, N, K, 128))
            else:
                outs.append(eng.gemm_forward_cuda_new(x, qw, s, sz))
        return outs

    def bytes_of(name, M, K, N):
        b = algo_bytes(M, K, N)
        if name == "gate_up" and M <= 8 and args.layout == "cdna4":
            b -= M * (N // 2) * 2  # fused epilogue writes [M, N/2]
        return b

    g = torch.Generator(device=dev).manual_seed(1)

    def make_x(M):
        return {K: torch.randn(M, K, device=dev, generator=g).to(dtype) for K in (4096, 14336)}

    # ---------------- decode leg: K timed steps ----------------
    xs1 = make_x(1)
    side = torch.cuda.Stream(device=dev)
    graph = None
    with torch.cuda.stream(side):
        run_pass(xs1)  # lazy init outside capture
        torch.cuda.synchronize()
        if not args.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                keep = run_pass(xs1)
        step = (lambda: graph.replay()) if graph is not None else (lambda: run_pass(xs1))
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(side)
        for _ in range(args.steps):
            step()
        e1.record(side)
        torch.cuda.synchronize()
        wall_ms = (time.perf_counter() - t0) * 1e3
        ev_ms = e0.elapsed_time(e1)
    ms_per_step = wall_ms / args.steps
    launches = len(weights)
    bytes_step = sum(bytes_of(name, 1, K, N) for (name, K, N, *_r) in weights)
    avg_launch_us = ev_ms * 1e3 / (args.steps * launches)
    gbs = bytes_step * args.steps / (ev_ms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic("awq::gemv_cdna4_kernel" if args.layout == "cdna4" else "awq::gemv_w4a16_kernel")
    roofline = {"bound": "hbm", "kernel": "gemv_cdna4_kernel" if args.layout == "cdna4" else "gemv_w4a16_kernel<BF16>", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_us": round(avg_launch_us, 3), "algorithmic_bytes_per_launch": bytes_step // launches,
                "launches_per_step": launches, "timing": "hip events on the launch stream over the timed region"}
    tok_s = 1e3 / ms_per_step * (L / LAYERS)  # tokens/s of a full 32-layer model

    out = {"metric": "W4A16 decode+prefill tok/s, Llama-3-8B; achieved %HBM (GEMV) / %MFMA (GEMM)",
           "value": round(tok_s, 2),
           "unit": "decode tok/s (the 160 quantised linears of one token: 32 x {qkv, o, gate, up, down}; attention/norm/lm_head off-path)",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "Llama-3-8B W4A16 g128 " + ("bf16" if args.dtype == "bf16" else "fp16") + " activations on 1xMI355X (decode GEMV + prefill GEMM)",
                      "layers": L, "decode_m": 1, "prefill_m": args.prefill_m, "graph": graph is not None,
                      "layout": args.layout, "fused_gate_up_silu_mul": fused and args.layout == "cdna4",
                      "launches_per_token": launches, "parallelism": "tp1", **({"tune": args.tune} if args.tune else {})},
           "roofline": roofline, "device": torch.cuda.get_device_name(dev)}

    # ---------------- prefill leg ----------------
    def prefill(M):
        xsm = make_x(M)
        with torch.cuda.stream(side):
            run_pass(xsm)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            for _ in range(args.prefill_iters):
                run_pass(xsm)
            e1.record(side)
            torch.cuda.synchronize()
            pms = e0.elapsed_time(e1) / args.prefill_iters
        flops = sum(2.0 * M * K * N for (_nm, K, N, *_r) in weights)
        tfl = flops / (pms * 1e-3) / 1e12
        return {"m": M, "ms_per_pass": round(pms, 3), "tok_s": round(M / (pms * 1e-3) * (LAYERS / L), 1),
                "roofline": {"bound": "mfma", "kernel": "gemm_cdna4_v4_kernel (256-wide tiles) + gemm_cdna4_v4n_kernel (128-wide remainder)" if args.layout == "cdna4" else "gemm_w4a16_256x256_kernel",
                             "achieved": round(tfl, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(tfl / MFMA_PEAK_TFLOPS, 4), "traffic": None}}

    if not args.no_prefill:
        out["prefill"] = prefill(args.prefill_m)
        if args.prefill_m2 and args.prefill_m2 != args.prefill_m:
            out["prefill_m%d" % args.prefill_m2] = prefill(args.prefill_m2)
        if args.prefill_m3 and args.prefill_m3 not in (args.prefill_m, args.prefill_m2):
            out["prefill_m%d" % args.prefill_m3] = prefill(args.prefill_m3)

    # ---------------- CPU baseline (reference's pseudo-quant Linear on the host cores) ----------------
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))


def cpu_baseline(budget_s: float = 12.0):
    import torch.nn.functional as F
    from oracle import awq_oracle as O

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    lins = []
    gen = torch.Generator().manual_seed(0)
    for (_n, K, N) in SHAPES:
        w = torch.randn(N, K, generator=gen) * 0.02
        lins.append(O.pseudo_quant_linear(w, 4, 128, torch.bfloat16))
    xs = {4096: torch.randn(1, 4096).bfloat16(), 14336: torch.randn(1, 14336).bfloat16()}
    def one_block():
        for lin in lins:
            F.linear(xs[lin.in_features], lin.weight)
    with torch.no_grad():
        for _ in range(2):
            one_block()
        ts = []
        t_end = time.perf_counter() + budget_s
        while time.perf_counter() < t_end and len(ts) < 200:
            t0 = time.perf_counter()
            one_block()
            ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": round(1.0 / (med * LAYERS), 3), "unit": "decode tok/s (32 x one block's five pseudo-quant Linears, M=1)",
            "cores": cores, "kind": "port", "dtype": "bf16 F.linear",
            "sample": f"one Llama-3-8B decoder block (5 linears, M=1), median of {len(ts)} runs, x32 layers",
            "ms_per_block": round(med * 1e3, 3)}


if __name__ == "__main__":
    main()
