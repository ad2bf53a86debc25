#!/usr/bin/env python
"""bench.py -- Llama-3-8B W4A16 (g128, bf16 activations) WQLinear hot path on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.  With N > 1 and no launcher around it
(WORLD_SIZE unset) the script re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`.

  step      = one decode pass of the hot path: the 160 WQLinear GEMV calls (32 layers x
              {qkv 4096->6144, o 4096->4096, gate 4096->14336, up 4096->14336, down 14336->4096})
              of one Llama-3-8B token, M = 1, every layer with its OWN packed weights (3.7 GB total,
              >> the 256 MB Infinity Cache, so every byte comes from HBM), captured in one hipGraph.
              gate and up run as ONE launch with the SiLU*mul epilogue, exactly what llm_awq_amd.fused_mlp.QuantLlamaMLP
              (the build's tinychat/modules/fused_mlp.py) does: 128 launches per token.
              Attention / norms / lm_head are off-path and excluded.
  value     = decode tokens/s over the whole job (N GPUs = K-sharded tensor parallel, strong scaling).
  roofline  = the dominant kernel (decode GEMV): algorithmic bytes per launch / average launch
              duration (HIP events over the timed region, on the launch stream) vs 8 TB/s.
  prefill   = the same 160 linears at M = 2048 (the size BASELINE.md / SURVEY.md 8(d) quote), tok/s and fraction of the
              2.5 PFLOP/s dense bf16 MFMA peak; prefill_m4096 / prefill_m512 beside it.
  prefill_m16 / m64 / m128 / m256 / m1024 = shorter prompts on the same weights (fraction of the MFMA roofline; below ~128 rows they are weight-stream bound:
              `hbm_frac` = algorithmic bytes of the pass / time / 8 TB/s beside it).  65 .. 192 rows run the mid-M kernel (csrc/awq_midm_cdna4.hip).
  w3_llama2_7b / tp70b_world1 / moe_mixtral = BASELINE.json configs 3, 4 (world size 1) and 5 as compact legs (bench_extra.py).
  dropin    = the SAME work through the reference's own entry points on RAW reference-layout (v2) buffers:
              awq_inference_engine.gemv_forward_cuda_new / gemm_forward_cuda_new (pybind.cpp:22-23), 160 calls per token.
  cpu_baseline = the reference's pure-PyTorch pseudo-quant Linear (awq/quantize/quantizer.py:106-122,
              restated in oracle/awq_oracle.py) timed with F.linear on the host cores, rank 0, N = 1 only,
              on a bounded sample (one decoder block's five linears; decode M = 1 and prefill M = 2048; bf16 and fp32).
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
CLOCK_RAMP_STEPS = int(os.environ.get("AWQ_BENCH_CLOCK_RAMP", "60"))      # untimed decode steps (~60 ms) before the timed region, the W warmup steps included: the chip leaves its idle clocks
SHAPES = [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate", 4096, 14336), ("up", 4096, 14336), ("down", 14336, 4096)]


def pmc_traffic(kernel_prefixes):
    """HBM bytes per launch of the dominant kernel(s) from the committed PMC pass (profiles/*pmc_traffic.json, produced by
    tools/rocpd_pmc.py from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this script; FETCH_SIZE
    doubled per MI355X_MICROARCH.md).  Counters cannot be read from inside the timed run, so this is the last
    measured value for the same workload, or None if no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
    if not files:
        return None, None
    try:
        ks = [k for k in json.load(open(files[-1]))["kernels"] if any(k["kernel"].startswith(p) for p in kernel_prefixes)]
        calls = sum(k["calls"] for k in ks)
        if not calls:
            return None, None
        return int(sum(k["hbm_bytes_per_launch_corrected"] * k["calls"] for k in ks) / calls), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def profile_consistency(ms_per_step, layers):
    """The committed rocprofv3 record of the decode step AS TIMED HERE -- one hipGraph replay per step under `rocprofv3 --kernel-trace --stats`
    (tools/gpu_decode_graph_profile.sh -> profiles/*decode_graph_consistency.json): the sum of the decode kernels' average durations per step next to the
    ms_per_step the profiled process itself reported (`ratio_in_profile`: must be ~1), and next to this run's ms_per_step (`ratio_vs_this_run`: profiled
    passes run a few per cent lower clocks, MI355X_MICROARCH.md DVFS note)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*decode_graph_consistency.json")))
    if not files:
        return None
    try:
        rec = json.load(open(files[-1]))
        return {"source": os.path.relpath(files[-1], ROOT), "mode": rec.get("mode"), "decode_kernel_rows": len(rec["rows"]),
                "sum_kernel_avg_ms_per_step": rec["sum_kernel_avg_ms_per_step"], "ms_per_step_in_profile": rec["ms_per_step_reported"],
                "ratio_in_profile": rec["ratio"], "ms_per_step": round(ms_per_step, 4),
                "ratio_vs_this_run": round(rec["sum_kernel_avg_ms_per_step"] / ms_per_step, 4)}
    except Exception:
        return None


def launcher_command(n_gpus, argv, port=None):
    """argv of the re-exec: this script under `python -m torch.distributed.run`, one process per GPU of ONE node, rendezvous on 127.0.0.1 (the
    container hostname may not resolve), a free port unless given.  The driver's own form (README) is the same command line."""
    import socket
    if port is None:
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL / peer-mapped buffers across processes need it on this driver
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}", "--master-addr", "127.0.0.1",
            "--master-port", str(int(port)), os.path.abspath(__file__), *argv]


def algo_bytes(M, K, N, esz=2, group=128):
    """packed int4 + scales + scaled_zeros + x + out."""
    return N * K // 2 + 2 * (K // group) * N * esz + M * K * esz + M * N * esz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--prefill-m", type=int, default=2048, help="prefill rows of the headline GEMM figure (SURVEY.md 8(d) / BASELINE.md quote M = 2048)")
    ap.add_argument("--prefill-m2", type=int, default=4096, help="second prefill size reported beside it (0 = skip)")
    ap.add_argument("--prefill-m3", type=int, default=512, help="a short prompt (split-K territory) reported beside them (0 = skip)")
    ap.add_argument("--prefill-small", default="16,64,128,256,1024", help="short prompts (the reference's M <= 192 tile territory, gemm_cuda.cu:1155-1206) reported as prefill_m<M> beside the others ('' = skip)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the compact legs of BASELINE.json configs 3 / 4 (world size 1) / 5 (bench_extra.py: w3_llama2_7b, tp70b_world1, moe_mixtral)")
    ap.add_argument("--prefill-iters", type=int, default=10, help="timed prefill passes per size (after two untimed ones); median and min are reported")
    ap.add_argument("--no-prefill", action="store_true")
    ap.add_argument("--no-dropin", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-batched-decode", action="store_true", help="skip the M = 4 / M = 7 decode legs")
    ap.add_argument("--layers", type=int, default=LAYERS)
    ap.add_argument("--launch-check", action="store_true", help="print what the launcher handed this rank (RANK / WORLD_SIZE / MASTER_*) and exit: no GPU touched")
    ap.add_argument("--tune", action="append", default=[], help="experiments only: awq_tune_set knobs as key=value")
    ap.add_argument("--layout", default="cdna4", choices=["cdna4", "v2"], help="cdna4 = what the rewritten repacker emits (default); v2 = reference checkpoint layout through gemv/gemm_forward_cuda_new only")
    ap.add_argument("--overlap-probe", type=int, default=0, help="experiments (NOT a valid decode figure): issue the decode launches round-robin on this many streams inside the graph, i.e. drop the dependency between consecutive linears -- the upper bound of what cross-launch overlap could give")
    ap.add_argument("--repeat-layers", type=int, default=1, help="experiments: run the --layers layers this many times per step (with --layers 1/2 the weights stay in the 256 MB Infinity Cache)")
    ap.add_argument("--mlp", default="interleaved", choices=["interleaved", "stacked", "unfused"],
                    help="gate/up: one launch on the 8+8 row-interleaved stack QuantLlamaMLP builds (default), one launch on the plain [gate; up] stack, or two launches + F.silu * mul left out (160 launches per token)")
    ap.add_argument("--sz", default="half", choices=["half", "packed"], help="decode side buffer: sz_half (f16-mantissa dequant) or the T-typed sz_packed")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"], help="activation / scale dtype: bf16 is BASELINE.json's configuration (default); f16 is the reference's default WQLinear dtype (single-GPU leg only)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed (no launcher around it): re-exec the same command line under torch.distributed.run, one rank per GPU
        cmd = launcher_command(args.gpus, sys.argv[1:])
        sys.stdout.flush()
        os.execv(cmd[0], cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.launch_check:  # (tests: what the launcher hands every rank, no GPU touched)
        print(json.dumps({"launch_check": {"rank": rank, "local_rank": local_rank, "world": world, "master_addr": os.environ.get("MASTER_ADDR"),
                                           "master_port": os.environ.get("MASTER_PORT"), "ipc_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}}), flush=True)
        return
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_tp = os.environ.get("AWQ_BENCH_FORCE_TP") == "1"  # exercise the tensor-parallel leg on one GPU (world size 1)
    if world > 1 or force_tp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if force_tp and "WORLD_SIZE" not in os.environ:  # (no launcher around a world-1 run: the env:// rendezvous still wants its variables)
            os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
            os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=dev)

    import llm_awq_amd
    from llm_awq_amd import synth
    from llm_awq_amd.fused_mlp import interleave_gate_up
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

    # ---------------- weights: every layer distinct (HBM-resident, 3.7 GB).  `raw` = the reference checkpoint buffers (v2),
    # kept for the drop-in leg; `nat` = what the repacker / QuantLlamaMLP make of them (cdna4 interleave, side buffers) ----------------
    L = args.layers
    raw, nat = [], []
    for li in range(L):
        ws = {name: synth.random_wq(K, N, dtype=dtype, device=dev, seed=li * 16 + si, keep_q=False) for si, (name, K, N) in enumerate(SHAPES)}
        for name, K, N in SHAPES:
            raw.append((name, K, N, ws[name]["qweight"], ws[name]["scales"], ws[name]["scaled_zeros"]))
        if args.layout == "v2":
            continue

        def native(name, K, N, qw, s, z, epi):
            szh, exact = eng.pack_szh_cdna4(s, z, K)
            return (name, K, N, eng.repack_v2_to_cdna4(qw), s, z, eng.pack_sz_cdna4(s, z, K), szh if (exact and args.sz == "half") else None, epi)

        for name in ("qkv", "o"):
            K, N = synth.LLAMA3_8B[name]
            nat.append(native(name, K, N, ws[name]["qweight"], ws[name]["scales"], ws[name]["scaled_zeros"], 0))
        g, u = ws["gate"], ws["up"]
        if args.mlp == "interleaved":
            q, s, z = interleave_gate_up(g["qweight"], u["qweight"], g["scales"], u["scales"], g["scaled_zeros"], u["scaled_zeros"])
            nat.append(native("gate_up", 4096, 28672, q, s, z, 2))
        elif args.mlp == "stacked":
            q = torch.cat([g["qweight"], u["qweight"]], 0)
            s, z = torch.cat([g["scales"], u["scales"]], 1), torch.cat([g["scaled_zeros"], u["scaled_zeros"]], 1)
            nat.append(native("gate_up", 4096, 28672, q, s, z, 1))
        else:
            for name in ("gate", "up"):
                nat.append(native(name, 4096, 14336, ws[name]["qweight"], ws[name]["scales"], ws[name]["scaled_zeros"], 0))
        nat.append(native("down", 14336, 4096, ws["down"]["qweight"], ws["down"]["scales"], ws["down"]["scaled_zeros"], 0))
        del ws, g, u
    if args.no_dropin and args.layout != "v2":
        raw = []
    torch.cuda.synchronize()

    probe_streams = [torch.cuda.Stream(device=dev) for _ in range(args.overlap_probe)] if args.overlap_probe > 1 else []
    def run_native(xs):
        outs = []
        decode = xs[4096].numel() <= 8 * 4096
        if probe_streams and decode:
            cur = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(cur)
            for st in probe_streams:
                st.wait_event(ev)
        seq = nat * (args.repeat_layers if decode else 1)
        for li, (name, K, N, qw, s, z, szp, szh, epi) in enumerate(seq):
            if probe_streams and decode:
                with torch.cuda.stream(probe_streams[li % len(probe_streams)]):
                    outs.append(eng.mlp_gate_up_forward_cdna4(xs[K], qw, szp, szh) if epi == 2 else eng.decode_cdna4(xs[K], qw, szh, None, epi))
                continue
            x = xs[K]
            m = x.numel() // K
            if epi == 2:
                outs.append(eng.mlp_gate_up_forward_cdna4(x, qw, szp, szh))          # QuantLlamaMLP.our_llama_mlp, any row count
            elif m <= 8 and szh is not None:
                outs.append(eng.decode_cdna4(x, qw, szh, None, epi))                 # WQLinear.forward, decode
            elif m <= 8 and epi == 1:
                outs.append(eng.mlp_gate_up_cdna4(x, qw, szp))
            else:
                outs.append(eng.forward_cdna4(x, qw, s, z, szp, None, szh))   # WQLinear.forward, prompts of any length (with the layer's sz_half side buffer, as the module passes it)
        if probe_streams and decode:
            for st in probe_streams:
                e2 = torch.cuda.Event()
                e2.record(st)
                torch.cuda.current_stream().wait_event(e2)
        return outs

    def run_dropin(xs):
        outs = []
        for (name, K, N, qw, s, z) in raw:
            x = xs[K]
            m = x.numel() // K
            if m < 8:
                outs.append(eng.gemv_forward_cuda_new(x, qw, s, z, m, N, K, 128))    # qmodule.py:206-216
            else:
                outs.append(eng.gemm_forward_cuda_new(x, qw, s, z))                  # qmodule.py:217-220
        return outs

    def bytes_native(M):
        tot = 0
        for (name, K, N, *_r, epi) in nat:
            b = algo_bytes(M, K, N)
            if epi:
                b -= M * (N // 2) * 2  # fused epilogue writes [M, N/2]
            tot += b
        return tot

    g = torch.Generator(device=dev).manual_seed(1)

    def make_x(M):
        return {K: torch.randn(M, K, device=dev, generator=g).to(dtype) for K in (4096, 14336)}

    side = torch.cuda.Stream(device=dev)

    def timed_decode(run_pass, steps, warmup, use_graph=True, M=1):
        xs1 = make_x(M)
        graph = None
        with torch.cuda.stream(side):
            run_pass(xs1)  # lazy init / cache builds outside capture
            torch.cuda.synchronize()
            if use_graph:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    keep = run_pass(xs1)  # noqa: F841
            step = (lambda: graph.replay()) if graph is not None else (lambda: run_pass(xs1))
            # clock ramp: a token is ~1 ms of GPU time, and the driver's --warmup 5 ends before the chip has left its idle clocks (--steps 5 --warmup 2 reads
            # 860 tok/s where --steps 50 --warmup 10 reads 1030 on the same box).  Untimed replays up to CLOCK_RAMP_STEPS in total come first; the W warmup
            # steps and the EXACTLY K timed steps follow unchanged (config.clock_ramp_steps in the JSON line)
            for _ in range(max(0, CLOCK_RAMP_STEPS - warmup)):
                step()
            for _ in range(warmup):
                step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(side)
            for _ in range(steps):
                step()
            e1.record(side)
            torch.cuda.synchronize()
            wall_ms = (time.perf_counter() - t0) * 1e3
            ev_ms = e0.elapsed_time(e1)
        return wall_ms, ev_ms, graph is not None

    def timed_prefill(run_pass, M, n_lin):
        """two untimed passes (lazy init, clock ramp: the first timed leg of a run used to read 3 % slow), then --prefill-iters passes each
        bracketed by its own pair of events on the launch stream: median (the reported figure) and min"""
        xsm = make_x(M)
        iters = max(1, args.prefill_iters)
        with torch.cuda.stream(side):
            for _ in range(2):
                run_pass(xsm)
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
            for (e0, e1) in evs:
                e0.record(side)
                run_pass(xsm)
                e1.record(side)
            torch.cuda.synchronize()
            ts = sorted(e0.elapsed_time(e1) for (e0, e1) in evs)
        pms, pmin = ts[len(ts) // 2], ts[0]
        flops = sum(2.0 * M * K * N for (_nm, K, N) in SHAPES) * L
        tfl = flops / (pms * 1e-3) / 1e12
        return pms, tfl, pmin, flops / (pmin * 1e-3) / 1e12

    def per_kernel_decode(steps, warmup):
        """the decode launch kinds one by one (events, graph replay of that kind's L launches, every layer its own weights): where the
        token's time goes.  Each kind alone re-reads L x its bytes per replay (o_proj: 285 MB, just above the 256 MB Infinity Cache)."""
        xs1 = make_x(1)
        kinds = []
        for (name, *_r) in nat:
            if name not in kinds:
                kinds.append(name)
        res = {}
        for kind in kinds:
            sub = [t for t in nat if t[0] == kind]

            def run_kind():
                outs = []
                for (name, K, N, qw, s, z, szp, szh, epi) in sub:
                    if epi == 2:
                        outs.append(eng.mlp_gate_up_forward_cdna4(xs1[K], qw, szp, szh))
                    elif szh is not None:
                        outs.append(eng.decode_cdna4(xs1[K], qw, szh, None, epi))
                    elif epi == 1:
                        outs.append(eng.mlp_gate_up_cdna4(xs1[K], qw, szp))
                    else:
                        outs.append(eng.forward_cdna4(xs1[K], qw, s, z, szp, None))
                return outs

            with torch.cuda.stream(side):
                run_kind()
                torch.cuda.synchronize()
                gk = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gk, stream=side):
                    keep = run_kind()  # noqa: F841
                for _ in range(warmup):
                    gk.replay()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(side)
                for _ in range(steps):
                    gk.replay()
                e1.record(side)
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / (steps * len(sub))
            (_nm, K, N, *_r2, epi) = sub[0]
            b = algo_bytes(1, K, N) - ((N // 2) * 2 if epi else 0)
            res[kind] = {"k": K, "n": N, "algorithmic_bytes": b, "avg_launch_us": round(us, 3), "frac": round(b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
            del gk
        return res

    native_leg = args.layout == "cdna4"
    run_main = run_native if native_leg else run_dropin
    # ---------------- decode leg: K timed steps ----------------
    wall_ms, ev_ms, graphed = timed_decode(run_main, args.steps, args.warmup, not args.no_graph)
    ms_per_step = wall_ms / args.steps
    launches = len(nat) * args.repeat_layers if native_leg else len(raw)
    bytes_step = bytes_native(1) * args.repeat_layers if native_leg else sum(algo_bytes(1, K, N) for (_n, K, N, *_r) in raw)
    avg_launch_us = ev_ms * 1e3 / (args.steps * launches)
    gbs = bytes_step * args.steps / (ev_ms * 1e-3) / 1e9
    kname = ("awq::gemv_dma_kernel (o, gate/up, down) + awq::skinny_cdna4_kernel (qkv)" if native_leg
             else "awq::gemv_dma_kernel / awq::skinny_cdna4_kernel (via gemv_forward_cuda_new + the engine's repack cache)")
    traffic, traffic_src = pmc_traffic(["awq::gemv_dma_kernel", "awq::gemv_cdna4_kernel", "awq::skinny_cdna4_kernel"])
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_us": round(avg_launch_us, 3), "algorithmic_bytes_per_launch": bytes_step // launches,
                "launches_per_step": launches, "timing": "hip events on the launch stream over the timed region"}
    if native_leg and graphed and not probe_streams and args.repeat_layers == 1:
        roofline["per_kernel"] = per_kernel_decode(max(5, args.steps // 2), max(2, args.warmup // 2))
    tok_s = 1e3 / ms_per_step * (L * args.repeat_layers / LAYERS)  # tokens/s of a full 32-layer model

    out = {"metric": "W4A16 decode+prefill tok/s, Llama-3-8B; achieved %HBM (GEMV) / %MFMA (GEMM)",
           "value": round(tok_s, 2),
           "unit": "decode tok/s (the 160 quantised linears of one token: 32 x {qkv, o, gate, up, down}; attention/norm/lm_head off-path)",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": "Llama-3-8B W4A16 g128 " + ("bf16" if args.dtype == "bf16" else "fp16") + " activations on 1xMI355X (decode GEMV + prefill GEMM)",
                      "layers": L, "decode_m": 1, "prefill_m": args.prefill_m, "graph": graphed,
                      "layout": args.layout,
                      "clock_ramp_steps": max(CLOCK_RAMP_STEPS, args.warmup),
                      "decode_mlp": "gate/up + SiLU*mul launch, then down_proj" if native_leg else "separate",
                      "prefill_mlp": "SiLU*mul fused into the gate/up GEMM epilogue" if (native_leg and args.mlp == "interleaved") else "separate",
                      "mlp": args.mlp if native_leg else "unfused (two gemv_forward_cuda_new calls, as tinychat issues them)",
                      "decode_side_buffer": ("sz_half" if args.sz == "half" else "sz_packed") if native_leg else "engine cache",
                      "launches_per_token": launches, "parallelism": "tp1", **({"tune": args.tune} if args.tune else {})},
           "roofline": roofline, "device": torch.cuda.get_device_name(dev)}

    # ---------------- batched decode (the reference GEMV serves 1..7 rows, gemv_cuda.cu:291-329): same launches, M = 4 and 7 ----------------
    if native_leg and not args.no_batched_decode:
        for Mb in (4, 7):
            b_wall, b_ev, _g = timed_decode(run_main, max(5, args.steps // 2), max(2, args.warmup // 2), not args.no_graph, M=Mb)
            b_steps = max(5, args.steps // 2)
            b_bytes = bytes_native(Mb) * args.repeat_layers
            b_gbs = b_bytes * b_steps / (b_ev * 1e-3) / 1e9
            out["decode_m%d" % Mb] = {"m": Mb, "ms_per_step": round(b_wall / b_steps, 4), "tok_s": round(Mb * 1e3 / (b_wall / b_steps) * (L * args.repeat_layers / LAYERS), 1),
                                      "roofline": {"bound": "hbm", "achieved": round(b_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b_gbs / HBM_PEAK_GBS, 4),
                                                   "avg_launch_us": round(b_ev * 1e3 / (b_steps * launches), 3)}}
    # ---------------- self-check: the committed rocprofv3 kernel-trace summary of this command against this run ----------------
    out["profile_consistency"] = profile_consistency(ms_per_step, L)

    # ---------------- prefill leg ----------------
    def prefill(M, run_pass, kernel):
        pms, tfl, pmin, tfl_best = timed_prefill(run_pass, M, launches)
        ptraffic, psrc = pmc_traffic(["awq::gemm_cdna4_v6", "awq::gemm_cdna4_v4"]) if M == 2048 else (None, None)
        return {"m": M, "ms_per_pass": round(pms, 3), "ms_per_pass_min": round(pmin, 3), "passes_timed": max(1, args.prefill_iters), "statistic": "median",
                "tok_s": round(M / (pms * 1e-3) * (LAYERS / L), 1),
                "hbm_frac": round(bytes_native(M) / (pms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),  # (the weight-stream side of the same pass: what bounds it below ~128 rows)
                "roofline": {"bound": "mfma", "kernel": kernel, "achieved": round(tfl, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(tfl / MFMA_PEAK_TFLOPS, 4), "frac_best_pass": round(tfl_best / MFMA_PEAK_TFLOPS, 4), "traffic": ptraffic,
                             **({"traffic_source": psrc, "traffic_note": "HBM bytes per launch, averaged over the tile-kernel launches of the M = 2048 pass (the PMC passes run with --prefill-m2 0 --prefill-m3 0)"} if ptraffic else {})}}

    pk = "gemm_cdna4_v6_kernel (256- / 192- / 128-wide blocks) + gemm_cdna4_v6_pair_kernel (down_proj at <= 2048 rows: pairs of 256-wide blocks, half of K each) + gemm_cdna4_v4n_kernel (split-K launches of short prompts); 9 .. 64 rows skinny_cdna4_kernel, 65 .. 192 rows midm_kernel"
    if not args.no_prefill:
        out["prefill"] = prefill(args.prefill_m, run_main, pk)
        small = [int(v) for v in args.prefill_small.split(",") if v.strip()]
        for extra in (args.prefill_m2, args.prefill_m3, *small):
            if extra and extra != args.prefill_m:
                out["prefill_m%d" % extra] = prefill(extra, run_main, pk)

    # ---------------- the same work through the reference's entry points on raw v2 buffers ----------------
    if native_leg and raw:
        def dropin_leg(with_prefill):
            d_wall, d_ev, _g = timed_decode(run_dropin, args.steps, args.warmup, not args.no_graph)
            d_bytes = sum(algo_bytes(1, K, N) for (_n, K, N, *_r) in raw)
            d_gbs = d_bytes * args.steps / (d_ev * 1e-3) / 1e9
            d = {"launches_per_token": len(raw), "decode_tok_s": round(1e3 / (d_wall / args.steps) * (L / LAYERS), 2),
                 "decode_vs_native": round((1e3 / (d_wall / args.steps)) / (1e3 / ms_per_step), 4),
                 "roofline": {"bound": "hbm", "achieved": round(d_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(d_gbs / HBM_PEAK_GBS, 4),
                              "avg_launch_us": round(d_ev * 1e3 / (args.steps * len(raw)), 3)},
                 "cache": {k: int(v) for k, v in eng.cdna4_cache_info().items()}}
            if with_prefill:
                d["prefill_m%d" % args.prefill_m] = prefill(args.prefill_m, run_dropin, pk + " via gemm_forward_cuda_new")
            return d

        # DEFAULT cache mode (the product default, SURVEY.md 8(b) "no in-place mutation of inputs"): the engine keeps a cdna4 COPY of every
        # qweight it has seen (cache.bytes = that second copy); the caller's v2 buffers are untouched
        eng.cdna4_cache_inplace(False)
        drop = {"entry_points": "awq_inference_engine.gemv_forward_cuda_new / gemm_forward_cuda_new on reference-layout (v2) buffers, engine repack cache on",
                "cache_mode": "default: a cdna4 copy beside the caller's untouched v2 buffers", **dropin_leg(not args.no_prefill)}
        # OPT-IN mode beside it (AWQ_CDNA4_INPLACE=1): the first call converts the caller's qweight where it lies -- no second copy of the
        # weights (cache.bytes 0), but the module's buffer then holds the cdna4 interleave until cdna4_restore(); `raw` is this leg's own copy
        if os.environ.get("AWQ_BENCH_DROPIN_INPLACE", "1") != "0":
            eng.cdna4_cache_inplace(True)
            drop["inplace_opt_in"] = {"cache_mode": "AWQ_CDNA4_INPLACE=1 (opt-in): the caller's qweight is permuted in place, restored by cdna4_restore()",
                                      **dropin_leg(False)}
            eng.cdna4_cache_inplace(False)  # (clears the cache: every qweight gets its v2 interleave back)
        out["dropin"] = drop

    # ---------------- BASELINE.json configs 3, 4 (world size 1) and 5 on the same clock (compact legs: bench_extra.py) ----------------
    if native_leg and not args.no_extra_configs and args.dtype == "bf16" and not args.tune:
        import bench_extra
        del raw[:], nat[:]
        torch.cuda.empty_cache()
        out.update(bench_extra.run_all(eng, dev, side, max(5, args.steps // 2), max(2, args.warmup // 2), max(3, args.prefill_iters // 2)))

    # ---------------- CPU baseline (reference's pseudo-quant Linear on the host cores) ----------------
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(budget_s: float = 24.0):
    """The reference's pseudo-quant Linear (quantizer.py:106-122): dense T weights on the quantisation grid, F.linear on the host.
    Bounded sample: ONE decoder block's five linears (x32 for a token / a prompt), decode M = 1 and prefill M = 2048, bf16 and fp32.
    Thread count: the best of {physical cores, 64, 32} on the decode sample (all logical cores oversubscribe the memory system)."""
    import torch.nn.functional as F
    from oracle import awq_oracle as O

    t_start = time.perf_counter()
    logical = os.cpu_count() or 1
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        physical = logical
    gen = torch.Generator().manual_seed(0)
    lins = {}
    for (_n, K, N) in SHAPES:
        w = torch.randn(N, K, generator=gen) * 0.02
        lin = O.pseudo_quant_linear(w, 4, 128, torch.bfloat16)
        lins.setdefault("bf16", []).append(lin.weight.detach())
        lins.setdefault("fp32", []).append(lin.weight.detach().float())
    xs = {("bf16", M): {K: torch.randn(M, K).bfloat16() for K in (4096, 14336)} for M in (1, 2048)}
    xs.update({("fp32", M): {K: v.float() for K, v in xs[("bf16", M)].items()} for M in (1, 2048)})

    def one_block(dt, M):
        for w in lins[dt]:
            F.linear(xs[(dt, M)][w.shape[1]], w)

    def median_time(dt, M, max_s, max_runs):
        ts = []
        with torch.no_grad():
            one_block(dt, M)
            t_end = time.perf_counter() + max_s
            while (time.perf_counter() < t_end and len(ts) < max_runs) or len(ts) < 2:
                t0 = time.perf_counter()
                one_block(dt, M)
                ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2], len(ts)

    # host streaming bandwidth (a 1 GiB fp32 reduction): the bound the M = 1 figure should sit near
    big = torch.ones(1 << 28)
    cands = sorted({c for c in (physical, 64, 32) if 1 <= c <= logical}, reverse=True)
    sweep = {}
    for c in cands:
        torch.set_num_threads(c)
        sweep[c] = median_time("bf16", 1, 1.5, 20)[0]
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    big.sum()
    t0 = time.perf_counter()
    big.sum()
    stream_gbs = big.numel() * 4 / (time.perf_counter() - t0) / 1e9
    del big
    res = {}
    left = max(4.0, budget_s - (time.perf_counter() - t_start))
    for (dt, M, share) in (("bf16", 1, 0.15), ("fp32", 1, 0.15), ("bf16", 2048, 0.35), ("fp32", 2048, 0.35)):
        med, n = median_time(dt, M, left * share, 30)
        res[f"{'decode' if M == 1 else 'prefill_m2048'}_{dt}"] = {"ms_per_block": round(med * 1e3, 3), "tok_s": round(M / (med * LAYERS), 3), "runs": n}
    wbytes = sum(w.numel() for w in lins["bf16"]) * 2
    dec = res["decode_bf16"]
    return {"value": dec["tok_s"], "unit": "decode tok/s (32 x one block's five pseudo-quant Linears, M=1, bf16 F.linear)",
            "cores": best, "kind": "port", "cpu_model": cpu_model(), "logical_cpus": logical, "physical_cores": physical,
            "threads_sweep_ms_per_block": {str(k): round(v * 1e3, 3) for k, v in sweep.items()},
            "sample": "one Llama-3-8B decoder block (qkv, o, gate, up, down as dense pseudo-quantised weights): median of bounded repeats, x32 layers; "
                      "decode M=1 and prefill M=2048, bf16 and fp32",
            "host_stream_gbs": round(stream_gbs, 1),
            "decode_bf16_effective_gbs": round(wbytes / (dec["ms_per_block"] * 1e-3) / 1e9, 1),
            **res}


if __name__ == "__main__":
    main()
