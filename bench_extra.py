"""bench_extra.py -- the other BASELINE.json configurations on the driver's clock, as compact legs of bench.py's JSON line (N = 1 only).

  w3_llama2_7b   cfg3: Llama-2-7B W3A16 g128 (w3c tiles, 0.375 B / weight): decode M = 1 and batched decode of 16 rows as a fraction of the HBM roofline, prefill M = 2048 as a
                 fraction of the bf16 MFMA roofline; the five WQLinear(w_bit=3) calls of a block (qkv, o, gate, up, down), 8 distinct layers (607 MB,
                 beyond the 256 MB Infinity Cache) for decode, 4 for prefill.
  tp70b_world1   cfg4 at world size 1: the UNSHARDED Llama-3-70B block shapes (8192 -> 10240, 8192 -> 8192, gate/up 8192 -> 2 x 28672 fused,
                 28672 -> 8192), 4 distinct layers (1.7 GB): decode M = 1 and prefill M = 2048.  (The sharded legs need the 8-GPU node: bench.py --gpus N.)
  moe_mixtral    cfg5: one Mixtral-8x7B expert block, E = 8, top-2, 2048 tokens = 4096 sorted rows, seeded routing: the fused w1 / w3 grouped launch
                 (SiLU * mul in the tile epilogue) and the w2 grouped launch, each as a fraction of the MFMA roofline.

Every fraction is SURVEY.md 8(d)'s: algorithmic bytes (packed weights + scales + zeros + x + out) / time / 8 TB/s, or 2 M N K / time / 2.5 PFLOP/s, timed
with HIP events on the launch stream; decode legs replay a hipGraph of the layers' launches.  Synthetic weights made on the GPU (llm_awq_amd.synth)."""
import torch

HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TFLOPS = 2500.0


def _graph_us(run, stream, steps, warmup):
    """average microseconds of one replay of the graph of run() (events on `stream`)"""
    with torch.cuda.stream(stream):
        run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            keep = run()  # noqa: F841
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # these graphs are 0.1 - 0.5 ms long: a handful of replays ends before the clocks have ramped (a leg timed right after an idle gap read 5 - 15 % slow).
        # Keep the GPU busy for ~40 ms of measured replay time before the timed region, as bench.py's own warm-up steps do for the headline legs
        busy_ms, rounds = 0.0, 0
        while busy_ms < 40.0 and rounds < 300:
            e0.record(stream)
            for _ in range(max(10, warmup)):
                g.replay()
            e1.record(stream)
            torch.cuda.synchronize()
            busy_ms += e0.elapsed_time(e1)
            rounds += 1
        steps = max(steps, 30)  # (sub-millisecond graphs: a timed region of at least ~8 ms)
        e0.record(stream)
        for _ in range(steps):
            g.replay()
        e1.record(stream)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / steps
    del g
    return us


def _median_us(run, stream, iters):
    with torch.cuda.stream(stream):
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for (e0, e1) in evs:
            e0.record(stream)
            run()
            e1.record(stream)
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for (e0, e1) in evs)
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3


def _hbm(bytes_, us):
    gbs = bytes_ / (us * 1e-6) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}


def _mfma(flops, us, us_min=None):
    tf = flops / (us * 1e-6) / 1e12
    d = {"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4)}
    if us_min:
        d["frac_best_pass"] = round(flops / (us_min * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4)
    return d


def _rand_sz(K, N, levels, dtype, dev, gen):
    """per-group (scale, scaled_zero) with the statistics of a N(0, 0.02^2) weight on the reference's grid (quantizer.py:61-103)"""
    G, gpad = K // 128, ((K // 128 + 7) // 8) * 8
    s = torch.zeros(gpad, N, dtype=dtype, device=dev)
    z = torch.zeros(gpad, N, dtype=dtype, device=dev)
    s[:G] = ((5.2 + 0.8 * torch.rand(G, N, device=dev, generator=gen)) * 0.02 / levels).to(dtype)
    lo, hi = (2, 6) if levels == 7 else (5, 11)
    z[:G] = -(s[:G].float() * torch.randint(lo, hi, (G, N), device=dev, generator=gen).float()).to(dtype)
    return s, z


def w3_llama2_7b(eng, dev, stream, steps, warmup, iters, dtype=torch.bfloat16):
    from llm_awq_amd import ops, synth
    gen = torch.Generator(device=dev).manual_seed(303)
    L_dec, L_pre = 8, 4
    layers, fused = [], []
    for _li in range(L_dec):
        lin, ints = {}, {}
        for nm in ("qkv", "o", "gate", "up", "down"):
            K, N = synth.LLAMA2_7B[nm]
            q = torch.randint(0, 8, (N, K), dtype=torch.uint8, device=dev, generator=gen)
            s, z = _rand_sz(K, N, 7, dtype, dev, gen)
            lin[nm] = (K, N, eng.pack_w3(q), s, z, eng.pack_sz_cdna4(s, z, K))
            if nm in ("gate", "up"):
                ints[nm] = q
        # QuantLlamaMLP(w_bit = 3)'s stream: integer rows interleaved 8 + 8 per slab, packed into w3c tiles (fused_mlp.interleave_gate_up_w3)
        K, F = synth.LLAMA2_7B["gate"]
        qi = torch.stack([ints["gate"].view(F // 8, 8, K), ints["up"].view(F // 8, 8, K)], 1).reshape(2 * F, K).contiguous()

        def cols(a, b):
            return torch.stack([a.view(-1, F // 8, 8), b.view(-1, F // 8, 8)], 2).reshape(a.shape[0], 2 * F).contiguous()

        si, zi = cols(lin["gate"][3], lin["up"][3]), cols(lin["gate"][4], lin["up"][4])
        fused.append((K, 2 * F, eng.pack_w3(qi), eng.pack_sz_cdna4(si, zi, K)))
        layers.append(lin)
        del ints, qi, si, zi
    xs = {M: {K: torch.randn(M, K, device=dev, generator=gen).to(dtype) for K in (4096, 11008)} for M in (1, 16, 2048)}

    def lin_fwd(M, ent):
        K, N, qw, s, z, szp = ent
        return eng.forward_w3(xs[M][K], qw, s, z, szp, None)

    def run(M, n_layers):  # the block as the module tree runs it: qkv, o, QuantLlamaMLP (fused gate/up + SiLU * mul, then down_proj)
        outs = []
        for li in range(n_layers):
            outs += [lin_fwd(M, layers[li]["qkv"]), lin_fwd(M, layers[li]["o"])]
            K, N2, qw, szp = fused[li]
            outs.append(ops.mlp_gate_up_forward_w3(xs[M][K], qw, szp))
            outs.append(lin_fwd(M, layers[li]["down"]))
        return outs

    def run_unfused(M, n_layers):  # five WQLinear calls (no fused module): reported beside it
        return [lin_fwd(M, layers[li][nm]) for li in range(n_layers) for nm in ("qkv", "o", "gate", "up", "down")]

    def wbytes(K, N):
        return N * K * 3 // 8 + 2 * (K // 128) * N * 2

    K, F = synth.LLAMA2_7B["gate"]
    by = sum(wbytes(*synth.LLAMA2_7B[nm]) + synth.LLAMA2_7B[nm][0] * 2 + synth.LLAMA2_7B[nm][1] * 2 for nm in ("qkv", "o", "down"))
    by = (by + wbytes(K, 2 * F) + K * 2 + F * 2) * L_dec
    us = _graph_us(lambda: run(1, L_dec), stream, steps, warmup)
    us_u = _graph_us(lambda: run_unfused(1, L_dec), stream, steps, warmup)
    us16 = _graph_us(lambda: run(16, L_dec), stream, steps, warmup)  # batched decode of 16 rows: one weight pass on the skinny kernel (w3c tiles)
    by16 = by + L_dec * sum(15 * (K_ + N_) * 2 for (K_, N_) in (synth.LLAMA2_7B[nm] for nm in ("qkv", "o", "down")))
    us_p, us_pmin = _median_us(lambda: run(2048, L_pre), stream, iters)
    us_pu, _ = _median_us(lambda: run_unfused(2048, L_pre), stream, iters)
    fl = sum(2.0 * 2048 * K_ * N_ for (K_, N_) in (synth.LLAMA2_7B[nm] for nm in ("qkv", "o", "gate", "up", "down"))) * L_pre
    return {"workload": "Llama-2-7B W3A16 g128 bf16 (w3c tiles): qkv, o, QuantLlamaMLP(w_bit=3) = fused gate/up + SiLU*mul, down; every layer its own weights",
            "decode_m1": {"layers": L_dec, "launches": 4 * L_dec, "us_per_layer": round(us / L_dec, 2), "tok_s_32_layers": round(1e6 / (us / L_dec * 32), 1),
                          "algorithmic_bytes_per_layer": by // L_dec, "roofline": _hbm(by, us), "us_per_layer_five_unfused_calls": round(us_u / L_dec, 2)},
            "decode_m16": {"layers": L_dec, "us_per_layer": round(us16 / L_dec, 2), "tok_s_32_layers": round(16e6 / (us16 / L_dec * 32), 1), "roofline": _hbm(by16, us16)},
            "prefill_m2048": {"layers": L_pre, "ms_per_layer": round(us_p / L_pre / 1e3, 4), "tok_s_32_layers": round(2048 / (us_p / L_pre * 32 * 1e-6), 1),
                              "roofline": _mfma(fl, us_p, us_pmin), "ms_per_layer_five_unfused_calls": round(us_pu / L_pre / 1e3, 4)}}


def tp70b_world1(eng, dev, stream, steps, warmup, iters, dtype=torch.bfloat16):
    from llm_awq_amd import synth
    from llm_awq_amd.fused_mlp import interleave_gate_up
    L = 4
    layers = []

    def native(K, N, w, epi):
        szh, exact = eng.pack_szh_cdna4(w["scales"], w["scaled_zeros"], K)
        return (K, N, eng.repack_v2_to_cdna4(w["qweight"]), w["scales"], w["scaled_zeros"], eng.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K),
                szh if exact else None, epi)

    for li in range(L):
        ws = {nm: synth.random_wq(K, N, dtype=dtype, device=dev, seed=7000 + li * 16 + si, keep_q=False)
              for si, (nm, (K, N)) in enumerate(synth.LLAMA3_70B.items())}
        lin = [native(*synth.LLAMA3_70B["qkv"], ws["qkv"], 0), native(*synth.LLAMA3_70B["o"], ws["o"], 0)]
        g, u = ws["gate"], ws["up"]
        q, s, z = interleave_gate_up(g["qweight"], u["qweight"], g["scales"], u["scales"], g["scaled_zeros"], u["scaled_zeros"])
        lin.append(native(8192, 2 * 28672, dict(qweight=q, scales=s, scaled_zeros=z), 2))
        lin.append(native(*synth.LLAMA3_70B["down"], ws["down"], 0))
        layers.append(lin)
        del ws, g, u, q, s, z
    gen = torch.Generator(device=dev).manual_seed(404)
    xs = {M: {K: torch.randn(M, K, device=dev, generator=gen).to(dtype) for K in (8192, 28672)} for M in (1, 2048)}

    def run(M):
        outs = []
        for lin in layers:
            for (K, N, qw, s, z, szp, szh, epi) in lin:
                x = xs[M][K]
                if epi == 2:
                    outs.append(eng.mlp_gate_up_forward_cdna4(x, qw, szp, szh))
                elif M <= 8 and szh is not None:
                    outs.append(eng.decode_cdna4(x, qw, szh, None, 0))
                else:
                    outs.append(eng.forward_cdna4(x, qw, s, z, szp, None, szh if M >= 256 else None))
        return outs

    def algo(M, K, N, epi):
        b = N * K // 2 + 2 * (K // 128) * N * 2 + M * K * 2 + M * N * 2
        return b - (M * (N // 2) * 2 if epi else 0)

    us = _graph_us(lambda: run(1), stream, steps, warmup)
    by = sum(algo(1, K, N, epi) for (K, N, *_r, epi) in layers[0]) * L
    us_p, us_pmin = _median_us(lambda: run(2048), stream, iters)
    fl = sum(2.0 * 2048 * K * N for (K, N, *_r) in layers[0]) * L
    return {"workload": "Llama-3-70B W4A16 g128 bf16, the UNSHARDED block shapes (world size 1 of BASELINE.json config 4): qkv, o, fused gate/up + SiLU*mul, down",
            "decode_m1": {"layers": L, "launches": 4 * L, "us_per_layer": round(us / L, 2), "tok_s_80_layers": round(1e6 / (us / L * 80), 1),
                          "algorithmic_bytes_per_layer": by // L, "roofline": _hbm(by, us)},
            "prefill_m2048": {"layers": L, "ms_per_layer": round(us_p / L / 1e3, 4), "tok_s_80_layers": round(2048 / (us_p / L * 80 * 1e-6), 1),
                              "roofline": _mfma(fl, us_p, us_pmin)}}


def moe_mixtral(eng, dev, stream, iters, dtype=torch.bfloat16):
    from llm_awq_amd import ops
    from llm_awq_amd.fused_mlp import interleave_gate_up
    from llm_awq_amd.moe import sort_by_expert
    E, H, F, T = 8, 4096, 14336, 2048
    gen = torch.Generator(device=dev).manual_seed(505)

    def rand_lin(K, N):
        q = torch.randint(0, 16, (N, K), dtype=torch.uint8, device=dev, generator=gen)
        s, z = _rand_sz(K, N, 15, dtype, dev, gen)
        return ops.pack_v2(q), s, z

    qi, si, zi, q2, s2, z2 = [], [], [], [], [], []
    for _e in range(E):
        (qa, sa, za), (qb, sb, zb) = rand_lin(H, F), rand_lin(H, F)
        qq, sq, zq = interleave_gate_up(qa, qb, sa, sb, za, zb)
        qi.append(eng.repack_v2_to_cdna4(qq))
        si.append(sq)
        zi.append(zq)
        qd, sd, zd = rand_lin(F, H)
        q2.append(eng.repack_v2_to_cdna4(qd))
        s2.append(sd)
        z2.append(zd)
        del qa, qb, qq, qd
    qwi, sI, zI = torch.stack(qi), torch.stack(si), torch.stack(zi)
    szpi = torch.stack([eng.pack_sz_cdna4(si[e], zi[e], H) for e in range(E)])
    qw2, sD, zD = torch.stack(q2), torch.stack(s2), torch.stack(z2)
    szp2 = torch.stack([eng.pack_sz_cdna4(s2[e], z2[e], F) for e in range(E)])

    def stacked_szh(ss, zz, K):  # the experts' sz_half side buffers (what GroupedGateUp / GroupedWQLinear build), None unless every expert is exact
        hs = [eng.pack_szh_cdna4(ss[e], zz[e], K) for e in range(E)]
        return torch.stack([h for (h, _ok) in hs]) if all(ok for (_h, ok) in hs) else None

    szhi, szh2 = stacked_szh(si, zi, H), stacked_szh(s2, z2, F)
    del qi, q2
    rgen = torch.Generator(device=dev).manual_seed(1234)  # the same routing in every run
    ids = torch.stack([torch.randperm(E, device=dev, generator=rgen)[:2] for _ in range(T)])
    _order, off = sort_by_expert(ids, E)
    cnts = (off[1:] - off[:-1]).tolist()
    xs = torch.randn(2 * T, H, device=dev, generator=gen).to(dtype)
    h = ops.moe_mlp_gate_up_cdna4(xs, qwi, sI, zI, szpi, off, sz_half=szhi)
    us1, us1m = _median_us(lambda: ops.moe_mlp_gate_up_cdna4(xs, qwi, sI, zI, szpi, off, sz_half=szhi), stream, iters)
    us2, us2m = _median_us(lambda: ops.moe_forward_cdna4(h, qw2, sD, zD, szp2, off, sz_half=szh2), stream, iters)
    f1, f2 = 2.0 * 2 * T * (2 * F) * H, 2.0 * 2 * T * H * F
    return {"workload": "Mixtral-8x7B W4A16 g128 bf16 expert block: E = 8, top-2, 2048 tokens = 4096 sorted rows, seeded routing",
            "rows_per_expert": cnts, "row_tiles_256": sum((c + 255) // 256 for c in cnts),
            "w1_w3_fused": {"launch": "one grouped launch, w1 / w3 rows interleaved 8 + 8, SiLU*mul in the tile epilogue", "us": round(us1, 1), "roofline": _mfma(f1, us1, us1m)},
            "w2": {"launch": "one grouped launch", "us": round(us2, 1), "roofline": _mfma(f2, us2, us2m)},
            "block": {"us": round(us1 + us2, 1), "tok_s_32_layers": round(T / ((us1 + us2) * 32 * 1e-6), 1), "roofline": _mfma(f1 + f2, us1 + us2)}}


def run_all(eng, dev, stream, steps, warmup, iters):
    """the three legs, each isolated: a failure is reported in its key instead of taking the headline line down"""
    out = {}
    for key, fn in (("w3_llama2_7b", lambda: w3_llama2_7b(eng, dev, stream, steps, warmup, iters)),
                    ("tp70b_world1", lambda: tp70b_world1(eng, dev, stream, steps, warmup, iters)),
                    ("moe_mixtral", lambda: moe_mixtral(eng, dev, stream, iters))):
        try:
            out[key] = fn()
        except Exception as exc:  # noqa: BLE001
            out[key] = {"error": f"{type(exc).__name__}: {exc}"}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out
