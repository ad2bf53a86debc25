"""Tensor-parallel WQLinear over RCCL / xGMI (one process per GPU, torch.distributed backend "nccl").

The reference has no multi-GPU path for this code (layer placement only), so this is
new capability defined by BASELINE.json's north star: the `[N, K]` weight is sharded and the partial
products are summed with ONE all-reduce.

  * "row"    parallel (K-sharded): rank r owns k in [r*K/p, (r+1)*K/p) -- a plain column slice of the
             int16 `[N/4, K]` buffer (cut at a multiple of 128 so whole quantisation groups and whole 64-k
             pack blocks stay together) and the matching rows of scales / scaled_zeros.  It consumes
             x[..., k-slice] and produces a partial y [M, N]; all-reduce(sum) over ranks; bias once, after.
  * "column" parallel (N-sharded): rank r owns output rows [r*N/p, ...): no communication, sharded output.
             Used for qkv / gate / up feeding row-parallel o / down (Megatron pairing: 2 all-reduces per
             decoder block instead of 5).

xGMI is point-to-point: decode all-reduces are tiny ([1, 4096] bf16 = 8 KiB, latency bound), prefill ones are
MBs (bandwidth bound per link).  RCCL picks the algorithm; we keep one fused tensor per collective.
"""
from __future__ import annotations

import time
from typing import Callable, Optional

import torch
import torch.nn as nn

from .qmodule import WQLinear, calculate_zeros_width

GROUP = 128


def shard_bounds(total: int, world: int, rank: int, multiple: int):
    """Even split of `total` in units of `multiple`; the first (units % world) ranks get one more unit."""
    units = total // multiple
    assert units * multiple == total, f"{total} is not a multiple of {multiple}"
    base, extra = divmod(units, world)
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo * multiple, hi * multiple


def shard_row_parallel(qweight, scales, scaled_zeros, world: int, rank: int):
    """K-shard of v2 buffers -> (qweight [N/4, Kr], scales [Gpad_r, N], scaled_zeros [Gpad_r, N], (k0, k1))."""
    K = qweight.shape[1]
    k0, k1 = shard_bounds(K, world, rank, GROUP)
    assert k1 > k0, "more ranks than 128-k groups"
    g0, g1 = k0 // GROUP, k1 // GROUP
    gpad = calculate_zeros_width(k1 - k0, GROUP) * 8
    N = scales.shape[1]
    s = torch.zeros(gpad, N, dtype=scales.dtype, device=scales.device)
    z = torch.zeros(gpad, N, dtype=scales.dtype, device=scales.device)
    s[: g1 - g0] = scales[g0:g1]
    z[: g1 - g0] = scaled_zeros[g0:g1]
    return qweight[:, k0:k1].contiguous(), s, z, (k0, k1)


def shard_column_parallel(qweight, scales, scaled_zeros, world: int, rank: int, multiple: int = 16):
    """N-shard (cut at multiples of 16 rows = one MFMA slab) -> (qweight, scales, scaled_zeros, (n0, n1))."""
    N = qweight.shape[0] * 4
    n0, n1 = shard_bounds(N, world, rank, multiple)
    return (qweight[n0 // 4: n1 // 4].contiguous(), scales[:, n0:n1].contiguous(),
            scaled_zeros[:, n0:n1].contiguous(), (n0, n1))


def shard_stacked_column_parallel(qweight, scales, scaled_zeros, world: int, rank: int, parts: int = 2, multiple: int = 16):
    """N-shard of a buffer that stacks `parts` projections along N (the fused gate/up pair, or q/k/v as tinychat's
    make_quant_attn concatenates them, fused_attn.py:566-572): every projection is sharded on its own and the rank's
    pieces are re-stacked, so rank r holds [gate_r; up_r] and the fused SiLU*mul epilogue still pairs matching rows."""
    N = qweight.shape[0] * 4
    assert N % parts == 0
    sec = N // parts
    qs, ss, zs, bounds = [], [], [], []
    for p in range(parts):
        n0, n1 = shard_bounds(sec, world, rank, multiple)
        lo, hi = p * sec + n0, p * sec + n1
        qs.append(qweight[lo // 4: hi // 4])
        ss.append(scales[:, lo:hi])
        zs.append(scaled_zeros[:, lo:hi])
        bounds.append((lo, hi))
    return torch.cat(qs, 0).contiguous(), torch.cat(ss, 1).contiguous(), torch.cat(zs, 1).contiguous(), bounds


def shard_w3c(qweight, scales, scaled_zeros, mode: str, world: int, rank: int):
    """K- ("row") or N- ("column") shard of a 3-bit layer.  The w3c buffer is a [N/16, K/128] grid of self-contained 768-byte tiles
    (include/awq_cdna4.h), so a K cut at a multiple of 128 / an N cut at a multiple of 16 is a plain slice of that grid: no
    unpacking.  -> (qweight int16 [Nr/4, 3 Kr/4], scales, scaled_zeros, (lo, hi))."""
    N, K = qweight.shape[0] * 4, qweight.shape[1] * 4 // 3
    tiles = qweight.contiguous().reshape(N // 16, K // GROUP, 384)  # 384 int16 = one tile
    if mode == "row":
        k0, k1 = shard_bounds(K, world, rank, GROUP)
        assert k1 > k0, "more ranks than 128-k groups"
        g0, g1 = k0 // GROUP, k1 // GROUP
        gpad = calculate_zeros_width(k1 - k0, GROUP) * 8
        s = torch.zeros(gpad, N, dtype=scales.dtype, device=scales.device)
        z = torch.zeros(gpad, N, dtype=scales.dtype, device=scales.device)
        s[: g1 - g0] = scales[g0:g1]
        z[: g1 - g0] = scaled_zeros[g0:g1]
        return tiles[:, g0:g1].contiguous().reshape(N // 4, (k1 - k0) * 3 // 4), s, z, (k0, k1)
    n0, n1 = shard_bounds(N, world, rank, 16)
    return (tiles[n0 // 16: n1 // 16].contiguous().reshape((n1 - n0) // 4, K * 3 // 4), scales[:, n0:n1].contiguous(),
            scaled_zeros[:, n0:n1].contiguous(), (n0, n1))


class TPWQLinear(nn.Module):
    """A WQLinear shard + its collective.

    Row mode (K-sharded, the north star's form): the shard computes its product as an UNROUNDED fp32 partial
    (`awq_w4a16_partial_cdna4`, `awq_w3a16_partial` for a 3-bit layer: the same decode / skinny / prefill kernels with an fp32 epilogue), the partials are summed in fp32 --
    `OneShotAllReduce.reduce_f32` for the latency-class messages, the group's all-reduce on the float tensor above that -- and the sum is
    rounded to T ONCE, then the bias is added in T: what the single-device kernel does with its one accumulator.  (Rounding every rank's
    partial to T first puts a bf16 output 2.6-2.9e-3 norm-wise from the single-device result, outside the 1e-3 budget of SURVEY.md 8(e);
    fp32 partials: ~2e-6.)
    Column mode (N-sharded): the shard is a plain WQLinear on its output rows; no communication.

    The arithmetic hooks `_shard_product` (column mode: T output of the shard), `_shard_partial` (row mode: fp32 partial) and `_round_bias`
    run the HIP kernels and nothing else; the CPU tests of the sharding / collective logic (no GPU in that container) override them with the
    oracle in a subclass that lives in tests/helpers.py, not here."""

    def __init__(self, full: WQLinear, mode: str, group=None, world: Optional[int] = None, rank: Optional[int] = None,
                 reducer: Optional[Callable] = None):
        super().__init__()
        import torch.distributed as dist

        assert mode in ("row", "column")
        self.mode, self.group = mode, group
        have_dist = dist.is_available() and dist.is_initialized()
        if (world is None or rank is None) and not have_dist:
            raise RuntimeError("TPWQLinear needs an initialised process group, or explicit world= and rank= (offline sharding)")
        self.world = world if world is not None else dist.get_world_size(group)
        self.rank = rank if rank is not None else dist.get_rank(group)
        self.in_features, self.out_features = full.in_features, full.out_features
        w3 = getattr(full, "w_bit", 4) == 3
        if hasattr(full, "_refuse_converted"):
            full._refuse_converted("TPWQLinear")
        # W4: the slicing is defined on the REFERENCE (v2) interleave, where a K cut at a multiple of 64 / an N cut at a multiple of 4
        # is a plain column / row slice of the int16 buffer (the cdna4 tiling permutes across those cuts: converted back and forth);
        # W3: the w3c tiles are cut on their own grid (shard_w3c)
        relayout = not w3 and getattr(full, "layout", "v2") == "cdna4"
        if relayout:
            full.to_v2()
        if w3:
            qw, s, z, self.bounds = shard_w3c(full.qweight, full.scales, full.scaled_zeros, mode, self.world, self.rank)
            k_local, n_local = qw.shape[1] * 4 // 3, qw.shape[0] * 4
        else:
            fn = shard_row_parallel if mode == "row" else shard_column_parallel
            qw, s, z, self.bounds = fn(full.qweight, full.scales, full.scaled_zeros, self.world, self.rank)
            k_local, n_local = qw.shape[1], qw.shape[0] * 4
        self.shard = WQLinear(full.w_bit, full.group_size, k_local, n_local, False, qw.device, dtype=s.dtype)
        self.shard.qweight, self.shard.scales, self.shard.scaled_zeros = qw, s, z
        if relayout:
            full.to_cdna4()
        cdna4_able = n_local % 16 == 0 and k_local % 128 == 0 and self.shard.group_size == 128 and s.dtype in (torch.float16, torch.bfloat16)
        if qw.is_cuda and not w3:
            if mode == "row" and not cdna4_able:
                raise ValueError("a K-sharded WQLinear runs the cdna4 kernels' fp32-partial epilogue: it needs out_features % 16 == 0, "
                                 "group_size 128 and fp16 / bf16 scales")
            if cdna4_able and (relayout or mode == "row"):
                self.shard.to_cdna4()  # the shard runs the same kernels the unsharded module did
        if full.bias is None:
            self.bias = None
        else:
            self.bias = full.bias if mode == "row" else full.bias[self.bounds[0]: self.bounds[1]].contiguous()
        # reducer: an llm_awq_amd.oneshot.OneShotAllReduce (reduce_f32) or None = torch.distributed.all_reduce on the fp32 partial (RCCL on
        # GPUs, gloo in the CPU tests).  Default on GPUs with world > 1: the group's shared one-shot reducer, built on the SHARD's device
        # (AWQ_ONESHOT=0, a box without fine-grained memory / hipIpc, or a world that is not the group's -> None)
        if (reducer is None and mode == "row" and self.world > 1 and qw.is_cuda and have_dist
                and self.world == dist.get_world_size(group)):
            reducer = default_reducer(dist, group, qw.device)
        self._reducer = reducer

    def _reduce_round(self, y32, dtype):
        """fp32 partial -> T(sum over ranks) + bias.  Small messages (decode): the one-shot reducer / one all-reduce on the fp32 partial.  From
        RS_AG_MIN_BYTES (prompts: 64 MiB of fp32 at 2048 rows x 8192) the sum is taken as reduce-scatter(fp32) -> round once to T (+ bias) on the
        rank's own rows -> all-gather(T): the same single rounding, 25 % fewer bytes on the links than an fp32 all-reduce."""
        import torch.distributed as dist
        if self.world > 1 and y32.numel() * 4 >= RS_AG_MIN_BYTES and (self._reducer is None or hasattr(self._reducer, "reduce_f32")):
            y = reduce_scatter_round_gather(y32, dtype, self.bias, self._round_rows, dist, self.group, self.world)
            if y is not None:
                return y
        if self._reducer is not None and hasattr(self._reducer, "reduce_f32"):
            return self._reducer.reduce_f32(y32, dtype, self.bias)
        if self.world > 1:
            if self._reducer is not None:
                y32 = self._reducer(y32)
            else:
                dist.all_reduce(y32, op=dist.ReduceOp.SUM, group=self.group)
        return self._round_bias(y32, dtype)

    def _round_rows(self, y32_rows, dtype, bias):
        """T(fp32 rows) + bias in T on a row block of the sum (the reduce-scatter path's middle step): awq_round_bias_f32"""
        from . import ops
        return ops.round_bias_f32(y32_rows, dtype, bias)

    def _round_bias(self, y32, dtype):
        """T(fp32 sum) + bias in T: awq_round_bias_f32"""
        from . import ops
        return ops.round_bias_f32(y32, dtype, self.bias)

    def _shard_product(self, x):
        """column mode: the shard's T output (a plain WQLinear on its output rows)"""
        return self.shard(x)

    @torch.no_grad()
    def forward(self, x, input_is_sharded: bool = False):
        if self.mode == "column":
            y = self._shard_product(x)
            return y + self.bias if self.bias is not None else y
        return self._reduce_round(self.partial(x, input_is_sharded), x.dtype)

    @torch.no_grad()
    def partial(self, x, input_is_sharded: bool = False):
        """row mode: this rank's fp32 partial [..., N] of x . W^T (unrounded, no bias)"""
        assert self.mode == "row"
        if not input_is_sharded:
            x = x[..., self.bounds[0]: self.bounds[1]].contiguous()
        return self._shard_partial(x)

    def _shard_partial(self, x):
        """the K shard's product as unrounded fp32 (awq_w4a16_partial_cdna4 on the shard's cdna4 buffers, awq_w3a16_partial on w3c tiles)"""
        from . import ops
        sh = self.shard
        if not x.is_contiguous():
            x = x.contiguous()
        key = sh._side_key()
        if sh.sz_cdna4 is None or getattr(sh, "_sz_key", None) != key:
            sh.sz_cdna4 = ops.pack_sz_cdna4(sh.scales, sh.scaled_zeros, sh.in_features)
            sh.szh_cdna4, sh._sz_key = None, key
        if sh.w_bit == 3:
            return ops.partial_w3(x, sh.qweight, sh.sz_cdna4)
        if sh.szh_cdna4 is None and not torch.cuda.is_current_stream_capturing():
            sh._build_szh(ops)
        szh = sh.szh_cdna4 if (sh.szh_cdna4 is not None and sh.szh_cdna4 is not False) else None
        return ops.partial_cdna4(x, sh.qweight, sh.sz_cdna4, szh)

    def check(self):
        """after a synchronize: raise if a one-shot round of this module's reducer timed out (its output was poisoned with NaNs; every
        later call of that communicator poisons its output too).  forward() cannot poll the status word without a host sync: serving
        loops call this at their own sync points (once per generated token is enough)."""
        if self._reducer is not None and hasattr(self._reducer, "check"):
            self._reducer.check()


RS_AG_MIN_BYTES = 1 << 20  # fp32 partials from this size on are summed by reduce-scatter -> round -> all-gather (below: latency-bound, one all-reduce)


def reduce_scatter_round_gather(y32, dtype, bias, round_rows, dist, group, world):
    """[..., N] fp32 partial of every rank -> [..., N] T(sum over ranks) (+ bias) on every rank:
       reduce-scatter the fp32 partials over row blocks (rank r ends up with the fp32 SUM of rows [r R, (r + 1) R)), round that block once to T and
       add the bias (`round_rows`: awq_round_bias_f32 on the GPU), all-gather the T blocks.  Bytes per rank on the links: (w - 1) / w x (4 + 2) per
       element against 2 (w - 1) / w x 4 of a ring all-reduce in fp32; the numerics are the all-reduce path's (fp32 sum, ONE rounding, bias in T).
       Rows are padded to a multiple of the world size.  Returns None when the layout does not allow it (no rows to split)."""
    n = y32.shape[-1]
    flat = y32.reshape(-1, n)
    m = flat.shape[0]
    if m < world:
        return None
    rows = (m + world - 1) // world
    if rows * world != m:
        pad = torch.zeros(rows * world - m, n, dtype=flat.dtype, device=flat.device)
        flat = torch.cat([flat, pad], 0)
    mine = torch.empty(rows, n, dtype=torch.float32, device=flat.device)
    dist.reduce_scatter_tensor(mine, flat.contiguous(), op=dist.ReduceOp.SUM, group=group)
    block = round_rows(mine, dtype, bias)
    out = torch.empty(rows * world, n, dtype=dtype, device=flat.device)
    dist.all_gather_into_tensor(out, block.contiguous(), group=group)
    return out[:m].reshape(*y32.shape[:-1], n)


_REDUCERS = {}  # id(group) -> (weakref to the group or None, reducer): an entry whose group object died is rebuilt, never reused


def default_reducer(dist, group=None, device=None):
    """One OneShotAllReduce per (process group, device), built at the first row-parallel module (a collective: every rank builds its
    modules in the same order), or None where it is disabled / unavailable."""
    import weakref
    key = (id(group) if group is not None else 0, str(device))
    ent = _REDUCERS.get(key)
    if ent is not None:
        ref, red = ent
        if (ref is None and group is None) or (ref is not None and ref() is group):
            if red is None or red.local is not None:
                return red
        if red is not None:
            red.close()
        del _REDUCERS[key]
    from . import oneshot
    red = oneshot.make_reducer(dist, group, device=device)
    try:
        ref = weakref.ref(group) if group is not None else None
    except TypeError:  # (a group object that cannot be weakly referenced: keyed by id only, closed reducers are still never reused)
        ref = None if group is None else (lambda g=group: g)
    _REDUCERS[key] = (ref, red)
    return red


def close_reducers():
    """close every cached one-shot reducer (call before destroying the process group)"""
    for _ref, red in list(_REDUCERS.values()):
        if red is not None:
            red.close()
    _REDUCERS.clear()


# ---------------------------------------------------------------------------------------------------
# bench leg for N > 1 (called from bench.py): Megatron-paired tensor parallelism of the decoder block's quantised linears
# ---------------------------------------------------------------------------------------------------

def _block(shapes):
    """(name, K, N, mode) of one decoder block: qkv and the gate/up pair N-sharded (no communication), o / down K-sharded + all-reduce"""
    return [("qkv",) + shapes["qkv"] + ("column",), ("o",) + shapes["o"] + ("row",),
            ("gate_up", shapes["gate"][0], 2 * shapes["gate"][1], "column"), ("down",) + shapes["down"] + ("row",)]


def _build_shards(eng, block, layers, shard_world, rank, dev, dtype, seed0=0):
    from . import synth
    shards = []
    for li in range(layers):
        for si, (name, K, N, mode) in enumerate(block):
            if mode == "row":
                k0, k1 = shard_bounds(K, shard_world, rank, GROUP)
                kl, nl = k1 - k0, N
            else:
                n0, n1 = shard_bounds(N, shard_world, rank, 32)  # 32: the stacked gate/up pair splits in matching 16-row slabs
                kl, nl = K, n1 - n0
            w = synth.random_wq(kl, nl, dtype=dtype, device=dev, seed=seed0 + (li * 16 + si) * 64 + rank, keep_q=False)
            qw = eng.repack_v2_to_cdna4(w["qweight"])
            szp = eng.pack_sz_cdna4(w["scales"], w["scaled_zeros"], kl)
            szh, exact = eng.pack_szh_cdna4(w["scales"], w["scaled_zeros"], kl)  # the streaming decode kernel's side buffer
            shards.append((name, kl, nl, qw, w["scales"], w["scaled_zeros"], szp, mode, szh if exact else None))
            del w
    return shards


def _make_pass(eng, shards, xs, reducer, dist, world):
    from . import ops

    def run_pass():
        outs = []
        for (name, kl, nl, qw, s, sz, szp, mode, szh) in shards:
            x = xs[kl]
            m = x.numel() // kl
            if mode == "row":
                # K shard: fp32 partial -> sum over ranks in fp32 -> ONE rounding to T (TPWQLinear.forward's path, no module overhead)
                y32 = ops.partial_cdna4(x, qw, szp, szh)
                y = None
                if world > 1 and y32.numel() * 4 >= RS_AG_MIN_BYTES:  # prompts: reduce-scatter(fp32) -> one rounding -> all-gather(T), as TPWQLinear does
                    y = reduce_scatter_round_gather(y32, x.dtype, None, ops.round_bias_f32, dist, None, world)
                if y is not None:
                    pass
                elif reducer is not None:
                    y = reducer.reduce_f32(y32, x.dtype)
                else:
                    if world > 1:
                        dist.all_reduce(y32)
                    y = ops.round_bias_f32(y32, x.dtype)
            elif name == "gate_up":  # the shard's rows are taken as QuantLlamaMLP's 8 + 8 interleaved gate / up pair (synthetic weights)
                y = eng.mlp_gate_up_forward_cdna4(x, qw, szp, szh)
            elif m <= 8 and szh is not None:
                y = eng.decode_cdna4(x, qw, szh, None, 0)
            else:
                y = eng.forward_cdna4(x, qw, s, sz, szp, None)
            outs.append(y)
        return outs
    return run_pass


_CLOCK_RAMP_STEPS = 60  # untimed steps before the timed region, the W warmup steps included (every rank runs the same count: the collectives stay matched)


def _timed(run_pass, steps, warmup, dist, dev, use_graph, rank):
    """K timed steps bracketed by barrier + synchronize on both sides, max over ranks; returns (ms_per_step, graphed)"""
    side = torch.cuda.Stream(device=dev)
    graph = None
    with torch.cuda.stream(side):
        for _ in range(2):
            run_pass()  # communicator + lazy init outside capture
        torch.cuda.synchronize()
        if use_graph:  # (the one-shot reducer keeps its round counter on the device: replay-safe)
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    keep = run_pass()  # noqa: F841
                graph.replay()
                torch.cuda.synchronize()
            except Exception as e:  # RCCL capture not available -> eager launches
                import sys
                print(f"[bench rank {rank}] graph capture of the TP pass failed ({type(e).__name__}: {e}); eager", file=sys.stderr)
                graph = None
                torch.cuda.synchronize()
        # all ranks must agree on the mode
        flag = torch.tensor([1 if graph is not None else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() == 0:
            graph = None
        step = (lambda: graph.replay()) if graph is not None else run_pass
        for _ in range(max(0, _CLOCK_RAMP_STEPS - warmup)):  # (untimed, as in bench.py's single-GPU leg: a --warmup 5 ends before the chip has left its idle clocks)
            step()
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return tmax.item() * 1e3 / steps, graph is not None


def _allreduce_record(dist, reducer, dev, world, numel, dtype, iters=200):
    """one message class on its own: microseconds per reduction of `numel` fp32 partials to T (max over ranks) -- the group's all-reduce on
    the float tensor + the rounding kernel, and, where it serves the size, the one-shot exchange (sum + rounding in one kernel)"""
    from . import ops
    rec = {"bytes": numel * 4, "partials": "fp32 (rounded to T once, after the sum)", "ranks": world}
    if world <= 1:
        return rec
    t = torch.ones(numel, device=dev, dtype=torch.float32)

    def timeit(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return dt.item() * 1e6 / iters

    def rccl():
        dist.all_reduce(t.fill_(1.0))
        return ops.round_bias_f32(t, dtype)

    rec["rccl_us"] = round(timeit(rccl), 2)
    if reducer is not None and numel * 4 <= reducer.max_bytes:
        rec["oneshot_us"] = round(timeit(lambda: reducer.reduce_f32(t, dtype)), 2)
    return rec


def run_tp_bench(args, eng, dist, rank, world, dev, shapes, algo_bytes):
    """Headline: Llama-3-8B decode (M = 1) under Megatron-paired tensor parallelism in the layout the rewritten repacker emits.
    Beside it (`tp70b`): the configuration BASELINE.json names for TP -- Llama-3-70B shapes sharded the same way, decode M = 1 and
    prefill M = 2048 -- and (`allreduce`) the two message classes on their own: 16 KiB (decode, one-shot / RCCL) and 32 MiB
    (prefill, RCCL).  Every rank builds ONLY its own shards (weights are synthetic)."""
    from . import oneshot, synth

    import os
    dtype = torch.bfloat16
    L = args.layers
    # AWQ_BENCH_SHARD_WORLD=W on fewer ranks: every rank times the shard shapes of a W-way split (rank r of W); the all-reduces still
    # run over the real ranks.  It gives the compute floor of a W-GPU step on boxes that have fewer GPUs; `config.shard_world` says so.
    shard_world = int(os.environ.get("AWQ_BENCH_SHARD_WORLD", world))
    reducer = oneshot.make_reducer(dist, None, 64 * 1024, dev)  # default for <= 64 KiB when world > 1 (AWQ_ONESHOT=0: RCCL)
    g = torch.Generator(device=dev).manual_seed(1 + rank)

    def make_xs(shards, M):
        xs = {}
        for (_nm, kl, *_r) in shards:
            if kl not in xs:
                xs[kl] = torch.randn(M, kl, device=dev, generator=g).to(dtype)
        return xs

    def local_ms(run_pass, steps, warmup):
        """rank-local timing (no collective inside): graph replay where it captures"""
        side = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(side):
            run_pass()
            torch.cuda.synchronize()
            gr = None
            if not args.no_graph:
                try:
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr, stream=side):
                        keep = run_pass()  # noqa: F841
                except Exception:  # noqa: BLE001
                    gr = None
                    torch.cuda.synchronize()
            step = (lambda: gr.replay()) if gr is not None else run_pass
            for _ in range(max(warmup, _CLOCK_RAMP_STEPS)):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) * 1e3 / steps

    def measure(reducer):
        # ---------------- Llama-3-70B TP (BASELINE.json configs[3]): the HEADLINE of the N > 1 line -- decode M = 1 (K timed steps), prefill M = 2048 beside it ----------------
        L70 = max(1, int(os.environ.get("AWQ_BENCH_TP70B_LAYERS", "8")))
        sh70 = _build_shards(eng, _block(synth.LLAMA3_70B), L70, shard_world, rank, dev, dtype, seed0=1 << 20)
        b70 = sum(algo_bytes(1, kl, nl) for (_nm, kl, nl, *_r) in sh70)
        launches70 = len(sh70)
        d_ms, d_graph = _timed(_make_pass(eng, sh70, make_xs(sh70, 1), reducer, dist, world), args.steps, args.warmup, dist, dev, not args.no_graph, rank)
        Mp = 2048
        p_ms, _pg = _timed(_make_pass(eng, sh70, make_xs(sh70, Mp), reducer, dist, world), 2, 1, dist, dev, False, rank)
        flops_rank = sum(2.0 * Mp * kl * nl for (_nm, kl, nl, *_r) in sh70)
        tp70 = {"workload": f"Llama-3-70B W4A16 g128 bf16, {L70} of 80 decoder blocks timed, tensor parallel over {world} GPUs"
                            + (f" (shard shapes of a {shard_world}-way split)" if shard_world != world else ""),
                "layers_timed": L70, "ms_per_step_timed_layers": d_ms, "graph": d_graph, "launches_per_step": launches70, "bytes_rank": b70,
                "decode": {"m": 1, "ms_per_step_80_layers": round(d_ms * 80 / L70, 4), "tok_s": round(1e3 / (d_ms * 80 / L70), 2), "graph": d_graph,
                           "hbm_gbs_per_gpu": round(b70 / (d_ms * 1e-3) / 1e9, 1), "hbm_frac_per_gpu": round(b70 / (d_ms * 1e-3) / 1e9 / 8000.0, 4),
                           "allreduce_bytes": 8192 * 4, "allreduces_per_token": 2 * 80},
                "prefill": {"m": Mp, "ms_per_pass_80_layers": round(p_ms * 80 / L70, 3), "tok_s": round(Mp / (p_ms * 80 / L70 * 1e-3), 1),
                            "mfma_tflops_per_gpu": round(flops_rank / (p_ms * 1e-3) / 1e12, 1),
                            "mfma_frac_per_gpu": round(flops_rank / (p_ms * 1e-3) / 1e12 / 2500.0, 4),
                            "reduction": "reduce-scatter (fp32) -> one rounding -> all-gather (T)" if world > 1 and Mp * 8192 * 4 >= RS_AG_MIN_BYTES else "all-reduce (fp32) -> one rounding",
                            "allreduce_bytes": Mp * 8192 * 4, "allreduces_per_pass": 2 * 80}}
        del sh70
        torch.cuda.empty_cache()
        # the SAME workload on ONE GPU (rank 0 alone, unsharded shapes, no collective): the N = 1 point of this line's own scaling curve
        if world > 1 and shard_world == world and os.environ.get("AWQ_BENCH_N1_REF", "1") != "0":
            if rank == 0:
                sh1 = _build_shards(eng, _block(synth.LLAMA3_70B), L70, 1, 0, dev, dtype, seed0=1 << 20)
                ms1 = local_ms(_make_pass(eng, sh1, make_xs(sh1, 1), None, dist, 1), max(5, args.steps // 2), max(2, args.warmup // 2))
                tp70["one_gpu_reference"] = {"note": "the unsharded 70B shapes on rank 0's GPU alone, same layers, same kernels, measured in this run",
                                             "decode_tok_s": round(1e3 / (ms1 * 80 / L70), 2), "ms_per_step_80_layers": round(ms1 * 80 / L70, 4)}
                del sh1
                torch.cuda.empty_cache()
            dist.barrier()

        # ---------------- Llama-3-8B decode under the same sharding (the N = 1 line's model; 64 all-reduces of 8 KiB per token: latency bound) ----------------
        shards = _build_shards(eng, _block(synth.LLAMA3_8B), L, shard_world, rank, dev, dtype)
        ms_per_step, graphed = _timed(_make_pass(eng, shards, make_xs(shards, 1), reducer, dist, world), max(5, args.steps // 2), max(2, args.warmup // 2),
                                      dist, dev, not args.no_graph, rank)
        bytes_rank = sum(algo_bytes(1, kl, nl) for (_nm, kl, nl, *_r) in shards)
        gbs_rank = bytes_rank / (ms_per_step * 1e-3) / 1e9
        launches = len(shards)
        del shards
        torch.cuda.empty_cache()

        ar = {"kind": "oneshot (peer-mapped exchange buffers, csrc/awq_oneshot.hip) for <= 64 KiB, RCCL above" if reducer is not None
                      else "rccl (torch.distributed.all_reduce)",
              "rccl_ranks": world, "per_step": 2 * 80,
              "decode_8b": _allreduce_record(dist, reducer, dev, world, 4096, dtype),
              "decode_70b": _allreduce_record(dist, reducer, dev, world, 8192, dtype),
              "prefill_70b_m2048": _allreduce_record(dist, reducer, dev, world, 2048 * 8192, dtype, iters=10),
              "weight_bytes_per_rank": int(bytes_rank)}
        timed_out = 0
        if reducer is not None:  # a timed-out round poisons its output and must not be reported as a timing: every rank learns of it
            torch.cuda.synchronize()
            bad = reducer.status.clone()
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            timed_out = int(bad.item())
        return ms_per_step, graphed, gbs_rank, launches, bytes_rank, tp70, ar, timed_out

    ms_per_step, graphed, gbs_rank, launches, bytes_rank, tp70, ar, timed_out = measure(reducer)
    if timed_out:  # (agreed by all ranks) the peer-mapped reducer lost a round on this fabric: the whole leg again over RCCL
        import sys
        print(f"[bench rank {rank}] one-shot all-reduce timed out waiting for a peer; re-running the leg with RCCL", file=sys.stderr)
        reducer.close()
        ms_per_step, graphed, gbs_rank, launches, bytes_rank, tp70, ar, _t = measure(None)
        ar["oneshot"] = "timed out on this box: figures above are RCCL"
    L70, d_ms = tp70["layers_timed"], tp70.pop("ms_per_step_timed_layers")
    b70, launches70 = tp70.pop("bytes_rank"), tp70.pop("launches_per_step")
    gbs70 = b70 / (d_ms * 1e-3) / 1e9
    return {"metric": "W4A16 decode+prefill tok/s, Llama-3-70B TP over xGMI (BASELINE.json config 4); achieved %HBM (GEMV) / %MFMA (GEMM) per GPU",
            "value": round(1e3 / (d_ms * 80 / L70), 2),
            "unit": "decode tok/s of the whole job (the 400 quantised linears of one Llama-3-70B token: 80 x {qkv, o, gate, up, down}; attention/norm/lm_head off-path; "
                    f"{L70} of 80 blocks timed, scaled)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(d_ms, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"Llama-3-70B W4A16 TP={world} over xGMI (column-sharded WQLinear + RCCL all-reduce): qkv and gate/up N-sharded, "
                                   "o / down K-sharded, fp32 partials summed over the ranks and rounded once",
                       "layers_timed": L70, "layers_model": 80, "decode_m": 1, "graph": tp70["graph"], "layout": "cdna4",
                       "fused_gate_up_silu_mul": True, "launches_per_step": launches70, "parallelism": f"tp{world}",
                       "allreduces_per_step": 2 * L70,
                       "n1_line": "the N = 1 line of this script times Llama-3-8B on one GPU (BASELINE.json config 2); `tp70b.one_gpu_reference` is THIS workload on one GPU",
                       **({"shard_world": shard_world, "note": "shard shapes of a larger split timed on fewer ranks: a compute floor, not a scaling point"} if shard_world != world else {})},
            "allreduce": ar,
            "tp70b": tp70,
            "llama3_8b_tp": {"workload": f"Llama-3-8B W4A16 g128 bf16, decode M=1, tensor parallel over {world} GPUs (same sharding)",
                             "layers": L, "decode_tok_s": round(1e3 / ms_per_step * (L / 32), 2), "ms_per_step": round(ms_per_step, 4), "graph": graphed,
                             "launches_per_token": launches, "allreduces_per_step": 2 * L,
                             "hbm_gbs_per_gpu": round(gbs_rank, 1), "hbm_frac_per_gpu": round(gbs_rank / 8000.0, 4)},
            "roofline": {"bound": "hbm", "kernel": "awq::gemv_dma_kernel", "achieved": round(gbs70, 1),
                         "peak": 8000.0, "unit": "GB/s per GPU (incl. all-reduce time)", "frac": round(gbs70 / 8000.0, 4),
                         "traffic": None},
            "device": torch.cuda.get_device_name(dev)}
