"""Shared test helpers: host-independent seeded synthetic cases built by the ORACLE (CPU) and error bounds.

Inputs.  Every CPU-side random draw of the suite comes from numpy's PCG64 (`Gen`), not from torch's CPU generator: torch's
CPU `randn` runs vectorised math whose last bits depend on the host's SIMD level, so two boxes would test different data.
`AWQ_TEST_SEED` (default 0) is mixed into every seed, so the whole suite can be re-run on other data.

Bounds.  Hard criteria are always elementwise / norm-wise error bounds.  "How many elements are bit-identical" is a
STATISTICAL criterion: the expected number of flipped roundings is modelled (or stated by the caller) and the observed count
must stay inside a Poisson tail (lambda + 6 sqrt(lambda) + 3: false-failure probability below 1e-6 per assertion)."""
import json
import math
import os

import numpy as np
import torch

from oracle import awq_oracle as O

MANT = {torch.float16: 10, torch.bfloat16: 7}
SEED0 = int(os.environ.get("AWQ_TEST_SEED", "0"))
_STATS = os.environ.get("AWQ_TEST_STATS")  # path of a jsonl file: observed flip counts next to what was allowed


class Gen:
    """numpy PCG64 stream -> torch tensors (bit-identical on every host)."""

    def __init__(self, seed):
        self.g = np.random.Generator(np.random.PCG64([SEED0, int(seed)]))

    def randn(self, *shape):
        return torch.from_numpy(self.g.standard_normal(size=shape, dtype=np.float32))

    def rand(self, *shape):
        return torch.from_numpy(self.g.random(size=shape, dtype=np.float32))

    def randint(self, lo, hi, shape):
        return torch.from_numpy(self.g.integers(lo, hi, size=tuple(shape), dtype=np.int64))


def cuda_gen(seed):
    """device-side Philox stream (the algorithm runs on the GPU: the same on every MI355X box)."""
    return torch.Generator(device="cuda").manual_seed(SEED0 * 1000003 + int(seed))


_BIG = 4 << 20      # from this many weights on, the ORACLE's quantise + pack of a case costs seconds on the test box's host
_big_cases = {}     # (N, K, dtype, variant) -> the oracle's buffers of that matrix (a handful of full model shapes; ~1 GB at most)


def make_case(N, K, dtype, seed=0, bias=False, M=1, x_scale=1.0):
    """oracle-built case.  Matrices below _BIG weights are drawn from `seed` as always.  The few full-size shapes (Llama layer sizes, tens of
    millions of weights) are quantised and packed ONCE per (shape, dtype, seed parity) and shared by the tests that ask for them -- the
    activations and the bias are still drawn from `seed` -- which takes minutes of host-side oracle time off the GPU suite."""
    if N * K >= _BIG:
        key = (N, K, dtype, int(seed) & 1)  # two distinct matrices per shape: the fused gate / up tests ask for seed and seed + 1
        if key not in _big_cases:
            if len(_big_cases) >= 8:
                _big_cases.pop(next(iter(_big_cases)))
            gw = Gen(1000003 * (int(seed) & 1) + N * 31 + K)
            dw = O.quantize_linear(gw.randn(N, K) * 0.02, dtype=dtype, n_bit=4, group_size=128)
            dw["q"] = dw["intweight"].numpy().astype(np.uint8)
            _big_cases[key] = dw
        d = dict(_big_cases[key])
        g = Gen(seed)
    else:
        g = Gen(seed)
        w = g.randn(N, K) * 0.02
        d = O.quantize_linear(w, dtype=dtype, n_bit=4, group_size=128)
        d["q"] = d["intweight"].numpy().astype(np.uint8)
    d["x"] = (g.randn(M, K) * x_scale).to(dtype)
    d["bias"] = (g.randn(N) * 0.02).to(dtype) if bias else None
    return d


MIN_EXP = {torch.float16: -14, torch.bfloat16: -126}  # exponent of the smallest normal number: below it the spacing is constant (subnormals)


def ulp(v: torch.Tensor, dtype) -> torch.Tensor:
    e = torch.floor(torch.log2(v.abs().double().clamp(min=1e-300))).clamp(min=MIN_EXP[dtype])
    return torch.pow(2.0, e - MANT[dtype])


def flips_allowed(lam: float) -> float:
    return lam + 6.0 * math.sqrt(lam) + 3.0


def _record(what, count, lam, numel):
    if _STATS:
        with open(_STATS, "a") as f:
            f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", ""), "what": what, "count": int(count),
                                "lambda": float(lam), "allowed": flips_allowed(lam), "numel": int(numel)}) + "\n")


def assert_bits(a: torch.Tensor, b: torch.Tensor, rate: float, what: str = "", ulps: float = None, dtype=None):
    """`a` and `b` state the same math with different rounding points / reduction orders: at most `rate` of the elements are
    EXPECTED to differ (the caller's model); the observed count must stay inside the Poisson tail of rate * numel.  With `ulps`
    every differing pair must also be within that many ulps of T of each other (plus 1e-4 of the rms value for results that
    cancel to ~0, where the fp32 accumulation error exceeds an ulp of the tiny result)."""
    assert a.shape == b.shape, (a.shape, b.shape)
    n = a.numel()
    count = int((a != b).sum().item())
    lam = rate * n
    _record(what or "bits", count, lam, n)
    assert count <= flips_allowed(lam), f"{what}: {count} of {n} elements differ ({100.0 * count / n:.3f} %), expected <= {100 * rate:.2f} %"
    if ulps is not None:
        dtype = dtype or a.dtype
        ad, bd = a.double(), b.double()
        tol = ulps * 1.001 * ulp(torch.maximum(ad.abs(), bd.abs()), dtype) + 1e-4 * bd.pow(2).mean().sqrt()
        worst = ((ad - bd).abs() / tol).max().item()
        assert worst <= 1.0, f"{what}: elementwise distance {worst:.2f} x ({ulps} ulp + 1e-4 rms)"


_w_memo = []  # [(q, scales, scaled_zeros, W float64)]: tests loop over row counts / bias on ONE case; its dequantised matrix is built once


def _dequant_f64(q, scales, scaled_zeros):
    for (q0, s0, z0, W) in _w_memo:
        if q0 is q and s0 is scales and z0 is scaled_zeros:
            return W
    W = O.dequant_weight(q, scales, scaled_zeros, 128).double()
    _w_memo.append((q, scales, scaled_zeros, W))
    if len(_w_memo) > 2:
        _w_memo.pop(0)
    return W


def check_forward(y_gpu: torch.Tensor, x, q, scales, scaled_zeros, dtype, bias=None, x_unc=None):
    """y_gpu (T, on cpu) against the oracle:
    (1) HARD, elementwise: within half an ulp of T around the float64 contraction of the T-rounded weights, plus the fp32
        accumulation slack 2e-6 * sum|x||w| (plus one more rounding when a bias add follows);
    (2) HARD, norm-wise <= 1e-3 against the oracle's T-rounded output (BASELINE.json);
    (3) STATISTICAL: the fp32-accumulate oracle agrees bit for bit except where the two fp32 accumulation orders land on
        different sides of a rounding boundary of T.  Model: the two fp32 sums differ by ~ eps32 * sqrt(K)/4 * sqrt(sum (x w)^2)
        (random-walk bound of a sequential fp32 accumulation; blocked / tree orders do better), so output i flips with
        probability p_i = min(1, that / ulp_T(y_i)); the number of flips must stay inside the Poisson tail of sum p_i.

    `x_unc` [M, K] (fused-norm callers): absolute uncertainty of the activations the kernel really multiplied -- elements of
    the normalised x whose rounding to T depends on the last fp32 bits of rstd.  It widens (1) by x_unc @ |W|^T and (3) by the
    flips that slack can cause; (2) is unchanged."""
    return check_forward_rows(y_gpu, forward_oracle(x, q, scales, scaled_zeros, dtype, bias=bias, x_unc=x_unc))


def forward_oracle(x, q, scales, scaled_zeros, dtype, bias=None, x_unc=None):
    """the oracle side of check_forward for ALL rows of x, computed once: output rows are independent, so a test that runs several row counts m on the
    first m rows of one x checks each result against the first m rows of this (check_forward_rows)"""
    Wd = _dequant_f64(q, scales, scaled_zeros)
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    y64 = x2.double() @ Wd.t()
    S = x2.double().abs() @ Wd.abs().t()
    Q = ((x2.double() ** 2) @ (Wd ** 2).t()).sqrt()
    u = ulp(y64, dtype)
    bound = 0.501 * u + 2e-6 * S + 1e-30
    acc = (2.0 ** -24) * (math.sqrt(K) / 4.0 + 1.0) * Q
    if x_unc is not None:
        extra = x_unc.reshape(-1, K).double() @ Wd.abs().t()
        bound = bound + extra
        acc = acc + extra
    if bias is not None:
        y64 = y64 + bias.double()
        u = torch.minimum(u, ulp(y64, dtype))
        bound = bound + 0.501 * ulp(y64, dtype) + ulp(y64, dtype) * 0.5
    # BASELINE.json: "<= 1e-3 rel on the fp16/bf16 matmul result" -- norm-wise against the oracle's T-rounded
    # output (the final rounding to bf16 alone is ~1.1e-3 rms per element, so it must be on both sides)
    y_or = O.wqlinear_forward(x, None, scales, scaled_zeros, bias, 128, q_int=q).reshape(y64.shape).double()
    return dict(y64=y64, bound=bound, acc=acc, u=u, y_or=y_or)


def check_forward_rows(y_gpu: torch.Tensor, pre, rows=None):
    """y_gpu = the kernel's result for the first `rows` rows of the x `pre` (forward_oracle) was built from (default: all)"""
    n = pre["y64"].shape[0] if rows is None else rows
    y64, bound, acc, u, y_or = (pre[k][:n] for k in ("y64", "bound", "acc", "u", "y_or"))
    yg = y_gpu.reshape(y64.shape).double()
    err = (yg - y64).abs()
    worst = (err / bound).max().item()
    assert worst <= 1.0, f"elementwise bound violated: worst err/bound = {worst}"
    rel = ((yg - y_or).norm() / y_or.norm()).item()
    assert rel <= 1e-3, rel
    count = int((y_or != yg).sum().item())
    lam = torch.clamp(acc / u, max=1.0).sum().item()
    _record("check_forward", count, lam, y_or.numel())
    assert count <= flips_allowed(lam), (f"{count} of {y_or.numel()} elements differ from the fp32-accumulate oracle; the "
                                         f"accumulation-order model expects {lam:.1f} (allowed {flips_allowed(lam):.1f})")
    return worst, rel, count / y_or.numel()


def rmsnorm_uncertainty(x: torch.Tensor, gamma: torch.Tensor, eps: float) -> torch.Tensor:
    """[M, K] uncertainty of T((x * rstd) * gamma) under a relative perturbation of 2^-21 of the fp32 product (one ulp of the
    hardware rsqrt, the order of the fp32 sum of squares, two fp32 multiplies): where the rounding to T is not decided, one
    ulp of T; elsewhere 0.  (layernorm.cu:48-60: the reference kernel's own rsqrtf / reduction order are just as free.)"""
    xd = x.double()
    rstd = torch.rsqrt((xd * xd).mean(-1, keepdim=True) + eps)
    v = xd * rstd * gamma.double()
    d = 2.0 ** -21
    lo, hi = (v * (1 - d)).to(x.dtype), (v * (1 + d)).to(x.dtype)
    return torch.where(lo != hi, ulp(v, x.dtype), torch.zeros_like(v))


# ---------------- oracle-backed subclasses for the CPU tests of the host logic ----------------
# The product classes (llm_awq_amd.parallel.TPWQLinear, llm_awq_amd.moe.GroupedWQLinear / SparseMoeMLP) run the HIP kernels and nothing else.
# What the CPU suite tests about them -- shard slicing, the fp32 sum over ranks, bias placement, routing / sorting glue -- needs SOME arithmetic
# behind the hooks; it comes from the oracle, here, in test code.
def oracle_tp_linear(full, mode, rounded_partials=False, **kw):
    from llm_awq_amd.parallel import TPWQLinear

    class OracleTPWQLinear(TPWQLinear):
        def _shard_product(self, x):
            sh = self.shard
            return O.wqlinear_forward(x, sh.qweight, sh.scales, sh.scaled_zeros, None, 128)

        def _shard_partial(self, x):
            sh = self.shard
            if rounded_partials:  # the pre-round-4 numerics (every rank's partial rounded to T before the sum): kept for the comparison
                return O.wqlinear_forward(x, sh.qweight, sh.scales, sh.scaled_zeros, None, 128).float()
            return O.wqlinear_partial_f32(x, sh.qweight, sh.scales, sh.scaled_zeros, 128)

        def _round_bias(self, y32, dtype):
            y = y32.to(dtype)
            return y + self.bias if self.bias is not None else y

        def _round_rows(self, y32_rows, dtype, bias):
            y = y32_rows.to(dtype)
            return y + bias if bias is not None else y

    return OracleTPWQLinear(full, mode, **kw)


def oracle_grouped_linear(experts):
    from llm_awq_amd.moe import GroupedWQLinear

    class OracleGroupedWQLinear(GroupedWQLinear):
        def forward(self, x_sorted, expert_offsets):
            off = expert_offsets.tolist()
            outs = [O.wqlinear_forward(x_sorted[off[e]: off[e + 1]], self.qweight[e], self.scales[e], self.scaled_zeros[e], None, 128)
                    for e in range(self.num_experts) if off[e + 1] > off[e]]
            return torch.cat(outs) if outs else x_sorted.new_zeros(0, self.out_features)

    return OracleGroupedWQLinear(experts)


def oracle_sparse_moe(w1, w3, w2, top_k=2):
    from llm_awq_amd.moe import SparseMoeMLP

    class OracleSparseMoeMLP(SparseMoeMLP):
        @staticmethod
        def _silu_mul(a, b):
            return torch.nn.functional.silu(a) * b

    return OracleSparseMoeMLP(w1, w3, w2, top_k)


# ---------------- compositions of T-rounded ops (fused gate/up tail) ----------------
def _nbrs(t: torch.Tensor, slack=None):
    """(below, t, above) as T tensors: one ulp of T away, or `slack` (absolute, >= 0, broadcastable) where that is more -- float64 arithmetic,
    rounded to T (exact for normal values one ulp away)"""
    d = t.double()
    u = ulp(d, t.dtype)
    if slack is not None:
        u = torch.maximum(u, slack.double())
    return (d - u).to(t.dtype), t, (d + u).to(t.dtype)


def acc_slack(x: torch.Tensor, w_row_norm: torch.Tensor) -> torch.Tensor:
    """[M, N] absolute distance two fp32 accumulations of the same products may lie apart before the rounding to T: 2e-6 * sum |x||w| (the slack
    check_forward's elementwise bound (1) grants), with sum |x||w| <= ||x||_2 ||w||_2.  It matters where a product cancels to ~0: there it exceeds
    an ulp of T of the tiny result, and the kernel's T(gate') is more than one neighbour away from the oracle's."""
    return 2e-6 * x.reshape(-1, x.shape[-1]).double().norm(dim=1, keepdim=True) * w_row_norm.double().reshape(1, -1)


def weight_row_norms(case) -> torch.Tensor:
    """||w_n||_2 of the dequantised (T-rounded) weight rows of an oracle-built case (memoised on the shared big-case entry)"""
    if "_wnorm" not in case:
        wn = O.dequant_weight(case["q"], case["scales"], case["scaled_zeros"], 128).float().norm(dim=1)
        case["_wnorm"] = wn
        for d in _big_cases.values():  # (make_case hands out copies of the shared entry: remember it there too)
            if d.get("q") is case["q"]:
                d["_wnorm"] = wn
    return case["_wnorm"]


def record_rel(what: str, rel: float, allowed: float):
    if _STATS:
        with open(_STATS, "a") as f:
            f.write(json.dumps({"test": os.environ.get("PYTEST_CURRENT_TEST", ""), "what": what, "rel": float(rel), "allowed": float(allowed)}) + "\n")


def check_fused_tail(y: torch.Tensor, gate: torch.Tensor, up: torch.Tensor, rel_max: float, what: str = "fused tail", slack_g=None, slack_u=None):
    """y = the kernel's T(T(silu(T(gate'))) * T(up')) from ITS fp32 sums gate' / up'; `gate` / `up` = the oracle's T-rounded products
    (fused_mlp.py:79-82: every op rounded to T).
    HARD, elementwise -- the model that replaces a flat norm-wise tolerance: the kernel's fp32 sums lie within `slack_g` / `slack_u` (acc_slack:
    2e-6 sum |x||w|, the accumulation-order slack of check_forward (1)) of the oracle's, so its T(gate'), T(up') lie between the oracle's values
    moved by max(one ulp of T, that slack) either way; its fp32 silu (hardware exp2 / rcp) is within a few ulp of fp32 of torch's, so T(silu) is
    torch's value or a neighbour one ulp of T away; the multiply and its rounding are deterministic.  silu is monotone on either side of its minimum
    (-0.27846 at -1.27846) and the product is monotone in each factor, so y must lie in the hull of the ORACLE tail over
    {gate -, gate, gate +, the minimum if the gate interval holds it} x {silu -, silu, silu +} x {up -, up, up +} -- no slack on top.
    Norm-wise: against the oracle's tail, recorded (AWQ_TEST_STATS) and held to `rel_max` (BASELINE.json's 1e-3; measured <= 3.7e-4, profiles/r05_test_stats.txt)."""
    dtype = y.dtype
    lo = hi = None
    centre = None
    g_lo, g_mid, g_hi = _nbrs(gate, slack_g)
    g_min = torch.clamp(torch.full_like(gate, -1.2784645), min=g_lo, max=g_hi)  # an endpoint again when the interval does not hold the minimum
    for gi, gg in enumerate((g_lo, g_mid, g_hi, g_min)):
        for si, sg in enumerate(_nbrs(torch.nn.functional.silu(gg))):  # silu: T in -> fp32 inside -> one rounding to T
            for ui, uu in enumerate(_nbrs(up, slack_u)):
                v = (sg * uu).float()            # (T values are exact in fp32)
                lo = v if lo is None else torch.minimum(lo, v)
                hi = v if hi is None else torch.maximum(hi, v)
                if gi == 1 and si == 1 and ui == 1:
                    centre = v
    yd = y.float()
    bad = (yd < lo) | (yd > hi)
    nbad = int(bad.sum().item())
    if nbad:
        k = int(torch.nonzero(bad.flatten())[0])
        detail = f"first: y={yd.flatten()[k].item()!r} hull=[{lo.flatten()[k].item()!r}, {hi.flatten()[k].item()!r}] gate={gate.flatten()[k].item()!r} up={up.flatten()[k].item()!r}"
    assert nbad == 0, f"{what}: {nbad} of {y.numel()} outputs outside the hull of the oracle tail over the one-ulp neighbours of (gate, silu, up); {detail}"
    rel = ((yd.double() - centre.double()).norm() / centre.double().norm()).item()
    record_rel(what, rel, rel_max)
    assert rel <= rel_max, f"{what}: norm-wise {rel:.3e} > {rel_max:.3e}"
    return rel
