"""Grouped per-expert GEMM (BASELINE.json config 5, Mixtral-8x7B shapes scaled down): host routing logic on CPU with
the oracle injected, and the HIP grouped kernel through the C ABI against the per-expert oracle (-m gpu)."""
import numpy as np
import pytest
import torch

from llm_awq_amd import moe as MOE
from llm_awq_amd.qmodule import WQLinear
from oracle import awq_oracle as O
from tests.helpers import check_forward, make_case, Gen, cuda_gen, assert_bits, oracle_grouped_linear, oracle_sparse_moe


def _experts(E, N, K, dtype, seed):
    mods, cases = [], []
    for e in range(E):
        c = make_case(N, K, dtype, seed=seed + e)
        m = WQLinear(4, 128, K, N, False, "cpu", dtype=dtype)
        m.qweight, m.scales, m.scaled_zeros = c["qweight"], c["scales"], c["scaled_zeros"]
        mods.append(m)
        cases.append(c)
    return mods, cases


def test_sort_by_expert():
    ids = torch.tensor([[2, 0], [1, 2], [0, 3], [2, 1]])
    order, off = MOE.sort_by_expert(ids, 5)
    assert off.tolist() == [0, 2, 4, 7, 8, 8] and off.dtype == torch.int32
    assert ids.reshape(-1)[order].tolist() == sorted(ids.reshape(-1).tolist())
    assert order.tolist() == [1, 4, 2, 7, 0, 3, 6, 5]  # stable


def test_sparse_moe_block_with_oracle_matmul():
    dtype, E, H, F, T = torch.bfloat16, 4, 128, 256, 9
    # (the product classes run the HIP kernels only; the oracle-backed subclasses of the routing glue live in tests/helpers.py)
    w1 = oracle_grouped_linear(_experts(E, F, H, dtype, 10)[0])
    w3 = oracle_grouped_linear(_experts(E, F, H, dtype, 20)[0])
    w2 = oracle_grouped_linear(_experts(E, H, F, dtype, 30)[0])
    assert w1.qweight.shape == (E, F // 4, H) and w1.scales.shape == (E, 8, F)
    blk = oracle_sparse_moe(w1, w3, w2, top_k=2)
    g = Gen(0)
    x = g.randn(T, H).to(dtype)
    logits = g.randn(T, E)
    logits[:, 3] = -1e9  # expert 3 gets no tokens (empty group)
    y = blk(x, logits)
    # dense reference: every token through its two experts, one at a time
    probs = torch.softmax(logits, -1)
    pw, ids = torch.topk(probs, 2, -1)
    pw = (pw / pw.sum(-1, keepdim=True)).to(dtype)
    ref = torch.zeros(T, H, dtype=dtype)
    for t in range(T):
        acc = []
        for j in range(2):
            e = int(ids[t, j])
            f = lambda mod, v: O.wqlinear_forward(v, mod.qweight[e], mod.scales[e], mod.scaled_zeros[e], None, 128)
            h = torch.nn.functional.silu(f(w1, x[t:t + 1])) * f(w3, x[t:t + 1])
            acc.append(f(w2, h) * pw[t, j])
        ref[t] = torch.stack(acc, 1).sum(1)
    assert torch.equal(y, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["v2", "cdna4"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("counts", [[5, 0, 130, 1], [0, 0, 0, 7], [128, 128, 1, 300], [0, 0, 0, 0]])
def test_gpu_grouped_gemm_vs_oracle(layout, dtype, counts):
    from llm_awq_amd import ops
    E, N, K = len(counts), 384, 512
    mods, cases = _experts(E, N, K, dtype, seed=sum(counts) + 7)
    grp = MOE.GroupedWQLinear(mods)
    T = sum(counts)
    g = Gen(T)
    x = g.randn(T, K).to(dtype)
    off = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32)
    qw = grp.qweight.cuda()
    if layout == "cdna4":
        qw = torch.stack([ops.repack_v2_to_cdna4(qw[e].contiguous()) for e in range(E)]).contiguous()
    y = ops.moe_gemm(x.cuda(), qw, grp.scales.cuda(), grp.scaled_zeros.cuda(), off.cuda(), layout=layout).cpu()
    assert y.shape == (T, N)
    for e in range(E):
        lo, hi = int(off[e]), int(off[e + 1])
        if hi > lo:
            check_forward(y[lo:hi], x[lo:hi], cases[e]["q"], cases[e]["scales"], cases[e]["scaled_zeros"], dtype)


@pytest.mark.gpu
def test_gpu_mixtral_block_shapes():
    """Mixtral-8x7B expert shapes (w1/w3 4096 -> 14336, w2 14336 -> 4096), 8 experts, top-2, 64 tokens: the grouped kernel
    against per-expert calls of the (separately verified) plain GEMM on the same buffers."""
    from llm_awq_amd import ops, synth
    E, H, F, T = 8, 4096, 14336, 64
    dev = "cuda"
    ws = [synth.random_wq(H, F, dtype=torch.bfloat16, seed=e, keep_q=False) for e in range(E)]
    qw = torch.stack([w["qweight"] for w in ws])
    s = torch.stack([w["scales"] for w in ws])
    z = torch.stack([w["scaled_zeros"] for w in ws])
    g = cuda_gen(1)
    x = torch.randn(T, H, device=dev, generator=g).to(torch.bfloat16)
    ids = torch.stack([torch.randperm(E, device=dev, generator=g)[:2] for _ in range(T)])
    order, off = MOE.sort_by_expert(ids, E)
    xs = x[order // 2].contiguous()
    y = ops.moe_gemm(xs, qw, s, z, off)
    offc = off.tolist()
    for e in range(E):
        lo, hi = offc[e], offc[e + 1]
        if hi > lo:
            ref = ops.gemm(xs[lo:hi].contiguous(), qw[e], s[e], z[e])
            assert_bits(ref, y[lo:hi], 0.02)


@pytest.mark.gpu
@pytest.mark.parametrize("counts", [[1, 0, 0, 1], [2, 2, 0, 0], [0, 0, 0, 3], [3, 1, 2, 2], [0, 8, 0, 0], [4, 3, 2, 4]])
def test_gpu_grouped_decode_and_module(counts):
    """decode batches: <= 8 sorted rows take the grouped GEMV (block = (expert, slab)); more rows the grouped GEMM; both
    through GroupedWQLinear.to_cdna4().forward and against the per-expert oracle."""
    dtype = torch.bfloat16
    E, N, K = len(counts), 384, 1280
    mods, cases = _experts(E, N, K, dtype, seed=sum(counts) + 3)
    grp = MOE.GroupedWQLinear(mods).cuda().to_cdna4()
    T = sum(counts)
    g = Gen(T + 1)
    x = g.randn(T, K).to(dtype)
    off = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32)
    y = grp(x.cuda(), off.cuda()).cpu()
    assert y.shape == (T, N)
    for e in range(E):
        lo, hi = int(off[e]), int(off[e + 1])
        if hi > lo:
            check_forward(y[lo:hi], x[lo:hi], cases[e]["q"], cases[e]["scales"], cases[e]["scaled_zeros"], dtype)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("counts", [[300, 0, 1, 255], [256, 256], [1, 700, 3, 40], [0, 0, 0, 256], [511, 2, 257, 0, 90], [300, 256, 10, 0, 270, 511, 63, 64],
                                    [40, 40, 40, 40, 40, 40, 40, 40]])
def test_gpu_grouped_prefill_v6_and_128(counts, dtype):
    """>= 256 sorted rows: the grouped kernel on the v6 tile (default; awq_gemm_v6.hip) -- tiles straddling expert boundaries, the shifted
    last tile, empty experts, segments shorter than a tile -- against the per-expert oracle and against the 128 x 128 grouped kernel (knob
    moe_v6=0), with which it agrees up to the association inside one 32-k MFMA."""
    from llm_awq_amd import ops
    E, N, K = len(counts), 400, 512
    mods, cases = _experts(E, N, K, dtype, seed=sum(counts) + 11)
    grp = MOE.GroupedWQLinear(mods).cuda().to_cdna4()
    T = sum(counts)
    g = Gen(T + 5)
    x = g.randn(T, K).to(dtype).cuda()
    off = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32).cuda()
    y6 = grp(x, off)
    try:
        ops._capi.tune(moe_v6=0)
        y_ref = grp(x, off)
    finally:
        ops._capi.tune(moe_v6=1)
    assert_bits(y6, y_ref, 0.01, what="v6 tile vs 128 x 128 grouped kernel")
    # round 5: partial row tiles of fewer than 64 rows are served by the grouped skinny kernel's tail pass behind the tile launch (knob moe_tail = 0:
    # every partial tile is a tile) -- the same products, another split of K over the waves for those rows
    try:
        ops._capi.tune(moe_tail=0)
        y_all_tiles = grp(x, off)
    finally:
        ops._capi.tune(moe_tail=64)
    assert_bits(y6, y_all_tiles, 0.01, what="tail pass vs every partial tile a tile")
    x = x.cpu()
    for y in (y6.cpu(), y_ref.cpu(), y_all_tiles.cpu()):
        for e in range(E):
            lo, hi = int(off[e]), int(off[e + 1])
            if hi > lo:
                check_forward(y[lo:hi], x[lo:hi], cases[e]["q"], cases[e]["scales"], cases[e]["scaled_zeros"], dtype)


@pytest.mark.gpu
def test_gpu_grouped_prefill_mixtral_shape_v6():
    """Mixtral-8x7B w1 (4096 -> 14336), 8 experts, 1024 tokens x top-2 = 2048 sorted rows with a ragged split: the v6 grouped tile
    against per-expert calls of the plain (dense, separately verified) prefill GEMM on the same buffers."""
    from llm_awq_amd import ops, synth
    E, H, F = 8, 4096, 14336
    counts = [300, 212, 256, 1, 511, 0, 257, 511]
    ws = [synth.random_wq(H, F, dtype=torch.bfloat16, seed=40 + e, keep_q=False) for e in range(E)]
    c4 = torch.stack([ops.repack_v2_to_cdna4(w["qweight"]) for w in ws])
    s = torch.stack([w["scales"] for w in ws])
    z = torch.stack([w["scaled_zeros"] for w in ws])
    szp = torch.stack([ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], H) for w in ws])
    T = sum(counts)
    x = torch.randn(T, H, device="cuda", generator=cuda_gen(3)).to(torch.bfloat16)
    off = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32).cuda()
    y = ops.moe_forward_cdna4(x, c4, s, z, szp, off)
    for e in range(E):
        lo, hi = int(off[e]), int(off[e + 1])
        if hi - lo >= 1:
            ref = ops.gemm_cdna4(x[lo:hi].contiguous(), c4[e], s[e], z[e], None, szp[e])
            assert_bits(ref, y[lo:hi], 0.01, what=f"expert {e}")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("counts", [[3, 2, 2, 2], [16, 0, 3, 40], [70, 1, 0, 9], [1, 1, 1, 6, 0, 0, 0, 0], [33, 31, 30, 34], [0, 0, 255, 0],
                                    [64, 64, 64, 63]])
def test_gpu_grouped_skinny(counts, dtype):
    """9 <= sorted rows <= 255 (batched MoE decode): the grouped skinny kernel -- experts without rows, experts with more rows
    than one pass holds, every column-block count -- against the per-expert oracle and the 128 x 128 grouped kernel."""
    from llm_awq_amd import ops
    E, N, K = len(counts), 400, 1280
    mods, cases = _experts(E, N, K, dtype, seed=sum(counts) + 17)
    grp = MOE.GroupedWQLinear(mods).cuda().to_cdna4()
    T = sum(counts)
    g = Gen(T + 9)
    x = g.randn(T, K).to(dtype).cuda()
    off = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32).cuda()
    y = grp(x, off)
    ops._capi.tune(moe_v4=0)
    try:
        y_ref = grp(x, off)
    finally:
        ops._capi.tune(moe_v4=1)
    assert_bits(y, y_ref, 0.03)  # different K split -> a few 1-ulp flips
    y, x = y.cpu(), x.cpu()
    for e in range(E):
        lo, hi = int(off[e]), int(off[e + 1])
        if hi > lo:
            check_forward(y[lo:hi], x[lo:hi], cases[e]["q"], cases[e]["scales"], cases[e]["scaled_zeros"], dtype)


def _fused_ref(x, c1, c3, dtype):
    """per-expert oracle of the fused half: T(T(silu(T(x W1^T))) * T(x W3^T)) (fused_mlp.py:79-82, every op rounded to T)"""
    a = O.wqlinear_forward(x, None, c1["scales"], c1["scaled_zeros"], None, 128, q_int=c1["q"])
    b = O.wqlinear_forward(x, None, c3["scales"], c3["scaled_zeros"], None, 128, q_int=c3["q"])
    return torch.nn.functional.silu(a) * b


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("counts", [[300, 0, 1, 255], [256, 256], [1, 700, 3, 40], [3, 2, 0, 2], [64, 64, 64, 63], [0, 5, 0, 0]])
def test_gpu_fused_gate_up_grouped(counts, dtype):
    """GroupedGateUp (awq_w4a16_moe_mlp_gate_up_cdna4): w1 / w3 interleaved 8 + 8 per expert, SiLU * mul in the grouped tile's epilogue from 256
    sorted rows on (ragged segments, empty experts, the shifted last tile), through scratch + the tail kernel below -- against the per-expert
    oracle SEQUENCE w1 -> w3 -> silu * mul, and against the unfused module path (two grouped launches + awq_silu_mul)."""
    E, F, K = len(counts), 272, 512
    m1, c1 = _experts(E, F, K, dtype, seed=sum(counts) + 101)
    m3, c3 = _experts(E, F, K, dtype, seed=sum(counts) + 202)
    gu = MOE.GroupedGateUp(m1, m3).cuda()
    T = sum(counts)
    x = Gen(T + 31).randn(T, K).to(dtype)
    off = torch.tensor([0] + list(np.cumsum(counts)), dtype=torch.int32)
    h = gu(x.cuda(), off.cuda()).cpu()
    assert h.shape == (T, F) and h.dtype == dtype
    w1 = MOE.GroupedWQLinear(m1).cuda().to_cdna4()
    w3 = MOE.GroupedWQLinear(m3).cuda().to_cdna4()
    blk = MOE.SparseMoeMLP(w1, w3, w1, 2)
    h_unfused = blk._h(x.cuda(), off.cuda()).cpu()
    assert_bits(h, h_unfused, 0.03, "fused grouped gate/up vs two grouped launches + tail")
    for e in range(E):
        lo, hi = int(off[e]), int(off[e + 1])
        if hi > lo:
            ref = _fused_ref(x[lo:hi], c1[e], c3[e], dtype)
            # gate / up are each within the forward bound of the oracle; the tail is computed from T-rounded values on both sides: a flipped
            # rounding of gate or up moves h by ~1 ulp, so bit-equality is statistical and the hard bound is a few ulps of T
            assert_bits(h[lo:hi], ref, 0.03, f"expert {e}", ulps=4, dtype=dtype)
            rel = ((h[lo:hi].float() - ref.float()).norm() / ref.float().norm()).item()
            assert rel < 1e-3, (e, rel)


@pytest.mark.gpu
def test_gpu_silu_mul_kernel():
    from llm_awq_amd import ops
    for dtype in (torch.bfloat16, torch.float16):
        g = cuda_gen(9)
        a = (torch.randn(37, 264, device="cuda", generator=g) * 3).to(dtype)
        b = (torch.randn(37, 264, device="cuda", generator=g) * 3).to(dtype)
        ref = (torch.nn.functional.silu(a.float()).to(dtype).float() * b.float()).to(dtype)
        # (the device silu is x / (1 + exp(-x)) with the hardware exp / rcp: the rounding of T(silu) flips on a few boundary cases)
        assert_bits(ops.silu_mul(a, b), ref, 0.01, "silu * mul", ulps=2, dtype=dtype)


@pytest.mark.gpu
def test_gpu_sparse_moe_block_fused_equals_unfused():
    """SparseMoeMLP.fused (one grouped launch for h) against SparseMoeMLP on separate w1 / w3 modules: same routing, same experts"""
    dtype, E, H, F, T = torch.bfloat16, 4, 256, 512, 300
    m1, _ = _experts(E, F, H, dtype, 310)
    m3, _ = _experts(E, F, H, dtype, 320)
    m2, _ = _experts(E, H, F, dtype, 330)
    w2 = MOE.GroupedWQLinear(m2).cuda().to_cdna4()
    fused = MOE.SparseMoeMLP.fused(m1, m3, w2, 2)
    fused.gate_up.cuda()
    plain = MOE.SparseMoeMLP(MOE.GroupedWQLinear(m1).cuda().to_cdna4(), MOE.GroupedWQLinear(m3).cuda().to_cdna4(), w2, 2)
    g = Gen(5)
    x = g.randn(T, H).to(dtype).cuda()
    logits = g.randn(T, E).cuda()
    y_f, y_p = fused(x, logits), plain(x, logits)
    assert y_f.shape == (T, H)
    assert ((y_f.float() - y_p.float()).norm() / y_p.float().norm()).item() < 2e-3


def test_stacked_modules_mark_their_layout_in_the_state_dict():
    """a stacked module that holds the cdna4 interleave says so in its state_dict (`qweight_layout`, as WQLinear's native checkpoints do); a fresh
    module that loads it takes the layout over, a v2 dict resets it"""
    dtype, E, F, K = torch.bfloat16, 2, 64, 128
    m1, _ = _experts(E, F, K, dtype, 5)
    m3, _ = _experts(E, F, K, dtype, 6)
    for make in (lambda: MOE.GroupedWQLinear(m1), lambda: MOE.GroupedGateUp(m1, m3)):
        a, b = make(), make()
        sd = a.state_dict()
        assert "qweight_layout" not in sd and set(sd) == {"qweight", "scales", "scaled_zeros"}
        a.layout = "cdna4"  # (what the first GPU forward / to_cdna4 leaves behind)
        sd = a.state_dict()
        assert int(sd["qweight_layout"]) == 1
        b.load_state_dict(sd)
        assert b.layout == "cdna4" and b.sz_cdna4 is None
        sd.pop("qweight_layout")
        b.load_state_dict(sd)
        assert b.layout == "v2"
