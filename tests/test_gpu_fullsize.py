"""-m gpu: BASELINE.json's full Llama-3-8B layer shapes.  The oracle is too slow here, so use
size-independent checks: (a) the kernels against a plain torch fp32 matmul on the weights produced by
the (separately bit-exact-verified) dequant kernel; (b) GEMV and GEMM agree with each other on the same
rows; (c) linearity in x for power-of-two scalings (exact in floating point)."""
import pytest
import torch

from tests.helpers import acc_slack, cuda_gen, assert_bits, check_fused_tail, record_rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from llm_awq_amd import ops, synth
    ops._capi.lib()
    return ops, synth


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("K,N", [(4096, 6144), (4096, 4096), (4096, 14336), (14336, 4096), (11008, 4096)])
def test_fullsize_vs_torch_fp32(env, dtype, K, N):
    ops, synth = env
    w = synth.random_wq(K, N, dtype=dtype, seed=K + N, keep_q=False)
    W = ops.dequant_v2(w["qweight"], w["scales"], w["scaled_zeros"]).float()
    g = cuda_gen(7)
    for M, fn in [(1, ops.gemv), (7, ops.gemv), (16, ops.gemv), (64, ops.gemm), (300, ops.gemm), (2048, ops.gemm)]:
        x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
        y = fn(x, w["qweight"], w["scales"], w["scaled_zeros"]).float()
        ref = (x.float() @ W.t()).to(dtype).float()  # fp32 accumulate, ONE rounding to T: the rounding is on both sides
        rel = ((y - ref).norm() / ref.norm()).item()
        assert rel <= 1e-3, (M, rel)                  # BASELINE.json's tolerance
        # and almost all elements are identical
        assert_bits(ref, y, 0.03, what=str(M))


def test_gemv_gemm_agree_and_scaling(env):
    ops, synth = env
    K, N = 4096, 14336
    w = synth.random_wq(K, N, dtype=torch.bfloat16, seed=1, keep_q=False)
    x = torch.randn(64, K, device="cuda").bfloat16()
    yg = ops.gemm(x, w["qweight"], w["scales"], w["scaled_zeros"])
    yv = ops.gemv(x[:7].contiguous(), w["qweight"], w["scales"], w["scaled_zeros"])
    assert_bits(yg[:7], yv, 0.02)
    y2 = ops.gemv((x[:7] * 4).contiguous(), w["qweight"], w["scales"], w["scaled_zeros"])
    assert torch.equal(y2, yv * 4)  # power-of-two scaling commutes with every rounding


# ---------------- the product path of bench.py: cdna4 interleave, decode fast path, fused MLP, GEMM v3 ----------------
@pytest.mark.parametrize("K,N", [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096), (11008, 4096), (8192, 10240)])
def test_fullsize_cdna4_vs_torch_fp32(env, K, N):
    """BASELINE.json shapes (Llama-3-8B, the stacked gate/up pair, Llama-2-7B's 11008, Llama-3-70B's 8192 x 10240) on the
    layout the repacker emits: dequant kernels of both layouts agree bit for bit, and every M bucket of forward_cdna4
    (decode fast path <= 8, old GEMV <= 16, 128x128 GEMM, GEMM v3 with both tile widths) matches a torch fp32 matmul."""
    ops, synth = env
    dtype = torch.bfloat16
    w = synth.random_wq(K, N, dtype=dtype, seed=K + N + 1, keep_q=False)
    c4 = ops.repack_v2_to_cdna4(w["qweight"])
    szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
    W2 = ops.dequant_v2(w["qweight"], w["scales"], w["scaled_zeros"])
    W4 = ops.dequant_cdna4(c4, w["scales"], w["scaled_zeros"])
    assert torch.equal(W2, W4)
    W = W4.float()
    g = cuda_gen(11)
    bias = (torch.randn(N, device="cuda", generator=g) * 0.02).to(dtype)
    for M in (1, 3, 8, 13, 64, 256, 300, 2048):
        x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
        for b in (None, bias):
            y = ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], b, szp)
            ref = (x.float() @ W.t()).to(dtype)
            if b is not None:
                ref = ref + b
            rel = ((y.float() - ref.float()).norm() / ref.float().norm()).item()
            assert rel < 1e-3, (M, rel)
            assert_bits(ref, y, 0.03, what=str(M))


@pytest.mark.parametrize("variant", [4, 5])  # GEMM v3: force 256 x 256 / 256 x 128 tiles
def test_gemm_v3_tile_widths_identical(env, variant):
    ops, synth = env
    K, N = 4096, 6144
    w = synth.random_wq(K, N, dtype=torch.bfloat16, seed=3, keep_q=False)
    c4 = ops.repack_v2_to_cdna4(w["qweight"])
    szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
    for M in (256, 257, 1000, 2048):
        x = torch.randn(M, K, device="cuda").bfloat16()
        ops._capi.tune(gemm_variant=1, gemm_splitk=0)  # (split-K re-associates the sum: tests/test_gpu_splitk.py)
        ref = ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], None, szp)
        ops._capi.tune(gemm_variant=variant)
        try:
            y = ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], None, szp)
        finally:
            ops._capi.tune(gemm_variant=0, gemm_splitk=1)
        assert torch.equal(y, ref), M  # same K order, same numerics: bit-identical to the 128 x 128 kernel


def test_fused_mlp_fullsize(env):
    """Llama-3-8B gate/up (2 x 14336 x 4096) in one launch == two GEMVs + F.silu + mul on the same buffers."""
    ops, synth = env
    K, F = 4096, 14336
    w = synth.random_wq(K, 2 * F, dtype=torch.bfloat16, seed=5, keep_q=False)
    c4 = ops.repack_v2_to_cdna4(w["qweight"])
    szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
    for M in (1, 4, 7):
        x = torch.randn(M, K, device="cuda").bfloat16()
        y = ops.mlp_gate_up_cdna4(x, c4, szp)
        full = ops.gemm_cdna4(x, c4, w["scales"], w["scaled_zeros"], None, szp)
        ref = torch.nn.functional.silu(full[:, :F]) * full[:, F:]
        assert y.shape == (M, F)
        assert_bits(ref, y, 0.02)
        # the fused launch's T(gate'), T(up') against the plain GEMM's: the same values or one-ulp neighbours -> the tail's elementwise hull
        wn = ops.dequant_cdna4(c4, w["scales"], w["scaled_zeros"]).float().norm(dim=1).cpu()
        check_fused_tail(y.cpu(), full[:, :F].cpu(), full[:, F:].cpu(), 1e-3, what=f"stacked gate/up vs GEMM M={M}", slack_g=acc_slack(x.cpu(), wn[:F]),
                         slack_u=acc_slack(x.cpu(), wn[F:]))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_llama3_70b_tp8_shards_sum_to_the_unsharded_layer(env, dtype):
    """BASELINE.json config 4 (Llama-3-70B, TP = 8): the N-sharded stacked gate/up pair (8192 -> 2 x 28672, 2 x 3584 rows per rank) at
    its real shard size -- every rank's shard through the cdna4 kernels; the concatenation reproduces the unsharded layer."""
    ops, synth = env
    from llm_awq_amd import parallel as P
    world = 8
    g = cuda_gen(5)
    # (row parallel -- down_proj, one K slice per rank, fp32 partials rounded once -- lives in tests/test_gpu_tp_partial.py)
    # ---- column parallel: the stacked gate/up pair, matching gate and up rows on every rank, fused SiLU*mul epilogue ----
    K, F = 8192, 28672
    w = synth.random_wq(K, 2 * F, dtype=dtype, seed=9, keep_q=False)
    x = torch.randn(2, K, device="cuda", generator=g).to(dtype)
    c4 = ops.repack_v2_to_cdna4(w["qweight"])
    szp = ops.pack_sz_cdna4(w["scales"], w["scaled_zeros"], K)
    full = ops.mlp_gate_up_cdna4(x, c4, szp)
    parts = []
    for r in range(world):
        qw, s, z, bounds = P.shard_stacked_column_parallel(w["qweight"], w["scales"], w["scaled_zeros"], world, r, parts=2)
        assert qw.shape[0] * 4 == 2 * 3584
        parts.append(ops.mlp_gate_up_cdna4(x, ops.repack_v2_to_cdna4(qw), ops.pack_sz_cdna4(s, z, K)))
    got = torch.cat(parts, 1)
    # a shard splits K over a different number of waves than the full matrix does: same products, another fp32 summation order
    assert_bits(got, full, 0.03)
    rel = ((got.float() - full.float()).norm() / full.float().norm()).item()
    record_rel("70B gate/up shards vs full", rel, 1e-3)
    assert rel <= 1e-3, rel  # (measured 1.8e-5: profiles/r05_test_stats.txt)
