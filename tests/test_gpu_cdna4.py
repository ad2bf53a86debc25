"""-m gpu: the "cdna4" interleave (this repository's MI355X-native int4 layout) and the kernels that
consume it, through the C ABI, against the oracle.  Index work is bit exact; the matrix-core dequant
must reproduce round_bf16(q*s+sz) bit for bit; matmul within the bounds of tests/helpers.py."""
import numpy as np
import pytest
import torch

from oracle import awq_oracle as O
from tests.helpers import check_forward, make_case, Gen, assert_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from llm_awq_amd import ops as _ops
    _ops._capi.lib()
    return _ops


@pytest.mark.parametrize("N,K", [(16, 128), (32, 256), (64, 768), (768, 3072), (1024, 4096)])
def test_repack_bit_exact_and_roundtrip(ops, N, K):
    rng = np.random.default_rng(N + K)
    q = rng.integers(0, 16, size=(N, K)).astype(np.uint8)
    v2 = torch.from_numpy(O.pack_v2(q)).cuda()
    c4 = ops.repack_v2_to_cdna4(v2)
    assert (c4.cpu().numpy() == O.pack_cdna4(q)).all()
    assert (ops.unpack_cdna4(c4).cpu().numpy() == q).all()
    assert torch.equal(ops.repack_cdna4_to_v2(c4), v2)


def test_structured_tile_is_transpose_detecting(ops):
    N, K = 32, 256
    q = ((np.arange(N)[:, None] * 5 + np.arange(K)[None, :] * 3) % 16).astype(np.uint8)
    c4 = ops.repack_v2_to_cdna4(torch.from_numpy(O.pack_v2(q)).cuda())
    assert (c4.cpu().numpy() == O.pack_cdna4(q)).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,K", [(16, 128), (64, 768), (256, 1280), (512, 4096)])
def test_matrix_core_dequant_bit_exact(ops, dtype, N, K):
    c = make_case(N, K, dtype, seed=N + K)
    W = O.dequant_weight(c["q"], c["scales"], c["scaled_zeros"], 128)
    c4 = ops.repack_v2_to_cdna4(c["qweight"].cuda())
    got = ops.dequant_cdna4(c4, c["scales"].cuda(), c["scaled_zeros"].cuda()).cpu()
    assert torch.equal(got.view(torch.int16), W.view(torch.int16))


@pytest.mark.parametrize("dtype,emin,emax", [(torch.bfloat16, -30, 8), (torch.float16, -12, 3)])
def test_matrix_core_dequant_adversarial_scales(ops, dtype, emin, emax):
    """scales over the dtype's exponent range, every zero point 0..15: (offset + q) * s + (sz - offset * s) stays exact in
    fp32 (bf16: offset 128, 8 x 8 significant bits; fp16: offset 1024, 11 x 11), so the single rounding is the reference's"""
    g = Gen(9)
    N, K = 64, 512
    q = g.randint(0, 16, (N, K)).numpy().astype(np.uint8)
    scales = torch.zeros(8, N, dtype=dtype)
    scales[:4] = ((g.rand(4, N) * 2 + 0.5) * torch.pow(2.0, g.randint(emin, emax, (4, N)).float())).to(dtype)
    zeros = g.randint(0, 16, (4, N))
    sz = torch.zeros(8, N, dtype=dtype)
    sz[:4] = -(scales[:4].float() * zeros.float()).to(dtype)
    W = O.dequant_weight(q, scales, sz, 128)
    c4 = ops.repack_v2_to_cdna4(torch.from_numpy(O.pack_v2(q)).cuda())
    got = ops.dequant_cdna4(c4, scales.cuda(), sz.cuda()).cpu()
    assert torch.equal(got.view(torch.int16), W.view(torch.int16))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 2, 4, 5, 8, 13, 16])
@pytest.mark.parametrize("N,K", [(768, 768), (3072, 768), (768, 3072), (256, 4096), (1040, 1280), (64, 11008)])
def test_gemv_cdna4_vs_oracle(ops, dtype, M, N, K):
    c = make_case(N, K, dtype, seed=M * 131 + N + K, M=M)
    c4 = ops.repack_v2_to_cdna4(c["qweight"].cuda())
    y = ops.gemv_cdna4(c["x"].cuda(), c4, c["scales"].cuda(), c["scaled_zeros"].cuda()).cpu()
    check_forward(y, c["x"], c["q"], c["scales"], c["scaled_zeros"], dtype)
    # same through the packed {scale | zero} array (one dword load per step)
    szp = ops.pack_sz_cdna4(c["scales"].cuda(), c["scaled_zeros"].cuda(), K)
    y2 = ops.gemv_cdna4(c["x"].cuda(), c4, c["scales"].cuda(), c["scaled_zeros"].cuda(), szp).cpu()
    check_forward(y2, c["x"], c["q"], c["scales"], c["scaled_zeros"], dtype)
    assert_bits(y2, y, 0.03)  # different split-K order, same rounding almost everywhere


def test_pack_sz_cdna4(ops):
    c = make_case(64, 768, torch.bfloat16, seed=1)
    szp = ops.pack_sz_cdna4(c["scales"].cuda(), c["scaled_zeros"].cuda(), 768).cpu().numpy().view(np.uint32)
    s = c["scales"].view(torch.int16).numpy().view(np.uint16).astype(np.uint32)
    z = c["scaled_zeros"].view(torch.int16).numpy().view(np.uint16).astype(np.uint32)
    for nb in range(4):
        for kg in range(6):
            assert (szp[nb, kg] == (s[kg, nb * 16:(nb + 1) * 16] | (z[kg, nb * 16:(nb + 1) * 16] << 16))).all()


@pytest.mark.parametrize("knobs", [dict(gemv_waves=4, gemv_pf=2), dict(gemv_waves=8, gemv_pf=4), dict(gemv_waves=16, gemv_pf=8),
                                   dict(gemv_waves=4, gemv_pf=8, gemv_x_budget_kib=8)])
def test_gemv_knobs_do_not_change_results(ops, knobs):
    """every (waves, prefetch depth, x-segment) configuration: both layouts, M = 1 and 7, odd step counts."""
    try:
        for (N, K) in [(64, 11008), (128, 4096), (48, 1280)]:
            for M in (1, 7):
                c = make_case(N, K, torch.bfloat16, seed=N + M, M=M)
                c4 = ops.repack_v2_to_cdna4(c["qweight"].cuda())
                ops._capi.tune(gemv_waves=0, gemv_pf=0, gemv_x_budget_kib=64)
                ref4 = ops.gemv_cdna4(c["x"].cuda(), c4, c["scales"].cuda(), c["scaled_zeros"].cuda())
                ops._capi.tune(**knobs)
                y4 = ops.gemv_cdna4(c["x"].cuda(), c4, c["scales"].cuda(), c["scaled_zeros"].cuda())
                y2 = ops.gemv(c["x"].cuda(), c["qweight"].cuda(), c["scales"].cuda(), c["scaled_zeros"].cuda())
                check_forward(y4.cpu(), c["x"], c["q"], c["scales"], c["scaled_zeros"], torch.bfloat16)
                check_forward(y2.cpu(), c["x"], c["q"], c["scales"], c["scaled_zeros"], torch.bfloat16)
                assert_bits(y4, ref4, 0.02)
    finally:
        ops._capi.tune(gemv_waves=0, gemv_pf=0, gemv_x_budget_kib=64)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("variant,M", [(0, m) for m in (17, 64, 128, 200, 512, 777)] + [(v, m) for v in (1, 2) for m in (17, 128, 777)])
@pytest.mark.parametrize("N,K", [(768, 768), (3072, 768), (768, 3072), (144, 1280)])
def test_gemm_cdna4_vs_oracle(ops, dtype, variant, M, N, K):
    c = make_case(N, K, dtype, seed=M * 17 + N + K, M=M, bias=(M == 64))
    c4 = ops.repack_v2_to_cdna4(c["qweight"].cuda())
    ops._capi.tune(gemm_variant=variant)
    try:
        y = ops.gemm_cdna4(c["x"].cuda(), c4, c["scales"].cuda(), c["scaled_zeros"].cuda(),
                           c["bias"].cuda() if c["bias"] is not None else None).cpu()
    finally:
        ops._capi.tune(gemm_variant=0)
    check_forward(y, c["x"], c["q"], c["scales"], c["scaled_zeros"], dtype, bias=c["bias"])


@pytest.mark.parametrize("knobs", [dict(gemvc_pipe=2, gemvc_pipe_s=1, gemvc_waves=4), dict(gemvc_pipe=2, gemvc_pipe_s=2, gemvc_waves=4),
                                   dict(gemvc_pipe=2, gemvc_pipe_s=1, gemvc_waves=8), dict(gemvc_pipe=2, gemvc_pipe_s=2, gemvc_waves=8),
                                   dict(gemvc_pipe=2, gemvc_pipe_s=2, gemvc_waves=16), dict(gemvc_pipe=2, gemvc_pipe_s=1, gemvc_waves=16)])
def test_fast_gemv_knobs(ops, knobs):
    """the register-ring decode kernel (awq_gemv_cdna4.hip: W3, fused norm, grouped decode and the fallback of the streaming kernel):
    every shipped (waves, chunk) configuration incl. ragged step counts and more waves than steps."""
    try:
        ops._capi.tune(gemv_dma=0)
        for (N, K) in [(64, 11008), (128, 4096), (48, 1280), (32, 128)]:
            for M in (1, 3, 4, 5, 8):
                c = make_case(N, K, torch.bfloat16, seed=N + M, M=M, bias=True)
                c4 = ops.repack_v2_to_cdna4(c["qweight"].cuda())
                szp = ops.pack_sz_cdna4(c["scales"].cuda(), c["scaled_zeros"].cuda(), K)
                ops._capi.tune(**knobs)
                y = ops.gemm_cdna4(c["x"].cuda(), c4, c["scales"].cuda(), c["scaled_zeros"].cuda(), c["bias"].cuda(), szp)
                check_forward(y.cpu(), c["x"], c["q"], c["scales"], c["scaled_zeros"], torch.bfloat16, bias=c["bias"])
    finally:
        ops._capi.tune(gemvc_waves=0, gemvc_s=0, gemvc_pipe=-1, gemvc_pipe_s=0, gemv_dma=1)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 2, 4, 7, 8])
@pytest.mark.parametrize("F,K", [(256, 768), (1376, 512), (64, 4096)])
def test_fused_gate_up_silu_mul(ops, dtype, M, F, K):
    """one launch == the reference's QuantLlamaMLP sequence (fused_mlp.py:36-83): two GEMVs, F.silu, multiply, all in T."""
    cg = make_case(F, K, dtype, seed=F + K + M, M=M)
    cu = make_case(F, K, dtype, seed=F + K + M + 1, M=M)
    x = cg["x"]
    qgu = torch.cat([cg["qweight"], cu["qweight"]], 0).cuda()        # exactly how tinychat fuses q/k/v buffers
    s = torch.cat([cg["scales"], cu["scales"]], 1).cuda()
    z = torch.cat([cg["scaled_zeros"], cu["scaled_zeros"]], 1).cuda()
    c4 = ops.repack_v2_to_cdna4(qgu)
    szp = ops.pack_sz_cdna4(s, z, K)
    y = ops.mlp_gate_up_cdna4(x.cuda(), c4, szp).cpu()
    g = O.wqlinear_forward(x, None, cg["scales"], cg["scaled_zeros"], None, 128, q_int=cg["q"])
    u = O.wqlinear_forward(x, None, cu["scales"], cu["scaled_zeros"], None, 128, q_int=cu["q"])
    ref = torch.nn.functional.silu(g) * u
    rel = ((y.float() - ref.float()).norm() / ref.float().norm()).item()
    assert rel <= 1e-3 * 3, rel   # three bf16 roundings deep; exact-match fraction is the sharper check
    assert_bits(y, ref, 0.05)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [9, 16, 17, 32, 33, 48, 49, 64, 65, 129, 200, 255])
@pytest.mark.parametrize("N,K", [(768, 768), (1040, 1280), (64, 11008), (8192, 512), (16400, 256)])
def test_skinny_gemm_vs_oracle(ops, dtype, M, N, K):
    """9 <= M <= 255 (short prompts / batched decode; above 64 rows in chunks): the skinny kernel, every column-block count, slab counts that do not
    divide the slabs-per-block, ragged K splits, wide N (both slab groupings), bias fused."""
    c = make_case(N, K, dtype, seed=M * 7 + N + K, M=M, bias=(M % 2 == 1))
    c4 = ops.repack_v2_to_cdna4(c["qweight"].cuda())
    szp = ops.pack_sz_cdna4(c["scales"].cuda(), c["scaled_zeros"].cuda(), K)
    for small_m in (0, 1):  # 0: always the skinny kernel; 1 (default): rows >= 72 with m * K >= 0.6 M go to the prefill GEMM
        ops._capi.tune(gemm_small_m=small_m)
        try:
            y = ops.gemm_cdna4(c["x"].cuda(), c4, c["scales"].cuda(), c["scaled_zeros"].cuda(),
                               c["bias"].cuda() if c["bias"] is not None else None, szp).cpu()
        finally:
            ops._capi.tune(gemm_small_m=1)
        check_forward(y, c["x"], c["q"], c["scales"], c["scaled_zeros"], dtype, bias=c["bias"])
