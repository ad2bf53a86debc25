"""-m gpu: the HIP path against the CPU oracle, called THROUGH THE C ABI (llm_awq_amd.ops -> ctypes
-> libawq_cdna4.so).  Integer / index work must be bit exact; matmul within the bounds in
tests/helpers.py (half an ulp of T around the exact contraction + fp32 accumulation slack, and
<= 1e-3 norm-wise as BASELINE.json states)."""
import numpy as np
import pytest
import torch

from oracle import awq_oracle as O
from tests.helpers import check_forward, make_case, Gen

pytestmark = pytest.mark.gpu
DTYPES = [torch.float16, torch.bfloat16]


@pytest.fixture(scope="module")
def ops():
    from llm_awq_amd import ops as _ops
    _ops._capi.lib()  # fail loudly if the HIP library is missing
    return _ops


@pytest.mark.parametrize("N,K", [(4, 64), (16, 128), (64, 256), (128, 768), (768, 3072), (1024, 4096)])
def test_unpack_and_pack_bit_exact(ops, N, K):
    rng = np.random.default_rng(N * 7 + K)
    q = rng.integers(0, 16, size=(N, K)).astype(np.uint8)
    packed = O.pack_v2(q)
    got = ops.unpack_v2(torch.from_numpy(packed).cuda()).cpu().numpy()
    assert (got == q).all()
    got_p = ops.pack_v2(torch.from_numpy(q).cuda()).cpu().numpy()
    assert (got_p == packed).all()


def test_unpack_golden(ops, golden):
    g = golden("pack_v2.npz")
    for key in ["0", "1", "2", "3", "4", "_struct"]:
        q, p = g["q" + key], g["p" + key]
        assert (ops.unpack_v2(torch.from_numpy(p).cuda()).cpu().numpy() == q).all()
        assert (ops.pack_v2(torch.from_numpy(q).cuda()).cpu().numpy() == p).all()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,K", [(16, 128), (64, 768), (256, 1280), (768, 3072), (512, 4096)])
def test_dequant_bit_exact(ops, dtype, N, K):
    c = make_case(N, K, dtype, seed=N + K)
    W = O.dequant_weight(c["q"], c["scales"], c["scaled_zeros"], 128)
    got = ops.dequant_v2(c["qweight"].cuda(), c["scales"].cuda(), c["scaled_zeros"].cuda()).cpu()
    assert torch.equal(got.view(torch.int16), W.view(torch.int16))


@pytest.mark.parametrize("dtype", DTYPES)
def test_dequant_adversarial_scales(ops, dtype):
    """random scales over a wide exponent range + every zero point: rounding ties, subnormal fp16."""
    g = Gen(5)
    N, K = 64, 512
    q = g.randint(0, 16, (N, K)).numpy().astype(np.uint8)
    scales = torch.zeros(8, N, dtype=dtype)
    scales[:4] = (g.rand(4, N) * 2 + 0.5) * torch.pow(2.0, g.randint(-14, 3, (4, N)).float())
    zeros = g.randint(0, 16, (4, N))
    sz = torch.zeros(8, N, dtype=dtype)
    sz[:4] = -(scales[:4] * zeros.float()).to(dtype)
    W = O.dequant_weight(q, scales, sz, 128)
    got = ops.dequant_v2(torch.from_numpy(O.pack_v2(q)).cuda(), scales.cuda(), sz.cuda()).cpu()
    assert torch.equal(got.view(torch.int16), W.view(torch.int16))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [1, 2, 4, 5, 7, 8, 13, 16])
@pytest.mark.parametrize("N,K", [(768, 768), (3072, 768), (768, 3072), (256, 4096), (1032, 1280)])
def test_gemv_vs_oracle(ops, dtype, M, N, K):
    c = make_case(N, K, dtype, seed=M * 131 + N + K, M=M)
    y = ops.gemv(c["x"].cuda(), c["qweight"].cuda(), c["scales"].cuda(), c["scaled_zeros"].cuda()).cpu()
    check_forward(y, c["x"], c["q"], c["scales"], c["scaled_zeros"], dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [1, 4, 5, 8])
@pytest.mark.parametrize("N,K", [(768, 768), (1024, 4096), (512, 14336)])
def test_gemv_reference_layout_kernels_agree(ops, dtype, M, N, K):
    """The pipelined reference-layout decode kernel (awq_v2_kernels.hip part 1, default for M <= 8, N % 16 == 0) and the older
    kernel of awq_gemv.hip (knob gemv_v2fast=0) both meet the oracle; they differ only in fp32 summation order."""
    c = make_case(N, K, dtype, seed=M + N + K, M=M)
    args = (c["x"].cuda(), c["qweight"].cuda(), c["scales"].cuda(), c["scaled_zeros"].cuda())
    y_new = ops.gemv(*args).cpu()
    ops._capi.tune(gemv_v2fast=0)
    try:
        y_old = ops.gemv(*args).cpu()
    finally:
        ops._capi.tune(gemv_v2fast=1)
    check_forward(y_new, c["x"], c["q"], c["scales"], c["scaled_zeros"], dtype)
    check_forward(y_old, c["x"], c["q"], c["scales"], c["scaled_zeros"], dtype)
    # one-ulp-of-T differences at most (same products, same fp32 accumulator width)
    d = (y_new.float() - y_old.float()).abs().max().item()
    ref = y_old.float().abs().max().item()
    assert d <= ref * (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -9)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [9, 17, 33, 48, 64, 65, 255])
@pytest.mark.parametrize("N,K", [(768, 768), (1040, 1280), (64, 11008), (8192, 512)])
def test_skinny_reference_layout_vs_oracle(ops, dtype, M, N, K):
    """9 <= M <= 255 on un-repacked buffers (fp16 and bf16): the v2-layout skinny kernel behind gemm_forward_cuda_new / forward,
    every column-block count, slab counts that do not divide the slabs-per-block, ragged K splits, row chunks, fused bias."""
    c = make_case(N, K, dtype, seed=M * 5 + N + K, M=M, bias=(M % 2 == 1))
    args = (c["x"].cuda(), c["qweight"].cuda(), c["scales"].cuda(), c["scaled_zeros"].cuda())
    if c["bias"] is not None:
        y = ops.forward(*args, c["bias"].cuda()).cpu()
    else:
        y = ops.gemm(*args).cpu()
    check_forward(y, c["x"], c["q"], c["scales"], c["scaled_zeros"], dtype, bias=c["bias"])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("variant,M", [(0, m) for m in (8, 17, 64, 100, 128, 129, 200, 512, 777)] +   # auto: every row count
                         [(v, m) for v in (1, 2) for m in (17, 128, 129, 777)])                       # forced 128x128 / 256x256 tiles
@pytest.mark.parametrize("N,K", [(768, 768), (3072, 768), (768, 3072), (136, 1280)])
def test_gemm_vs_oracle(ops, dtype, variant, M, N, K):
    c = make_case(N, K, dtype, seed=M * 17 + N + K, M=M)
    ops._capi.tune(gemm_variant=variant)
    try:
        y = ops.gemm(c["x"].cuda(), c["qweight"].cuda(), c["scales"].cuda(), c["scaled_zeros"].cuda()).cpu()
    finally:
        ops._capi.tune(gemm_variant=0)
    check_forward(y, c["x"], c["q"], c["scales"], c["scaled_zeros"], dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [1, 7, 8, 300])
def test_forward_dispatch_with_bias(ops, dtype, M):
    c = make_case(768, 768, dtype, seed=M, M=M, bias=True)
    y = ops.forward(c["x"].cuda(), c["qweight"].cuda(), c["scales"].cuda(), c["scaled_zeros"].cuda(),
                    c["bias"].cuda()).cpu()
    check_forward(y, c["x"], c["q"], c["scales"], c["scaled_zeros"], dtype, bias=c["bias"])


def test_identity_activation_is_transpose_detecting(ops):
    """x = I (K=128 rows of the identity): out[m, n] must equal W_T[n, m] exactly (catches an m/n swap
    or any k permutation mismatch between the two MFMA operands)."""
    for dtype in DTYPES:
        N, K = 64, 128
        c = make_case(N, K, dtype, seed=3)
        W = O.dequant_weight(c["q"], c["scales"], c["scaled_zeros"], 128)
        x = torch.eye(K, dtype=dtype)
        for fn, rows in ((ops.gemm, K), (ops.gemv, 16)):
            y = fn(x[:rows].contiguous().cuda(), c["qweight"].cuda(), c["scales"].cuda(), c["scaled_zeros"].cuda()).cpu()
            assert torch.equal(y.view(torch.int16), W.t()[:rows].contiguous().view(torch.int16))


def test_error_codes(ops):
    from llm_awq_amd import _capi
    c = make_case(64, 256, torch.float16, M=1)
    args = [t.cuda() for t in (c["x"], c["qweight"], c["scales"], c["scaled_zeros"])]
    with pytest.raises(_capi.AwqNativeError, match="group size"):
        ops.gemv(*args, group_size=64)
    x17 = torch.zeros(17, 256, dtype=torch.float16).cuda()
    with pytest.raises(_capi.AwqNativeError, match="batch size"):
        ops.gemv(x17, *args[1:])
    with pytest.raises(_capi.AwqNativeError):
        ops.gemv(c["x"], c["qweight"], c["scales"], c["scaled_zeros"])  # CPU tensors: no fallback


def test_repack_v1_to_v2_golden(ops, golden):
    g = golden("repack_v1_v2.npz")
    for i in range(2):
        dt = torch.float16 if int(g[f"dtype_{i}"][0]) == 0 else torch.bfloat16
        qw1 = torch.from_numpy(g[f"qw1_{i}"]).cuda()
        qz1 = torch.from_numpy(g[f"qz1_{i}"]).cuda()
        sc1 = torch.from_numpy(g[f"sc1_{i}"]).view(dt).cuda()
        qw2, s2, sz2 = ops.repack_v1_to_v2(qw1, sc1, qz1)
        assert (qw2.cpu().numpy() == g[f"qw2_{i}"]).all()
        assert (s2.cpu().view(torch.int16).numpy() == g[f"sc2_{i}"]).all()
        assert (sz2.cpu().view(torch.int16).numpy() == g[f"sz2_{i}"]).all()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M", [1, 7, 8, 9, 64, 255, 256, 257, 600])
@pytest.mark.parametrize("N,K", [(16, 128), (32, 256), (48, 384), (272, 128)])
def test_smallest_shapes_every_dispatch_boundary(ops, dtype, M, N, K):
    """One or two quantisation groups, one to seventeen slabs, M on both sides of every kernel boundary (GEMV <= 8, skinny
    9..255, tiled GEMM >= 256), reference layout and cdna4, with and without bias."""
    c = make_case(N, K, dtype, seed=M + N + K, M=M, bias=(M % 2 == 0))
    b = c["bias"].cuda() if c["bias"] is not None else None
    qw, s, z, x = c["qweight"].cuda(), c["scales"].cuda(), c["scaled_zeros"].cuda(), c["x"].cuda()
    y2 = ops.forward(x, qw, s, z, b).cpu() if b is not None else (ops.gemv(x, qw, s, z) if M < 8 else ops.gemm(x, qw, s, z)).cpu()
    check_forward(y2, c["x"], c["q"], c["scales"], c["scaled_zeros"], dtype, bias=c["bias"])
    c4 = ops.repack_v2_to_cdna4(qw)
    szp = ops.pack_sz_cdna4(s, z, K)
    y4 = ops.gemm_cdna4(x, c4, s, z, b, szp).cpu()
    check_forward(y4, c["x"], c["q"], c["scales"], c["scaled_zeros"], dtype, bias=c["bias"])
