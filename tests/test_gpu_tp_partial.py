"""-m gpu: the K-shard fp32 partial of a tensor-parallel row split (awq_w4a16_partial_cdna4, awq_round_bias_f32).

  * every kernel the entry point dispatches to (streaming decode, skinny decode, skinny, masked narrow tile, 128 / 192 / 256-wide prefill
    blocks) writes its fp32 accumulators: against the float64 contraction of the oracle's T-rounded weights, within fp32 accumulation error;
  * a `world`-way row split, every rank's shard run one after the other on the one device (TPWQLinear with explicit world / rank), summed in
    fp32 and rounded once by awq_round_bias_f32: within 1e-3 norm-wise of the SINGLE-DEVICE ORACLE (oracle.wqlinear_forward) in bf16 and fp16
    -- the budget T-rounded partials missed (2.6-2.9e-3 in bf16) -- and nearly bit-identical to it.
"""
import pytest
import torch

from oracle import awq_oracle as O
from tests.helpers import Gen, assert_bits, cuda_gen, make_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from llm_awq_amd import ops as _ops
    return _ops


def _bufs(ops, c, K):
    qw = ops.repack_v2_to_cdna4(c["qweight"].cuda())
    s, z = c["scales"].cuda(), c["scaled_zeros"].cuda()
    szp = ops.pack_sz_cdna4(s, z, K)
    szh, exact = ops.pack_szh_cdna4(s, z, K)
    return qw, szp, (szh if exact else None)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,K", [(64, 512), (256, 3584), (4096, 1024), (1024, 14336)])
def test_partial_is_the_unrounded_fp32_product(ops, dtype, N, K):
    c = make_case(N, K, dtype, seed=N + K, M=600)
    qw, szp, szh = _bufs(ops, c, K)
    W = O.dequant_weight(c["q"], c["scales"], c["scaled_zeros"], 128).double().cuda()
    for M in (1, 3, 5, 8, 9, 40, 100, 255, 256, 300, 600):
        x = c["x"][:M].contiguous().cuda()
        for side in ((szh, None) if szh is not None and M <= 8 else (None,)):
            y = ops.partial_cdna4(x, qw, szp, side)
            assert y.dtype == torch.float32 and y.shape == (M, N)
            y64 = x.double() @ W.t()
            S = x.double().abs() @ W.abs().t()
            worst = ((y.double() - y64).abs() / (3e-6 * S + 1e-30)).max().item()
            assert worst <= 1.0, (M, side is not None, worst)
            # ... and NOT the T-rounded value: the product path's output is its rounding
            y_t = ops.gemm_cdna4(x, qw, c["scales"].cuda(), c["scaled_zeros"].cuda(), None, szp)
            assert_bits(y.to(dtype), y_t, 0.02, "T(partial) vs the T kernel (another fp32 summation order on some paths)")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_round_bias_f32(ops, dtype):
    g = cuda_gen(3)
    for (m, n) in ((1, 8), (3, 4096), (300, 264)):
        y32 = torch.randn(m, n, device="cuda", generator=g) * 3
        b = (torch.randn(n, device="cuda", generator=g) * 0.1).to(dtype)
        assert torch.equal(ops.round_bias_f32(y32, dtype), y32.to(dtype))
        assert torch.equal(ops.round_bias_f32(y32, dtype, b), y32.to(dtype) + b)
    assert ops.round_bias_f32(torch.empty(0, 64, device="cuda"), dtype).shape == (0, 64)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_row_split_sums_to_the_single_device_oracle(ops, dtype, world):
    """the whole row-parallel layer against O.wqlinear_forward on the unsharded buffers"""
    from llm_awq_amd import parallel as P
    from llm_awq_amd.qmodule import WQLinear
    K, N = 4096, 512
    c = make_case(N, K, dtype, seed=77 + world, M=300, bias=True)
    full = WQLinear(4, 128, K, N, True, "cuda", dtype=dtype)
    full.qweight, full.scales, full.scaled_zeros, full.bias = c["qweight"].cuda(), c["scales"].cuda(), c["scaled_zeros"].cuda(), c["bias"].cuda()
    shards = [P.TPWQLinear(full, "row", world=world, rank=r) for r in range(world)]
    assert all(t.shard.layout == "cdna4" and t._reducer is None for t in shards)
    for M in (1, 7, 64, 300):
        x = c["x"][:M].contiguous()
        ref = O.wqlinear_forward(x, None, c["scales"], c["scaled_zeros"], c["bias"], 128, q_int=c["q"]).float()
        xg = x.cuda()
        acc = torch.zeros(M, N, device="cuda")
        acc_t = torch.zeros(M, N, device="cuda")
        for t in shards:
            p32 = t.partial(xg)
            assert p32.dtype == torch.float32
            acc += p32
            acc_t += p32.to(dtype).float()
        y = ops.round_bias_f32(acc, dtype, full.bias).float().cpu()
        rel = ((y - ref).norm() / ref.norm()).item()
        assert rel < 1e-3, (M, rel)  # the norm-wise budget, bf16 included
        assert_bits(y, ref, 0.01, "row split vs single-device oracle")
        if dtype == torch.bfloat16:
            rel_t = (((acc_t.to(dtype) + full.bias).float().cpu() - ref).norm() / ref.norm()).item()
            assert rel < rel_t


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_llama3_70b_tp8_down_proj_row_split(ops, dtype):
    """BASELINE.json config 4 at its real shard size: down_proj 28672 -> 8192 over 8 ranks (3584 k each), decode and prefill rows.  The
    reference here is the float64 contraction of the device-dequantised weights (awq_dequant_cdna4 is bit-pinned to the oracle in
    test_gpu_oracle_fullsize.py) rounded to T once: the single-device oracle's definition, evaluated on the GPU for the 235 M weights."""
    from llm_awq_amd import parallel as P, synth
    K, N, world = 28672, 8192, 8
    w = synth.random_wq(K, N, dtype=dtype, seed=7, keep_q=False)
    W = ops.dequant_v2(w["qweight"], w["scales"], w["scaled_zeros"])
    g = cuda_gen(5)
    parts = []
    for r in range(world):
        qw, s, z, (k0, k1) = P.shard_row_parallel(w["qweight"], w["scales"], w["scaled_zeros"], world, r)
        assert k1 - k0 == 3584
        parts.append((ops.repack_v2_to_cdna4(qw), ops.pack_sz_cdna4(s, z, k1 - k0), k0, k1))
    for M in (1, 300, 2048):
        x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
        ref = torch.empty(M, N, device="cuda", dtype=dtype)
        for n0 in range(0, N, 1024):  # float64 in column panels
            ref[:, n0:n0 + 1024] = (x.double() @ W[n0:n0 + 1024].double().t()).to(dtype)
        acc = torch.zeros(M, N, device="cuda")
        for (c4, szp, k0, k1) in parts:
            acc += ops.partial_cdna4(x[:, k0:k1].contiguous(), c4, szp, None)
        y = ops.round_bias_f32(acc, dtype)
        rel = ((y.float() - ref.float()).norm() / ref.float().norm()).item()
        assert rel < 1e-3, (M, rel)
        assert_bits(y, ref, 0.01, "8-way row split vs float64")
