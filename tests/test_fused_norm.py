"""RMSNorm fused in front of the quantised linear (awq/kernels/csrc/layernorm/layernorm.cu:39-61 +
tinychat/modules/fused_norm.py:7-21 feeding WQLinear.forward / QuantLlamaMLP)."""
import numpy as np
import pytest
import torch

from oracle import awq_oracle as O
from tests.helpers import Gen, assert_bits, check_forward, make_case, record_rel, rmsnorm_uncertainty

REL_NORM_TAIL = 1e-3  # fused norm + gate/up tail, norm-wise: BASELINE.json's tolerance (measured on MI355X: <= 5.4e-5, profiles/r05_test_stats.txt)


def test_oracle_rmsnorm_matches_llama_rmsnorm_formula():
    """FTLlamaRMSNorm "is equivalent to T5LayerNorm" (fused_norm.py:10-13): the oracle restatement against the textbook
    LlamaRMSNorm computation in fp64 -- at most one ulp of T apart (different rounding points)."""
    g = Gen(0)
    for dtype, tol in ((torch.bfloat16, 2.0 ** -7), (torch.float16, 2.0 ** -10)):
        x = (g.randn(3, 512) * 3).to(dtype)
        gamma = (1 + 0.1 * g.randn(512)).to(dtype)
        got = O.rmsnorm(x, gamma, 1e-6).double()
        xd = x.double()
        want = xd * torch.rsqrt((xd * xd).mean(-1, keepdim=True) + 1e-6) * gamma.double()
        assert ((got - want).abs() <= tol * want.abs() + 1e-12).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 2, 3, 4])
@pytest.mark.parametrize("N,K", [(768, 768), (256, 4096), (1024, 2048), (64, 8192)])
def test_gpu_rmsnorm_linear_vs_oracle(dtype, M, N, K):
    from llm_awq_amd import ops
    c = make_case(N, K, dtype, seed=M + N + K, M=M, bias=(M == 2))
    g = Gen(N + M)
    x = (g.randn(M, K) * 2.5).to(dtype)
    gamma = (1 + 0.2 * g.randn(K)).to(dtype)
    eps = 1e-5
    c4 = ops.repack_v2_to_cdna4(c["qweight"].cuda())
    szp = ops.pack_sz_cdna4(c["scales"].cuda(), c["scaled_zeros"].cuda(), K)
    y = ops.rmsnorm_forward_cdna4(x.cuda(), gamma.cuda(), eps, c4, szp, c["bias"].cuda() if c["bias"] is not None else None).cpu()
    xn = O.rmsnorm(x, gamma, eps)
    ref = O.wqlinear_forward(xn, None, c["scales"], c["scaled_zeros"], c["bias"], 128, q_int=c["q"])
    # the normalised activations can differ in the last bit where the fp32 sum of squares is reduced in another order / the
    # hardware rsqrt rounds the other way: those x carry one ulp of T of slack (tests/helpers.rmsnorm_uncertainty)
    check_forward(y, xn, c["q"], c["scales"], c["scaled_zeros"], dtype, bias=c["bias"], x_unc=rmsnorm_uncertainty(x, gamma, eps))
    # and against the two-launch product path: oracle-normalised x through the plain kernel
    y2 = ops.gemm_cdna4(xn.cuda(), c4, c["scales"].cuda(), c["scaled_zeros"].cuda(), c["bias"].cuda() if c["bias"] is not None else None, szp).cpu()
    assert_bits(y, y2, 0.1)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 4])
def test_gpu_rmsnorm_gate_up_vs_oracle(dtype, M):
    from llm_awq_amd import ops
    F, K = 1376, 2048
    cg = make_case(F, K, dtype, seed=F + M, M=M)
    cu = make_case(F, K, dtype, seed=F + M + 1, M=M)
    g = Gen(M)
    x = (g.randn(M, K) * 1.7).to(dtype)
    gamma = (1 + 0.2 * g.randn(K)).to(dtype)
    qgu = torch.cat([cg["qweight"], cu["qweight"]], 0).cuda()
    s = torch.cat([cg["scales"], cu["scales"]], 1).cuda()
    z = torch.cat([cg["scaled_zeros"], cu["scaled_zeros"]], 1).cuda()
    c4 = ops.repack_v2_to_cdna4(qgu)
    szp = ops.pack_sz_cdna4(s, z, K)
    y = ops.rmsnorm_forward_cdna4(x.cuda(), gamma.cuda(), 1e-6, c4, szp, None, fused_gate_up=True).cpu()
    xn = O.rmsnorm(x, gamma, 1e-6)
    gt = O.wqlinear_forward(xn, None, cg["scales"], cg["scaled_zeros"], None, 128, q_int=cg["q"])
    up = O.wqlinear_forward(xn, None, cu["scales"], cu["scaled_zeros"], None, 128, q_int=cu["q"])
    ref = torch.nn.functional.silu(gt) * up
    # (the normalised x may differ in the last bit of T where rstd's last fp32 bits decide -- rmsnorm_uncertainty -- so gate / up can move by more
    # than one ulp and the one-ulp hull of check_fused_tail does not apply; the norm-wise bound is BASELINE.json's 1e-3 -- measured <= 5.4e-5,
    # profiles/r05_test_stats.txt -- beside the flip count)
    rel = ((y.float() - ref.float()).norm() / ref.float().norm()).item()
    record_rel(f"rmsnorm + gate/up M={M}", rel, REL_NORM_TAIL)
    assert rel <= REL_NORM_TAIL, rel
    assert_bits(y, ref, 0.15)


@pytest.mark.gpu
def test_gpu_rmsnorm_forward_rejects_what_it_cannot_serve():
    from llm_awq_amd import ops, _capi
    c = make_case(256, 512, torch.bfloat16, seed=1, M=5)
    c4 = ops.repack_v2_to_cdna4(c["qweight"].cuda())
    szp = ops.pack_sz_cdna4(c["scales"].cuda(), c["scaled_zeros"].cuda(), 512)
    gamma = torch.ones(512, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(_capi.AwqNativeError):
        ops.rmsnorm_forward_cdna4(c["x"].cuda(), gamma, 1e-6, c4, szp)  # M = 5 > 4


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_module_matches_norm_then_linear(dtype):
    """llm_awq_amd.fused_norm.RMSNormWQLinear through the torch extension's export: decode rows take the fused launch, more rows
    the separate norm + WQLinear; both equal oracle norm -> oracle forward."""
    from llm_awq_amd.fused_norm import RMSNormWQLinear
    from llm_awq_amd.qmodule import WQLinear
    from oracle import awq_oracle as O
    from tests.helpers import check_forward, make_case, rmsnorm_uncertainty
    N, K, eps = 256, 4096, 1e-5
    c = make_case(N, K, dtype, seed=21, M=16, bias=True)
    gamma = (1.0 + 0.1 * Gen(22).randn(K)).to(dtype)
    lin = WQLinear(4, 128, K, N, True, "cuda", dtype=dtype)
    lin.load_state_dict(dict(qweight=c["qweight"], scales=c["scales"], scaled_zeros=c["scaled_zeros"], bias=c["bias"]))
    lin.to_cdna4()
    mod = RMSNormWQLinear(gamma.cuda(), eps, lin)
    for M in (1, 3, 4, 5, 16):
        x = c["x"][:M].contiguous()
        xn = O.rmsnorm(x, gamma, eps)
        # the fused launch (M <= 4) normalises with the hardware rsqrt and its own reduction order (as layernorm.cu:48-60 does
        # on its hardware): where T(x * rstd * gamma) hangs on the last fp32 bits of rstd, one ulp of T of slack on that x
        unc = rmsnorm_uncertainty(x, gamma, eps)
        check_forward(mod(x.cuda()).cpu(), xn, c["q"], c["scales"], c["scaled_zeros"], dtype, bias=c["bias"], x_unc=unc)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K", [(1, 8), (3, 768), (5, 4096), (300, 4096), (17, 14336), (2, 11008)])
def test_standalone_rmsnorm_vs_oracle(dtype, M, K):
    """awq_rmsnorm (any row count; layernorm.cu:39-61's arithmetic) against the oracle: identical except where the rounding of
    T(x * rstd * gamma) hangs on the last fp32 bits of rstd (hardware rsqrt, another order of the fp32 sum of squares)."""
    from llm_awq_amd import ops
    g = Gen(M * 7 + K)
    x = (g.randn(M, K) * 3).to(dtype)
    gamma = (1 + 0.2 * g.randn(K)).to(dtype)
    eps = 1e-5
    y = ops.rmsnorm(x.cuda(), gamma.cuda(), eps).cpu()
    ref = O.rmsnorm(x, gamma, eps)
    unc = rmsnorm_uncertainty(x, gamma, eps)
    diff = (y.double() - ref.double()).abs()
    assert (diff <= unc + 1e-30).all(), "a difference outside the elements whose rounding is undecided"
    assert_bits(y, ref, 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -13, what="rmsnorm")


def test_module_has_no_torch_fallback():
    """what the HIP norm does not take raises (CPU tensors here): nothing under llm_awq_amd/ computes the step in torch"""
    from llm_awq_amd.fused_norm import RMSNormWQLinear
    import llm_awq_amd.fused_norm as FN
    assert not hasattr(FN, "rmsnorm_reference_semantics")
    mod = RMSNormWQLinear(torch.ones(128, dtype=torch.bfloat16), 1e-5, torch.nn.Identity())
    with pytest.raises(RuntimeError, match="no CPU / PyTorch fallback"):
        mod(torch.zeros(9, 128, dtype=torch.bfloat16))
