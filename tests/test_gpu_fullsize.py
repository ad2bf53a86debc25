"""-m gpu: BASELINE.json's full Llama-3-8B layer shapes.  The oracle is too slow here, so use
size-independent checks: (a) the kernels against a plain torch fp32 matmul on the weights produced by
the (separately bit-exact-verified) dequant kernel; (b) GEMV and GEMM agree with each other on the same
rows; (c) linearity in x for power-of-two scalings (exact in floating point)."""
import pytest
import torch

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
    g = torch.Generator(device="cuda").manual_seed(7)
    for M, fn in [(1, ops.gemv), (7, ops.gemv), (16, ops.gemv), (64, ops.gemm), (300, ops.gemm), (2048, ops.gemm)]:
        x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
        y = fn(x, w["qweight"], w["scales"], w["scaled_zeros"]).float()
        ref = x.float() @ W.t()
        rel = ((y - ref).norm() / ref.norm()).item()
        # y is rounded to T (rel. spacing 2^-8 bf16 / 2^-11 fp16 -> rms ~ 1.1e-3 / 1.4e-4 of each element)
        tol = 2.5e-3 if dtype == torch.bfloat16 else 4e-4
        assert rel < tol, (M, rel)
        # against the reference rounded the same way, almost all elements are identical
        same = (ref.to(dtype).float() == y).float().mean().item()
        assert same > 0.97, (M, same)


def test_gemv_gemm_agree_and_scaling(env):
    ops, synth = env
    K, N = 4096, 14336
    w = synth.random_wq(K, N, dtype=torch.bfloat16, seed=1, keep_q=False)
    x = torch.randn(64, K, device="cuda").bfloat16()
    yg = ops.gemm(x, w["qweight"], w["scales"], w["scaled_zeros"])
    yv = ops.gemv(x[:7].contiguous(), w["qweight"], w["scales"], w["scaled_zeros"])
    same = (yg[:7] == yv).float().mean().item()
    assert same > 0.98, same
    y2 = ops.gemv((x[:7] * 4).contiguous(), w["qweight"], w["scales"], w["scaled_zeros"])
    assert torch.equal(y2, yv * 4)  # power-of-two scaling commutes with every rounding
