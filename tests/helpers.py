"""Shared test helpers: seeded synthetic cases built by the ORACLE (CPU) and error bounds."""
import numpy as np
import torch

from oracle import awq_oracle as O

MANT = {torch.float16: 10, torch.bfloat16: 7}


def make_case(N, K, dtype, seed=0, bias=False, M=1, x_scale=1.0):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(N, K, generator=g) * 0.02
    d = O.quantize_linear(w, dtype=dtype, n_bit=4, group_size=128)
    d["x"] = (torch.randn(M, K, generator=g) * x_scale).to(dtype)
    d["bias"] = (torch.randn(N, generator=g) * 0.02).to(dtype) if bias else None
    d["q"] = d["intweight"].numpy().astype(np.uint8)
    return d


def ulp(v: torch.Tensor, dtype) -> torch.Tensor:
    e = torch.floor(torch.log2(v.abs().double().clamp(min=1e-30)))
    return torch.pow(2.0, e - MANT[dtype])


def check_forward(y_gpu: torch.Tensor, x, q, scales, scaled_zeros, dtype, bias=None):
    """y_gpu (T, on cpu) against the oracle: (1) elementwise within half an ulp of T around the
    float64 contraction of the T-rounded weights, plus the fp32 accumulation slack 2e-6*sum|x||w|
    (plus one more rounding when a bias add follows); (2) norm-wise <= 1e-3 (BASELINE.json);
    (3) the fp32-accumulate oracle agrees on almost every element bit for bit."""
    W = O.dequant_weight(q, scales, scaled_zeros, 128)
    K = x.shape[-1]
    x2 = x.reshape(-1, K)
    y64 = x2.double() @ W.double().t()
    S = x2.double().abs() @ W.double().abs().t()
    yg = y_gpu.reshape(y64.shape).double()
    bound = 0.501 * ulp(y64, dtype) + 2e-6 * S + 1e-30
    if bias is not None:
        y64 = y64 + bias.double()
        bound = bound + 0.501 * ulp(y64, dtype) + ulp(y64, dtype) * 0.5
    err = (yg - y64).abs()
    worst = (err / bound).max().item()
    assert worst <= 1.0, f"elementwise bound violated: worst err/bound = {worst}"
    # BASELINE.json: "<= 1e-3 rel on the fp16/bf16 matmul result" -- norm-wise against the oracle's T-rounded
    # output (the final rounding to bf16 alone is ~1.1e-3 rms per element, so it must be on both sides)
    y_or = O.wqlinear_forward(x, None, scales, scaled_zeros, bias, 128, q_int=q).reshape(y64.shape).double()
    rel = ((yg - y_or).norm() / y_or.norm()).item()
    assert rel <= 1e-3, rel
    mism = (y_or != yg).double().mean().item()
    # (two elements are always allowed: tiny outputs -- 64 values -- would otherwise fail on a pair of 1-ulp flips)
    assert mism <= max(0.02, 2.5 / y_or.numel()), f"{mism*100:.2f}% of elements differ from the fp32-accumulate oracle"
    return worst, rel, mism
