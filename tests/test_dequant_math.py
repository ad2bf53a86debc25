"""not-gpu: the arithmetic of the matrix-core dequant (llm_awq_amd/csrc/awq_device.hpp, Cdna4DequantT) emulated step by step
in numpy -- every intermediate rounded to the precision the hardware keeps -- against the reference's contract
W = round_T(fma(q, s, sz)) (gemv_cuda.cu:159-166), evaluated exactly in float64.

  bf16:  A = 128 + q as a bf16 bit pattern (0x4300 | q), B = s, C = sz - 128 s (v_dot2_f32_bf16, fp32), D = A B + C in the
         MFMA's fp32 accumulator, then ONE rounding to bf16 (v_cvt_pk_bf16_f32)
  fp16:  offset 1024 (0x6400 | q), C = sz - 1024 s

and, for the next round's candidate, the f16-mantissa form for bf16 models: nibbles at mantissa bits 7:4 enter as
1024 + 16 q against s / 16 and C = sz - 64 s, with s and sz pre-scaled by 2^P into the f16 range."""
import numpy as np
import pytest
import torch

from tests.helpers import Gen


def _cases(dtype, emin, emax, seed):
    g = Gen(seed)
    n = 4096
    s = ((g.rand(n) * 2 + 0.5) * torch.pow(2.0, g.randint(emin, emax, (n,)).float())).to(dtype)
    z = g.randint(0, 16, (n,))
    sz = (-(s.float() * z.float())).to(dtype)  # qmodule.py:191-197: scaled_zeros = -(scales * zeros.float()).to(T): ROUNDED to T
    q = torch.arange(16).view(16, 1).expand(16, n)
    return q.numpy().astype(np.float64), s.double().numpy(), sz.double().numpy()


def _round_T(x64, dtype):
    return torch.from_numpy(np.ascontiguousarray(x64)).to(dtype)  # one rounding, float64 -> T


def _f32(x64):
    return x64.astype(np.float32).astype(np.float64)


@pytest.mark.parametrize("dtype,offset,emin,emax", [(torch.bfloat16, 128.0, -30, 8), (torch.float16, 1024.0, -12, 3)])
def test_offset_form_is_exact_before_the_single_rounding(dtype, offset, emin, emax):
    q, s, sz = _cases(dtype, emin, emax, seed=3)
    exact = q * s + sz                               # float64: exact (<= 12 + a few bits)
    want = _round_T(exact, dtype)
    c = _f32(_f32(sz) + _f32(-offset * s))            # the dot2 in fp32, each step rounded
    d = _f32(_f32((offset + q) * s) + c)              # product (exact in fp32) + C in the fp32 accumulator
    assert np.array_equal(d, exact), "the fp32 value in front of the rounding must be the exact q*s + sz"
    got = _round_T(d, dtype)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


def test_f16_mantissa_form_for_bf16_models_is_exact_in_its_range():
    P = 8
    q, s, sz = _cases(torch.bfloat16, -17, 3, seed=5)     # 2^-18 <= s <= 10; 15 s 2^P must stay below 65504 (the range a pack-time check must enforce)
    s_p, sz_p = s * 2.0 ** P, sz * 2.0 ** P
    for v in (s_p, sz_p, s_p / 16):                     # exactly representable f16 operands (normal range, 8-bit significands)
        assert np.array_equal(v.astype(np.float16).astype(np.float64), v)
    exact = (q * s + sz) * 2.0 ** P
    lo = _f32(_f32((1024.0 + q) * s_p) + _f32(_f32(sz_p) + _f32(-1024.0 * s_p)))                 # nibble at mantissa bits 3:0
    hi = _f32(_f32((1024.0 + 16.0 * q) * (s_p / 16)) + _f32(_f32(sz_p) + _f32(-64.0 * s_p)))     # nibble at bits 7:4, s / 16
    assert np.array_equal(lo, exact) and np.array_equal(hi, exact)
    want = _round_T(q * s + sz, torch.bfloat16).double() * 2.0 ** P     # round_T(2^P w) = 2^P round_T(w)
    assert torch.equal(_round_T(hi, torch.bfloat16).double(), want)
