"""not-gpu: the oracle (oracle/awq_oracle.py) against the golden vectors produced by RUNNING the
reference's own Python (oracle/gen_golden.py).  Integer / index work is bit exact."""
import numpy as np
import pytest
import torch

from tests.helpers import Gen
from hypothesis import given, settings, strategies as st

from oracle import awq_oracle as O
from tests.conftest import as_t

CASES = [("f16_a", torch.float16, True), ("bf16_a", torch.bfloat16, False), ("f16_b", torch.float16, False),
         ("bf16_b", torch.bfloat16, True)]


def test_pack_unpack_golden(golden):
    g = golden("pack_v2.npz")
    for key in ["0", "1", "2", "3", "4", "_struct"]:
        q, p = g["q" + key], g["p" + key]
        assert (O.pack_v2(q) == p).all()
        assert (O.unpack_v2(p) == q).all()


def test_word_view_of_v2(golden):
    """word w of a row's 32-k chunk holds k = [2w, 2w+8, 2w+16, 2w+24, 2w+1, ...]."""
    g = golden("pack_v2.npz")
    q, p = g["q_struct"], g["p_struct"]
    words = p.view(np.uint32)  # [N/4, K/2]
    N, K = q.shape
    for n in (0, 5, 15):
        for chunk in (0, 1, 3):
            base = (chunk // 2) * 32 + (n % 4) * 8 + (chunk % 2) * 4
            for w in range(4):
                word = int(words[n // 4, base + w])
                ks = [2 * w, 2 * w + 8, 2 * w + 16, 2 * w + 24, 2 * w + 1, 2 * w + 9, 2 * w + 17, 2 * w + 25]
                for nib, kl in enumerate(ks):
                    assert (word >> (4 * nib)) & 0xF == q[n, chunk * 32 + kl]


@settings(max_examples=25, deadline=None)
@given(st.integers(1, 6), st.integers(1, 6), st.integers(0, 2**31 - 1))
def test_pack_roundtrip_property(nq, kb, seed):
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 16, size=(4 * nq, 64 * kb)).astype(np.uint8)
    assert (O.unpack_v2(O.pack_v2(q)) == q).all()


def test_zeros_width_golden(golden):
    for K, G, w in golden("zeros_width.npz")["table"]:
        assert O.zeros_width(int(K), int(G)) == int(w)
    assert O.padded_groups(768) == 8 and O.padded_groups(11008) == 88 and O.padded_groups(14336) == 112


@pytest.mark.parametrize("name,dt,has_bias", CASES)
def test_pseudo_quant_and_from_linear_golden(golden, name, dt, has_bias):
    g = golden("from_linear.npz")
    w0 = as_t(g[name + "_w0"], dt)
    fake, s, z = O.pseudo_quantize(w0, 4, 128)
    assert torch.equal(fake, as_t(g[name + "_wfake"], dt))
    assert torch.equal(s, as_t(g[name + "_s"], dt)) and torch.equal(z, as_t(g[name + "_z"], dt))
    qw, sc, sz, iw = O.wq_buffers_from_fake(fake, s, z, 128)
    assert (qw.numpy() == g[name + "_qweight"]).all()
    assert torch.equal(sc, as_t(g[name + "_scales"], dt))
    assert torch.equal(sz, as_t(g[name + "_scaled_zeros"], dt))
    assert iw.min() >= 0 and iw.max() <= 15


@pytest.mark.parametrize("name,dt,has_bias", CASES)
def test_dequant_reproduces_fake_weight(golden, name, dt, has_bias):
    """pin (2) of SURVEY 8(c): dequant(from_linear(fake)) == fake to <= 1 ulp of T."""
    g = golden("from_linear.npz")
    fake = as_t(g[name + "_wfake"], dt)
    q = O.unpack_v2(g[name + "_qweight"])
    sc, sz = as_t(g[name + "_scales"], dt), as_t(g[name + "_scaled_zeros"], dt)
    W = O.dequant_weight(q, sc, sz, 128)
    # both sides are roundings to T of (almost) the same real number; the operands q*s and sz are up to
    # 16x larger than the result, so the bound is one ulp of T at the magnitude of the largest operand
    gi = torch.arange(fake.shape[1]) // 128
    mag = torch.maximum(sz.float().abs()[gi].t(), 15.0 * sc.float()[gi].t())
    ulp = 2.0 ** (torch.floor(torch.log2(mag.clamp(min=1e-20))) - (10 if dt == torch.float16 else 7))
    assert ((W.float() - fake.float()).abs() <= ulp).all()


@pytest.mark.parametrize("name,dt,has_bias", CASES)
def test_real_forward_close_to_fake_forward(golden, name, dt, has_bias):
    """pin (3): WQLinear oracle(x) ~= F.linear(x, pseudo_quantize_tensor(w)) (README 'fake' vs 'real')."""
    g = golden("from_linear.npz")
    x = as_t(g[name + "_x"], dt)
    bias = as_t(g[name + "_bias"], dt) if has_bias else None
    y = O.wqlinear_forward(x, torch.from_numpy(g[name + "_qweight"]), as_t(g[name + "_scales"], dt),
                           as_t(g[name + "_scaled_zeros"], dt), bias, 128).float()
    yf = torch.from_numpy(g[name + "_yfake32"])
    rel = ((y - yf).norm() / yf.norm()).item()
    assert rel < (2e-3 if dt == torch.float16 else 8e-3), rel


def test_pseudo_w3_golden(golden):
    g = golden("pseudo_w3.npz")
    for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16), ("f32", torch.float32)):
        w0 = as_t(g[name + "_w0"], dt)
        fake, s, z = O.pseudo_quantize(w0, 3, 128)
        assert torch.equal(fake, as_t(g[name + "_wfake"], dt))
        assert torch.equal(s, as_t(g[name + "_s"], dt)) and torch.equal(z, as_t(g[name + "_z"], dt))
        assert z.max() <= 7


def test_repack_v1_v2_golden(golden):
    g = golden("repack_v1_v2.npz")
    for i in range(2):
        dt = torch.float16 if int(g[f"dtype_{i}"][0]) == 0 else torch.bfloat16
        assert (O.unpack_v1(g[f"qw1_{i}"]) == g[f"unpacked_{i}"]).all()
        assert (O.pack_v1(g[f"unpacked_{i}"]) == g[f"qw1_{i}"]).all()
        qw2, s2, sz2 = O.repack_v1_to_v2(g[f"qw1_{i}"], as_t(g[f"sc1_{i}"], dt), g[f"qz1_{i}"])
        assert (qw2.numpy() == g[f"qw2_{i}"]).all()
        assert (s2.view(torch.int16).numpy() == g[f"sc2_{i}"]).all()
        assert (sz2.view(torch.int16).numpy() == g[f"sz2_{i}"]).all()


def test_dequant_is_single_rounded_fma():
    """the fp32 mul+add used by the oracle is exact before the one rounding to T: compare with float64."""
    g = Gen(0)
    for dt in (torch.float16, torch.bfloat16):
        s = ((g.rand(4, 64) + 0.5) * 2.0 ** g.randint(-12, 2, (4, 64)).float()).to(dt)
        z = g.randint(0, 16, (4, 64))
        sz = -(s * z.float()).to(dt)
        q = g.randint(0, 16, (64, 512)).numpy()
        W = O.dequant_weight(q, s, sz, 128)
        gi = torch.arange(512) // 128
        exact = torch.from_numpy(q).double() * s.double()[gi].t() + sz.double()[gi].t()
        assert torch.equal(W, exact.to(dt))
