"""W3A16 (BASELINE.json config 3, Llama-2-7B shapes): the reference only defines the 3-bit GRID
(pseudo_quantize_tensor, awq/quantize/quantizer.py:61-103, pinned by tests/golden/pseudo_w3.npz); the packed
"w3c" format is this repository's.  not-gpu: oracle grid vs golden, oracle pack <-> unpack, product host pack vs
oracle, WQLinear(w_bit=3) contract.  -m gpu: HIP pack/unpack/dequant bit exact, GEMV / expand+GEMM vs oracle."""
import numpy as np
import pytest
import torch

from oracle import awq_oracle as O
from tests.conftest import as_t
from tests.helpers import check_forward, Gen, cuda_gen, assert_bits


def make_case_w3(N, K, seed=0, M=1, bias=False, dtype=torch.bfloat16):
    g = Gen(seed)
    d = O.quantize_linear_w3(g.randn(N, K) * 0.02, dtype=dtype)
    d["x"] = g.randn(M, K).to(dtype)
    d["bias"] = (g.randn(N) * 0.02).to(dtype) if bias else None
    d["q"] = d["intweight"].numpy().astype(np.uint8)
    return d


def test_grid_matches_reference_golden(golden):
    g = golden("pseudo_w3.npz")
    for name, dt in [("f16", torch.float16), ("bf16", torch.bfloat16), ("f32", torch.float32)]:
        cv = (lambda a: as_t(a, dt)) if dt != torch.float32 else torch.from_numpy
        fake, s, z = O.pseudo_quantize(cv(g[name + "_w0"]), 3, 128)
        assert torch.equal(fake, cv(g[name + "_wfake"])) and torch.equal(s, cv(g[name + "_s"])) and torch.equal(z, cv(g[name + "_z"]))
        assert z.max() <= 7 and z.min() >= 0


@pytest.mark.parametrize("N,K", [(16, 128), (32, 256), (48, 1280), (64, 11008)])
def test_pack_unpack_oracle_and_host(N, K):
    from llm_awq_amd import qmodule as Q
    rng = np.random.default_rng(N + K)
    q = rng.integers(0, 8, size=(N, K)).astype(np.uint8)
    p = O.pack_w3(q)
    assert p.shape == (N // 4, 3 * K // 4) and p.dtype == np.int16 and p.nbytes == N * K * 3 // 8
    assert (O.unpack_w3(p) == q).all()
    assert (Q.pack_w3c(torch.from_numpy(q.astype(np.int32))).numpy() == p).all()
    # a single flipped integer changes exactly the words of its own lane
    q2 = q.copy()
    q2[N // 2, K // 3] ^= 5
    diff = (O.pack_w3(q2).view(np.uint32) != p.view(np.uint32))
    assert 1 <= diff.sum() <= 2


def test_wqlinear_w3_contract():
    from llm_awq_amd import qmodule as Q
    m = Q.WQLinear(3, 128, 11008, 4096, True, "cpu", dtype=torch.bfloat16)
    sd = m.state_dict()
    assert list(sd) == ["qweight", "scales", "scaled_zeros", "bias"]
    assert sd["qweight"].shape == (1024, 8256) and sd["qweight"].dtype == torch.int16
    assert sd["scales"].shape == (88, 4096) and m.layout == "w3c" and m.w_bit == 3
    assert Q.WQLinear(3, 128, 256, 64, False, "cpu", dtype=torch.float16).layout == "w3c"  # fp16 models too
    with pytest.raises(NotImplementedError):
        Q.WQLinear(3, 128, 256, 64, False, "cpu", dtype=torch.float32)
    with pytest.raises(NotImplementedError):
        Q.WQLinear(2, 128, 256, 64, False, "cpu", dtype=torch.bfloat16)
    d = make_case_w3(32, 256, seed=3)
    lin = torch.nn.Linear(256, 32, bias=False).to(torch.bfloat16)
    lin.weight.data = d["w_fake"]
    q = Q.WQLinear.from_linear(lin, 3, 128, False, d["s"], d["z"])
    assert torch.equal(q.qweight, d["qweight"]) and torch.equal(q.scaled_zeros, d["scaled_zeros"])
    with pytest.raises(RuntimeError):
        q(torch.zeros(1, 256, dtype=torch.bfloat16))  # no CPU fallback


def test_w3_dequant_equals_fake_weight():
    """dequant(from_linear(fake)) == fake to <= 1 ulp: the 'real == fake quantisation' equivalence the README evaluates."""
    d = make_case_w3(64, 512, seed=11)
    W = O.dequant_weight(d["q"], d["scales"], d["scaled_zeros"], 128)
    # (q - z) * s vs q * s + round_T(-s * z): they differ by the rounding of the folded zero point (<= 1/2 ulp of
    # s*z, z <= 7) plus the final rounding -- an ABSOLUTE error of the order of ulp_T(7 s), not a relative one
    err = (W.float() - d["w_fake"].float()).abs()
    assert err.max() <= 2.0 ** -7 * 7 * d["scales"].float().max()


# ---------------------------------------------- GPU ----------------------------------------------
@pytest.fixture(scope="module")
def ops():
    from llm_awq_amd import ops as _ops
    _ops._capi.lib()
    return _ops


@pytest.mark.gpu
@pytest.mark.parametrize("N,K", [(16, 128), (64, 768), (256, 1280), (4096, 11008)])
def test_gpu_pack_unpack_bit_exact(ops, N, K):
    rng = np.random.default_rng(N * 3 + K)
    q = rng.integers(0, 8, size=(N, K)).astype(np.uint8)
    p = ops.pack_w3(torch.from_numpy(q).cuda())
    if N * K <= 1 << 20:
        assert (p.cpu().numpy() == O.pack_w3(q)).all()
    assert (ops.unpack_w3(p).cpu().numpy() == q).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,K", [(16, 128), (64, 768), (256, 1280)])
def test_gpu_dequant_bit_exact(ops, N, K, dtype):
    d = make_case_w3(N, K, seed=N + K, dtype=dtype)
    W = O.dequant_weight(d["q"], d["scales"], d["scaled_zeros"], 128)
    got = ops.dequant_w3(d["qweight"].cuda(), d["scales"].cuda(), d["scaled_zeros"].cuda()).cpu()
    assert torch.equal(got.view(torch.int16), W.view(torch.int16))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 2, 4, 7, 8, 9, 64, 200, 300, 777])
@pytest.mark.parametrize("N,K", [(768, 768), (256, 4096), (64, 11008), (1040, 1280)])
def test_gpu_forward_vs_oracle(ops, M, N, K, dtype):
    """every row count: decode GEMV (<= 8), the masked single row tile (9..255), full row tiles with and without split-K -- all of
    them read the 768-byte tiles natively (no expanded copy: the workspace is the optional split-K scratch only)."""
    d = make_case_w3(N, K, seed=M * 13 + N + K, M=M, bias=(M in (4, 64, 300)), dtype=dtype)
    szp = ops.pack_sz_cdna4(d["scales"].cuda(), d["scaled_zeros"].cuda(), K)
    y = ops.forward_w3(d["x"].cuda(), d["qweight"].cuda(), d["scales"].cuda(), d["scaled_zeros"].cuda(), szp,
                       d["bias"].cuda() if d["bias"] is not None else None).cpu()
    check_forward(y, d["x"], d["q"], d["scales"], d["scaled_zeros"], dtype, bias=d["bias"])
    L = ops._capi.lib()
    # the workspace is the OPTIONAL split-K scratch now (no N*K/2 expanded copy): decode and chip-filling prompts need none
    assert L.awq_w3a16_forward_workspace_bytes(8, N, K) == 0 and L.awq_w3a16_forward_workspace_bytes(2048, 22016, 4096) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gpu_w3_prefill_full_shape_both_tile_widths(ops, dtype):
    """Llama-2-7B's stacked gate/up (4096 -> 22016) at M = 2048 and 300: 256-wide tiles for the full rounds + 128-wide for the
    rest, against fp32 torch on the weights of dequant_w3 (bit exact vs the oracle: test_gpu_dequant_bit_exact and
    tests/test_gpu_oracle_fullsize.py)."""
    K, N = 4096, 22016
    g = cuda_gen(5)
    q = torch.randint(0, 8, (N, K), dtype=torch.uint8, device="cuda", generator=g)
    qw = ops.pack_w3(q)
    s = ((5.2 + 0.8 * torch.rand(K // 128, N, device="cuda", generator=g)) * 0.02 / 7).to(dtype)
    z = -(s * torch.randint(2, 6, (K // 128, N), device="cuda", generator=g).float()).to(dtype)
    szp = ops.pack_sz_cdna4(s, z, K)
    W = ops.dequant_w3(qw, s, z).float()
    for M in (2048, 300):
        x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
        y = ops.forward_w3(x, qw, s, z, szp)
        ref = (x.float() @ W.t()).to(dtype)  # fp32 accumulate, ONE rounding to T: the rounding is on both sides
        rel = ((y.float() - ref.float()).norm() / ref.float().norm()).item()
        assert rel <= 1e-3, (M, rel)         # BASELINE.json's tolerance
        assert_bits(ref, y, 0.03)


@pytest.mark.gpu
def test_gpu_wqlinear_w3_module(ops):
    """WQLinear(w_bit=3) end to end on Llama-2-7B's down_proj shape: module forward == C-ABI forward, and the
    fake-vs-real equivalence against F.linear on the fake-quantised weight."""
    from llm_awq_amd.qmodule import WQLinear
    K, N = 11008, 4096
    g = cuda_gen(0)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    # the grid on the GPU with torch ops (same formulae as quantizer.py:61-103)
    grp = w.reshape(-1, 128)
    hi, lo = grp.amax(1, keepdim=True), grp.amin(1, keepdim=True)
    s = (hi - lo).clamp(min=1e-5) / 7
    z = (-torch.round(lo / s)).clamp_(0, 7)
    fake = ((torch.clamp(torch.round(grp / s) + z, 0, 7) - z) * s).reshape(N, K)
    lin = torch.nn.Linear(K, N, bias=False, device="cuda", dtype=torch.bfloat16)
    lin.weight.data = fake
    m = WQLinear.from_linear(lin, 3, 128, False, s.view(N, -1), z.view(N, -1))
    assert m.qweight.shape == (N // 4, 3 * K // 4)
    W = ops.dequant_w3(m.qweight, m.scales, m.scaled_zeros)
    assert (W.float() - fake.float()).abs().max() <= 2.0 ** -7 * 7 * m.scales.float().max()
    for M in (1, 7, 64):
        x = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
        y = m(x)
        ref = (x.float() @ W.float().t()).to(torch.bfloat16)  # one rounding to T on both sides
        rel = ((y.float() - ref.float()).norm() / ref.float().norm()).item()
        assert rel <= 1e-3, (M, rel)
        assert_bits(ref, y, 0.03)


# ---- tensor-parallel sharding of a 3-bit layer (BASELINE config 3 x config 4) ----

@pytest.mark.parametrize("world", [2, 3])
def test_shard_w3c_is_packing_the_sliced_integers(world):
    """the w3c buffer is a [N/16, K/128] grid of self-contained tiles: a shard cut on that grid equals packing the sliced integers"""
    from llm_awq_amd import parallel as P, qmodule as Q
    d = make_case_w3(80, 640, seed=5)
    q = torch.from_numpy(d["q"].astype(np.int32))
    covered = {"row": 0, "column": 0}
    for mode in ("row", "column"):
        for r in range(world):
            qw, s, z, (lo, hi) = P.shard_w3c(d["qweight"], d["scales"], d["scaled_zeros"], mode, world, r)
            covered[mode] += hi - lo
            if mode == "row":
                assert lo % 128 == 0 and torch.equal(qw, Q.pack_w3c(q[:, lo:hi].contiguous()))
                g0, g1 = lo // 128, hi // 128
                assert torch.equal(s[: g1 - g0], d["scales"][g0:g1]) and torch.equal(z[: g1 - g0], d["scaled_zeros"][g0:g1])
                assert s.shape[0] % 8 == 0 and not s[g1 - g0:].any()
            else:
                assert lo % 16 == 0 and torch.equal(qw, Q.pack_w3c(q[lo:hi].contiguous()))
                assert torch.equal(s, d["scales"][:, lo:hi]) and torch.equal(z, d["scaled_zeros"][:, lo:hi])
    assert covered == {"row": 640, "column": 80}


def test_tp_wqlinear_takes_a_w3_module_offline():
    from llm_awq_amd import parallel as P, qmodule as Q
    d = make_case_w3(64, 512, seed=6, bias=True)
    full = Q.WQLinear(3, 128, 512, 64, True, "cpu", dtype=torch.bfloat16)
    full.qweight, full.scales, full.scaled_zeros, full.bias = d["qweight"], d["scales"], d["scaled_zeros"], d["bias"]
    row = P.TPWQLinear(full, "row", world=2, rank=1)
    assert row.shard.w_bit == 3 and row.shard.layout == "w3c" and (row.shard.in_features, row.shard.out_features) == (256, 64)
    assert row.bounds == (256, 512) and row.bias is full.bias
    col = P.TPWQLinear(full, "column", world=4, rank=2)
    assert (col.shard.in_features, col.shard.out_features) == (512, 16) and torch.equal(col.bias, d["bias"][32:48])


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,K", [(64, 512), (256, 3584), (1024, 11008)])
def test_gpu_w3_partial_is_the_unrounded_fp32_product(ops, dtype, N, K):
    """awq_w3a16_partial: the register-ring decode kernel (m <= 8), the masked narrow tile and the prefill tiles write their fp32 sums"""
    d = make_case_w3(N, K, seed=N + K, M=600, dtype=dtype)
    qw, s, z = d["qweight"].cuda(), d["scales"].cuda(), d["scaled_zeros"].cuda()
    szp = ops.pack_sz_cdna4(s, z, K)
    W = ops.dequant_w3(qw, s, z).double()
    for M in (1, 3, 4, 5, 8, 9, 40, 255, 256, 300, 600):
        x = d["x"][:M].contiguous().cuda()
        y = ops.partial_w3(x, qw, szp)
        assert y.dtype == torch.float32 and y.shape == (M, N)
        y64 = x.double() @ W.t()
        S = x.double().abs() @ W.abs().t()
        worst = ((y.double() - y64).abs() / (3e-6 * S + 1e-30)).max().item()
        assert worst <= 1.0, (M, worst)
        assert_bits(y.to(dtype), ops.forward_w3(x, qw, s, z, szp), 0.02, "T(partial) vs the T kernel")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("world", [2, 4, 8])
def test_gpu_w3_row_and_column_split_vs_the_single_device_oracle(ops, dtype, world):
    """every rank's shard run one after the other on the one device: the row split's fp32 partials summed and rounded once, the column
    split's outputs concatenated -- both against O.wqlinear_forward on the unsharded layer"""
    from llm_awq_amd import parallel as P, qmodule as Q
    K, N = 4096, 512
    d = make_case_w3(N, K, seed=31 + world, M=300, bias=True, dtype=dtype)
    full = Q.WQLinear(3, 128, K, N, True, "cuda", dtype=dtype)
    full.qweight, full.scales, full.scaled_zeros, full.bias = d["qweight"].cuda(), d["scales"].cuda(), d["scaled_zeros"].cuda(), d["bias"].cuda()
    rows = [P.TPWQLinear(full, "row", world=world, rank=r) for r in range(world)]
    cols = [P.TPWQLinear(full, "column", world=world, rank=r) for r in range(world)]
    assert all(t.shard.layout == "w3c" and t._reducer is None for t in rows + cols)
    for M in (1, 7, 64, 300):
        x = d["x"][:M].contiguous()
        ref = O.wqlinear_forward(x, None, d["scales"], d["scaled_zeros"], d["bias"], 128, q_int=d["q"]).float()
        xg = x.cuda()
        acc = torch.zeros(M, N, device="cuda")
        for t in rows:
            p32 = t.partial(xg)
            assert p32.dtype == torch.float32
            acc += p32
        y = ops.round_bias_f32(acc, dtype, full.bias).float().cpu()
        rel = ((y - ref).norm() / ref.norm()).item()
        assert rel < 1e-3, (M, rel)
        assert_bits(y, ref, 0.01, "W3 row split vs single-device oracle")
        yc = torch.cat([t(xg) for t in cols], dim=-1).cpu()
        check_forward(yc, x, d["q"], d["scales"], d["scaled_zeros"], dtype, bias=d["bias"])
        # a world-1 "split" is the module itself
    one = P.TPWQLinear(full, "row", world=1, rank=0)
    xg = d["x"][:5].contiguous().cuda()
    assert_bits(one(xg), full(xg), 0.02, "world-1 row shard vs the module")


# ---- QuantLlamaMLP on 3-bit projections (tinychat/modules/fused_mlp.py:33-83 with w_bit = 3): awq_w3a16_mlp_gate_up_forward ----

def _w3_pair(F, K, dtype, seed, M):
    cg = make_case_w3(F, K, seed=seed, M=M, dtype=dtype)
    cu = make_case_w3(F, K, seed=seed + 1, M=M, dtype=dtype)
    x = cg["x"]
    g = O.wqlinear_forward(x, None, cg["scales"], cg["scaled_zeros"], None, 128, q_int=cg["q"])
    u = O.wqlinear_forward(x, None, cu["scales"], cu["scaled_zeros"], None, 128, q_int=cu["q"])
    return cg, cu, x, torch.nn.functional.silu(g) * u, g, u


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 4, 5, 8, 9, 64, 300, 2048])
@pytest.mark.parametrize("F,K", [(256, 768), (1376, 512), (11008, 4096)])
def test_gpu_w3_gate_up_entry_every_row_count(ops, dtype, M, F, K):
    """<= 8 rows: the register-ring decode kernel pairs gate row r with up row r of a slab in its epilogue; more rows: the prefill tiles'
    fused tail on the w3c stream.  Checker: the oracle's statement of the reference sequence (two forwards, F.silu, multiply, all rounded to T)."""
    from llm_awq_amd.fused_mlp import deinterleave_gate_up_w3, interleave_gate_up_w3
    from tests.helpers import acc_slack, check_fused_tail, weight_row_norms
    if F >= 4096 and (M not in (1, 9, 300) or dtype != torch.bfloat16):
        pytest.skip("full-size case: bf16, M = 1, 9, 300 only")
    if M == 2048 and F != 1376:
        pytest.skip("M = 2048 on the (1376, 512) pair only")
    cg, cu, x, ref, gt, up = _w3_pair(F, K, dtype, F + K + M, M)
    dev = [t.cuda() for t in (cg["qweight"], cu["qweight"], cg["scales"], cu["scales"], cg["scaled_zeros"], cu["scaled_zeros"])]
    qi, si, zi = interleave_gate_up_w3(*dev)
    assert qi.shape == (2 * F // 4, K * 3 // 4) and si.shape[1] == 2 * F
    # the stream holds slab j = gate rows 8j..8j+7, then up rows 8j..8j+7, and the inverse gives the projections back bit for bit
    qint = ops.unpack_w3(qi).cpu().view(F // 8, 2, 8, K)
    assert (qint[:, 0].reshape(F, K).numpy() == cg["q"]).all() and (qint[:, 1].reshape(F, K).numpy() == cu["q"]).all()
    for a, b in zip(deinterleave_gate_up_w3(qi, si, zi), dev):
        assert torch.equal(a, b)
    szp = ops.pack_sz_cdna4(si, zi, K)
    y = ops.mlp_gate_up_forward_w3(x.cuda(), qi, szp).cpu()
    assert y.shape == (M, F)
    check_fused_tail(y, gt, up, 1e-3, what=f"W3 gate_up entry F={F} K={K} M={M}", slack_g=acc_slack(x, weight_row_norms(cg)),
                     slack_u=acc_slack(x, weight_row_norms(cu)))
    assert_bits(y, ref, (0.05 if M <= 300 else 0.07))
    # the fused tail == the unfused product path on the same interleaved stream
    full = ops.forward_w3(x.cuda(), qi, si, zi, szp).view(M, F // 8, 2, 8)
    unfused = (torch.nn.functional.silu(full[:, :, 0, :]) * full[:, :, 1, :]).reshape(M, F).cpu()
    assert_bits(y, unfused, (0.002 if dtype == torch.bfloat16 else 0.01))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gpu_w3_quant_llama_mlp_module(ops, dtype):
    """QuantLlamaMLP built from three WQLinear(w_bit = 3) modules through make_fused_mlp: reference-named state dict, every row count, down_proj behind it"""
    import torch.nn as nn
    from llm_awq_amd import qmodule as Q
    from llm_awq_amd.fused_mlp import QuantLlamaMLP, make_fused_mlp
    H, F = 1024, 2816
    cg, cu, x, act, _gt, _up = _w3_pair(F, H, dtype, 177, 40)
    cd = make_case_w3(H, F, seed=179, M=1, dtype=dtype)

    def lin(c, k, n):
        m = Q.WQLinear(3, 128, k, n, False, "cuda", dtype=dtype)
        m.load_state_dict(dict(qweight=c["qweight"], scales=c["scales"], scaled_zeros=c["scaled_zeros"]))
        return m

    class LlamaMLP(nn.Module):
        def __init__(self):
            super().__init__()
            self.gate_proj, self.up_proj, self.down_proj = lin(cg, H, F), lin(cu, H, F), lin(cd, F, H)

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.mlp = LlamaMLP()

    blk = make_fused_mlp(Block())
    mlp = blk.mlp
    assert isinstance(mlp, QuantLlamaMLP) and mlp.w_bit == 3
    for M in (1, 5, 8, 9, 40):
        xm = x[:M].contiguous()
        a = mlp.our_llama_mlp(xm.cuda()).cpu()
        assert_bits(a, act[:M], 0.05)
        y = mlp(xm.cuda()).cpu()
        check_forward(y, a, cd["q"], cd["scales"], cd["scaled_zeros"], dtype)
    # the six reference-named buffers were released once the fused stream existed; state_dict() rebuilds them bit for bit
    sd = mlp.state_dict()
    for name, c in (("gate_proj", cg), ("up_proj", cu)):
        assert torch.equal(sd[name + "_qweight"].cpu(), c["qweight"]) and torch.equal(sd[name + "_scales"].cpu(), c["scales"])
        assert torch.equal(sd[name + "_scaled_zeros"].cpu(), c["scaled_zeros"])
    mlp.load_state_dict(sd)
    assert_bits(mlp.our_llama_mlp(x[:9].contiguous().cuda()).cpu(), act[:9], 0.05)
    assert mlp(x[:12].view(2, 6, H).cuda()).shape == (2, 6, H)
    # mixed bit widths are refused
    with pytest.raises(ValueError):
        QuantLlamaMLP(Q.WQLinear(4, 128, H, F, False, "cuda", dtype=dtype), lin(cd, F, H), lin(cu, H, F))
