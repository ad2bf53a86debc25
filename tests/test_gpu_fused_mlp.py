"""-m gpu: QuantLlamaMLP / make_fused_mlp (tinychat/modules/fused_mlp.py:11-101) and the C-ABI entry behind it,
`awq_w4a16_mlp_gate_up_forward_cdna4`, for every row count: <= 8 rows the streaming decode launch, more rows the prefill tile
kernels with the SiLU * mul tail fused into their epilogue.  The checker is the oracle's statement of the reference sequence:
two WQLinear forwards, F.silu, multiply, every op rounded to T (fused_mlp.py:36-83), then down_proj."""
import pytest
import torch

from oracle import awq_oracle as O
from tests.helpers import acc_slack, check_forward, check_fused_tail, make_case, assert_bits, record_rel, weight_row_norms, _dequant_f64

pytestmark = pytest.mark.gpu

# norm-wise distance of the fused tail from the oracle's tail: BASELINE.json's 1e-3 (measured on MI355X: <= 3.7e-4 over every case of the suite,
# profiles/r05_test_stats.txt -- 2.7 x below it); the HARD criterion is check_fused_tail's elementwise hull
REL_TAIL = {torch.bfloat16: 1e-3, torch.float16: 1e-3}


def _pair(F, K, dtype, seed, M):
    cg = make_case(F, K, dtype, seed=seed, M=M)
    cu = make_case(F, K, dtype, seed=seed + 1, M=M)
    x = cg["x"]
    g = O.wqlinear_forward(x, None, cg["scales"], cg["scaled_zeros"], None, 128, q_int=cg["q"])
    u = O.wqlinear_forward(x, None, cu["scales"], cu["scaled_zeros"], None, 128, q_int=cu["q"])
    return cg, cu, x, torch.nn.functional.silu(g) * u, g, u


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 7, 8, 9, 16, 17, 64, 300, 2048])
@pytest.mark.parametrize("F,K", [(256, 768), (1376, 512), (2048, 2048), (14336, 4096)])
def test_gate_up_entry_every_row_count(dtype, M, F, K):
    from llm_awq_amd import ops
    from llm_awq_amd.fused_mlp import interleave_gate_up
    if F >= 4096 and (M not in (1, 9, 300) or dtype != torch.bfloat16):
        pytest.skip("full-size case: bf16, M = 1, 9, 300 only (2048 rows of this shape against the CPU oracle: tests/test_gpu_oracle_fullsize.py, the plain linear)")
    if M == 2048 and F != 2048:
        pytest.skip("M = 2048 on the (2048, 2048) pair only")
    cg, cu, x, ref, gt, up = _pair(F, K, dtype, F + K + M, M)
    qi, si, zi = interleave_gate_up(cg["qweight"].cuda(), cu["qweight"].cuda(), cg["scales"].cuda(), cu["scales"].cuda(),
                                    cg["scaled_zeros"].cuda(), cu["scaled_zeros"].cuda())
    c4 = ops.repack_v2_to_cdna4(qi)
    szp = ops.pack_sz_cdna4(si, zi, K)
    szh, exact = ops.pack_szh_cdna4(si, zi, K)
    assert exact
    ys = [ops.mlp_gate_up_forward_cdna4(x.cuda(), c4, szp, szh).cpu(), ops.mlp_gate_up_forward_cdna4(x.cuda(), c4, szp, None).cpu()]
    for y in ys:
        assert y.shape == (M, F)
        check_fused_tail(y, gt, up, REL_TAIL[dtype], what=f"gate_up entry F={F} K={K} M={M}", slack_g=acc_slack(x, weight_row_norms(cg)),
                         slack_u=acc_slack(x, weight_row_norms(cu)))
        assert_bits(y, ref, (0.05 if M <= 300 else 0.07))
    assert_bits(ys[0], ys[1], 0.01)
    # the fused tail == the unfused product path on the same interleaved stream: GEMM, de-interleave, F.silu * up, all in T
    full = ops.gemm_cdna4(x.cuda(), c4, si, zi, None, szp)
    full = full.view(M, F // 8, 2, 8)
    unfused = (torch.nn.functional.silu(full[:, :, 0, :]) * full[:, :, 1, :]).reshape(M, F).cpu()
    if M > 8:
        # same accumulators, same roundings; only the fp32 silu is another implementation (hardware exp2 / rcp vs torch's kernel):
        # a few ulp of fp32, visible in ~0.2 % of the fp16 roundings
        assert_bits(ys[1], unfused, (0.001 if dtype == torch.bfloat16 else 0.005))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_module_matches_reference_sequence(dtype):
    """QuantLlamaMLP built from three WQLinear modules (the reference's constructor), through make_fused_mlp."""
    import torch.nn as nn
    from llm_awq_amd.fused_mlp import QuantLlamaMLP, make_fused_mlp
    from llm_awq_amd.qmodule import WQLinear
    H, F = 1024, 2816
    cg, cu, x, act, _gt, _up = _pair(F, H, dtype, 77, 40)
    cd = make_case(H, F, dtype, seed=79, M=1)

    def lin(c, k, n):
        m = WQLinear(4, 128, k, n, False, "cuda", dtype=dtype)
        m.load_state_dict(dict(qweight=c["qweight"], scales=c["scales"], scaled_zeros=c["scaled_zeros"]))
        return m

    class LlamaMLP(nn.Module):  # the class name make_fused_mlp keys on (fused_mlp.py:87)
        def __init__(self):
            super().__init__()
            self.gate_proj, self.up_proj, self.down_proj = lin(cg, H, F), lin(cu, H, F), lin(cd, F, H)

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.mlp = LlamaMLP()

    blk = make_fused_mlp(Block())
    assert isinstance(blk.mlp, QuantLlamaMLP)
    sd = blk.mlp.state_dict()
    assert {"gate_proj_qweight", "gate_proj_scales", "gate_proj_scaled_zeros", "up_proj_qweight", "up_proj_scales",
            "up_proj_scaled_zeros"} <= set(sd)  # fused_mlp.py:19-27
    for M in (1, 5, 8, 9, 40):
        xm = x[:M].contiguous()
        a = blk.mlp.our_llama_mlp(xm.cuda()).cpu()
        assert_bits(a, act[:M], 0.05)
        y = blk.mlp(xm.cuda()).cpu()
        # down_proj on the module's own activations must satisfy the forward bound; against the oracle's activations the few
        # last-bit differences of `a` pass through a 2816-term dot product
        check_forward(y, a, cd["q"], cd["scales"], cd["scaled_zeros"], dtype)
        # against the oracle run on the ORACLE's activations: y - ref = (y - W a) + W (a - act); the first term is held to 1e-3 above, the second is
        # the module's few last-bit differences of `a` through down_proj -- computed, not guessed
        ref = O.wqlinear_forward(act[:M], None, cd["scales"], cd["scaled_zeros"], None, 128, q_int=cd["q"])
        Wd = _dequant_f64(cd["q"], cd["scales"], cd["scaled_zeros"])
        through = ((a.double() - act[:M].double()) @ Wd.t()).norm().item() / ref.double().norm().item()
        rel = ((y.double() - ref.double()).norm() / ref.double().norm()).item()
        record_rel(f"module down_proj M={M}", rel, 1e-3 + through + 1.2e-3)
        assert rel <= 1e-3 + through + 1.2e-3, (rel, through)  # (+ the T rounding of y(a) vs y(act) where they differ: one ulp rms ~1.1e-3)
    # 3-D input [batch, seq, hidden] like the model passes
    y3 = blk.mlp(x[:12].view(2, 6, H).cuda())
    assert y3.shape == (2, 6, H)


def test_entry_rejects_bad_arguments():
    from llm_awq_amd import ops, _capi
    c = make_case(64, 256, torch.bfloat16, seed=1, M=4)
    c4 = ops.repack_v2_to_cdna4(c["qweight"].cuda())
    szp = ops.pack_sz_cdna4(c["scales"].cuda(), c["scaled_zeros"].cuda(), 256)
    with pytest.raises(_capi.AwqNativeError):
        ops.mlp_gate_up_forward_cdna4(c["x"].cuda(), c4, szp, None, group_size=64)
    with pytest.raises((TypeError, _capi.AwqNativeError)):
        ops.mlp_gate_up_forward_cdna4(c["x"].cuda().float(), c4, szp, None)


def _fused_block(cg, cu, cd, H, F, dtype):
    import torch.nn as nn
    from llm_awq_amd.fused_mlp import make_fused_mlp
    from llm_awq_amd.qmodule import WQLinear

    def lin(c, k, n):
        m = WQLinear(4, 128, k, n, False, "cuda", dtype=dtype)
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

    return make_fused_mlp(Block()).mlp


def _granule_h(state, F, dtype):
    """the activations the one-launch kernel handed over: data halves of the {2 x T, tag} granules behind the counter block of its state"""
    from llm_awq_amd import ops
    g = state[ops._capi.AWQ_MLP_DECODE_COUNTER_BYTES // 4:][: F].view(F // 2, 2)
    return g[:, 0].contiguous().view(torch.int16).view(dtype).reshape(1, F).cpu(), g[:, 1].cpu()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_v2_gate_up_buffers_are_released_and_state_dict_round_trips(dtype):
    """fused_mlp.py:19-27 registers the six v2 gate / up buffers: here they are released once the fused cdna4 stream exists (no second
    copy of two thirds of the block's weights) and state_dict() / load_state_dict() still speak the reference's keys bit-exactly."""
    import torch.nn as nn
    from llm_awq_amd.fused_mlp import QuantLlamaMLP
    from llm_awq_amd.qmodule import WQLinear
    H, F = 1024, 1408
    cg, cu, x, act, _gt, _up = _pair(F, H, dtype, 31, 12)
    cd = make_case(H, F, dtype, seed=33, M=1)

    def lin(c, k, n):
        m = WQLinear(4, 128, k, n, False, "cuda", dtype=dtype)
        m.load_state_dict(dict(qweight=c["qweight"], scales=c["scales"], scaled_zeros=c["scaled_zeros"]))
        return m

    mlp = QuantLlamaMLP(lin(cg, H, F), lin(cd, F, H), lin(cu, H, F))
    before = {k: v.clone() for k, v in mlp.state_dict().items()}
    y0 = mlp(x.cuda()).cpu()
    held = sum(getattr(mlp, n).numel() for n in ("gate_proj_qweight", "up_proj_qweight", "gate_proj_scales", "up_proj_scales",
                                                 "gate_proj_scaled_zeros", "up_proj_scaled_zeros"))
    assert held == 0, "the v2 copies are gone once the fused stream is built"
    after = mlp.state_dict()
    assert set(before) <= set(after)  # (+ down_proj's layout marker: QuantLlamaMLP moved it to the cdna4 interleave)
    for k in before:
        if not k.startswith("down_proj."):
            assert after[k].shape == before[k].shape and torch.equal(after[k].cpu(), before[k].cpu()), k
    # load into a fresh module, and back into the one whose buffers were released
    fresh = QuantLlamaMLP(lin(cu, H, F), lin(cd, F, H), lin(cg, H, F))  # (gate / up swapped on purpose: the load must overwrite them)
    fresh(x.cuda())
    fresh.load_state_dict(after)
    assert torch.equal(fresh(x.cuda()).cpu(), y0)
    mlp.load_state_dict(before)
    assert torch.equal(mlp(x.cuda()).cpu(), y0)


@pytest.mark.parametrize("F,K", [(14336, 4096), (7168, 2048)])
def test_gate_up_short_prompts_on_the_skinny_kernel_wide_shapes(F, K):
    """9..64 rows of the fused pair run on the skinny kernel's paired epilogue; at Llama-3-8B's width (2 F / 16 = 1792 slabs) in blocks of SEVEN slabs, below
    that of four.  The CPU oracle at this size is held by test_gate_up_entry_every_row_count (M = 9); here every row-count class against (a) the unfused
    product path on the same interleaved stream (same accumulators, same roundings: only the fp32 silu differs in its last bits) and (b) fp32 torch on the
    device-dequantised weights (bit-pinned to the oracle in test_gpu_oracle_fullsize.py), and the masked-tile route the knob mlp_skinny_max = 8 selects."""
    from llm_awq_amd import ops, synth
    from llm_awq_amd.fused_mlp import interleave_gate_up
    dtype = torch.bfloat16
    g, u = (synth.random_wq(K, F, dtype=dtype, seed=s, keep_q=False) for s in (21, 22))
    qi, si, zi = interleave_gate_up(g["qweight"], u["qweight"], g["scales"], u["scales"], g["scaled_zeros"], u["scaled_zeros"])
    c4 = ops.repack_v2_to_cdna4(qi)
    szp = ops.pack_sz_cdna4(si, zi, K)
    Wg, Wu = (ops.dequant_v2(w["qweight"], w["scales"], w["scaled_zeros"]).float() for w in (g, u))
    gen = torch.Generator(device="cuda").manual_seed(5)
    for M in (9, 16, 17, 33, 48, 64):
        x = torch.randn(M, K, device="cuda", generator=gen).to(dtype)
        y = ops.mlp_gate_up_forward_cdna4(x, c4, szp, None)
        full = ops.gemm_cdna4(x, c4, si, zi, None, szp).view(M, F // 8, 2, 8)
        unfused = (torch.nn.functional.silu(full[:, :, 0, :]) * full[:, :, 1, :]).reshape(M, F)
        assert_bits(y, unfused, 0.001, what=f"fused vs unfused skinny M={M}")
        gt, up = (x.float() @ Wg.t()).to(dtype), (x.float() @ Wu.t()).to(dtype)
        ref = (torch.nn.functional.silu(gt) * up).float()
        rel = ((y.float() - ref).norm() / ref.norm()).item()
        assert rel < 2e-3, (M, rel)  # (two T roundings of fp32 sums taken in another order than torch's: the hull model lives in the oracle tests)
        ops._capi.tune(mlp_skinny_max=8)
        try:
            y_tile = ops.mlp_gate_up_forward_cdna4(x, c4, szp, None)
        finally:
            ops._capi.tune(mlp_skinny_max=64)
        assert_bits(y, y_tile, 0.02, what=f"skinny vs masked tile M={M}")
