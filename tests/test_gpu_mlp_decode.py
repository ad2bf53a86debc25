"""-m gpu: QuantLlamaMLP.forward at decode in ONE launch (awq_w4a16_mlp_decode_cdna4: gate/up + SiLU * mul + down_proj, the down blocks
gated on a device-side count of finished gate/up blocks) against the oracle's statement of the reference sequence
(tinychat/modules/fused_mlp.py:33-83) and against the two-launch product path; repeated calls with changing inputs check that no
call reads a previous call's activations (cross-XCD visibility of h) and that the counters return to zero."""
import pytest
import torch

from oracle import awq_oracle as O
from tests.helpers import make_case, Gen, assert_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from llm_awq_amd import ops as o
    return o


def _build(ops, H, F, NO, dtype, seed):
    from llm_awq_amd.fused_mlp import interleave_gate_up
    cg = make_case(F, H, dtype, seed=seed, M=8)
    cu = make_case(F, H, dtype, seed=seed + 1, M=8)
    cd = make_case(NO, F, dtype, seed=seed + 2, M=1, bias=True)
    qi, si, zi = interleave_gate_up(cg["qweight"].cuda(), cu["qweight"].cuda(), cg["scales"].cuda(), cu["scales"].cuda(),
                                    cg["scaled_zeros"].cuda(), cu["scaled_zeros"].cuda())
    gu = ops.repack_v2_to_cdna4(qi)
    gu_szh, e1 = ops.pack_szh_cdna4(si, zi, H)
    dq = ops.repack_v2_to_cdna4(cd["qweight"].cuda())
    d_szh, e2 = ops.pack_szh_cdna4(cd["scales"].cuda(), cd["scaled_zeros"].cuda(), F)
    assert e1 and e2
    return cg, cu, cd, gu, gu_szh, dq, d_szh


def _oracle(x, cg, cu, cd):
    g = O.wqlinear_forward(x, None, cg["scales"], cg["scaled_zeros"], None, 128, q_int=cg["q"])
    u = O.wqlinear_forward(x, None, cu["scales"], cu["scaled_zeros"], None, 128, q_int=cu["q"])
    h = torch.nn.functional.silu(g) * u
    return h, O.wqlinear_forward(h, None, cd["scales"], cd["scaled_zeros"], cd["bias"], 128, q_int=cd["q"])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("H,F,NO", [(4096, 2048, 1024), (4096, 14336, 4096)])
def test_one_launch_mlp_vs_oracle_and_two_launches(ops, dtype, H, F, NO):
    cg, cu, cd, gu, gu_szh, dq, d_szh = _build(ops, H, F, NO, dtype, 11 + F)
    ctr = torch.zeros(4096, dtype=torch.int32, device="cuda")
    bias = cd["bias"].cuda()
    g = Gen(F)
    for M in ((1, 2, 8) if F < 4096 else (1, 4)):
        for it in range(6):  # new activations every call: a stale h (or a missed wait) cannot pass
            x = g.randn(M, H).to(dtype)
            y = ops.mlp_decode_cdna4(x.cuda(), gu, gu_szh, dq, d_szh, ctr, bias)
            h2 = ops.decode_cdna4(x.cuda(), gu, gu_szh, None, 2)
            y2 = ops.decode_cdna4(h2, dq, d_szh, bias, 0)
            torch.cuda.synchronize()
            assert int(ctr.abs().sum()) == 0, ctr[:4].cpu().tolist()
            # against the two-launch product path: same arithmetic, another split of K over the waves of a block
            rel2 = ((y.float() - y2.float()).norm() / y2.float().norm()).item()
            assert rel2 < 2e-3, (M, it, rel2)
            assert_bits(y, y2, (0.1 if dtype == torch.bfloat16 else 0.2))  # last-bit differences of h pass through a 14336-term dot product
            if it == 0 and F < 4096:
                _h, ref = _oracle(x, cg, cu, cd)
                rel = ((y.cpu().float() - ref.float()).norm() / ref.float().norm()).item()
                assert rel < 4e-3, (M, rel)


def test_rows_that_do_not_fit_are_refused(ops):
    """8 rows of a 14336-wide h do not fit the down blocks' staging: the entry point says so and the module issues two launches"""
    from llm_awq_amd import _capi
    cg, cu, cd, gu, gu_szh, dq, d_szh = _build(ops, 4096, 14336, 4096, torch.bfloat16, 5)
    ctr = torch.zeros(4096, dtype=torch.int32, device="cuda")
    with pytest.raises(_capi.AwqNativeError):
        ops.mlp_decode_cdna4(cg["x"].cuda(), gu, gu_szh, dq, d_szh, ctr, None)


def test_module_decode_takes_the_one_launch_path(monkeypatch):
    import torch.nn as nn
    monkeypatch.setenv("AWQ_MLP_ONE_LAUNCH", "1")
    from llm_awq_amd.fused_mlp import QuantLlamaMLP
    from llm_awq_amd.qmodule import WQLinear
    dtype, H, F = torch.bfloat16, 4096, 2048
    cg = make_case(F, H, dtype, seed=3, M=4)
    cu = make_case(F, H, dtype, seed=4, M=4)
    cd = make_case(H, F, dtype, seed=5, M=1)

    def lin(c, k, n):
        m = WQLinear(4, 128, k, n, False, "cuda", dtype=dtype)
        m.load_state_dict(dict(qweight=c["qweight"], scales=c["scales"], scaled_zeros=c["scaled_zeros"]))
        return m

    mlp = QuantLlamaMLP(lin(cg, H, F), lin(cd, F, H), lin(cu, H, F))
    x = cg["x"]
    _h, ref = _oracle(x, cg, cu, cd)
    y = mlp(x.cuda())
    assert mlp._ctr is not None and int(mlp._ctr.abs().sum()) == 0, "the one-launch path ran and left its counters clean"
    assert ((y.cpu().float() - ref.float()).norm() / ref.float().norm()).item() < 4e-3
    y9 = mlp(torch.cat([x, x, x], 0)[:9].cuda())  # 9 rows: the two-kernel path
    assert ((y9[:4].float() - y.float()).norm() / y.float().norm()).item() < 4e-3
