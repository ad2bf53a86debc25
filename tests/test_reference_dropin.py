"""not-gpu, needs the reference mount (skipped on the GPU box, where /root/reference does not exist): the REFERENCE's own classes on THIS engine.

`install_as_awq_inference_engine()` puts the MI355X build under the module name the reference imports (awq/quantize/qmodule.py:4,
tinychat/modules/fused_mlp.py:8); then the reference's `awq.quantize.qmodule` is imported from /root/reference and compared with
`llm_awq_amd.qmodule` member by member: constructor signature, attributes, registered buffers (names, shapes, dtypes), state_dict keys in
both directions, `from_linear` buffers bit for bit, `extra_repr`, and the two entry points' arity as the reference's forward() calls them
(qmodule.py:206-220).  Nothing is copied from the mount; the test only imports it."""
import inspect
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "awq", "quantize")), reason="reference mount absent")


@pytest.fixture(scope="module")
def ref_qmodule():
    import llm_awq_amd
    eng = llm_awq_amd.install_as_awq_inference_engine()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import awq.quantize.qmodule as RQ  # (prints a harmless "VILA is not installed" notice)
    assert RQ.awq_inference_engine is eng, "the reference module did not bind the MI355X engine"
    return RQ, eng


CASES = [(4, 128, 768, 768, True, torch.float16), (4, 128, 4096, 6144, False, torch.bfloat16), (4, 128, 11008, 4096, False, torch.float16),
         (4, -1, 256, 64, True, torch.float16)]


@pytest.mark.parametrize("w_bit,group,K,N,bias,dtype", CASES)
def test_ctor_attributes_buffers_state_dict(ref_qmodule, w_bit, group, K, N, bias, dtype):
    RQ, _eng = ref_qmodule
    from llm_awq_amd import qmodule as MQ
    assert list(inspect.signature(RQ.WQLinear.__init__).parameters) == list(inspect.signature(MQ.WQLinear.__init__).parameters)
    assert (list(inspect.signature(RQ.WQLinear.from_linear).parameters) == list(inspect.signature(MQ.WQLinear.from_linear).parameters))
    r, m = RQ.WQLinear(w_bit, group, K, N, bias, "cpu", dtype=dtype), MQ.WQLinear(w_bit, group, K, N, bias, "cpu", dtype=dtype)
    for a in ("in_features", "out_features", "w_bit", "group_size", "split_k_iters", "interleave"):
        assert getattr(r, a) == getattr(m, a), a
    rb, mb = dict(r.named_buffers()), dict(m.named_buffers())
    assert list(rb) == list(mb)
    for k in rb:
        assert rb[k].shape == mb[k].shape and rb[k].dtype == mb[k].dtype, k
    assert (r.bias is None) == (m.bias is None)
    assert r.extra_repr() == m.extra_repr()
    # state dicts are interchangeable in both directions (the v2 checkpoint contract)
    sd = {k: torch.randint(-100, 100, v.shape).to(v.dtype) for k, v in r.state_dict().items()}
    m.load_state_dict(sd)
    r.load_state_dict(m.state_dict())
    for k, v in sd.items():
        assert torch.equal(r.state_dict()[k], v) and torch.equal(m.state_dict()[k], v)
    for fn in ("pack_intweight", "calculate_zeros_width", "make_divisible", "ScaledActivation"):
        assert hasattr(MQ, fn) and hasattr(RQ, fn)
    with pytest.raises(NotImplementedError):
        RQ.WQLinear(2, 128, K, N, bias, "cpu")
    with pytest.raises(NotImplementedError):
        MQ.WQLinear(2, 128, K, N, bias, "cpu")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_from_linear_buffers_bit_for_bit(ref_qmodule, dtype):
    RQ, _eng = ref_qmodule
    from llm_awq_amd import qmodule as MQ
    from oracle import awq_oracle as O
    from tests.helpers import Gen
    K, N = 768, 96
    g = Gen(5)
    w = (g.randn(N, K) * 0.02).to(dtype)
    fake, s, z = O.pseudo_quantize(w, 4, 128)
    lin = torch.nn.Linear(K, N, bias=True, dtype=dtype)
    with torch.no_grad():
        lin.weight.data = fake.clone()
        lin.bias.data = (g.randn(N) * 0.02).to(dtype)
    r = RQ.WQLinear.from_linear(lin, 4, 128, False, s.clone(), z.clone())
    m = MQ.WQLinear.from_linear(lin, 4, 128, False, s.clone(), z.clone())
    for k in ("qweight", "scales", "scaled_zeros", "bias"):
        assert torch.equal(getattr(r, k), getattr(m, k)), k


def test_entry_points_take_the_calls_the_reference_forward_makes(ref_qmodule):
    """qmodule.py:206-220: gemv_forward_cuda_new(x, qweight, scales, scaled_zeros, M, N, K, group_size) for fewer than 8 rows, else
    gemm_forward_cuda_new(x, qweight, scales, scaled_zeros); tinychat/modules/fused_mlp.py:48-77 makes the same two calls.  Here (no GPU):
    the bound functions exist under those names with that arity and reject CPU tensors loudly instead of computing anything."""
    RQ, eng = ref_qmodule
    doc_v, doc_m = eng.gemv_forward_cuda_new.__doc__, eng.gemm_forward_cuda_new.__doc__
    sig_v, sig_m = doc_v.split("->")[0], doc_m.split("->")[0]
    assert sig_v.count("torch.Tensor") == 4 and sig_v.count("SupportsInt") == 4 and sig_v.count("arg") == 8   # 4 tensors + m, n, k, group_size
    assert sig_m.count("torch.Tensor") == 4 and sig_m.count("arg") == 4
    r = RQ.WQLinear(4, 128, 256, 64, False, "cpu", dtype=torch.float16)
    for rows in (1, 7, 8, 33):  # both branches of the reference's dispatch reach this engine and are refused on CPU tensors
        with pytest.raises((RuntimeError, ValueError, TypeError)):
            r(torch.zeros(rows, 256, dtype=torch.float16))


def test_reference_quant_llama_mlp_binds_this_engine(ref_qmodule):
    """tinychat/modules/fused_mlp.py:8 imports awq_inference_engine at module import: with the MI355X build installed under that name the
    reference's QuantLlamaMLP class constructs from this repository's WQLinear modules and registers the same buffer names as
    llm_awq_amd.fused_mlp.QuantLlamaMLP."""
    _RQ, eng = ref_qmodule
    import importlib.util
    try:  # the FILE alone: the tinychat.modules package __init__ pulls flash_attn, which this image does not have
        spec = importlib.util.spec_from_file_location("_ref_fused_mlp", os.path.join(REF, "tinychat", "modules", "fused_mlp.py"))
        RF = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(RF)
    except Exception as e:  # noqa: BLE001
        pytest.skip(f"tinychat/modules/fused_mlp.py not importable here: {type(e).__name__}: {e}")
    assert RF.awq_inference_engine is eng
    from llm_awq_amd.fused_mlp import QuantLlamaMLP
    from llm_awq_amd.qmodule import WQLinear
    mk = lambda k, n: WQLinear(4, 128, k, n, False, "cpu", dtype=torch.float16)  # noqa: E731
    ref = RF.QuantLlamaMLP(mk(256, 512), mk(512, 256), mk(256, 512))
    ours = QuantLlamaMLP(mk(256, 512), mk(512, 256), mk(256, 512))
    assert set(dict(ref.named_buffers())) == set(dict(ours.named_buffers()))
    for a in ("in_features", "intermediate_size", "out_features", "w_bit", "split_k_iters"):
        assert getattr(ref, a) == getattr(ours, a), a
