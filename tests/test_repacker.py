"""The rewritten offline repacker (llm_awq_amd/repacker.py vs tinychat/offline-weight-repacker.py:111-152):
key handling on CPU with the oracle injected as the kernel set (not-gpu), and the real HIP kernels through the
C ABI against the reference's golden v1 -> v2 vectors (-m gpu)."""
import numpy as np
import pytest
import torch

from llm_awq_amd import repacker as R
from oracle import awq_oracle as O
from tests.conftest import as_t


class OracleKernels:
    """test-only stand-in for GpuKernels: the CPU oracle."""

    def v1_to_v2(self, qw1, s1, qz1):
        return O.repack_v1_to_v2(qw1.numpy(), s1, qz1.numpy())

    def v2_to_cdna4(self, qw2):
        return torch.from_numpy(O.v2_to_cdna4(qw2.numpy()))


def _v1_checkpoint(golden):
    g = golden("repack_v1_v2.npz")
    sd, want = {}, {}
    for idx, prefix in [(0, "model.layers.0.self_attn.q_proj"), (1, "model.layers.0.mlp.down_proj")]:
        dt = torch.float16 if int(g[f"dtype_{idx}"][0]) == 0 else torch.bfloat16
        sd[prefix + ".qweight"] = torch.from_numpy(g[f"qw1_{idx}"])
        sd[prefix + ".qzeros"] = torch.from_numpy(g[f"qz1_{idx}"])
        sd[prefix + ".scales"] = as_t(g[f"sc1_{idx}"], dt)
        want[prefix] = (g[f"qw2_{idx}"], as_t(g[f"sc2_{idx}"], dt), as_t(g[f"sz2_{idx}"], dt), g[f"unpacked_{idx}"])
    sd["model.norm.weight"] = torch.ones(8)
    return sd, want


def _check_v2(out, want):
    for prefix, (qw2, s2, sz2, _q) in want.items():
        assert (out[prefix + ".qweight"].numpy() == qw2).all()
        assert torch.equal(out[prefix + ".scales"], s2) and torch.equal(out[prefix + ".scaled_zeros"], sz2)
        assert prefix + ".qzeros" not in out
    assert torch.equal(out["model.norm.weight"], torch.ones(8))


def test_v1_to_v2_keys_and_values(golden):
    sd, want = _v1_checkpoint(golden)
    out = R.repack_state_dict(sd, target="v2", kernels=OracleKernels())
    _check_v2(out, want)
    assert not any("layout" in k for k in out)
    # idempotent on v2 input
    again = R.repack_state_dict(out, target="v2", kernels=OracleKernels())
    assert list(again) == list(out) and all(torch.equal(again[k], out[k]) for k in out)


def test_v1_to_cdna4_marks_only_eligible_layers(golden):
    sd, want = _v1_checkpoint(golden)
    out = R.repack_state_dict(sd, target="cdna4", kernels=OracleKernels())
    for prefix, (qw2, s2, _sz2, q) in want.items():
        n, k = q.shape
        ok = n % 16 == 0 and k % 128 == 0  # bf16 and fp16 alike
        assert ((prefix + ".qweight_layout") in out) == ok
        got = out[prefix + ".qweight"].numpy()
        assert got.shape == qw2.shape and got.dtype == qw2.dtype  # same contract either way
        assert (O.unpack_cdna4(got) == q).all() if ok else (got == qw2).all()


def test_cdna4_checkpoint_loads_into_wqlinear():
    from llm_awq_amd.qmodule import WQLinear
    N, K = 32, 256
    rng = np.random.default_rng(0)
    q = rng.integers(0, 16, size=(N, K)).astype(np.uint8)
    sd = {"fc.qweight": torch.from_numpy(O.pack_v2(q)), "fc.scales": torch.rand(8, N).bfloat16(),
          "fc.scaled_zeros": -torch.rand(8, N).bfloat16()}
    out = R.repack_state_dict(sd, target="cdna4", kernels=OracleKernels())
    assert int(out["fc.qweight_layout"]) == 1
    with pytest.raises(ValueError):  # already interleaved
        R.repack_state_dict(out, target="cdna4", kernels=OracleKernels())
    m = torch.nn.Module()
    m.fc = WQLinear(4, 128, K, N, False, "cpu", dtype=torch.bfloat16)
    m.load_state_dict(out)  # strict: the marker key is consumed by WQLinear._load_from_state_dict
    assert m.fc.layout == "cdna4" and (O.unpack_cdna4(m.fc.qweight.numpy()) == q).all()
    m.load_state_dict(sd)
    assert m.fc.layout == "v2"
    assert "fc.qweight_layout" not in m.state_dict()


def test_cli_roundtrip(tmp_path, golden, monkeypatch):
    sd, want = _v1_checkpoint(golden)
    src, dst = tmp_path / "in.pt", tmp_path / "out-v2.pt"
    torch.save(sd, src)
    monkeypatch.setattr(R, "GpuKernels", lambda device="cuda": OracleKernels())
    R.main(["--input", str(src), "--output", str(dst), "--target", "v2"])
    _check_v2(torch.load(dst), want)


@pytest.mark.gpu
def test_gpu_kernels_match_reference_golden(golden):
    sd, want = _v1_checkpoint(golden)
    out = R.repack_state_dict(sd, target="v2", device="cuda")
    _check_v2(out, want)
    out4 = R.repack_state_dict(sd, target="cdna4", device="cuda")
    ref4 = R.repack_state_dict(sd, target="cdna4", kernels=OracleKernels())
    assert list(out4) == list(ref4) and all(torch.equal(out4[k], ref4[k]) for k in ref4)
