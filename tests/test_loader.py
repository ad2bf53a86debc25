"""Streaming loader (llm_awq_amd/loader.py vs tinychat/utils/load_quant.py:27-97 + tinychat/split_ckpt.py): every on-disk
format, v1 and v2 inputs, tensor-parallel slicing and the cdna4 interleave, with the oracle injected as the kernel set on
CPU; the real HIP kernels under -m gpu."""
import json
import os

import numpy as np
import pytest
import torch

from llm_awq_amd import loader as L
from llm_awq_amd import parallel as P
from oracle import awq_oracle as O
from tests.helpers import make_case
from tests.test_repacker import OracleKernels, _v1_checkpoint, _check_v2

H, F = 512, 1024
LAYERS = {
    "model.layers.0.self_attn.q_proj": (H, H, False),
    "model.layers.0.self_attn.o_proj": (H, H, True),
    "model.layers.0.mlp.gate_proj": (F, H, False),
    "model.layers.0.mlp.up_proj": (F, H, False),
    "model.layers.0.mlp.down_proj": (H, F, False),
    "model.layers.0.mlp.gate_up_proj": (2 * F, H, True),   # tinychat-style stacked buffer
    "lm_head": (264, H, False),                           # N % 16 != 0: stays v2, replicated
}


def _v2_state_dict(dtype=torch.bfloat16):
    sd, cases = {}, {}
    for i, (p, (n, k, bias)) in enumerate(LAYERS.items()):
        c = make_case(n, k, dtype, seed=100 + i, M=2, bias=bias)
        sd[p + ".qweight"], sd[p + ".scales"], sd[p + ".scaled_zeros"] = c["qweight"], c["scales"], c["scaled_zeros"]
        if bias:
            sd[p + ".bias"] = c["bias"]
        cases[p] = c
    sd["model.norm.weight"] = torch.arange(H, dtype=torch.float32).to(dtype)
    sd["model.embed_tokens.weight"] = torch.randn(32, H).to(dtype)
    return sd, cases


def _write(fmt, sd, d):
    from safetensors.torch import save_file

    if fmt == "pt":
        f = os.path.join(d, "model-v2.pt")
        torch.save(sd, f)
        return f
    if fmt == "safetensors":
        f = os.path.join(d, "model.safetensors")
        save_file({k: v.contiguous() for k, v in sd.items()}, f)
        return f
    if fmt == "split":  # tinychat/split_ckpt.py: one {key: tensor} file per key
        out = os.path.join(d, "split")
        os.makedirs(out)
        for k, v in sd.items():
            torch.save({k: v}, os.path.join(out, k + ".pt"))
        return out
    if fmt == "hf_sharded":
        out = os.path.join(d, "hf")
        os.makedirs(out)
        keys = list(sd)
        parts = [keys[0::2], keys[1::2]]
        wm = {}
        for i, ks in enumerate(parts):
            fn = f"model-{i + 1:05d}-of-00002.safetensors"
            save_file({k: sd[k].contiguous() for k in ks}, os.path.join(out, fn))
            wm.update({k: fn for k in ks})
        with open(os.path.join(out, "model.safetensors.index.json"), "w") as fh:
            json.dump({"metadata": {}, "weight_map": wm}, fh)
        return out
    raise AssertionError(fmt)


@pytest.mark.parametrize("fmt", ["pt", "safetensors", "split", "hf_sharded"])
def test_formats_roundtrip_to_v2(tmp_path, fmt):
    sd, _ = _v2_state_dict()
    path = _write(fmt, sd, str(tmp_path))
    out = L.load_quantized_state_dict(path, target="v2", device="cpu", kernels=OracleKernels())
    assert set(out) == set(sd)
    for k in sd:
        assert torch.equal(out[k], sd[k]), k


def test_cdna4_target_marks_and_interleaves(tmp_path):
    sd, _ = _v2_state_dict()
    path = _write("safetensors", sd, str(tmp_path))
    out = L.load_quantized_state_dict(path, target="cdna4", device="cpu", kernels=OracleKernels())
    for p, (n, k, _b) in LAYERS.items():
        eligible = n % 16 == 0 and k % 128 == 0
        assert ((p + ".qweight_layout") in out) == eligible
        want = O.v2_to_cdna4(sd[p + ".qweight"].numpy()) if eligible else sd[p + ".qweight"].numpy()
        assert (out[p + ".qweight"].numpy() == want).all(), p
        assert torch.equal(out[p + ".scales"], sd[p + ".scales"])
    # fp16 checkpoints take the same interleave (the matrix-core dequant has an fp16 form: offset 1024)
    sd16, _ = _v2_state_dict(torch.float16)
    out16 = L.load_quantized_state_dict(_write("pt", sd16, str(tmp_path)), target="cdna4", device="cpu", kernels=OracleKernels())
    assert sum(k.endswith("qweight_layout") for k in out16) == sum(n % 16 == 0 and k % 128 == 0 for (n, k, _b) in LAYERS.values())


def test_v1_checkpoint_on_the_fly(tmp_path, golden):
    sd, want = _v1_checkpoint(golden)
    path = _write("split", sd, str(tmp_path))
    out = L.load_quantized_state_dict(path, target="v2", device="cpu", kernels=OracleKernels())
    _check_v2(out, want)


@pytest.mark.parametrize("world", [2, 4])
def test_tensor_parallel_shards_reassemble(tmp_path, world):
    """Every rank loads only its shard; the shards equal parallel.shard_* of the full v2 buffers, and the oracle matmul over
    the shards reproduces the unsharded result (column: concatenate, row: sum)."""
    sd, cases = _v2_state_dict()
    path = _write("hf_sharded", sd, str(tmp_path))
    outs = [L.load_quantized_state_dict(path, target="v2", device="cpu", kernels=OracleKernels(), tp_rank=r, tp_world=world)
            for r in range(world)]
    for p, (n, k, has_bias) in LAYERS.items():
        c = cases[p]
        mode = L.llama_tp_plan(p)
        full = O.wqlinear_forward(c["x"], c["qweight"], c["scales"], c["scaled_zeros"], None, 128).float()
        ys = []
        for r, o in enumerate(outs):
            qw, s, z = o[p + ".qweight"], o[p + ".scales"], o[p + ".scaled_zeros"]
            if mode == "column":
                eq, es, ez, _ = P.shard_column_parallel(c["qweight"], c["scales"], c["scaled_zeros"], world, r)
                xr = c["x"]
            elif mode == "row":
                eq, es, ez, (k0, k1) = P.shard_row_parallel(c["qweight"], c["scales"], c["scaled_zeros"], world, r)
                xr = c["x"][:, k0:k1].contiguous()
            elif mode == "stacked:2":
                eq, es, ez, _ = P.shard_stacked_column_parallel(c["qweight"], c["scales"], c["scaled_zeros"], world, r, parts=2)
                xr = c["x"]
            else:
                eq, es, ez, xr = c["qweight"], c["scales"], c["scaled_zeros"], c["x"]
            assert torch.equal(qw, eq) and torch.equal(s, es) and torch.equal(z, ez), (p, r)
            ys.append(O.wqlinear_forward(xr, qw, s, z, None, 128).float())
        if mode == "row":
            got = sum(ys)
            assert ((got - full).norm() / full.norm()).item() < 6e-3  # every partial is rounded to bf16 before the sum (as in TP)
            if has_bias:  # the bias survives on exactly one rank
                assert sum(int(o[p + ".bias"].abs().sum() > 0) for o in outs) == 1
        elif mode == "column":
            assert torch.equal(torch.cat(ys, 1), full)
        elif mode == "stacked:2":
            half = full.shape[1] // 2
            g = torch.cat([y[:, : y.shape[1] // 2] for y in ys], 1)
            u = torch.cat([y[:, y.shape[1] // 2:] for y in ys], 1)
            assert torch.equal(g, full[:, :half]) and torch.equal(u, full[:, half:])
            if has_bias:
                b = torch.cat([o[p + ".bias"][: o[p + ".bias"].numel() // 2] for o in outs])
                assert torch.equal(b, c["bias"][:half])
        else:
            assert all(torch.equal(y, full) for y in ys)


def test_load_into_wqlinear_module(tmp_path):
    from llm_awq_amd.qmodule import WQLinear

    sd, cases = _v2_state_dict()
    path = _write("pt", sd, str(tmp_path))

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj = WQLinear(4, 128, H, H, False, "cpu", dtype=torch.bfloat16)
            self.down_proj = WQLinear(4, 128, F, H, False, "cpu", dtype=torch.bfloat16)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = torch.nn.Module()
            self.model.layers = torch.nn.ModuleList([torch.nn.Module()])
            blk = Block()
            self.model.layers[0].self_attn = torch.nn.Module()
            self.model.layers[0].self_attn.q_proj = blk.q_proj
            self.model.layers[0].mlp = torch.nn.Module()
            self.model.layers[0].mlp.down_proj = blk.down_proj
            self.model.norm = torch.nn.LayerNorm(H, bias=False)

    net = Net()
    missing = L.load_into(net, path, target="cdna4", device="cpu", kernels=OracleKernels(), tp_rank=1, tp_world=2)
    q, d = net.model.layers[0].self_attn.q_proj, net.model.layers[0].mlp.down_proj
    assert q.layout == "cdna4" and q.out_features == H // 2 and q.in_features == H
    assert d.layout == "cdna4" and d.in_features == F // 2 and d.out_features == H
    assert torch.equal(net.model.norm.weight.data, sd["model.norm.weight"].float())
    assert "model.embed_tokens.weight" in missing and "lm_head.qweight" in missing


@pytest.mark.gpu
def test_gpu_loader_end_to_end(tmp_path):
    """real kernels: load rank shards as cdna4 on the GPU and run them through WQLinear against the oracle."""
    from llm_awq_amd import ops

    sd, cases = _v2_state_dict()
    path = _write("safetensors", sd, str(tmp_path))
    world = 2
    p = "model.layers.0.mlp.down_proj"
    c = cases[p]
    full = O.wqlinear_forward(c["x"], c["qweight"], c["scales"], c["scaled_zeros"], None, 128).float()
    acc = 0
    for r in range(world):
        o = L.load_quantized_state_dict(path, target="cdna4", device="cuda", tp_rank=r, tp_world=world)
        assert int(o[p + ".qweight_layout"]) == 1 and o[p + ".qweight"].is_cuda
        k0, k1 = P.shard_bounds(F, world, r, 128)
        szp = ops.pack_sz_cdna4(o[p + ".scales"], o[p + ".scaled_zeros"], k1 - k0)
        y = ops.gemm_cdna4(c["x"][:, k0:k1].contiguous().cuda(), o[p + ".qweight"], o[p + ".scales"], o[p + ".scaled_zeros"], None, szp)
        acc = acc + y.float().cpu()
    assert ((acc - full).norm() / full.norm()).item() < 6e-3
