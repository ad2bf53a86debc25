"""Streaming checkpoint loader: AWQ checkpoint on disk -> this build's buffers on the GPU, one linear at a time.

What it replaces in the reference:
  * tinychat/utils/load_quant.py:27-58  `mem_efficient_load_checkpoint` -- a folder with one `<key>.pt` file per tensor
    (written by tinychat/split_ckpt.py:8-21), loaded file by file;
  * tinychat/utils/load_quant.py:61-97   `load_awq_model` -- a single `.pt` state dict through accelerate;
  * tinychat/offline-weight-repacker.py:111-152 -- the offline v1 -> v2 conversion, which here happens on the fly.
It also reads `.safetensors` files and Hugging Face sharded folders (`*.safetensors` / `*.bin` + `*.index.json`), the
formats `examples/convert_to_hf.py:44-50` produces.

Per packed linear (`<p>.qweight` + `<p>.scales` + `<p>.qzeros` | `<p>.scaled_zeros` [+ `<p>.bias`]) the loader
  1. converts v1 -> v2 when the checkpoint is v1 (HIP kernel, bit-exact against the reference's golden vectors),
  2. cuts the rank's tensor-parallel shard out of the v2 buffers (`tp_plan`: module prefix -> "column" | "row" |
     "stacked:<parts>" | "replicate"; cuts are whole 16-row slabs / whole 128-k groups, llm_awq_amd/parallel.py),
  3. interleaves the shard to the cdna4 layout when it is eligible (N % 16 == 0, K % 128 == 0) and the target asks
     for it, and emits the same keys / shapes / dtypes as a v2 checkpoint plus the one-byte `<p>.qweight_layout` marker
     `WQLinear` reads at load time.
Only one linear's tensors are resident on the host at a time (safetensors and `torch.load(mmap=True)` map the file).
The arithmetic is the product's HIP kernels (`repacker.GpuKernels`); there is no CPU fallback (tests inject the oracle).
"""
from __future__ import annotations

import json
import os
import re
from collections import OrderedDict
from typing import Callable, Dict, Iterable, Iterator, List, Optional, Tuple

import torch

from . import parallel as P
from .repacker import GpuKernels, cdna4_eligible


# ------------------------------------------------------------------------------------------------
# readers: key -> tensor, lazily
# ------------------------------------------------------------------------------------------------
class CheckpointReader:
    """Uniform lazy view of a checkpoint: `keys()` in file order, `get(key)` -> CPU tensor."""

    def __init__(self, path: str):
        self.path = path
        self._index: "OrderedDict[str, Tuple[str, str]]" = OrderedDict()  # key -> (kind, file)
        self._open_pt: Dict[str, Dict[str, torch.Tensor]] = {}
        self._open_st: Dict[str, object] = {}
        if os.path.isdir(path):
            self._scan_dir(path)
        else:
            self._scan_file(path)
        if not self._index:
            raise FileNotFoundError(f"no checkpoint tensors found under {path}")

    # -- discovery --
    def _scan_file(self, f: str):
        if f.endswith(".safetensors"):
            for k in self._st(f).keys():
                self._index[k] = ("st", f)
        elif f.endswith((".pt", ".bin", ".pth")):
            for k in self._pt(f).keys():
                self._index[k] = ("pt", f)
        else:
            raise ValueError(f"unsupported checkpoint file {f} (want .pt / .bin / .pth / .safetensors)")

    def _scan_dir(self, d: str):
        names = sorted(os.listdir(d))
        idx = [n for n in names if n.endswith(".index.json")]
        if idx:  # Hugging Face sharded checkpoint: weight_map says which file holds which key
            with open(os.path.join(d, idx[0])) as fh:
                wm = json.load(fh)["weight_map"]
            for k, fn in wm.items():
                self._index[k] = ("st" if fn.endswith(".safetensors") else "pt", os.path.join(d, fn))
            return
        st = [n for n in names if n.endswith(".safetensors")]
        pt = [n for n in names if n.endswith((".pt", ".bin", ".pth"))]
        if st:
            for n in st:
                self._scan_file(os.path.join(d, n))
        elif pt:
            # split_ckpt.py layout: one file per tensor, named <key>.pt and holding {key: tensor} -- trust the name, do
            # not open thousands of files up front; anything else: read its keys
            for n in pt:
                key = re.sub(r"\.(pt|bin|pth)$", "", n)
                f = os.path.join(d, n)
                if len(pt) > 8 and "." in key:
                    self._index[key] = ("pt1", f)
                else:
                    self._scan_file(f)

    # -- file handles --
    def _pt(self, f: str) -> Dict[str, torch.Tensor]:
        if f not in self._open_pt:
            try:
                sd = torch.load(f, map_location="cpu", mmap=True, weights_only=True)
            except (RuntimeError, TypeError, ValueError):  # legacy (non-zip) pickles cannot be mapped
                sd = torch.load(f, map_location="cpu", weights_only=True)
            if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
                sd = sd["state_dict"]
            self._open_pt[f] = sd
        return self._open_pt[f]

    def _st(self, f: str):
        if f not in self._open_st:
            from safetensors import safe_open

            self._open_st[f] = safe_open(f, framework="pt", device="cpu")
        return self._open_st[f]

    # -- access --
    def keys(self) -> List[str]:
        return list(self._index)

    def __contains__(self, key: str) -> bool:
        return key in self._index

    def get(self, key: str) -> torch.Tensor:
        kind, f = self._index[key]
        if kind == "st":
            return self._st(f).get_tensor(key)
        if kind == "pt1":
            sd = torch.load(f, map_location="cpu", weights_only=True)
            return sd[key] if key in sd else next(iter(sd.values()))
        return self._pt(f)[key]

    def close(self):
        self._open_pt.clear()
        self._open_st.clear()


# ------------------------------------------------------------------------------------------------
# tensor-parallel plans
# ------------------------------------------------------------------------------------------------
def llama_tp_plan(prefix: str) -> str:
    """Megatron pairing for Llama-style blocks (DESIGN.md "(f) Multi-GPU"): q/k/v/gate/up N-sharded, o/down K-sharded;
    tinychat's fused buffers (`qkv_proj`, fused_attn.py:566-572; a stacked `gate_up_proj`) shard per projection."""
    leaf = prefix.rsplit(".", 1)[-1]
    if leaf in ("q_proj", "k_proj", "v_proj", "gate_proj", "up_proj", "w1", "w3"):
        return "column"
    if leaf in ("o_proj", "down_proj", "w2", "out_proj", "dense_4h_to_h"):
        return "row"
    if leaf == "gate_up_proj":
        return "stacked:2"
    return "replicate"  # unknown packed linears (lm_head, fused qkv with unequal sections, ...) stay whole on every rank


def _shard(mode: str, qw, s, z, bias, world: int, rank: int):
    """v2 buffers of one linear -> this rank's v2 buffers."""
    if world == 1 or mode == "replicate":
        return qw, s, z, bias
    if mode == "column":
        qw, s, z, (n0, n1) = P.shard_column_parallel(qw, s, z, world, rank)
        return qw, s, z, (bias[n0:n1].contiguous() if bias is not None else None)
    if mode == "row":
        qw, s, z, _ = P.shard_row_parallel(qw, s, z, world, rank)
        return qw, s, z, (bias if rank == 0 or bias is None else torch.zeros_like(bias))  # added once after the all-reduce
    if mode.startswith("stacked:"):
        parts = int(mode.split(":")[1])
        qw, s, z, bounds = P.shard_stacked_column_parallel(qw, s, z, world, rank, parts=parts)
        if bias is not None:
            bias = torch.cat([bias[lo:hi] for (lo, hi) in bounds]).contiguous()
        return qw, s, z, bias
    raise ValueError(f"unknown tensor-parallel mode {mode!r}")


# ------------------------------------------------------------------------------------------------
# the loader
# ------------------------------------------------------------------------------------------------
def iter_quantized(path: str, target: str = "cdna4", device: str = "cuda", tp_rank: int = 0, tp_world: int = 1,
                   tp_plan: Optional[Callable[[str], str]] = None, kernels=None,
                   log: Optional[Callable[[str], None]] = None) -> Iterator[Tuple[str, torch.Tensor]]:
    """Yield (key, tensor on `device`) for every tensor of the checkpoint, packed linears converted as described in the
    module docstring.  Keys come out in checkpoint order with each linear's tensors together."""
    assert target in ("v2", "cdna4")
    assert 0 <= tp_rank < tp_world
    reader = CheckpointReader(path)
    kernels = kernels or GpuKernels(device)
    plan = tp_plan or llama_tp_plan
    log = log or (lambda s: None)
    keys = reader.keys()
    is_v1 = any(k.endswith("qzeros") for k in keys)
    dev = torch.device(device)
    consumed = set()
    try:
        for key in keys:
            if key in consumed or key.endswith(".qweight_layout"):
                continue
            if not key.endswith("qweight"):
                if key.endswith(("scales", "qzeros", "scaled_zeros")) and key.rsplit(".", 1)[0] + ".qweight" in reader:
                    continue  # emitted with their qweight
                if key.endswith(".bias") and key[:-5] + ".qweight" in reader:
                    continue
                yield key, reader.get(key).to(dev)
                continue
            prefix = key[: -len(".qweight")]
            lay = prefix + ".qweight_layout"
            if lay in reader and int(reader.get(lay)) == 1:
                raise ValueError(f"{key} is already cdna4-interleaved: shard and repack from the v2 checkpoint instead")
            qw, s = reader.get(key), reader.get(prefix + ".scales")
            bias = reader.get(prefix + ".bias") if (prefix + ".bias") in reader else None
            if is_v1:
                log(f"v1 -> v2: {prefix}")
                qw, s, z = kernels.v1_to_v2(qw, s, reader.get(prefix + ".qzeros"))
            else:
                z = reader.get(prefix + ".scaled_zeros")
            mode = plan(prefix)
            qw, s, z, bias = _shard(mode, qw, s, z, bias, tp_world, tp_rank)
            marker = None
            if target == "cdna4" and cdna4_eligible(qw, s):
                log(f"cdna4 interleave: {prefix} ({mode}, rank {tp_rank}/{tp_world})")
                qw = kernels.v2_to_cdna4(qw)
                marker = torch.tensor(1, dtype=torch.uint8)
            yield key, qw.to(dev)
            if marker is not None:
                yield lay, marker
            yield prefix + ".scales", s.to(dev)
            yield prefix + ".scaled_zeros", z.to(dev)
            if bias is not None:
                yield prefix + ".bias", bias.to(dev)
            consumed.update((key, prefix + ".scales", prefix + ".qzeros", prefix + ".scaled_zeros", prefix + ".bias"))
    finally:
        reader.close()


def load_quantized_state_dict(path: str, **kw) -> "OrderedDict[str, torch.Tensor]":
    """`iter_quantized` collected into a state dict (device tensors): what `model.load_state_dict` / `WQLinear` buffers
    take.  For models too large to hold twice, consume `iter_quantized` and assign buffers as they arrive."""
    return OrderedDict(iter_quantized(path, **kw))


def load_into(model: torch.nn.Module, path: str, strict: bool = False, **kw) -> List[str]:
    """Assign the checkpoint into `model` tensor by tensor (buffers of WQLinear modules are re-assigned, so sharded shapes and
    the layout marker take effect -- the reference overwrites them the same way, fused_attn.py:581-588).  Returns the keys
    the model had no slot for."""
    from .qmodule import WQLinear

    mods = dict(model.named_modules())
    params = dict(model.named_parameters())
    bufs = dict(model.named_buffers())
    missing = []
    for key, t in iter_quantized(path, **kw):
        prefix, _, leaf = key.rpartition(".")
        m = mods.get(prefix)
        if isinstance(m, WQLinear) and leaf in ("qweight", "scales", "scaled_zeros", "bias", "qweight_layout"):
            if leaf == "qweight_layout":
                m.layout = "cdna4" if int(t) == 1 else "v2"
            else:
                setattr(m, leaf, t)
                if leaf == "qweight":  # a marker, if any, follows; the packed {scale | zero} side buffer is rebuilt lazily
                    m.out_features, m.in_features = t.shape[0] * 4, t.shape[1]
                    m.layout, m.sz_cdna4 = "v2", None
            continue
        if key in params:
            params[key].data = t.to(params[key].dtype)
        elif key in bufs:
            bufs[key].data = t
        else:
            missing.append(key)
            if strict:
                raise KeyError(f"{key}: no such parameter or buffer in the model")
    return missing
