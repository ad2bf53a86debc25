"""Grouped (per-expert) WQLinear for MoE layers -- BASELINE.json config 5 (Mixtral-8x7B W4A16 g128).

The reference has no MoE path, so this is new capability built from the same contract:
every expert is a `WQLinear` whose buffers are stacked along a leading expert dimension
    qweight int16 [E, N/4, K]   scales / scaled_zeros T [E, Gpad, N]
and one launch runs all experts over tokens sorted by expert (`awq_w4a16_moe_gemm`).  Routing (top-k, sort,
un-sort, weighting) is ordinary torch glue around the path; it never synchronises the host: the per-expert row
ranges go to the kernel as a device int32 [E + 1] offsets array.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import load_engine
from .qmodule import WQLinear


def sort_by_expert(topk_ids: torch.Tensor, num_experts: int):
    """topk_ids [T, k] -> (order [T*k] token-slot indices sorted by expert, stable; offsets int32 [E + 1])."""
    flat = topk_ids.reshape(-1)
    order = torch.argsort(flat, stable=True)
    counts = torch.bincount(flat, minlength=num_experts)
    offsets = torch.zeros(num_experts + 1, dtype=torch.int32, device=flat.device)
    offsets[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return order, offsets


class _LayoutMarker:
    """state_dict contract of the stacked modules: `qweight` holds the reference (v2) interleave unless the dict carries `qweight_layout` = 1
    (the cdna4 interleave, as WQLinear's native checkpoints do) -- a module that was converted on its first GPU forward saves the marker, and a
    fresh module that loads such a dict takes the layout over instead of permuting the bytes a second time."""

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self.layout == "cdna4":
            destination[prefix + "qweight_layout"] = torch.tensor(1, dtype=torch.uint8)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        marker = state_dict.pop(prefix + "qweight_layout", None)
        had_qweight = (prefix + "qweight") in state_dict
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        # the layout follows the qweight bytes: a partial load (strict=False, a dict that only carries scales) leaves them -- and the layout
        # they are in -- untouched; resetting it would have an already converted module permute its bytes a second time
        if had_qweight:
            self.layout = "cdna4" if (marker is not None and int(marker) == 1) else "v2"
        self.sz_cdna4 = self.szh_cdna4 = None  # rebuilt from the (possibly new) scales at the next forward


def _stack_side_buffers(scales, scaled_zeros, in_features, pack_sz, pack_szh):
    """stacked sz_packed [E, N/16, K/128, 16] and, when EVERY expert's scales are f16-exact, the stacked sz_half (else None)"""
    E = scales.shape[0]
    szp = torch.stack([pack_sz(scales[e].contiguous(), scaled_zeros[e].contiguous(), in_features) for e in range(E)])
    hs = []
    for e in range(E):
        h, exact = pack_szh(scales[e].contiguous(), scaled_zeros[e].contiguous(), in_features)
        if not exact:
            return szp, None
        hs.append(h)
    return szp, torch.stack(hs)


class GroupedWQLinear(_LayoutMarker, nn.Module):
    """E stacked WQLinear experts with one grouped launch on the HIP engine (no other arithmetic path: the CPU tests of the routing glue
    subclass this in tests/helpers.py and override `forward` with the oracle)."""

    def __init__(self, experts: Sequence[WQLinear]):
        super().__init__()
        e0 = experts[0]
        assert all(e.w_bit == 4 and e.layout == e0.layout and e.bias is None for e in experts)
        assert all((e.in_features, e.out_features) == (e0.in_features, e0.out_features) for e in experts)
        self.num_experts, self.in_features, self.out_features = len(experts), e0.in_features, e0.out_features
        self.layout = e0.layout
        self.register_buffer("qweight", torch.stack([e.qweight for e in experts]).contiguous())
        self.register_buffer("scales", torch.stack([e.scales for e in experts]).contiguous())
        self.register_buffer("scaled_zeros", torch.stack([e.scaled_zeros for e in experts]).contiguous())
        self.sz_cdna4 = None  # stacked packed scales int32 [E, N/16, K/128, 16], built lazily for the cdna4 layout
        self.szh_cdna4 = None  # stacked sz_half (f16-mantissa dequant of the grouped tile launch) when every expert's scales are f16-exact

    @torch.no_grad()
    def to_cdna4(self):
        """Permute every expert's qweight into the cdna4 interleave (bf16; what the repacker emits) -- idempotent."""
        if self.layout != "cdna4":
            eng = load_engine()
            self.qweight = torch.stack([eng.repack_v2_to_cdna4(self.qweight[e].contiguous()) for e in range(self.num_experts)])
            self.layout = "cdna4"
        return self

    @torch.no_grad()
    def forward(self, x_sorted: torch.Tensor, expert_offsets: torch.Tensor) -> torch.Tensor:
        eng = load_engine()
        if self.layout == "cdna4":
            if self.sz_cdna4 is None or self.sz_cdna4.device != self.scales.device:
                # (building sz_half reads an exactness flag back per expert: once per module, not inside a graph capture)
                self.sz_cdna4, self.szh_cdna4 = _stack_side_buffers(self.scales, self.scaled_zeros, self.in_features, eng.pack_sz_cdna4, eng.pack_szh_cdna4)
            from . import ops
            return ops.moe_forward_cdna4(x_sorted.contiguous(), self.qweight, self.scales, self.scaled_zeros, self.sz_cdna4, expert_offsets,
                                         sz_half=getattr(self, "szh_cdna4", None))
        return eng.moe_gemm_forward(x_sorted.contiguous(), self.qweight, self.scales, self.scaled_zeros, expert_offsets, False)


class GroupedGateUp(_LayoutMarker, nn.Module):
    """The expert MLP's first half as ONE grouped launch: every expert's w1 (gate) and w3 (up) WQLinear stacked with their rows interleaved
    8 + 8 per 16-row slab -- the layout `llm_awq_amd.fused_mlp.QuantLlamaMLP` gives the dense pair (tinychat/modules/fused_mlp.py:36-83) --
    so that `silu(w1 x) * (w3 x)` is the epilogue of the grouped tile (`awq_w4a16_moe_mlp_gate_up_cdna4`): the [T, 2F] intermediate is never
    written and the block's w1 / w3 / F.silu * mul launches become one.  Built from v2 (reference layout) experts."""

    def __init__(self, w1: Sequence[WQLinear], w3: Sequence[WQLinear]):
        super().__init__()
        from .fused_mlp import interleave_gate_up
        assert len(w1) == len(w3) and len(w1) >= 1
        e0 = w1[0]
        assert all(e.w_bit == 4 and e.layout == "v2" and e.bias is None and e.group_size == 128 for e in list(w1) + list(w3)), \
            "GroupedGateUp is built from v2 (reference layout) 4-bit experts without bias, group size 128"
        assert all((e.in_features, e.out_features) == (e0.in_features, e0.out_features) for e in list(w1) + list(w3))
        assert e0.out_features % 8 == 0
        self.num_experts, self.in_features, self.out_features = len(w1), e0.in_features, e0.out_features
        qs, ss, zs = [], [], []
        for g, u in zip(w1, w3):
            q, s, z = interleave_gate_up(g.qweight, u.qweight, g.scales, u.scales, g.scaled_zeros, u.scaled_zeros)
            qs.append(q)
            ss.append(s)
            zs.append(z)
        self.register_buffer("qweight", torch.stack(qs).contiguous())        # int16 [E, 2F/4, K], v2 interleave until the first GPU forward
        self.register_buffer("scales", torch.stack(ss).contiguous())          # T [E, Gpad, 2F]
        self.register_buffer("scaled_zeros", torch.stack(zs).contiguous())
        self.layout = "v2"
        self.sz_cdna4 = self.szh_cdna4 = None

    @torch.no_grad()
    def _to_cdna4(self):
        from . import ops
        self.qweight = torch.stack([ops.repack_v2_to_cdna4(self.qweight[e].contiguous()) for e in range(self.num_experts)])
        self.sz_cdna4, self.szh_cdna4 = _stack_side_buffers(self.scales, self.scaled_zeros, self.in_features, ops.pack_sz_cdna4, ops.pack_szh_cdna4)
        self.layout = "cdna4"

    @torch.no_grad()
    def forward(self, x_sorted: torch.Tensor, expert_offsets: torch.Tensor) -> torch.Tensor:
        from . import ops
        if self.layout != "cdna4":
            self._to_cdna4()
        elif self.sz_cdna4 is None or self.sz_cdna4.device != self.qweight.device:  # (a loaded cdna4 checkpoint, or the module moved: only the side buffers)
            self.sz_cdna4, self.szh_cdna4 = _stack_side_buffers(self.scales, self.scaled_zeros, self.in_features, ops.pack_sz_cdna4, ops.pack_szh_cdna4)
        return ops.moe_mlp_gate_up_cdna4(x_sorted.contiguous(), self.qweight, self.scales, self.scaled_zeros, self.sz_cdna4, expert_offsets,
                                         sz_half=getattr(self, "szh_cdna4", None))


class SparseMoeMLP(nn.Module):
    """Mixtral-style block: y = sum_k p_k * w2_e( silu(w1_e x) * w3_e x ) over each token's top-k experts."""

    def __init__(self, w1, w3, w2: GroupedWQLinear, top_k: int = 2, gate_up: Optional[GroupedGateUp] = None):
        """w1 / w3: GroupedWQLinear modules (two grouped launches + the SiLU * mul tail as a third), or None with `gate_up` = a
        GroupedGateUp built from the same experts (ONE grouped launch for h: `SparseMoeMLP.fused(w1_experts, w3_experts, w2, top_k)`)."""
        super().__init__()
        self.w1, self.w3, self.w2, self.top_k, self.gate_up = w1, w3, w2, top_k, gate_up
        assert gate_up is not None or (w1 is not None and w3 is not None)

    @classmethod
    def fused(cls, w1_experts: Sequence[WQLinear], w3_experts: Sequence[WQLinear], w2: GroupedWQLinear, top_k: int = 2):
        return cls(None, None, w2, top_k, GroupedGateUp(w1_experts, w3_experts))

    def _h(self, xs, offsets):
        if self.gate_up is not None:
            return self.gate_up(xs, offsets)
        return self._silu_mul(self.w1(xs, offsets), self.w3(xs, offsets))

    @staticmethod
    def _silu_mul(a, b):
        """the SiLU * mul tail as the HIP kernel of the dense path's epilogue (no torch arithmetic on the hot path; raises without a GPU)"""
        from . import ops
        return ops.silu_mul(a, b)

    @torch.no_grad()
    def forward(self, x: torch.Tensor, router_logits: torch.Tensor) -> torch.Tensor:
        T = x.shape[0]
        probs = torch.softmax(router_logits.float(), dim=-1)
        w, ids = torch.topk(probs, self.top_k, dim=-1)
        w = (w / w.sum(-1, keepdim=True)).to(x.dtype)
        order, offsets = sort_by_expert(ids, self.w2.num_experts)
        xs = x[order // self.top_k]
        h = self._h(xs, offsets)
        ys = self.w2(h, offsets)
        out = torch.zeros(T * self.top_k, ys.shape[1], dtype=ys.dtype, device=ys.device)
        out[order] = ys
        return (out.view(T, self.top_k, -1) * w.unsqueeze(-1)).sum(1)
