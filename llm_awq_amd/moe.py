"""Grouped (per-expert) WQLinear for MoE layers -- BASELINE.json config 5 (Mixtral-8x7B W4A16 g128).

The reference has no MoE path, so this is new capability built from the same contract:
every expert is a `WQLinear` whose buffers are stacked along a leading expert dimension
    qweight int16 [E, N/4, K]   scales / scaled_zeros T [E, Gpad, N]
and one launch runs all experts over tokens sorted by expert (`awq_w4a16_moe_gemm`).  Routing (top-k, sort,
un-sort, weighting) is ordinary torch glue around the path; it never synchronises the host: the per-expert row
ranges go to the kernel as a device int32 [E + 1] offsets array.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence

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


class GroupedWQLinear(nn.Module):
    """E stacked WQLinear experts with one grouped launch.  `matmul(x_sorted, qweight, scales, scaled_zeros, offsets)`
    defaults to the HIP engine; CPU tests inject the oracle."""

    def __init__(self, experts: Sequence[WQLinear], matmul: Optional[Callable] = None):
        super().__init__()
        e0 = experts[0]
        assert all(e.w_bit == 4 and e.layout == e0.layout and e.bias is None for e in experts)
        assert all((e.in_features, e.out_features) == (e0.in_features, e0.out_features) for e in experts)
        self.num_experts, self.in_features, self.out_features = len(experts), e0.in_features, e0.out_features
        self.layout = e0.layout
        self.register_buffer("qweight", torch.stack([e.qweight for e in experts]).contiguous())
        self.register_buffer("scales", torch.stack([e.scales for e in experts]).contiguous())
        self.register_buffer("scaled_zeros", torch.stack([e.scaled_zeros for e in experts]).contiguous())
        self._matmul = matmul
        self.sz_cdna4 = None  # stacked packed scales int32 [E, N/16, K/128, 16], built lazily for the cdna4 layout

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
        if self._matmul is not None:
            return self._matmul(x_sorted, self.qweight, self.scales, self.scaled_zeros, expert_offsets)
        eng = load_engine()
        if self.layout == "cdna4":
            if self.sz_cdna4 is None or self.sz_cdna4.device != self.scales.device:
                self.sz_cdna4 = torch.stack([eng.pack_sz_cdna4(self.scales[e].contiguous(), self.scaled_zeros[e].contiguous(),
                                                                self.in_features) for e in range(self.num_experts)])
            return eng.moe_forward_cdna4(x_sorted.contiguous(), self.qweight, self.scales, self.scaled_zeros, self.sz_cdna4,
                                         expert_offsets)
        return eng.moe_gemm_forward(x_sorted.contiguous(), self.qweight, self.scales, self.scaled_zeros, expert_offsets, False)


class SparseMoeMLP(nn.Module):
    """Mixtral-style block: y = sum_k p_k * w2_e( silu(w1_e x) * w3_e x ) over each token's top-k experts."""

    def __init__(self, w1: GroupedWQLinear, w3: GroupedWQLinear, w2: GroupedWQLinear, top_k: int = 2):
        super().__init__()
        self.w1, self.w3, self.w2, self.top_k = w1, w3, w2, top_k

    @torch.no_grad()
    def forward(self, x: torch.Tensor, router_logits: torch.Tensor) -> torch.Tensor:
        T = x.shape[0]
        probs = torch.softmax(router_logits.float(), dim=-1)
        w, ids = torch.topk(probs, self.top_k, dim=-1)
        w = (w / w.sum(-1, keepdim=True)).to(x.dtype)
        order, offsets = sort_by_expert(ids, self.w1.num_experts)
        xs = x[order // self.top_k]
        h = torch.nn.functional.silu(self.w1(xs, offsets)) * self.w3(xs, offsets)
        ys = self.w2(h, offsets)
        out = torch.zeros(T * self.top_k, ys.shape[1], dtype=ys.dtype, device=ys.device)
        out[order] = ys
        return (out.view(T, self.top_k, -1) * w.unsqueeze(-1)).sum(1)
