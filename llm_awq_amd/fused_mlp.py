"""QuantLlamaMLP / make_fused_mlp -- the MI355X build of tinychat/modules/fused_mlp.py:11-101.

The reference module keeps gate_proj / up_proj as raw v2 buffers, issues two `gemv_forward_cuda_new` (or two
`gemm_forward_cuda_new`) calls, `F.silu` and a multiply, then calls `down_proj`.  Here the pair is ONE weight stream:

  * construction interleaves the gate and up rows 8 + 8 inside every 16-row slab (slab j = gate rows 8j..8j+7, then up rows
    8j..8j+7; in the v2 buffers that is a permutation of whole packed rows, 4 logical rows each) and repacks the result to the
    cdna4 interleave.  2 * ffn / 16 blocks of one tile stream each (1792 for Llama-3-8B: 7 per CU) instead of ffn / 16 blocks of
    two streams (896: 3.5 per CU, so half the CUs carried 4 blocks and the rest 3);
  * decode (<= 8 rows): `decode_cdna4(..., epilogue=2)` -- gate, up, SiLU and the multiply in one launch, every intermediate
    rounded to T exactly like the reference's separate ops (fused_mlp.py:39-61, :79-82) -- then down_proj's launch.  (A whole-module one-launch engine was built
    in rounds 2, 4 and 5 and measured 30-37 % slower each time -- the all-gather of h to every CU costs more than the kernel boundary it removes; it was
    taken out of the library in round 6: tools/EXPERIMENTS.md, profiles/r05_mlp_engine.txt.);
  * prefill (>= 8 rows): one GEMM over the interleaved weight (x is read once for both projections) whose tile epilogue pairs
    column n with column n + 8 and stores silu(gate) * up directly -- the [rows, 2 * ffn] intermediate of the reference's two
    GEMMs + F.silu + multiply is never written.

3-bit projections (`WQLinear(w_bit=3)`, this repository's extension): the same module.  The w3c tiles hold 16 rows each, so the 8 + 8
interleave is done on the INTEGER rows (`awq_unpack_w3` -> permute -> `awq_pack_w3`) and the fused stream is served by
`awq_w3a16_mlp_gate_up_forward` (<= 8 rows: the register-ring decode kernel pairs the rows in its epilogue; more: the same tile epilogue).
Llama-2-7B: 11.2-11.7 us for the fused decode launch against 2 x 6.5-6.7 us (profiles/r05_w3_fused_mlp.txt).

`scaled_zeros - 8 * scales` (fused_mlp.py:69,76): the reference's GEMM branch shifts the zero point by 8 while its GEMV
branch, `WQLinear.forward` (qmodule.py:220, shift commented out), `from_linear` and the offline repacker (`zp_shift = 0`,
offline-weight-repacker.py:142) all treat the stored nibbles as UNSIGNED 0..15 (dequantize.cuh:59-69 yields 0..15).  Both
cannot be right for the same buffers; the unsigned convention is the checkpoint contract, so this module
uses it for every row count -- decode and prefill agree with each other and with `WQLinear.forward`, which the reference's two
branches do not.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import load_engine


def interleave_gate_up(gq, uq, gs, us, gz, uz):
    """v2 buffers of gate and up (qweight int16 [F/4, K], scales / scaled_zeros [Gpad, F]) -> the 8 + 8 row-interleaved
    stack (qweight int16 [2F/4, K], scales / scaled_zeros [Gpad, 2F]).  F % 8 == 0."""
    F4, K = gq.shape
    assert uq.shape == gq.shape and F4 % 2 == 0, "intermediate size must be a multiple of 8"
    q = torch.stack([gq.view(F4 // 2, 2, K), uq.view(F4 // 2, 2, K)], 1).reshape(2 * F4, K).contiguous()
    Fo = gs.shape[1]

    def cols(a, b):
        return torch.stack([a.view(-1, Fo // 8, 8), b.view(-1, Fo // 8, 8)], 2).reshape(a.shape[0], 2 * Fo).contiguous()

    return q, cols(gs, us), cols(gz, uz)


def deinterleave_gate_up(q, s, z):
    """inverse of interleave_gate_up: (gate qweight, up qweight, gate scales, up scales, gate scaled_zeros, up scaled_zeros)"""
    F2, K = q.shape
    q4 = q.view(F2 // 4, 2, 2, K)
    Fo = s.shape[1] // 2

    def cols(a):
        a4 = a.view(a.shape[0], Fo // 8, 2, 8)
        return a4[:, :, 0, :].reshape(a.shape[0], Fo).contiguous(), a4[:, :, 1, :].reshape(a.shape[0], Fo).contiguous()

    (gs, us), (gz, uz) = cols(s), cols(z)
    return q4[:, 0].reshape(F2 // 2, K).contiguous(), q4[:, 1].reshape(F2 // 2, K).contiguous(), gs, us, gz, uz


def interleave_gate_up_w3(gq3, uq3, gs, us, gz, uz):
    """the same 8 + 8 interleave for 3-bit projections: the w3c tiles hold 16 rows each, so the rows are interleaved as INTEGERS
    (awq_unpack_w3 -> permute -> awq_pack_w3, GPU kernels) -> (qweight int16 [2F/4, 3K/4], scales / scaled_zeros [Gpad, 2F])."""
    from . import ops
    g, u = ops.unpack_w3(gq3.contiguous()), ops.unpack_w3(uq3.contiguous())  # uint8 [F, K]
    Fo, K = g.shape
    assert u.shape == g.shape and Fo % 8 == 0, "intermediate size must be a multiple of 8"
    q = torch.stack([g.view(Fo // 8, 8, K), u.view(Fo // 8, 8, K)], 1).reshape(2 * Fo, K).contiguous()

    def cols(a, b):
        return torch.stack([a.view(-1, Fo // 8, 8), b.view(-1, Fo // 8, 8)], 2).reshape(a.shape[0], 2 * Fo).contiguous()

    return ops.pack_w3(q), cols(gs, us), cols(gz, uz)


def deinterleave_gate_up_w3(q3, s, z):
    """inverse of interleave_gate_up_w3"""
    from . import ops
    q = ops.unpack_w3(q3.contiguous())
    F2, K = q.shape
    q8 = q.view(F2 // 16, 2, 8, K)
    Fo = F2 // 2

    def cols(a):
        a4 = a.view(a.shape[0], Fo // 8, 2, 8)
        return a4[:, :, 0, :].reshape(a.shape[0], Fo).contiguous(), a4[:, :, 1, :].reshape(a.shape[0], Fo).contiguous()

    (gs, us), (gz, uz) = cols(s), cols(z)
    return (ops.pack_w3(q8[:, 0].reshape(Fo, K).contiguous()), ops.pack_w3(q8[:, 1].reshape(Fo, K).contiguous()), gs, us, gz, uz)


_V2_NAMES = ("gate_proj_qweight", "gate_proj_scales", "gate_proj_scaled_zeros", "up_proj_qweight", "up_proj_scales", "up_proj_scaled_zeros")


class QuantLlamaMLP(nn.Module):
    """Same constructor and attributes as the reference (gate_proj, down_proj, up_proj are WQLinear modules); the v2 buffers are
    registered under the reference's names so state dicts are interchangeable (fused_mlp.py:19-27).  Once the fused cdna4 stream
    has been built on the GPU the six v2 buffers are RELEASED (they would be a second copy of two thirds of the block's weights:
    14 GB on Llama-3-70B) -- `state_dict()` rebuilds them bit-exactly from the fused stream, `load_state_dict()` re-materialises them
    first.  AWQ_MLP_KEEP_V2=1 keeps both copies."""

    def __init__(self, gate_proj, down_proj, up_proj):
        super().__init__()
        self.register_buffer("gate_proj_qweight", gate_proj.qweight)
        self.register_buffer("gate_proj_scales", gate_proj.scales)
        self.register_buffer("gate_proj_scaled_zeros", gate_proj.scaled_zeros)
        self.register_buffer("up_proj_qweight", up_proj.qweight)
        self.register_buffer("up_proj_scales", up_proj.scales)
        self.register_buffer("up_proj_scaled_zeros", up_proj.scaled_zeros)
        if gate_proj.w_bit != up_proj.w_bit:
            raise ValueError("QuantLlamaMLP: gate_proj and up_proj must share w_bit")
        want = "w3c" if gate_proj.w_bit == 3 else "v2"  # (3-bit projections only exist as w3c tiles; their rows are interleaved as integers)
        if getattr(gate_proj, "layout", "v2") != want or getattr(up_proj, "layout", "v2") != want:
            raise ValueError("QuantLlamaMLP is built from v2 (reference layout) gate / up projections; it makes its own cdna4 stream")
        if gate_proj.w_bit == 3 and (gate_proj.out_features % 16 != 0 or gate_proj.group_size != 128):
            raise ValueError("QuantLlamaMLP (w_bit = 3): intermediate size % 16 == 0 and group_size 128")
        self.in_features = gate_proj.in_features
        self.intermediate_size = gate_proj.out_features
        self.out_features = down_proj.out_features
        self.w_bit = gate_proj.w_bit
        self.down_proj = down_proj
        self.split_k_iters = down_proj.split_k_iters
        self._fused = None  # (qweight cdna4, scales, scaled_zeros, sz_packed, sz_half or None): built on the first GPU forward
        self._v2_released = False
        self._v2_meta = None  # (shape, dtype) of the six released buffers
        self._register_state_dict_hook(QuantLlamaMLP._fill_state_dict)
        self._register_load_state_dict_pre_hook(self._rematerialise_v2)

    def _apply(self, fn, *args, **kwargs):
        """.to() / .cuda() / .half(): the fused stream is a plain tuple nn.Module does not move, and while the six reference-named buffers are
        released it is the only copy -- they are rematerialised first, moved by nn.Module like any buffer, and the stream (with the device it
        was pinned on) is dropped and rebuilt at the next forward."""
        if self._fused is not None:
            self._rematerialise_v2()
        return super()._apply(fn, *args, **kwargs)

    # ---- the v2 buffers while they are released ----
    @torch.no_grad()
    def _v2_from_fused(self):
        c4, s, z, _szp, _szh = self._fused
        names = ("gate_proj_qweight", "up_proj_qweight", "gate_proj_scales", "up_proj_scales", "gate_proj_scaled_zeros", "up_proj_scaled_zeros")
        if self.w_bit == 3:
            return dict(zip(names, deinterleave_gate_up_w3(c4, s, z)))
        q = load_engine().repack_cdna4_to_v2(c4)
        return dict(zip(names, deinterleave_gate_up(q, s, z)))

    @staticmethod
    def _fill_state_dict(module, state_dict, prefix, local_metadata):
        if module._v2_released:
            for name, t in module._v2_from_fused().items():
                state_dict[prefix + name] = t
        return state_dict

    def _rematerialise_v2(self, *args, **kwargs):
        """before load_state_dict: the incoming tensors need full-size buffers to be copied into; the fused stream is rebuilt after"""
        if self._v2_released:
            for name, t in self._v2_from_fused().items():
                setattr(self, name, t)
            self._v2_released = False
        self._fused = None

    @torch.no_grad()
    def _build(self, device=None):
        eng = load_engine()
        if self._v2_released:  # (the module moved to another device after the buffers were released: carry them over)
            for name, t in self._v2_from_fused().items():
                setattr(self, name, t if device is None else t.to(device))
            self._v2_released = False
        if hasattr(eng, "cdna4_is_converted") and self.gate_proj_qweight.is_cuda and (
                eng.cdna4_is_converted(self.gate_proj_qweight) or eng.cdna4_is_converted(self.up_proj_qweight)):
            raise RuntimeError("QuantLlamaMLP: gate_proj / up_proj qweight was converted in place by the engine cache (AWQ_CDNA4_INPLACE); "
                               "call awq_inference_engine.cdna4_restore(qweight) on both before the fused stream is built")
        if self.w_bit == 3:
            c4, s, z = interleave_gate_up_w3(self.gate_proj_qweight, self.up_proj_qweight, self.gate_proj_scales, self.up_proj_scales,
                                             self.gate_proj_scaled_zeros, self.up_proj_scaled_zeros)
            self._fused = (c4, s, z, eng.pack_sz_cdna4(s, z, self.in_features), None)
        else:
            q, s, z = interleave_gate_up(self.gate_proj_qweight, self.up_proj_qweight, self.gate_proj_scales, self.up_proj_scales,
                                         self.gate_proj_scaled_zeros, self.up_proj_scaled_zeros)
            c4 = eng.repack_v2_to_cdna4(q)
            szp = eng.pack_sz_cdna4(s, z, self.in_features)
            szh, exact = eng.pack_szh_cdna4(s, z, self.in_features)
            self._fused = (c4, s, z, szp, szh if exact else None)
        if getattr(self.down_proj, "layout", None) == "v2" and self.down_proj.out_features % 16 == 0:
            self.down_proj.to_cdna4()
        if c4.is_cuda and os.environ.get("AWQ_MLP_KEEP_V2") != "1":
            for name in _V2_NAMES:  # (registered names stay: zero-size placeholders on the same device)
                t = getattr(self, name)
                setattr(self, name, torch.empty(0, dtype=t.dtype, device=t.device))
            self._v2_released = True

    @torch.no_grad()
    def forward(self, x):
        return self.down_proj(self.our_llama_mlp(x))

    @torch.no_grad()
    def our_llama_mlp(self, x):
        eng = load_engine()
        if self._fused is None or self._fused[0].device != x.device:
            self._build(x.device)
        c4, s, z, szp, szh = self._fused
        if not x.is_contiguous():
            x = x.contiguous()
        # one entry point for every row count: <= 8 rows the streaming decode launch, more the tile kernels with the fused tail
        if self.w_bit == 3:
            from . import ops
            return ops.mlp_gate_up_forward_w3(x, c4, szp)
        return eng.mlp_gate_up_forward_cdna4(x, c4, szp, szh)


def make_fused_mlp(m, parent_name=""):
    """tinychat/modules/fused_mlp.py:86-101: replace every LlamaMLP whose projections are WQLinear modules."""
    if m.__class__.__name__ in ["LlamaMLP"]:
        return QuantLlamaMLP(m.gate_proj, m.down_proj, m.up_proj)
    for name, child in m.named_children():
        child = make_fused_mlp(child, parent_name=f"{parent_name}.{name}")
        if isinstance(child, QuantLlamaMLP):
            setattr(m, name, child)
    return m
