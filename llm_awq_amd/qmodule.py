"""`WQLinear` for MI355X: same Python API and packed-buffer contract as the reference's
awq/quantize/qmodule.py:78-235, with forward() running the hand-written gfx950 kernels.

Mirrored surface: ctor signature, attributes (in_features, out_features, w_bit,
group_size, split_k_iters, interleave), registered buffers `qweight int16 [N/4, K]`,
`scales T [Gpad, N]`, `scaled_zeros T [Gpad, N]`, `bias T [N]`, `from_linear`, `forward`,
`extra_repr`; module-level `pack_intweight`, `calculate_zeros_width`, `make_divisible`,
`ScaledActivation`.  Importing this module needs no GPU; forward() does (no CPU fallback).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import load_engine


def make_divisible(c: int, divisor: int) -> int:
    """qmodule.py:7-8 -- a ceiling division (the reference's name is kept for drop-in imports)."""
    return -(-c // divisor)


def calculate_zeros_width(in_features: int, group_size: int = 128, pack_num: int = 8) -> int:
    """qmodule.py:11-23: int32 zero-point words per row in v1; the v2 scale buffers have 8x this many rows."""
    if group_size >= 128:
        mult = 1
    elif group_size in (64, 32):
        mult = 128 // group_size
    else:
        raise NotImplementedError
    return make_divisible(make_divisible(in_features // group_size, pack_num), mult) * mult


def _v2_nibble_source(K: int, device) -> torch.Tensor:
    """For one packed row-quad of the v2 layout: index table src[c, j] -> (rr * K + k) of the logical
    weight stored in nibble j of int16 column c.  Closed form of the three permutations of
    qmodule.py:31-57 (see include/awq_cdna4.h / DESIGN.md 'v2 interleave')."""
    c = torch.arange(K, device=device).view(K, 1)
    j = torch.arange(4, device=device).view(1, 4)
    p = (c % 64) * 4 + j  # nibble position inside the 256-nibble block
    rr, t = p // 64, p % 32
    chunk = (p % 64) // 32
    a, u = t // 8, t % 8
    k = (c // 64) * 64 + chunk * 32 + 8 * (u % 4) + 2 * a + u // 4
    return rr * K + k  # [K, 4]


def pack_intweight(unpacked_qweight: torch.Tensor, interleave: int = 4, kstride: int = 64) -> torch.Tensor:
    """int [N, K] (values 0..15) -> v2 `int16 [N/4, K]`, bit-identical to qmodule.py:26-65.
    Pure index arithmetic in torch (runs where the tensor lives; the reference round-trips via numpy)."""
    if interleave != 4 or kstride != 64:
        raise NotImplementedError("the v2 format is defined for interleave=4, kstride=64 only")
    N, K = unpacked_qweight.shape
    assert N % 4 == 0 and K % 64 == 0
    q = unpacked_qweight.to(torch.int32).reshape(N // 4, 4 * K)
    src = _v2_nibble_source(K, q.device)  # [K, 4]
    nib = q[:, src.reshape(-1)].reshape(N // 4, K, 4) & 0xF
    word = nib[..., 0] | (nib[..., 1] << 4) | (nib[..., 2] << 8) | (nib[..., 3] << 12)
    # two's-complement wrap to int16 exactly like numpy's astype("int16") in the reference
    return ((word + 0x8000) % 0x10000 - 0x8000).to(torch.int16).contiguous()


def unpack_intweight(qweight: torch.Tensor) -> torch.Tensor:
    """Inverse of :func:`pack_intweight` -> int32 [N, K] (host-side format tool)."""
    R, K = qweight.shape
    src = _v2_nibble_source(K, qweight.device)  # [K, 4]
    w = qweight.to(torch.int32) & 0xFFFF
    nib = torch.stack([(w >> (4 * j)) & 0xF for j in range(4)], dim=-1).reshape(R, K * 4)
    out = torch.empty(R, 4 * K, dtype=torch.int32, device=qweight.device)
    out[:, src.reshape(-1)] = nib
    return out.reshape(R * 4, K)


def _cdna4_gather_index(N: int, K: int, device) -> torch.Tensor:
    """flat index n*K + k of the logical weight held by nibble p of logical word a of lane `lane` of cdna4 tile
    (nb, kg) -> int64 [N/16, K/128, 64, 4, 8] (the interleave defined in include/awq_cdna4.h / DESIGN.md)."""
    ar = lambda n, pos: torch.arange(n, device=device).view([n if d == pos else 1 for d in range(5)])
    nb, kg, lane, a, p = ar(N // 16, 0), ar(K // 128, 1), ar(64, 2), ar(4, 3), ar(8, 4)
    g, nq, r, i, hi = lane // 16, (lane // 4) % 4, lane % 4, p & 3, p >> 2
    n = 16 * nb + 4 * nq + 2 * (i & 1) + hi
    k = 128 * kg + 32 * a + 8 * g + 4 * (i >> 1) + r
    return n * K + k


def pack_w3c(unpacked_qweight: torch.Tensor) -> torch.Tensor:
    """int [N, K] (values 0..7) -> `int16 [N/4, 3K/4]` w3c tiles (this repository's 3-bit format, see
    include/awq_cdna4.h): the cdna4 tile of the same integers with its fourth word folded into bit 3 of every
    nibble of the other three.  Host-side torch index arithmetic (the GPU twin is awq_pack_w3)."""
    N, K = unpacked_qweight.shape
    assert N % 16 == 0 and K % 128 == 0, "w3c tiles need out_features % 16 == 0 and in_features % 128 == 0"
    q = unpacked_qweight.reshape(-1).to(torch.int64)
    idx = _cdna4_gather_index(N, K, q.device)
    sh = (4 * torch.arange(8, device=q.device)).view(1, 1, 1, 1, 8)
    w = ((q[idx] & 7) << sh).sum(-1)  # [N/16, K/128, 64, 4] logical words
    w3 = w[..., 3]
    words = torch.stack([(w[..., c] & 0x77777777) | (((w3 >> c) & 0x11111111) << 3) for c in range(3)], dim=-1)
    words = ((words + 0x80000000) % 0x100000000 - 0x80000000).to(torch.int32)  # two's-complement wrap
    return words.contiguous().view(torch.int16).reshape(N // 4, K * 3 // 4)


class ScaledActivation(nn.Module):
    """qmodule.py:68-75 (imported by awq/quantize/quantizer.py:5 and auto_scale.py:11)."""

    def __init__(self, module, scales):
        super().__init__()
        self.act = module
        self.scales = nn.Parameter(scales.data)

    def forward(self, x):
        return self.act(x) / self.scales.view(1, 1, -1).to(x.device)


class WQLinear(nn.Module):
    def __init__(self, w_bit, group_size, in_features, out_features, bias, dev, dtype=torch.float16):
        super().__init__()
        if w_bit not in [3, 4]:
            # the reference supports 4 only (qmodule.py:82-83); 3 is this repository's INT3 extension (bf16, w3c tiles)
            raise NotImplementedError("Only 4-bit (and the MI355X build's 3-bit) are supported for now.")
        if w_bit == 3 and dtype not in (torch.bfloat16, torch.float16):
            raise NotImplementedError("w_bit=3 (w3c tiles, matrix-core dequant) is defined for bfloat16 / float16")
        self.in_features = in_features
        self.out_features = out_features
        self.w_bit = w_bit
        self.group_size = group_size if group_size != -1 else in_features
        self.split_k_iters = 8  # kept writable for tinychat/utils/tune.py:51-65; unused by the HIP kernels
        self.interleave = 4
        # "v2" = reference interleave; "cdna4" after to_cdna4() (same buffers, qweight permuted for the matrix-core
        # dequant); "w3c" = the 3-bit tiles (w_bit == 3, always)
        self.layout = "v2" if w_bit == 4 else "w3c"
        self.sz_cdna4 = None
        self.szh_cdna4 = None  # decode side buffer ("sz_half"): None = not built yet, False = this layer's scales are not f16-exact
        self._decode_served = {}  # rows -> does the streaming decode entry serve this (rows, N, K)?  (awq_w4a16_decode_cdna4_plan)
        assert self.in_features % self.group_size == 0
        assert out_features % 8 == 0  # 32 // w_bit for the reference's w_bit = 4 (qmodule.py:93)
        assert out_features % self.interleave == 0
        gpad = calculate_zeros_width(in_features, self.group_size) * 8
        if w_bit == 3:
            assert out_features % 16 == 0 and in_features % 128 == 0 and self.group_size == 128
        cols = in_features if w_bit == 4 else in_features * 3 // 4  # int16 [N/4, K] (4 bit) / [N/4, 3K/4] (3 bit)
        self.register_buffer("qweight", torch.zeros((out_features // self.interleave, cols), dtype=torch.int16, device=dev))
        self.register_buffer("scales", torch.zeros((gpad, out_features), dtype=dtype, device=dev))
        self.register_buffer("scaled_zeros", torch.zeros((gpad, out_features), dtype=dtype, device=dev))
        if bias:
            self.register_buffer("bias", torch.zeros((out_features), dtype=dtype, device=dev))
        else:
            self.bias = None

    @classmethod
    def from_linear(cls, linear, w_bit, group_size, init_only=False, scales=None, zeros=None):
        """qmodule.py:139-199.  `linear.weight` must already be fake-quantised (on the grid)."""
        q = cls(w_bit, group_size, linear.in_features, linear.out_features, linear.bias is not None,
                linear.weight.device, dtype=linear.weight.data.dtype)
        if init_only:
            return q
        assert scales is not None and zeros is not None
        G = q.group_size
        dtype = scales.dtype
        gpad = calculate_zeros_width(linear.in_features, group_size) * 8
        qscales = torch.zeros((scales.shape[0], gpad), dtype=dtype, device=scales.device)
        qscales[:, : scales.shape[1]] = scales
        q.scales = qscales.transpose(1, 0).contiguous()
        if linear.bias is not None:
            q.bias = linear.bias.clone().to(dtype)
        # integer recovery, same arithmetic (in the tensors' dtype, no clamp) as the reference's
        # per-column loop at qmodule.py:176-184, vectorised over columns
        gi = torch.arange(q.in_features, device=scales.device) // G
        scale_zeros = zeros * scales
        intweight = torch.round((linear.weight.data + scale_zeros[:, gi]) / qscales[:, gi]).to(torch.int32)
        if w_bit == 4:
            q.qweight = pack_intweight(intweight.contiguous(), interleave=4, kstride=64)
        else:
            q.qweight = pack_w3c(intweight.contiguous())
        zi = zeros.to(dtype=torch.int32)
        sz = torch.zeros_like(qscales)
        sz[:, : scales.shape[1]] = -(qscales[:, : scales.shape[1]] * zi.to(torch.float32)).to(dtype)
        q.scaled_zeros = sz.transpose(1, 0).contiguous()
        return q

    def engine_converted(self) -> bool:
        """True while the engine's cache holds THIS qweight converted in place (AWQ_CDNA4_INPLACE=1: the reference entry points permuted the
        bytes where they lie; `layout` still says "v2").  Format tools -- to_cdna4, tensor-parallel sharding, QuantLlamaMLP's stacking,
        checkpoint writers -- must `awq_inference_engine.cdna4_restore(qweight)` first; they check this and raise."""
        if not self.qweight.is_cuda or self.layout != "v2":
            return False
        eng = load_engine()
        return bool(eng.cdna4_is_converted(self.qweight)) if hasattr(eng, "cdna4_is_converted") else False

    def _refuse_converted(self, what: str):
        if self.engine_converted():
            raise RuntimeError(f"{what}: this module's qweight was converted in place by the engine cache (AWQ_CDNA4_INPLACE); call "
                               "awq_inference_engine.cdna4_restore(module.qweight) first")

    # ---- MI355X-native layout (no reference counterpart; what llm_awq_amd.repacker emits) ----
    @torch.no_grad()
    def to_cdna4(self):
        """Permute `qweight` (same shape / dtype, so the checkpoint contract is unchanged) into the cdna4
        interleave and build the packed {scale | scaled_zero} side buffer.  bf16 / fp16, out_features % 16 == 0,
        group_size 128; buffers must live on the GPU.  Idempotent."""
        if self.layout in ("cdna4", "w3c"):
            return self
        self._refuse_converted("to_cdna4")
        if self.scales.dtype not in (torch.bfloat16, torch.float16):
            raise TypeError("the cdna4 interleave (matrix-core dequant) is defined for bfloat16 / float16 WQLinear only")
        if self.out_features % 16 or self.group_size != 128:
            raise ValueError("cdna4 interleave needs out_features % 16 == 0 and group_size == 128")
        eng = load_engine()
        self.qweight = eng.repack_v2_to_cdna4(self.qweight.contiguous())
        self.scales, self.scaled_zeros = self.scales.contiguous(), self.scaled_zeros.contiguous()
        self.sz_cdna4 = eng.pack_sz_cdna4(self.scales, self.scaled_zeros, self.in_features)
        self._sz_key = self._side_key()  # the first forward finds both side buffers current (no rebuild, no host sync)
        self._build_szh(eng)
        self._decode_served = {}
        self.layout = "cdna4"
        return self

    def _side_key(self):
        """identity of what the side buffers were derived from: rebuilt when scales / scaled_zeros moved, changed dtype or were edited in place"""
        return (self.scales.device, self.scales.dtype, self.scales.data_ptr(), self.scales._version, self.scaled_zeros.data_ptr(),
                self.scaled_zeros._version)

    def _build_szh(self, eng):
        """the decode side buffer; reads one flag back from the device (a host sync): never called while a stream is capturing"""
        szh, exact = eng.pack_szh_cdna4(self.scales.contiguous(), self.scaled_zeros.contiguous(), self.in_features)
        self.szh_cdna4 = szh if exact else False

    @torch.no_grad()
    def to_v2(self):
        if self.layout in ("v2", "w3c"):
            return self
        self.qweight = load_engine().repack_cdna4_to_v2(self.qweight)
        self.sz_cdna4 = self.szh_cdna4 = None
        self._decode_served = {}
        self.layout = "v2"
        return self

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self.layout == "cdna4":  # marker key only for native checkpoints; v2 state dicts are unchanged
            destination[prefix + "qweight_layout"] = torch.tensor(1, dtype=torch.uint8)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        marker = state_dict.pop(prefix + "qweight_layout", None)
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)
        self.szh_cdna4 = None
        self._decode_served = {}  # (the plan verdicts are memoised per row count: re-asked after a load / layout change, e.g. when test knobs moved)
        if self.w_bit == 3:
            self.layout, self.sz_cdna4 = "w3c", None
        elif marker is not None and int(marker) == 1:
            if self.group_size != 128 or self.out_features % 16:
                raise ValueError("a cdna4-interleaved checkpoint needs group_size 128 and out_features % 16 == 0")
            self.layout = "cdna4"
            self.sz_cdna4 = None  # rebuilt lazily on the first forward
        else:
            self.layout, self.sz_cdna4 = "v2", None

    @torch.no_grad()
    def forward(self, x):
        """qmodule.py:201-224: fewer than 8 rows -> decode GEMV, else prefill GEMM; bias added after."""
        eng = load_engine()
        if not x.is_contiguous():
            x = x.contiguous()
        if self.layout in ("cdna4", "w3c"):
            if self.group_size != 128:
                raise ValueError("the cdna4 / w3c kernels implement group_size 128 only")
            # side buffers are derived from scales / scaled_zeros: rebuild when those moved, changed dtype or were edited in place
            key = self._side_key()
            if self.sz_cdna4 is None or getattr(self, "_sz_key", None) != key:
                self.sz_cdna4 = eng.pack_sz_cdna4(self.scales, self.scaled_zeros, self.in_features)
                self.szh_cdna4 = None
                self._sz_key = key
            if self.layout == "cdna4" and x.numel() // x.shape[-1] <= 8:
                # (building sz_half reads a flag back: inside a graph capture this call keeps the T-typed sz_packed instead)
                if self.szh_cdna4 is None and not (x.is_cuda and torch.cuda.is_current_stream_capturing()):
                    self._build_szh(eng)
                if self.szh_cdna4 is not None and self.szh_cdna4 is not False:
                    rows = x.numel() // x.shape[-1]
                    # (host-side plan query, decided once per row count: a shape the streaming kernel does not serve goes to the general entry)
                    served = self._decode_served.get(rows)
                    if served is None:
                        served = self._decode_served[rows] = rows >= 1 and eng.decode_cdna4_plan(rows, self.out_features, self.in_features, 0)[0] > 0
                    if served:
                        return eng.decode_cdna4(x, self.qweight, self.szh_cdna4, self.bias, 0)
            if self.layout == "cdna4":
                # (prompts: the tile kernels take the layer's sz_half side buffer too when it exists -- the f16-mantissa dequant form, same results)
                szh = self.szh_cdna4 if (self.szh_cdna4 is not None and self.szh_cdna4 is not False) else None
                return eng.forward_cdna4(x, self.qweight, self.scales, self.scaled_zeros, self.sz_cdna4, self.bias, szh)
            return eng.forward_w3(x, self.qweight, self.scales, self.scaled_zeros, self.sz_cdna4, self.bias)
        rows = x.numel() // x.shape[-1]
        if rows < 8:
            out = eng.gemv_forward_cuda_new(x, self.qweight, self.scales, self.scaled_zeros, rows,
                                            self.out_features, self.in_features, self.group_size)
        else:
            out = eng.gemm_forward_cuda_new(x, self.qweight, self.scales, self.scaled_zeros)
        return out + self.bias if self.bias is not None else out

    def extra_repr(self) -> str:
        return "in_features={}, out_features={}, bias={}, w_bit={}, group_size={}".format(
            self.in_features, self.out_features, self.bias is not None, self.w_bit, self.group_size)
