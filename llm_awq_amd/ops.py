"""Torch-tensor conveniences over the C ABI (device memory + current stream are the only things
torch provides here).  Every function requires GPU tensors and the built HIP library."""
from __future__ import annotations

import torch

from . import _capi


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float16:
        return _capi.AWQ_F16
    if t.dtype == torch.bfloat16:
        return _capi.AWQ_BF16
    raise TypeError(f"expected float16/bfloat16, got {t.dtype}")


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _capi.AwqNativeError("llm_awq_amd ops run on the GPU only (no CPU fallback)")
        if t is not None and not t.is_contiguous():
            raise ValueError("tensors must be contiguous")


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def gemv(x, qweight, scales, scaled_zeros, group_size: int = 128):
    """C-ABI awq_w4a16_gemv: x [..., K] with 1 <= M <= 16 rows."""
    _need_gpu(x, qweight, scales, scaled_zeros)
    k = x.shape[-1]
    m = x.numel() // k
    n = qweight.shape[0] * 4
    out = torch.empty(*x.shape[:-1], n, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().awq_w4a16_gemv(x.data_ptr(), qweight.data_ptr(), scales.data_ptr(),
                                                scaled_zeros.data_ptr(), out.data_ptr(), m, n, k, group_size, _dt(x),
                                                _stream(x)))
    return out


def gemm(x, qweight, scales, scaled_zeros, group_size: int = 128):
    """C-ABI awq_w4a16_gemm: any M >= 1."""
    _need_gpu(x, qweight, scales, scaled_zeros)
    k = x.shape[-1]
    m = x.numel() // k
    n = qweight.shape[0] * 4
    out = torch.empty(*x.shape[:-1], n, dtype=x.dtype, device=x.device)
    L = _capi.lib()
    wsb = L.awq_w4a16_gemm_workspace_bytes(m, n, k)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
    with torch.cuda.device(x.device):
        _capi.check(L.awq_w4a16_gemm(x.data_ptr(), qweight.data_ptr(), scales.data_ptr(), scaled_zeros.data_ptr(),
                                     out.data_ptr(), m, n, k, group_size, _dt(x), ws.data_ptr() if wsb else None, wsb,
                                     _stream(x)))
    return out


def forward(x, qweight, scales, scaled_zeros, bias=None, group_size: int = 128):
    """C-ABI awq_w4a16_forward: WQLinear.forward's dispatch + optional bias."""
    _need_gpu(x, qweight, scales, scaled_zeros, bias)
    k = x.shape[-1]
    m = x.numel() // k
    n = qweight.shape[0] * 4
    out = torch.empty(*x.shape[:-1], n, dtype=x.dtype, device=x.device)
    L = _capi.lib()
    wsb = L.awq_w4a16_gemm_workspace_bytes(m, n, k)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
    with torch.cuda.device(x.device):
        _capi.check(L.awq_w4a16_forward(x.data_ptr(), qweight.data_ptr(), scales.data_ptr(), scaled_zeros.data_ptr(),
                                        bias.data_ptr() if bias is not None else None, out.data_ptr(), m, n, k,
                                        group_size, _dt(x), ws.data_ptr() if wsb else None, wsb, _stream(x)))
    return out


def unpack_v2(qweight):
    """int16 [N/4, K] -> uint8 [N, K] logical 4-bit integers (GPU kernel, same unpack code as the matmuls)."""
    _need_gpu(qweight)
    n, k = qweight.shape[0] * 4, qweight.shape[1]
    out = torch.empty(n, k, dtype=torch.uint8, device=qweight.device)
    with torch.cuda.device(qweight.device):
        _capi.check(_capi.lib().awq_unpack_v2(qweight.data_ptr(), out.data_ptr(), n, k, _stream(qweight)))
    return out


def dequant_v2(qweight, scales, scaled_zeros, group_size: int = 128):
    """-> T [N, K] = round_T(q*s + sz) (GPU kernel, same dequant code as the matmuls)."""
    _need_gpu(qweight, scales, scaled_zeros)
    n, k = qweight.shape[0] * 4, qweight.shape[1]
    out = torch.empty(n, k, dtype=scales.dtype, device=qweight.device)
    with torch.cuda.device(qweight.device):
        _capi.check(_capi.lib().awq_dequant_v2(qweight.data_ptr(), scales.data_ptr(), scaled_zeros.data_ptr(),
                                                out.data_ptr(), n, k, group_size, _dt(scales), _stream(qweight)))
    return out


def pack_v2(q_u8):
    """uint8 [N, K] -> int16 [N/4, K] (GPU pack_intweight)."""
    _need_gpu(q_u8)
    assert q_u8.dtype == torch.uint8
    n, k = q_u8.shape
    out = torch.empty(n // 4, k, dtype=torch.int16, device=q_u8.device)
    with torch.cuda.device(q_u8.device):
        _capi.check(_capi.lib().awq_pack_v2(q_u8.data_ptr(), out.data_ptr(), n, k, _stream(q_u8)))
    return out


def repack_v1_to_v2(qweight_v1, scales_v1, qzeros_v1):
    """v1 (qweight int32 [N,K/8], scales T [N,Gpad], qzeros int32 [N,Gpad/8]) -> v2 triple."""
    _need_gpu(qweight_v1, scales_v1, qzeros_v1)
    n, k = qweight_v1.shape[0], qweight_v1.shape[1] * 8
    gpad = scales_v1.shape[1]
    dev = qweight_v1.device
    qw2 = torch.empty(n // 4, k, dtype=torch.int16, device=dev)
    s2 = torch.empty(gpad, n, dtype=scales_v1.dtype, device=dev)
    sz2 = torch.empty(gpad, n, dtype=scales_v1.dtype, device=dev)
    with torch.cuda.device(dev):
        _capi.check(_capi.lib().awq_repack_v1_to_v2(qweight_v1.data_ptr(), scales_v1.data_ptr(), qzeros_v1.data_ptr(),
                                                     qw2.data_ptr(), s2.data_ptr(), sz2.data_ptr(), n, k, gpad,
                                                     _dt(scales_v1), _stream(qweight_v1)))
    return qw2, s2, sz2


# ---- cdna4 interleave (bf16) ----

def repack_v2_to_cdna4(qweight_v2):
    _need_gpu(qweight_v2)
    n, k = qweight_v2.shape[0] * 4, qweight_v2.shape[1]
    out = torch.empty_like(qweight_v2)
    with torch.cuda.device(qweight_v2.device):
        _capi.check(_capi.lib().awq_repack_v2_to_cdna4(qweight_v2.data_ptr(), out.data_ptr(), n, k, _stream(qweight_v2)))
    return out


def repack_cdna4_to_v2(qweight_cdna4):
    _need_gpu(qweight_cdna4)
    n, k = qweight_cdna4.shape[0] * 4, qweight_cdna4.shape[1]
    out = torch.empty_like(qweight_cdna4)
    with torch.cuda.device(qweight_cdna4.device):
        _capi.check(_capi.lib().awq_repack_cdna4_to_v2(qweight_cdna4.data_ptr(), out.data_ptr(), n, k, _stream(qweight_cdna4)))
    return out


def unpack_cdna4(qweight):
    _need_gpu(qweight)
    n, k = qweight.shape[0] * 4, qweight.shape[1]
    out = torch.empty(n, k, dtype=torch.uint8, device=qweight.device)
    with torch.cuda.device(qweight.device):
        _capi.check(_capi.lib().awq_unpack_cdna4(qweight.data_ptr(), out.data_ptr(), n, k, _stream(qweight)))
    return out


def dequant_cdna4(qweight, scales, scaled_zeros, group_size: int = 128):
    _need_gpu(qweight, scales, scaled_zeros)
    n, k = qweight.shape[0] * 4, qweight.shape[1]
    out = torch.empty(n, k, dtype=scales.dtype, device=qweight.device)
    with torch.cuda.device(qweight.device):
        _capi.check(_capi.lib().awq_dequant_cdna4(qweight.data_ptr(), scales.data_ptr(), scaled_zeros.data_ptr(),
                                                   out.data_ptr(), n, k, group_size, _dt(scales), _stream(qweight)))
    return out


def pack_sz_cdna4(scales, scaled_zeros, in_features: int):
    """-> int32 [N/16, K/128, 16] packed {scale | scaled_zero << 16} (bit patterns)."""
    _need_gpu(scales, scaled_zeros)
    n, k = scales.shape[1], in_features
    out = torch.empty(n // 16, k // 128, 16, dtype=torch.int32, device=scales.device)
    with torch.cuda.device(scales.device):
        _capi.check(_capi.lib().awq_pack_sz_cdna4(scales.data_ptr(), scaled_zeros.data_ptr(), out.data_ptr(), n, k,
                                                   _stream(scales)))
    return out


def pack_szh_cdna4(scales, scaled_zeros, in_features: int):
    """-> (int32 [N/16, K/128, 16] "sz_half" {f16(s') | f16(sz) << 16}, exact: bool).  `exact` False means a scale of this
    layer is not representable as a normal f16 number: keep sz_packed (pack_sz_cdna4) for it.  Synchronises (reads the flag)."""
    _need_gpu(scales, scaled_zeros)
    n, k = scales.shape[1], in_features
    out = torch.empty(n // 16, k // 128, 16, dtype=torch.int32, device=scales.device)
    flag = torch.zeros(1, dtype=torch.int32, device=scales.device)
    with torch.cuda.device(scales.device):
        _capi.check(_capi.lib().awq_pack_szh_cdna4(scales.data_ptr(), scaled_zeros.data_ptr(), out.data_ptr(), flag.data_ptr(), n, k,
                                                    _dt(scales), _stream(scales)))
    return out, int(flag.item()) == 0


def decode_cdna4(x, qweight, sz_half, bias=None, epilogue: int = 0, group_size: int = 128):
    """C-ABI awq_w4a16_decode_cdna4: 1 <= M <= 8 on cdna4 weights + sz_half.  epilogue 0: x.W^T (+ bias); 1: stacked [gate; up]
    -> silu(gate) * up; 2: the same with gate / up rows interleaved 8 + 8 per slab."""
    _need_gpu(x, qweight, sz_half, bias)
    k = x.shape[-1]
    m = x.numel() // k
    n = qweight.shape[0] * 4
    out = torch.empty(*x.shape[:-1], n // 2 if epilogue else n, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().awq_w4a16_decode_cdna4(x.data_ptr(), qweight.data_ptr(), sz_half.data_ptr(),
                                                        bias.data_ptr() if bias is not None else None, out.data_ptr(), m, n, k,
                                                        group_size, _dt(x), int(epilogue), _stream(x)))
    return out


def partial_cdna4(x, qweight, sz_packed, sz_half=None, group_size: int = 128):
    """C-ABI awq_w4a16_partial_cdna4: the K shard's product x . W^T as fp32 [..., N], unrounded, no bias (tensor-parallel row split)."""
    _need_gpu(x, qweight, sz_packed, sz_half)
    k = x.shape[-1]
    m = x.numel() // k
    n = qweight.shape[0] * 4
    out = torch.empty(*x.shape[:-1], n, dtype=torch.float32, device=x.device)
    if m == 0:
        return out
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().awq_w4a16_partial_cdna4(x.data_ptr(), qweight.data_ptr(), sz_packed.data_ptr(),
                                                         sz_half.data_ptr() if sz_half is not None else None, out.data_ptr(), m, n, k,
                                                         group_size, _dt(x), _stream(x)))
    return out


def round_bias_f32(y32, dtype, bias=None):
    """C-ABI awq_round_bias_f32: T(y32) (+ bias in T) -- the single rounding after the fp32 partials of a row split were summed."""
    _need_gpu(y32, bias)
    if y32.dtype != torch.float32:
        raise TypeError("expected the float32 sum of the partials")
    n = y32.shape[-1]
    m = y32.numel() // n
    out = torch.empty(y32.shape, dtype=dtype, device=y32.device)
    if m == 0:
        return out
    with torch.cuda.device(y32.device):
        _capi.check(_capi.lib().awq_round_bias_f32(y32.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), m, n,
                                                    _dt(out), _stream(y32)))
    return out


def gemv_cdna4(x, qweight, scales, scaled_zeros, sz_packed=None, group_size: int = 128):
    _need_gpu(x, qweight, scales, scaled_zeros, sz_packed)
    k = x.shape[-1]
    m = x.numel() // k
    n = qweight.shape[0] * 4
    out = torch.empty(*x.shape[:-1], n, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().awq_w4a16_gemv_cdna4(x.data_ptr(), qweight.data_ptr(), scales.data_ptr(),
                                                      scaled_zeros.data_ptr(),
                                                      sz_packed.data_ptr() if sz_packed is not None else None,
                                                      out.data_ptr(), m, n, k, group_size, _dt(x), _stream(x)))
    return out


def gemm_cdna4(x, qweight, scales, scaled_zeros, bias=None, sz_packed=None, group_size: int = 128, sz_half=None):
    """C-ABI awq_w4a16_forward_cdna4 (awq_w4a16_forward_cdna4_szh when the layer's sz_half side buffer is given): any M (M <= 16 -> GEMV), optional bias."""
    _need_gpu(x, qweight, scales, scaled_zeros, bias, sz_packed, sz_half)
    k = x.shape[-1]
    m = x.numel() // k
    n = qweight.shape[0] * 4
    out = torch.empty(*x.shape[:-1], n, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        ws_bytes = _capi.lib().awq_w4a16_forward_cdna4_workspace_bytes(m, n, k)  # (inside the device context: the plan asks the CURRENT device for its CU count)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device) if ws_bytes else None
        if sz_half is not None:
            _capi.check(_capi.lib().awq_w4a16_forward_cdna4_szh(x.data_ptr(), qweight.data_ptr(), scales.data_ptr(), scaled_zeros.data_ptr(),
                                                                 sz_packed.data_ptr() if sz_packed is not None else None, sz_half.data_ptr(),
                                                                 bias.data_ptr() if bias is not None else None, out.data_ptr(), m, n, k, group_size,
                                                                 _dt(x), ws.data_ptr() if ws is not None else None, ws_bytes, _stream(x)))
            return out
        _capi.check(_capi.lib().awq_w4a16_forward_cdna4(x.data_ptr(), qweight.data_ptr(), scales.data_ptr(),
                                                         scaled_zeros.data_ptr(),
                                                         sz_packed.data_ptr() if sz_packed is not None else None,
                                                         bias.data_ptr() if bias is not None else None,
                                                         out.data_ptr(), m, n, k, group_size, _dt(x),
                                                         ws.data_ptr() if ws is not None else None, ws_bytes, _stream(x)))
    return out


def mlp_gate_up_cdna4(x, qweight_gate_up, sz_packed, group_size: int = 128):
    """C-ABI awq_w4a16_mlp_gate_up_cdna4: silu(x.Wg^T) * (x.Wu^T) in one launch; qweight_gate_up = the gate and up
    cdna4 buffers stacked along N, sz_packed from the equally stacked scales.  1 <= M <= 8, bf16."""
    _need_gpu(x, qweight_gate_up, sz_packed)
    k = x.shape[-1]
    m = x.numel() // k
    n2 = qweight_gate_up.shape[0] * 4
    out = torch.empty(*x.shape[:-1], n2 // 2, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().awq_w4a16_mlp_gate_up_cdna4(x.data_ptr(), qweight_gate_up.data_ptr(), sz_packed.data_ptr(),
                                                             out.data_ptr(), m, n2, k, group_size, _dt(x), _stream(x)))
    return out


def mlp_gate_up_forward_cdna4(x, qweight_interleaved, sz_packed, sz_half=None, group_size: int = 128):
    """C-ABI awq_w4a16_mlp_gate_up_forward_cdna4: QuantLlamaMLP.our_llama_mlp for any row count on the 8 + 8 interleaved pair."""
    _need_gpu(x, qweight_interleaved, sz_packed, sz_half)
    k = x.shape[-1]
    m = x.numel() // k
    n2 = qweight_interleaved.shape[0] * 4
    out = torch.empty(*x.shape[:-1], n2 // 2, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        ws_bytes = _capi.lib().awq_w4a16_mlp_gate_up_forward_cdna4_workspace_bytes(m, n2, k)
        ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=x.device) if ws_bytes else None
        _capi.check(_capi.lib().awq_w4a16_mlp_gate_up_forward_cdna4_ws(x.data_ptr(), qweight_interleaved.data_ptr(), sz_packed.data_ptr(),
                                                                        sz_half.data_ptr() if sz_half is not None else None, out.data_ptr(),
                                                                        m, n2, k, group_size, _dt(x), ws.data_ptr() if ws is not None else None,
                                                                        ws_bytes, _stream(x)))
    return out


def rmsnorm(x, gamma, eps: float):
    """C-ABI awq_rmsnorm: T((float(x) * rsqrt(mean(x^2) + eps)) * float(gamma)) for every row of x [.., k] (layernorm.cu:39-61)."""
    _need_gpu(x, gamma)
    k = x.shape[-1]
    out = torch.empty_like(x)
    if x.numel() == 0:  # (an empty batch: as the torch binding, nothing to launch)
        return out
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().awq_rmsnorm(x.data_ptr(), gamma.data_ptr(), float(eps), out.data_ptr(), x.numel() // k, k, _dt(x), _stream(x)))
    return out


def rmsnorm_forward_cdna4(x, gamma, eps: float, qweight, sz_packed, bias=None, fused_gate_up: bool = False, group_size: int = 128):
    """C-ABI awq_w4a16_rmsnorm_forward_cdna4: T5/Llama RMSNorm (FTLlamaRMSNorm, fused_norm.py:7-21) fused in front of the
    quantised linear -- or, with fused_gate_up, of the gate/up pair + SiLU*mul.  x: un-normalised [.., K], 1 <= M <= 4."""
    _need_gpu(x, gamma, qweight, sz_packed)
    k = x.shape[-1]
    m = x.numel() // k
    n = qweight.shape[0] * 4
    out = torch.empty(*x.shape[:-1], n // 2 if fused_gate_up else n, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().awq_w4a16_rmsnorm_forward_cdna4(x.data_ptr(), gamma.data_ptr(), float(eps), qweight.data_ptr(),
                                                                 sz_packed.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                                 out.data_ptr(), m, n, k, group_size, _dt(x), 1 if fused_gate_up else 0,
                                                                 _stream(x)))
    return out


# ---- W3 ("w3c" tiles, bf16) ----

def pack_w3(q_u8):
    """uint8 [N, K] (0..7) -> int16 [N/4, 3K/4] w3c tiles (GPU kernel)."""
    _need_gpu(q_u8)
    assert q_u8.dtype == torch.uint8
    n, k = q_u8.shape
    out = torch.empty(n // 4, k * 3 // 4, dtype=torch.int16, device=q_u8.device)
    with torch.cuda.device(q_u8.device):
        _capi.check(_capi.lib().awq_pack_w3(q_u8.data_ptr(), out.data_ptr(), n, k, _stream(q_u8)))
    return out


def unpack_w3(qweight_w3):
    _need_gpu(qweight_w3)
    n, k = qweight_w3.shape[0] * 4, qweight_w3.shape[1] * 4 // 3
    out = torch.empty(n, k, dtype=torch.uint8, device=qweight_w3.device)
    with torch.cuda.device(qweight_w3.device):
        _capi.check(_capi.lib().awq_unpack_w3(qweight_w3.data_ptr(), out.data_ptr(), n, k, _stream(qweight_w3)))
    return out


def dequant_w3(qweight_w3, scales, scaled_zeros, group_size: int = 128):
    _need_gpu(qweight_w3, scales, scaled_zeros)
    n, k = qweight_w3.shape[0] * 4, qweight_w3.shape[1] * 4 // 3
    out = torch.empty(n, k, dtype=scales.dtype, device=qweight_w3.device)
    with torch.cuda.device(qweight_w3.device):
        _capi.check(_capi.lib().awq_dequant_w3(qweight_w3.data_ptr(), scales.data_ptr(), scaled_zeros.data_ptr(),
                                                out.data_ptr(), n, k, group_size, _dt(scales), _stream(qweight_w3)))
    return out


def forward_w3(x, qweight_w3, scales, scaled_zeros, sz_packed, bias=None, group_size: int = 128):
    """C-ABI awq_w3a16_forward: any M; every kernel reads the 3-bit tiles natively (the workspace is the optional split-K scratch)."""
    _need_gpu(x, qweight_w3, scales, scaled_zeros, sz_packed, bias)
    k = x.shape[-1]
    m = x.numel() // k
    n = qweight_w3.shape[0] * 4
    out = torch.empty(*x.shape[:-1], n, dtype=x.dtype, device=x.device)
    L = _capi.lib()
    with torch.cuda.device(x.device):
        wsb = L.awq_w3a16_forward_workspace_bytes(m, n, k)
        ws = torch.empty(wsb, dtype=torch.uint8, device=x.device) if wsb else None
        _capi.check(L.awq_w3a16_forward(x.data_ptr(), qweight_w3.data_ptr(), scales.data_ptr(), scaled_zeros.data_ptr(),
                                        sz_packed.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                        m, n, k, group_size, _dt(x), ws.data_ptr() if wsb else None, wsb, _stream(x)))
    return out


def mlp_gate_up_forward_w3(x, qweight_w3_interleaved, sz_packed, group_size: int = 128):
    """C-ABI awq_w3a16_mlp_gate_up_forward: QuantLlamaMLP.our_llama_mlp on 3-bit projections (rows interleaved 8 + 8, w3c tiles), any row count."""
    _need_gpu(x, qweight_w3_interleaved, sz_packed)
    k = x.shape[-1]
    m = x.numel() // k
    n2 = qweight_w3_interleaved.shape[0] * 4
    out = torch.empty(*x.shape[:-1], n2 // 2, dtype=x.dtype, device=x.device)
    L = _capi.lib()
    with torch.cuda.device(x.device):
        wsb = L.awq_w3a16_mlp_gate_up_forward_workspace_bytes(m, n2, k)
        ws = torch.empty(wsb // 4, dtype=torch.float32, device=x.device) if wsb else None
        _capi.check(L.awq_w3a16_mlp_gate_up_forward(x.data_ptr(), qweight_w3_interleaved.data_ptr(), sz_packed.data_ptr(), out.data_ptr(), m, n2, k,
                                                     group_size, _dt(x), ws.data_ptr() if ws is not None else None, wsb, _stream(x)))
    return out


def partial_w3(x, qweight_w3, sz_packed, group_size: int = 128):
    """C-ABI awq_w3a16_partial: the K shard's product of a 3-bit layer as fp32 [..., N], unrounded, no bias (tensor-parallel row split)."""
    _need_gpu(x, qweight_w3, sz_packed)
    k = x.shape[-1]
    m = x.numel() // k
    n = qweight_w3.shape[0] * 4
    out = torch.empty(*x.shape[:-1], n, dtype=torch.float32, device=x.device)
    if m == 0:
        return out
    with torch.cuda.device(x.device):
        _capi.check(_capi.lib().awq_w3a16_partial(x.data_ptr(), qweight_w3.data_ptr(), sz_packed.data_ptr(), out.data_ptr(), m, n, k,
                                                   group_size, _dt(x), _stream(x)))
    return out


# ---- grouped (per-expert) GEMM for MoE layers ----

def moe_gemm(x_sorted, qweight, scales, scaled_zeros, expert_offsets, layout: str = "v2", group_size: int = 128):
    """C-ABI awq_w4a16_moe_gemm.  x_sorted [T, K] (tokens sorted by expert), qweight int16 [E, N/4, K],
    scales / scaled_zeros T [E, Gpad, N], expert_offsets int32 [E + 1] on the device -> out [T, N]."""
    _need_gpu(x_sorted, qweight, scales, scaled_zeros, expert_offsets)
    assert expert_offsets.dtype == torch.int32 and qweight.dim() == 3 and scales.dim() == 3
    e, n, k = qweight.shape[0], qweight.shape[1] * 4, qweight.shape[2]
    t = x_sorted.shape[0]
    out = torch.empty(t, n, dtype=x_sorted.dtype, device=x_sorted.device)
    with torch.cuda.device(x_sorted.device):
        _capi.check(_capi.lib().awq_w4a16_moe_gemm(x_sorted.data_ptr(), qweight.data_ptr(), scales.data_ptr(),
                                                    scaled_zeros.data_ptr(), expert_offsets.data_ptr(), out.data_ptr(), t, e,
                                                    n, k, scales.shape[1], group_size, _dt(x_sorted),
                                                    1 if layout == "cdna4" else 0, _stream(x_sorted)))
    return out


def moe_forward_cdna4(x_sorted, qweight, scales, scaled_zeros, sz_packed, expert_offsets, group_size: int = 128, sz_half=None):
    """C-ABI awq_w4a16_moe_forward_cdna4(_szh): grouped GEMV for <= 8 sorted rows (decode), grouped GEMM otherwise; sz_half = the experts' stacked
    sz_half side buffers (every expert exact) for the f16-mantissa dequant form of the grouped tile launch."""
    _need_gpu(x_sorted, qweight, scales, scaled_zeros, sz_packed, expert_offsets, sz_half)
    assert expert_offsets.dtype == torch.int32 and qweight.dim() == 3 and sz_packed.dtype == torch.int32
    e, n, k = qweight.shape[0], qweight.shape[1] * 4, qweight.shape[2]
    t = x_sorted.shape[0]
    out = torch.empty(t, n, dtype=x_sorted.dtype, device=x_sorted.device)
    with torch.cuda.device(x_sorted.device):
        _capi.check(_capi.lib().awq_w4a16_moe_forward_cdna4_szh(x_sorted.data_ptr(), qweight.data_ptr(), scales.data_ptr(),
                                                                 scaled_zeros.data_ptr(), sz_packed.data_ptr(),
                                                                 sz_half.data_ptr() if sz_half is not None else None,
                                                                 expert_offsets.data_ptr(), out.data_ptr(), t, e, n, k,
                                                                 scales.shape[1], group_size, _dt(x_sorted), _stream(x_sorted)))
    return out


def moe_mlp_gate_up_cdna4(x_sorted, qweight_interleaved, scales, scaled_zeros, sz_packed, expert_offsets, group_size: int = 128, sz_half=None):
    """C-ABI awq_w4a16_moe_mlp_gate_up_cdna4: silu(x . W1_e^T) * (x . W3_e^T) for tokens sorted by expert, every expert's w1 / w3 rows
    interleaved 8 + 8 per slab (qweight int16 [E, 2F/4, K]); out [T, F].  One grouped launch from 256 sorted rows on."""
    _need_gpu(x_sorted, qweight_interleaved, scales, scaled_zeros, sz_packed, expert_offsets, sz_half)
    e, n2, k = qweight_interleaved.shape[0], qweight_interleaved.shape[1] * 4, qweight_interleaved.shape[2]
    t = x_sorted.shape[0]
    out = torch.empty(t, n2 // 2, dtype=x_sorted.dtype, device=x_sorted.device)
    # the fused grouped launch serves >= 256 sorted rows and needs no scratch; below that -- and whenever the launch declines a large call
    # (t * k or n * k / 8 beyond 2^31, knob moe_v6 = 0: AWQ_ERR_WORKSPACE) -- the unfused route wants a [T, 2F] buffer: allocate and retry once
    scratch = torch.empty(t, n2, dtype=x_sorted.dtype, device=x_sorted.device) if 0 < t < 256 else None
    with torch.cuda.device(x_sorted.device):
        for attempt in (0, 1):
            rc = _capi.lib().awq_w4a16_moe_mlp_gate_up_cdna4_szh(
                x_sorted.data_ptr(), qweight_interleaved.data_ptr(), scales.data_ptr(), scaled_zeros.data_ptr(), sz_packed.data_ptr(),
                sz_half.data_ptr() if sz_half is not None else None,
                expert_offsets.data_ptr(), out.data_ptr(), scratch.data_ptr() if scratch is not None else None,
                scratch.numel() * 2 if scratch is not None else 0, t, e, n2, k, scales.shape[1], group_size, _dt(x_sorted), _stream(x_sorted))
            if rc == _capi.AWQ_ERR_WORKSPACE and scratch is None and attempt == 0 and t > 0:
                scratch = torch.empty(t, n2, dtype=x_sorted.dtype, device=x_sorted.device)
                continue
            _capi.check(rc)
            break
    return out


def silu_mul(gate, up):
    """C-ABI awq_silu_mul: T(T(silu(gate)) * up), elementwise (fused_mlp.py:79-82)."""
    _need_gpu(gate, up)
    if gate.shape != up.shape or gate.dtype != up.dtype or gate.numel() % 8:
        raise ValueError("silu_mul: gate and up must have the same shape / dtype and a multiple of 8 elements")
    out = torch.empty_like(gate)
    with torch.cuda.device(gate.device):
        _capi.check(_capi.lib().awq_silu_mul(gate.data_ptr(), up.data_ptr(), out.data_ptr(), gate.numel(), _dt(gate), _stream(gate)))
    return out


def pair_lost_count(device=None) -> int:
    """C-ABI awq_w4a16_gemm_cdna4_pair_lost: blocks of the block-pair K split (down_proj-shaped prefill launches) of `device` that gave up waiting
    for their partner since the library was loaded -- their outputs are NaN.  0 on a healthy run; synchronises with the device.  A serving loop
    that shares the GPU (CU masks, other tenants) polls this between batches and sets the knob `gemm_v6_pair` = 0 when it ever moves."""
    import ctypes
    if not torch.cuda.is_available():
        raise RuntimeError("pair_lost_count needs the GPU the library runs on")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    c = ctypes.c_uint(0)
    with torch.cuda.device(dev):
        _capi.check(_capi.lib().awq_w4a16_gemm_cdna4_pair_lost(ctypes.byref(c)))
    return int(c.value)
