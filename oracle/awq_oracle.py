"""CPU restatement of mit-han-lab/llm-awq's W4A16 ``WQLinear`` hot path.

TEST INFRASTRUCTURE ONLY -- never imported by the product path (see oracle/__init__.py).

Every function cites the reference file:line (relative to the reference repo root) that it
restates.  The restatement is index-arithmetic based (closed forms derived in SURVEY.md
section 8(a)), not a transcription of the reference's reshape/transposes, so that agreement
with the reference (tests/golden/*, produced by oracle/gen_golden.py from the reference's own
Python) is a real check.

Parity status: the reference ships no tests / golden vectors for this path,
and its CUDA kernels cannot be built here (inline PTX, no nvcc).  The oracle is therefore
PINNED against outputs of the reference's own *Python* side (pack_intweight, from_linear,
pseudo_quantize_tensor, the offline repacker) executed in the authoring container, and the
kernel semantics (dequant = one fma rounded to T, fp32 accumulate) are restated from
gemv_cuda.cu / gemm_cuda.cu / dequantize.cuh by reading them ("kernel parity unpinned by
execution").
"""
from __future__ import annotations

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# shape helpers -- awq/quantize/qmodule.py:7-23
# --------------------------------------------------------------------------------------


def ceil_div(c: int, d: int) -> int:
    """qmodule.py:7-8 (`make_divisible` is a ceil-div despite its name)."""
    return -(-c // d)


def zeros_width(in_features: int, group_size: int = 128, pack_num: int = 8) -> int:
    """qmodule.py:11-23: number of int32 zero-point words per row in the v1 format; the v2
    scales/scaled_zeros buffers have ``zeros_width * 8`` rows (zero padded)."""
    if group_size >= 128:
        mult = 1
    elif group_size == 64:
        mult = 2
    elif group_size == 32:
        mult = 4
    else:
        raise NotImplementedError(group_size)
    w = ceil_div(in_features // group_size, pack_num)
    return ceil_div(w, mult) * mult


def padded_groups(in_features: int, group_size: int = 128) -> int:
    return zeros_width(in_features, group_size) * 8


# --------------------------------------------------------------------------------------
# v2 ("gemv_new"/"gemm_new") int4 interleave -- qmodule.py:26-65
# --------------------------------------------------------------------------------------


def v2_position(n, k):
    """Where does logical weight Q[n, k] live in the v2 packed tensor ``int16 [N/4, K]``?

    Returns (row, col, nib): the int16 element [row, col] and the nibble index (bits
    4*nib .. 4*nib+3).  Closed form of qmodule.py:31-57 (three permutations + 4-nibble pack):
      * rows are interleaved by 4 at a granularity of 64 k (qmodule.py:43-50);
      * inside each 32-k chunk, the nibble stream order t = 8*a + u holds
        k_local = 8*(u % 4) + 2*a + (u // 4)  (qmodule.py:31-40 composed).
    """
    n = np.asarray(n)
    k = np.asarray(k)
    row, rr = n // 4, n % 4
    kb, ch, kl = k // 64, (k % 64) // 32, k % 32
    u = (kl // 8) + 4 * (kl % 2)
    a = (kl % 8) // 2
    p = rr * 64 + ch * 32 + 8 * a + u  # nibble position inside the 256-nibble block
    return row, kb * 64 + p // 4, p % 4


_ROW_CHUNK = 512  # rows per pass of the index-map pack / unpack (a multiple of 16): bounds the index arrays at full model sizes


def _row_chunks(N, step=_ROW_CHUNK):
    for n0 in range(0, N, step):
        yield n0, min(N, n0 + step)


def pack_v2(q) -> np.ndarray:
    """Logical ints ``[N, K]`` (0..15) -> v2 ``int16 [N/4, K]`` (qmodule.py:26-65).  Row-chunked (a packed row only
    holds nibbles of its own four logical rows), so Llama-3-70B-sized matrices stay within a few hundred MB."""
    q = np.asarray(q)
    N, K = q.shape
    assert N % 4 == 0 and K % 64 == 0
    out = np.zeros((N // 4, K), dtype=np.uint16)
    kk1 = np.arange(K, dtype=np.int32)
    for n0, n1 in _row_chunks(N):
        nn, kk = np.meshgrid(np.arange(n0, n1, dtype=np.int32), kk1, indexing="ij")
        row, col, nib = v2_position(nn, kk)
        val = ((q[n0:n1].astype(np.uint16) & 0xF) << (4 * nib).astype(np.uint16)).astype(np.uint16)
        # every (row, col) receives exactly four nibbles (nib = 0..3) with disjoint bits: OR == ADD, one pass per nibble
        for j in range(4):
            m = nib == j
            out[row[m], col[m]] |= val[m]
    return out.view(np.int16)


def unpack_v2(qweight) -> np.ndarray:
    """v2 ``int16 [N/4, K]`` -> logical ``uint8 [N, K]``: inverse of :func:`pack_v2`; this is
    what dequantize.cuh:18-123 + the shuffle at gemv_cuda.cu:150-174 compute per thread."""
    qw = np.asarray(qweight).view(np.uint16)
    R, K = qw.shape
    N = R * 4
    out = np.empty((N, K), dtype=np.uint8)
    kk1 = np.arange(K, dtype=np.int32)
    for n0, n1 in _row_chunks(N):
        nn, kk = np.meshgrid(np.arange(n0, n1, dtype=np.int32), kk1, indexing="ij")
        row, col, nib = v2_position(nn, kk)
        out[n0:n1] = ((qw[row, col] >> (4 * nib).astype(np.uint16)) & 0xF).astype(np.uint8)
    return out


# --------------------------------------------------------------------------------------
# v1 format -- tinychat/offline-weight-repacker.py:8-19, 64-73
# --------------------------------------------------------------------------------------


def unpack_v1(qweight_i32) -> np.ndarray:
    """v1 ``int32 [N, K/8]``: nibble kk%8 of word kk//8 is Q[n, kk] (repacker :8-19)."""
    w = np.asarray(qweight_i32).view(np.uint32).astype(np.int64)
    N, W = w.shape
    sh = 4 * np.arange(8)
    return ((w[:, :, None] >> sh[None, None, :]) & 0xF).reshape(N, W * 8).astype(np.uint8)


def pack_v1(q) -> np.ndarray:
    q = np.asarray(q).astype(np.int64)
    N, K = q.shape
    sh = 4 * np.arange(8)
    w = ((q.reshape(N, K // 8, 8) & 0xF) << sh[None, None, :]).sum(-1)
    return w.astype(np.uint32).view(np.int32)


def scaled_zeros_from_v1(scales: torch.Tensor, qzeros_i32, zp_shift: float = 0.0) -> torch.Tensor:
    """repacker :64-73 with the ``zp_shift=0`` used at :142: ``-(scales*zero + zp_shift*scales)``
    evaluated in the dtype of ``scales``; scales is v1-shaped ``[N, Gpad]``."""
    z = unpack_v1(qzeros_i32)  # [N, Gpad] same sequential nibble order over groups
    G = scales.shape[1]
    zt = torch.from_numpy(z[:, :G].astype(np.float32)).to(scales.dtype)
    sz = scales * zt
    return -(sz + (zp_shift * scales))


def repack_v1_to_v2(qweight_i32, scales: torch.Tensor, qzeros_i32):
    """repacker :111-152: returns (qweight_v2 int16 [N/4,K], scales [Gpad,N], scaled_zeros [Gpad,N])."""
    q = unpack_v1(qweight_i32)
    qw2 = torch.from_numpy(pack_v2(q))
    s2 = scales.transpose(1, 0).contiguous()
    sz2 = scaled_zeros_from_v1(scales, qzeros_i32, 0.0).transpose(1, 0).contiguous()
    return qw2, s2, sz2


# --------------------------------------------------------------------------------------
# quantisation grid -- awq/quantize/quantizer.py:61-103
# --------------------------------------------------------------------------------------


def pseudo_quantize(w: torch.Tensor, n_bit: int = 4, group_size: int = 128):
    """quantizer.py:61-103 with zero_point=True, evaluated in w's dtype.
    Returns (w_fake [N,K], scales [N,K/G], zeros [N,K/G])."""
    shape = w.shape
    g = w.reshape(-1, group_size) if group_size > 0 else w.reshape(-1, shape[-1])
    hi = g.amax(dim=1, keepdim=True)
    lo = g.amin(dim=1, keepdim=True)
    qmax = 2 ** n_bit - 1
    scales = (hi - lo).clamp(min=1e-5) / qmax
    zeros = (-torch.round(lo / scales)).clamp_(0, qmax)
    fake = (torch.clamp(torch.round(g / scales) + zeros, 0, qmax) - zeros) * scales
    return fake.reshape(shape), scales.view(shape[0], -1), zeros.view(shape[0], -1)


def intweight_from_fake(w_fake: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, group_size: int):
    """qmodule.py:155-185: recover the integers column by column, in the tensors' dtype,
    without clamping: round((w + zeros*scales) / scales)."""
    sz = zeros * scales
    K = w_fake.shape[1]
    gi = torch.arange(K) // group_size
    return torch.round((w_fake + sz[:, gi]) / scales[:, gi]).to(torch.int32)


def wq_buffers_from_fake(w_fake, scales, zeros, group_size, n_bit: int = 4):
    """qmodule.py:139-199 (`from_linear`): (qweight int16 [N/4,K], scales [Gpad,N], scaled_zeros [Gpad,N])."""
    assert n_bit == 4
    N, K = w_fake.shape
    T = scales.dtype
    gp = padded_groups(K, group_size)
    qs = torch.zeros((N, gp), dtype=T)
    qs[:, : scales.shape[1]] = scales
    iw = intweight_from_fake(w_fake, scales, zeros, group_size)
    qweight = torch.from_numpy(pack_v2(iw.numpy()))
    zi = zeros.to(torch.int32)
    sz = torch.zeros_like(qs)
    sz[:, : scales.shape[1]] = -(qs[:, : scales.shape[1]] * zi.to(torch.float32)).to(T)
    return qweight, qs.t().contiguous(), sz.t().contiguous(), iw


def quantize_linear(w: torch.Tensor, dtype=torch.float16, n_bit: int = 4, group_size: int = 128):
    """The `real_quantize_model_weight` recipe (quantizer.py:143-157) for one weight matrix:
    cast to T, pseudo-quantise in T, build the v2 buffers.  Returns a dict."""
    wt = w.to(dtype)
    fake, s, z = pseudo_quantize(wt, n_bit, group_size)
    qweight, scales, scaled_zeros, iw = wq_buffers_from_fake(fake, s, z, group_size, 4) if n_bit == 4 else (None,) * 4
    return dict(w_fake=fake, s=s, z=z, qweight=qweight, scales=scales, scaled_zeros=scaled_zeros, intweight=iw)


# --------------------------------------------------------------------------------------
# kernel-faithful forward -- gemv_cuda.cu:128-228, gemm_cuda.cu:263-310 + bf16 mma (fp32 acc)
# --------------------------------------------------------------------------------------


def dequant_weight(q_int, scales: torch.Tensor, scaled_zeros: torch.Tensor, group_size: int = 128) -> torch.Tensor:
    """W_T[n,k] = round_T( fma(q[n,k], scales[k/G, n], scaled_zeros[k/G, n]) ).

    gemv_cuda.cu:159-166 / gemm_cuda.cu:304-307: one `__hfma2` in T on the exact integer
    produced by dequantize.cuh (unsigned 0..15, no -8 shift).  q*s is exact in fp32
    (4 x 11 significant bits) and the sum needs <= 24 bits because |sz| = s*z with z<=15, so an
    fp32 multiply-add followed by one rounding to T equals the single-rounded fma."""
    T = scales.dtype
    q = torch.as_tensor(np.asarray(q_int).astype(np.float32))
    N, K = q.shape
    gi = torch.arange(K) // group_size
    s = scales.float()[gi, :].t()  # [N, K]
    z = scaled_zeros.float()[gi, :].t()
    return (q * s + z).to(T)


def wqlinear_forward(x: torch.Tensor, qweight, scales, scaled_zeros, bias=None, group_size: int = 128,
                     q_int=None) -> torch.Tensor:
    """Oracle of `WQLinear.forward` (qmodule.py:201-224): y = round_T(sum_k fp32(x)*fp32(W_T)) (+ bias in T).

    fp32 accumulation is what the reference's bf16 tensor-core path and the GEMV's final
    reduction use; the fp16 paths' T-precision partial sums are an artefact that is NOT emulated
    (SURVEY.md 8(c))."""
    T = x.dtype
    if q_int is None:
        q_int = unpack_v2(qweight.numpy() if isinstance(qweight, torch.Tensor) else qweight)
    W = dequant_weight(q_int, scales, scaled_zeros, group_size)
    K = x.shape[-1]
    y = (x.reshape(-1, K).float() @ W.float().t())
    y = y.to(T).reshape(*x.shape[:-1], W.shape[0])
    if bias is not None:
        y = y + bias
    return y


def wqlinear_partial_f32(x: torch.Tensor, qweight, scales, scaled_zeros, group_size: int = 128, q_int=None) -> torch.Tensor:
    """The K shard's product of a tensor-parallel row split, as THIS repository defines it (the reference has no multi-GPU path for
    qmodule.py:201-224): the same contraction as `wqlinear_forward` on the shard's k range, left in fp32 -- not rounded to
    T, no bias.  The ranks' partials are summed in fp32 and rounded once, so the sharded layer reproduces the single-device forward to fp32
    reassociation error."""
    if q_int is None:
        q_int = unpack_v2(qweight.numpy() if isinstance(qweight, torch.Tensor) else qweight)
    W = dequant_weight(q_int, scales, scaled_zeros, group_size)
    K = x.shape[-1]
    return (x.reshape(-1, K).float() @ W.float().t()).reshape(*x.shape[:-1], W.shape[0])


def wqlinear_forward_f64(x, q_int, scales, scaled_zeros, group_size=128):
    """Same contraction in float64 on the T-rounded weights: the 'exact' value the fp32
    accumulations approximate (used to size tolerances)."""
    W = dequant_weight(q_int, scales, scaled_zeros, group_size)
    K = x.shape[-1]
    return x.reshape(-1, K).double() @ W.double().t()


# --------------------------------------------------------------------------------------
# the CPU baseline of BASELINE.json: pseudo-quant nn.Linear (quantizer.py:106-122)
# --------------------------------------------------------------------------------------


def pseudo_quant_linear(w: torch.Tensor, n_bit=4, group_size=128, dtype=torch.bfloat16):
    lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=False, dtype=dtype)
    with torch.no_grad():
        lin.weight.data = pseudo_quantize(w.to(dtype), n_bit, group_size)[0]
    return lin


# --------------------------------------------------------------------------------------
# "cdna4" interleave -- THIS repository's MI355X-native int4 layout (no reference counterpart; it is
# what the rewritten repacker emits, see DESIGN.md "cdna4 interleave").  Same size/dtype as v2
# (int16 [N/4, K]), a pure permutation of nibbles:
#   u32 words [N/16][K/128][64 lanes][4 words]; one 1-KiB tile = 16 rows x 128 k (one group).
#   lane = 16*g + 4*nq + r, word a, nibble p  (i = p & 3, hi = p >> 2)  holds
#       Q[n = 16*nb + 4*nq + 2*(i & 1) + hi][k = 128*kg + 32*a + 8*g + 4*(i >> 1) + r]
# so that (word >> 4*i) & 0x000F000F | 0x43004300 is directly an A-operand register of the 16-block MFMA 4x4x4 bf16
# (block = lane // 4 = (k octet g, row quad nq), rows = k, inner = the quad's four n) of the "dequantise on the matrix
# core" step  W^T = (128+Q)^T . diag(s) + (sz-128 s), whose result lands with lane 16*g + n holding k = 32a + 8g + 0..7.
# --------------------------------------------------------------------------------------


def cdna4_position(n, k, K):
    """(word index into the flat u32 buffer, nibble index) of logical weight Q[n, k]."""
    n = np.asarray(n)
    k = np.asarray(k)
    nb, c = n // 16, n % 16
    g, j = c // 4, c % 4
    kg, kk = k // 128, k % 128
    a, r32 = kk // 32, kk % 32
    b8, e = r32 // 8, r32 % 8
    th, rr = e // 4, e % 4
    lane = 16 * b8 + 4 * g + rr  # g = row quad of the slab, b8 = k octet of the 32-k word range
    i = 2 * th + (j >> 1)
    p = i + 4 * (j & 1)
    word = ((nb * (K // 128) + kg) * 64 + lane) * 4 + a
    return word, p


def pack_cdna4(q) -> np.ndarray:
    """Logical ints [N, K] (0..15) -> cdna4 interleave, returned as int16 [N/4, K] (same shape as v2)."""
    q = np.asarray(q)
    N, K = q.shape
    assert N % 16 == 0 and K % 128 == 0
    out = np.zeros(N * K // 8, dtype=np.uint32)
    kk1 = np.arange(K, dtype=np.int64)
    for n0, n1 in _row_chunks(N):  # a slab's tiles only hold nibbles of its own 16 rows
        nn, kk = np.meshgrid(np.arange(n0, n1, dtype=np.int64), kk1, indexing="ij")
        word, p = cdna4_position(nn, kk, K)
        val = (q[n0:n1].astype(np.uint32) & 0xF) << (4 * p).astype(np.uint32)
        for j in range(8):  # each word receives one nibble per p: disjoint bits, one pass per nibble index
            m = p == j
            out[word[m]] |= val[m]
    return out.view(np.int16).reshape(N // 4, K)


def unpack_cdna4(qweight) -> np.ndarray:
    w = np.ascontiguousarray(np.asarray(qweight)).view(np.uint32).reshape(-1)
    N, K = qweight.shape[0] * 4, qweight.shape[1]
    out = np.empty((N, K), dtype=np.uint8)
    kk1 = np.arange(K, dtype=np.int64)
    for n0, n1 in _row_chunks(N):
        nn, kk = np.meshgrid(np.arange(n0, n1, dtype=np.int64), kk1, indexing="ij")
        word, p = cdna4_position(nn, kk, K)
        out[n0:n1] = ((w[word] >> (4 * p).astype(np.uint32)) & 0xF).astype(np.uint8)
    return out


def v2_to_cdna4(qweight_v2) -> np.ndarray:
    return pack_cdna4(unpack_v2(qweight_v2))


def pack_sz_half(scales: torch.Tensor, scaled_zeros: torch.Tensor, K: int):
    """THIS repository's "sz_half" side buffer of the decode kernels (no reference counterpart; DESIGN.md "f16-mantissa dequant"):
    u32 [N/16][K/128][16] = {f16(s') | f16(sz) << 16} with s' = s for rows n % 4 < 2 and s / 16 for the others.
    Returns (int32 array, exact) -- exact is False when a value is not representable as a normal f16 number (or 0)."""
    G = K // 128
    s = scales.float()[:G].t().contiguous()        # [N, G]
    z = scaled_zeros.float()[:G].t().contiguous()
    N = s.shape[0]
    sp = torch.where((torch.arange(N) % 4 >= 2)[:, None], s * 0.0625, s)
    sh, zh = sp.to(torch.float16), z.to(torch.float16)

    def ok(v, h):
        return bool(((h.float() == v) & ((v == 0) | (v.abs() >= 2.0 ** -14))).all())

    exact = ok(sp, sh) and ok(z, zh)
    lo = sh.view(torch.int16).numpy().view(np.uint16).astype(np.uint32)
    hi = zh.view(torch.int16).numpy().view(np.uint16).astype(np.uint32)
    packed = (lo | (hi << 16)).reshape(N // 16, 16, G).transpose(0, 2, 1)  # [slab][group][row in slab]
    return np.ascontiguousarray(packed).view(np.int32), exact


# --------------------------------------------------------------------------------------
# W3 ("w3c" tiles) -- THIS repository's 3-bit format; the reference has none (qmodule.py:82-83 raises for
# w_bit != 4) and only defines the 3-bit GRID: pseudo_quantize_tensor(n_bit=3), quantizer.py:61-103, restated by
# pseudo_quantize() above and pinned by tests/golden/pseudo_w3.npz.  A tile is the cdna4 tile of the same integers
# (values 0..7) whose fourth word (a = 3) is folded into the free bit 3 of every nibble of words 0..2:
# bit 3 of nibble p of stored word c = bit c of nibble p of logical word 3.  768 B per tile, int16 [N/4, 3K/4] overall.
# --------------------------------------------------------------------------------------


def pack_w3(q) -> np.ndarray:
    q = np.asarray(q).astype(np.int64)
    assert q.max(initial=0) <= 7 and q.min(initial=0) >= 0
    N, K = q.shape
    w4 = np.ascontiguousarray(pack_cdna4(q)).view(np.uint32).reshape(-1, 4).astype(np.int64)  # [tiles*64, 4]
    out = np.empty((w4.shape[0], 3), dtype=np.int64)
    for c in range(3):
        bit_c = (w4[:, 3] >> c) & 0x11111111      # bit c of every nibble of word 3, at nibble bit 0
        out[:, c] = (w4[:, c] & 0x77777777) | (bit_c << 3)
    return out.astype(np.uint32).view(np.int16).reshape(N // 4, K * 3 // 4)


def unpack_w3(qweight_w3) -> np.ndarray:
    w3 = np.ascontiguousarray(np.asarray(qweight_w3)).view(np.uint32).reshape(-1, 3).astype(np.int64)
    N, K = qweight_w3.shape[0] * 4, qweight_w3.shape[1] * 4 // 3
    w4 = np.zeros((w3.shape[0], 4), dtype=np.int64)
    for c in range(3):
        w4[:, c] = w3[:, c] & 0x77777777
        w4[:, 3] |= ((w3[:, c] >> 3) & 0x11111111) << c
    return unpack_cdna4(w4.astype(np.uint32).view(np.int16).reshape(N // 4, K))


def quantize_linear_w3(w: torch.Tensor, dtype=torch.bfloat16, group_size: int = 128):
    """real-quantise recipe (quantizer.py:143-157 with n_bit = 3) + from_linear's integer recovery and scale layout
    (qmodule.py:155-197) + the w3c packing."""
    wt = w.to(dtype)
    fake, s, z = pseudo_quantize(wt, 3, group_size)
    N, K = wt.shape
    gp = padded_groups(K, group_size)
    qs = torch.zeros((N, gp), dtype=dtype)
    qs[:, : s.shape[1]] = s
    iw = intweight_from_fake(fake, s, z, group_size)
    sz = torch.zeros_like(qs)
    sz[:, : s.shape[1]] = -(qs[:, : s.shape[1]] * z.to(torch.int32).to(torch.float32)).to(dtype)
    return dict(w_fake=fake, s=s, z=z, qweight=torch.from_numpy(pack_w3(iw.numpy())), scales=qs.t().contiguous(),
                scaled_zeros=sz.t().contiguous(), intweight=iw)


# --------------------------------------------------------------------------------------
# RMSNorm in front of the linear: awq/kernels/csrc/layernorm/layernorm.cu:39-61
# (generalT5LayerNorm: no mean subtraction, no bias), called by tinychat/modules/fused_norm.py:7-21.
# --------------------------------------------------------------------------------------
def rmsnorm(x: "torch.Tensor", gamma: "torch.Tensor", eps: float) -> "torch.Tensor":
    """out = T((float(x) * rsqrt(mean(x^2) + eps)) * float(gamma)): fp32 sum of squares (layernorm.cu:48-52; the
    reduction order is the kernel's business -- fp32 either way), rsqrtf(variance / n + eps) (:55), the two multiplies in
    fp32 in that order and ONE rounding to T (:60)."""
    xf = x.float()
    var = (xf * xf).sum(-1, keepdim=True) / xf.shape[-1]
    rstd = torch.rsqrt(var + eps)
    return ((xf * rstd) * gamma.float()).to(x.dtype)
