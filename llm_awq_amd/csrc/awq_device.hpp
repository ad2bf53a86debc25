// Device-side building blocks shared by every kernel of the W4A16 hot path (gfx950 only).
//
//  * v2 int4 interleave addressing (reference: awq/quantize/qmodule.py:26-65, closed form in
//    SURVEY.md 8(a) a2): a lane's 16-byte load = one output row n x one 32-k chunk.
//  * word -> four "pair registers": extraction i of word w yields the two adjacent weights
//    k = 8*i + 2*w + {0,1} of the chunk, so the four words of a chunk give, for every i, eight
//    consecutive k -- exactly one MFMA 16x16x32 operand (4 VGPRs).
//  * dequantisation with the reference's numerics (dequantize.cuh:18-123 + the hfma2 at
//    gemv_cuda.cu:159-166 / gemm_cuda.cu:304-307): W = round_T(q * s + sz), q unsigned 0..15.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace awq {

using u32 = uint32_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

constexpr int kGroup = 128;  // the only group size the reference kernels implement (gemv_cuda.cu:332-335)

// ---------------------------------------------------------------------------------------------
// v2 layout addressing.  qweight is int16 [N/4, K] viewed as u32 words: a packed row has K/2 words;
// every 32-word block holds 64 k of 4 rows: words [8*rr + 4*c, +4) = row 4r+rr, 32-k chunk c.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t v2_chunk_word(int n, int chunk /* = k/32 */, int K) {
  return (size_t)(n >> 2) * (size_t)(K >> 1) + (size_t)(chunk >> 1) * 32 + (n & 3) * 8 + (chunk & 1) * 4;
}

// logical k (inside the 32-k chunk) of nibble `nib` of word `w` (0..3) of a chunk
__device__ __forceinline__ int v2_nibble_k(int w, int nib) { return 8 * (nib & 3) + 2 * w + (nib >> 2); }

// ---------------------------------------------------------------------------------------------
// dtype traits
// ---------------------------------------------------------------------------------------------
struct F16 {
  using elem = _Float16;
  using vec8 = f16x8;
  static constexpr int id = 0;
  // per-(n, group) dequant constants, prepared once per 16-byte chunk
  struct SZ {
    f16x2 s, z;
  };
  static __device__ __forceinline__ SZ make_sz(uint16_t s_bits, uint16_t z_bits) {
    SZ r;
    r.s = __builtin_bit_cast(f16x2, (u32)s_bits * 0x00010001u);
    r.z = __builtin_bit_cast(f16x2, (u32)z_bits * 0x00010001u);
    return r;
  }
  // word -> 4 packed pairs (pair i = weights k=8i+2w, 8i+2w+1), already dequantised & rounded to fp16.
  // Exact int->fp16 via the 0x6400 magic (1024+q) and, for the high nibbles, (1024+16q)/16-64;
  // then ONE packed fma per pair == the reference's __hfma2(q, s, sz).
  static __device__ __forceinline__ void dequant_word(u32 w, const SZ& c, u32 (&out)[4]) {
    const f16x2 k1024 = {(_Float16)1024.f, (_Float16)1024.f};
    const f16x2 k16th = {(_Float16)0.0625f, (_Float16)0.0625f};
    const f16x2 kneg64 = {(_Float16)-64.f, (_Float16)-64.f};
    const u32 w8 = w >> 8;
    f16x2 q0 = __builtin_bit_cast(f16x2, (w & 0x000F000Fu) | 0x64006400u) - k1024;
    f16x2 q1 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, (w & 0x00F000F0u) | 0x64006400u), k16th, kneg64);
    f16x2 q2 = __builtin_bit_cast(f16x2, (w8 & 0x000F000Fu) | 0x64006400u) - k1024;
    f16x2 q3 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, (w8 & 0x00F000F0u) | 0x64006400u), k16th, kneg64);
    out[0] = __builtin_bit_cast(u32, __builtin_elementwise_fma(q0, c.s, c.z));
    out[1] = __builtin_bit_cast(u32, __builtin_elementwise_fma(q1, c.s, c.z));
    out[2] = __builtin_bit_cast(u32, __builtin_elementwise_fma(q2, c.s, c.z));
    out[3] = __builtin_bit_cast(u32, __builtin_elementwise_fma(q3, c.s, c.z));
  }
  static __device__ __forceinline__ f32x4 mfma(const vec8& a, const vec8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint16_t from_float(float f) {
    return __builtin_bit_cast(uint16_t, (_Float16)f);
  }
  static __device__ __forceinline__ float to_float(uint16_t b) { return (float)__builtin_bit_cast(_Float16, b); }
  // ---- matrix-core dequant (cdna4 interleave): A operand = 1024 + q (0x6400 | q), C = sz - 1024 s ----
  static constexpr u32 kDqMagic = 0x64006400u;
  static __device__ __forceinline__ f32x4 mfma4(u32x2 a, u32x2 b, const f32x4& c) {
    typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(f16x4, a), __builtin_bit_cast(f16x4, b), c, 0, 0, 0);
  }
  // {s | sz << 16} -> sz - 1024 s: both products are exact in fp32 and so is their sum (|sz| = s z, z <= 15: 22 bits)
  static __device__ __forceinline__ float dq_offset(u32 sz) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, sz), __builtin_bit_cast(f16x2, 0x3C00E400u), 0.0f, false);
  }
  static __device__ __forceinline__ vec8 pack8(const f32x4& d0, const f32x4& d1) {
    vec8 r = {(_Float16)d0[0], (_Float16)d0[1], (_Float16)d0[2], (_Float16)d0[3],
              (_Float16)d1[0], (_Float16)d1[1], (_Float16)d1[2], (_Float16)d1[3]};
    return r;
  }
  typedef float f32x16_t __attribute__((ext_vector_type(16)));
  static __device__ __forceinline__ f32x16_t mfma32(const vec8& a, const vec8& b, const f32x16_t& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

struct BF16 {
  using elem = __bf16;
  using vec8 = bf16x8;
  static constexpr int id = 1;
  struct SZ {
    float s, z;
  };
  static __device__ __forceinline__ SZ make_sz(uint16_t s_bits, uint16_t z_bits) {
    SZ r;
    r.s = __builtin_bit_cast(float, (u32)s_bits << 16);
    r.z = __builtin_bit_cast(float, (u32)z_bits << 16);
    return r;
  }
  // gfx950 has no packed bf16 fma.  q*s (4 x 8 significant bits) + sz is exactly representable in
  // fp32 (|sz| = s*z, z <= 15 keeps the exponent gap <= 5 bits), so fmaf() is exact and the single
  // RNE rounding of v_cvt_pk_bf16_f32 reproduces the reference's bf16 __hfma2 bit for bit.
  static __device__ __forceinline__ u32 pair(u32 qlo, u32 qhi, const SZ& c) {
    float a = __builtin_fmaf((float)qlo, c.s, c.z);
    float b = __builtin_fmaf((float)qhi, c.s, c.z);
    bf16x2 r = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(u32, r);
  }
  static __device__ __forceinline__ void dequant_word(u32 w, const SZ& c, u32 (&out)[4]) {
    u32 e = w & 0x0F0F0F0Fu;         // nibbles 0,2,4,6 -> bytes 0..3
    u32 o = (w >> 4) & 0x0F0F0F0Fu;  // nibbles 1,3,5,7 -> bytes 0..3
    // keep the byte-lane form so the conversions lower to v_cvt_f32_ubyte{0..3}
    asm volatile("" : "+v"(e), "+v"(o));
    out[0] = pair(e & 0xFFu, (e >> 16) & 0xFFu, c);  // nibbles (0,4)
    out[1] = pair(o & 0xFFu, (o >> 16) & 0xFFu, c);  // nibbles (1,5)
    out[2] = pair((e >> 8) & 0xFFu, e >> 24, c);     // nibbles (2,6)
    out[3] = pair((o >> 8) & 0xFFu, o >> 24, c);     // nibbles (3,7)
  }
  static __device__ __forceinline__ f32x4 mfma(const vec8& a, const vec8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint16_t from_float(float f) {
    return __builtin_bit_cast(uint16_t, (__bf16)f);
  }
  static __device__ __forceinline__ float to_float(uint16_t b) { return __builtin_bit_cast(float, (u32)b << 16); }
  // ---- matrix-core dequant (cdna4 interleave): A operand = 128 + q (0x4300 | q), C = sz - 128 s ----
  static constexpr u32 kDqMagic = 0x43004300u;
  static __device__ __forceinline__ f32x4 mfma4(u32x2 a, u32x2 b, const f32x4& c) {
    typedef short s16x4_t __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4_t, a), __builtin_bit_cast(s16x4_t, b), c, 0, 0, 0);
  }
  // {s | sz << 16} -> sz - 128 s (s * -128 and sz * 1 are exact products and their sum is exactly representable:
  // |sz| = s * z, z <= 15)
  static __device__ __forceinline__ float dq_offset(u32 sz) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, sz), __builtin_bit_cast(bf16x2, 0x3F80C300u), 0.0f, false);
  }
  static __device__ __forceinline__ vec8 pack8(const f32x4& d0, const f32x4& d1) {
    vec8 r = {(__bf16)d0[0], (__bf16)d0[1], (__bf16)d0[2], (__bf16)d0[3],
              (__bf16)d1[0], (__bf16)d1[1], (__bf16)d1[2], (__bf16)d1[3]};
    return r;
  }
  typedef float f32x16_t __attribute__((ext_vector_type(16)));
  static __device__ __forceinline__ f32x16_t mfma32(const vec8& a, const vec8& b, const f32x16_t& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// Dequantise one 16-byte chunk (4 words = 32 k of one row) into four MFMA operands:
// op[j] = weights k0+8j .. k0+8j+7 in natural order.
template <typename DT>
__device__ __forceinline__ void dequant_chunk(const u32x4& w, const typename DT::SZ& c, typename DT::vec8 (&op)[4]) {
  u32 p0[4], p1[4], p2[4], p3[4];
  DT::dequant_word(w.x, c, p0);
  DT::dequant_word(w.y, c, p1);
  DT::dequant_word(w.z, c, p2);
  DT::dequant_word(w.w, c, p3);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u32x4 v = {p0[j], p1[j], p2[j], p3[j]};
    op[j] = __builtin_bit_cast(typename DT::vec8, v);
  }
}

// raw (un-dequantised) nibble extraction in the same pair order: used by the unpack-index check.
__device__ __forceinline__ void unpack_word_pairs(u32 w, u32 (&lo)[4], u32 (&hi)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    lo[i] = (w >> (4 * i)) & 0xFu;
    hi[i] = (w >> (4 * i + 16)) & 0xFu;
  }
}

__device__ __forceinline__ u32x4 ldg_nt_u32x4(const u32* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}


// QuantLlamaMLP's elementwise tail on eight packed T values (fused_mlp.py:79-82): c = T(T(silu(gate)) * up), silu in fp32 on the
// T-rounded gate -- the same roundings as the reference's separate F.silu and multiply, and as the decode epilogues.  This is
// the prefill form: 32 k evaluations per 256 x 256 tile sit in the tile's exposed epilogue, so silu is x * rcp(1 + exp2(-x log2 e))
// on the hardware transcendentals (5 VALU, ~3 ulp of fp32 -- it changes the T rounding of ~0.05 % (bf16) / 0.2 % (fp16) of the
// outputs by one ulp of T against an exact silu, which is also how far torch's own GPU silu sits from its CPU one); libm's expf and
// an IEEE division cost ~40 instructions = +9 % on the gate/up GEMM at M = 2048.  The decode epilogues use the same form (silu_f32).
// The ONE fp32 silu of the library (decode and prefill epilogues alike, so a row's QuantLlamaMLP output does not depend on how many
// rows were batched with it): x * rcp(1 + exp2(-x log2 e)) on the hardware transcendentals.
__device__ __forceinline__ float silu_f32(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}
template <typename DT>
__device__ __forceinline__ u32 silu_mul_pair(u32 gate2, u32 up2) {
  auto one = [](uint16_t gb, uint16_t ub) {
    const float gt = DT::to_float(gb), up = DT::to_float(ub);
    const float sl = DT::to_float(DT::from_float(silu_f32(gt)));
    return (u32)DT::from_float(sl * up);
  };
  return one((uint16_t)(gate2 & 0xFFFFu), (uint16_t)(up2 & 0xFFFFu)) | (one((uint16_t)(gate2 >> 16), (uint16_t)(up2 >> 16)) << 16);
}
template <typename DT>
__device__ __forceinline__ u32x4 silu_mul_octet(const u32x4& g, const u32x4& u) {
  return u32x4{silu_mul_pair<DT>(g.x, u.x), silu_mul_pair<DT>(g.y, u.y), silu_mul_pair<DT>(g.z, u.z), silu_mul_pair<DT>(g.w, u.w)};
}

// =============================================================================================
// "cdna4" interleave (this repository's MI355X-native layout; emitted by the rewritten repacker).
// Same bytes/shape as v2, nibbles permuted so that a 1-KiB tile = 16 rows x 128 k is ONE contiguous
// wave-load and every extraction (word >> 4i) & 0x000F000F | 0x43004300 is directly an A-operand
// register of a v_mfma_f32_4x4x4_16B_bf16 / _f16 (16 independent 4 x 4 x 4 blocks, block = lane / 4) that dequantises
// ON THE MATRIX CORE -- per block (k octet g, row quad nq), rows = 4 k, inner = the quad's 4 rows n':
//      D[k][n] = sum_n' (128 + Q[n'][k]) * (s_n [n'==n])  +  (sz_n - 128 s_n)  =  Q[n][k] s_n + sz_n   (exact in fp32;
//      fp16: magic 0x6400 = 1024 + q and offset 1024 -- 11 x 11 significant bits, still exact)
// lane l = 16 g + 4 nq + r, word a, nibble p (i = p & 3, hi = p >> 2):
//      n = 16 nb + 4 nq + 2 (i & 1) + hi,   k = 128 kg + 32 a + 8 g + 4 (i >> 1) + r
// D comes out with lane (n = l % 16, g = l / 16) holding k = 32 a + 8 g + {0..3} (first MFMA) and
// + {4..7} (second): after v_cvt_pk_bf16_f32 that IS the operand of the matmul MFMA 16x16x32.
// (The 4x4x4 form issues at twice the rate of a 16x16x16 with a diagonal B operand -- measured 3.9 vs 7.5 ns per
// instruction per SIMD, tools/ubench/mfma4x4_probe.hip -- and wastes 3/4 instead of 15/16 of its products.)
// =============================================================================================
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ size_t cdna4_tile_word(int nb, int kg, int nit) { return ((size_t)nb * nit + kg) * 256; }

template <typename DT>
struct Cdna4DequantT {
  using vec8 = typename DT::vec8;
  u32 m01, m23;  // lane masks selecting where s_n sits in the diagonal B operand
  u32 kMagic, kMask;
  __device__ __forceinline__ void init(int lane, u32 nibble_mask = 0x000F000Fu) {
    const int pos = lane & 3;  // B operand of block lane / 4: column lane % 4 = this lane's row n, nonzero at inner index pos
    m01 = pos == 0 ? 0x0000FFFFu : (pos == 1 ? 0xFFFF0000u : 0u);
    m23 = pos == 2 ? 0x0000FFFFu : (pos == 3 ? 0xFFFF0000u : 0u);
    // gfx950 VOP3 takes no 32-bit literal: park the magic in a VGPR and the mask in an SGPR so that
    // (w & mask) | magic is ONE v_and_or_b32 instead of v_and_b32 + v_or_b32 with literals
    kMagic = DT::kDqMagic;
    kMask = nibble_mask;  // 0x00070007 for W3 tiles (bit 3 of every nibble carries the folded fourth word)
    asm volatile("" : "+v"(kMagic));
    asm volatile("" : "+s"(kMask));
  }
  // the two dequant MFMAs of one word, and the rounding of their results, as separate halves for kernels that put other
  // work in between
  struct Pending {
    f32x4 d0, d1;
  };
  __device__ __forceinline__ Pending word_issue(u32 w, u32 b01, u32 b23, float cv) const {
    const u32x2 a0 = {(w & kMask) | kMagic, ((w >> 4) & kMask) | kMagic};
    const u32x2 a1 = {((w >> 8) & kMask) | kMagic, ((w >> 12) & kMask) | kMagic};
    const u32x2 b = {b01, b23};
    const f32x4 c = {cv, cv, cv, cv};
    Pending p;
    p.d0 = DT::mfma4(a0, b, c);
    p.d1 = DT::mfma4(a1, b, c);
    return p;
  }
  static __device__ __forceinline__ vec8 word_finish(const Pending& p) { return DT::pack8(p.d0, p.d1); }
  // one word (two 4x4x4 16-block MFMAs) -> one 8-element operand: W[n = lane%16][k = 32a + 8g + 0..7]
  __device__ __forceinline__ vec8 word(u32 w, u32 b01, u32 b23, float cv) const { return word_finish(word_issue(w, b01, b23, cv)); }
  // whole 1-KiB tile -> 4 operands (op[a] covers k = 32a + 8g + 0..7 of the tile's 128 k)
  __device__ __forceinline__ void tile(const u32x4& w, uint16_t s_bits, uint16_t z_bits, vec8 (&op)[4]) const {
    tile_packed(w, (u32)s_bits | ((u32)z_bits << 16), op);
  }
  // word-by-word form (awq_midm_cdna4.hip): prep() once per (tile, lane), word(w, p) per 32-k slice
  struct Prep {
    u32 b01, b23;
    float cv;
  };
  __device__ __forceinline__ Prep prep(u32 sz) const {
    const u32 sdup = __builtin_amdgcn_perm(sz, sz, 0x01000100u);  // {s, s}
    return Prep{sdup & m01, sdup & m23, DT::dq_offset(sz)};
  }
  __device__ __forceinline__ vec8 word(u32 w, const Prep& p) const { return word(w, p.b01, p.b23, p.cv); }
  // same, from the packed {scale | scaled_zero << 16} dword: one v_perm for the splat, one v_dot2 for sz - offset * s
  __device__ __forceinline__ void tile_packed(const u32x4& w, u32 sz, vec8 (&op)[4]) const {
    const u32 sdup = __builtin_amdgcn_perm(sz, sz, 0x01000100u);  // {s, s}
    const u32 b01 = sdup & m01, b23 = sdup & m23;
    const float cv = DT::dq_offset(sz);
    op[0] = word(w.x, b01, b23, cv);
    op[1] = word(w.y, b01, b23, cv);
    op[2] = word(w.z, b01, b23, cv);
    op[3] = word(w.w, b01, b23, cv);
  }
};
using Cdna4Dequant = Cdna4DequantT<BF16>;

// ---------------------------------------------------------------------------------------------
// "f16-mantissa" form of the matrix-core dequant (decode kernels; bf16 AND fp16 models).  The 4x4x4 dequant MFMA runs in
// its f16 form whatever T is: a 10-bit mantissa takes a nibble at bits 3:0 (1024 + q) and one at bits 7:4 (1024 + 16 q) of
// each half WITHOUT a shift, so a word costs 1 shift + 4 v_and_or instead of 3 + 4 (the decode kernel is issue bound:
// profiles/r02_tile_ubench.txt, 142 -> 117 ns per tile per SIMD).  The x16 of the high nibbles is folded into the
// diagonal operand: rows n % 4 >= 2 of a quad -- inner slots 2, 3 of the block, fed by the (w & 0x00F000F0) extractions --
// carry s / 16.  Per (row, group) the side buffer "sz_half" holds {f16(s') | f16(sz) << 16}, s' = s or s / 16, and the
// offset is C = sz - 1024 s' (one v_dot2_f32_f16), so D = (1024 + 16^e q) s' + sz - 1024 s' = q s + sz exactly in fp32
// (11 x 11-bit products, |sz| = s z <= 15 s); one v_cvt_pk to T is the reference's single rounding.  Requires s' and sz to
// be exactly representable as NORMAL f16 numbers (or 0): awq_pack_szh_cdna4 checks that per layer and the callers keep
// the T-typed form (Cdna4DequantT) for layers that fail (bf16 scales below 2^-10).
// ---------------------------------------------------------------------------------------------
template <typename DT>
struct Cdna4DequantH {
  using vec8 = typename DT::vec8;
  u32 sel01, sel23;  // v_perm selectors placing s' at inner index lane % 4 of the diagonal B operand
  u32 kMagic, kMaskLo, kMaskHi, kDotC;
  __device__ __forceinline__ void init(int lane) {
    const int pos = lane & 3;
    // v_perm_b32(S0, S1, sel): result byte i = byte sel[i] of {S0: 4..7, S1: 0..3}; selector 0x0C = constant 0x00
    sel01 = pos == 0 ? 0x0C0C0100u : (pos == 1 ? 0x01000C0Cu : 0x0C0C0C0Cu);
    sel23 = pos == 2 ? 0x0C0C0100u : (pos == 3 ? 0x01000C0Cu : 0x0C0C0C0Cu);
    kMagic = 0x64006400u;
    kMaskLo = 0x000F000Fu;
    kMaskHi = 0x00F000F0u;
    kDotC = 0x3C00E400u;  // f16 pair {-1024 (lo), 1 (hi)}
    asm volatile("" : "+v"(kMagic));
    asm volatile("" : "+s"(kMaskLo));
    asm volatile("" : "+s"(kMaskHi));
    asm volatile("" : "+v"(kDotC));
  }
  // the same tile word by word, for kernels that spread a tile's dequant over the product MFMAs of the tile before it (awq_midm_cdna4.hip):
  // prep() once per (tile, lane), word() per 32-k slice
  struct Prep {
    u32x2 b;
    f32x4 c;
  };
  __device__ __forceinline__ Prep prep(u32 szh) const {
    Prep p;
    p.b = u32x2{__builtin_amdgcn_perm(szh, szh, sel01), __builtin_amdgcn_perm(szh, szh, sel23)};
    const float cv = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, szh), __builtin_bit_cast(f16x2, kDotC), 0.0f, false);
    p.c = f32x4{cv, cv, cv, cv};
    return p;
  }
  __device__ __forceinline__ vec8 word(u32 w, const Prep& p) const {
    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
    const u32 w8 = w >> 8;
    const u32x2 a0 = {(w & kMaskLo) | kMagic, (w & kMaskHi) | kMagic};
    const u32x2 a1 = {(w8 & kMaskLo) | kMagic, (w8 & kMaskHi) | kMagic};
    const f32x4 d0 = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h16x4, a0), __builtin_bit_cast(h16x4, p.b), p.c, 0, 0, 0);
    const f32x4 d1 = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h16x4, a1), __builtin_bit_cast(h16x4, p.b), p.c, 0, 0, 0);
    return DT::pack8(d0, d1);
  }
  // whole 1-KiB tile -> 4 operands (op[a] covers k = 32a + 8g + 0..7 of the tile's 128 k), szh = this lane's sz_half dword
  __device__ __forceinline__ void tile(const u32x4& w, u32 szh, vec8 (&op)[4]) const {
    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
    const u32 b01 = __builtin_amdgcn_perm(szh, szh, sel01), b23 = __builtin_amdgcn_perm(szh, szh, sel23);
    const float cv = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, szh), __builtin_bit_cast(f16x2, kDotC), 0.0f, false);
    const f32x4 c = {cv, cv, cv, cv};
    const u32x2 b = {b01, b23};
    const u32 ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const u32 w8 = ws[a] >> 8;
      const u32x2 a0 = {(ws[a] & kMaskLo) | kMagic, (ws[a] & kMaskHi) | kMagic};
      const u32x2 a1 = {(w8 & kMaskLo) | kMagic, (w8 & kMaskHi) | kMagic};
      const f32x4 d0 = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h16x4, a0), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
      const f32x4 d1 = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h16x4, a1), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
      op[a] = DT::pack8(d0, d1);
    }
  }
};


// =============================================================================================
// W3 ("w3c") tiles: the repository's 3-bit format (the reference has none: qmodule.py:82-83 raises for
// w_bit != 4; INT3 exists only as pseudo-quantisation, quantizer.py:61-103 with n_bit = 3).
// One tile = 16 rows x 128 k = 64 lanes x 3 words (768 B, contiguous).  It is the cdna4 W4 tile of the same
// integers (values 0..7, so bit 3 of every nibble is free) with logical word 3 folded into those free
// bits: bit 3 of nibble j of stored word c holds bit c of nibble j of logical word 3.
// =============================================================================================
__device__ __forceinline__ size_t w3_tile_word(int nb, int kg, int nit) { return ((size_t)nb * nit + kg) * 192; }

// stored (W0, W1, W2) -> the four logical words; words 0..2 keep the foreign bit 3 (consumers mask with 0x0007)
__device__ __forceinline__ u32x4 w3_expand(u32 W0, u32 W1, u32 W2) {
  u32 t = (W0 >> 3) & 0x11111111u;
  t |= (W1 >> 2) & 0x22222222u;
  t |= (W2 >> 1) & 0x44444444u;
  return u32x4{W0, W1, W2, t};
}
__device__ __forceinline__ void w3_fold(const u32x4& w, u32 (&out)[3]) {
  out[0] = (w.x & 0x77777777u) | ((w.w & 0x11111111u) << 3);
  out[1] = (w.y & 0x77777777u) | ((w.w & 0x22222222u) << 2);
  out[2] = (w.z & 0x77777777u) | ((w.w & 0x44444444u) << 1);
}

}  // namespace awq
