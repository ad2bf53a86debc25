// Format / parity helper kernels.  They run the SAME device unpack + dequant routines as the
// GEMV / GEMM kernels (awq_device.hpp), so a bit-exact match of their outputs with the oracle
// pins the index arithmetic and the dequant numerics of the hot path.
//
// Reference behaviour restated: awq/quantize/qmodule.py:26-65 (pack), dequantize.cuh:18-123 +
// gemv_cuda.cu:150-174 (unpack order), gemv_cuda.cu:159-166 (dequant),
// tinychat/offline-weight-repacker.py:8-19,64-73,111-152 (v1 -> v2).
#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

// one thread per 16-byte chunk (row n, 32 k)
__global__ void unpack_v2_kernel(const u32* __restrict__ qw, uint8_t* __restrict__ out, int N, int K) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int chunks = K / 32;
  if (t >= (size_t)N * chunks) return;
  const int n = (int)(t / chunks), c = (int)(t % chunks);
  const u32x4 w = *reinterpret_cast<const u32x4*>(qw + v2_chunk_word(n, c, K));
  const u32 ws[4] = {w.x, w.y, w.z, w.w};
  uint8_t* o = out + (size_t)n * K + (size_t)c * 32;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    u32 lo[4], hi[4];
    unpack_word_pairs(ws[a], lo, hi);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[8 * i + 2 * a] = (uint8_t)lo[i];
      o[8 * i + 2 * a + 1] = (uint8_t)hi[i];
    }
  }
}

template <typename DT>
__global__ void dequant_v2_kernel(const u32* __restrict__ qw, const uint16_t* __restrict__ scales,
                                  const uint16_t* __restrict__ zeros, uint16_t* __restrict__ out, int N, int K) {
  using vec8 = typename DT::vec8;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int chunks = K / 32;
  if (t >= (size_t)N * chunks) return;
  const int n = (int)(t / chunks), c = (int)(t % chunks);
  const u32x4 w = *reinterpret_cast<const u32x4*>(qw + v2_chunk_word(n, c, K));
  const int grp = (c * 32) / kGroup;
  vec8 op[4];
  dequant_chunk<DT>(w, DT::make_sz(scales[(size_t)grp * N + n], zeros[(size_t)grp * N + n]), op);
  vec8* o = reinterpret_cast<vec8*>(out + (size_t)n * K + (size_t)c * 32);
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = op[j];
}

// logical u8 [N,K] -> v2.  One thread per output u32 word.
__global__ void pack_v2_kernel(const uint8_t* __restrict__ q, u32* __restrict__ qw, int N, int K) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t words = (size_t)N * K / 8;
  if (t >= words) return;
  // invert v2_chunk_word: word index -> (n, chunk, a)
  const size_t row_words = (size_t)K / 2;
  const int r = (int)(t / row_words);
  const int wi = (int)(t % row_words);
  const int kb = wi / 32, in = wi % 32;
  const int rr = in / 8, cc = (in % 8) / 4, a = in % 4;
  const int n = r * 4 + rr, k0 = kb * 64 + cc * 32;
  const uint8_t* src = q + (size_t)n * K + k0;
  u32 w = 0;
#pragma unroll
  for (int nib = 0; nib < 8; ++nib) w |= (u32)(src[v2_nibble_k(a, nib)] & 0xF) << (4 * nib);
  qw[t] = w;
}

// v1 qweight int32 [N, K/8] (nibble kk%8 of word kk/8) -> v2.  One thread per v2 word.
__global__ void repack_qweight_v1_to_v2_kernel(const u32* __restrict__ qw1, u32* __restrict__ qw2, int N, int K) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t words = (size_t)N * K / 8;
  if (t >= words) return;
  const size_t row_words = (size_t)K / 2;
  const int r = (int)(t / row_words);
  const int wi = (int)(t % row_words);
  const int kb = wi / 32, in = wi % 32;
  const int rr = in / 8, cc = (in % 8) / 4, a = in % 4;
  const int n = r * 4 + rr, k0 = kb * 64 + cc * 32;
  const u32* src = qw1 + (size_t)n * (K / 8) + k0 / 8;  // 4 v1 words = this 32-k chunk
  u32 w = 0;
#pragma unroll
  for (int nib = 0; nib < 8; ++nib) {
    const int kl = v2_nibble_k(a, nib);
    w |= ((src[kl >> 3] >> (4 * (kl & 7))) & 0xFu) << (4 * nib);
  }
  qw2[t] = w;
}

// scales_v1 T [N, gpad] -> scales_v2 T [gpad, N];  scaled_zeros_v2 = -(scales * zero) in T
// (multiply_scale_qzero_negative with zp_shift = 0, repacker :64-73,:142: product rounded to T, then negated)
template <typename DT>
__global__ void repack_scales_v1_to_v2_kernel(const uint16_t* __restrict__ s1, const u32* __restrict__ qz1,
                                              uint16_t* __restrict__ s2, uint16_t* __restrict__ sz2, int N, int gpad) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)N * gpad) return;
  const int n = (int)(t % N), gi = (int)(t / N);  // write-coalesced over n
  const uint16_t sb = s1[(size_t)n * gpad + gi];
  const u32 zq = (qz1[(size_t)n * (gpad / 8) + gi / 8] >> (4 * (gi % 8))) & 0xFu;
  float s;
  if (DT::id == 0)
    s = (float)__builtin_bit_cast(_Float16, sb);
  else
    s = __builtin_bit_cast(float, (u32)sb << 16);
  u32 prod = DT::from_float(s * (float)zq);  // exact product, one rounding to T
  s2[(size_t)gi * N + n] = sb;
  // -(x + 0*s): x + (+0) keeps x, unary minus flips the sign bit (also of zero: -(+0) = -0).
  // The empty asm keeps the compiler from folding the sign flip into the multiply (it would emit
  // fma(s, -z, +0), which turns the reference's -0.0 into +0.0).
  asm volatile("" : "+v"(prod));
  sz2[(size_t)gi * N + n] = (uint16_t)(prod ^ 0x8000u);
}

template <typename DT>
__global__ void bias_add_kernel(uint16_t* __restrict__ out, const uint16_t* __restrict__ bias, size_t total, int N) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  float a, b;
  if (DT::id == 0) {
    a = (float)__builtin_bit_cast(_Float16, out[t]);
    b = (float)__builtin_bit_cast(_Float16, bias[t % N]);
  } else {
    a = __builtin_bit_cast(float, (u32)out[t] << 16);
    b = __builtin_bit_cast(float, (u32)bias[t % N] << 16);
  }
  out[t] = DT::from_float(a + b);  // == T-precision add (sum of two T values rounded once)
}

// ---------------------------------------------------------------------------------------------
// cdna4 interleave <-> v2 (pure nibble permutations, one thread per destination u32 word)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 v2_read_nibble(const u32* qw, int n, int k, int K) {
  const int kl = k & 31;
  const int a = (kl & 7) >> 1, nib = (kl >> 3) + 4 * (kl & 1);
  return (qw[v2_chunk_word(n, k >> 5, K) + a] >> (4 * nib)) & 0xFu;
}
__device__ __forceinline__ u32 cdna4_read_nibble(const u32* qw, int n, int k, int K) {
  const int nb = n >> 4, c = n & 15, g = c >> 2, j = c & 3;
  const int kg = k >> 7, kk = k & 127, a = kk >> 5, r32 = kk & 31;
  const int b8 = r32 >> 3, e = r32 & 7, th = e >> 2, rr = e & 3;
  const int lane = 16 * b8 + 4 * g + rr;  // k octet | row quad | k within the quartet
  const int p = (2 * th + (j >> 1)) + 4 * (j & 1);
  return (qw[cdna4_tile_word(nb, kg, K >> 7) + lane * 4 + a] >> (4 * p)) & 0xFu;
}

__global__ void repack_v2_to_cdna4_kernel(const u32* __restrict__ src, u32* __restrict__ dst, int N, int K) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)N * K / 8) return;
  const int a = (int)(t & 3), lane = (int)((t >> 2) & 63);
  const size_t tile = t >> 8;
  const int nit = K >> 7;
  const int nb = (int)(tile / nit), kg = (int)(tile % nit);
  const int g = lane >> 4, nq = (lane >> 2) & 3, r = lane & 3;
  u32 w = 0;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int i = p & 3, hi = p >> 2;
    const int n = 16 * nb + 4 * nq + 2 * (i & 1) + hi;
    const int k = 128 * kg + 32 * a + 8 * g + 4 * (i >> 1) + r;
    w |= v2_read_nibble(src, n, k, K) << (4 * p);
  }
  dst[t] = w;
}

__global__ void repack_cdna4_to_v2_kernel(const u32* __restrict__ src, u32* __restrict__ dst, int N, int K) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)N * K / 8) return;
  const size_t row_words = (size_t)K / 2;
  const int r = (int)(t / row_words);
  const int wi = (int)(t % row_words);
  const int kb = wi / 32, in = wi % 32;
  const int rr = in / 8, cc = (in % 8) / 4, a = in % 4;
  const int n = r * 4 + rr, k0 = kb * 64 + cc * 32;
  u32 w = 0;
#pragma unroll
  for (int nib = 0; nib < 8; ++nib) w |= cdna4_read_nibble(src, n, k0 + v2_nibble_k(a, nib), K) << (4 * nib);
  dst[t] = w;
}

__global__ void unpack_cdna4_kernel(const u32* __restrict__ qw, uint8_t* __restrict__ out, int N, int K) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)N * K) return;
  const int n = (int)(t / K), k = (int)(t % K);
  out[t] = (uint8_t)cdna4_read_nibble(qw, n, k, K);
}

// one wave per 1-KiB tile, through the SAME matrix-core dequant as the cdna4 GEMV / GEMM
template <typename DT>
__global__ __launch_bounds__(64) void dequant_cdna4_kernel(const u32* __restrict__ qw, const uint16_t* __restrict__ scales,
                                                            const uint16_t* __restrict__ zeros, uint16_t* __restrict__ out,
                                                            int N, int K) {
  using vec8 = typename DT::vec8;
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  const int nit = K >> 7;
  const int nb = blockIdx.x / nit, kg = blockIdx.x % nit;
  Cdna4DequantT<DT> cd;
  cd.init(lane);
  const u32x4 w = *reinterpret_cast<const u32x4*>(qw + cdna4_tile_word(nb, kg, nit) + lane * 4);
  const int n = nb * 16 + c;
  vec8 op[4];
  cd.tile(w, scales[(size_t)kg * N + n], zeros[(size_t)kg * N + n], op);
#pragma unroll
  for (int a = 0; a < 4; ++a)
    *reinterpret_cast<vec8*>(out + (size_t)n * K + (size_t)kg * 128 + 32 * a + 8 * g) = op[a];
}

// packed {scale | scaled_zero << 16} per (16-row slab, group, row): one dword load per lane per step
__global__ void pack_sz_cdna4_kernel(const uint16_t* __restrict__ s, const uint16_t* __restrict__ z, u32* __restrict__ out,
                                     int N, int nit) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)N * nit) return;
  const int c = (int)(t & 15);
  const size_t tile = t >> 4;
  const int nb = (int)(tile / nit), kg = (int)(tile % nit);
  const size_t src = (size_t)kg * N + nb * 16 + c;
  out[t] = (u32)s[src] | ((u32)z[src] << 16);
}

// "sz_half" side buffer of the decode kernels (Cdna4DequantH, awq_device.hpp): {f16(s') | f16(sz) << 16}, s' = s for rows
// n % 4 < 2 and s / 16 for the others.  *inexact is set when a value is not exactly representable as a normal f16 number
// (or 0): the caller then keeps the T-typed sz_packed form for this layer.
template <typename DT>
__global__ void pack_szh_cdna4_kernel(const uint16_t* __restrict__ s, const uint16_t* __restrict__ z, u32* __restrict__ out,
                                      int* __restrict__ inexact, int N, int nit) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)N * nit) return;
  const int c = (int)(t & 15);
  const size_t tile = t >> 4;
  const int nb = (int)(tile / nit), kg = (int)(tile % nit);
  const size_t src = (size_t)kg * N + nb * 16 + c;
  const float sf = DT::to_float(s[src]), zf = DT::to_float(z[src]);
  const float sp = (c & 3) >= 2 ? sf * 0.0625f : sf;
  const _Float16 sh = (_Float16)sp, zh = (_Float16)zf;
  auto ok = [](float v, _Float16 h) { return (float)h == v && (v == 0.0f || fabsf(v) >= 6.103515625e-05f); };
  if (!ok(sp, sh) || !ok(zf, zh)) atomicOr(inexact, 1);
  out[t] = (u32)__builtin_bit_cast(uint16_t, sh) | ((u32)__builtin_bit_cast(uint16_t, zh) << 16);
}

static inline unsigned nblk(size_t n, unsigned b) { return (unsigned)((n + b - 1) / b); }

int launch_unpack_v2(const void* qw, void* out_u8, int n, int k, hipStream_t st) {
  const size_t items = (size_t)n * (k / 32);
  hipLaunchKernelGGL(unpack_v2_kernel, dim3(nblk(items, 256)), dim3(256), 0, st, (const u32*)qw, (uint8_t*)out_u8, n, k);
  return 0;
}

int launch_dequant_v2(const void* qw, const void* s, const void* z, void* out, int n, int k, int dtype, hipStream_t st) {
  const size_t items = (size_t)n * (k / 32);
  if (dtype == 0)
    hipLaunchKernelGGL((dequant_v2_kernel<F16>), dim3(nblk(items, 256)), dim3(256), 0, st, (const u32*)qw,
                       (const uint16_t*)s, (const uint16_t*)z, (uint16_t*)out, n, k);
  else
    hipLaunchKernelGGL((dequant_v2_kernel<BF16>), dim3(nblk(items, 256)), dim3(256), 0, st, (const u32*)qw,
                       (const uint16_t*)s, (const uint16_t*)z, (uint16_t*)out, n, k);
  return 0;
}

int launch_pack_v2(const void* q_u8, void* qw, int n, int k, hipStream_t st) {
  const size_t words = (size_t)n * k / 8;
  hipLaunchKernelGGL(pack_v2_kernel, dim3(nblk(words, 256)), dim3(256), 0, st, (const uint8_t*)q_u8, (u32*)qw, n, k);
  return 0;
}

int launch_repack_v1_to_v2(const void* qw1, const void* s1, const void* qz1, void* qw2, void* s2, void* sz2, int n, int k,
                           int gpad, int dtype, hipStream_t st) {
  const size_t words = (size_t)n * k / 8;
  hipLaunchKernelGGL(repack_qweight_v1_to_v2_kernel, dim3(nblk(words, 256)), dim3(256), 0, st, (const u32*)qw1,
                     (u32*)qw2, n, k);
  const size_t items = (size_t)n * gpad;
  if (dtype == 0)
    hipLaunchKernelGGL((repack_scales_v1_to_v2_kernel<F16>), dim3(nblk(items, 256)), dim3(256), 0, st,
                       (const uint16_t*)s1, (const u32*)qz1, (uint16_t*)s2, (uint16_t*)sz2, n, gpad);
  else
    hipLaunchKernelGGL((repack_scales_v1_to_v2_kernel<BF16>), dim3(nblk(items, 256)), dim3(256), 0, st,
                       (const uint16_t*)s1, (const u32*)qz1, (uint16_t*)s2, (uint16_t*)sz2, n, gpad);
  return 0;
}

int launch_repack_v2_cdna4(const void* src, void* dst, int n, int k, int to_cdna4, hipStream_t st) {
  const size_t words = (size_t)n * k / 8;
  if (to_cdna4)
    hipLaunchKernelGGL(repack_v2_to_cdna4_kernel, dim3(nblk(words, 256)), dim3(256), 0, st, (const u32*)src, (u32*)dst, n, k);
  else
    hipLaunchKernelGGL(repack_cdna4_to_v2_kernel, dim3(nblk(words, 256)), dim3(256), 0, st, (const u32*)src, (u32*)dst, n, k);
  return 0;
}

int launch_pack_sz_cdna4(const void* s, const void* z, void* szp, int n, int k, hipStream_t st) {
  const size_t items = (size_t)n * (k / kGroup);
  hipLaunchKernelGGL(pack_sz_cdna4_kernel, dim3(nblk(items, 256)), dim3(256), 0, st, (const uint16_t*)s, (const uint16_t*)z,
                     (u32*)szp, n, k / kGroup);
  return 0;
}

int launch_pack_szh_cdna4(const void* s, const void* z, void* szh, int* inexact, int n, int k, int dtype, hipStream_t st) {
  const size_t items = (size_t)n * (k / kGroup);
  auto kern = dtype == 0 ? pack_szh_cdna4_kernel<F16> : pack_szh_cdna4_kernel<BF16>;
  hipLaunchKernelGGL(kern, dim3(nblk(items, 256)), dim3(256), 0, st, (const uint16_t*)s, (const uint16_t*)z, (u32*)szh, inexact, n,
                     k / kGroup);
  return 0;
}

int launch_unpack_cdna4(const void* qw, void* out_u8, int n, int k, hipStream_t st) {
  hipLaunchKernelGGL(unpack_cdna4_kernel, dim3(nblk((size_t)n * k, 256)), dim3(256), 0, st, (const u32*)qw, (uint8_t*)out_u8, n, k);
  return 0;
}

int launch_dequant_cdna4(const void* qw, const void* s, const void* z, void* out, int n, int k, int dtype, hipStream_t st) {
  auto kern = dtype == 0 ? dequant_cdna4_kernel<F16> : dequant_cdna4_kernel<BF16>;
  hipLaunchKernelGGL(kern, dim3((n / 16) * (k / 128)), dim3(64), 0, st, (const u32*)qw, (const uint16_t*)s,
                     (const uint16_t*)z, (uint16_t*)out, n, k);
  return 0;
}

// RMSNorm of whole rows (prefill side of RMSNormWQLinear; decode rows are fused into the GEMV, awq_gemv_cdna4.hip NORM = 1): one 256-thread
// block per row, 16-byte loads, fp32 sum of squares (lane -> wave -> block), rsqrtf(mean + eps), (x * rstd) * gamma with ONE rounding
// to T -- layernorm.cu:48-60's arithmetic (generalT5LayerNorm: no mean subtraction, no bias).  k % 8 == 0.
template <typename DT>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ gamma,
                                                      uint16_t* __restrict__ out, int K, float eps) {
  __shared__ float part[4];
  const uint16_t* xr = x + (size_t)blockIdx.x * K;
  uint16_t* orow = out + (size_t)blockIdx.x * K;
  const int ng = K >> 3;  // 16-byte granules per row
  float ss = 0.f;
  for (int gi = threadIdx.x; gi < ng; gi += 256) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(xr + (size_t)gi * 8);
    const u32 w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = DT::to_float((uint16_t)(w4[e] & 0xFFFFu)), hi = DT::to_float((uint16_t)(w4[e] >> 16));
      ss = __builtin_fmaf(lo, lo, ss);
      ss = __builtin_fmaf(hi, hi, ss);
    }
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) ss += __shfl_xor(ss, d, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float tot = (part[0] + part[1]) + (part[2] + part[3]);
  const float rstd = rsqrtf(tot / (float)K + eps);  // layernorm.cu:55
  for (int gi = threadIdx.x; gi < ng; gi += 256) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(xr + (size_t)gi * 8), gv = *reinterpret_cast<const u32x4*>(gamma + (size_t)gi * 8);
    const u32 w4[4] = {v.x, v.y, v.z, v.w}, g4[4] = {gv.x, gv.y, gv.z, gv.w};
    u32 o4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // layernorm.cu:60: T((float(x) * s_variance) * float(gamma))
      const float lo = (DT::to_float((uint16_t)(w4[e] & 0xFFFFu)) * rstd) * DT::to_float((uint16_t)(g4[e] & 0xFFFFu));
      const float hi = (DT::to_float((uint16_t)(w4[e] >> 16)) * rstd) * DT::to_float((uint16_t)(g4[e] >> 16));
      o4[e] = (u32)DT::from_float(lo) | ((u32)DT::from_float(hi) << 16);
    }
    *reinterpret_cast<u32x4*>(orow + (size_t)gi * 8) = u32x4{o4[0], o4[1], o4[2], o4[3]};
  }
}

int launch_rmsnorm(const void* x, const void* gamma, float eps, void* out, int m, int k, int dtype, hipStream_t st) {
  if (m < 1 || k < 8 || (k % 8) != 0) return -1;
  if (dtype == 0) hipLaunchKernelGGL((rmsnorm_kernel<F16>), dim3(m), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)gamma, (uint16_t*)out, k, eps);
  else hipLaunchKernelGGL((rmsnorm_kernel<BF16>), dim3(m), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)gamma, (uint16_t*)out, k, eps);
  return 0;
}

// the SiLU * mul tail (tinychat/modules/fused_mlp.py:79-82: c = F.silu(gate_output) * up_output, every op rounded to T) on a matmul result whose
// columns are the 8 + 8 interleaved gate / up pair: one thread = one output octet (reads 2 x 16 B, writes 16 B)
template <typename DT>
__global__ __launch_bounds__(256) void silu_mul_interleaved_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, size_t octets, int half_octets) {
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < octets; o += (size_t)gridDim.x * 256) {
    const size_t row = o / (size_t)half_octets, j = o - row * (size_t)half_octets;
    const uint16_t* src = in + (row * (size_t)half_octets * 2 + j * 2) * 8;
    const u32x4 g = *reinterpret_cast<const u32x4*>(src), u = *reinterpret_cast<const u32x4*>(src + 8);
    *reinterpret_cast<u32x4*>(out + o * 8) = silu_mul_octet<DT>(g, u);
  }
}

template <typename DT>
__global__ __launch_bounds__(256) void silu_mul_kernel(const uint16_t* __restrict__ gate, const uint16_t* __restrict__ up, uint16_t* __restrict__ out, size_t octets) {
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < octets; o += (size_t)gridDim.x * 256)
    *reinterpret_cast<u32x4*>(out + o * 8) = silu_mul_octet<DT>(*reinterpret_cast<const u32x4*>(gate + o * 8), *reinterpret_cast<const u32x4*>(up + o * 8));
}

// out = T(T(silu(gate)) * up), elementwise over `count` values (count % 8 == 0)
int launch_silu_mul(const void* gate, const void* up, void* out, size_t count, int dtype, hipStream_t st) {
  if (count == 0 || (count % 8) != 0) return -1;
  const size_t octets = count / 8;
  const unsigned blocks = (octets + 255) / 256 > 8192u ? 8192u : (unsigned)((octets + 255) / 256);
  if (dtype == 0) hipLaunchKernelGGL((silu_mul_kernel<F16>), dim3(blocks), dim3(256), 0, st, (const uint16_t*)gate, (const uint16_t*)up, (uint16_t*)out, octets);
  else hipLaunchKernelGGL((silu_mul_kernel<BF16>), dim3(blocks), dim3(256), 0, st, (const uint16_t*)gate, (const uint16_t*)up, (uint16_t*)out, octets);
  return 0;
}

int launch_silu_mul_interleaved(const void* in, void* out, int m, int n2, int dtype, hipStream_t st) {
  if (m < 1 || n2 < 16 || (n2 % 16) != 0) return -1;
  const size_t octets = (size_t)m * (n2 / 16);
  const unsigned blocks = (octets + 255) / 256 > 8192u ? 8192u : (unsigned)((octets + 255) / 256);
  if (dtype == 0) hipLaunchKernelGGL((silu_mul_interleaved_kernel<F16>), dim3(blocks), dim3(256), 0, st, (const uint16_t*)in, (uint16_t*)out, octets, n2 / 16);
  else hipLaunchKernelGGL((silu_mul_interleaved_kernel<BF16>), dim3(blocks), dim3(256), 0, st, (const uint16_t*)in, (uint16_t*)out, octets, n2 / 16);
  return 0;
}

// out[m, n] = T(in_f32[m, n]) (+ bias in T): the single rounding of a tensor-parallel row split after its fp32 partials were summed
// (by RCCL: the bandwidth-class messages; the latency-class ones are summed and rounded inside awq_oneshot_allreduce_f32)
template <typename DT>
__global__ __launch_bounds__(256) void round_bias_f32_kernel(const float* __restrict__ in, const uint16_t* __restrict__ bias, uint16_t* __restrict__ out,
                                                             size_t octets, int n) {
  for (size_t o = (size_t)blockIdx.x * 256 + threadIdx.x; o < octets; o += (size_t)gridDim.x * 256) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(in + o * 8), b = *reinterpret_cast<const f32x4*>(in + o * 8 + 4);
    uint16_t v[8] = {DT::from_float(a[0]), DT::from_float(a[1]), DT::from_float(a[2]), DT::from_float(a[3]),
                     DT::from_float(b[0]), DT::from_float(b[1]), DT::from_float(b[2]), DT::from_float(b[3])};
    if (bias != nullptr) {  // `out + self.bias` in T (qmodule.py:221)
      const u32x4 bv = *reinterpret_cast<const u32x4*>(bias + (o * 8) % (size_t)n);
      const u32 bw[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = DT::from_float(DT::to_float(v[e]) + DT::to_float((uint16_t)((bw[e >> 1] >> (16 * (e & 1))) & 0xFFFFu)));
    }
    *reinterpret_cast<u32x4*>(out + o * 8) = u32x4{(u32)v[0] | ((u32)v[1] << 16), (u32)v[2] | ((u32)v[3] << 16), (u32)v[4] | ((u32)v[5] << 16),
                                                   (u32)v[6] | ((u32)v[7] << 16)};
  }
}

int launch_round_bias_f32(const void* in_f32, const void* bias, void* out, int m, int n, int dtype, hipStream_t st) {
  if (m < 1 || n < 8 || (n % 8) != 0) return -1;
  const size_t octets = (size_t)m * n / 8;
  const unsigned blocks = (unsigned)(octets + 255) / 256 > 4096u ? 4096u : (unsigned)((octets + 255) / 256);
  if (dtype == 0) hipLaunchKernelGGL((round_bias_f32_kernel<F16>), dim3(blocks), dim3(256), 0, st, (const float*)in_f32, (const uint16_t*)bias, (uint16_t*)out, octets, n);
  else hipLaunchKernelGGL((round_bias_f32_kernel<BF16>), dim3(blocks), dim3(256), 0, st, (const float*)in_f32, (const uint16_t*)bias, (uint16_t*)out, octets, n);
  return 0;
}

int launch_bias_add(void* out, const void* bias, int m, int n, int dtype, hipStream_t st) {
  const size_t total = (size_t)m * n;
  if (dtype == 0)
    hipLaunchKernelGGL((bias_add_kernel<F16>), dim3(nblk(total, 256)), dim3(256), 0, st, (uint16_t*)out,
                       (const uint16_t*)bias, total, n);
  else
    hipLaunchKernelGGL((bias_add_kernel<BF16>), dim3(nblk(total, 256)), dim3(256), 0, st, (uint16_t*)out,
                       (const uint16_t*)bias, total, n);
  return 0;
}


// =============================================================================================
// W3 ("w3c" tiles) format helpers (formerly awq_w3.hip)
// W3 ("w3c") format kernels: pack / unpack / dequant / expand-to-W4.  The 3-bit format is this repository's
// (awq_device.hpp "W3 tiles"); the quantisation grid is the reference's pseudo_quantize_tensor with n_bit = 3
// (awq/quantize/quantizer.py:61-103).  unpack / dequant run the SAME device routines as the W3 GEMV, so their
// bit-exact agreement with the oracle pins the index arithmetic and the numerics of the matmul path.
// =============================================================================================


// logical (n, k) held by nibble p of logical word a of lane `lane` of a cdna4 tile (see awq_device.hpp)
__device__ __forceinline__ void cdna4_nibble_nk(int lane, int a, int p, int& n_in_slab, int& k_in_group) {
  const int g = lane >> 4, nq = (lane >> 2) & 3, r = lane & 3, i = p & 3, hi = p >> 2;
  n_in_slab = 4 * nq + 2 * (i & 1) + hi;
  k_in_group = 32 * a + 8 * g + 4 * (i >> 1) + r;
}

// one thread per (tile, lane): gather the 32 integers of the lane's four logical words, fold, store 3 words
__global__ void pack_w3_kernel(const uint8_t* __restrict__ q, u32* __restrict__ qw3, int N, int K) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nit = K >> 7;
  if (t >= (size_t)(N >> 4) * nit * 64) return;
  const int lane = (int)(t & 63);
  const size_t tile = t >> 6;
  const int nb = (int)(tile / nit), kg = (int)(tile % nit);
  u32 w[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    u32 v = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      int nn, kk;
      cdna4_nibble_nk(lane, a, p, nn, kk);
      v |= (u32)(q[(size_t)(nb * 16 + nn) * K + kg * 128 + kk] & 7u) << (4 * p);
    }
    w[a] = v;
  }
  u32 o[3];
  w3_fold(u32x4{w[0], w[1], w[2], w[3]}, o);
  u32* dst = qw3 + tile * 192 + lane * 3;
  dst[0] = o[0];
  dst[1] = o[1];
  dst[2] = o[2];
}

__global__ void unpack_w3_kernel(const u32* __restrict__ qw3, uint8_t* __restrict__ out, int N, int K) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nit = K >> 7;
  if (t >= (size_t)(N >> 4) * nit * 64) return;
  const int lane = (int)(t & 63);
  const size_t tile = t >> 6;
  const int nb = (int)(tile / nit), kg = (int)(tile % nit);
  const u32* src = qw3 + tile * 192 + lane * 3;
  const u32x4 w = w3_expand(src[0], src[1], src[2]);
  const u32 ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      int nn, kk;
      cdna4_nibble_nk(lane, a, p, nn, kk);
      out[(size_t)(nb * 16 + nn) * K + kg * 128 + kk] = (uint8_t)((ws[a] >> (4 * p)) & 7u);
    }
}

// one wave per tile, through the matrix-core dequant of the GEMV
template <typename DT>
__global__ __launch_bounds__(64) void dequant_w3_kernel(const u32* __restrict__ qw3, const uint16_t* __restrict__ scales,
                                                         const uint16_t* __restrict__ zeros, uint16_t* __restrict__ out,
                                                         int N, int K) {
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  const int nit = K >> 7;
  const int nb = blockIdx.x / nit, kg = blockIdx.x % nit;
  using vec8 = typename DT::vec8;
  Cdna4DequantT<DT> cd;
  cd.init(lane, 0x00070007u);
  const u32* src = qw3 + w3_tile_word(nb, kg, nit) + lane * 3;
  const u32x4 w = w3_expand(src[0], src[1], src[2]);
  const int n = nb * 16 + c;
  vec8 op[4];
  cd.tile(w, scales[(size_t)kg * N + n], zeros[(size_t)kg * N + n], op);
#pragma unroll
  for (int a = 0; a < 4; ++a)
    *reinterpret_cast<vec8*>(out + (size_t)n * K + (size_t)kg * 128 + 32 * a + 8 * g) = op[a];
}

// w3c tiles -> cdna4 W4 tiles of the same integers (prefill: the W4 GEMM kernels then run unchanged)
__global__ void expand_w3_kernel(const u32* __restrict__ qw3, u32* __restrict__ qw4, size_t lanes) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lanes) return;
  const u32* src = qw3 + t * 3;
  const u32x4 w = w3_expand(src[0], src[1], src[2]);
  *reinterpret_cast<u32x4*>(qw4 + t * 4) = u32x4{w.x & 0x77777777u, w.y & 0x77777777u, w.z & 0x77777777u, w.w};
}

static inline unsigned nblk3(size_t n, unsigned b) { return (unsigned)((n + b - 1) / b); }

int launch_pack_w3(const void* q_u8, void* qw3, int n, int k, hipStream_t st) {
  const size_t lanes = (size_t)(n / 16) * (k / 128) * 64;
  hipLaunchKernelGGL(pack_w3_kernel, dim3(nblk3(lanes, 256)), dim3(256), 0, st, (const uint8_t*)q_u8, (u32*)qw3, n, k);
  return 0;
}
int launch_unpack_w3(const void* qw3, void* out_u8, int n, int k, hipStream_t st) {
  const size_t lanes = (size_t)(n / 16) * (k / 128) * 64;
  hipLaunchKernelGGL(unpack_w3_kernel, dim3(nblk3(lanes, 256)), dim3(256), 0, st, (const u32*)qw3, (uint8_t*)out_u8, n, k);
  return 0;
}
int launch_dequant_w3(const void* qw3, const void* s, const void* z, void* out, int n, int k, int dtype, hipStream_t st) {
  auto kern = dtype == 0 ? dequant_w3_kernel<F16> : dequant_w3_kernel<BF16>;
  hipLaunchKernelGGL(kern, dim3((n / 16) * (k / 128)), dim3(64), 0, st, (const u32*)qw3, (const uint16_t*)s,
                     (const uint16_t*)z, (uint16_t*)out, n, k);
  return 0;
}
int launch_expand_w3_to_cdna4(const void* qw3, void* qw4, int n, int k, hipStream_t st) {
  const size_t lanes = (size_t)(n / 16) * (k / 128) * 64;
  hipLaunchKernelGGL(expand_w3_kernel, dim3(nblk3(lanes, 256)), dim3(256), 0, st, (const u32*)qw3, (u32*)qw4, lanes);
  return 0;
}

}  // namespace awq
