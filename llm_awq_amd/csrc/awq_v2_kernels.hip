// Kernels on the REFERENCE (v2) interleave for checkpoints that were not repacked (and for the engine with its cache off):
//   part 1: decode GEMV, m <= 8 (formerly awq_gemv_v2fast.hip);  part 2: skinny GEMM, 9 <= m <= 255 (formerly awq_skinny_v2.hip).
//
// ---- part 1 ----
// Decode GEMV on the REFERENCE (v2) interleave, 1 <= M <= 8, fp16 and bf16 (gfx950): the structure of the cdna4 fast
// path (awq_gemv_cdna4.hip: two-deep software pipeline, raw buffer loads with scalar tile offsets, wave-private swizzled
// x staging, split-K through LDS, fused bias) applied to un-repacked checkpoints, so gemv_forward_cuda_new on raw
// reference buffers (awq/kernels/csrc/quantization_new/gemv/gemv_cuda.cu:245-338) gets it for fp16 models too.
//
// Differences forced by the layout: a lane's 16 bytes are one output row x one 32-k chunk (qmodule.py:26-65), so a wave
// load is 4 x 256 B instead of one contiguous KiB, the weights are dequantised on the VALU with the reference's
// numerics (dequant_chunk: round_T(q*s + sz), packed fp16 math / fp32 fma + v_cvt_pk for bf16), and scales / zeros
// come from the [Gpad, N] tensors (two 16-bit loads per step).
//
// ---- part 2 ----
// Skinny GEMM on the REFERENCE (v2) interleave, 9 <= M <= 255, fp16 and bf16 (gfx950): the structure of
// awq_skinny_cdna4.hip for un-repacked checkpoints -- in practice fp16 models, the reference's default dtype, which the
// bf16-only matrix-core dequant cannot serve.  gemm_forward_cuda_new covers this range with 16/32-row tiles + split-K
// (awq/kernels/csrc/quantization_new/gemm/gemm_cuda.cu:1155-1193); the 128 x 128 tile kernel of awq_gemm.hip needed
// 62-207 us for it (profiles/r01_skinny_sweep.txt: one wave of tiles, a long serial K loop).
//
// A block owns NS 16-row slabs that share every x operand; WAVES waves split K in interleaved 128-k steps; per step a wave
//   * stages its x slice (16 CB rows x 128 k) through a wave-private XOR-swizzled LDS region, prefetched one step ahead;
//   * dequantises NS x 16 rows x 128 k on the VALU with the reference's numerics (dequant_chunk: round_T(q*s + sz));
//     a lane holds one row x one 32-k chunk, so a wave load is 4 x 256 B, scales / zeros two 16-bit loads per slab;
//   * issues NS x CB x 4 MFMA 16x16x32 (weights = A operand, 16 x rows = B operand), fp32 accumulation.
// Split-K partials are reduced through LDS in fp32; one rounding; bias fused.
#include <string.h>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {


template <typename DT, int WAVES, int S, int MB>
__global__ __launch_bounds__(64 * WAVES) void gemv_v2fast_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                                  const uint16_t* __restrict__ scales,
                                                                  const uint16_t* __restrict__ zeros,
                                                                  const uint16_t* __restrict__ bias,
                                                                  uint16_t* __restrict__ out, int M, int N, int K, int gpad) {
  using vec8 = typename DT::vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int nb = blockIdx.x, nit = K >> 7;
  const int xstep = M * 256;
  float(*red)[4][64] = reinterpret_cast<float(*)[4][64]>(smem);  // [WAVES][4][64]
  char* xs = smem + WAVES * 1024 + wv * (S * xstep);

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(qw), 0, (N >> 1) * K, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(scales), 0, gpad * N * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rzr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(zeros), 0, gpad * N * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, M * K * 2, 0x00020000);
  // v2 word of (row 16 nb + i, chunk 4 kg + g) = [4 nb (K/2) + 64 kg]  +  [(i>>2)(K/2) + 32 (g>>1) + 8 (i&3) + 4 (g&1)]
  const u32 wlane_b = ((u32)(i >> 2) * (u32)(K >> 1) + 32u * (g >> 1) + 8u * (i & 3) + 4u * (g & 1)) * 4u;
  const u32 wslab_b = (u32)nb * 4u * (u32)(K >> 1) * 4u;
  const u32 slane_b = (u32)i * 2u, sslab_b = (u32)nb * 32u;
  const int mrow = min(i, M - 1);
  const int cnt = (nit - wv + WAVES - 1) / WAVES;

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  struct Regs {
    u32x4 xr[S][MB];
    u32x4 w[S];
    uint16_t s[S], z[S];
  };
  auto load_chunk = [&](int c0, Regs& R) {
#pragma unroll
    for (int t = 0; t < S; ++t) {
      const int kg = min(wv + WAVES * (c0 + t), nit - 1);
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        const int r = min(4 * b + g, M - 1);
        const u32 xoff_b = ((u32)r * (u32)K + (u32)((i ^ (r & 15)) * 8)) * 2u;
        R.xr[t][b] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff_b, (u32)kg * 256u, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < S; ++t) {
      const int kg = min(wv + WAVES * (c0 + t), nit - 1);
      R.w[t] = __builtin_amdgcn_raw_buffer_load_b128(rw, wlane_b, wslab_b + (u32)kg * 256u, 2);  // aux 2 = nt
      const u32 so = ((u32)kg * (u32)N) * 2u + sslab_b;
      R.s[t] = __builtin_amdgcn_raw_buffer_load_b16(rsc, slane_b, so, 0);
      R.z[t] = __builtin_amdgcn_raw_buffer_load_b16(rzr, slane_b, so, 0);
    }
  };
  auto compute_chunk = [&](int c0, const Regs& R) {
#pragma unroll
    for (int t = 0; t < S; ++t)
#pragma unroll
      for (int b = 0; b < MB; ++b)
        *reinterpret_cast<u32x4*>(xs + t * xstep + min(4 * b + g, M - 1) * 256 + i * 16) = R.xr[t][b];
#pragma unroll
    for (int t = 0; t < S; ++t) {
      if (wv + WAVES * (c0 + t) >= nit) continue;
      const u32x4* xrow = reinterpret_cast<const u32x4*>(xs + t * xstep + mrow * 256);
      vec8 wop[4];
      dequant_chunk<DT>(R.w[t], DT::make_sz(R.s[t], R.z[t]), wop);  // wop[j] = k 32 g + 8 j .. + 7 of the step
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc = DT::mfma(wop[j], __builtin_bit_cast(vec8, xrow[(4 * g + j) ^ (mrow & 15)]), acc);
    }
  };
  Regs A, B;
  load_chunk(0, A);
  for (int c0 = 0; c0 < cnt; c0 += 2 * S) {
    load_chunk(c0 + S, B);
    compute_chunk(c0, A);
    load_chunk(c0 + 2 * S, A);
    compute_chunk(c0 + S, B);
  }

#pragma unroll
  for (int r = 0; r < 4; ++r) red[wv][r][lane] = acc[r];
  __syncthreads();
  if (wv < 4 && i < M) {
    const int r = wv;
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < WAVES; ++q) t += red[q][r][lane];
    const int nn = nb * 16 + 4 * g + r;
    uint16_t o = DT::from_float(t);
    if (bias != nullptr) {  // `out + self.bias` in T (qmodule.py:221)
      float a, b;
      if (DT::id == 0) {
        a = (float)__builtin_bit_cast(_Float16, o);
        b = (float)__builtin_bit_cast(_Float16, bias[nn]);
      } else {
        a = __builtin_bit_cast(float, (u32)o << 16);
        b = __builtin_bit_cast(float, (u32)bias[nn] << 16);
      }
      o = DT::from_float(a + b);
    }
    out[(size_t)i * N + nn] = o;
  }
}

template <typename DT, int WAVES, int S, int MB>
static void launch_v2fast(const void* x, const void* qw, const void* s, const void* z, const void* bias, void* out, int m, int n,
                          int k, int gpad, hipStream_t st) {
  const size_t smem = (size_t)WAVES * 1024 + (size_t)WAVES * S * m * 256;
  hipLaunchKernelGGL((gemv_v2fast_kernel<DT, WAVES, S, MB>), dim3(n / 16), dim3(64 * WAVES), smem, st, (const uint16_t*)x,
                     (const u32*)qw, (const uint16_t*)s, (const uint16_t*)z, (const uint16_t*)bias, (uint16_t*)out, m, n, k, gpad);
}

template <typename DT, int MB>
static int launch_v2fast_mb(const void* x, const void* qw, const void* s, const void* z, const void* bias, void* out, int m, int n,
                            int k, int gpad, hipStream_t st) {
  const int nit = k / kGroup, slabs = n / 16;
  int waves = slabs >= 768 ? 4 : (slabs >= 384 ? 8 : (nit >= 64 ? 8 : 16));  // as the cdna4 fast path
  while (waves > 4 && waves * 2 > nit) waves >>= 1;
  const int per = (nit + waves - 1) / waves;
  const int ps = per >= 8 ? 2 : 1;
#define AWQ_V2CASE(W_, S_)                                                  \
  if (waves == W_ && ps == S_) {                                            \
    launch_v2fast<DT, W_, S_, MB>(x, qw, s, z, bias, out, m, n, k, gpad, st); \
    return 0;                                                               \
  }
  AWQ_V2CASE(4, 1) AWQ_V2CASE(4, 2) AWQ_V2CASE(8, 1) AWQ_V2CASE(8, 2) AWQ_V2CASE(16, 1) AWQ_V2CASE(16, 2)
#undef AWQ_V2CASE
  return -1;
}

// reference-layout fast path: 1 <= m <= 8, n % 16 == 0, k % 128 == 0; gpad = rows of the scales / zeros tensors.  -1 if unsupported.
int launch_gemv_v2fast(const void* x, const void* qw, const void* s, const void* z, const void* bias, void* out, int m, int n, int k,
                       int gpad, int dtype, hipStream_t st) {
  if (m < 1 || m > 8 || (n % 16) != 0 || (k % 128) != 0 || k / 128 < 4 || gpad * 128 < k) return -1;
  if ((size_t)n * (size_t)k / 2 >= (1ull << 31) || (size_t)gpad * n * 2 >= (1ull << 31)) return -1;
  if (dtype == 0)
    return m <= 4 ? launch_v2fast_mb<F16, 1>(x, qw, s, z, bias, out, m, n, k, gpad, st)
                  : launch_v2fast_mb<F16, 2>(x, qw, s, z, bias, out, m, n, k, gpad, st);
  return m <= 4 ? launch_v2fast_mb<BF16, 1>(x, qw, s, z, bias, out, m, n, k, gpad, st)
                : launch_v2fast_mb<BF16, 2>(x, qw, s, z, bias, out, m, n, k, gpad, st);
}


// ============================================ part 2: skinny GEMM on v2 buffers ============================================


template <typename DT, int WAVES, int NS, int CB>
__global__ __launch_bounds__(64 * WAVES) void skinny_v2_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                                const uint16_t* __restrict__ scales,
                                                                const uint16_t* __restrict__ zeros,
                                                                const uint16_t* __restrict__ bias, uint16_t* __restrict__ out,
                                                                int M, int N, int K, int gpad) {
  using vec8 = typename DT::vec8;
  constexpr int XB = 4 * CB;             // staging pieces per step: 4 x rows (1 KiB) each
  constexpr int XBYTES = 16 * CB * 256;  // wave-private x region: 16 CB rows x 256 B
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int nb = blockIdx.x, nit = K >> 7, nslab = N >> 4;
  char* xs = smem + wv * XBYTES;

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(qw), 0, (N >> 1) * K, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(scales), 0, gpad * N * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rzr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(zeros), 0, gpad * N * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, M * K * 2, 0x00020000);
  // v2 word of (row 16 sl + i, chunk 4 kg + g) = [4 sl (K/2) + 64 kg] + [(i>>2)(K/2) + 32 (g>>1) + 8 (i&3) + 4 (g&1)]
  const u32 wlane_b = ((u32)(i >> 2) * (u32)(K >> 1) + 32u * (g >> 1) + 8u * (i & 3) + 4u * (g & 1)) * 4u;
  u32 wslab_b[NS], sslab_b[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const u32 sl = (u32)min(nb * NS + s, nslab - 1);
    wslab_b[s] = sl * 4u * (u32)(K >> 1) * 4u;
    sslab_b[s] = sl * 32u;
  }
  const u32 slane_b = (u32)i * 2u;
  u32 xsrc_b[XB];  // staging piece b: LDS row r = 4b + g, slot i  <-  source row min(r, M-1), granule i ^ (r & 15)
#pragma unroll
  for (int b = 0; b < XB; ++b) {
    const int r = 4 * b + g;
    xsrc_b[b] = ((u32)min(r, M - 1) * (u32)K + (u32)((i ^ (r & 15)) * 8)) * 2u;
  }
  const int cnt = (nit - wv + WAVES - 1) / WAVES;  // this wave's steps: kg = wv + WAVES * t

  f32x4 acc[NS][CB];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[s][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 w[NS], xr[XB];
  uint16_t sc[NS], zr[NS];
  auto load_step = [&](int t) {
    const int kg = min(wv + WAVES * t, nit - 1);
#pragma unroll
    for (int b = 0; b < XB; ++b) xr[b] = __builtin_amdgcn_raw_buffer_load_b128(rx, xsrc_b[b], (u32)kg * 256u, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      w[s] = __builtin_amdgcn_raw_buffer_load_b128(rw, wlane_b, wslab_b[s] + (u32)kg * 256u, 2);  // aux 2 = nt
      const u32 so = (u32)kg * (u32)N * 2u + sslab_b[s];
      sc[s] = __builtin_amdgcn_raw_buffer_load_b16(rsc, slane_b, so, 0);
      zr[s] = __builtin_amdgcn_raw_buffer_load_b16(rzr, slane_b, so, 0);
    }
  };
  if (cnt > 0) load_step(0);
  for (int t = 0; t < cnt; ++t) {
#pragma unroll
    for (int b = 0; b < XB; ++b) *reinterpret_cast<u32x4*>(xs + b * 1024 + lane * 16) = xr[b];
    u32x4 wc[NS];
    uint16_t scc[NS], zrc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      wc[s] = w[s];
      scc[s] = sc[s];
      zrc[s] = zr[s];
    }
    if (t + 1 < cnt) load_step(t + 1);  // next step's packed words and x slice stream in under this step's math
    // x operands of the step, shared by the block's NS slabs: xo[c][j] = rows 16c .. 16c+15, k = 32g + 8j .. +8
    vec8 xo[CB][4];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const char* xrow = xs + (c * 16 + i) * 256;
#pragma unroll
      for (int j = 0; j < 4; ++j) xo[c][j] = *reinterpret_cast<const vec8*>(xrow + (((4 * g + j) ^ i) << 4));
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      vec8 op[4];
      dequant_chunk<DT>(wc[s], DT::make_sz(scc[s], zrc[s]), op);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[s][c] = DT::mfma(op[j], xo[c][j], acc[s][c]);
    }
  }

  // ---- split-K reduction across the block's waves (fp32), x regions re-used.  acc[s][c][r] = C[n = 4g + r][m = 16c + i] ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);  // [wave][blk = s * CB + c][r][lane]
  constexpr int NBLK = NS * CB;
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wv * NBLK + s * CB + c) * 4 + r) * 64 + lane] = acc[s][c][r];
  __syncthreads();
  for (int blk = wv; blk < NBLK; blk += WAVES) {
    const int s = blk / CB, c = blk - s * CB;
    const int slab = nb * NS + s, m = 16 * c + i;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < WAVES; ++q) t += red[((q * NBLK + blk) * 4 + r) * 64 + lane];
      v[r] = t;
    }
    if (slab < nslab && m < M) {
      const int nn = slab * 16 + 4 * g;
      uint16_t o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = DT::from_float(v[r]);
        if (bias != nullptr) {  // `out + self.bias` in T (qmodule.py:221)
          float a, b;
          if (DT::id == 0) {
            a = (float)__builtin_bit_cast(_Float16, o[r]);
            b = (float)__builtin_bit_cast(_Float16, bias[nn + r]);
          } else {
            a = __builtin_bit_cast(float, (u32)o[r] << 16);
            b = __builtin_bit_cast(float, (u32)bias[nn + r] << 16);
          }
          o[r] = DT::from_float(a + b);
        }
      }
      *reinterpret_cast<u32x2*>(out + (size_t)m * N + nn) = u32x2{(u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16)};
    }
  }
}

template <typename DT, int WAVES, int NS, int CB>
static void launch_skinny_v2(const void* x, const void* qw, const void* s, const void* z, const void* bias, void* out, int m, int n,
                             int k, int gpad, hipStream_t st) {
  const size_t xbytes = (size_t)WAVES * 16 * CB * 256, rbytes = (size_t)WAVES * NS * CB * 1024;
  const size_t smem = xbytes > rbytes ? xbytes : rbytes;
  auto kern = skinny_v2_kernel<DT, WAVES, NS, CB>;
  static LdsOptIn optin;  // per (kernel instantiation, device)
  if (smem > 64 * 1024) optin.ensure(reinterpret_cast<const void*>(kern));
  const int nslab = n / 16;
  hipLaunchKernelGGL(kern, dim3((nslab + NS - 1) / NS), dim3(64 * WAVES), smem, st, (const uint16_t*)x, (const u32*)qw,
                     (const uint16_t*)s, (const uint16_t*)z, (const uint16_t*)bias, (uint16_t*)out, m, n, k, gpad);
}

template <typename DT>
static void launch_skinny_v2_64(const void* x, const void* qw, const void* s, const void* z, const void* bias, void* out, int m, int n,
                                int k, int gpad, hipStream_t st) {
  const int nslab = n / 16;
  if (m <= 16) {
    if (nslab >= 1024) launch_skinny_v2<DT, 8, 2, 1>(x, qw, s, z, bias, out, m, n, k, gpad, st);
    else launch_skinny_v2<DT, 8, 1, 1>(x, qw, s, z, bias, out, m, n, k, gpad, st);
  } else if (m <= 32) {
    if (nslab >= 512) launch_skinny_v2<DT, 8, 2, 2>(x, qw, s, z, bias, out, m, n, k, gpad, st);
    else launch_skinny_v2<DT, 8, 1, 2>(x, qw, s, z, bias, out, m, n, k, gpad, st);
  } else if (m <= 48) {
    if (nslab >= 512) launch_skinny_v2<DT, 4, 4, 3>(x, qw, s, z, bias, out, m, n, k, gpad, st);
    else launch_skinny_v2<DT, 8, 2, 3>(x, qw, s, z, bias, out, m, n, k, gpad, st);
  } else {
    if (nslab >= 512) launch_skinny_v2<DT, 4, 4, 4>(x, qw, s, z, bias, out, m, n, k, gpad, st);
    else launch_skinny_v2<DT, 8, 2, 4>(x, qw, s, z, bias, out, m, n, k, gpad, st);
  }
}

// 9 <= m <= 255, reference layout, fp16 / bf16; gpad = rows of scales / zeros that may be read (>= k / 128).  -1 if unsupported.
// 65 <= m <= 255 runs as row chunks of <= 64 (the weights are re-streamed per chunk), as on the cdna4 path.
int launch_skinny_v2(const void* x, const void* qw, const void* s, const void* z, const void* bias, void* out, int m, int n, int k,
                     int gpad, int dtype, hipStream_t st) {
  if (m < 1 || m > 255 || (n % 16) != 0 || (k % 128) != 0 || gpad * 128 < k) return -1;
  if ((size_t)m * (size_t)k >= (1ull << 31) || (size_t)n * (size_t)k / 2 >= (1ull << 31) || (size_t)gpad * n * 2 >= (1ull << 31)) return -1;
  if (m > 64 && n >= 16384) return -1;  // wide N: re-streaming the weights per 64-row chunk loses to the 128 x 128 tiles (profiles/r01_skinny_sweep.txt)
  const int chunks = (m + 63) / 64, rows = (m + chunks - 1) / chunks;
  for (int r0 = 0; r0 < m; r0 += rows) {
    const int mr = m - r0 < rows ? m - r0 : rows;
    const uint16_t* xr = static_cast<const uint16_t*>(x) + (size_t)r0 * k;
    uint16_t* orow = static_cast<uint16_t*>(out) + (size_t)r0 * n;
    if (dtype == 0) launch_skinny_v2_64<F16>(xr, qw, s, z, bias, orow, mr, n, k, gpad, st);
    else launch_skinny_v2_64<BF16>(xr, qw, s, z, bias, orow, mr, n, k, gpad, st);
  }
  return 0;
}

}  // namespace awq
