// Decode GEMV on the REFERENCE (v2) interleave, 1 <= M <= 8, fp16 and bf16 (gfx950): the structure of the cdna4 fast
// path (awq_gemv_cdna4.hip: two-deep software pipeline, raw buffer loads with scalar tile offsets, wave-private swizzled
// x staging, split-K through LDS, fused bias) applied to un-repacked checkpoints, so gemv_forward_cuda_new on raw
// reference buffers (awq/kernels/csrc/quantization_new/gemv/gemv_cuda.cu:245-338) gets it for fp16 models too.
//
// Differences forced by the layout: a lane's 16 bytes are one output row x one 32-k chunk (qmodule.py:26-65), so a wave
// load is 4 x 256 B instead of one contiguous KiB, the weights are dequantised on the VALU with the reference's
// numerics (dequant_chunk: round_T(q*s + sz), packed fp16 math / fp32 fma + v_cvt_pk for bf16), and scales / zeros
// come from the [Gpad, N] tensors (two 16-bit loads per step).
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

}  // namespace awq
