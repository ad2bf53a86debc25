// Decode GEMV / skinny GEMM (1 <= M <= 16) for W4A16, gfx950.
//
// Replaces gemv_kernel<NPerBlock,Batch,BlockSize,GroupSize,T> (reference
// awq/kernels/csrc/quantization_new/gemv/gemv_cuda.cu:74-229).  Design ("gemv"):
//   * HBM-bound: every packed weight byte is read exactly once with 16-byte non-temporal loads.
//   * one wave owns 16 output rows x a K slice; per 16-byte load a lane holds 32 k of ONE row
//     (the v2 interleave already has this shape), so it needs one (scale, scaled_zero) pair.
//   * weights are dequantised in registers with the reference's exact numerics
//     (round_T(q*s+sz)), then fed to v_mfma_f32_16x16x32 as the A operand; the (<=16) activation
//     rows are the B operand.  The MACs cost one MFMA per 512 weights whatever M is, which keeps
//     the VALU budget for unpack+dequant (~1.6 ops/weight fp16, ~2.9 bf16).
//   * K is split across the 4 waves of a block (interleaved 128-k steps) and reduced through LDS
//     in fp32; one rounding to T at the end.
#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

template <typename DT, int U>
__global__ __launch_bounds__(256) void gemv_w4a16_kernel(const uint16_t* __restrict__ x,
                                                         const u32* __restrict__ qw,
                                                         const uint16_t* __restrict__ scales,
                                                         const uint16_t* __restrict__ zeros,
                                                         uint16_t* __restrict__ out, int M, int N, int K) {
  using vec8 = typename DT::vec8;
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int i = lane & 15;  // weight row inside the 16-row slab  /  activation row
  const int g = lane >> 4;  // which 32-k chunk of the 128-k step
  const int n0 = blockIdx.x * 16;
  const int n = min(n0 + i, N - 1);
  const int nit = K / kGroup;  // 128-k steps == quantisation groups

  // per-lane base pointers (advance by `it`)
  const u32* wp = qw + v2_chunk_word(n, g, K);                         // + it*64 words
  const uint16_t* xp = x + (size_t)min(i, M - 1) * K + g * 32;         // + it*128 elements
  const uint16_t* sp = scales + n;                                     // + it*N
  const uint16_t* zp = zeros + n;

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};

  for (int base = wv; base < nit; base += 4 * U) {
    u32x4 wq[U];
    uint16_t sb[U], zb[U];
    u32x4 xa[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = min(base + 4 * u, nit - 1);
      wq[u] = ldg_nt_u32x4(wp + (size_t)it * 64);
      sb[u] = sp[(size_t)it * N];
      zb[u] = zp[(size_t)it * N];
      const u32x4* xv = reinterpret_cast<const u32x4*>(xp + (size_t)it * 128);
#pragma unroll
      for (int j = 0; j < 4; ++j) xa[u][j] = xv[j];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (base + 4 * u < nit) {  // wave-uniform
        vec8 wop[4];
        dequant_chunk<DT>(wq[u], DT::make_sz(sb[u], zb[u]), wop);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = DT::mfma(wop[j], __builtin_bit_cast(vec8, xa[u][j]), acc);
      }
    }
  }

  // cross-wave (split-K) reduction in fp32.  acc[r] = C[n = 4g + r][m = i]
  __shared__ float red[4][4][64];
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wv][r][lane] = acc[r];
  __syncthreads();
  {
    const int r = wv;  // thread (lane, wv) finalises register r of lane
    float s = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
    const int m = i;
    const int nn = n0 + 4 * g + r;
    if (m < M && nn < N) out[(size_t)m * N + nn] = DT::from_float(s);
  }
}

template <typename DT>
static int launch_gemv_t(const void* x, const void* qw, const void* s, const void* z, void* out, int m, int n, int k,
                         hipStream_t st) {
  dim3 grid((n + 15) / 16), block(256);
  const int nit = k / kGroup;
  if (nit >= 16) {
    hipLaunchKernelGGL((gemv_w4a16_kernel<DT, 4>), grid, block, 0, st, (const uint16_t*)x, (const u32*)qw,
                       (const uint16_t*)s, (const uint16_t*)z, (uint16_t*)out, m, n, k);
  } else {
    hipLaunchKernelGGL((gemv_w4a16_kernel<DT, 2>), grid, block, 0, st, (const uint16_t*)x, (const u32*)qw,
                       (const uint16_t*)s, (const uint16_t*)z, (uint16_t*)out, m, n, k);
  }
  return 0;
}

int launch_gemv(const void* x, const void* qw, const void* s, const void* z, void* out, int m, int n, int k, int dtype,
                hipStream_t st) {
  return dtype == 0 ? launch_gemv_t<F16>(x, qw, s, z, out, m, n, k, st)
                    : launch_gemv_t<BF16>(x, qw, s, z, out, m, n, k, st);
}

}  // namespace awq
