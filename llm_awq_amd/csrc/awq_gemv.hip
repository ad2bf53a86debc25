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
//   * K is split across the WAVES waves of a block (interleaved 128-k steps) and reduced through LDS
//     in fp32; one rounding to T at the end.  WAVES is chosen so that ~16 waves per CU are resident
//     even for N = 4096 (256 slabs only).
#include <string.h>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

// XMODE: how the activation (B) operand reaches the lanes
//   0 = every lane loads (rows clamped to M-1)            1 = only lanes with row < M load (others keep 0)
//   2 = x staged once per block in LDS (padded rows), ds_read_b128 per MFMA
template <typename DT, int U, int WAVES, int XMODE, bool STREAM_ONLY>
__global__ __launch_bounds__(64 * WAVES) void gemv_w4a16_kernel(const uint16_t* __restrict__ x,
                                                                 const u32* __restrict__ qw,
                                                                 const uint16_t* __restrict__ scales,
                                                                 const uint16_t* __restrict__ zeros,
                                                                 uint16_t* __restrict__ out, int M, int N, int K) {
  using vec8 = typename DT::vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int i = lane & 15;  // weight row inside the 16-row slab  /  activation row
  const int g = lane >> 4;  // which 32-k chunk of the 128-k step
  const int n0 = blockIdx.x * 16;
  const int n = min(n0 + i, N - 1);
  const int nit = K / kGroup;  // 128-k steps == quantisation groups

  // per-lane base pointers (advance by `it`)
  const u32* wp = qw + v2_chunk_word(n, g, K);                  // + it*64 words
  const int mrow = min(i, M - 1);
  const uint16_t* xp = x + (size_t)mrow * K + g * 32;           // + it*128 elements
  const uint16_t* sp = scales + n;                              // + it*N
  const uint16_t* zp = zeros + n;
  const int xrow_bytes = 2 * K + 16;                            // padded LDS row (XMODE 2)
  char* xs = smem + WAVES * 1024;                               // after the reduction buffer

  if (XMODE == 2) {
    // cooperative stage of x[0..M) into LDS, 16 B per thread per step
    const int per_row = K / 8;  // 16-byte granules per row
    for (int q = threadIdx.x; q < M * per_row; q += 64 * WAVES) {
      const int r = q / per_row, c = q % per_row;
      *reinterpret_cast<u32x4*>(xs + r * xrow_bytes + c * 16) = *reinterpret_cast<const u32x4*>(x + (size_t)r * K + c * 8);
    }
    __syncthreads();
  }

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  u32x4 xa[U][4];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) xa[u][j] = u32x4{0u, 0u, 0u, 0u};
  u32 sink = 0;

  for (int base = wv; base < nit; base += WAVES * U) {
    u32x4 wq[U];
    uint16_t sb[U], zb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = min(base + WAVES * u, nit - 1);
      wq[u] = ldg_nt_u32x4(wp + (size_t)it * 64);
      sb[u] = sp[(size_t)it * N];
      zb[u] = zp[(size_t)it * N];
      if (XMODE == 0) {
        const u32x4* xv = reinterpret_cast<const u32x4*>(xp + (size_t)it * 128);
#pragma unroll
        for (int j = 0; j < 4; ++j) xa[u][j] = xv[j];
      } else if (XMODE == 1) {
        if (i < M) {
          const u32x4* xv = reinterpret_cast<const u32x4*>(xp + (size_t)it * 128);
#pragma unroll
          for (int j = 0; j < 4; ++j) xa[u][j] = xv[j];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (base + WAVES * u < nit) {  // wave-uniform
        if (STREAM_ONLY) {
          sink ^= wq[u].x ^ wq[u].y ^ wq[u].z ^ wq[u].w ^ sb[u] ^ zb[u];
        } else {
          if (XMODE == 2) {
            const u32x4* xv =
                reinterpret_cast<const u32x4*>(xs + mrow * xrow_bytes + ((base + WAVES * u) * 128 + g * 32) * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) xa[u][j] = xv[j];
          }
          vec8 wop[4];
          dequant_chunk<DT>(wq[u], DT::make_sz(sb[u], zb[u]), wop);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = DT::mfma(wop[j], __builtin_bit_cast(vec8, xa[u][j]), acc);
        }
      }
    }
  }
  if (STREAM_ONLY) acc[0] = __builtin_bit_cast(float, sink & 0x3fffffffu);

  // cross-wave (split-K) reduction in fp32.  acc[r] = C[n = 4g + r][m = i]
  float(*red)[4][64] = reinterpret_cast<float(*)[4][64]>(smem);  // [WAVES][4][64]
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wv][r][lane] = acc[r];
  __syncthreads();
  if (wv < 4) {
    const int r = wv;  // thread (lane, wv) finalises register r of lane
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s += red[w][r][lane];
    const int m = i;
    const int nn = n0 + 4 * g + r;
    if (m < M && nn < N) out[(size_t)m * N + nn] = DT::from_float(s);
  }
}

namespace {
struct GemvTune {
  int waves = 0;  // 0 = auto
  int unroll = 0;
  int xmode = 1;
  int stream_only = 0;
} g_tune;
}  // namespace

int gemv_tune_set(const char* key, int value) {
  if (!strcmp(key, "gemv_waves")) g_tune.waves = value;
  else if (!strcmp(key, "gemv_unroll")) g_tune.unroll = value;
  else if (!strcmp(key, "gemv_xmode")) g_tune.xmode = value;
  else if (!strcmp(key, "gemv_stream_only")) g_tune.stream_only = value;
  else return -1;
  return 0;
}

template <typename DT, int U, int WAVES, int XMODE, bool SO>
static void launch_one(const void* x, const void* qw, const void* s, const void* z, void* out, int m, int n, int k,
                       hipStream_t st) {
  dim3 grid((n + 15) / 16), block(64 * WAVES);
  size_t smem = WAVES * 1024 + (XMODE == 2 ? (size_t)m * (2 * k + 16) : 0);
  hipLaunchKernelGGL((gemv_w4a16_kernel<DT, U, WAVES, XMODE, SO>), grid, block, smem, st, (const uint16_t*)x,
                     (const u32*)qw, (const uint16_t*)s, (const uint16_t*)z, (uint16_t*)out, m, n, k);
}

template <typename DT, int U, int WAVES>
static void launch_x(int xmode, bool so, const void* x, const void* qw, const void* s, const void* z, void* out, int m,
                     int n, int k, hipStream_t st) {
  if (so) return launch_one<DT, U, WAVES, 1, true>(x, qw, s, z, out, m, n, k, st);
  switch (xmode) {
    case 0: return launch_one<DT, U, WAVES, 0, false>(x, qw, s, z, out, m, n, k, st);
    case 2: return launch_one<DT, U, WAVES, 2, false>(x, qw, s, z, out, m, n, k, st);
    default: return launch_one<DT, U, WAVES, 1, false>(x, qw, s, z, out, m, n, k, st);
  }
}

template <typename DT>
static int launch_gemv_t(const void* x, const void* qw, const void* s, const void* z, void* out, int m, int n, int k,
                         hipStream_t st) {
  const int nit = k / kGroup;
  const int slabs = (n + 15) / 16;
  int waves = g_tune.waves;
  if (waves == 0) waves = slabs >= 768 ? 4 : (slabs >= 320 ? 8 : 16);  // ~16 resident waves per CU
  while (waves > 4 && waves > nit) waves >>= 1;
  int unroll = g_tune.unroll;
  if (unroll == 0) unroll = (nit / waves >= 4) ? 4 : 2;
  int xmode = g_tune.xmode;
  if (xmode == 2 && (size_t)m * (2 * k + 16) + waves * 1024 > 60 * 1024) xmode = 1;
  const bool so = g_tune.stream_only != 0;
#define AWQ_GEMV_CASE(W_, U_)                                                   \
  if (waves == W_ && unroll == U_) {                                            \
    launch_x<DT, U_, W_>(xmode, so, x, qw, s, z, out, m, n, k, st);             \
    return 0;                                                                   \
  }
  AWQ_GEMV_CASE(4, 2) AWQ_GEMV_CASE(4, 4) AWQ_GEMV_CASE(8, 2) AWQ_GEMV_CASE(8, 4) AWQ_GEMV_CASE(16, 2)
  AWQ_GEMV_CASE(16, 4) AWQ_GEMV_CASE(4, 8) AWQ_GEMV_CASE(8, 1) AWQ_GEMV_CASE(16, 1)
#undef AWQ_GEMV_CASE
  launch_x<DT, 2, 4>(xmode, so, x, qw, s, z, out, m, n, k, st);
  return 0;
}

int launch_gemv(const void* x, const void* qw, const void* s, const void* z, void* out, int m, int n, int k, int dtype,
                hipStream_t st) {
  return dtype == 0 ? launch_gemv_t<F16>(x, qw, s, z, out, m, n, k, st)
                    : launch_gemv_t<BF16>(x, qw, s, z, out, m, n, k, st);
}

}  // namespace awq
