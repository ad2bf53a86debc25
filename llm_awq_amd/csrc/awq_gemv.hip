// Decode GEMV / skinny GEMM (1 <= M <= 16) for W4A16, gfx950.
//
// Replaces gemv_kernel<NPerBlock,Batch,BlockSize,GroupSize,T> (reference
// awq/kernels/csrc/quantization_new/gemv/gemv_cuda.cu:74-229).  Design ("gemv"):
//   * HBM-bound: every packed weight byte is read exactly once with 16-byte non-temporal loads,
//     kept PF deep in flight per wave (register ring) so the stream never drains while a wave
//     dequantises.
//   * one wave owns 16 output rows x a K slice; per 16-byte load a lane holds 32 k of ONE row
//     (the v2 interleave already has this shape), so it needs one (scale, scaled_zero) pair.
//   * x (<= 16 rows), and the slab's scales / scaled_zeros are staged ONCE per block in LDS; the only
//     global loads in the loop are the packed weights.
//   * weights are dequantised in registers with the reference's exact numerics
//     (round_T(q*s+sz)), then fed to v_mfma_f32_16x16x32 as the A operand; the activation rows are
//     the B operand.  One MFMA per 512 weights whatever M is.
//   * K is split across the WAVES waves of a block (interleaved 128-k steps) and reduced through LDS
//     in fp32; one rounding to T at the end.
#include <string.h>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

// XLDS: true  = x rows staged in LDS (padded rows), read with ds_read_b128 per MFMA
//       false = only lanes with row < M load x from global (L2) per step (others keep 0)
template <typename DT, int PF, int WAVES, bool XLDS, int PROBE>
__global__ __launch_bounds__(64 * WAVES) void gemv_w4a16_kernel(const uint16_t* __restrict__ x,
                                                                 const u32* __restrict__ qw,
                                                                 const uint16_t* __restrict__ scales,
                                                                 const uint16_t* __restrict__ zeros,
                                                                 uint16_t* __restrict__ out, int M, int N, int K,
                                                                 int order) {
  using vec8 = typename DT::vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int i = lane & 15;  // weight row inside the 16-row slab  /  activation row
  const int g = lane >> 4;  // which 32-k chunk of the 128-k step
  const int n0 = blockIdx.x * 16;
  const int n = min(n0 + i, N - 1);
  const int nit = K / kGroup;  // 128-k steps == quantisation groups

  // this wave's steps: it(t) = (first + stride * t + rot) % nit, t in [0, cnt)
  int first, stride, cnt;
  if (order == 1) {  // contiguous K slice per wave
    const int per = (nit + WAVES - 1) / WAVES;
    first = wv * per;
    stride = 1;
    cnt = max(0, min(per, nit - first));
  } else {  // interleaved
    first = wv;
    stride = WAVES;
    cnt = (nit - wv + WAVES - 1) / WAVES;
  }
  const int rot = (order == 2) ? (int)((blockIdx.x * 5u) % (unsigned)nit) : 0;
  auto step_of = [&](int t) {
    int it = first + stride * t + rot;
    return it >= nit ? it - nit : it;
  };

  const u32* wp = qw + v2_chunk_word(n, g, K);  // + it*64 words
  const int mrow = min(i, M - 1);

  // ---- LDS carve: [reduce WAVES*1KiB][sz pairs nit*16*4B][x rows] ----
  float(*red)[4][64] = reinterpret_cast<float(*)[4][64]>(smem);
  u32* szs = reinterpret_cast<u32*>(smem + WAVES * 1024);  // [nit][16] {scale | zero << 16}
  char* xs = smem + WAVES * 1024 + nit * 64;
  const int xrow_bytes = 2 * K + 16;

  // ---- prologue: start the weight stream first, then stage the small operands ----
  u32x4 wq[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int it = min(step_of(min(u, max(cnt - 1, 0))), nit - 1);
    wq[u] = ldg_nt_u32x4(wp + (size_t)it * 64);
  }
  for (int q = threadIdx.x; q < nit * 16; q += 64 * WAVES) {
    const int gi = q >> 4, c = q & 15;
    const int nn = min(n0 + c, N - 1);
    szs[q] = (u32)scales[(size_t)gi * N + nn] | ((u32)zeros[(size_t)gi * N + nn] << 16);
  }
  if (XLDS) {
    const int per_row = K / 8;  // 16-byte granules per row
    for (int q = threadIdx.x; q < M * per_row; q += 64 * WAVES) {
      const int r = q / per_row, c = q - r * per_row;
      *reinterpret_cast<u32x4*>(xs + r * xrow_bytes + c * 16) = *reinterpret_cast<const u32x4*>(x + (size_t)r * K + c * 8);
    }
  }
  __syncthreads();

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  u32 sink = 0;
  u32x4 xg[4] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};

  for (int t0 = 0; t0 < cnt; t0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int t = t0 + u;
      if (t < cnt) {  // wave-uniform
        const int it = step_of(t);
        const u32x4 w = wq[u];
        if (t + PF < cnt) wq[u] = ldg_nt_u32x4(wp + (size_t)step_of(t + PF) * 64);
        if (PROBE) {
          sink ^= w.x ^ w.y ^ w.z ^ w.w;
        } else {
          const u32 szv = szs[it * 16 + i];
          const u32x4* xv;
          if (XLDS) {
            xv = reinterpret_cast<const u32x4*>(xs + mrow * xrow_bytes + (it * 128 + g * 32) * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) xg[j] = xv[j];
          } else if (i < M) {
            xv = reinterpret_cast<const u32x4*>(x + (size_t)mrow * K + it * 128 + g * 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) xg[j] = xv[j];
          }
          vec8 wop[4];
          dequant_chunk<DT>(w, DT::make_sz((uint16_t)(szv & 0xFFFFu), (uint16_t)(szv >> 16)), wop);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc = DT::mfma(wop[j], __builtin_bit_cast(vec8, xg[j]), acc);
        }
      }
    }
  }
  if (PROBE) acc[0] = __builtin_bit_cast(float, sink & 0x3fffffffu);

  // cross-wave (split-K) reduction in fp32.  acc[r] = C[n = 4g + r][m = i]
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

// ---- calibration probes (experiments only): ideal linear 16-byte streaming read, and an empty kernel ----
__global__ __launch_bounds__(256) void probe_linear_read_kernel(const u32x4* __restrict__ p, size_t n16, u32* out) {
  u32 acc = 0;
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; idx + 3 * stride < n16; idx += 4 * stride) {
    u32x4 a = __builtin_nontemporal_load(p + idx), b = __builtin_nontemporal_load(p + idx + stride);
    u32x4 c = __builtin_nontemporal_load(p + idx + 2 * stride), d = __builtin_nontemporal_load(p + idx + 3 * stride);
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; idx < n16; idx += stride) {
    u32x4 a = __builtin_nontemporal_load(p + idx);
    acc ^= a.x ^ a.y ^ a.z ^ a.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void probe_null_kernel(u32* out) {
  if (out == nullptr) out[1] = 0;
}

namespace {
struct GemvTune {
  int waves = 0;  // 0 = auto
  int pf = 0;
  int xlds = 1;
  int probe = 0;  // 1 = stream-only in the real access pattern, 2 = linear read, 3 = null kernel
  int order = 0;
  int probe_blocks = 2048;
} g_tune;
bool g_attr_done = false;
}  // namespace

int gemv_tune_set(const char* key, int value) {
  if (!strcmp(key, "gemv_waves")) g_tune.waves = value;
  else if (!strcmp(key, "gemv_pf")) g_tune.pf = value;
  else if (!strcmp(key, "gemv_xlds")) g_tune.xlds = value;
  else if (!strcmp(key, "gemv_probe")) g_tune.probe = value;
  else if (!strcmp(key, "gemv_order")) g_tune.order = value;
  else if (!strcmp(key, "gemv_probe_blocks")) g_tune.probe_blocks = value;
  else return -1;
  return 0;
}

template <typename DT, int PF, int WAVES, bool XLDS, int PROBE>
static void launch_one(const void* x, const void* qw, const void* s, const void* z, void* out, int m, int n, int k,
                       hipStream_t st) {
  dim3 grid((n + 15) / 16), block(64 * WAVES);
  const size_t smem = WAVES * 1024 + (size_t)(k / kGroup) * 64 + (XLDS ? (size_t)m * (2 * k + 16) : 0);
  auto kern = gemv_w4a16_kernel<DT, PF, WAVES, XLDS, PROBE>;
  if (smem > 64 * 1024) {
    static bool done = false;  // per instantiation
    if (!done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      done = true;
    }
  }
  hipLaunchKernelGGL(kern, grid, block, smem, st, (const uint16_t*)x, (const u32*)qw, (const uint16_t*)s,
                     (const uint16_t*)z, (uint16_t*)out, m, n, k, g_tune.order);
}

template <typename DT, int PF, int WAVES>
static void launch_x(bool xlds, int probe, const void* x, const void* qw, const void* s, const void* z, void* out, int m,
                     int n, int k, hipStream_t st) {
  if (probe == 1) return launch_one<DT, PF, WAVES, false, 1>(x, qw, s, z, out, m, n, k, st);
  if (xlds) return launch_one<DT, PF, WAVES, true, 0>(x, qw, s, z, out, m, n, k, st);
  return launch_one<DT, PF, WAVES, false, 0>(x, qw, s, z, out, m, n, k, st);
}

template <typename DT>
static int launch_gemv_t(const void* x, const void* qw, const void* s, const void* z, void* out, int m, int n, int k,
                         hipStream_t st) {
  const int nit = k / kGroup;
  const int slabs = (n + 15) / 16;
  if (g_tune.probe == 2) {
    hipLaunchKernelGGL(probe_linear_read_kernel, dim3(g_tune.probe_blocks), dim3(256), 0, st, (const u32x4*)qw,
                       (size_t)n * k / 32, (u32*)out);
    return 0;
  }
  if (g_tune.probe == 3) {
    hipLaunchKernelGGL(probe_null_kernel, dim3(256), dim3(256), 0, st, (u32*)out);
    return 0;
  }
  int waves = g_tune.waves;
  if (waves == 0) waves = slabs >= 768 ? 4 : (slabs >= 320 ? 8 : 16);  // ~16 resident waves per CU
  while (waves > 4 && waves > nit) waves >>= 1;
  int pf = g_tune.pf;
  if (pf == 0) pf = (nit / waves >= 8) ? 8 : 4;
  // x in LDS when it fits next to the reduce + scale buffers (<= 150 KiB per block)
  bool xlds = g_tune.xlds != 0;
  if (xlds && (size_t)m * (2 * k + 16) + (size_t)nit * 64 + waves * 1024 > 150 * 1024) xlds = false;
#define AWQ_GEMV_CASE(W_, P_)                                                      \
  if (waves == W_ && pf == P_) {                                                   \
    launch_x<DT, P_, W_>(xlds, g_tune.probe, x, qw, s, z, out, m, n, k, st);       \
    return 0;                                                                      \
  }
  AWQ_GEMV_CASE(4, 4) AWQ_GEMV_CASE(4, 8) AWQ_GEMV_CASE(8, 4) AWQ_GEMV_CASE(8, 8) AWQ_GEMV_CASE(16, 4)
  AWQ_GEMV_CASE(16, 8) AWQ_GEMV_CASE(4, 2) AWQ_GEMV_CASE(8, 2) AWQ_GEMV_CASE(16, 2)
#undef AWQ_GEMV_CASE
  launch_x<DT, 4, 4>(xlds, g_tune.probe, x, qw, s, z, out, m, n, k, st);
  return 0;
}

int launch_gemv(const void* x, const void* qw, const void* s, const void* z, void* out, int m, int n, int k, int dtype,
                hipStream_t st) {
  return dtype == 0 ? launch_gemv_t<F16>(x, qw, s, z, out, m, n, k, st)
                    : launch_gemv_t<BF16>(x, qw, s, z, out, m, n, k, st);
}

}  // namespace awq
