// Standalone decode-GEMV structure microbenchmark (experiments only; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I llm_awq_amd/csrc tools/ubench/gemv_ubench.hip -o gemv_ubench
// Times, per (N, K) shape, over R rotating weight copies (> the 256 MB Infinity Cache):
//   lin     ideal 16-B/lane linear read (floor)
//   g<DQ>   slab kernel: block = one 16-row slab, WAVES waves split K, every wave issues ALL its loads up
//           front (S steps of 1 KiB), wave-private x staging (no block barrier before the reduce)
//           DQ 0 = stream only, 1 = matrix-core dequant (cdna4 nibble order), 2 = dot2 dequant (v2 chunk order)
//   km      tile order: 0 = slab-major ((nb*nit + kg) KiB), 1 = k-major ((kg*nslab + nb) KiB)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "awq_device.hpp"

using namespace awq;

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);   \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

__global__ __launch_bounds__(256) void lin_kernel(const u32x4* __restrict__ p, size_t n16, u32* out) {
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
__global__ void null_kernel(u32* out) {
  if (out == nullptr) out[1] = 0;
}

// dot2 dequant of one v2-order chunk (lane = row i, 32 k): 4 operands, op[j] = k 8j..8j+7 of the chunk
__device__ __forceinline__ void dot2_chunk(const u32x4& w, u32 sb, u32 zb, bf16x8 (&op)[4]) {
  const float sf = __builtin_bit_cast(float, sb << 16), zf = __builtin_bit_cast(float, zb << 16);
  const float c = __builtin_fmaf(-128.0f, sf, zf);
  const bf16x2 slo = __builtin_bit_cast(bf16x2, sb), shi = __builtin_bit_cast(bf16x2, sb << 16);
  u32 kMagic = 0x43004300u, kMask = 0x000F000Fu;
  asm volatile("" : "+v"(kMagic));
  asm volatile("" : "+s"(kMask));
  const u32 ws[4] = {w.x, w.y, w.z, w.w};
  u32 r[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16x2 pr = __builtin_bit_cast(bf16x2, ((ws[a] >> (4 * i)) & kMask) | kMagic);
      const float lo = __builtin_amdgcn_fdot2_f32_bf16(pr, slo, c, false);
      const float hi = __builtin_amdgcn_fdot2_f32_bf16(pr, shi, c, false);
      bf16x2 o = {(__bf16)lo, (__bf16)hi};
      r[i][a] = __builtin_bit_cast(u32, o);
    }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u32x4 v = {r[j][0], r[j][1], r[j][2], r[j][3]};
    op[j] = __builtin_bit_cast(bf16x8, v);
  }
}

template <int WAVES, int S, int DQ, int NT>
__global__ __launch_bounds__(64 * WAVES) void gemv2(const u32* __restrict__ qw, const u32* __restrict__ szp,
                                                    const uint16_t* __restrict__ x, uint16_t* __restrict__ out, int M,
                                                    int N, int K, int kmajor) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int nb = blockIdx.x, nit = K >> 7, nslab = N >> 4;
  // LDS: [WAVES][S][4 rows][256 B] x slices (wave private), reused for the reduction
  const int rows = min(M, 4), xstep = rows * 256;
  char* xs = smem + wv * (S * xstep);

  if (kmajor & 4) __builtin_amdgcn_s_setprio(3);  // experiment: the load-issue phase of a new wave outranks older waves' math
  // ---- x slices of this wave's steps: lane -> (row lane>>4 (clamped), granule lane&15) ----
  u32x4 xr[S];
  if (DQ != 0) {
    const int xrow = min(lane >> 4, M - 1);
#pragma unroll
    for (int t = 0; t < S; ++t) {
      const int kg = min(wv + WAVES * t, nit - 1);
      xr[t] = *reinterpret_cast<const u32x4*>(x + (size_t)xrow * K + kg * 128 + (lane & 15) * 8);
    }
  }
  u32x4 w[S];
  u32 sz[S];
#pragma unroll
  for (int t = 0; t < S; ++t) {
    const int kg = min(wv + WAVES * t, nit - 1);
    size_t tile = (kmajor & 1) ? (size_t)kg * nslab + nb : (size_t)nb * nit + kg;
    if (kmajor & 2) tile &= 1023;
    const u32x4* p = reinterpret_cast<const u32x4*>(qw + tile * 256 + lane * 4);
    w[t] = NT ? __builtin_nontemporal_load(p) : *p;
    sz[t] = szp[tile * 16 + i];
  }
  if (kmajor & 4) __builtin_amdgcn_s_setprio(0);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (DQ == 6 || DQ == 7) {
    // independent VALU-only (6) / MFMA-only (7) work of roughly the dequant's instruction count per step
    u32 a0 = lane * 0x9E3779B9u, a1 = wv, a2 = nb, a3 = 0x12345u;
    u32x4 fake = {a0, a1, a2, a3};
    bf16x8 xo = __builtin_bit_cast(bf16x8, fake);
#pragma unroll
    for (int t = 0; t < S; ++t) {
      if (DQ == 6) {
#pragma unroll
        for (int r = 0; r < 13; ++r) {
          a0 = ((a0 >> 4) & 0x000F000Fu) | 0x43004300u;
          a1 = ((a1 >> 8) & 0x000F000Fu) | (a0 + r);
          a2 = ((a2 >> 12) & 0x000F000Fu) | (a1 ^ t);
          a3 = ((a3 >> 4) & 0x000F000Fu) | a2;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 12; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xo, xo, acc, 0, 0, 0);
      }
    }
    u32 sink = a0 ^ a1 ^ a2 ^ a3;
#pragma unroll
    for (int t = 0; t < S; ++t) sink ^= w[t].x ^ w[t].y ^ w[t].z ^ w[t].w ^ sz[t];
    acc[1] += __builtin_bit_cast(float, sink & 0x3fffffffu);
  } else if (DQ == 5) {
    Cdna4Dequant cd;
    cd.init(lane);
    u32x4 fake = {(u32)lane * 0x9E3779B9u, (u32)wv, (u32)nb, 0x12345u};
    bf16x8 xo = __builtin_bit_cast(bf16x8, fake);
#pragma unroll
    for (int t = 0; t < S; ++t) {
      bf16x8 op[4];
      cd.tile(fake, (uint16_t)(0x3C00 + t), (uint16_t)0xBC00, op);
#pragma unroll
      for (int a = 0; a < 4; ++a) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op[a], xo, acc, 0, 0, 0);
      fake.x += 0x01010101u;
      fake.z ^= fake.x >> 3;
    }
    u32 sink = 0;
#pragma unroll
    for (int t = 0; t < S; ++t) sink ^= w[t].x ^ w[t].y ^ w[t].z ^ w[t].w ^ sz[t];
    acc[1] += __builtin_bit_cast(float, sink & 0x3fffffffu);
  } else if (DQ == 0) {
    u32 sink = 0;
#pragma unroll
    for (int t = 0; t < S; ++t) sink ^= w[t].x ^ w[t].y ^ w[t].z ^ w[t].w ^ sz[t];
    acc[0] = __builtin_bit_cast(float, sink & 0x3fffffffu);
  } else {
#pragma unroll
    for (int t = 0; t < S; ++t)
      if ((lane >> 4) < rows) *reinterpret_cast<u32x4*>(xs + t * xstep + lane * 16) = xr[t];
    Cdna4Dequant cd;
    if (DQ == 1 || DQ == 4) cd.init(lane);
    const int mrow = min(i, rows - 1);
#pragma unroll
    for (int t = 0; t < S; ++t) {
      const bool valid = wv + WAVES * t < nit;
      const u32x4* xrow = reinterpret_cast<const u32x4*>(xs + t * xstep + mrow * 256);
      bf16x8 op[4];
      const u32 sb = valid ? (sz[t] & 0xFFFFu) : 0u, zb = valid ? (sz[t] >> 16) : 0u;
      if (DQ == 3) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          u32x4 v = {w[t].x + a, w[t].y ^ sb, w[t].z, w[t].w};
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v), __builtin_bit_cast(bf16x8, xrow[4 * a + g]), acc, 0, 0, 0);
        }
      } else if (DQ == 4) {  // dequant only: operands xor-folded, one product MFMA per step
        cd.tile(w[t], (uint16_t)sb, (uint16_t)zb, op);
        u32x4 v = __builtin_bit_cast(u32x4, op[0]) ^ __builtin_bit_cast(u32x4, op[1]) ^ __builtin_bit_cast(u32x4, op[2]) ^ __builtin_bit_cast(u32x4, op[3]);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, v), __builtin_bit_cast(bf16x8, xrow[g]), acc, 0, 0, 0);
      } else if (DQ == 1) {
        cd.tile(w[t], (uint16_t)sb, (uint16_t)zb, op);
#pragma unroll
        for (int a = 0; a < 4; ++a)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op[a], __builtin_bit_cast(bf16x8, xrow[4 * a + g]), acc, 0, 0, 0);
      } else {
        dot2_chunk(w[t], sb, zb, op);
#pragma unroll
        for (int a = 0; a < 4; ++a)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(op[a], __builtin_bit_cast(bf16x8, xrow[4 * g + a]), acc, 0, 0, 0);
      }
    }
  }
  // ---- cross-wave reduction (fp32) ----
  __syncthreads();
  float(*red)[4][64] = reinterpret_cast<float(*)[4][64]>(smem);
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wv][r][lane] = acc[r];
  __syncthreads();
  if (wv < 4) {
    const int r = wv;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < WAVES; ++q) s += red[q][r][lane];
    const int nn = nb * 16 + 4 * g + r;
    if (i < M) out[(size_t)i * N + nn] = __builtin_bit_cast(uint16_t, (__bf16)s);
  }
}

template <typename F>
static float time_us(F&& launch, int R, int iters);

template <int KIND>
__global__ __launch_bounds__(256) void mfma_rate_kernel(float* out, int iters) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
  u32x4 a = {threadIdx.x, 2, 3, 4}, b = {5, 6, 7, threadIdx.x};
  typedef short s16x4_ __attribute__((ext_vector_type(4)));
  u32x2 a2 = {threadIdx.x, 1}, b2 = {3, threadIdx.x};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (KIND == 0) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[u], 0, 0, 0);
      if (KIND == 1) acc[u] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_, a2), __builtin_bit_cast(s16x4_, b2), acc[u], 0, 0, 0);
      if (KIND == 2) acc[u] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(__builtin_bit_cast(s16x4_, a2), __builtin_bit_cast(s16x4_, b2), acc[u], 0, 0, 0);
      if (KIND == 3) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(__builtin_bit_cast(long, a2), __builtin_bit_cast(long, b2), acc[u], 0, 0, 0);
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}
template <int KIND>
static void mfma_rate(const char* name) {
  float* o;
  CK(hipMalloc(&o, 1024 * 256 * 4));
  const int iters = 4096;
  auto l = [&](int) { hipLaunchKernelGGL(mfma_rate_kernel<KIND>, dim3(1024), dim3(256), 0, 0, o, iters); };
  const float us = time_us(l, 1, 2);
  // 1024 blocks x 4 waves = 4096 waves over 1024 SIMDs = 4 waves per SIMD, each 4*iters MFMAs
  const double cyc = (double)us * 1e-6 * 2.4e9 / (4.0 * 4 * iters);
  printf("mfma rate %-28s %8.1f us  -> %.2f cycles per MFMA per SIMD (at 2.4 GHz)\n", name, us, cyc);
  CK(hipFree(o));
}

// ------------------------------------------------------------------------------------------------
static uint32_t rng_state = 12345;
static inline uint32_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 17;
  rng_state ^= rng_state << 5;
  return rng_state;
}
static inline float bf2f(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

struct Shape {
  int N, K;
};

template <typename F>
static float time_us(F&& launch, int R, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int r = 0; r < R; ++r) launch(r);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0, 0));
    for (int it = 0; it < iters; ++it)
      for (int r = 0; r < R; ++r) launch(r);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = fminf(best, ms * 1e3f / (iters * R));
  }
  CK(hipGetLastError());
  return best;
}

struct Ctx {
  int N, K, M, R;
  std::vector<u32*> qw, szp;
  uint16_t* x;
  uint16_t* out;
  std::vector<u32> h_qw, h_szp;
  std::vector<uint16_t> h_x;
};

// host decode of weight (n, k) of tile layouts
static int q_cdna4(const std::vector<u32>& qw, size_t tile, int c, int kk) {
  const int g = c >> 2, j = c & 3, a = kk >> 5, r32 = kk & 31, b8 = r32 >> 3, e = r32 & 7, th = e >> 2, rr = e & 3;
  const int lane = 16 * g + 4 * b8 + rr, p = (2 * th + (j >> 1)) + 4 * (j & 1);
  return (qw[tile * 256 + lane * 4 + a] >> (4 * p)) & 0xF;
}
static int q_v2chunk(const std::vector<u32>& qw, size_t tile, int c, int kk) {
  const int g = kk >> 5, kl = kk & 31, lane = 16 * g + c;
  const int a = (kl & 7) >> 1, nib = (kl >> 3) + 4 * (kl & 1);
  return (qw[tile * 256 + lane * 4 + a] >> (4 * nib)) & 0xF;
}

template <int WAVES, int S, int DQ, int NT>
static void run(Ctx& c, int kmajor, const char* tag) {
  const int nit = c.K / 128;
  if (WAVES * S < nit || WAVES * (S - 1) >= nit) return;  // S must be ceil(nit / WAVES)
  auto kern = gemv2<WAVES, S, DQ, NT>;
  const size_t xb = (size_t)WAVES * S * 256 * (c.M < 4 ? c.M : 4);
  const size_t smem = xb > (size_t)WAVES * 1024 ? xb : (size_t)WAVES * 1024;
  if (smem > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  auto launch = [&](int r) {
    hipLaunchKernelGGL(kern, dim3(c.N / 16), dim3(64 * WAVES), smem, 0, c.qw[r], c.szp[r], c.x, c.out, c.M, c.N, c.K, kmajor);
  };
  const float us = time_us(launch, c.R, 4);
  const double bytes = (double)c.N * c.K / 2 + 4.0 * c.N * nit + 2.0 * c.M * c.K + 2.0 * c.M * c.N;
  // correctness of the full variants against a host fp64 contraction of the T-rounded weights (copy 0, first rows)
  double maxrel = -1;
  if (DQ == 1 || DQ == 2) {
    launch(0);
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> h_out((size_t)c.M * c.N);
    CK(hipMemcpy(h_out.data(), c.out, h_out.size() * 2, hipMemcpyDeviceToHost));
    const int nslab = c.N / 16;
    maxrel = 0;
    double ref_norm = 0, err_norm = 0;
    for (int n = 0; n < c.N; n += 37) {
      const int nb = n / 16, cc = n % 16;
      for (int m = 0; m < c.M; ++m) {
        double s = 0;
        for (int k = 0; k < c.K; ++k) {
          const int kg = k / 128;
          const size_t tile = kmajor ? (size_t)kg * nslab + nb : (size_t)nb * nit + kg;
          const int q = DQ == 1 ? q_cdna4(c.h_qw, tile, cc, k % 128) : q_v2chunk(c.h_qw, tile, cc, k % 128);
          const u32 sz = c.h_szp[tile * 16 + cc];
          const float wf = fmaf((float)q, bf2f(sz & 0xFFFF), bf2f(sz >> 16));
          s += (double)bf2f(f2bf(wf)) * (double)bf2f(c.h_x[(size_t)m * c.K + k]);
        }
        const double got = bf2f(h_out[(size_t)m * c.N + n]);
        ref_norm += s * s;
        err_norm += (got - s) * (got - s);
      }
    }
    maxrel = sqrt(err_norm / (ref_norm + 1e-30));
  }
  printf("N=%6d K=%6d M=%d %-5s waves=%2d S=%2d dq=%d nt=%d kmajor=%d  %7.2f us  %7.1f GB/s  %5.1f%%  relerr=%.2e\n", c.N, c.K,
         c.M, tag, WAVES, S, DQ, NT, kmajor, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0, maxrel);
  fflush(stdout);
}

template <int DQ, int NT>
static void run_all_ws(Ctx& c, int kmajor, const char* tag) {
  run<4, 8, DQ, NT>(c, kmajor, tag);
  run<8, 4, DQ, NT>(c, kmajor, tag);
  run<16, 2, DQ, NT>(c, kmajor, tag);
  run<8, 14, DQ, NT>(c, kmajor, tag);
  run<16, 7, DQ, NT>(c, kmajor, tag);
  run<8, 8, DQ, NT>(c, kmajor, tag);   // K = 8192
  run<16, 4, DQ, NT>(c, kmajor, tag);  // K = 8192
}

int main(int argc, char** argv) {
  int M = argc > 1 ? atoi(argv[1]) : 1;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
  Shape shapes[] = {{4096, 4096}, {14336, 4096}, {4096, 14336}};
  for (auto sh : shapes) {
    Ctx c;
    c.N = sh.N;
    c.K = sh.K;
    c.M = M;
    const size_t wbytes = (size_t)c.N * c.K / 2;
    const size_t rot_mb = getenv("UB_ROT_MB") ? (size_t)atoi(getenv("UB_ROT_MB")) : 600;
    c.R = (int)((rot_mb << 20) / wbytes);
    if (c.R < 6) c.R = 6;
    if (c.R > 400) c.R = 400;
    const int nit = c.K / 128;
    const size_t words = wbytes / 4, nsz = (size_t)(c.N / 16) * nit * 16;
    c.h_qw.resize(words);
    c.h_szp.resize(nsz);
    for (auto& v : c.h_qw) v = rnd();
    for (auto& v : c.h_szp) {
      const float s = (5.2f + 0.8f * (rnd() % 1000) / 1000.f) * 0.02f / 15.f;
      const uint16_t sb = f2bf(s);
      const int z = 5 + rnd() % 6;
      const uint16_t zb = f2bf(-(bf2f(sb) * (float)z));
      v = (u32)sb | ((u32)zb << 16);
    }
    c.h_x.resize((size_t)16 * c.K);
    for (auto& v : c.h_x) v = f2bf(((int)(rnd() % 2001) - 1000) / 500.f);
    for (int r = 0; r < c.R; ++r) {
      u32 *a, *b;
      CK(hipMalloc(&a, wbytes));
      CK(hipMalloc(&b, nsz * 4));
      CK(hipMemcpy(a, c.h_qw.data(), wbytes, hipMemcpyHostToDevice));  // same content in every copy (different addresses)
      CK(hipMemcpy(b, c.h_szp.data(), nsz * 4, hipMemcpyHostToDevice));
      c.qw.push_back(a);
      c.szp.push_back(b);
    }
    CK(hipMalloc(&c.x, c.h_x.size() * 2));
    CK(hipMemcpy(c.x, c.h_x.data(), c.h_x.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&c.out, (size_t)16 * c.N * 2));
    const double bytes = (double)wbytes;
    {
      auto l0 = [&](int) { hipLaunchKernelGGL(null_kernel, dim3(256), dim3(256), 0, 0, (u32*)c.out); };
      printf("N=%6d K=%6d null kernel %7.2f us/launch\n", c.N, c.K, time_us(l0, c.R, 4));
      for (int blocks : {1024, 2048, 4096, 8192}) {
        auto l1 = [&](int r) {
          hipLaunchKernelGGL(lin_kernel, dim3(blocks), dim3(256), 0, 0, (const u32x4*)c.qw[r], wbytes / 16, (u32*)c.out);
        };
        const float us = time_us(l1, c.R, 4);
        printf("N=%6d K=%6d lin blocks=%5d  %7.2f us  %7.1f GB/s  %5.1f%%\n", c.N, c.K, blocks, us, bytes / us / 1e3,
               bytes / us / 1e3 / 80.0);
      }
    }
    run_all_ws<0, 1>(c, 0, "strm");
    run_all_ws<5, 1>(c, 0, "indep");
    run_all_ws<6, 1>(c, 0, "valu");
    run_all_ws<6, 1>(c, 2, "valuL2");
    run_all_ws<7, 1>(c, 0, "mfmao");
    run_all_ws<7, 1>(c, 2, "mfmaoL2");
    for (int r = 0; r < c.R; ++r) {
      CK(hipFree(c.qw[r]));
      CK(hipFree(c.szp[r]));
    }
    CK(hipFree(c.x));
    CK(hipFree(c.out));
  }
  return 0;
}
