// Register-resident tile-compute microbenchmark (experiments only; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I llm_awq_amd/csrc tools/ubench/tile_ubench.hip -o tools/ubench/tile_ubench
// Question: what does ONE 1-KiB cdna4 tile (16 rows x 128 k: dequant on the matrix core + 4 product MFMAs) cost a SIMD
// when nothing waits for memory, by dequant variant and by the number of waves sharing the SIMD?  The decode GEMV is
// issue-bound (profiles/r01_gemv_ubench_m1.txt: 8.3 us with the weights L2-resident against 3.7 us for the same loads
// without math), so cycles per tile per SIMD is the number that bounds it.
//   V0  product kernel's dequant (bf16 magic 0x4300, 3 shifts + 4 and_or per word, v_dot2c offset, 3-instruction B operand)
//   V1  f16-mantissa dequant (magic 0x6400: nibbles at bits 3:0 and 7:4 of a half need no shift -> 1 shift + 4 and_or per
//       word; rows 2,3 of a quad carry s/16), offset by one VOP3P dot2, B operand by two v_perm
//   V2  V1 + hand interleave hints (sched_group_barrier: 1 MFMA : 3 VALU)
//   V3  V0 without the product MFMAs      V4  V0 extraction only        V5  the 12 MFMAs only
//   V6  V1 without product MFMAs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#include "awq_device.hpp"
using namespace awq;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));

// f16-mantissa matrix-core dequant producing bf16 (or f16) operands
template <typename DT>
struct F16MantDequant {
  using vec8 = typename DT::vec8;
  u32 sel01, sel23;   // v_perm selectors placing s' at inner index lane % 4 of the diagonal B operand
  u32 kMagic, kMaskLo, kMaskHi, kDotC;
  __device__ __forceinline__ void init(int lane) {
    const int pos = lane & 3;
    // v_perm_b32(S0, S1, sel): byte i of the result = byte sel[i] of {S0 (4..7), S1 (0..3)}; 0x0C = 0x00
    sel01 = pos == 0 ? 0x0C0C0100u : (pos == 1 ? 0x01000C0Cu : 0x0C0C0C0Cu);
    sel23 = pos == 2 ? 0x0C0C0100u : (pos == 3 ? 0x01000C0Cu : 0x0C0C0C0Cu);
    kMagic = 0x64006400u;
    kMaskLo = 0x000F000Fu;
    kMaskHi = 0x00F000F0u;
    kDotC = 0x3C00E400u;  // {-1024, 1} as f16 pair (lo = -1024)
    asm volatile("" : "+v"(kMagic));
    asm volatile("" : "+s"(kMaskLo));
    asm volatile("" : "+s"(kMaskHi));
    asm volatile("" : "+v"(kDotC));
  }
  // szp = {s' (f16) | sz (f16) << 16}: s' = s for rows n % 4 < 2, s / 16 for the others
  __device__ __forceinline__ void tile(const u32x4& w, u32 szp, vec8 (&op)[4]) const {
    const u32 b01 = __builtin_amdgcn_perm(szp, szp, sel01), b23 = __builtin_amdgcn_perm(szp, szp, sel23);
    const float cv = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, szp), __builtin_bit_cast(f16x2, kDotC), 0.0f, false);
    const f32x4 c = {cv, cv, cv, cv};
    const u32 ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const u32 w8 = ws[a] >> 8;
      const u32x2 a0 = {(ws[a] & kMaskLo) | kMagic, (ws[a] & kMaskHi) | kMagic};
      const u32x2 a1 = {(w8 & kMaskLo) | kMagic, (w8 & kMaskHi) | kMagic};
      const u32x2 b = {b01, b23};
      const f32x4 d0 = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h16x4, a0), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
      const f32x4 d1 = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(h16x4, a1), __builtin_bit_cast(h16x4, b), c, 0, 0, 0);
      op[a] = DT::pack8(d0, d1);
    }
  }
};

template <int V>
__global__ __launch_bounds__(256) void tile_kernel(const u32* __restrict__ src, u32* __restrict__ out, long long* __restrict__ cyc,
                                                    int iters) {
  __shared__ __attribute__((aligned(16))) char xs[4][1024];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int i = lane & 15, g = lane >> 4;
  u32x4 w = *reinterpret_cast<const u32x4*>(src + (size_t)(blockIdx.x * 256 + threadIdx.x) * 4);
  u32 sz = src[1 << 20 | (blockIdx.x * 256 + threadIdx.x)];
  *reinterpret_cast<u32x4*>(xs[wv] + lane * 16) = w;
  __syncthreads();
  Cdna4DequantT<BF16> cd;
  cd.init(lane);
  F16MantDequant<BF16> fd;
  fd.init(lane);
  using vec8 = BF16::vec8;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  u32 fold = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    asm volatile("" : "+v"(w), "+v"(sz));  // opaque: the tile "changes" every iteration, nothing is hoisted
    vec8 xop[4];
    const u32x4* xrow = reinterpret_cast<const u32x4*>(xs[wv]);
#pragma unroll
    for (int a = 0; a < 4; ++a) xop[a] = __builtin_bit_cast(vec8, xrow[(4 * a + g) ^ (i & 3)]);
    vec8 op[4];
    if (V == 0 || V == 3) {
      cd.tile_packed(w, sz, op);
    } else if (V == 1 || V == 2 || V == 6) {
      fd.tile(w, sz, op);
    } else if (V == 4) {
      const u32 ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        u32x4 e = {(ws[a] & cd.kMask) | cd.kMagic, ((ws[a] >> 4) & cd.kMask) | cd.kMagic, ((ws[a] >> 8) & cd.kMask) | cd.kMagic,
                   ((ws[a] >> 12) & cd.kMask) | cd.kMagic};
        op[a] = __builtin_bit_cast(vec8, e);
      }
    } else if (V == 5) {
      const u32x2 a0 = {w.x, w.y}, b = {w.z, w.w};
      const f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const f32x4 d0 = BF16::mfma4(a0, b, c), d1 = BF16::mfma4(a0, b, c);
        fold ^= __builtin_bit_cast(u32, d0[0]) ^ __builtin_bit_cast(u32, d1[1]);
        op[a] = __builtin_bit_cast(vec8, w);
      }
    }
    if (V == 3 || V == 4 || V == 6) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const u32x4 e = __builtin_bit_cast(u32x4, op[a]);
        fold ^= e.x ^ e.y ^ e.z ^ e.w;
      }
      fold ^= __builtin_bit_cast(u32x4, xop[it & 3]).x;
    } else {
#pragma unroll
      for (int a = 0; a < 4; ++a) acc = BF16::mfma(op[a], xop[a], acc);
    }
    if (V == 2) {
      // interleave hint: groups of 1 MFMA : 3 VALU
#pragma unroll
      for (int r = 0; r < 12; ++r) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);  // VALU
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = fold ^ __builtin_bit_cast(u32, acc[0] + acc[1] + acc[2] + acc[3]);
  if (lane == 0) cyc[blockIdx.x * 4 + wv] = t1 - t0;
}

template <int V>
void run(const char* name, const u32* src, u32* out, long long* cyc, int iters) {
  const int occs[] = {1, 2, 3, 4, 6, 8};
  for (int occ : occs) {
    const int blocks = 256 * occ;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(tile_kernel<V>, dim3(blocks), dim3(256), 0, 0, src, out, cyc, iters);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(tile_kernel<V>, dim3(blocks), dim3(256), 0, 0, src, out, cyc, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks * 4);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    // wall: every SIMD ran occ waves x iters tiles
    const double ns_tile_simd = (double)ms * 1e6 / ((double)occ * iters);
    printf("%-34s waves/SIMD=%d  wall %8.1f us  %7.1f ns per tile per SIMD   wave-clock: %7.1f ticks per tile per wave (%7.1f per SIMD-tile)\n",
           name, occ, ms * 1e3, ns_tile_simd, med / iters, med / iters / occ);
  }
}

int main() {
  u32 *src, *out;
  long long* cyc;
  const size_t n = (size_t)3 << 20;
  CK(hipMalloc(&src, n * 4));
  CK(hipMalloc(&out, (size_t)2048 * 256 * 4));
  CK(hipMalloc(&cyc, (size_t)2048 * 4 * 8));
  std::vector<u32> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (u32)(i * 2654435761u) ^ 0x3c003c00u;
  CK(hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice));
  const int iters = 4096;
  run<0>("V0 product dequant (bf16 magic)", src, out, cyc, iters);
  run<1>("V1 f16-mantissa dequant", src, out, cyc, iters);
  run<2>("V2 V1 + 1 MFMA : 3 VALU hints", src, out, cyc, iters);
  run<3>("V3 V0 without product MFMAs", src, out, cyc, iters);
  run<6>("V6 V1 without product MFMAs", src, out, cyc, iters);
  run<4>("V4 extraction only", src, out, cyc, iters);
  run<5>("V5 the 12 MFMAs only", src, out, cyc, iters);
  return 0;
}
