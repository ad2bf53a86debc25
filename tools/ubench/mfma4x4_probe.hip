// Probe: operand / result lane layout and issue rate of v_mfma_f32_4x4x4_16B_bf16 against 16x16x16 (gfx950).
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/mfma4x4_probe.hip -o tools/ubench/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
static __host__ __device__ uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }

__global__ void layout_kernel(float* out) {
  const int l = threadIdx.x, b = l >> 2, j = l & 3;
  s16x4 a, bb;
  for (int kk = 0; kk < 4; ++kk) {
    a[kk] = (short)f2bf((float)(1 + l * 4 + kk));       // lane (b, i = l % 4) holds A[b][i][kk]  (assumed)
    bb[kk] = (short)f2bf(kk == j ? 1.0f : 0.0f);         // lane (b, j) holds B[b][kk][j]          (assumed)
  }
  f32x4 c = {0, 0, 0, 0};
  f32x4 d = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, bb, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
  (void)b;
}
template <int MODE>
__global__ void rate_kernel(float* out, int iters) {
  s16x4 a = {(short)threadIdx.x, 1, 2, 3}, b = {3, 2, 1, (short)threadIdx.x};
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      c0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c3, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c3, 0, 0, 0);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  float* d;
  hipMalloc(&d, 1 << 22);
  layout_kernel<<<1, 64>>>(d);
  float h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int b = l >> 2, j = l & 3;
      const float want = (float)(1 + (4 * b + r) * 4 + j);  // D[b][i = r][j] = A[b][r][kk = j]
      if (h[l * 4 + r] != want) {
        if (bad < 8) printf("lane %d r %d got %g want %g\n", l, r, h[l * 4 + r], want);
        ++bad;
      }
    }
  printf("layout check: %s (%d mismatches)\n", bad ? "MISMATCH" : "as assumed", bad);
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 4, threads = 256;  // 4 waves per SIMD
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) rate_kernel<0><<<blocks, threads>>>(d, iters);
      else rate_kernel<1><<<blocks, threads>>>(d, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)iters * 4 * (blocks * (threads / 64)) / (256.0 * 4);
    printf("%s: %.3f ms, %.2f ns per MFMA per SIMD\n", mode == 0 ? "4x4x4_16B bf16_1k" : "16x16x16 bf16_1k", ms, ms * 1e6 / inst_per_simd);
  }
  return 0;
}
