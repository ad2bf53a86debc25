// VALU / conversion instruction issue-rate probe (experiments only): cycles per wave-instruction per SIMD at the
// nominal 2.4 GHz, 4 waves per SIMD, independent chains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef uint32_t u32;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(u32* out, int iters, u32 seed) {
  u32 a[8];
  unsigned long long q[4];
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i + seed; f[i] = (float)(a[i] & 1023); }
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = ((unsigned long long)a[i] << 32) | a[i + 4];
  const u32 mask = 0x000F000Fu ^ (seed & 1), magic = 0x43004300u ^ (seed & 2);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) a[i] = (a[i] >> 4) ^ seed;                       // v_lshrrev_b32 + v_xor (2 instr)
      if (KIND == 1) a[i] = (a[i] & mask) | magic;                    // v_and_or_b32
      if (KIND == 2) { q[i & 3] = (q[i & 3] >> 4) ^ seed; }           // v_lshrrev_b64 + xor
      if (KIND == 3) { bf16x2 r = {(__bf16)f[i], (__bf16)f[(i + 1) & 7]}; a[i] ^= __builtin_bit_cast(u32, r); }  // cvt_pk + xor
      if (KIND == 4) f[i] = __builtin_fmaf(f[i], 1.0001f, 0.5f);      // v_fma_f32
      if (KIND == 5) a[i] = __builtin_amdgcn_perm(a[i], a[(i + 1) & 7], 0x05040100u ^ (seed & 3));  // v_perm_b32
      if (KIND == 6) a[i] = a[i] ^ seed;                               // v_xor only (baseline for the +xor kinds)
      if (KIND == 7) f[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[i]), __builtin_bit_cast(bf16x2, a[(i + 1) & 7]), f[i], false);
    }
  }
  u32 r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r ^= a[i] ^ __builtin_bit_cast(u32, f[i]) ^ (u32)q[i & 3] ^ (u32)(q[i & 3] >> 32);
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
int run(const char* name, int per_iter) {
  u32* o;
  CK(hipMalloc(&o, 1024 * 256 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 8192;
  hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(256), 0, 0, o, iters, 1u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k<KIND>, dim3(1024), dim3(256), 0, 0, o, iters, 3u);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  // 1024 blocks x 4 waves over 1024 SIMDs = 4 waves per SIMD; each wave issues iters * per_iter instructions
  const double cyc = (double)ms * 1e-3 * 2.4e9 / (4.0 * iters * per_iter);
  printf("%-34s %8.1f us  %.2f cycles per wave-instruction per SIMD (nominal 2.4 GHz; %d instr/iter assumed)\n", name, ms * 1e3, cyc, per_iter);
  CK(hipFree(o));
  return 0;
}
int main() {
  run<6>("v_xor_b32", 8);
  run<0>("v_lshrrev_b32 + v_xor", 16);
  run<1>("v_and_or_b32", 8);
  run<2>("v_lshrrev_b64 + xor(2x32)", 8 * 3);
  run<3>("v_cvt_pk_bf16_f32 + v_xor", 16);
  run<4>("v_fma_f32", 8);
  run<5>("v_perm_b32", 8);
  run<7>("v_dot2_f32_bf16", 8);
  return 0;
}
