// Instruction issue-rate probe with pinned instruction streams (inline asm; experiments only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/issue_ubench.hip -o tools/ubench/issue_ubench
// Every kernel runs `iters` iterations of a block of independent instructions on 16 private registers, with 1..8 waves per
// SIMD; reports shader-clock ticks (s_memtime) per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int KIND>
__global__ __launch_bounds__(256) void k(u32* out, long long* cyc, int iters, u32 seed) {
  u32 a[8], b[8];
  f32x4 d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = threadIdx.x * 2654435761u + i + seed;
    b[i] = a[i] ^ 0x3c003c00u;
    d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  u32 smask = 0x000F000Fu ^ (seed & 1);
  u32 vmagic = 0x43004300u ^ (seed & 2);
  asm volatile("" : "+s"(smask));
  asm volatile("" : "+v"(vmagic));
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#define X_XOR(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(vmagic));
#define X_SHR(i) asm volatile("v_lshrrev_b32 %0, 4, %1" : "=v"(b[i]) : "v"(a[i]));
#define X_ANDOR_S(i) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(b[i]) : "v"(a[i]), "s"(smask), "v"(vmagic));
#define X_ANDOR_V(i) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(b[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"(vmagic));
#define X_AND(i) asm volatile("v_and_b32 %0, %1, %2" : "=v"(b[i]) : "s"(smask), "v"(a[i]));
#define X_CVT(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(b[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
#define X_PERM(i) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(b[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]), "v"(vmagic));
#define X_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(b[i]) : "v"(a[i]));
#define X_FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(b[i]) : "v"(a[i]), "v"(vmagic));
#define X_PKFMA16(i) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(b[i]) : "v"(a[i]), "v"(vmagic));
#define X_PKMUL16(i) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(b[i]) : "v"(a[i]), "v"(vmagic));
#define X_DOT2(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, 0" : "=v"(b[i]) : "v"(a[i]), "v"(vmagic));
#define X_MFMA4(i) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, 0" : "=v"(d[i]) : "v"(*(u32x2*)&a[i & 6]), "v"(*(u32x2*)&b[i & 6]));
#define X_MFMA16(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d[i]) : "v"(*(u32x4*)&a[i & 4]), "v"(*(u32x4*)&b[i & 4]));
    if (KIND == 0) { REP8(X_XOR) }
    if (KIND == 1) { REP8(X_SHR) }
    if (KIND == 2) { REP8(X_ANDOR_S) }
    if (KIND == 3) { REP8(X_ANDOR_V) }
    if (KIND == 4) { REP8(X_AND) }
    if (KIND == 5) { REP8(X_CVT) }
    if (KIND == 6) { REP8(X_PERM) }
    if (KIND == 7) { REP8(X_MOV) }
    if (KIND == 8) { REP8(X_FMA) }
    if (KIND == 9) { REP8(X_PKFMA16) }
    if (KIND == 10) { REP8(X_PKMUL16) }
    if (KIND == 11) { REP8(X_DOT2) }
    if (KIND == 12) { REP8(X_MFMA4) }
    if (KIND == 13) { REP8(X_MFMA16) }
    if (KIND == 14) {  // 1 MFMA 4x4x4 : 4 VALU (and_or), independent
      X_MFMA4(0) X_ANDOR_S(0) X_ANDOR_S(1) X_CVT(2) X_CVT(3) X_MFMA4(1) X_ANDOR_S(4) X_ANDOR_S(5) X_CVT(6) X_CVT(7)
    }
    if (KIND == 15) {  // 1 MFMA 16x16x32 : 4 VALU
      X_MFMA16(0) X_ANDOR_S(0) X_ANDOR_S(1) X_CVT(2) X_CVT(3) X_MFMA16(1) X_ANDOR_S(4) X_ANDOR_S(5) X_CVT(6) X_CVT(7)
    }
    if (KIND == 16) {  // 1 MFMA 16x16x32 : 8 VALU
      X_MFMA16(0) X_ANDOR_S(0) X_ANDOR_S(1) X_CVT(2) X_CVT(3) X_ANDOR_S(4) X_ANDOR_S(5) X_CVT(6) X_CVT(7)
    }
    if (KIND == 17) {  // 1 MFMA 4x4x4 : 2 VALU
      X_MFMA4(0) X_ANDOR_S(0) X_CVT(1) X_MFMA4(1) X_ANDOR_S(2) X_CVT(3) X_MFMA4(2) X_ANDOR_S(4) X_CVT(5) X_MFMA4(3) X_ANDOR_S(6) X_CVT(7)
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  u32 r = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r ^= a[i] ^ b[i] ^ __builtin_bit_cast(u32, d[i][0] + d[i][1] + d[i][2] + d[i][3]);
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int n_instr, u32* out, long long* cyc) {
  const int iters = 2048;
  printf("%-44s", name);
  for (int occ : {1, 2, 4, 8}) {
    const int blocks = 256 * occ;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1u);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 3u);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(blocks * 4);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    printf("  w%d: %6.2f tk/instr/SIMD (%5.2f ns/blk)", occ, med / ((double)iters * n_instr * occ), ms * 1e6 / ((double)iters * occ));
  }
  printf("\n");
}

int main() {
  u32* out;
  long long* cyc;
  CK(hipMalloc(&out, (size_t)2048 * 256 * 4));
  CK(hipMalloc(&cyc, (size_t)2048 * 4 * 8));
  run<0>("v_xor_b32 (VOP2)", 8, out, cyc);
  run<1>("v_lshrrev_b32", 8, out, cyc);
  run<2>("v_and_or_b32 v, s, v", 8, out, cyc);
  run<3>("v_and_or_b32 v, v, v", 8, out, cyc);
  run<4>("v_and_b32 s, v", 8, out, cyc);
  run<5>("v_cvt_pk_bf16_f32", 8, out, cyc);
  run<6>("v_perm_b32", 8, out, cyc);
  run<7>("v_mov_b32", 8, out, cyc);
  run<8>("v_fma_f32", 8, out, cyc);
  run<9>("v_pk_fma_f16", 8, out, cyc);
  run<10>("v_pk_mul_f16", 8, out, cyc);
  run<11>("v_dot2_f32_f16", 8, out, cyc);
  run<12>("v_mfma_f32_4x4x4_16b_f16 (indep)", 8, out, cyc);
  run<13>("v_mfma_f32_16x16x32_bf16 (8 accumulators)", 8, out, cyc);
  run<14>("mix 2 mfma4 + 8 valu (per 10 instr)", 10, out, cyc);
  run<15>("mix 2 mfma16 + 8 valu (per 10 instr)", 10, out, cyc);
  run<16>("mix 1 mfma16 + 8 valu (per 9 instr)", 9, out, cyc);
  run<17>("mix 4 mfma4 + 8 valu (per 12 instr)", 12, out, cyc);
  return 0;
}
