// Prefill GEMM experiment driver over the C ABI (no torch: fast turnaround on the GPU box).
//   hipcc -O2 -std=c++17 tools/ubench/gemm_ubench.cpp -I include -L llm_awq_amd/lib -lawq_cdna4 -Wl,-rpath,'$ORIGIN/../../llm_awq_amd/lib' -o tools/ubench/gemm_ubench
// usage: gemm_ubench [variants...]   (gemm_variant knob values; the 128x128 kernel (1) is the correctness reference)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "awq_cdna4.h"

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)
#define AQ(x)                                                                  \
  do {                                                                         \
    int s_ = (x);                                                              \
    if (s_ != 0) {                                                             \
      printf("awq error %d (%s) at line %d: %s\n", s_, awq_status_string(s_), __LINE__, awq_last_hip_error()); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

static uint32_t rs = 777;
static inline uint32_t rnd() {
  rs ^= rs << 13;
  rs ^= rs >> 17;
  rs ^= rs << 5;
  return rs;
}
static inline uint16_t f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  std::vector<int> variants;
  for (int i = 1; i < argc; ++i) variants.push_back(atoi(argv[i]));
  if (variants.empty()) variants = {1, 2};
  struct Shape { int M, K, N; };
  Shape shapes[] = {{2048, 4096, 4096}, {2048, 4096, 6144}, {2048, 4096, 28672}, {2048, 14336, 4096},
                    {4096, 4096, 4096}, {4096, 4096, 28672}, {4096, 14336, 4096}, {512, 4096, 14336}};
  for (auto sh : shapes) {
    const int M = sh.M, K = sh.K, N = sh.N, G = K / 128;
    std::vector<uint8_t> hq((size_t)N * K);
    for (auto& v : hq) v = rnd() & 15;
    std::vector<uint16_t> hs((size_t)G * N), hz((size_t)G * N), hx((size_t)M * K);
    for (size_t i = 0; i < hs.size(); ++i) {
      const float s = (5.2f + 0.8f * (rnd() % 1000) / 1000.f) * 0.02f / 15.f;
      hs[i] = f2bf(s);
      hz[i] = f2bf(-(bf2f(hs[i]) * (float)(5 + rnd() % 6)));
    }
    for (auto& v : hx) v = f2bf(((int)(rnd() % 2001) - 1000) / 500.f);
    uint8_t* dq;
    void *qw2, *qw4, *ds, *dz, *dszp, *dx, *dout, *dref;
    CK(hipMalloc(&dq, hq.size()));
    CK(hipMalloc(&qw2, (size_t)N * K / 2));
    CK(hipMalloc(&qw4, (size_t)N * K / 2));
    CK(hipMalloc(&ds, hs.size() * 2));
    CK(hipMalloc(&dz, hz.size() * 2));
    CK(hipMalloc(&dszp, (size_t)N * G * 4));
    CK(hipMalloc(&dx, hx.size() * 2));
    CK(hipMalloc(&dout, (size_t)M * N * 2));
    CK(hipMalloc(&dref, (size_t)M * N * 2));
    CK(hipMemcpy(dq, hq.data(), hq.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dz, hz.data(), hz.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    AQ(awq_pack_v2(dq, qw2, N, K, nullptr));
    AQ(awq_repack_v2_to_cdna4(qw2, qw4, N, K, nullptr));
    AQ(awq_pack_sz_cdna4(ds, dz, dszp, N, K, nullptr));
    AQ(awq_tune_set("gemm_variant", 1));
    AQ(awq_w4a16_gemm_cdna4(dx, qw4, ds, dz, dszp, dref, M, N, K, 128, AWQ_BF16, nullptr, 0, nullptr));
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> href((size_t)M * N), hout((size_t)M * N);
    CK(hipMemcpy(href.data(), dref, href.size() * 2, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int v : variants) {
      (void)awq_tune_set("gemm_v6", v / 1000000);  // N000xxx = awq_gemm_v6.hip for the 256-wide tiles (2: every tile)
      (void)awq_tune_set("gemm_v5", (v / 100000) % 10);  // probe builds only: N00xxx = awq_gemm_v5.hip variant N for the 256-wide tiles
      (void)awq_tune_set("gemm_v4_probe", (v / 1000) % 100);  // probe builds only (AWQ_PROBES=1)  // Nxxx: timing-only probes of the v4 kernel (results are wrong by design)
      AQ(awq_tune_set("gemm_v4", (v % 1000) >= 100));  // 1xx: wide tiles run the hand-scheduled K loop (awq_gemm_v4.hip)  // 1xx: wide tiles run the hand-scheduled K loop (awq_gemm_v4.hip)
      AQ(awq_tune_set("gemm_variant", v % 100));
      CK(hipMemset(dout, 0xFF, (size_t)M * N * 2));
      AQ(awq_w4a16_gemm_cdna4(dx, qw4, ds, dz, dszp, dout, M, N, K, 128, AWQ_BF16, nullptr, 0, nullptr));
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(hout.data(), dout, hout.size() * 2, hipMemcpyDeviceToHost));
      size_t mism = 0;
      double maxd = 0;
      for (size_t i = 0; i < hout.size(); ++i)
        if (hout[i] != href[i]) {
          ++mism;
          const double d = fabs((double)bf2f(hout[i]) - (double)bf2f(href[i])) / (fabs((double)bf2f(href[i])) + 1e-3);
          if (d > maxd || d != d) maxd = d != d ? 1e9 : d;
        }
      for (int it = 0; it < 3; ++it)
        AQ(awq_w4a16_gemm_cdna4(dx, qw4, ds, dz, dszp, dout, M, N, K, 128, AWQ_BF16, nullptr, 0, nullptr));
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        const int iters = 5;
        for (int it = 0; it < iters; ++it)
          AQ(awq_w4a16_gemm_cdna4(dx, qw4, ds, dz, dszp, dout, M, N, K, 128, AWQ_BF16, nullptr, 0, nullptr));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms * 1e3f / iters);
      }
      const double tf = 2.0 * M * N * K / best / 1e6;
      printf("M=%5d K=%6d N=%6d variant=%d  %9.1f us  %7.1f TFLOP/s  %5.1f%%   mismatch vs 128x128: %.4f%% (max rel %.2e)\n", M, K,
             N, v, best, tf, tf / 25.0, 100.0 * mism / hout.size(), maxd);
      fflush(stdout);
    }
    AQ(awq_tune_set("gemm_variant", 0));
    AQ(awq_tune_set("gemm_v4", 1));
    (void)awq_tune_set("gemm_v4_probe", 0);
    (void)awq_tune_set("gemm_v5", 0);
    (void)awq_tune_set("gemm_v6", 0);
    hipFree(dq); hipFree(qw2); hipFree(qw4); hipFree(ds); hipFree(dz); hipFree(dszp); hipFree(dx); hipFree(dout); hipFree(dref);
  }
  return 0;
}
