// Prefill GEMM v3 on cdna4-interleaved weights (bf16, gfx950): 256 x 256 x 64 tile, 8 waves (2 along M x 4 along N,
// 128 x 64 each), v_mfma_f32_32x32x16_bf16, double-buffered LDS (2 x 64 KiB).
//
// Replaces gemm_w4a16_T1 / gemm_w4a16_T2 (reference awq/kernels/csrc/quantization_new/gemm/gemm_cuda.cu:312-1124) for
// the layout the rewritten repacker emits.  What changed against the 256x256 kernel in awq_gemm.hip (DESIGN.md "gemm"):
//   * the block barrier sits BEFORE the last k-step of a K-tile, not after it: the fragments of the next tile's first
//     k-step are read while the last 8 MFMAs of the current tile run, so no wave ever waits for LDS with an idle
//     matrix pipe (fragments are double-buffered in registers, one k-step ahead everywhere);
//   * the weight tile of the NEXT K-tile is produced one 32-bit word (8 weights per lane, two dequant MFMAs, four
//     v_cvt_pk, one ds_write_b128) per k-step instead of in one block, and its packed words are fetched one
//     quantisation group (two K-tiles) ahead with one 16-byte load per slab;
//   * 32x32x16 MFMAs: half the matrix instructions per flop and the higher measured ceiling of the two shapes.
// Numerics are those of every other kernel here: W = round_bf16(q*s + sz) exactly (matrix-core dequant), fp32
// accumulation, one rounding of the result.
#include <type_traits>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

namespace {
constexpr int TM = 256, TN = 256, TK = 64;
constexpr int kTile = TM * TK * 2;           // 32 KiB: one [256][64] bf16 tile
constexpr int kWBase = 2 * kTile;            // LDS: x stage 0 | x stage 1 | w stage 0 | w stage 1 (every ds offset stays < 64 KiB from its base)
constexpr int kEpiRow = 144;                 // bytes per staged output row (64 n x 2 B + 16 pad)
constexpr int kSmemV3 = 8 * 128 * kEpiRow;   // 147456 >= 4 * kTile
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int tile_off(int row, int gc) { return row * 128 + ((gc ^ ((row >> 1) & 7)) << 4); }

struct Group {       // one quantisation group (128 k) of the wave's two slabs
  u32x4 w0, w1;      // slab 0 / slab 1: the lane's 16 bytes of the 1-KiB tile
  u32 b01_0, b23_0, b01_1, b23_1;  // diagonal scale operands of the dequant MFMA
  float c0, c1;      // sz - 128 s
};
}  // namespace

template <int BARMID>
__global__ __launch_bounds__(512) void gemm_cdna4_v3_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                            const u32* __restrict__ szp,
                                                            uint16_t* __restrict__ out, int M, int N, int K, int tiles_m,
                                                            int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int l32 = lane & 31, hk = lane >> 5;
  const int wm = wv >> 2, wn = wv & 3;

  // XCD-aware tile order (bijective for any tile count): every XCD walks a contiguous range of tiles
  const int T = tiles_m * tiles_n;
  int tile;
  {
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int q = T >> 3, r = T & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  // the last row tile is shifted up to end at row M - 1 (M >= 256): it recomputes a few rows of its neighbour with
  // identical results, and no row index ever needs clamping, so the four x granule addresses differ by constants
  const int m0 = min(tm * TM, M - TM), n0 = tn * TN;
  const int nit = K >> 7;

  // ---- x tile: LDS-DMA, 4 x 16 B per thread per K-tile; swizzle applied to the SOURCE granule ----
  u32 a_off0;  // element offset into x of granule q = 0 (M * K < 2^31); granule q is 64 rows further down
  {
    const int row = tid >> 3, gcp = tid & 7;  // rows q * 64 + row share (row >> 1) & 7
    const int gc = gcp ^ ((row >> 1) & 7);
    a_off0 = (u32)(m0 + row) * (u32)K + gc * 8;
  }
  auto issue_a = [&](int kt, int stage) {
    char* dst = smem + stage * kTile + wv * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint16_t* xq = x + (size_t)kt * TK + (size_t)q * 64 * K;  // wave-uniform part (SGPRs)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xq + a_off0),
                                       (__attribute__((address_space(3))) void*)(dst + q * 8192), 16, 0, 0);
    }
  };

  // ---- weight tile: wave wv owns slabs 2*wv, 2*wv+1 (rows 32*wv .. +31 of the 256-row tile) ----
  const int nslab = N >> 4;
  const int sl0 = min((n0 >> 4) + 2 * wv, nslab - 1), sl1 = min((n0 >> 4) + 2 * wv + 1, nslab - 1);
  // word offsets (N * K / 8 < 2^31 words for every supported shape): tile (slab, grp) at (slab * nit + grp) * 256 + 4 * lane
  const u32 b_off0 = (u32)sl0 * nit * 256 + lane * 4, b_off1 = (u32)sl1 * nit * 256 + lane * 4;
  const u32 sz_off0 = (u32)sl0 * nit * 16 + i, sz_off1 = (u32)sl1 * nit * 16 + i;  // packed {scale | scaled_zero << 16}
  const int nl = 32 * wv + i;  // tile row of slab 0's lane row; slab 1 = + 16
  Cdna4Dequant cd;
  cd.init(lane);

  struct Raw {
    u32x4 w0, w1;
    u32 sz0, sz1;
  };
  auto load_group = [&](int grp) {
    Raw r;
    const u32* qg = qw + (size_t)grp * 256;
    const u32* sg = szp + (size_t)grp * 16;
    r.w0 = *reinterpret_cast<const u32x4*>(qg + b_off0);
    r.w1 = *reinterpret_cast<const u32x4*>(qg + b_off1);
    r.sz0 = sg[sz_off0];
    r.sz1 = sg[sz_off1];
    return r;
  };
  auto prep = [&](const Raw& r) {
    Group gq;
    gq.w0 = r.w0;
    gq.w1 = r.w1;
    const u32 sd0 = (r.sz0 & 0xFFFFu) * 0x00010001u, sd1 = (r.sz1 & 0xFFFFu) * 0x00010001u;
    gq.b01_0 = sd0 & cd.m01;
    gq.b23_0 = sd0 & cd.m23;
    gq.b01_1 = sd1 & cd.m01;
    gq.b23_1 = sd1 & cd.m23;
    gq.c0 = __builtin_fmaf(-128.0f, __builtin_bit_cast(float, r.sz0 << 16), __builtin_bit_cast(float, r.sz0 & 0xFFFF0000u));
    gq.c1 = __builtin_fmaf(-128.0f, __builtin_bit_cast(float, r.sz1 << 16), __builtin_bit_cast(float, r.sz1 & 0xFFFF0000u));
    return gq;
  };
  // job j of K-tile half h: word 2h + (j & 1) of slab (j >> 1) -> weight tile of `stage`, granule 4 (j & 1) + g
  auto job = [&](const Group& gq, int h, int j, int stage) {
    char* Bs = smem + kWBase + stage * kTile;
    const int widx = 2 * h + (j & 1);
    const u32x4& wsl = (j >> 1) ? gq.w1 : gq.w0;
    const u32 word = widx == 0 ? wsl.x : (widx == 1 ? wsl.y : (widx == 2 ? wsl.z : wsl.w));
    const bf16x8 v = (j >> 1) ? cd.word(word, gq.b01_1, gq.b23_1, gq.c1) : cd.word(word, gq.b01_0, gq.b23_0, gq.c0);
    *reinterpret_cast<bf16x8*>(Bs + tile_off(nl + 16 * (j >> 1), 4 * (j & 1) + g)) = v;
  };
  // fragments (single set, 24 VGPRs): each one is re-read for the NEXT k-step right after the last MFMA of this
  // k-step that consumes it -- wf[0] after the a = 0 sweep, xf[b] after its a = 1 MFMA, wf[1] at the end -- so every
  // ds_read has at least three MFMAs (plus the other wave of the SIMD) to land
  bf16x8 wf[2], xf[4];
  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  auto w_addr = [&](int stage, int ks, int t) {
    return smem + kWBase + stage * kTile + tile_off(wn * 64 + t * 32 + l32, 2 * ks + hk);
  };
  auto x_addr = [&](int stage, int ks, int t) { return smem + stage * kTile + tile_off(wm * 128 + t * 32 + l32, 2 * ks + hk); };
  // one k-step: 8 MFMAs while the fragments of k-step (stage_n, ks_n) stream in
  // `bar`: the block barrier sits AFTER the a = 0 sweep of the last k-step of a tile: the fragment reads issued at the
  // end of the previous k-step have had four MFMAs to land, so the lgkmcnt(0) in front of the barrier is (nearly) free,
  // and every read of the next tile's stage comes after it
  auto step = [&](int stage_n, int ks_n, bool rd, bool bar = false) {
    if (bar && !BARMID) {  // experiment arm: barrier in front of the whole k-step
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __syncthreads();
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], xf[b], acc[0][b], 0, 0, 0);
    if (bar && BARMID) {
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
      __syncthreads();
    }
    if (rd) wf[0] = *reinterpret_cast<const bf16x8*>(w_addr(stage_n, ks_n, 0));
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], xf[b], acc[1][b], 0, 0, 0);
      if (rd) xf[b] = *reinterpret_cast<const bf16x8*>(x_addr(stage_n, ks_n, b));
    }
    if (rd) wf[1] = *reinterpret_cast<const bf16x8*>(w_addr(stage_n, ks_n, 1));
  };

  // ---------------- prologue: tile 0 complete in stage 0, tile 1's x tile in flight, its first weight word written ----
  issue_a(0, 0);
  Group gc = prep(load_group(0));  // the group whose words are being written (one expanded group live at a time)
  Raw rn = load_group(nit > 1 ? 1 : 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) job(gc, 0, j, 0);
  __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and the ds_writes
#pragma unroll
  for (int t = 0; t < 2; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(w_addr(0, 0, t));
#pragma unroll
  for (int t = 0; t < 4; ++t) xf[t] = *reinterpret_cast<const bf16x8*>(x_addr(0, 0, t));
  // pin the prefetched group into registers BEFORE the next LDS-DMA goes out: otherwise the loop header inherits a
  // pending ordinary load from this path and hipcc drains the DMA queue (vmcnt(0)) at the top of every iteration
  asm volatile("" : "+v"(rn.w0), "+v"(rn.w1), "+v"(rn.sz0), "+v"(rn.sz1));
  issue_a(1, 1);  // K is a multiple of 128: there are always at least two K-tiles
  job(gc, 1, 0, 1);

  // One iteration = one quantisation group = two K-tiles (2q in stage 0, 2q + 1 in stage 1).  `more` is a compile-time
  // flag (the last group is peeled) so that every iteration is ONE basic block the scheduler can interleave freely.
  auto group_iter = [&](int q, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value;
    // ---------- K-tile 2q (stage 0); writes words 1..3 of tile 2q+1 = (group q, half 1) into stage 1 ----------
    step(0, 1, true);
    job(gc, 1, 1, 1);
    step(0, 2, true);
    job(gc, 1, 2, 1);
    step(0, 3, true);
    job(gc, 1, 3, 1);
    // last k-step of tile 2q; barrier inside: tile 2q+1 complete in stage 1, every read of stage 0 retired, loads drained
    step(1, 0, true, true);
    if (more) {
      gc = prep(rn);             // group q+1: loaded one iteration ago, drained by the barrier above
      rn = load_group(min(q + 2, nit - 1));  // consumed after the NEXT iteration's first barrier
    }
    if (more) {
      issue_a(2 * q + 2, 0);
      job(gc, 0, 0, 0);          // first word of tile 2q+2 = (group q+1, half 0)
    }
    // ---------- K-tile 2q+1 (stage 1); writes words 1..3 of tile 2q+2 into stage 0 ----------
    step(1, 1, true);
    if (more) job(gc, 0, 1, 0);
    step(1, 2, true);
    if (more) job(gc, 0, 2, 0);
    step(1, 3, true);
    if (more) job(gc, 0, 3, 0);
    step(0, 0, more, true);      // barrier inside: tile 2q+2 complete in stage 0; every read of stage 1 retired
    if (more) {
      issue_a(2 * q + 3, 1);
      job(gc, 1, 0, 1);          // first word of tile 2q+3 = (group q+1, half 1)
    }
  };
  for (int q = 0; q + 1 < nit; ++q) {
    group_iter(q, std::true_type{});
  }
  group_iter(nit - 1, std::false_type{});

  // ---------------- epilogue through LDS: acc[a][b][r] = C[n = wn*64 + a*32 + (r&3) + 8 (r>>2) + 4 hk][m = wm*128 + b*32 + l32] ----
  __syncthreads();  // stage memory is re-used as the output staging area
  char* eb = smem + wv * (128 * kEpiRow);
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x2 v;
        v.x = (u32)BF16::from_float(acc[a][b][4 * j + 0]) | ((u32)BF16::from_float(acc[a][b][4 * j + 1]) << 16);
        v.y = (u32)BF16::from_float(acc[a][b][4 * j + 2]) | ((u32)BF16::from_float(acc[a][b][4 * j + 3]) << 16);
        *reinterpret_cast<u32x2*>(eb + (b * 32 + l32) * kEpiRow + (a * 32 + 8 * j + 4 * hk) * 2) = v;
      }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's own LDS writes (region is wave-private)
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int ps = 0; ps < 16; ++ps) {
    const int row = ps * 8 + (lane >> 3), gc = lane & 7;
    const int m = m0 + wm * 128 + row, nn = n0 + wn * 64 + gc * 8;
    const u32x4 v = *reinterpret_cast<const u32x4*>(eb + row * kEpiRow + gc * 16);
    if (m < M && nn < N) *reinterpret_cast<u32x4*>(out + (size_t)m * N + nn) = v;
  }
}

int g_v3_barmid = 1;
int launch_gemm_cdna4_v3(const void* x, const void* qw, const void* szp, void* out, int m, int n, int k, hipStream_t st) {
  if (!szp || m < TM || (n % 16) != 0 || (k % 128) != 0 || (size_t)m * (size_t)k >= (1ull << 31)) return -1;
  const int tiles_m = (m + TM - 1) / TM, tiles_n = (n + TN - 1) / TN;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_cdna4_v3_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmemV3);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_cdna4_v3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, kSmemV3);
    attr = true;
  }
  if (g_v3_barmid)
    hipLaunchKernelGGL(gemm_cdna4_v3_kernel<1>, dim3(tiles_m * tiles_n), dim3(512), kSmemV3, st, (const uint16_t*)x,
                       (const u32*)qw, (const u32*)szp, (uint16_t*)out, m, n, k, tiles_m, tiles_n);
  else
    hipLaunchKernelGGL(gemm_cdna4_v3_kernel<0>, dim3(tiles_m * tiles_n), dim3(512), kSmemV3, st, (const uint16_t*)x,
                       (const u32*)qw, (const u32*)szp, (uint16_t*)out, m, n, k, tiles_m, tiles_n);
  return 0;
}

}  // namespace awq
