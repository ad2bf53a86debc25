// Prefill GEMM v3 on cdna4-interleaved weights (bf16 and fp16, gfx950): 256 x (128 * NSL) x 64 tile, 8 waves (2 along M x 4
// along N, 128 x (32 * NSL) each), v_mfma_f32_32x32x16_bf16, double-buffered LDS.
//
// Replaces gemm_w4a16_T1 / gemm_w4a16_T2 (reference awq/kernels/csrc/quantization_new/gemm/gemm_cuda.cu:312-1124) for
// the layout the rewritten repacker emits.  Against the 256x256 kernel in awq_gemm.hip (DESIGN.md "gemm"):
//   * the block barrier sits INSIDE the last k-step of a K-tile, so the fragments of the next tile's first k-step are
//     read while the remaining MFMAs of the current tile run and the lgkmcnt(0) in front of the barrier is nearly free;
//   * the weight tile of the NEXT K-tile is produced one 32-bit word (8 weights per lane, two dequant MFMAs, four
//     v_cvt_pk, one ds_write_b128) per k-step instead of in one block, and its packed words are fetched one
//     quantisation group (two K-tiles) ahead with one 16-byte load per slab;
//   * no ordinary load is ever consumed while an LDS-DMA is in flight (hipcc would drain the DMA queue there);
//   * 32x32x16 MFMAs: half the matrix instructions per flop and the higher measured ceiling of the two shapes;
//   * NSL = 1 (256 x 128 tiles) doubles the tile count for shapes that would fill only half the chip with 256 x 256.
// Numerics are those of every other kernel here: W = round_T(q*s + sz) exactly (matrix-core dequant), fp32
// accumulation in K order, one rounding of the result -- bit-identical to the 128x128 kernel.
#include <string.h>

#include <type_traits>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

namespace {
constexpr int TM = 256, TK = 64;
constexpr int kTileX = TM * TK * 2;  // 32 KiB: the x tile [256][64] bf16; LDS: x stage 0 | x stage 1 | w stage 0 | w stage 1
constexpr int kWBase = 2 * kTileX;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int tile_off(int row, int gc) { return row * 128 + ((gc ^ ((row >> 1) & 7)) << 4); }

template <int NSL>
struct Group {  // one quantisation group (128 k) of the wave's NSL slabs
  u32x4 w[NSL];   // the lane's 16 bytes of each slab's 1-KiB tile
  u32 b01[NSL], b23[NSL];  // diagonal scale operands of the dequant MFMA
  float c[NSL];   // sz - 128 s
};
template <int NSL>
struct Raw {
  u32x4 w[NSL];
  u32 sz[NSL];
};
}  // namespace

template <typename DT, int NSL>
__global__ __launch_bounds__(512) void gemm_cdna4_v3_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                            const u32* __restrict__ szp, const uint16_t* __restrict__ bias,
                                                            uint16_t* __restrict__ out, int M, int N, int K, int tiles_m,
                                                            int tiles_n, int n_begin, int n_end) {
  constexpr int TN = 128 * NSL;            // weight rows per block
  constexpr int WN = 32 * NSL;             // weight rows per wave
  constexpr int kTileW = TN * TK * 2;      // 16 / 32 KiB per weight stage
  constexpr int kEpiRow = 2 * WN + 16;     // bytes per staged output row (+16 pad)
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
  // within that order tiles are walked in bands of two row tiles, column tiles fastest: the ~32 tiles an XCD runs at
  // once then cover 2 x panels (2 MiB each at K = 4096) x 16 weight panels (0.5 MiB each) instead of 16 x 2 -- about
  // a third of the L2-miss traffic of a row-tile-fastest walk
  int tm, tn;
  {
    const int full = (tiles_m >> 1) * 2 * tiles_n;  // tiles inside complete two-row bands
    if (tile < full) {
      const int band = tile / (2 * tiles_n), rem = tile - band * 2 * tiles_n;
      tn = rem >> 1;
      tm = 2 * band + (rem & 1);
    } else {
      tn = tile - full;
      tm = tiles_m - 1;
    }
  }
  // the last row tile is shifted up to end at row M - 1 (M >= 256): it recomputes a few rows of its neighbour with
  // identical results, and no row index ever needs clamping, so the four x granule addresses differ by constants
  const int m0 = min(tm * TM, M - TM), n0 = n_begin + tn * TN;  // this launch covers weight rows [n_begin, n_end)
  const int nit = K >> 7;

  // ---- x tile: LDS-DMA, 4 x 16 B per thread per K-tile; swizzle applied to the SOURCE granule ----
  u32 a_off0;  // element offset into x of granule q = 0 (M * K < 2^31); granule q is 64 rows further down
  {
    const int row = tid >> 3, gcp = tid & 7;  // rows q * 64 + row share (row >> 1) & 7
    const int gc = gcp ^ ((row >> 1) & 7);
    a_off0 = (u32)(m0 + row) * (u32)K + gc * 8;
  }
  auto issue_a = [&](int kt, int stage) {
    char* dst = smem + stage * kTileX + wv * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint16_t* xq = x + (size_t)kt * TK + (size_t)q * 64 * K;  // wave-uniform part (SGPRs)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xq + a_off0),
                                       (__attribute__((address_space(3))) void*)(dst + q * 8192), 16, 0, 0);
    }
  };

  // ---- weight tile: wave wv owns slabs NSL*wv .. NSL*wv + NSL-1 (rows 16*NSL*wv .. of the TN-row tile) ----
  const int nslab = N >> 4;
  u32 b_off[NSL], sz_off[NSL];  // word offsets (N * K / 8 < 2^31): tile (slab, grp) at (slab * nit + grp) * 256 + 4 * lane
#pragma unroll
  for (int s = 0; s < NSL; ++s) {
    const int sl = min((n0 >> 4) + NSL * wv + s, min(nslab, n_end >> 4) - 1);
    b_off[s] = (u32)sl * nit * 256 + lane * 4;
    sz_off[s] = (u32)sl * nit * 16 + i;  // packed {scale | scaled_zero << 16}
  }
  const int nl = 16 * NSL * wv + i;  // tile row of slab 0's lane row; slab s = + 16 s
  using vec8 = typename DT::vec8;
  Cdna4DequantT<DT> cd;
  cd.init(lane);

  auto load_group = [&](int grp) {
    Raw<NSL> r;
    const u32* qg = qw + (size_t)grp * 256;
    const u32* sg = szp + (size_t)grp * 16;
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      r.w[s] = *reinterpret_cast<const u32x4*>(qg + b_off[s]);
      r.sz[s] = sg[sz_off[s]];
    }
    return r;
  };
  auto prep = [&](const Raw<NSL>& r) {
    Group<NSL> gq;
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      gq.w[s] = r.w[s];
      const u32 sd = (r.sz[s] & 0xFFFFu) * 0x00010001u;
      gq.b01[s] = sd & cd.m01;
      gq.b23[s] = sd & cd.m23;
      gq.c[s] = DT::dq_offset(r.sz[s]);
    }
    return gq;
  };
  // job j (0 .. 2 NSL - 1) of K-tile half h: word 2h + (j & 1) of slab (j >> 1) -> weight tile of `stage`, granule 4 (j & 1) + g
  auto job = [&](const Group<NSL>& gq, int h, int j, int stage) {
    char* Bs = smem + kWBase + stage * kTileW;
    const int widx = 2 * h + (j & 1), s = j >> 1;
    const u32 word = widx == 0 ? gq.w[s].x : (widx == 1 ? gq.w[s].y : (widx == 2 ? gq.w[s].z : gq.w[s].w));
    const vec8 v = cd.word(word, gq.b01[s], gq.b23[s], gq.c[s]);
    *reinterpret_cast<vec8*>(Bs + tile_off(nl + 16 * s, 4 * (j & 1) + g)) = v;
  };
  // the 2 NSL word jobs of a K-tile are spread over its four production slots (slot 0 = right after the barrier of
  // the previous tile, slots 1..3 = its own first three k-steps)
  auto slot = [&](const Group<NSL>& gq, int h, int sl, int stage) {
    if (NSL == 2) job(gq, h, sl, stage);
    else if ((sl & 1) == 0) job(gq, h, sl >> 1, stage);
  };

  // fragments (single set): each one is re-read for the NEXT k-step right after the last MFMA of this k-step that
  // consumes it, so every ds_read has several MFMAs (plus the other wave of the SIMD) to land
  vec8 wf[NSL], xf[4];
  f32x16 acc[NSL][4];
#pragma unroll
  for (int a = 0; a < NSL; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  auto w_addr = [&](int stage, int ks, int t) {
    return smem + kWBase + stage * kTileW + tile_off(wn * WN + t * 32 + l32, 2 * ks + hk);
  };
  auto x_addr = [&](int stage, int ks, int t) { return smem + stage * kTileX + tile_off(wm * 128 + t * 32 + l32, 2 * ks + hk); };
  // one k-step: 4 NSL MFMAs while the fragments of k-step (stage_n, ks_n) stream in.  `bar`: the block barrier sits
  // after the first MFMAs of the last k-step of a tile: the reads issued at the end of the previous k-step have had
  // time to land, so the lgkmcnt(0) in front of the barrier is (nearly) free, and every read of the next tile's stage
  // comes after it
  auto step = [&](int stage_n, int ks_n, bool rd, bool bar = false) {
    if (NSL == 2) {
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[0][b] = DT::mfma32(wf[0], xf[b], acc[0][b]);
      if (bar) {
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __syncthreads();
      }
      if (rd) wf[0] = *reinterpret_cast<const vec8*>(w_addr(stage_n, ks_n, 0));
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        acc[NSL - 1][b] = DT::mfma32(wf[NSL - 1], xf[b], acc[NSL - 1][b]);
        if (rd) xf[b] = *reinterpret_cast<const vec8*>(x_addr(stage_n, ks_n, b));
      }
      if (rd) wf[NSL - 1] = *reinterpret_cast<const vec8*>(w_addr(stage_n, ks_n, NSL - 1));
    } else {
      // one weight fragment: the first two MFMAs run in front of the barrier, x fragments are re-read behind it
      vec8 xn[2];
      acc[0][0] = DT::mfma32(wf[0], xf[0], acc[0][0]);
      acc[0][1] = DT::mfma32(wf[0], xf[1], acc[0][1]);
      if (bar) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __syncthreads();
      }
      if (rd) {
        xn[0] = *reinterpret_cast<const vec8*>(x_addr(stage_n, ks_n, 0));
        xn[1] = *reinterpret_cast<const vec8*>(x_addr(stage_n, ks_n, 1));
      }
      acc[0][2] = DT::mfma32(wf[0], xf[2], acc[0][2]);
      if (rd) xf[2] = *reinterpret_cast<const vec8*>(x_addr(stage_n, ks_n, 2));
      acc[0][3] = DT::mfma32(wf[0], xf[3], acc[0][3]);
      if (rd) {
        xf[3] = *reinterpret_cast<const vec8*>(x_addr(stage_n, ks_n, 3));
        wf[0] = *reinterpret_cast<const vec8*>(w_addr(stage_n, ks_n, 0));
        xf[0] = xn[0];
        xf[1] = xn[1];
      }
    }
  };

  // ---------------- prologue: tile 0 complete in stage 0, tile 1's x tile in flight, its first weight word written ----
  issue_a(0, 0);
  Group<NSL> gc = prep(load_group(0));  // the group whose words are being written (one expanded group live at a time)
  Raw<NSL> rn = load_group(nit > 1 ? 1 : 0);
#pragma unroll
  for (int j = 0; j < 2 * NSL; ++j) job(gc, 0, j, 0);
  __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and the ds_writes
#pragma unroll
  for (int t = 0; t < NSL; ++t) wf[t] = *reinterpret_cast<const vec8*>(w_addr(0, 0, t));
#pragma unroll
  for (int t = 0; t < 4; ++t) xf[t] = *reinterpret_cast<const vec8*>(x_addr(0, 0, t));
  // pin the prefetched group into registers BEFORE the next LDS-DMA goes out: otherwise the loop header inherits a
  // pending ordinary load from this path and hipcc drains the DMA queue (vmcnt(0)) at the top of every iteration
#pragma unroll
  for (int s = 0; s < NSL; ++s) asm volatile("" : "+v"(rn.w[s]), "+v"(rn.sz[s]));
  issue_a(1, 1);  // K is a multiple of 128: there are always at least two K-tiles
  slot(gc, 1, 0, 1);

  // One iteration = one quantisation group = two K-tiles (2q in stage 0, 2q + 1 in stage 1).  `more` is a compile-time
  // flag (the last group is peeled) so that every iteration is ONE basic block the scheduler can interleave freely.
  auto group_iter = [&](int q, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value;
    // ---------- K-tile 2q (stage 0); writes the remaining words of tile 2q+1 = (group q, half 1) into stage 1 ----------
    step(0, 1, true);
    slot(gc, 1, 1, 1);
    step(0, 2, true);
    slot(gc, 1, 2, 1);
    step(0, 3, true);
    slot(gc, 1, 3, 1);
    // last k-step of tile 2q; barrier inside: tile 2q+1 complete in stage 1, every read of stage 0 retired, loads drained
    step(1, 0, true, true);
    if (more) {
      gc = prep(rn);             // group q+1: loaded one iteration ago, drained by the barrier above
      rn = load_group(min(q + 2, nit - 1));  // consumed after the NEXT iteration's first barrier
      issue_a(2 * q + 2, 0);
      slot(gc, 0, 0, 0);         // first word of tile 2q+2 = (group q+1, half 0)
    }
    // ---------- K-tile 2q+1 (stage 1); writes the remaining words of tile 2q+2 into stage 0 ----------
    step(1, 1, true);
    if (more) slot(gc, 0, 1, 0);
    step(1, 2, true);
    if (more) slot(gc, 0, 2, 0);
    step(1, 3, true);
    if (more) slot(gc, 0, 3, 0);
    step(0, 0, more, true);      // barrier inside: tile 2q+2 complete in stage 0; every read of stage 1 retired
    if (more) {
      issue_a(2 * q + 3, 1);
      slot(gc, 1, 0, 1);         // first word of tile 2q+3 = (group q+1, half 1)
    }
  };
  for (int q = 0; q + 1 < nit; ++q) group_iter(q, std::true_type{});
  group_iter(nit - 1, std::false_type{});

  // ---------------- epilogue through LDS: acc[a][b][r] = C[n = wn*WN + a*32 + (r&3) + 8 (r>>2) + 4 hk][m = wm*128 + b*32 + l32] ----
  __syncthreads();  // stage memory is re-used as the output staging area
  char* eb = smem + wv * (128 * kEpiRow);
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int a = 0; a < NSL; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x2 v;
        v.x = (u32)DT::from_float(acc[a][b][4 * j + 0]) | ((u32)DT::from_float(acc[a][b][4 * j + 1]) << 16);
        v.y = (u32)DT::from_float(acc[a][b][4 * j + 2]) | ((u32)DT::from_float(acc[a][b][4 * j + 3]) << 16);
        *reinterpret_cast<u32x2*>(eb + (b * 32 + l32) * kEpiRow + (a * 32 + 8 * j + 4 * hk) * 2) = v;
      }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's own LDS writes (region is wave-private)
  __builtin_amdgcn_wave_barrier();
  constexpr int GR = WN / 8;        // 16-byte granules per staged row (4 or 8)
  constexpr int RP = 64 / GR;       // rows per pass
#pragma unroll
  for (int ps = 0; ps < 128 / RP; ++ps) {
    const int row = ps * RP + lane / GR, gc2 = lane % GR;
    const int m = m0 + wm * 128 + row, nn = n0 + wn * WN + gc2 * 8;
    u32x4 v = *reinterpret_cast<const u32x4*>(eb + row * kEpiRow + gc2 * 16);
    if (nn < n_end) {
      if (bias != nullptr) {  // `out + self.bias` in T (qmodule.py:221): the matmul result was already rounded to bf16
        const u32x4 bv = *reinterpret_cast<const u32x4*>(bias + nn);
        auto add2 = [](u32 a, u32 b) {
          const float lo = DT::to_float((uint16_t)(a & 0xFFFFu)) + DT::to_float((uint16_t)(b & 0xFFFFu));
          const float hi = DT::to_float((uint16_t)(a >> 16)) + DT::to_float((uint16_t)(b >> 16));
          return (u32)DT::from_float(lo) | ((u32)DT::from_float(hi) << 16);
        };
        v = u32x4{add2(v.x, bv.x), add2(v.y, bv.y), add2(v.z, bv.z), add2(v.w, bv.w)};
      }
      *reinterpret_cast<u32x4*>(out + (size_t)m * N + nn) = v;
    }
  }
}

namespace {
template <typename DT, int NSL>
void launch_v3(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int n_begin,
               int n_end, hipStream_t st) {
  constexpr int TN = 128 * NSL;
  constexpr int smem_main = 2 * kTileX + 2 * TN * TK * 2;
  constexpr int smem_epi = 8 * 128 * (64 * NSL + 16);
  constexpr int smem = smem_main > smem_epi ? smem_main : smem_epi;
  const int tiles_m = (m + TM - 1) / TM, tiles_n = (n_end - n_begin + TN - 1) / TN;
  static LdsOptIn optin;  // per (kernel instantiation, device)
  optin.ensure(reinterpret_cast<const void*>(gemm_cdna4_v3_kernel<DT, NSL>), smem);
  hipLaunchKernelGGL((gemm_cdna4_v3_kernel<DT, NSL>), dim3(tiles_m * tiles_n), dim3(512), smem, st, (const uint16_t*)x, (const u32*)qw,
                     (const u32*)szp, (const uint16_t*)bias, (uint16_t*)out, m, n, k, tiles_m, tiles_n, n_begin, n_end);
}
constexpr double kNarrowRate = 0.80;  // 256 x 128 tiles (awq_gemm_v4n.hip) vs 256 x 256 (awq_gemm_v6.hip) at equal chip fill (0.83 against awq_gemm_v4.hip, profiles/r01_gemm_v4.txt)
int g_small_m = 1;  // knob gemm_small_m: 0 = the prefill GEMM only takes m >= 256 (see gemm_cdna4_v3_takes)
int g_splitk = 1;  // narrow tiles: split K over blocks when the tiles fill less than half of the chip and a workspace is given
int g_v5 = 0;  // 1: 256-wide tiles run awq_gemm_v5.hip (weights never touch LDS); knob gemm_v5
int g_v6 = 1;  // 1 (default): 256-wide tiles of m >= 256 run awq_gemm_v6.hip (one software-pipelined wave per SIMD); 0: awq_gemm_v4.hip
int g_tile_n = 0;  // knob gemm_tile_n: 128 / 256 force one tile width for callers that pass tile_n = 0 (tests of a specific kernel)
int g_v6_192 = 1;  // knob gemm_v6_192: 0 = no 192-wide blocks in the tile plan
int g_v6_szh = 0;  // knob gemm_v6_szh: 1 = v6 dequantises in the f16-mantissa form when the caller hands its sz_half buffer (-40 VALU per K tile; measured neutral, profiles/r02_gemm_v6.txt)
int g_v4 = 1;  // 1 (default): 256-wide tiles run the hand-scheduled K loop of awq_gemm_v4.hip; 0: v3's compiler-scheduled loop
void launch_wide(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int n_begin,
                 int n_end, int dtype, hipStream_t st, int bits, int epi, const void* szh = nullptr, int tile_n = 256) {
#ifdef AWQ_ENABLE_PROBES  // awq_gemm_v5.hip is an evaluated alternative (profiles/r02_gemm_v5_sweep.txt), not a product path
  if (g_v5) {
    launch_gemm_cdna4_v5(x, qw, szp, bias, out, m, n, k, n_begin, n_end, dtype, st, bits, g_v5 == 3 ? 8 : (g_v5 == 4 ? 17 : 16));
    return;
  }
#endif
  if (g_v6 && m >= 256) {  // (szh: the caller's sz_half side buffer, reported exact for this layer -> the f16-mantissa dequant form)
    if (szh != nullptr && bits == 4 && g_v6_szh) launch_gemm_cdna4_v6(x, qw, szh, bias, out, m, n, k, n_begin, n_end, dtype, st, bits, epi, 1);
    else launch_gemm_cdna4_v6(x, qw, szp, bias, out, m, n, k, n_begin, n_end, dtype, st, bits, epi, 0, tile_n);
    return;
  }
  if (g_v4 || bits == 3 || epi) launch_gemm_cdna4_v4(x, qw, szp, bias, out, m, n, k, n_begin, n_end, dtype, st, bits, epi);
  else if (dtype == 0) launch_v3<F16, 2>(x, qw, szp, bias, out, m, n, k, n_begin, n_end, st);
  else launch_v3<BF16, 2>(x, qw, szp, bias, out, m, n, k, n_begin, n_end, st);
}
void launch_narrow(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int n_begin,
                   int n_end, int dtype, void* ws, size_t ws_bytes, hipStream_t st, int bits, int epi) {
  if (g_v4 || bits == 3 || epi) launch_gemm_cdna4_v4n(x, qw, szp, bias, out, m, n, k, n_begin, n_end, dtype, g_splitk ? ws : nullptr, ws_bytes, st, bits, epi);
  else if (dtype == 0) launch_v3<F16, 1>(x, qw, szp, bias, out, m, n, k, n_begin, n_end, st);
  else launch_v3<BF16, 1>(x, qw, szp, bias, out, m, n, k, n_begin, n_end, st);
}
}  // namespace

namespace {
int g_moe_v4 = 1;  // grouped prefill GEMM on 256 x 256 tiles (awq_gemm_v4.hip); 0: the 128 x 128 grouped kernel
}
bool moe_v4_enabled() { return g_moe_v4 != 0; }
namespace {
int g_moe_v6 = 1;  // grouped prefill GEMM on the v6 tile (awq_gemm_v6.hip); 0: the v4 loop above
}
bool moe_v6_enabled() { return g_moe_v6 != 0; }

int gemm_v3_tune_set(const char* key, int value) {
  if (!strcmp(key, "moe_v6")) g_moe_v6 = value;
  else if (!strcmp(key, "gemm_v4")) g_v4 = value;
  else if (!strcmp(key, "gemm_v6")) {  // units: 0 off, 1 = the 256-wide tiles, 2 = every tile; tens (probe builds): timing-only probe of the kernel
    g_v6 = value % 10;
    gemm_v6_set_probe(value / 10);
  }
#ifdef AWQ_ENABLE_PROBES
  else if (!strcmp(key, "gemm_v5")) g_v5 = value;
#endif
#ifdef AWQ_ENABLE_PROBES
  else if (!strcmp(key, "gemm_v4_probe")) gemm_v4_set_probe(value);
#endif
  else if (!strcmp(key, "gemm_tile_n")) g_tile_n = value;
  else if (!strcmp(key, "gemm_v6_szh")) g_v6_szh = value;
  else if (!strcmp(key, "gemm_v6_192")) g_v6_192 = value;
  else if (!strcmp(key, "moe_v4")) g_moe_v4 = value;
  else if (!strcmp(key, "gemm_small_m")) g_small_m = value;
  else if (!strcmp(key, "gemm_splitk")) {  // 0 = off, 1 = auto, n > 1 = force n K ranges
    g_splitk = value;
    g_v4n_ksplit_force = value > 1 ? value : 0;
  }
  else return -1;
  return 0;
}

// tile_n: 0 = pick by chip fill (256 CUs, one block per CU), 128 / 256 = force one width for the whole matrix.
// In auto mode a matrix whose 256-wide tile count is k full rounds plus a partial one runs the full rounds with 256-wide
// tiles and the remaining weight rows with 128-wide tiles in a second launch when that is faster.
namespace {
struct Plan {
  int mode;        // 0 = all 256-wide, 1 = all 128-wide, 2 = 256-wide for [0, cols_main * 256) + 128-wide for the rest, 3 = all 192-wide
  long cols_main;
};
constexpr double k192Rate = 0.97;  // 256 x 192 blocks of awq_gemm_v6.hip (three slabs per wave) vs its 256 x 256 blocks at equal chip fill
Plan plan_tiles(int m, int n, int tile_n, bool allow192 = false) {
  if (tile_n == 128) return {1, 0};
  if (tile_n == 256) return {0, 0};
  if (tile_n == 192) return {allow192 ? 3 : 0, 0};
  const long tiles_m = (m + TM - 1) / TM;
  auto rounds = [](long t) { return (double)((t + 255) / 256); };
  const long cols256 = (n + 255) / 256, t256 = tiles_m * cols256;
  const double cost_wide = rounds(t256);
  const double cost_narrow = rounds(tiles_m * ((n + 127) / 128)) * 0.5 / kNarrowRate;
  // mixed: as many whole rounds of 256-wide tiles as fit, the remaining rows 128-wide
  const long cols_main = (t256 / 256) * 256 / tiles_m;  // 256-wide column tiles that fill whole rounds
  double cost_mixed = 1e30;
  if (cols_main > 0 && cols_main < cols256) {
    const long n_rest = n - cols_main * 256;
    cost_mixed = rounds(tiles_m * cols_main) + rounds(tiles_m * ((n_rest + 127) / 128)) * 0.5 / kNarrowRate + 0.02;
  }
  // 192-wide blocks: three quarters of the work per block; they win where they turn a partial round into a full one
  const double cost_192 = allow192 ? rounds(tiles_m * ((n + 191) / 192)) * 0.75 / k192Rate : 1e30;
  if (cost_192 < cost_wide && cost_192 < cost_narrow && cost_192 < cost_mixed) return {3, 0};
  if (cost_mixed < cost_wide && cost_mixed < cost_narrow) return {2, cols_main};
  return {cost_narrow < cost_wide ? 1 : 0, 0};
}
}  // namespace

// Which m the prefill GEMM takes.  m >= 256 always; below that the narrow-tile kernel runs ONE row tile whose missing rows are
// not stored, at the cost of a 256-row tile whatever m is, while the skinny kernel (awq_skinny_cdna4.hip) re-streams and
// re-dequantises the weights once per 64 rows: the crossover measured on the Llama shapes (profiles/r01_small_m_sweep.txt) sits
// at m ~ 75 for K >= 8192 and ~150 for K = 4096, which m * K >= 0.6 M approximates; at m = 255 the GEMM is 1.6 - 3.1x faster.
bool gemm_cdna4_v3_takes(int m, int k) {
  if (m >= TM) return true;
  if (g_small_m == 2) return g_v4 && m > 8;  // experiments: every m the skinny kernel would take
  return g_small_m && g_v4 && m >= 72 && (long)m * k >= 600000;
}

// fp32 workspace the call below can use to split K when its tiles under-fill the chip (0 = none needed)
size_t gemm_cdna4_v3_workspace_bytes(int m, int n, int k) {
  if (!gemm_cdna4_v3_takes(m, k) || (n % 16) != 0 || (k % 128) != 0 || !g_v4 || !g_splitk) return 0;
  if (m < TM) return gemm_v4n_workspace_bytes(m, n, k);
  const Plan p = plan_tiles(m, n, 0);
  if (p.mode == 1) return gemm_v4n_workspace_bytes(m, n, k);
  if (p.mode == 2) return gemm_v4n_workspace_bytes(m, n - (int)(p.cols_main * 256), k);
  return 0;
}

size_t gemm_cdna4_v3_workspace_bytes_w3(int m, int n, int k) {
  if (m <= 8 || (n % 16) != 0 || (k % 128) != 0 || !g_splitk) return 0;
  if (m < TM) return gemm_v4n_workspace_bytes(m, n, k);
  const Plan p = plan_tiles(m, n, 0);
  if (p.mode == 1) return gemm_v4n_workspace_bytes(m, n, k);
  if (p.mode == 2) return gemm_v4n_workspace_bytes(m, n - (int)(p.cols_main * 256), k);
  return 0;
}

// which tiles the prefill GEMM would launch for (m, n) of a `bits`-bit matrix: *mode = 0 all 256-wide, 1 all 128-wide, 2 = 256-wide for the
// first *cols_main column tiles + 128-wide for the rest, 3 all 192-wide; returns the number of thread blocks (0: the GEMM does not take m)
int gemm_cdna4_v3_plan(int m, int n, int bits, int* mode, int* cols_main) {
  if (m <= 8 || (n % 16) != 0) return 0;
  const bool allow192 = g_v6 != 0 && g_v6_192 != 0 && bits == 4 && m >= TM;
  const Plan p = m < TM ? Plan{1, 0} : plan_tiles(m, n, g_tile_n, allow192);
  if (mode) *mode = p.mode;
  if (cols_main) *cols_main = (int)p.cols_main;
  const long tm = (m + TM - 1) / TM;
  if (p.mode == 0) return (int)(tm * ((n + 255) / 256));
  if (p.mode == 1) return (int)(tm * ((n + 127) / 128));
  if (p.mode == 3) return (int)(tm * ((n + 191) / 192));
  return (int)(tm * p.cols_main + tm * ((n - p.cols_main * 256 + 127) / 128));
}

int launch_gemm_cdna4_v3(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                         int tile_n, int dtype, void* ws, size_t ws_bytes, hipStream_t st, int bits, int epi, const void* szh) {
  // (w3c tiles have no skinny kernel: the masked single-row-tile path of the narrow kernel serves every m > 8)
  if (!szp || !(gemm_cdna4_v3_takes(m, k) || ((bits == 3 || epi) && m > 8)) || (n % 16) != 0 || (k % 128) != 0 || (size_t)m * (size_t)k >= (1ull << 31)) return -1;
  const bool allow192 = g_v6 != 0 && g_v6_192 != 0 && bits == 4 && m >= TM;
  Plan p = m < TM ? Plan{1, 0} : plan_tiles(m, n, tile_n ? tile_n : g_tile_n, allow192);  // m < 256: only the narrow-tile kernel masks rows
  if (g_v5 >= 2 || g_v6 >= 2) p = Plan{0, 0};                            // experiments: every tile through awq_gemm_v5.hip (it masks rows itself)
  if (p.mode == 3) {
    launch_wide(x, qw, szp, bias, out, m, n, k, 0, n, dtype, st, bits, epi, nullptr, 192);
  } else if (p.mode == 2) {
    launch_wide(x, qw, szp, bias, out, m, n, k, 0, (int)(p.cols_main * 256), dtype, st, bits, epi, szh);
    launch_narrow(x, qw, szp, bias, out, m, n, k, (int)(p.cols_main * 256), n, dtype, ws, ws_bytes, st, bits, epi);
  } else if (p.mode == 1) {
    launch_narrow(x, qw, szp, bias, out, m, n, k, 0, n, dtype, ws, ws_bytes, st, bits, epi);
  } else {
    launch_wide(x, qw, szp, bias, out, m, n, k, 0, n, dtype, st, bits, epi, szh);
  }
  return 0;
}

}  // namespace awq
