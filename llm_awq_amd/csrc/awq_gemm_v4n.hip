// Prefill GEMM v4, narrow tiles: 256 x 128 x 64 (8 waves = 2 along M x 4 along N, 128 x 32 each) with the K loop of
// awq_gemm_v4.hip -- every LDS access placed by hand -- for the shapes whose 256-wide tile count under-fills the 256 CUs:
// every prompt up to ~1 k tokens and the o / down projections at M = 2048 (launch_gemm_cdna4_v3 picks the width).  It
// replaces v3's NSL = 1 instantiation there, whose compiler-scheduled loop exposes the LDS latency of its five fragment
// reads per four MFMAs.  bf16 and fp16; numerics and accumulation order are v3's: results are bit-identical.
//
// Split-K (KSPLIT = 1): when even the narrow tiles fill less than half of the chip (prompts of 256 .. ~1 k tokens against the
// N = 4096 projections: 32 tiles on 256 CUs, a serial K loop of up to 224 K-tiles) every tile is cut into `ksplit` K ranges
// of whole quantisation groups; the blocks store their fp32 partial tiles to a caller-provided workspace and
// splitk_reduce_kernel adds them in split order (deterministic), rounds once, adds the bias.  This is the job of the reference's
// split_k_iters + semaphore (gemm_cuda.cu:546-619, semaphore.h:44-103) without the in-kernel ordering.
//
// PARTIAL = 1 (m < 256, a single row tile; launch_gemm_cdna4_v3 sends prompts of 72 .. 255 rows here when that beats the skinny
// kernel): x rows >= m are not fetched, the product MFMAs of 32-row fragments without rows are skipped, stores are masked.
// The PARTIAL = 0 instantiation is instruction-for-instruction the kernel without these paths.
//
// A wave has ONE weight fragment per k-step, so its successor cannot be re-read into the same registers before the
// step's last MFMA: the weight fragment is double-buffered (wa / wb by step parity) and read at the START of the step
// before; the four x fragments are single-buffered and re-read right after the MFMA that consumes them:
//
//   k-step:   R(wn) | D1 D2 | A1 | cvt W | R(x0') A2 R(x1') A3 R(x2') A4 R(x3')
//   last k-step of a K-tile:  D1 D2 | A1 A2 | barrier | cvt W | R(wn) R(x0') R(x1') A3 R(x2') A4 R(x3')
//
// (A_b = MFMA(w, x_b), D = the dequant MFMAs of this step's weight word, W = its ds_write, R = ds_read_b128 for the next
// k-step.)  LDS operations return in order; queue in front of A1: [.. x0' x1' x2' x3' wn] -> lgkmcnt(4); in front of
// A2 / A3 / A4: four younger reads plus this step's W -> lgkmcnt(4 + job).
#include <type_traits>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

namespace {
constexpr int TM = 256, TN = 128, TK = 64;
constexpr int kTileX = TM * TK * 2;  // 32 KiB x tile [256][64]; LDS: x stage 0 | x stage 1 | w stage 0 | w stage 1
constexpr int kTileW = TN * TK * 2;  // 16 KiB
constexpr int kWBase = 2 * kTileX;
constexpr int WN = 32;  // weight rows per wave
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16-byte granule gc of tile row `row` (128-byte rows).  The XOR term serves both access shapes (MI355X_MICROARCH.md, LDS):
//   * ds_read_b128 fragments (64 banks, lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} of the 32 rows a half-wave reads): the
//     eight even and the eight odd rows of a group must hit eight different granules -- rows & 6 alone repeat, bit 4 separates them;
//   * ds_write_b128 of a dequantised word (32 banks = ONE row width, eight consecutive lanes = eight consecutive rows, same gc):
//     row & 7 must differ.  (Round 1 used (row >> 1) & 7: conflict-free reads, 2-way conflicts on every write = the 21 % of
//     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE in profiles/r01_pmc_gemm_v4.txt.)
__device__ __forceinline__ int swz(int row) { return (row & 7) ^ ((row >> 4) & 1); }
__device__ __forceinline__ int tile_off(int row, int gc) { return row * 128 + ((gc ^ swz(row)) << 4); }

struct Group {  // one quantisation group (128 k) of the wave's slab
  u32x4 w;
  u32 b01, b23;
  float c;
};
struct Raw {
  u32x4 w;
  u32 sz;
};
template <int V>
using ic = std::integral_constant<int, V>;
template <bool V>
using bc = std::integral_constant<bool, V>;
template <int WIDX, int ST>
struct JobT {  // word WIDX (0..3) of the group -> granule 4 (WIDX & 1) + g of weight stage ST
  static constexpr bool has = true;
  static constexpr int widx = WIDX, st = ST;
};
struct NoJob {
  static constexpr bool has = false;
  static constexpr int widx = 0, st = 0;
};
}  // namespace

#define V4N_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define V4N_WRITE(addr, val, off) asm volatile("ds_write_b128 %0, %1 offset:%2\n\ts_nop 1" : : "v"(addr), "v"(val), "n"(off) : "memory")
#define V4N_FENCE() __builtin_amdgcn_sched_barrier(0)

// BITS 4: cdna4 W4 tiles (1 KiB);  BITS 3: w3c tiles (768 B: three words per lane, the fourth rebuilt by w3_expand once per group)
template <typename DT, int KSPLIT, int PARTIAL, int BITS = 4>
__global__ __launch_bounds__(512) void gemm_cdna4_v4n_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                             const u32* __restrict__ szp, const uint16_t* __restrict__ bias,
                                                             uint16_t* __restrict__ out, int M, int N, int K, int tiles_m,
                                                             int tiles_n, int n_begin, int n_end, int ksplit,
                                                             float* __restrict__ partial, int epi) {
  using vec8 = typename DT::vec8;
  constexpr int kEpiRow = 2 * WN + 16;  // bytes per staged output row (+16 pad)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int l32 = lane & 31, hk = lane >> 5;
  const int wm = wv >> 2, wn = wv & 3;

  // XCD-aware, two-row-band tile order: as v3 (awq_gemm_v3.hip)
  const int T = tiles_m * tiles_n;
  const int split = KSPLIT ? (int)(blockIdx.x % (unsigned)ksplit) : 0;
  int tile;
  {
    const int b = KSPLIT ? (int)(blockIdx.x / (unsigned)ksplit) : (int)blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int q = T >> 3, r = T & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int tm, tn;
  {
    const int full = (tiles_m >> 1) * 2 * tiles_n;
    if (tile < full) {
      const int band = tile / (2 * tiles_n), rem = tile - band * 2 * tiles_n;
      tn = rem >> 1;
      tm = 2 * band + (rem & 1);
    } else {
      tn = tile - full;
      tm = tiles_m - 1;
    }
  }
  const int m0 = max(min(tm * TM, M - TM), 0), n0 = n_begin + tn * TN;  // (M < 256: one row tile, rows >= M clamped / not stored)
  const int nit_all = K >> 7;                                 // quantisation groups of the matrix
  const int g0 = KSPLIT ? split * nit_all / ksplit : 0;       // this block's K range: groups [g0, g0 + nit), ranges differ by <= 1
  const int nit = KSPLIT ? (split + 1) * nit_all / ksplit - g0 : nit_all;

  // ---- x tile: LDS-DMA, 4 x 16 B per thread per K-tile; swizzle applied to the SOURCE granule ----
  u32 a_off[4];
  {
    const int row = tid >> 3, gcp = tid & 7;
    const int gc = gcp ^ swz(row);
#pragma unroll
    for (int q = 0; q < 4; ++q) a_off[q] = (u32)min(m0 + row + 64 * q, M - 1) * (u32)K + gc * 8;  // (64 q keeps swz(row))
  }
  // PARTIAL: rows >= M are not even fetched (their LDS rows keep stale bits; output rows depend on their own x row only and
  // rows >= M are never stored)
  const int a_rows = PARTIAL ? M - m0 - (tid >> 3) : 256;  // DMA q is live iff 64 q < a_rows
  auto issue_a = [&](int kt, int stage) {
    char* dst = smem + stage * kTileX + wv * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint16_t* xq = x + (size_t)(kt + 2 * g0) * TK;
      if (!PARTIAL || 64 * q < a_rows)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xq + a_off[q]),
                                       (__attribute__((address_space(3))) void*)(dst + q * 8192), 16, 0, 0);
    }
  };

  // ---- weight tile: wave wv owns slab wv of the 128-row tile ----
  const int nslab = N >> 4;
  const int sl = min((n0 >> 4) + wv, min(nslab, n_end >> 4) - 1);
  constexpr int kTileWords = BITS == 4 ? 256 : 192, kLaneWords = BITS == 4 ? 4 : 3;
  const u32 b_off = (u32)sl * nit_all * kTileWords + lane * kLaneWords, sz_off = (u32)sl * nit_all * 16 + i;
  const int nl = 16 * wv + i;  // tile row of the lane's weight row
  Cdna4DequantT<DT> cd;
  cd.init(lane, BITS == 4 ? 0x000F000Fu : 0x00070007u);
  auto load_group = [&](int grp) {
    Raw r;
    const u32* wp = qw + (size_t)(g0 + grp) * kTileWords + b_off;
    if (BITS == 4) {
      r.w = *reinterpret_cast<const u32x4*>(wp);
    } else {
      typedef u32 u32x3 __attribute__((ext_vector_type(3)));
      const u32x3 w3 = *reinterpret_cast<const u32x3*>(wp);
      r.w = u32x4{w3.x, w3.y, w3.z, 0u};
    }
    r.sz = szp[(size_t)(g0 + grp) * 16 + sz_off];
    return r;
  };
  auto prep = [&](const Raw& r) {
    Group gq;
    gq.w = BITS == 4 ? r.w : w3_expand(r.w.x, r.w.y, r.w.z);
    const u32 sd = (r.sz & 0xFFFFu) * 0x00010001u;
    gq.b01 = sd & cd.m01;
    gq.b23 = sd & cd.m23;
    gq.c = DT::dq_offset(r.sz);
    return gq;
  };

  // ---- LDS byte addresses ----
  const u32 lds0 = (u32)(size_t)(__attribute__((address_space(3))) char*)smem;
  u32 xa[4], wa[4], ja[2];  // per k-step fragment addresses (stage 0, fragment 0); job destinations (word parity)
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    xa[ks] = lds0 + tile_off(wm * 128 + l32, 2 * ks + hk);
    wa[ks] = lds0 + kWBase + tile_off(wn * WN + l32, 2 * ks + hk);
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) ja[b] = lds0 + kWBase + tile_off(nl, 4 * b + g);

  u32x4 wf[2], x0, x1, x2, x3;  // weight fragment by step parity; x fragments (single set)
  f32x16 acc[4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  // PARTIAL (m < 256, one row tile): a wave's 32-row x fragment b has rows only if wm * 128 + 32 b < M; product MFMAs of the
  // others are skipped behind a wave-uniform branch (their LDS reads stay: the lgkmcnt ladders count them)
  const int nb = PARTIAL ? __builtin_amdgcn_readfirstlane(min(max((M - wm * 128 + 31) >> 5, 0), 4)) : 4;
  auto mfb = [&](auto b_, const u32x4& a, const u32x4& b) {
    constexpr int B = decltype(b_)::value;
    if (!PARTIAL || nb > B) acc[B] = DT::mfma32(__builtin_bit_cast(vec8, a), __builtin_bit_cast(vec8, b), acc[B]);
  };

  Group gc;
  // one k-step (header comment).  P: parity (which weight register is current); SN / KN: weight stage (x stage = the same
  // index) and k-step whose fragments are read for the next step; RD: read them; BAR: last k-step of a K-tile; job.
  auto step = [&](auto p_, auto sn_, auto kn_, auto rd_, auto bar_, auto job_) {
    constexpr int P = decltype(p_)::value, SN = decltype(sn_)::value, KN = decltype(kn_)::value;
    constexpr bool RD = decltype(rd_)::value, BAR = decltype(bar_)::value;
    using J = decltype(job_);
    constexpr int JW = J::has ? 1 : 0;
    u32x4& wc = wf[P];
    u32x4& wnx = wf[P ^ 1];
    typename Cdna4DequantT<DT>::Pending pj;
    if constexpr (!BAR) {
      if constexpr (RD) V4N_READ(wnx, wa[KN], SN * kTileW);
      if constexpr (J::has) {
        const u32 word = J::widx == 0 ? gc.w.x : (J::widx == 1 ? gc.w.y : (J::widx == 2 ? gc.w.z : gc.w.w));
        pj = cd.word_issue(word, gc.b01, gc.b23, gc.c);
      }
      V4N_FENCE();
      asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(wc), "+v"(x0) : "n"(RD ? 4 : 3));
      mfb(ic<0>{}, wc, x0);
      if constexpr (J::has) {
        const vec8 v = Cdna4DequantT<DT>::word_finish(pj);
        V4N_WRITE(ja[J::widx & 1], __builtin_bit_cast(u32x4, v), J::st * kTileW);
      }
      V4N_FENCE();
      if constexpr (RD) V4N_READ(x0, xa[KN], SN * kTileX);
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x1) : "n"(RD ? 4 + JW : 2 + JW));
      mfb(ic<1>{}, wc, x1);
      V4N_FENCE();
      if constexpr (RD) V4N_READ(x1, xa[KN], SN * kTileX + 4096);
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x2) : "n"(RD ? 4 + JW : 1 + JW));
      mfb(ic<2>{}, wc, x2);
      V4N_FENCE();
      if constexpr (RD) V4N_READ(x2, xa[KN], SN * kTileX + 8192);
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x3) : "n"(RD ? 4 + JW : 0 + JW));
      mfb(ic<3>{}, wc, x3);
      V4N_FENCE();
      if constexpr (RD) V4N_READ(x3, xa[KN], SN * kTileX + 12288);
      V4N_FENCE();
    } else {
      // the next tile's stage may be read only behind the barrier: queue in front of A1 is [x0' x1' x2' x3'] (the weight
      // fragment of this step was read a step ago)
      if constexpr (J::has) {
        const u32 word = J::widx == 0 ? gc.w.x : (J::widx == 1 ? gc.w.y : (J::widx == 2 ? gc.w.z : gc.w.w));
        pj = cd.word_issue(word, gc.b01, gc.b23, gc.c);
      }
      V4N_FENCE();
      asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(wc), "+v"(x0));
      mfb(ic<0>{}, wc, x0);
      V4N_FENCE();
      asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(x1));
      mfb(ic<1>{}, wc, x1);
      V4N_FENCE();
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x2), "+v"(x3));
      __syncthreads();  // + vmcnt(0): the next tile's x DMA and packed words have landed; every read of this tile's stages retired
      V4N_FENCE();
      if constexpr (J::has) {
        const vec8 v = Cdna4DequantT<DT>::word_finish(pj);
        V4N_WRITE(ja[J::widx & 1], __builtin_bit_cast(u32x4, v), J::st * kTileW);
      }
      if constexpr (RD) {
        V4N_READ(wnx, wa[KN], SN * kTileW);
        V4N_READ(x0, xa[KN], SN * kTileX);
        V4N_READ(x1, xa[KN], SN * kTileX + 4096);
      }
      V4N_FENCE();
      mfb(ic<2>{}, wc, x2);
      V4N_FENCE();
      if constexpr (RD) V4N_READ(x2, xa[KN], SN * kTileX + 8192);
      mfb(ic<3>{}, wc, x3);
      V4N_FENCE();
      if constexpr (RD) V4N_READ(x3, xa[KN], SN * kTileX + 12288);
      V4N_FENCE();
    }
  };

  // ---------------- prologue: tile 0 complete in stage 0, tile 1's x tile in flight, its first weight word written ----
  issue_a(0, 0);
  gc = prep(load_group(0));
  Raw rn = load_group(nit > 1 ? 1 : 0);
  {
    char* Bs = smem + kWBase;
    *reinterpret_cast<vec8*>(Bs + tile_off(nl, 0 + g)) = cd.word(gc.w.x, gc.b01, gc.b23, gc.c);
    *reinterpret_cast<vec8*>(Bs + tile_off(nl, 4 + g)) = cd.word(gc.w.y, gc.b01, gc.b23, gc.c);
  }
  __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and the ds_writes
  asm volatile("" : "+v"(rn.w), "+v"(rn.sz));  // as v3: no pending ordinary load enters the loop
  issue_a(1, 1);
  V4N_FENCE();
  {
    const vec8 v = cd.word(gc.w.z, gc.b01, gc.b23, gc.c);  // first word of tile 1 = (group 0, half 1) -> stage 1
    V4N_WRITE(ja[0], __builtin_bit_cast(u32x4, v), kTileW);
  }
  V4N_READ(wf[0], wa[0], 0);
  V4N_READ(x0, xa[0], 0);
  V4N_READ(x1, xa[0], 4096);
  V4N_READ(x2, xa[0], 8192);
  V4N_READ(x3, xa[0], 12288);
  V4N_FENCE();

  // One iteration = one quantisation group = two K-tiles (2q in stage 0, 2q + 1 in stage 1); the last group is peeled.
  // Word w of a group: tile half h = w >> 1, granule parity w & 1.  Words of tile t+1 are produced during tile t: its word
  // (w & 1) == 0 right behind the barrier that ends tile t-1 ... i.e. one word every other k-step (v3's slots 0 and 2).
  auto group_iter = [&](int q, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value;
    using Tt = bc<true>;
    using Ff = bc<false>;
    // ---------- K-tile 2q (stage 0); the second word of tile 2q+1 = (group q, word 3) goes to stage 1 in its k-step 2 ----------
    step(ic<0>{}, ic<0>{}, ic<1>{}, Tt{}, Ff{}, NoJob{});
    step(ic<1>{}, ic<0>{}, ic<2>{}, Tt{}, Ff{}, JobT<3, 1>{});
    step(ic<0>{}, ic<0>{}, ic<3>{}, Tt{}, Ff{}, NoJob{});
    if constexpr (more) {
      gc = prep(rn);  // group q+1: loaded one iteration ago, retired by the previous barrier's vmcnt(0)
      step(ic<1>{}, ic<1>{}, ic<0>{}, Tt{}, Tt{}, JobT<0, 0>{});  // barrier inside; first word of tile 2q+2 -> stage 0
      rn = load_group(min(q + 2, nit - 1));
      issue_a(2 * q + 2, 0);
      V4N_FENCE();
      step(ic<0>{}, ic<1>{}, ic<1>{}, Tt{}, Ff{}, NoJob{});
      step(ic<1>{}, ic<1>{}, ic<2>{}, Tt{}, Ff{}, JobT<1, 0>{});
      step(ic<0>{}, ic<1>{}, ic<3>{}, Tt{}, Ff{}, NoJob{});
      step(ic<1>{}, ic<0>{}, ic<0>{}, Tt{}, Tt{}, JobT<2, 1>{});  // barrier inside; first word of tile 2q+3 -> stage 1
      issue_a(2 * q + 3, 1);
      V4N_FENCE();
    } else {
      step(ic<1>{}, ic<1>{}, ic<0>{}, Tt{}, Tt{}, NoJob{});
      step(ic<0>{}, ic<1>{}, ic<1>{}, Tt{}, Ff{}, NoJob{});
      step(ic<1>{}, ic<1>{}, ic<2>{}, Tt{}, Ff{}, NoJob{});
      step(ic<0>{}, ic<1>{}, ic<3>{}, Tt{}, Ff{}, NoJob{});
      step(ic<1>{}, ic<0>{}, ic<0>{}, Ff{}, Tt{}, NoJob{});
    }
  };
  for (int q = 0; q + 1 < nit; ++q) group_iter(q, std::true_type{});
  group_iter(nit - 1, std::false_type{});

  if constexpr (KSPLIT != 0) {
    // fp32 partial tile -> workspace, in register order: [tile][split][wave][b][j][lane] float4 (1 KiB per wave store);
    // splitk_reduce_kernel reads it back with the same indexing
    float* base = partial + (((size_t)tile * ksplit + split) * 8 + wv) * (4 * 4 * 64 * 4);
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<f32x4*>(base + ((b * 4 + j) * 64 + lane) * 4) =
            f32x4{acc[b][4 * j + 0], acc[b][4 * j + 1], acc[b][4 * j + 2], acc[b][4 * j + 3]};
    return;
  }
  if (epi == 3) {
    // K shard of a tensor-parallel row split (awq_w4a16_partial_cdna4): out is float [M, N], the fp32 accumulators unrounded, no bias
    float* o32 = reinterpret_cast<float*>(out);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int m = m0 + wm * 128 + b * 32 + l32;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int nn = n0 + wn * WN + 8 * j + 4 * hk;
        if (nn < n_end && m < M)
          __builtin_nontemporal_store(f32x4{acc[b][4 * j + 0], acc[b][4 * j + 1], acc[b][4 * j + 2], acc[b][4 * j + 3]}, reinterpret_cast<f32x4*>(o32 + (size_t)m * N + nn));
      }
    }
    return;
  }
  // ---------------- epilogue through LDS: acc[b][r] = C[n = wn*32 + (r&3) + 8 (r>>2) + 4 hk][m = wm*128 + b*32 + l32] ----
  __syncthreads();
  char* eb = smem + wv * (128 * kEpiRow);
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u32x2 v;
      v.x = (u32)DT::from_float(acc[b][4 * j + 0]) | ((u32)DT::from_float(acc[b][4 * j + 1]) << 16);
      v.y = (u32)DT::from_float(acc[b][4 * j + 2]) | ((u32)DT::from_float(acc[b][4 * j + 3]) << 16);
      *reinterpret_cast<u32x2*>(eb + (b * 32 + l32) * kEpiRow + (8 * j + 4 * hk) * 2) = v;
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's own LDS writes (region is wave-private)
  __builtin_amdgcn_wave_barrier();
  constexpr int GR = WN / 8, RP = 64 / GR;  // 4 granules per staged row, 16 rows per pass
#pragma unroll
  for (int ps = 0; ps < 128 / RP; ++ps) {
    const int row = ps * RP + lane / GR, gc2 = lane % GR;
    const int m = m0 + wm * 128 + row, nn = n0 + wn * WN + gc2 * 8;
    u32x4 v = *reinterpret_cast<const u32x4*>(eb + row * kEpiRow + gc2 * 16);
    if (epi == 2) {  // QuantLlamaMLP's interleaved gate / up pair: see awq_gemm_v4.hip
      if ((gc2 & 1) == 0 && nn < n_end && m < M) {
        const u32x4 u = *reinterpret_cast<const u32x4*>(eb + row * kEpiRow + (gc2 + 1) * 16);
        __builtin_nontemporal_store(silu_mul_octet<DT>(v, u), reinterpret_cast<u32x4*>(out + (size_t)m * (N >> 1) + (nn >> 1)));
      }
      continue;
    }
    if (nn < n_end && m < M) {
      if (bias != nullptr) {  // `out + self.bias` in T (qmodule.py:221)
        const u32x4 bv = *reinterpret_cast<const u32x4*>(bias + nn);
        auto add2 = [](u32 a, u32 b) {
          const float lo = DT::to_float((uint16_t)(a & 0xFFFFu)) + DT::to_float((uint16_t)(b & 0xFFFFu));
          const float hi = DT::to_float((uint16_t)(a >> 16)) + DT::to_float((uint16_t)(b >> 16));
          return (u32)DT::from_float(lo) | ((u32)DT::from_float(hi) << 16);
        };
        v = u32x4{add2(v.x, bv.x), add2(v.y, bv.y), add2(v.z, bv.z), add2(v.w, bv.w)};
      }
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out + (size_t)m * N + nn));  // streamed: keep the x / weight panels in L2
    }
  }
}

// out tile = T(sum over the K ranges of the fp32 partials, in range order) (+ bias in T).  One wave per (tile, wave, b):
// a 32-row x 32-column piece of the output; the sums are staged through LDS so that the stores are 16 B per lane, four
// lanes per output row, as in the unsplit epilogue.
template <typename DT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, const uint16_t* __restrict__ bias,
                                                            uint16_t* __restrict__ out, int M, int N, int tiles_m, int tiles_n,
                                                            int n_begin, int n_end, int ksplit) {
  constexpr int kRow = 2 * WN + 16;  // bytes per staged row (+16 pad)
  __shared__ __attribute__((aligned(16))) char stage[4][32 * kRow];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int piece = blockIdx.x * 4 + w;  // over tiles * 8 waves * 4 b
  const int b = piece & 3, wv = (piece >> 2) & 7, tile = piece >> 5;
  if (tile >= tiles_m * tiles_n) return;
  const int l32 = lane & 31, hk = lane >> 5, wm = wv >> 2, wn = wv & 3;
  int tm, tn;  // the tile walk of gemm_cdna4_v4n_kernel (the linear index is the one the partials were stored under)
  {
    const int full = (tiles_m >> 1) * 2 * tiles_n;
    if (tile < full) {
      const int band = tile / (2 * tiles_n), rem = tile - band * 2 * tiles_n;
      tn = rem >> 1;
      tm = 2 * band + (rem & 1);
    } else {
      tn = tile - full;
      tm = tiles_m - 1;
    }
  }
  constexpr size_t kWaveFloats = 4 * 4 * 64 * 4;  // one wave's accumulators
  const float* p = partial + (((size_t)tile * ksplit) * 8 + wv) * kWaveFloats + ((size_t)(b * 4) * 64 + lane) * 4;
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s2 = 0; s2 < ksplit; ++s2) {
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(p + (size_t)s2 * 8 * kWaveFloats + j * 256);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += v[j];
  }
  char* eb = stage[w];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    u32x2 v;
    v.x = (u32)DT::from_float(acc[j][0]) | ((u32)DT::from_float(acc[j][1]) << 16);
    v.y = (u32)DT::from_float(acc[j][2]) | ((u32)DT::from_float(acc[j][3]) << 16);
    *reinterpret_cast<u32x2*>(eb + l32 * kRow + (8 * j + 4 * hk) * 2) = v;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0); the staging region is wave-private
  __builtin_amdgcn_wave_barrier();
  const int m_base = max(min(tm * TM, M - TM), 0) + wm * 128 + b * 32, n_base = n_begin + tn * TN + wn * WN;
#pragma unroll
  for (int ps = 0; ps < 2; ++ps) {
    const int row = ps * 16 + (lane >> 2), gc2 = lane & 3;
    const int nn = n_base + gc2 * 8;
    u32x4 v = *reinterpret_cast<const u32x4*>(eb + row * kRow + gc2 * 16);
    if (nn < n_end && m_base + row < M) {
      if (bias != nullptr) {  // `out + self.bias` in T (qmodule.py:221)
        const u32x4 bv = *reinterpret_cast<const u32x4*>(bias + nn);
        auto add2 = [](u32 a, u32 c) {
          const float lo = DT::to_float((uint16_t)(a & 0xFFFFu)) + DT::to_float((uint16_t)(c & 0xFFFFu));
          const float hi = DT::to_float((uint16_t)(a >> 16)) + DT::to_float((uint16_t)(c >> 16));
          return (u32)DT::from_float(lo) | ((u32)DT::from_float(hi) << 16);
        };
        v = u32x4{add2(v.x, bv.x), add2(v.y, bv.y), add2(v.z, bv.z), add2(v.w, bv.w)};
      }
      *reinterpret_cast<u32x4*>(out + (size_t)(m_base + row) * N + nn) = v;
    }
  }
}

// How many K ranges a launch over n_cols weight rows should use (1 = none).  Costs in units of one K-group iteration of one
// block (~1.6 us), fitted to profiles/r01_splitk_sweep.txt: an unsplit block pays its groups + ~2 (pipeline fill, epilogue);
// a split block its groups + ~0.6; blocks run in rounds of 256 (one per CU); every (tile, range) costs 128 KiB of fp32
// partials written and read back (~0.041 per block) and the second launch ~1.3.
int g_v4n_ksplit_cap = 1;   // knob gemm_splitk_cap: 0 = the round-1 rule without the skip below
int g_v4n_ksplit_force = 0;  // > 1: use exactly this many ranges (knob gemm_splitk = n; experiments)
int gemm_v4n_ksplit(int m, int n_cols, int k) {
  const long tiles = (long)((m + TM - 1) / TM) * ((n_cols + TN - 1) / TN);
  const int nit = k / 128;
  if (g_v4n_ksplit_force > 1) return g_v4n_ksplit_force <= nit && tiles * g_v4n_ksplit_force <= 1024 ? g_v4n_ksplit_force : 1;
  auto rounds = [](long blocks) { return (double)((blocks + 255) / 256); };
  const double unsplit = rounds(tiles) * (nit + 2.0);
  double best_cost = 0.9 * unsplit;  // a split has to be clearly better
  int best = 1;
  for (int d = 2; d <= 16 && nit / d >= 2; ++d) {
    // (round 6 re-sweep, profiles/r06_splitk_sweep.txt: a full round of blocks with eight or fewer K groups each loses to three quarters of a round with
    // longer loops -- o_proj at 512 rows: 4 ranges x 64 tiles 37.2 us, 3 ranges 31.9 us; at 256 rows 8 ranges 32.9, 6 ranges 26.4)
    if (g_v4n_ksplit_cap && tiles * d > 192 && tiles * d <= 256 && nit / d <= 8) continue;
    const double cost = rounds(tiles * d) * ((nit + d - 1) / d + 0.6) + 0.041 * (double)(tiles * d) + 1.3;
    if (cost < best_cost) {
      best_cost = cost;
      best = d;
    }
  }
  return best;
}
size_t gemm_v4n_workspace_bytes(int m, int n_cols, int k) {
  const int ks = gemm_v4n_ksplit(m, n_cols, k);
  if (ks <= 1) return 0;
  const size_t tiles = (size_t)((m + TM - 1) / TM) * ((n_cols + TN - 1) / TN);
  return tiles * ks * (size_t)TM * TN * 4;
}

// weight rows [n_begin, n_end) of the matrix with 256 x 128 tiles; same contract as v3's launch_v3<DT, 1>, and also m < 256
// (one row tile whose missing rows are computed from row m - 1 and not stored).
// ws / ws_bytes: optional fp32 workspace; when it holds gemm_v4n_workspace_bytes() the K loop is split (see the header)
void launch_gemm_cdna4_v4n(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                           int n_begin, int n_end, int dtype, void* ws, size_t ws_bytes, hipStream_t st, int bits, int epi) {
  constexpr int smem_main = 2 * kTileX + 2 * kTileW;
  constexpr int smem_epi = 8 * 128 * (2 * WN + 16);
  constexpr int smem = smem_main > smem_epi ? smem_main : smem_epi;
  const int tiles_m = (m + TM - 1) / TM, tiles_n = (n_end - n_begin + TN - 1) / TN;
  using Kern = void (*)(const uint16_t*, const u32*, const u32*, const uint16_t*, uint16_t*, int, int, int, int, int, int, int, int, float*, int);
  static const Kern kerns4[2][2][2] = {  // [dtype][split][partial]
      {{gemm_cdna4_v4n_kernel<F16, 0, 0>, gemm_cdna4_v4n_kernel<F16, 0, 1>}, {gemm_cdna4_v4n_kernel<F16, 1, 0>, gemm_cdna4_v4n_kernel<F16, 1, 1>}},
      {{gemm_cdna4_v4n_kernel<BF16, 0, 0>, gemm_cdna4_v4n_kernel<BF16, 0, 1>}, {gemm_cdna4_v4n_kernel<BF16, 1, 0>, gemm_cdna4_v4n_kernel<BF16, 1, 1>}}};
  static const Kern kerns3[2][2][2] = {
      {{gemm_cdna4_v4n_kernel<F16, 0, 0, 3>, gemm_cdna4_v4n_kernel<F16, 0, 1, 3>}, {gemm_cdna4_v4n_kernel<F16, 1, 0, 3>, gemm_cdna4_v4n_kernel<F16, 1, 1, 3>}},
      {{gemm_cdna4_v4n_kernel<BF16, 0, 0, 3>, gemm_cdna4_v4n_kernel<BF16, 0, 1, 3>}, {gemm_cdna4_v4n_kernel<BF16, 1, 0, 3>, gemm_cdna4_v4n_kernel<BF16, 1, 1, 3>}}};
  const Kern (*kerns)[2][2] = bits == 3 ? kerns3 : kerns4;
  static LdsOptIn optin[2][8];  // per (kernel, device)
  for (int a = 0; a < 8; ++a) optin[bits == 3][a].ensure(reinterpret_cast<const void*>(kerns[a >> 2][(a >> 1) & 1][a & 1]), smem);
  const int dt = dtype == 0 ? 0 : 1, partial = m <= TM - 32 ? 1 : 0;  // a whole 32-row fragment of the single row tile is empty
  const int ks = gemm_v4n_ksplit(m, n_end - n_begin, k);
  const size_t need = ks > 1 ? (size_t)tiles_m * tiles_n * ks * TM * TN * 4 : 0;
  if (epi == 0 && ks > 1 && ws != nullptr && ws_bytes >= need && (reinterpret_cast<uintptr_t>(ws) & 15) == 0) {  // (the split-K reduce has no fused tail)
    hipLaunchKernelGGL(kerns[dt][1][partial], dim3(tiles_m * tiles_n * ks), dim3(512), smem, st, (const uint16_t*)x, (const u32*)qw,
                       (const u32*)szp, (const uint16_t*)nullptr, (uint16_t*)out, m, n, k, tiles_m, tiles_n, n_begin, n_end, ks, (float*)ws, 0);
    auto red = dtype == 0 ? splitk_reduce_kernel<F16> : splitk_reduce_kernel<BF16>;
    hipLaunchKernelGGL(red, dim3((unsigned)(tiles_m * tiles_n * 8)), dim3(256), 0, st, (const float*)ws, (const uint16_t*)bias, (uint16_t*)out,
                       m, n, tiles_m, tiles_n, n_begin, n_end, ks);
    return;
  }
  hipLaunchKernelGGL(kerns[dt][0][partial], dim3(tiles_m * tiles_n), dim3(512), smem, st, (const uint16_t*)x, (const u32*)qw,
                     (const u32*)szp, (const uint16_t*)bias, (uint16_t*)out, m, n, k, tiles_m, tiles_n, n_begin, n_end, 1, (float*)nullptr, epi);
}

}  // namespace awq
