// Prefill GEMM v4 on cdna4-interleaved weights (bf16 and fp16, gfx950): the tiling and data flow of awq_gemm_v3.hip's 256 x 256 x 64
// kernel (8 waves = 2 along M x 4 along N, 128 x 64 each, v_mfma_f32_32x32x16_bf16, x tile by LDS-DMA, weight tile
// dequantised on the matrix core one word per k-step, double-buffered LDS) with every LDS access of the K loop placed by
// hand.
//
// Why: at 256 VGPRs hipcc shortens live ranges by sinking v3's fragment reads next to their uses -- its K loop has two
// exposed LDS round trips per k-step ([4 MFMA] read wait [4 MFMA] 5 reads wait) and the matrix pipe is busy 66 % of the
// time (profiles/r01_pmc_gemm_v3.txt).  Here the fragment reads and the weight-tile writes are `asm volatile`
// statements hipcc neither counts nor moves, each fragment is re-read for the NEXT k-step right after the last MFMA
// that consumes it, and the waits are counted `s_waitcnt lgkmcnt(N)` ladders naming the registers they guard:
//
//   k-step:   A1 A2 A3 A4 | D1 D2 | [barrier] | R(w0') | B1 | cvt, W | R(x0') B2 R(x1') B3 R(x2') B4 R(x3') R(w1')
//
// (A_b = MFMA(w0, x_b), B_b = MFMA(w1, x_b), D = the two dequant MFMAs of this step's weight word, W = its ds_write,
// R = ds_read_b128 of a fragment of the next k-step.)  LDS operations return in order, so in front of A1 the queue is
// [w0' W x0' x1' x2' x3' w1']: A1 needs lgkmcnt(4), A2 (3), A3 (2), A4 (1), B1 needs w1' = lgkmcnt(1) behind R(w0'').
// Numerics are v3's (same products, same fp32 accumulation order): results are bit-identical.
#include <type_traits>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

namespace {
constexpr int TM = 256, TN = 256, TK = 64;
constexpr int kTileX = TM * TK * 2;  // 32 KiB: x tile [256][64] bf16; LDS: x stage 0 | x stage 1 | w stage 0 | w stage 1
constexpr int kTileW = TN * TK * 2;  // 32 KiB
constexpr int kWBase = 2 * kTileX;
constexpr int WN = 64;               // weight rows per wave
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16-byte granule gc of tile row `row` (128-byte rows).  The XOR term serves both access shapes (MI355X_MICROARCH.md, LDS):
//   * ds_read_b128 fragments (64 banks, lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} of the 32 rows a half-wave reads): the
//     eight even and the eight odd rows of a group must hit eight different granules -- rows & 6 alone repeat, bit 4 separates them;
//   * ds_write_b128 of a dequantised word (32 banks = ONE row width, eight consecutive lanes = eight consecutive rows, same gc):
//     row & 7 must differ.  (Round 1 used (row >> 1) & 7: conflict-free reads, 2-way conflicts on every write = the 21 % of
//     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE in profiles/r01_pmc_gemm_v4.txt.)
__device__ __forceinline__ int swz(int row) { return (row & 7) ^ ((row >> 4) & 1); }
__device__ __forceinline__ int tile_off(int row, int gc) { return row * 128 + ((gc ^ swz(row)) << 4); }

struct Group {  // one quantisation group (128 k) of the wave's two slabs
  u32x4 w[2];
  u32 b01[2], b23[2];
  float c[2];
};
struct Raw {
  u32x4 w[2];
  u32 sz[2];
};
template <int V>
using ic = std::integral_constant<int, V>;
template <bool V>
using bc = std::integral_constant<bool, V>;
template <int H, int J, int ST>
struct JobT {  // word 2 H + (J & 1) of slab (J >> 1) -> weight stage ST
  static constexpr bool has = true;
  static constexpr int h = H, j = J, st = ST;
};
struct NoJob {
  static constexpr bool has = false;
  static constexpr int h = 0, j = 0, st = 0;
};
}  // namespace

// LDS accesses outside hipcc's s_waitcnt bookkeeping and scheduling (the "memory" clobber keeps their mutual order)
#define V4_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define V4_WRITE(addr, val, off) asm volatile("ds_write_b128 %0, %1 offset:%2\n\ts_nop 1" : : "v"(addr), "v"(val), "n"(off) : "memory")
#define V4_FENCE() __builtin_amdgcn_sched_barrier(0)

// (A variant with double-buffered x fragments -- the reads of k-step s+1 issued in cluster A of k-step s, 248 VGPRs -- measured
// 5-8 % SLOWER, profiles/r01_gemm_v4.txt: read-to-use distance is not what limits this loop.)
// PROBE (only with -DAWQ_ENABLE_PROBES): experiments (timing only, wrong results): 1 = the x DMA always fetches K-tile 0 (cache hits), 2 = no x DMA,
// 3 = no epilogue (one dword per lane is stored so that the accumulators stay live), 4 = no block barrier inside the K loop (the
// vmcnt / lgkmcnt waits stay), 5 = no weight production inside the K loop (no dequant MFMAs, cvt, ds_write), 6 = production
// arithmetic without its ds_write, 7 = the ds_write without the arithmetic, 8 = neither production nor x DMA
// one 256 x 256 output tile: rows [m0, m0 + 256) of x (all of them must exist), weight rows [n0, n0 + 256) clipped to n_end;
// stores are masked to rows [row_lo, row_hi) (dense: every row of the tile; grouped: the expert's rows inside it)
template <typename DT, int PROBE, int BITS = 4>
__device__ __forceinline__ void v4_tile(char* smem, const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                        const u32* __restrict__ szp, const uint16_t* __restrict__ bias,
                                        uint16_t* __restrict__ out, int N, int K, int m0, int n0, int n_end, int row_lo,
                                        int row_hi, int epi = 0) {
  constexpr int kEpiRow = 2 * WN + 16;  // bytes per staged output row (+16 pad)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int l32 = lane & 31, hk = lane >> 5;
  const int wm = wv >> 2, wn = wv & 3;
  const int nit = K >> 7;

  // ---- x tile: LDS-DMA, 4 x 16 B per thread per K-tile; swizzle applied to the SOURCE granule ----
  u32 a_off0;
  {
    const int row = tid >> 3, gcp = tid & 7;
    const int gc = gcp ^ swz(row);
    a_off0 = (u32)(m0 + row) * (u32)K + gc * 8;
  }
  // pieces [q0, q1) of the x tile of K-tile kt (piece q = rows 64 q .. 64 q + 63 of the tile, 8 rows per wave)
  auto issue_a_pieces = [&](int kt, int stage, int q0, int q1) {
    if (PROBE == 2 || PROBE == 8) return;
    if (PROBE == 1) kt = 0;
    char* dst = smem + stage * kTileX + wv * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q < q0 || q >= q1) continue;
      const uint16_t* xq = x + (size_t)kt * TK + (size_t)q * 64 * K;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xq + a_off0),
                                       (__attribute__((address_space(3))) void*)(dst + q * 8192), 16, 0, 0);
    }
  };
  auto issue_a = [&](int kt, int stage) { issue_a_pieces(kt, stage, 0, 4); };
  // SPREAD: how the four pieces of a tile's DMA are spaced over the k-steps behind the barrier that frees their stage (each
  // LDS-DMA instruction holds the wave's issue for ~100-185 cycles next to MFMAs and ds_reads -- MI355X_MICROARCH.md -- and both
  // waves of a SIMD stand at the same point of the loop): slot 0 = right behind the barrier, slots 1..3 = behind the next k-steps
  constexpr int SPREAD = PROBE == 9 ? 1 : (PROBE == 10 ? 2 : 0);
  auto issue_a_slot = [&](int kt, int stage, int slot) {
    if (SPREAD == 0) { if (slot == 0) issue_a_pieces(kt, stage, 0, 4); }
    else if (SPREAD == 1) { if (slot == 0) issue_a_pieces(kt, stage, 0, 2); else if (slot == 1) issue_a_pieces(kt, stage, 2, 3); else if (slot == 2) issue_a_pieces(kt, stage, 3, 4); }
    else { issue_a_pieces(kt, stage, slot, slot + 1); }
  };

  // ---- weight tile: wave wv owns slabs 2 wv, 2 wv + 1 of the 256-row tile ----
  const int nslab = N >> 4;
  u32 b_off[2], sz_off[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int sl = min((n0 >> 4) + 2 * wv + s, min(nslab, n_end >> 4) - 1);
    b_off[s] = (u32)sl * nit * (BITS == 4 ? 256 : 192) + lane * (BITS == 4 ? 4 : 3);
    sz_off[s] = (u32)sl * nit * 16 + i;
  }
  const int nl = 32 * wv + i;  // tile row of slab 0's lane row; slab 1 = + 16
  using vec8 = typename DT::vec8;
  Cdna4DequantT<DT> cd;
  cd.init(lane, BITS == 4 ? 0x000F000Fu : 0x00070007u);

  auto load_group = [&](int grp) {
    Raw r;
    const u32* qg = qw + (size_t)grp * (BITS == 4 ? 256 : 192);
    const u32* sg = szp + (size_t)grp * 16;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (BITS == 4) {
        r.w[s] = *reinterpret_cast<const u32x4*>(qg + b_off[s]);
      } else {  // w3c tile: three words per lane, the fourth is rebuilt in prep()
        typedef u32 u32x3 __attribute__((ext_vector_type(3)));
        const u32x3 w3 = *reinterpret_cast<const u32x3*>(qg + b_off[s]);
        r.w[s] = u32x4{w3.x, w3.y, w3.z, 0u};
      }
      r.sz[s] = sg[sz_off[s]];
    }
    return r;
  };
  auto prep = [&](const Raw& r) {
    Group gq;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      gq.w[s] = BITS == 4 ? r.w[s] : w3_expand(r.w[s].x, r.w[s].y, r.w[s].z);
      const u32 sd = (r.sz[s] & 0xFFFFu) * 0x00010001u;
      gq.b01[s] = sd & cd.m01;
      gq.b23[s] = sd & cd.m23;
      gq.c[s] = DT::dq_offset(r.sz[s]);
    }
    return gq;
  };

  // ---- LDS byte addresses (the dynamic segment starts at LDS offset of `smem`) ----
  const u32 lds0 = (u32)(size_t)(__attribute__((address_space(3))) char*)smem;
  u32 xa[4], wa[4], ja[2][2];  // per k-step fragment addresses (stage 0, fragment 0); job destinations [word parity][slab]
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    xa[ks] = lds0 + tile_off(wm * 128 + l32, 2 * ks + hk);
    wa[ks] = lds0 + kWBase + tile_off(wn * WN + l32, 2 * ks + hk);
  }
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) ja[b][sl] = lds0 + kWBase + tile_off(nl + 16 * sl, 4 * b + g);  // (row + 16 flips the swizzle)

  u32x4 w0, w1, x0, x1, x2, x3;  // fragments (single set)
  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  auto mf = [](const u32x4& a, const u32x4& b, const f32x16& c) {
    return DT::mfma32(__builtin_bit_cast(vec8, a), __builtin_bit_cast(vec8, b), c);
  };

  Group gc;
  // one k-step (header comment).  SN / KN: stage and k-step whose fragments are read for the next step; RD: read them;
  // BAR: the block barrier of the K-tile sits behind cluster A; job: the weight word produced in this step.
  auto step = [&](auto sn_, auto kn_, auto rd_, auto bar_, auto job_) {
    constexpr int SN = decltype(sn_)::value, KN = decltype(kn_)::value;
    constexpr bool RD = decltype(rd_)::value, BAR = decltype(bar_)::value;
    using J = decltype(job_);
    // cluster A behind its wait ladder; the fences keep each MFMA between "its" wait and the next one (without them
    // hipcc bunches the four waits in front of A1, i.e. waits for x3' -- the read issued last -- before any MFMA)
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(w0), "+v"(x0));
    acc[0][0] = mf(w0, x0, acc[0][0]);
    V4_FENCE();
    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(x1));
    acc[0][1] = mf(w0, x1, acc[0][1]);
    V4_FENCE();
    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(x2));
    acc[0][2] = mf(w0, x2, acc[0][2]);
    V4_FENCE();
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(x3));
    acc[0][3] = mf(w0, x3, acc[0][3]);
    typename Cdna4DequantT<DT>::Pending pj;
    constexpr bool kJob = J::has && PROBE != 5 && PROBE != 7 && PROBE != 8;
    if constexpr (J::has && PROBE == 7) V4_WRITE(ja[J::j & 1][J::j >> 1], x3, J::st * kTileW);  // the LDS write without the arithmetic
    if constexpr (kJob) {  // the two dequant MFMAs of this step's weight word queue behind A4
      constexpr int widx = 2 * J::h + (J::j & 1), s = J::j >> 1;
      const u32 word = widx == 0 ? gc.w[s].x : (widx == 1 ? gc.w[s].y : (widx == 2 ? gc.w[s].z : gc.w[s].w));
      pj = cd.word_issue(word, gc.b01[s], gc.b23[s], gc.c[s]);
    }
    if constexpr (BAR) {
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w1));
      if constexpr (PROBE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else __syncthreads();  // + vmcnt(0): the next tile's x DMA and packed words have landed; every read of the other stage retired
    }
    V4_FENCE();
    if constexpr (RD) V4_READ(w0, wa[KN], SN * kTileW);
    if constexpr (!BAR) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(w1) : "n"(RD ? 1 : 0));
    V4_FENCE();
    acc[1][0] = mf(w1, x0, acc[1][0]);
    if constexpr (kJob) {
      const vec8 v = Cdna4DequantT<DT>::word_finish(pj);
      if constexpr (PROBE == 6) asm volatile("" : : "v"(v));  // dequant arithmetic without the LDS write
      else V4_WRITE(ja[J::j & 1][J::j >> 1], __builtin_bit_cast(u32x4, v), J::st * kTileW);
    }
    V4_FENCE();
    if constexpr (RD) V4_READ(x0, xa[KN], SN * kTileX);
    acc[1][1] = mf(w1, x1, acc[1][1]);
    V4_FENCE();
    if constexpr (RD) V4_READ(x1, xa[KN], SN * kTileX + 4096);
    acc[1][2] = mf(w1, x2, acc[1][2]);
    V4_FENCE();
    if constexpr (RD) V4_READ(x2, xa[KN], SN * kTileX + 8192);
    acc[1][3] = mf(w1, x3, acc[1][3]);
    V4_FENCE();
    if constexpr (RD) {
      V4_READ(x3, xa[KN], SN * kTileX + 12288);
      V4_READ(w1, wa[KN], SN * kTileW + 4096);
    }
    V4_FENCE();
  };

  auto kstep = [&](auto, auto sn_, auto kn_, auto rd_, auto bar_, auto job_) { step(sn_, kn_, rd_, bar_, job_); };

  // ---------------- prologue: tile 0 complete in stage 0, tile 1's x tile in flight, its first weight word written ----
  issue_a(0, 0);
  gc = prep(load_group(0));
  Raw rn = load_group(nit > 1 ? 1 : 0);
  {
    char* Bs = smem + kWBase;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = j >> 1;
      const u32 word = (j & 1) ? gc.w[s].y : gc.w[s].x;
      *reinterpret_cast<vec8*>(Bs + tile_off(nl + 16 * s, 4 * (j & 1) + g)) = cd.word(word, gc.b01[s], gc.b23[s], gc.c[s]);
    }
  }
  __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and the ds_writes
#pragma unroll
  for (int s = 0; s < 2; ++s) asm volatile("" : "+v"(rn.w[s]), "+v"(rn.sz[s]));  // as v3: no pending ordinary load enters the loop
  issue_a_slot(1, 1, 0);
  V4_FENCE();
  // first word of tile 1 = (group 0, half 1) -> stage 1, then the fragments of k-step 0 in the order the ladder expects
  {
    const vec8 v = cd.word(gc.w[0].z, gc.b01[0], gc.b23[0], gc.c[0]);
    V4_WRITE(ja[0][0], __builtin_bit_cast(u32x4, v), kTileW);
  }
  V4_READ(w0, wa[0], 0);
  V4_READ(x0, xa[0], 0);
  V4_READ(x1, xa[0], 4096);
  V4_READ(x2, xa[0], 8192);
  V4_READ(x3, xa[0], 12288);
  V4_READ(w1, wa[0], 4096);
  V4_FENCE();

  // One iteration = one quantisation group = two K-tiles (2q in stage 0, 2q + 1 in stage 1); the last group is peeled.
  auto group_iter = [&](int q, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value;
    using T = bc<true>;
    using F = bc<false>;
    // ---------- K-tile 2q (stage 0); writes words 1..3 of tile 2q+1 = (group q, half 1) into stage 1 ----------
    kstep(ic<0>{}, ic<0>{}, ic<1>{}, T{}, F{}, JobT<1, 1, 1>{});
    issue_a_slot(2 * q + 1, 1, 1);
    V4_FENCE();
    kstep(ic<1>{}, ic<0>{}, ic<2>{}, T{}, F{}, JobT<1, 2, 1>{});
    issue_a_slot(2 * q + 1, 1, 2);
    V4_FENCE();
    kstep(ic<0>{}, ic<0>{}, ic<3>{}, T{}, F{}, JobT<1, 3, 1>{});
    issue_a_slot(2 * q + 1, 1, 3);
    V4_FENCE();
    if constexpr (more) {
      gc = prep(rn);  // group q+1: loaded one iteration ago, retired by the previous barrier's vmcnt(0)
      kstep(ic<1>{}, ic<1>{}, ic<0>{}, T{}, T{}, JobT<0, 0, 0>{});  // barrier inside; first word of tile 2q+2 -> stage 0
      rn = load_group(min(q + 2, nit - 1));
      issue_a_slot(2 * q + 2, 0, 0);
      V4_FENCE();
      kstep(ic<0>{}, ic<1>{}, ic<1>{}, T{}, F{}, JobT<0, 1, 0>{});
      issue_a_slot(2 * q + 2, 0, 1);
      V4_FENCE();
      kstep(ic<1>{}, ic<1>{}, ic<2>{}, T{}, F{}, JobT<0, 2, 0>{});
      issue_a_slot(2 * q + 2, 0, 2);
      V4_FENCE();
      kstep(ic<0>{}, ic<1>{}, ic<3>{}, T{}, F{}, JobT<0, 3, 0>{});
      issue_a_slot(2 * q + 2, 0, 3);
      V4_FENCE();
      kstep(ic<1>{}, ic<0>{}, ic<0>{}, T{}, T{}, JobT<1, 0, 1>{});  // barrier inside; first word of tile 2q+3 -> stage 1
      issue_a_slot(2 * q + 3, 1, 0);
      V4_FENCE();
    } else {
      kstep(ic<1>{}, ic<1>{}, ic<0>{}, T{}, T{}, NoJob{});
      kstep(ic<0>{}, ic<1>{}, ic<1>{}, T{}, F{}, NoJob{});
      kstep(ic<1>{}, ic<1>{}, ic<2>{}, T{}, F{}, NoJob{});
      kstep(ic<0>{}, ic<1>{}, ic<3>{}, T{}, F{}, NoJob{});
      kstep(ic<1>{}, ic<0>{}, ic<0>{}, F{}, T{}, NoJob{});
    }
  };
  for (int q = 0; q + 1 < nit; ++q) group_iter(q, std::true_type{});
  group_iter(nit - 1, std::false_type{});

  if (PROBE == 3) {
    float t = 0.f;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[a][b][r];
    if (t == 12345.678f) out[tid] = 1;
    return;
  }
  // ---------------- epilogue through LDS: acc[a][b][r] = C[n = wn*64 + a*32 + (r&3) + 8 (r>>2) + 4 hk][m = wm*128 + b*32 + l32] ----
  __syncthreads();
  char* eb = smem + wv * (128 * kEpiRow);
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        u32x2 v;
        v.x = (u32)DT::from_float(acc[a][b][4 * j + 0]) | ((u32)DT::from_float(acc[a][b][4 * j + 1]) << 16);
        v.y = (u32)DT::from_float(acc[a][b][4 * j + 2]) | ((u32)DT::from_float(acc[a][b][4 * j + 3]) << 16);
        *reinterpret_cast<u32x2*>(eb + (b * 32 + l32) * kEpiRow + (a * 32 + 8 * j + 4 * hk) * 2) = v;
      }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's own LDS writes (region is wave-private)
  __builtin_amdgcn_wave_barrier();
  constexpr int GR = WN / 8, RP = 64 / GR;
#pragma unroll
  for (int ps = 0; ps < 128 / RP; ++ps) {
    const int row = ps * RP + lane / GR, gc2 = lane % GR;
    const int m = m0 + wm * 128 + row, nn = n0 + wn * WN + gc2 * 8;
    u32x4 v = *reinterpret_cast<const u32x4*>(eb + row * kEpiRow + gc2 * 16);
    if (epi == 2) {
      // QuantLlamaMLP's interleaved pair (llm_awq_amd/fused_mlp.py): columns 16 j .. + 7 are gate rows 8 j .., + 8 .. + 15 the matching
      // up rows -- the even granule of a pair stores silu(gate) * up to out[m, N/2], the odd one has nothing to store
      if ((gc2 & 1) == 0 && nn < n_end && m >= row_lo && m < row_hi) {
        const u32x4 u = *reinterpret_cast<const u32x4*>(eb + row * kEpiRow + (gc2 + 1) * 16);
        __builtin_nontemporal_store(silu_mul_octet<DT>(v, u), reinterpret_cast<u32x4*>(out + (size_t)m * (N >> 1) + (nn >> 1)));
      }
      continue;
    }
    if (nn < n_end && m >= row_lo && m < row_hi) {
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

// dense: XCD-aware, two-row-band tile order as v3 (awq_gemm_v3.hip); the last row tile is shifted up to end at row M - 1
template <typename DT, int PROBE, int BITS = 4>
__global__ __launch_bounds__(512) void gemm_cdna4_v4_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                            const u32* __restrict__ szp, const uint16_t* __restrict__ bias,
                                                            uint16_t* __restrict__ out, int M, int N, int K, int tiles_m,
                                                            int tiles_n, int n_begin, int n_end, int epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = tiles_m * tiles_n;
  int tile;
  {
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
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
  v4_tile<DT, PROBE, BITS>(smem, x, qw, szp, bias, out, N, K, min(tm * TM, M - TM), n_begin + tn * TN, n_end, 0, M, epi);
}

// grouped (MoE): expert e owns rows [offsets[e], offsets[e+1]) of the sorted x / out and the e-th slice of the stacked
// cdna4 weights / packed scales.  The grid is an upper bound (total / 256 + experts row tiles); a block finds its
// (expert, row tile) by walking the offsets and exits if there is none.  A tile always reads 256 existing rows of x
// (shifted up at the end of the buffer); rows of other experts inside it are computed and not stored.
template <typename DT>
__global__ __launch_bounds__(512) void moe_gemm_cdna4_v4_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                                const u32* __restrict__ szp, const int* __restrict__ offsets,
                                                                uint16_t* __restrict__ out, int total, int experts, int N,
                                                                int K, int row_tiles, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = row_tiles * tiles_n;
  int tile;
  {
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int q = T >> 3, r = T & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int rt = tile / tiles_n;
  const int tn = tile - rt * tiles_n;
  int e = 0, lo = 0, hi = 0;
  for (; e < experts; ++e) {
    lo = offsets[e];
    hi = offsets[e + 1];
    const int cnt = (hi - lo + TM - 1) / TM;
    if (rt < cnt) break;
    rt -= cnt;
  }
  if (e == experts) return;  // wave-uniform: no tile for this block
  const int r_lo = lo + rt * TM, r_hi = min(r_lo + TM, hi);
  const size_t ew = (size_t)(N >> 4) * (K >> 7);  // tiles per expert
  v4_tile<DT, 0>(smem, x, qw + (size_t)e * ew * 256, szp + (size_t)e * ew * 16, nullptr, out, N, K, min(r_lo, total - TM), tn * TN, N,
             r_lo, r_hi);
}

namespace {
int g_v4_probe = 0;
}
// weight rows [n_begin, n_end) of the matrix with 256 x 256 tiles (m >= 256); same contract as v3's launch_v3<2>
void launch_gemm_cdna4_v4(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                          int n_begin, int n_end, int dtype, hipStream_t st, int bits, int epi) {
  constexpr int smem_main = 2 * kTileX + 2 * kTileW;
  constexpr int smem_epi = 8 * 128 * (2 * WN + 16);
  constexpr int smem = smem_main > smem_epi ? smem_main : smem_epi;
  const int tiles_m = (m + TM - 1) / TM, tiles_n = (n_end - n_begin + TN - 1) / TN;
  using Kern = void (*)(const uint16_t*, const u32*, const u32*, const uint16_t*, uint16_t*, int, int, int, int, int, int, int, int);
#ifdef AWQ_ENABLE_PROBES
  constexpr int NP = 11;
#define V4K(T, P) gemm_cdna4_v4_kernel<T, P>
  static const Kern kerns[2][NP] = {
      {V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0)},
      {V4K(BF16, 0), V4K(BF16, 1), V4K(BF16, 2), V4K(BF16, 3), V4K(BF16, 4), V4K(BF16, 5), V4K(BF16, 6), V4K(BF16, 7), V4K(BF16, 8), V4K(BF16, 9), V4K(BF16, 10)}};
#else  // a default build has no knob that changes results
  constexpr int NP = 11;
#define V4K(T, P) gemm_cdna4_v4_kernel<T, 0>
  static const Kern kerns[2][NP] = {
      {V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0), V4K(F16, 0)},
      {V4K(BF16, 0), V4K(BF16, 0), V4K(BF16, 0), V4K(BF16, 0), V4K(BF16, 0), V4K(BF16, 0), V4K(BF16, 0), V4K(BF16, 0), V4K(BF16, 0)}};
#endif
#undef V4K
  static const Kern kerns3[2] = {gemm_cdna4_v4_kernel<F16, 0, 3>, gemm_cdna4_v4_kernel<BF16, 0, 3>};  // w3c tiles
  const int pi = g_v4_probe >= 0 && g_v4_probe < NP ? g_v4_probe : 0;
  const Kern kern = bits == 3 ? kerns3[dtype == 0 ? 0 : 1] : kerns[dtype == 0 ? 0 : 1][pi];
  static LdsOptIn optin[2][NP + 1];  // per (kernel, device)
  optin[dtype == 0 ? 0 : 1][bits == 3 ? NP : pi].ensure(reinterpret_cast<const void*>(kern), smem);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), smem, st, (const uint16_t*)x, (const u32*)qw,
                     (const u32*)szp, (const uint16_t*)bias, (uint16_t*)out, m, n, k, tiles_m, tiles_n, n_begin, n_end, epi);
}
void gemm_v4_set_probe(int v) { g_v4_probe = v; }

// grouped GEMM over sorted tokens, 256 x 256 tiles; needs total >= 256.  Returns -1 if unsupported.
int launch_moe_gemm_cdna4_v4(const void* x, const void* qw, const void* szp, const void* offsets, void* out, int total, int experts,
                             int n, int k, int dtype, hipStream_t st) {
  if (total < TM || experts < 1 || (n % 16) != 0 || (k % 128) != 0 || (size_t)total * (size_t)k >= (1ull << 31) ||
      (size_t)n * (size_t)k / 8 >= (1ull << 31))
    return -1;
  constexpr int smem_main = 2 * kTileX + 2 * kTileW;
  constexpr int smem_epi = 8 * 128 * (2 * WN + 16);
  constexpr int smem = smem_main > smem_epi ? smem_main : smem_epi;
  static LdsOptIn optin[2];  // per (kernel, device)
  optin[0].ensure(reinterpret_cast<const void*>(moe_gemm_cdna4_v4_kernel<F16>), smem);
  optin[1].ensure(reinterpret_cast<const void*>(moe_gemm_cdna4_v4_kernel<BF16>), smem);
  const int row_tiles = total / TM + experts, tiles_n = (n + TN - 1) / TN;
  auto mkern = dtype == 0 ? moe_gemm_cdna4_v4_kernel<F16> : moe_gemm_cdna4_v4_kernel<BF16>;
  hipLaunchKernelGGL(mkern, dim3(row_tiles * tiles_n), dim3(512), smem, st, (const uint16_t*)x, (const u32*)qw,
                     (const u32*)szp, (const int*)offsets, (uint16_t*)out, total, experts, n, k, row_tiles, tiles_n);
  return 0;
}

}  // namespace awq
