// Prefill GEMM v6 on cdna4-interleaved weights (bf16 and fp16, gfx950): one wave per SIMD, 256 x 64 per wave.
// Replaces gemm_w4a16_T1 / gemm_w4a16_T2 (reference awq/kernels/csrc/quantization_new/gemm/gemm_cuda.cu:312-1124) behind
// gemm_forward_cuda_new / WQLinear.forward for the 256-wide tiles of prompts with >= 256 rows.
//
// Why (profiles/r02_gemm_v4_probes.txt): the 8-wave kernels are bound by LDS occupancy -- per 16-k step v4 holds the LDS 192 cycles
// with fragment reads, 104 with the writes of the dequantised weight tile and ~115 with the x tile's LDS-DMA (411 of the 512 cycles
// the product MFMAs need), v5 256 + 115.  LDS bytes per flop fall with the number of output COLUMNS a wave owns (an x fragment is
// read once per column strip), and weights that stay in the registers of the wave that dequantised them cost no LDS at all: a wave
// that owns 64 columns for all 256 rows of the block reads 16 x fragments per 64 MFMAs (128 cycles per 16-k step) and the block needs
// no weight traffic -- but its accumulators are 256 registers, i.e. ONE wave per SIMD.  So this kernel is a 256-thread block whose
// four waves are software pipelined by hand instead of hiding each other's stalls:
//   block  = 256 rows x 256 columns, 4 waves; wave w = columns [64 w, 64 w + 64) = 4 weight slabs, all 256 rows
//   K tile = 128 (one quantisation group); x tile 256 x 128 bf16 = 64 KiB per LDS stage, two stages
//   x path = global -> registers -> ds_write_b128 (NOT LDS-DMA: a DMA instruction holds the issuing wave ~60-100 cycles, and here no
//            second wave fills the matrix pipe meanwhile); a wave stages the 64 rows it owns: 16 pieces of 4 rows per K tile, each loaded
//            two 32-k steps (~2 k cycles) before it is written (32 staging VGPRs), one piece per quarter-step
//   weights = 4 x (16 B + one scale dword) per lane per K tile straight into registers one tile ahead; dequantised on the matrix core
//            (Cdna4DequantT) into the A operand of v_mfma_f32_16x16x32 one 32-k step ahead of its use
//   per 32-k step and wave: 16 x fragments (two sets of 4, each read while the other feeds 16 MFMAs), 64 product MFMAs (1024 cycles)
//   block barrier: once per K tile, inside the last quarter-step (the fragments of the next tile's first quarter are read behind it
//            while 12 MFMAs still run); by then every read of the current stage has returned, so the stage can be overwritten
// All LDS operations of the K loop are inline asm in a fixed order (hipcc neither counts nor moves them); every quarter-step starts
// with lgkmcnt(0): the reads of its set were issued behind the first MFMA group of the previous quarter-step (~240 cycles earlier).
// Register budget (one wave per SIMD: 256 VGPRs + 256 AGPRs): accumulators 256 AGPRs; fragments 32, operands 32, weights of two
// groups 40, staging 32 VGPRs.
// Numerics: products and fp32 accumulation order along K are v5's (= v4's up to the association inside one 32-k MFMA).
#include <atomic>
#include <type_traits>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

// blocks of gemm_cdna4_v6_pair_kernel that gave up waiting for their partner since the library was loaded, on this device (their outputs are NaN): a library-owned
// counter beside the per-pair sticky word in the caller's workspace, which nobody can poll once the workspace is gone.  Host: gemm_v6_pair_lost().
__device__ unsigned int g_v6_pair_lost;

namespace {
constexpr int V6_TM = 256, V6_TN = 256, V6_TK = 128;
constexpr int kV6Stage = V6_TM * V6_TK * 2;  // 64 KiB
constexpr int kV6Pitch = 2 * V6_TN + 16;     // epilogue staging: bytes per output row (+16: consecutive rows start 4 banks apart); all widths
template <int V>
using ic6 = std::integral_constant<int, V>;
}  // namespace

#define V6_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define V6_WR(addr, val, off) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(val), "n"(off) : "memory")
#define V6_FENCE() __builtin_amdgcn_sched_barrier(0)
// product MFMA with the accumulator pinned in AGPRs: with 256 accumulator registers next to ~200 live VGPRs hipcc's allocator otherwise
// rotates accumulators between the two files every iteration (1580 v_accvgpr moves per K tile).  Opaque to the hazard recogniser:
// operands come from ds_read + s_waitcnt or from VALU results several instructions back, and an accumulator is reused 64 MFMAs later.
// the dequant MFMA in its VGPR form, as asm: once a function holds AGPR-constrained values hipcc selects the AGPR form for every builtin
// MFMA and would evict accumulators to make room for the 4x4x4 operands.  Early-clobber output (no partial overlap with C); the s_nop
// covers the VALU -> MFMA operand hazard the opaque statement hides from the hazard recogniser; results are read >= 8 MFMAs later.
template <typename DT, bool NOP = true, bool F16FORM = false>
__device__ __forceinline__ f32x4 v6_mfma4(const u32x2& a, const u32x2& b, const f32x4& c) {
  f32x4 d;
  if constexpr (NOP) {
    if constexpr (DT::id == 1 && !F16FORM) asm volatile("s_nop 3\n\tv_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    else asm volatile("s_nop 3\n\tv_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  } else {  // K loop (the nops stay: hipcc materialises the C splat with v_mov copies right in front of the statement, and a VALU write
            // followed directly by the MFMA's read of that register returned the stale value -- found in the 32x32x16 experiment, tools/EXPERIMENTS.md: awq_gemm_v7_32x32_rowswap)
    if constexpr (DT::id == 1 && !F16FORM) asm volatile("s_nop 3\n\tv_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    else asm volatile("s_nop 3\n\tv_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  }
  return d;
}
template <typename DT, typename V8>
__device__ __forceinline__ void v6_mfma(f32x4& acc, const V8& a, const u32x4& b) {
  if constexpr (DT::id == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// PROBE (AWQ_ENABLE_PROBES builds, timing only, wrong results): 1 = no x staging (no global loads / ds_writes of x in the K loop),
// 2 = no weight dequantisation in the K loop, 3 = no fragment reads in the K loop, 4 = all three (product MFMAs + barrier only)
// DQ 1: szp is the "sz_half" side buffer and the dequant runs in its f16-mantissa form (Cdna4DequantH: one shift + four v_and_or per word
// instead of three + four, two v_perm + one v_dot2 for the operands instead of perm / and / and / dot2c; exact for layers
// awq_pack_szh_cdna4 reports exact; W4 tiles only)
// NS = weight slabs (16 columns) per wave: 4 -> 256-column blocks; 2 -> 128-column blocks (knob gemm_v6_128: against awq_gemm_v4n.hip's
// 256 x 128 tiles where 256-wide tiles would fill half the chip); 3 -> 192-column blocks (accumulators 192 AGPRs) for matrices
// whose 256-wide tile count leaves a partial round that 192-wide tiles fill (qkv of Llama-3-8B: 6144 = 32 x 192 -> 8 x 32 = 256 tiles
// at M = 2048 instead of 192)
template <typename DT, int BITS, int PROBE, int DQ, int NS, int ROLE = 0>
__device__ __forceinline__ void v6_tile(char* smem, const uint16_t* __restrict__ x, const u32* __restrict__ qw, const u32* __restrict__ szp,
                                        const uint16_t* __restrict__ bias, uint16_t* __restrict__ out, int N, int K, int m0, int n0, int n_end,
                                        int epi, int row_lo, int row_hi, int g0 = 0, int nit_range = -1,
                                        float* __restrict__ part = nullptr, u32x2* flag = nullptr, u32 token = 0) {
  // one 256 x (64 NS) output tile: rows [m0, m0 + 256) of x (all readable), weight rows [n0, ...) below n_end; rows outside
  // [row_lo, row_hi) are computed but NOT stored (grouped GEMM: they belong to another expert's segment).
  // K split over a PAIR of blocks (gemm_cdna4_v6_pair_kernel; nit_range >= 0: only the quantisation groups [g0, g0 + nit_range) are summed), SYMMETRIC since
  // round 5: ROLE 1 sums the lower half of K and finishes rows [0, 128) of the tile, ROLE 2 the upper half and rows [128, 256).  Each block sends the eight
  // accumulator fragments it does NOT finish to `part` (register order, 1 KiB per wave store: 128 KiB each way, concurrently), raises the partner's flag,
  // waits for its own, adds the partner's partials to the eight fragments it keeps and runs the epilogue on its 128 rows.  (Round 4's form -- one block
  // sends all 256 KiB, the other adds and runs the whole epilogue -- left ~17 us of serial tail behind the K loops: profiles/r04_v6_pair.txt.)
  using vec8 = typename DT::vec8;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int nit_all = K >> 7, nit = nit_range < 0 ? nit_all : nit_range;  // nit_all: tile stride of a slab; nit: this block's K tiles
  constexpr int TN = 64 * NS;  // columns per block

  // ---- x staging: piece q (0..15) of this wave = rows 64 wv + 4 q .. + 3, one 16-byte granule per lane ----
  // LDS layout of a stage: row r (256 B = 16 granules of 8 k) stores logical granule p at slot p ^ (r & 15)
  const int r4 = lane >> 4, p16 = lane & 15;
  const u32 lds0 = (u32)(size_t)(__attribute__((address_space(3))) char*)smem;
  u32 wpat[4];  // LDS byte offset of this lane's slot inside piece q, q & 3 = c (rows 4 c + r4 of a 16-row group)
#pragma unroll
  for (int c = 0; c < 4; ++c) wpat[c] = (u32)(64 * wv) * 256u + (u32)r4 * 256u + (u32)((p16 ^ (4 * c + r4)) << 4);
  // global side: wave-uniform base (SGPRs) + one lane offset: global_load_dwordx4 v, v_off, s[base]
  const uint16_t* xw = x + (size_t)(m0 + 64 * wv) * (size_t)K;
  const u32 xlane_b = ((u32)r4 * (u32)K + (u32)p16 * 8u) * 2u;  // bytes, < 2^32: (SGPR base + 32-bit VGPR offset) is one instruction
  auto load_piece = [&](int kt, int q) {
    const uint16_t* base = xw + (size_t)(g0 + kt) * V6_TK + (size_t)(4 * q) * (size_t)K;
    return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base) + xlane_b);
  };

  // ---- weights: slabs 4 wv .. 4 wv + 3 of the block's 16 ----
  const int nslab = N >> 4, slab_end = min(nslab, n_end >> 4);
  constexpr int kTileWords = BITS == 4 ? 256 : 192, kLaneWords = BITS == 4 ? 4 : 3;
  u32 w_off[NS], s_off[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int slc = min((n0 >> 4) + NS * wv + s, slab_end - 1);
    w_off[s] = ((u32)slc * (u32)nit_all + (u32)g0) * kTileWords + lane * kLaneWords;
    s_off[s] = ((u32)slc * (u32)nit_all + (u32)g0) * 16 + i;
  }
  struct WG {
    u32x4 w[NS];
    u32 sz[NS];
  };
  auto load_w = [&](int grp) {
    WG r;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const u32* wp = qw + (size_t)grp * kTileWords + w_off[s];
      if (BITS == 4) {
        r.w[s] = *reinterpret_cast<const u32x4*>(wp);
      } else {
        typedef u32 u32x3 __attribute__((ext_vector_type(3)));
        const u32x3 w3 = *reinterpret_cast<const u32x3*>(wp);
        r.w[s] = w3_expand(w3.x, w3.y, w3.z);
      }
      r.sz[s] = szp[(size_t)grp * 16 + s_off[s]];
    }
    return r;
  };
  Cdna4DequantT<DT> cd;
  cd.init(lane, BITS == 4 ? 0x000F000Fu : 0x00070007u);
  struct DP {
    u32 b01, b23;
    float cv;
  };
  Cdna4DequantH<DT> ch;
  if (DQ == 1) ch.init(lane);
  auto params = [&](u32 sz) {
    DP d;
    if (DQ == 1) {
      d.b01 = __builtin_amdgcn_perm(sz, sz, ch.sel01);
      d.b23 = __builtin_amdgcn_perm(sz, sz, ch.sel23);
      d.cv = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, sz), __builtin_bit_cast(f16x2, ch.kDotC), 0.0f, false);
    } else {
      const u32 sdup = __builtin_amdgcn_perm(sz, sz, 0x01000100u);
      d.b01 = sdup & cd.m01;
      d.b23 = sdup & cd.m23;
      d.cv = DT::dq_offset(sz);
    }
    return d;
  };

  struct Pend {
    f32x4 d0, d1;
  };
  auto word_issue = [&](u32 w, const DP& d) {  // Cdna4DequantT / Cdna4DequantH ::word_issue with the asm MFMA (prologue)
    u32x2 a0, a1;
    if (DQ == 1) {
      const u32 w8 = w >> 8;
      a0 = u32x2{(w & ch.kMaskLo) | ch.kMagic, (w & ch.kMaskHi) | ch.kMagic};
      a1 = u32x2{(w8 & ch.kMaskLo) | ch.kMagic, (w8 & ch.kMaskHi) | ch.kMagic};
    } else {
      a0 = u32x2{(w & cd.kMask) | cd.kMagic, ((w >> 4) & cd.kMask) | cd.kMagic};
      a1 = u32x2{((w >> 8) & cd.kMask) | cd.kMagic, ((w >> 12) & cd.kMask) | cd.kMagic};
    }
    const u32x2 b = {d.b01, d.b23};
    const f32x4 c = {d.cv, d.cv, d.cv, d.cv};
    Pend p;
    p.d0 = v6_mfma4<DT, true, DQ == 1>(a0, b, c);
    p.d1 = v6_mfma4<DT, true, DQ == 1>(a1, b, c);
    return p;
  };

  // ---- x fragment addresses: fragment f (rows 16 f + i), 32-k step a: logical granule 4 a + g; + f * 4096 as immediate ----
  u32 xa[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) xa[a] = lds0 + i * 256 + (((4 * a + g) ^ i) << 4);

  // accumulators: every definition and every use inside the K loop is an asm statement with an AGPR constraint, so the allocator
  // has no choice of file for them (zeroed by an MFMA of zero operands with the inline constant 0 as C: 64 instructions, once)
  f32x4 acc[16][NS];
  {
    u32x4 zero = {0u, 0u, 0u, 0u};
    asm volatile("" : "+v"(zero));
#pragma unroll
    for (int f = 0; f < 16; ++f)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if constexpr (DT::id == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %1, 0" : "=a"(acc[f][s]) : "v"(zero));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, 0" : "=a"(acc[f][s]) : "v"(zero));
      }
  }

  // ---------------- prologue: x tile 0 in stage 0, x tile 1 in the staging registers, weights of group 0, operands of step 0 ----
  // staging registers: stg[0..3] carry the even quarters (pieces 0-3, 8-11) of an x tile, stg[4..7] the odd ones; a quarter is loaded two
  // 32-k steps (~2 k cycles) before it is written to LDS
  // every global load of the prologue goes out before the first wait: weights of group 0, x tile 0, the first half of x tile 1
  u32x4 stg[8];
  WG cur = load_w(0);
  {
    u32x4 t0[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) t0[q] = load_piece(0, q);
#pragma unroll
    for (int q = 0; q < 8; ++q) stg[q] = load_piece(nit > 1 ? 1 : 0, q);
#pragma unroll
    for (int q = 0; q < 16; ++q) V6_WR(lds0 + wpat[q & 3], t0[q], q * 1024);
  }
  vec8 op[2][NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const DP d0 = params(cur.sz[s]);
    Pend p0 = word_issue(cur.w[s].x, d0);
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(p0.d0), "+v"(p0.d1) : : "memory");  // (prologue only: let the dequant MFMAs retire; tied to the results so the rounding stays behind it)
    op[0][s] = DT::pack8(p0.d0, p0.d1);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
  __builtin_amdgcn_s_barrier();
  u32x4 xf[2][4];
  V6_RD(xf[0][0], xa[0], 0 * 4096);
  V6_RD(xf[0][1], xa[0], 1 * 4096);
  V6_RD(xf[0][2], xa[0], 2 * 4096);
  V6_RD(xf[0][3], xa[0], 3 * 4096);

  for (int t = 0; t < nit; ++t) {
    const u32 sbase = (u32)(t & 1) * (u32)kV6Stage, obase = sbase ^ (u32)kV6Stage;
    WG nxt = load_w(min(t + 1, nit - 1));
    const int kt1 = min(t + 1, nit - 1), kt2 = min(t + 2, nit - 1);

    // one quarter-step: 16 product MFMAs on the four fragments 4 Q .. 4 Q + 3 of 32-k step A (set Q & 1).  A single wave issues
    // everything, and the matrix pipe takes a new 16x16x32 every 16 cycles = one MFMA + at most three other instructions: the side
    // work of a quarter-step (four fragment reads of the next quarter, ONE dequantised word of step A + 1 -- slab Q --, one staged
    // piece of x tile t + 1 written and its registers reloaded) is spread over the 16 MFMA slots by hand, a scheduling fence per slot.
    auto quarter = [&](auto a_, auto q_) {
      constexpr int A = decltype(a_)::value, Q = decltype(q_)::value;
      constexpr bool kLast = A == 3 && Q == 3;
      constexpr int AN = Q < 3 ? A : ((A + 1) & 3), QN = (Q + 1) & 3, S = Q & 1, SN = S ^ 1;
      constexpr int RS = kLast ? 2 : 0;  // slot of the first fragment read (the last quarter of a tile reads behind the barrier)
      const u32 raddr = xa[AN] + (kLast ? obase : sbase);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[S][0]), "+v"(xf[S][1]), "+v"(xf[S][2]), "+v"(xf[S][3]) : : "memory");
      // dequant state of this quarter's word
      u32 word = 0, sz = 0;
      constexpr bool kDeq = PROBE != 2 && PROBE != 4 && Q < NS;  // (NS = 3: the fourth quarter of a step has no slab to dequantise)
      if constexpr (kDeq) {
        if (A == 3) {
          sz = nxt.sz[Q];
          word = nxt.w[Q].x;
        } else {
          sz = cur.sz[Q];
          word = A == 0 ? cur.w[Q].y : (A == 1 ? cur.w[Q].z : cur.w[Q].w);
        }
      }
      u32x2 a0, a1, bq;
      f32x4 cq;
      Pend pj;
      auto stage_piece = [&](auto q_c, auto part) {  // part 0: write piece q of tile t + 1; part 1: reload its registers
        constexpr int q = decltype(q_c)::value, r = 4 * ((q >> 2) & 1) + (q & 3), q2 = (q + 8) & 15;
        if (decltype(part)::value == 0) {
          const u32 waddr = lds0 + obase + wpat[q & 3];
          V6_WR(waddr, stg[r], q * 1024);
        } else {
          stg[r] = load_piece(q < 8 ? kt1 : kt2, q2);
        }
      };
      auto slot = [&](auto k_) {
        constexpr int k = decltype(k_)::value, j = k >> 2, s = k & 3;
        if constexpr (s < NS) v6_mfma<DT>(acc[4 * Q + j][s], op[A & 1][s], xf[S][j]);  // (NS = 3: every fourth slot carries side work only)
        if (kLast && k == 1) {
          // every read of this stage has returned (the set in use was complete at the top), this wave's writes of tile t + 1 are done:
          // meet the other waves, then the other stage is readable and this one writable
          asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
          __builtin_amdgcn_s_barrier();
        }
        if (PROBE != 3 && PROBE != 4) {
          if (k == RS) {
            V6_RD(xf[SN][0], raddr, (4 * QN + 0) * 4096);
            V6_RD(xf[SN][1], raddr, (4 * QN + 1) * 4096);
          }
          if (k == RS + 1) {
            V6_RD(xf[SN][2], raddr, (4 * QN + 2) * 4096);
            V6_RD(xf[SN][3], raddr, (4 * QN + 3) * 4096);
          }
        }
        if constexpr (kDeq) {
          if (DQ == 1) {
            if (k == 4) a0.x = (word & ch.kMaskLo) | ch.kMagic, a0.y = (word & ch.kMaskHi) | ch.kMagic;
            if (k == 5) {
              const u32 w8 = word >> 8;
              a1.x = (w8 & ch.kMaskLo) | ch.kMagic, a1.y = (w8 & ch.kMaskHi) | ch.kMagic;
            }
            if (k == 6) bq.x = __builtin_amdgcn_perm(sz, sz, ch.sel01), bq.y = __builtin_amdgcn_perm(sz, sz, ch.sel23);
            if (k == 7) {
              const float cv = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, sz), __builtin_bit_cast(f16x2, ch.kDotC), 0.0f, false);
              cq = f32x4{cv, cv, cv, cv};
            }
          } else {
            if (k == 4) a0.x = (word & cd.kMask) | cd.kMagic, a0.y = ((word >> 4) & cd.kMask) | cd.kMagic;
            if (k == 5) a1.x = ((word >> 8) & cd.kMask) | cd.kMagic, a1.y = ((word >> 12) & cd.kMask) | cd.kMagic;
            if (k == 6) {
              const u32 sdup = __builtin_amdgcn_perm(sz, sz, 0x01000100u);
              bq.x = sdup & cd.m01;
              bq.y = sdup & cd.m23;
            }
            if (k == 7) {
              const float cv = DT::dq_offset(sz);
              cq = f32x4{cv, cv, cv, cv};
            }
          }
          if (k == 8) pj.d0 = v6_mfma4<DT, false, DQ == 1>(a0, bq, cq);
          if (k == 9) pj.d1 = v6_mfma4<DT, false, DQ == 1>(a1, bq, cq);
          if (k == 14) op[(A + 1) & 1][Q] = DT::pack8(pj.d0, pj.d1);
        }
        if (PROBE != 1 && PROBE != 4) {
          // piece 4 A + Q in slots 10 / 11 (and piece 15 next to piece 14 in quarter (3, 2): quarter (3, 3) holds the barrier)
          if (!kLast && k == 10) stage_piece(ic6<4 * A + Q>{}, ic6<0>{});
          if (!kLast && k == 11) stage_piece(ic6<4 * A + Q>{}, ic6<1>{});
          if (A == 3 && Q == 2 && k == 12) stage_piece(ic6<15>{}, ic6<0>{});
          if (A == 3 && Q == 2 && k == 13) stage_piece(ic6<15>{}, ic6<1>{});
        }
        V6_FENCE();
      };
      slot(ic6<0>{});
      slot(ic6<1>{});
      slot(ic6<2>{});
      slot(ic6<3>{});
      slot(ic6<4>{});
      slot(ic6<5>{});
      slot(ic6<6>{});
      slot(ic6<7>{});
      slot(ic6<8>{});
      slot(ic6<9>{});
      slot(ic6<10>{});
      slot(ic6<11>{});
      slot(ic6<12>{});
      slot(ic6<13>{});
      slot(ic6<14>{});
      slot(ic6<15>{});
    };
    quarter(ic6<0>{}, ic6<0>{});
    quarter(ic6<0>{}, ic6<1>{});
    quarter(ic6<0>{}, ic6<2>{});
    quarter(ic6<0>{}, ic6<3>{});
    quarter(ic6<1>{}, ic6<0>{});
    quarter(ic6<1>{}, ic6<1>{});
    quarter(ic6<1>{}, ic6<2>{});
    quarter(ic6<1>{}, ic6<3>{});
    quarter(ic6<2>{}, ic6<0>{});
    quarter(ic6<2>{}, ic6<1>{});
    quarter(ic6<2>{}, ic6<2>{});
    quarter(ic6<2>{}, ic6<3>{});
    quarter(ic6<3>{}, ic6<0>{});
    quarter(ic6<3>{}, ic6<1>{});
    quarter(ic6<3>{}, ic6<2>{});
    quarter(ic6<3>{}, ic6<3>{});
    cur = nxt;
  }

  // rows of the tile this block finishes: fragments [F0, F0 + FN) (16 rows each)
  constexpr int F0 = ROLE == 2 ? 8 : 0, FN = ROLE ? 8 : 16;
  if constexpr (ROLE != 0) {
    static_assert(NS == 4, "pair split: 256-wide blocks");
    constexpr int H = ROLE - 1, O = 1 - H, OF0 = 8 * O;
    // flag line of the pair (8 x u32x2): [h] = "the partials for the block that finishes half h are complete" {token, ~token}; [2 + h] = mailbox of that
    // block {token, its XCC id}, written at its start (gemm_cdna4_v6_pair_kernel); [7].x = sticky error word
    // Where does the partner run?  Blocks of a pair share blockIdx & 7, which has been the XCD on every launch observed -- but placement is not a
    // contract, so the partner SAYS where it is: same XCD -> plain stores (the lines stay in the shared L2: the reader's sc1 loads hit there, and the
    // hand-over never reaches the fabric or the Infinity Cache the neighbouring launches live in); anything else, including "not heard from yet" ->
    // write-through stores.  The reader's loads are sc1 either way (L2-served; correct for both producer forms).
    u32 my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(my_xcc));
    u32x2 mb;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(mb) : "v"(flag + 2 + O) : "memory");
    const bool same_xcd = (u32)__builtin_amdgcn_readfirstlane(mb.x) == token && (u32)__builtin_amdgcn_readfirstlane(mb.y) == my_xcc;
    // (the last product MFMAs are opaque asm: let them retire before their accumulators are read)
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" : : : "memory");
    {
      float* pw = part + (size_t)O * (size_t)(V6_TM * V6_TN / 2) + (size_t)wv * (8 * NS * 256) + lane * 4;
      if (same_xcd) {
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
          for (int s2 = 0; s2 < NS; ++s2) {
            const f32x4 v = acc[OF0 + f][s2];
            const float* dst = pw + (f * NS + s2) * 256;
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
          }
      } else {
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
          for (int s2 = 0; s2 < NS; ++s2) {
            const f32x4 v = acc[OF0 + f][s2];
            const float* dst = pw + (f * NS + s2) * 256;
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
          }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");  // every store acknowledged by the L2 (plain) / by memory (write-through)
    __builtin_amdgcn_s_barrier();
    if (tid == 0) {
      const u32x2 fv = {token, token ^ 0xA5A5A5A5u}, z = {0u, 0u};
      asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(flag + O), "v"(fv) : "memory");
      asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(flag + 2 + O), "v"(z) : "memory");  // the partner's mailbox is read: free for the next launch / replay
    }
    // the partner's partials for MY half.  Bounded wait: a lost partner turns into NaN outputs and the sticky error word, not a hung queue
    bool lost = false;
    for (int spins = 0;; ++spins) {
      u32x2 fv;
      asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(fv) : "v"(flag + H) : "memory");
      if ((u32)__builtin_amdgcn_readfirstlane(fv.x) == token && (u32)__builtin_amdgcn_readfirstlane(fv.y) == (token ^ 0xA5A5A5A5u)) break;
      if (spins > (1 << 22)) {
        lost = true;
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
    // My own mailbox back to zero as well: the partner read it BEFORE it stored the partials whose flag I have just seen, so nobody needs it any
    // more -- and a mailbox written by a block that started after its partner's clearing store can no longer survive into the next replay of a captured
    // launch (same token), where it would have vouched for a placement of the previous replay.
    if (tid == 0 && !lost) {
      const u32x2 z = {0u, 0u};
      asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(flag + 2 + H), "v"(z) : "memory");
    }
    const float* pr = part + (size_t)H * (size_t)(V6_TM * V6_TN / 2) + (size_t)wv * (8 * NS * 256) + lane * 4;
#pragma unroll
    for (int f0 = 0; f0 < 8; f0 += 4) {
      f32x4 pv[4 * NS];
#pragma unroll
      for (int j = 0; j < 4 * NS; ++j) {  // (unconditional loads, one wait naming every destination: the compiler does not see the asynchronous writes)
        const float* src = pr + ((f0 + j / NS) * NS + (j % NS)) * 256;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(pv[j]) : "v"(src) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]), "+v"(pv[8]), "+v"(pv[9]),
                     "+v"(pv[10]), "+v"(pv[11]), "+v"(pv[12]), "+v"(pv[13]), "+v"(pv[14]), "+v"(pv[15])
                   :
                   : "memory");
      // lower half of K + upper half of K, in that order on both sides of the pair (fp32 addition commutes: the two blocks produce what ONE block
      // adding `lower + upper` would)
#pragma unroll
      for (int j = 0; j < 4 * NS; ++j) acc[F0 + f0 + j / NS][j % NS] = acc[F0 + f0 + j / NS][j % NS] + pv[j];
    }
    if (lost) {
      const float bad = __builtin_nanf("");
#pragma unroll
      for (int f = 0; f < FN; ++f)
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) acc[F0 + f][s2] = f32x4{bad, bad, bad, bad};
      if (tid == 0) {
        flag[7].x = 1u;  // (word 14 of the pair's 64-byte flag line: sticky)
        atomicAdd(&g_v6_pair_lost, 1u);
      }
    }
  }
  if (epi == 3) {
    // K shard of a tensor-parallel row split (awq_w4a16_partial_cdna4): out is float [M, N] and takes the fp32 accumulators unrounded,
    // no bias -- the ranks' partials are summed in fp32 and rounded to T once.  Straight from the registers: a lane holds four
    // consecutive columns of one row (16 bytes), the four g-lanes of a row one 64-byte run
    float* o32 = reinterpret_cast<float*>(out);
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      const int m = m0 + 16 * f + i;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int col = 16 * NS * wv + 16 * s + 4 * g, nn = n0 + col;
        if (nn < n_end && m >= row_lo && m < row_hi) __builtin_nontemporal_store(acc[f][s], reinterpret_cast<f32x4*>(o32 + (size_t)m * N + nn));
      }
    }
    return;
  }
  // ---------------- epilogue through LDS: acc[f][s][r] = C[n = n0 + 16 NS wv + 16 s + 4 g + r][m = m0 + 16 f + i], staged row-major ----
  asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
  __builtin_amdgcn_s_barrier();  // every wave is done with the x stages (the trailing reads of the unused stage have returned)
  if (ROLE != 0 && tid == 0) {  // (every wave of the block has seen the token: the flag is free for the next launch / the next replay of a graph)
    const u32x2 z = {0u, 0u};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(flag + (ROLE - 1)), "v"(z) : "memory");
  }
  constexpr int RW = 4 * FN, R0 = 16 * F0;  // rows per wave and first row of the block's share of the tile (64 / 0; a pair block: 32 / 0 or 128)
  {
    const u32 wbase = lds0 + i * kV6Pitch + (16 * NS * wv + 4 * g) * 2;
#pragma unroll
    for (int f = F0; f < F0 + FN; ++f)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        u32x2 v;
        v.x = (u32)DT::from_float(acc[f][s][0]) | ((u32)DT::from_float(acc[f][s][1]) << 16);
        v.y = (u32)DT::from_float(acc[f][s][2]) | ((u32)DT::from_float(acc[f][s][3]) << 16);
        asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(wbase + f * (16 * kV6Pitch)), "v"(v), "n"(s * 32) : "memory");
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
  if (epi == 2) {
    // QuantLlamaMLP's interleaved pair: columns 16 j .. + 7 are gate rows, + 8 .. + 15 the matching up rows.  A lane takes one pair
    // (32 staged bytes) and stores silu(gate) * up: 16 lanes = one 256-byte output row, four rows per wave-instruction
    const int pr = lane & 15, nn = n0 + 16 * pr;
    const bool ok = nn < n_end && 16 * pr < TN;
#pragma unroll
    for (int it0 = 0; it0 < RW / 4; it0 += 4) {
      u32x4 v[4], u[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const u32 ra = lds0 + (R0 + RW * wv + 4 * (it0 + b) + (lane >> 4)) * kV6Pitch + 32 * pr;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v[b]) : "v"(ra) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(u[b]) : "v"(ra) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(u[0]), "+v"(u[1]), "+v"(u[2]), "+v"(u[3]) : : "memory");
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int m = m0 + R0 + RW * wv + 4 * (it0 + b) + (lane >> 4);
        if (ok && m >= row_lo && m < row_hi) __builtin_nontemporal_store(silu_mul_octet<DT>(v[b], u[b]), reinterpret_cast<u32x4*>(out + (size_t)m * (N >> 1) + (nn >> 1)));
      }
    }
  } else {
    const int col = (lane & 31) * 8;  // 8 columns (16 B) per lane, two rows per wave-instruction; eight instructions per wait
    const int nn = n0 + col;
    const bool ncol_ok = nn < n_end && col < TN;
    u32x4 bv = {0u, 0u, 0u, 0u};
    if (bias != nullptr && ncol_ok) bv = *reinterpret_cast<const u32x4*>(bias + nn);
#pragma unroll
    for (int it0 = 0; it0 < RW / 2; it0 += 8) {
      u32x4 v[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const u32 ra = lds0 + (R0 + RW * wv + 2 * (it0 + b) + (lane >> 5)) * kV6Pitch + col * 2;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v[b]) : "v"(ra) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int m = m0 + R0 + RW * wv + 2 * (it0 + b) + (lane >> 5);
        u32x4 o = v[b];
        if (bias != nullptr) {  // `out + self.bias` in T (qmodule.py:221)
          auto add2 = [](u32 a, u32 b2) {
            const float lo = DT::to_float((uint16_t)(a & 0xFFFFu)) + DT::to_float((uint16_t)(b2 & 0xFFFFu));
            const float hi = DT::to_float((uint16_t)(a >> 16)) + DT::to_float((uint16_t)(b2 >> 16));
            return (u32)DT::from_float(lo) | ((u32)DT::from_float(hi) << 16);
          };
          o = u32x4{add2(o.x, bv.x), add2(o.y, bv.y), add2(o.z, bv.z), add2(o.w, bv.w)};
        }
        if (ncol_ok && m >= row_lo && m < row_hi) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(out + (size_t)m * N + nn));  // streamed: keep x / weight panels in L2
      }
    }
  }
}

template <typename DT, int BITS, int PROBE = 0, int DQ = 0, int NS = 4>
__global__ __launch_bounds__(256) void gemm_cdna4_v6_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                            const u32* __restrict__ szp, const uint16_t* __restrict__ bias,
                                                            uint16_t* __restrict__ out, int M, int N, int K, int tiles_m, int tiles_n,
                                                            int n_begin, int n_end, int epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware, two-row-band tile order (as awq_gemm_v4n.hip)
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
  // the last row tile is shifted up to end at row M - 1 (M >= 256: the launcher's contract): no row index needs clamping or masking
  v6_tile<DT, BITS, PROBE, DQ, NS>(smem, x, qw, szp, bias, out, N, K, min(tm * V6_TM, M - V6_TM), n_begin + tn * (64 * NS), n_end, epi, 0, M);
}

// Grouped (per-expert) GEMM on the same tile: tokens sorted by expert, `offsets[e] .. offsets[e + 1]` = expert e's rows, stacked cdna4
// weights [E][N/16][K/128] tiles + packed scales (Mixtral w1 / w3 / w2: SURVEY.md 8(e) MoE row; the reference has no grouped kernel --
// tinychat loops over experts).  Row tile rt of expert e covers its rows [lo + 256 rt, min(.. + 256, hi)); the tile itself always reads
// 256 in-range rows of x (shifted up at the end of the token list) and the epilogue stores only the segment's rows, so a short segment
// costs one tile and never sees another expert's weights in its outputs.
template <typename DT, int DQ = 0>
__global__ __launch_bounds__(256) void moe_gemm_cdna4_v6_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                                const u32* __restrict__ szp, const int* __restrict__ offsets,
                                                                uint16_t* __restrict__ out, int total, int experts, int N, int K,
                                                                int row_tiles, int tiles_n, int epi, int tail) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Tile walk (round 4).  The launch is sized for the host's upper bound (total / 256 + experts row tiles); the REAL count depends on the device-side
  // offsets, so every block first adds it up (experts + 1 scalar loads) and the eight XCDs split the real tiles into eight contiguous chunks -- the
  // phantom blocks end up as the tail of EVERY chunk instead of being one XCD's whole share (with 21 real row tiles of a bound of 24 the old walk
  // left XCD 7 idle).  Inside a chunk the order is expert-major, then column tile, then the expert's row tiles: the two or three row tiles of an
  // expert that share a weight column tile run back to back on one XCD (the dense kernel's two-row band), so only the first of them misses its L2.
  // `tail` (round 5): an expert's last partial row tile with fewer than `tail` rows is NOT a tile of this launch -- a 256-row tile for a handful of rows is
  // almost all waste (2048 tokens x top-2 over 8 experts: 21 row tiles for 16 tiles of work) -- those rows go to the grouped skinny kernel's tail pass
  // (awq_skinny_cdna4.hip, one weight stream of that expert); 0 = every partial tile is a tile
  const int tmin = tail > 1 ? tail : 1;
  auto tiles_of = [&](int c) { return (c >> 8) + ((c & 255) >= tmin ? 1 : 0); };
  int R = 0;
  for (int e2 = 0; e2 < experts; ++e2) R += tiles_of(offsets[e2 + 1] - offsets[e2]);
  const int T = R * tiles_n;
  int tile;
  {
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int q = T >> 3, r = T & 7;
    if (idx >= q + (xcd < r ? 1 : 0)) return;  // wave-uniform: a phantom block of this XCD's chunk
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int e = 0, lo = 0, hi = 0, rt = 0, tn = 0;
  for (; e < experts; ++e) {
    lo = offsets[e];
    hi = offsets[e + 1];
    const int cnt = tiles_of(hi - lo);
    if (tile < cnt * tiles_n) {
      tn = tile / cnt;  // (row-tile major inside the expert measured equal on the plain launch and 5 % slower on the fused w1 / w3 one: profiles/r04_moe_sweep.txt)
      rt = tile - tn * cnt;
      break;
    }
    tile -= cnt * tiles_n;
  }
  if (e == experts) return;  // (cannot happen: tile < T)
  const int r_lo = lo + rt * V6_TM, r_hi = min(r_lo + V6_TM, hi);
  const size_t ew = (size_t)(N >> 4) * (K >> 7);  // tiles per expert
  // epi 2: every expert's rows are its w1 / w3 pair interleaved 8 + 8 per 16-row slab (N = 2 x ffn): out [total, N / 2] = silu(w1 x) * (w3 x)
  // (DQ 1: szp is the experts' stacked sz_half side buffer -- the f16-mantissa dequant form, as the dense launches use it)
  v6_tile<DT, 4, 0, DQ, 4>(smem, x, qw + (size_t)e * ew * 256, szp + (size_t)e * ew * 16, nullptr, out, N, K, min(r_lo, total - V6_TM), tn * V6_TN,
                           N, epi, r_lo, r_hi);
}

// K split over a PAIR of 256 x 256 blocks.  A launch whose 256-wide tiles fill at most half the chip (down_proj and o_proj of Llama-3-8B at 2048 rows: 8 x 16 =
// 128 tiles; the 128 tiles the gate/up launch leaves behind its three full rounds) used to run as 256 x 128 blocks -- two slabs per wave: twice the x-fragment
// reads and twice the x staging per MFMA, 0.35-0.42 of the MFMA peak against 0.48-0.51 for the four-slab block (profiles/r04_f_pmc_mfma_m2048.txt).  Here every tile
// is two four-slab blocks on the SAME XCD (block index mod 8), each summing half of K and finishing half of the tile's rows: they swap the fp32 accumulators of the
// rows they do not finish through the workspace (128 KiB each way; one flag per direction carrying the launch's token, reset by its reader, so a captured launch
// replays) -- the role of the reference's split_k_iters + Semaphore (gemm_cuda.cu:546-619) inside one launch, no second kernel.  Both blocks of a pair must be
// resident at once: the launcher only takes grids of at most 256 blocks on a 256-CU device (one block per CU: 160 KiB of LDS, 512 registers per lane).
template <typename DT, int BITS, int DQ>
__global__ __launch_bounds__(256) void gemm_cdna4_v6_pair_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw, const u32* __restrict__ szp,
                                                                 const uint16_t* __restrict__ bias, uint16_t* __restrict__ out, int M, int N, int K,
                                                                 int tiles_m, int tiles_n, int n_begin, int n_end, int epi, float* __restrict__ ws, u32 token) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int T = tiles_m * tiles_n, per = T >> 3;  // T % 8 == 0 (launcher)
  // partners are NEIGHBOURS in dispatch order (blocks b and b + 8: same blockIdx & 7, consecutive idx): a resident block's partner is the next block
  // its XCD is handed, so a launch that does not own the whole chip (another stream's kernel holding CUs, a CU mask) cannot fill its CUs with lower
  // halves that all wait for undispatched upper halves (round 5 dispatched every lower half first)
  const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
  const bool upper = (idx & 1) != 0;
  const int tile = xcd * per + (idx >> 1);
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
  const int nit_all = K >> 7, n_lo = (nit_all + 1) >> 1, n_up = nit_all - n_lo;
  float* part = ws + (size_t)tile * (size_t)(V6_TM * V6_TN);
  u32x2* flag = reinterpret_cast<u32x2*>(ws + (size_t)T * (size_t)(V6_TM * V6_TN)) + (size_t)tile * 8;  // one 64-byte line per pair
  if (threadIdx.x == 0) {  // this block's mailbox: where it runs (read by the partner when its K loop is done, tens of microseconds from now)
    u32 my_xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(my_xcc));
    const u32x2 mv = {token, my_xcc};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(flag + 2 + (upper ? 1 : 0)), "v"(mv) : "memory");
  }
  const int m0 = min(tm * V6_TM, M - V6_TM), n0 = n_begin + tn * V6_TN;
  if (upper) v6_tile<DT, BITS, 0, DQ, 4, 2>(smem, x, qw, szp, bias, out, N, K, m0, n0, n_end, epi, 0, M, n_lo, n_up, part, flag, token);
  else v6_tile<DT, BITS, 0, DQ, 4, 1>(smem, x, qw, szp, bias, out, N, K, m0, n0, n_end, epi, 0, M, 0, n_lo, part, flag, token);
}

namespace {
int g_v6_probe = 0;
int g_v6_pair_min_nit = 64;  // knob gemm_v6_pair_min_nit: K tiles from which a half-filled launch takes the pair split (32 = o_proj / the gate/up remainder too, K = 4096: A/B in profiles/r05_v6_pair.txt)
}
void gemm_v6_set_probe(int v) { g_v6_probe = v; }
void gemm_v6_set_pair_lead(int) {}  // (round 4's asymmetric hand-over ran the producer half `lead` K tiles short; the symmetric pair splits K evenly)
void gemm_v6_set_pair_min_nit(int v) { g_v6_pair_min_nit = v < 8 ? 8 : v; }

// Does the block-pair K split serve [m, n_cols] x K?  W4 or W3 tiles; the 256-wide tiles fill between 3/8 and 1/2 of the 256 CUs (so the pairs fill 3/4 .. all
// of it in ONE round: both blocks of every pair are resident together), whole XCD shares, and a K loop long enough to pay for the hand-over
bool gemm_v6_pair_takes(int m, int n, int k) {
  if (m < V6_TM || (n % V6_TN) != 0 || (k % 128) != 0 || (k >> 7) < g_v6_pair_min_nit) return false;
  if (device_cu_count() != 256) return false;  // the one-round co-residency of a pair (and its XCD = blockIdx & 7 placement) is the whole MI355X's
  const long tiles = (long)((m + V6_TM - 1) / V6_TM) * (n / V6_TN);
  return tiles >= 96 && tiles <= 128 && (tiles % 8) == 0 && (size_t)m * (size_t)k < (1ull << 31);
}
size_t gemm_v6_pair_workspace_bytes(int m, int n, int k) {
  if (!gemm_v6_pair_takes(m, n, k)) return 0;
  const size_t tiles = (size_t)((m + V6_TM - 1) / V6_TM) * (n / V6_TN);
  return tiles * (size_t)(V6_TM * V6_TN * 4) + tiles * 64;
}
// host: how many pair blocks of the CURRENT device ever gave up waiting (a synchronising device-to-host copy of the counter); -1 on a runtime error
int gemm_v6_pair_lost(unsigned* count) {
  unsigned v = 0;
  if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_v6_pair_lost), sizeof(v), 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (count) *count = v;
  return 0;
}

// weight rows [n_begin, n_end) of the [n, k] matrix as block pairs; epi 0 / 2 as launch_gemm_cdna4_v6; szfmt 1: szp = sz_half (W4).  Returns -1 if it does
// not serve the call (shape, workspace): the caller runs the 256 x 128 blocks
int launch_gemm_cdna4_v6_pair(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int n_begin, int n_end,
                              int dtype, void* ws, size_t ws_bytes, hipStream_t st, int bits, int epi, int szfmt) {
  const size_t need = gemm_v6_pair_workspace_bytes(m, n_end - n_begin, k);
  if (need == 0 || ws == nullptr || ws_bytes < need || (reinterpret_cast<uintptr_t>(ws) & 63) != 0 || (epi != 0 && epi != 2) || (szfmt && bits != 4)) return -1;
  constexpr int stage2 = 2 * kV6Stage, stg_epi = V6_TM * kV6Pitch;
  constexpr int smem = stage2 > stg_epi ? stage2 : stg_epi;
  const int tiles_m = (m + V6_TM - 1) / V6_TM, tiles_n = (n_end - n_begin) / V6_TN;
  static std::atomic<u32> counter{0};
  const u32 token = (counter.fetch_add(1, std::memory_order_relaxed) % 0x7FFFFFFEu) + 1u;  // never 0 (= the reset value of a flag)
  using Kern = void (*)(const uint16_t*, const u32*, const u32*, const uint16_t*, uint16_t*, int, int, int, int, int, int, int, int, float*, u32);
  static const Kern kerns[2][3] = {{gemm_cdna4_v6_pair_kernel<F16, 4, 0>, gemm_cdna4_v6_pair_kernel<F16, 3, 0>, gemm_cdna4_v6_pair_kernel<F16, 4, 1>},
                                   {gemm_cdna4_v6_pair_kernel<BF16, 4, 0>, gemm_cdna4_v6_pair_kernel<BF16, 3, 0>, gemm_cdna4_v6_pair_kernel<BF16, 4, 1>}};
  static LdsOptIn optin[2][3];
  const int a = dtype == 0 ? 0 : 1, v = bits == 3 ? 1 : (szfmt ? 2 : 0);
  const Kern kern = kerns[a][v];
  optin[a][v].ensure(reinterpret_cast<const void*>(kern), smem);
  hipLaunchKernelGGL(kern, dim3(2 * tiles_m * tiles_n), dim3(256), smem, st, (const uint16_t*)x, (const u32*)qw, (const u32*)szp, (const uint16_t*)bias,
                     (uint16_t*)out, m, n, k, tiles_m, tiles_n, n_begin, n_end, epi, (float*)ws, token);
  return 0;
}

// weight rows [n_begin, n_end) with 256 x 256 blocks; any m >= 1 (rows past m are clamped / not stored)
void launch_gemm_cdna4_v6(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int n_begin,
                          int n_end, int dtype, hipStream_t st, int bits, int epi, int szfmt, int tile_n) {
  constexpr int stage2 = 2 * kV6Stage, stg_epi = V6_TM * kV6Pitch;
  constexpr int smem = stage2 > stg_epi ? stage2 : stg_epi;
  const int tn_cols = tile_n == 192 ? 192 : (tile_n == 128 ? 128 : V6_TN);
  const int tiles_m = (m + V6_TM - 1) / V6_TM, tiles_n = (n_end - n_begin + tn_cols - 1) / tn_cols;
  using Kern = void (*)(const uint16_t*, const u32*, const u32*, const uint16_t*, uint16_t*, int, int, int, int, int, int, int, int);
  static const Kern kerns[2][2] = {{gemm_cdna4_v6_kernel<F16, 4>, gemm_cdna4_v6_kernel<F16, 3>},
                                   {gemm_cdna4_v6_kernel<BF16, 4>, gemm_cdna4_v6_kernel<BF16, 3>}};
  const int a = dtype == 0 ? 0 : 1, b = bits == 3 ? 1 : 0;
  if (tn_cols == 192 && bits == 4) {  // 192-column blocks (three slabs per wave)
    static const Kern kerns_3[4] = {gemm_cdna4_v6_kernel<F16, 4, 0, 0, 3>, gemm_cdna4_v6_kernel<BF16, 4, 0, 0, 3>,
                                    gemm_cdna4_v6_kernel<F16, 4, 0, 1, 3>, gemm_cdna4_v6_kernel<BF16, 4, 0, 1, 3>};  // [2..3]: szp = sz_half
    static LdsOptIn optin_3[4];
    const int a3 = (dtype == 0 ? 0 : 1) + (szfmt == 1 ? 2 : 0);
    optin_3[a3].ensure(reinterpret_cast<const void*>(kerns_3[a3]), smem);
    hipLaunchKernelGGL(kerns_3[a3], dim3(tiles_m * tiles_n), dim3(256), smem, st, (const uint16_t*)x, (const u32*)qw, (const u32*)szp,
                       (const uint16_t*)bias, (uint16_t*)out, m, n, k, tiles_m, tiles_n, n_begin, n_end, epi);
    return;
  }
  if (tn_cols == 128 && bits == 4) {  // 128-column blocks (two slabs per wave): o_proj / down_proj at M = 2048 fill the chip with 256 of them
    static const Kern kerns_2[4] = {gemm_cdna4_v6_kernel<F16, 4, 0, 0, 2>, gemm_cdna4_v6_kernel<BF16, 4, 0, 0, 2>,
                                    gemm_cdna4_v6_kernel<F16, 4, 0, 1, 2>, gemm_cdna4_v6_kernel<BF16, 4, 0, 1, 2>};  // [2..3]: szp = sz_half
    static LdsOptIn optin_2[4];
    const int a2 = (dtype == 0 ? 0 : 1) + (szfmt == 1 ? 2 : 0);
    optin_2[a2].ensure(reinterpret_cast<const void*>(kerns_2[a2]), smem);
    hipLaunchKernelGGL(kerns_2[a2], dim3(tiles_m * tiles_n), dim3(256), smem, st, (const uint16_t*)x, (const u32*)qw, (const u32*)szp,
                       (const uint16_t*)bias, (uint16_t*)out, m, n, k, tiles_m, tiles_n, n_begin, n_end, epi);
    return;
  }
  static const Kern kerns_h[2] = {gemm_cdna4_v6_kernel<F16, 4, 0, 1>, gemm_cdna4_v6_kernel<BF16, 4, 0, 1>};  // szp = sz_half
  Kern kern = (szfmt == 1 && bits == 4) ? kerns_h[a] : kerns[a][b];
  static LdsOptIn optin[2][2], optin_h[2];
  if (szfmt == 1 && bits == 4) {
    optin_h[a].ensure(reinterpret_cast<const void*>(kern), smem);
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), smem, st, (const uint16_t*)x, (const u32*)qw, (const u32*)szp,
                       (const uint16_t*)bias, (uint16_t*)out, m, n, k, tiles_m, tiles_n, n_begin, n_end, epi);
    return;
  }
#ifdef AWQ_ENABLE_PROBES
  static const Kern probes[5] = {gemm_cdna4_v6_kernel<BF16, 4>, gemm_cdna4_v6_kernel<BF16, 4, 1>, gemm_cdna4_v6_kernel<BF16, 4, 2>,
                                 gemm_cdna4_v6_kernel<BF16, 4, 3>, gemm_cdna4_v6_kernel<BF16, 4, 4>};
  static LdsOptIn optin_p[5];
  if (g_v6_probe > 0 && g_v6_probe < 5) {
    kern = probes[g_v6_probe];
    optin_p[g_v6_probe].ensure(reinterpret_cast<const void*>(kern), smem);
  } else
#endif
  optin[a][b].ensure(reinterpret_cast<const void*>(kern), smem);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), smem, st, (const uint16_t*)x, (const u32*)qw, (const u32*)szp,
                     (const uint16_t*)bias, (uint16_t*)out, m, n, k, tiles_m, tiles_n, n_begin, n_end, epi);
}

// grouped GEMM over sorted tokens with the v6 tile; needs total >= 256.  Returns -1 if unsupported.
namespace {
int g_moe_tail = 64;  // knob moe_tail: partial row tiles below this many rows go to the grouped skinny kernel's tail pass (0 = every partial tile is a tile)
}
void moe_v6_set_tail(int v) { g_moe_tail = v < 0 ? 0 : (v > 64 ? 64 : v); }
int launch_moe_gemm_cdna4_v6(const void* x, const void* qw, const void* szp, const void* offsets, void* out, int total, int experts,
                             int n, int k, int dtype, hipStream_t st, int epi, const void* szh) {
  if ((epi != 0 && epi != 2) || (epi == 2 && (n % 32) != 0)) return -1;
  if (total < V6_TM || experts < 1 || (n % 16) != 0 || (k % 128) != 0 || (size_t)total * (size_t)k >= (1ull << 31) ||
      (size_t)n * (size_t)k / 8 >= (1ull << 31))
    return -1;
  constexpr int stage2 = 2 * kV6Stage, stg_epi = V6_TM * kV6Pitch;
  constexpr int smem = stage2 > stg_epi ? stage2 : stg_epi;
  static LdsOptIn optin[4];
  using MKern = void (*)(const uint16_t*, const u32*, const u32*, const int*, uint16_t*, int, int, int, int, int, int, int, int);
  static const MKern kerns[4] = {moe_gemm_cdna4_v6_kernel<F16, 0>, moe_gemm_cdna4_v6_kernel<BF16, 0>, moe_gemm_cdna4_v6_kernel<F16, 1>,
                                 moe_gemm_cdna4_v6_kernel<BF16, 1>};  // [2..3]: the tile launch reads the stacked sz_half buffer
  const int ki = (dtype == 0 ? 0 : 1) + (szh != nullptr ? 2 : 0);
  const MKern kern = kerns[ki];
  optin[ki].ensure(reinterpret_cast<const void*>(kern), smem);
  const int row_tiles = total / V6_TM + experts, tiles_n = (n + V6_TN - 1) / V6_TN;
  const int tail = experts <= 64 ? g_moe_tail : 0;  // (the tail pass keeps its list of qualifying experts in 64 LDS words)
  hipLaunchKernelGGL(kern, dim3(row_tiles * tiles_n), dim3(256), smem, st, (const uint16_t*)x, (const u32*)qw, (const u32*)(szh != nullptr ? szh : szp),
                     (const int*)offsets, (uint16_t*)out, total, experts, n, k, row_tiles, tiles_n, epi, tail);
  // the rows of the partial tiles this launch left out (the launches read the same device-side offsets and split every expert's rows the same way)
  if (tail > 0 && launch_moe_skinny_tail_cdna4(x, qw, szp, offsets, out, experts, n, k, dtype, st, epi, tail) != 0) return -1;
  return 0;
}

}  // namespace awq
