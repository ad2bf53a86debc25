// EXPERIMENT (AWQ_PROBES=1 builds only; not in the product library, not yet run on hardware): the v6 prefill loop on the PLANNED 32-row
// interleave "cdna4w" (specified and pinned by tests/test_cdna4w_layout.py, DESIGN.md "Planned layout"), whose matrix-core
// dequant emits the A operand of v_mfma_f32_32x32x16 directly: half the product-MFMA issue and operand reads per flop of awq_gemm_v6.hip's
// 16x16x32 loop.  Same role as gemm_w4a16_T1 / T2 (reference awq/kernels/csrc/quantization_new/gemm/gemm_cuda.cu:312-1124).
//
// Block = 256 rows x 256 columns, 4 waves, one per SIMD; wave w owns columns [64 w, 64 w + 64) = TWO slab pairs (32 weight rows each) for
// all 256 rows: 8 row blocks x 2 pairs x 16 accumulator registers = the whole AGPR file.  K tile = 128 (one quantisation group) = 8 steps
// of 16 k; per step and wave 16 product MFMAs of 32 cycles.  The schedule unit is a HALF-step (H = 0 / 1: row blocks 4 H .. 4 H + 3 of step
// S): 8 MFMA slots of 32 cycles, i.e. the 256 cycles of a v6 quarter-step with half the MFMA issues, and the same side work spread over
// them: the four x fragments of the next unit, ONE dequantised word (pair H, step S + 1) and one staged piece of the next x tile.
//   x path: as awq_gemm_v6.hip (global -> registers -> ds_write_b128, two 64-KiB stages, slot = granule ^ (row & 15)); a fragment is
//           rows 32 f + l % 32, granule 2 S + l / 32: conflict free for the reason the 16-row read is
//   weights: per pair and group two 1-KiB tiles (64 k each), 16 B per lane each, one group ahead; sz_packed dword of row l % 32 of the pair
//   accumulators: acc[f][p][r] = C[n = n0 + 64 wv + 32 p + (r & 3) + 8 (r >> 2) + 4 (l / 32)][m = m0 + 32 f + l % 32]
#include <type_traits>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

#ifdef AWQ_ENABLE_PROBES

namespace awq {

namespace {
constexpr int W_TM = 256, W_TN = 256, W_TK = 128;
constexpr int kWStage = W_TM * W_TK * 2;  // 64 KiB
constexpr int kWPitch = 2 * W_TN + 16;    // epilogue staging: bytes per output row
template <int V>
using icw = std::integral_constant<int, V>;
typedef float f32x16 __attribute__((ext_vector_type(16)));
}  // namespace

#define W_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define W_WR(addr, val, off) asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(val), "n"(off) : "memory")
#define W_FENCE() __builtin_amdgcn_sched_barrier(0)

// dequant MFMA in its VGPR form (see awq_gemm_v6.hip::v6_mfma4: the s_nop covers the VALU -> MFMA operand hazard the opaque statement hides)
template <typename DT>
__device__ __forceinline__ f32x4 w_mfma4(const u32x2& a, const u32x2& b, const f32x4& c) {
  f32x4 d;
  if constexpr (DT::id == 1) asm volatile("s_nop 3\n\tv_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  else asm volatile("s_nop 3\n\tv_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// product MFMA 32x32x16, accumulator pinned in AGPRs (A = 32 weight rows x 16 k, B = 16 k x 32 x rows)
template <typename DT, typename V8>
__device__ __forceinline__ void w_mfma(f32x16& acc, const V8& a, const u32x4& b) {
  if constexpr (DT::id == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <typename DT>
__device__ __forceinline__ void v6w_tile(char* smem, const uint16_t* __restrict__ x, const u32* __restrict__ qw, const u32* __restrict__ szp,
                                         const uint16_t* __restrict__ bias, uint16_t* __restrict__ out, int N, int K, int m0, int n0, int n_end) {
  using vec8 = typename DT::vec8;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, kb = lane >> 5;
  const int nit = K >> 7;

  // ---- x staging (as awq_gemm_v6.hip): piece q (0..15) of this wave = rows 64 wv + 4 q .. + 3, one 16-byte granule per lane ----
  const int r4 = lane >> 4, p16 = lane & 15;
  const u32 lds0 = (u32)(size_t)(__attribute__((address_space(3))) char*)smem;
  u32 wpat[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) wpat[c] = (u32)(64 * wv) * 256u + (u32)r4 * 256u + (u32)((p16 ^ (4 * c + r4)) << 4);
  const uint16_t* xw = x + (size_t)(m0 + 64 * wv) * (size_t)K;
  const u32 xlane_b = ((u32)r4 * (u32)K + (u32)p16 * 8u) * 2u;
  auto load_piece = [&](int kt, int q) {
    const uint16_t* base = xw + (size_t)kt * W_TK + (size_t)(4 * q) * (size_t)K;
    return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base) + xlane_b);
  };

  // ---- weights: slab pairs 2 wv, 2 wv + 1 of the block's 8; pair P of the matrix = rows 32 P .. 32 P + 31 ----
  const int npair = N >> 5, pair_end = min(npair, (n_end + 31) >> 5), nslab = N >> 4;
  u32 w_off[2], s_off[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int pc = min((n0 >> 5) + 2 * wv + p, pair_end - 1);
    w_off[p] = (u32)pc * (u32)(2 * nit) * 256u + lane * 4u;                       // tiles [pair][K / 64] of 256 words
    const int slab = min(2 * pc + (l32 >> 4), nslab - 1);                          // sz_packed is [N / 16][K / 128][16]
    s_off[p] = (u32)slab * (u32)nit * 16u + (u32)(lane & 15);
  }
  struct WG {
    u32x4 w[2][2];  // [pair][64-k tile of the group]
    u32 sz[2];
  };
  auto load_w = [&](int grp) {
    WG r;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int h = 0; h < 2; ++h) r.w[p][h] = *reinterpret_cast<const u32x4*>(qw + (size_t)(2 * grp + h) * 256 + w_off[p]);
      r.sz[p] = szp[(size_t)grp * 16 + s_off[p]];
    }
    return r;
  };
  Cdna4DequantT<DT> cd;
  cd.init(lane, 0x000F000Fu);
  struct Pend {
    f32x4 d0, d1;
  };
  auto word_issue = [&](u32 w, u32 sz) {  // prologue form
    const u32x2 a0 = {(w & cd.kMask) | cd.kMagic, ((w >> 4) & cd.kMask) | cd.kMagic};
    const u32x2 a1 = {((w >> 8) & cd.kMask) | cd.kMagic, ((w >> 12) & cd.kMask) | cd.kMagic};
    const u32 sdup = __builtin_amdgcn_perm(sz, sz, 0x01000100u);
    const u32x2 b = {sdup & cd.m01, sdup & cd.m23};
    const float cv = DT::dq_offset(sz);
    const f32x4 c = {cv, cv, cv, cv};
    Pend pd;
    pd.d0 = w_mfma4<DT>(a0, b, c);
    pd.d1 = w_mfma4<DT>(a1, b, c);
    return pd;
  };

  // ---- x fragment addresses: fragment f (rows 32 f + l32), step S: logical granule 2 S + kb; + f * 8192 as immediate ----
  u32 xa[8];
#pragma unroll
  for (int S = 0; S < 8; ++S) xa[S] = lds0 + l32 * 256 + (((2 * S + kb) ^ (lane & 15)) << 4);

  f32x16 acc[8][2];
  {
    u32x4 zero = {0u, 0u, 0u, 0u};
    asm volatile("" : "+v"(zero));
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if constexpr (DT::id == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=a"(acc[f][p]) : "v"(zero));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, 0" : "=a"(acc[f][p]) : "v"(zero));
      }
  }

  // ---------------- prologue: x tile 0 in stage 0, x tile 1's first half in the staging registers, weights of group 0, operands of step 0 ----
  u32x4 stg[8];
  WG cur = load_w(0);
  {
    u32x4 t0[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) t0[q] = load_piece(0, q);
#pragma unroll
    for (int q = 0; q < 8; ++q) stg[q] = load_piece(nit > 1 ? 1 : 0, q);
#pragma unroll
    for (int q = 0; q < 16; ++q) W_WR(lds0 + wpat[q & 3], t0[q], q * 1024);
  }
  vec8 op[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    Pend p0 = word_issue(cur.w[p][0].x, cur.sz[p]);
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(p0.d0), "+v"(p0.d1) : : "memory");
    op[0][p] = DT::pack8(p0.d0, p0.d1);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
  __builtin_amdgcn_s_barrier();
  u32x4 xf[2][4];
  W_RD(xf[0][0], xa[0], 0 * 8192);
  W_RD(xf[0][1], xa[0], 1 * 8192);
  W_RD(xf[0][2], xa[0], 2 * 8192);
  W_RD(xf[0][3], xa[0], 3 * 8192);

  for (int t = 0; t < nit; ++t) {
    const u32 sbase = (u32)(t & 1) * (u32)kWStage, obase = sbase ^ (u32)kWStage;
    WG nxt = load_w(min(t + 1, nit - 1));
    const int kt1 = min(t + 1, nit - 1), kt2 = min(t + 2, nit - 1);

    // one unit = half a 16-k step: 8 product MFMAs (row blocks 4 H .. 4 H + 3 x two pairs) + the side work of 256 cycles
    auto unit = [&](auto s_, auto h_) {
      constexpr int S = decltype(s_)::value, H = decltype(h_)::value;
      constexpr int U = 2 * S + H;                      // 0 .. 15: the unit's index in the K tile = the staged piece it carries
      constexpr bool kLast = U == 15;
      constexpr int SN = H == 0 ? S : ((S + 1) & 7), HN = H ^ 1, SET = U & 1, SETN = SET ^ 1;
      constexpr int RS = kLast ? 2 : 0;                 // the last unit of a tile reads behind the barrier
      const u32 raddr = xa[SN] + (kLast ? obase : sbase);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xf[SET][0]), "+v"(xf[SET][1]), "+v"(xf[SET][2]), "+v"(xf[SET][3]) : : "memory");
      // the word this unit dequantises: pair H, step S + 1 (step 8 = step 0 of the next group)
      u32 word, sz;
      if constexpr (S == 7) {
        sz = nxt.sz[H];
        word = nxt.w[H][0].x;
      } else {
        constexpr int S1 = S + 1, TI = S1 >> 2, WI = S1 & 3;
        sz = cur.sz[H];
        word = WI == 0 ? cur.w[H][TI].x : (WI == 1 ? cur.w[H][TI].y : (WI == 2 ? cur.w[H][TI].z : cur.w[H][TI].w));
      }
      u32x2 a0, a1, bq;
      f32x4 cq;
      Pend pj;
      auto stage_piece = [&](auto q_c, auto part) {
        constexpr int q = decltype(q_c)::value, r = 4 * ((q >> 2) & 1) + (q & 3), q2 = (q + 8) & 15;
        if (decltype(part)::value == 0) {
          const u32 waddr = lds0 + obase + wpat[q & 3];
          W_WR(waddr, stg[r], q * 1024);
        } else {
          stg[r] = load_piece(q < 8 ? kt1 : kt2, q2);
        }
      };
      auto slot = [&](auto k_) {
        constexpr int k = decltype(k_)::value, j = k >> 1, p = k & 1;
        w_mfma<DT>(acc[4 * H + j][p], op[S & 1][p], xf[SET][j]);
        if (kLast && k == 1) {
          asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
          __builtin_amdgcn_s_barrier();
        }
        if (k == RS) {
          W_RD(xf[SETN][0], raddr, (4 * HN + 0) * 8192);
          W_RD(xf[SETN][1], raddr, (4 * HN + 1) * 8192);
        }
        if (k == RS + 1) {
          W_RD(xf[SETN][2], raddr, (4 * HN + 2) * 8192);
          W_RD(xf[SETN][3], raddr, (4 * HN + 3) * 8192);
        }
        if (k == 2) {
          a0.x = (word & cd.kMask) | cd.kMagic, a0.y = ((word >> 4) & cd.kMask) | cd.kMagic;
          a1.x = ((word >> 8) & cd.kMask) | cd.kMagic, a1.y = ((word >> 12) & cd.kMask) | cd.kMagic;
        }
        if (k == 3) {
          const u32 sdup = __builtin_amdgcn_perm(sz, sz, 0x01000100u);
          bq.x = sdup & cd.m01;
          bq.y = sdup & cd.m23;
          const float cv = DT::dq_offset(sz);
          cq = f32x4{cv, cv, cv, cv};
        }
        if (k == 4) {
          pj.d0 = w_mfma4<DT>(a0, bq, cq);
          pj.d1 = w_mfma4<DT>(a1, bq, cq);
        }
        if (k == 7) op[(S + 1) & 1][H] = DT::pack8(pj.d0, pj.d1);
        // piece U in slots 5 / 6 (piece 15 rides with piece 14: the last unit holds the barrier)
        if (!kLast && k == 5) stage_piece(icw<U>{}, icw<0>{});
        if (!kLast && k == 6) stage_piece(icw<U>{}, icw<1>{});
        if (U == 14 && k == 6) stage_piece(icw<15>{}, icw<0>{});
        if (U == 14 && k == 7) stage_piece(icw<15>{}, icw<1>{});
        W_FENCE();
      };
      slot(icw<0>{});
      slot(icw<1>{});
      slot(icw<2>{});
      slot(icw<3>{});
      slot(icw<4>{});
      slot(icw<5>{});
      slot(icw<6>{});
      slot(icw<7>{});
    };
    unit(icw<0>{}, icw<0>{});
    unit(icw<0>{}, icw<1>{});
    unit(icw<1>{}, icw<0>{});
    unit(icw<1>{}, icw<1>{});
    unit(icw<2>{}, icw<0>{});
    unit(icw<2>{}, icw<1>{});
    unit(icw<3>{}, icw<0>{});
    unit(icw<3>{}, icw<1>{});
    unit(icw<4>{}, icw<0>{});
    unit(icw<4>{}, icw<1>{});
    unit(icw<5>{}, icw<0>{});
    unit(icw<5>{}, icw<1>{});
    unit(icw<6>{}, icw<0>{});
    unit(icw<6>{}, icw<1>{});
    unit(icw<7>{}, icw<0>{});
    unit(icw<7>{}, icw<1>{});
    cur = nxt;
  }

  // ---------------- epilogue through LDS, staged row-major (rows = x rows m, columns = weight rows n) ----
  asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
  __builtin_amdgcn_s_barrier();
  {
    const u32 wbase = lds0 + l32 * kWPitch + (64 * wv + 4 * kb) * 2;
#pragma unroll
    for (int f = 0; f < 8; ++f)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {  // registers 4 rg .. 4 rg + 3 = four consecutive weight rows 8 rg + 4 kb + 0..3 of the pair
          u32x2 v;
          v.x = (u32)DT::from_float(acc[f][p][4 * rg + 0]) | ((u32)DT::from_float(acc[f][p][4 * rg + 1]) << 16);
          v.y = (u32)DT::from_float(acc[f][p][4 * rg + 2]) | ((u32)DT::from_float(acc[f][p][4 * rg + 3]) << 16);
          asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(wbase + f * (32 * kWPitch)), "v"(v), "n"(p * 64 + rg * 16) : "memory");
        }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
  {
    const int col = (lane & 31) * 8;  // 8 columns (16 B) per lane, two rows per wave-instruction
    const int nn = n0 + col;
    const bool ncol_ok = nn < n_end;
    u32x4 bv = {0u, 0u, 0u, 0u};
    if (bias != nullptr && ncol_ok) bv = *reinterpret_cast<const u32x4*>(bias + nn);
#pragma unroll
    for (int it0 = 0; it0 < 32; it0 += 8) {
      u32x4 v[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const u32 ra = lds0 + (64 * wv + 2 * (it0 + b) + (lane >> 5)) * kWPitch + col * 2;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v[b]) : "v"(ra) : "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int m = m0 + 64 * wv + 2 * (it0 + b) + (lane >> 5);
        u32x4 o = v[b];
        if (bias != nullptr) {  // `out + self.bias` in T (qmodule.py:221)
          auto add2 = [](u32 a, u32 b2) {
            const float lo = DT::to_float((uint16_t)(a & 0xFFFFu)) + DT::to_float((uint16_t)(b2 & 0xFFFFu));
            const float hi = DT::to_float((uint16_t)(a >> 16)) + DT::to_float((uint16_t)(b2 >> 16));
            return (u32)DT::from_float(lo) | ((u32)DT::from_float(hi) << 16);
          };
          o = u32x4{add2(o.x, bv.x), add2(o.y, bv.y), add2(o.z, bv.z), add2(o.w, bv.w)};
        }
        if (ncol_ok) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(out + (size_t)m * N + nn));
      }
    }
  }
}

template <typename DT>
__global__ __launch_bounds__(256) void gemm_cdna4w_v6_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw, const u32* __restrict__ szp,
                                                             const uint16_t* __restrict__ bias, uint16_t* __restrict__ out, int M, int N, int K,
                                                             int tiles_m, int tiles_n) {
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
  v6w_tile<DT>(smem, x, qw, szp, bias, out, N, K, min(tm * W_TM, M - W_TM), tn * W_TN, N);
}

// v2 -> cdna4w (one thread per destination u32 word, as awq_util.hip's repack_v2_to_cdna4_kernel): destination word t = tile * 256 + lane * 4 + a
// of tile (pair, kt); lane = 32 kb + 4 nq + r; nibble p (i = p & 3, hi = p >> 2) <- Q[32 pair + 4 nq + 2 (i & 1) + hi][64 kt + 16 a + 8 kb + 4 (i >> 1) + r]
__global__ void repack_v2_to_cdna4w_kernel(const u32* __restrict__ src, u32* __restrict__ dst, int N, int K) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)N * K / 8) return;
  const int a = (int)(t & 3), lane = (int)((t >> 2) & 63);
  const size_t tile = t >> 8;
  const int nt64 = K >> 6;
  const int pair = (int)(tile / nt64), kt = (int)(tile % nt64);
  const int kb = lane >> 5, nq = (lane >> 2) & 7, r = lane & 3;
  u32 w = 0;
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int i = p & 3, hi = p >> 2;
    const int n = 32 * pair + 4 * nq + 2 * (i & 1) + hi;
    const int k = 64 * kt + 16 * a + 8 * kb + 4 * (i >> 1) + r;
    // the v2 read of awq_util.hip::v2_read_nibble
    const int kl = k & 31;
    const int wa = (kl & 7) >> 1, nib = (kl >> 3) + 4 * (kl & 1);
    w |= ((src[v2_chunk_word(n, k >> 5, K) + wa] >> (4 * nib)) & 0xFu) << (4 * p);
  }
  dst[t] = w;
}

int launch_repack_v2_to_cdna4w(const void* src, void* dst, int n, int k, hipStream_t st) {
  if ((n % 32) != 0 || (k % 128) != 0) return -1;
  const size_t words = (size_t)n * k / 8;
  hipLaunchKernelGGL(repack_v2_to_cdna4w_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, (const u32*)src, (u32*)dst, n, k);
  return 0;
}

// qw: cdna4w-interleaved weights (tools/v6w_try.py packs them on the host); szp: the usual sz_packed [N / 16][K / 128][16].  m >= 256, n % 32 == 0, k % 128 == 0.
int launch_gemm_cdna4w_v6(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int dtype,
                          hipStream_t st) {
  if (m < W_TM || (n % 32) != 0 || (k % 128) != 0 || (size_t)m * (size_t)k >= (1ull << 31)) return -1;
  constexpr int stage2 = 2 * kWStage, stg_epi = W_TM * kWPitch;
  constexpr int smem = stage2 > stg_epi ? stage2 : stg_epi;
  const int tiles_m = (m + W_TM - 1) / W_TM, tiles_n = (n + W_TN - 1) / W_TN;
  static LdsOptIn optin[2];
  auto kern = dtype == 0 ? gemm_cdna4w_v6_kernel<F16> : gemm_cdna4w_v6_kernel<BF16>;
  optin[dtype == 0 ? 0 : 1].ensure(reinterpret_cast<const void*>(kern), smem);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(256), smem, st, (const uint16_t*)x, (const u32*)qw, (const u32*)szp, (const uint16_t*)bias,
                     (uint16_t*)out, m, n, k, tiles_m, tiles_n);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Decode GEMV on cdna4w weights at SLAB granularity (EXPERIMENT; a reduced copy of awq_gemv_dma.hip's body: EPI 0, sz_half dequant form).
// A 16-row slab nb is half h = nb & 1 of every tile of pair nb >> 1 (tile lanes 32 kb + 16 h + 0..15), so one 1-KiB wave load per 128-k
// step fetches the two halves of two adjacent pair tiles: lane l = 16 g + i, kb = g & 1, tsel = g >> 1 reads 16 bytes at
// tsel * 1024 + (32 kb + 16 h + i) * 16 behind the wave-uniform (pair * K / 64 + 2 step) * 1024.  The dequant then yields the 16x16x32
// operand of row i over k = 128 step + 64 tsel + 16 a + 8 kb + 0..7, and the x operand is read at the matching offsets
// (tests/test_cdna4w_layout.py::test_decode_kernels_can_read_cdna4w_at_slab_granularity).  Everything else -- K split over waves, ring,
// counted vmcnt, sz staging, split-K reduction through LDS, bias -- is the product kernel's.
// ---------------------------------------------------------------------------------------------------------------------------------
#define DMAW_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
template <int SIZE, int AUX>
__device__ __forceinline__ void dmaw_to_lds(const __amdgpu_buffer_rsrc_t& rsrc, char* dst, u32 voff, u32 soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (SIZE == 16 && AUX == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, DMAW_LDS_PTR(dst), 16, voff, soff, 0, 2);
  if constexpr (SIZE == 16 && AUX == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, DMAW_LDS_PTR(dst), 16, voff, soff, 0, 0);
  if constexpr (SIZE == 4 && AUX == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, DMAW_LDS_PTR(dst), 4, voff, soff, 0, 0);
#endif
}
template <int N_>
__device__ __forceinline__ void dmaw_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N_) : "memory");
}
template <int J, int E, class F>
__device__ __forceinline__ void dmaw_static_for(F&& f) {
  if constexpr (J < E) {
    f(std::integral_constant<int, J>{});
    dmaw_static_for<J + 1, E>(f);
  }
}

template <typename DT, int WAVES, int D>
__global__ __launch_bounds__(64 * WAVES) void gemv_dmaw_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                                const u32* __restrict__ szh, const uint16_t* __restrict__ bias,
                                                                uint16_t* __restrict__ out, int M, int N, int K, int TX) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nb = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int kbl = g & 1, tsel = g >> 1;
  const int nit = K >> 7;
  const int TXp = (TX + 3) & ~3;
  const int xrow = TXp * 256 + 16;
  const int wave_bytes = D * 1024 + TXp * 64 + M * xrow;
  char* wbase = smem + wv * wave_bytes;
  char* ring = wbase;
  char* szs = wbase + D * 1024;
  char* xs = szs + TXp * 64;
  const int s0 = wv * TX;

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(qw), 0, (N >> 4) * nit * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(szh), 0, (N >> 4) * nit * 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, M * K * 2, 0x00020000);
  const u32 pair_tile = (u32)(nb >> 1) * (u32)(2 * nit);                                 // pair tiles are [N / 32][K / 64]
  const u32 wvoff = (u32)tsel * 1024u + (u32)(32 * kbl + 16 * (nb & 1) + i) * 16u;        // this lane's 16 bytes inside the two tiles of a step
  const u32 slab_sz = (u32)nb * (u32)nit;                                                // sz_half stays [N / 16][K / 128][16]
  const u32 lane16 = lane * 16u, lane4 = lane * 4u;
  auto issue = [&](int t, int slot) {
    const u32 kg = (u32)min(s0 + t, nit - 1);
    dmaw_to_lds<16, 2>(rw, ring + slot * 1024, wvoff, (pair_tile + 2u * kg) * 1024u);
  };
  issue(0, 0);
  for (int q = 0; q < TXp; q += 4) dmaw_to_lds<4, 0>(rs, szs + q * 64, lane4, (slab_sz + (u32)(s0 + q)) * 64u);
  for (int r = 0; r < M; ++r)
    for (int q = 0; q < TXp; q += 4) dmaw_to_lds<16, 0>(rx, xs + r * xrow + q * 256, lane16, ((u32)r * (u32)K + (u32)(s0 + q) * 128u) * 2u);
#pragma unroll
  for (int d = 1; d < D; ++d) issue(d, d);

  using vec8 = typename DT::vec8;
  Cdna4DequantH<DT> ch;
  ch.init(lane);
  const int mrow = min(i, M - 1);
  const u32 lds0 = (u32)(size_t)(__attribute__((address_space(3))) char*)wbase;
  const u32 ring_lane = lds0 + lane16;
  const u32 sz_lane = lds0 + D * 1024 + i * 4;
  const u32 x_lane = lds0 + D * 1024 + TXp * 64 + mrow * xrow + tsel * 128 + kbl * 16;   // k = 64 tsel + 16 a + 8 kb: a * 32 bytes apart
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int slot = 0;
  auto step = [&](int t, auto vm_, auto reissue_) {
    constexpr int VM = decltype(vm_)::value;
    constexpr bool REISSUE = decltype(reissue_)::value;
    u32x4 w, xo[4];
    u32 sz;
    const u32 ra = ring_lane + slot * 1024, sa = sz_lane + t * 64, xa = x_lane + t * 256;
    dmaw_wait_vm<VM>();
    asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(w) : "v"(ra) : "memory");
    asm volatile("ds_read_b32 %0, %1" : "=v"(sz) : "v"(sa) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(xo[0]) : "v"(xa) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:32" : "=v"(xo[1]) : "v"(xa) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(xo[2]) : "v"(xa) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:96" : "=v"(xo[3]) : "v"(xa) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w), "+v"(sz), "+v"(xo[0]), "+v"(xo[1]), "+v"(xo[2]), "+v"(xo[3]) : : "memory");
    if (REISSUE) issue(t + D, slot);
    if (s0 + t < nit) {
      vec8 op[4];
      ch.tile(w, sz, op);
#pragma unroll
      for (int a = 0; a < 4; ++a) acc = DT::mfma(op[a], __builtin_bit_cast(vec8, xo[a]), acc);
    }
    slot = slot + 1 == D ? 0 : slot + 1;
  };
  int t = 0;
  for (; t < TX - D; ++t) step(t, std::integral_constant<int, D - 1>{}, std::true_type{});
  dmaw_static_for<0, D>([&](auto j_) {
    constexpr int J = decltype(j_)::value;
    step(t + J, std::integral_constant<int, D - 1 - J>{}, std::false_type{});
  });
  // split-K reduction across the block's waves through each wave's own (now idle) first ring slot: acc[r] = C[n = 4 g + r][m = i]
#pragma unroll
  for (int r = 0; r < 4; ++r) reinterpret_cast<float*>(wbase)[r * 64 + lane] = acc[r];
  __syncthreads();
  if (wv < 4 && i < M) {
    const int r = wv;
    float tsum = 0.f;
#pragma unroll
    for (int q = 0; q < WAVES; ++q) tsum += reinterpret_cast<const float*>(smem + q * wave_bytes)[r * 64 + lane];
    const int nn = nb * 16 + 4 * g + r;
    uint16_t o = DT::from_float(tsum);
    if (bias != nullptr) o = DT::from_float(DT::to_float(o) + DT::to_float(bias[nn]));
    out[(size_t)i * N + nn] = o;
  }
}

// waves / d: the product's choice for the shape (8 / 2: one or two slabs per CU; 16 / 1: K >= 12288; 4 / 7: many slabs per CU).  m <= 8.
int launch_gemv_dmaw(const void* x, const void* qw, const void* szh, const void* bias, void* out, int m, int n, int k, int dtype, int waves,
                     int d, hipStream_t st) {
  if (m < 1 || m > 8 || (k % 128) != 0 || (n % 32) != 0 || !szh) return -1;
  const int nit = k / 128;
  while (waves > 4 && waves > nit) waves >>= 1;
  const int tx = (nit + waves - 1) / waves, txp = (tx + 3) & ~3;
  if (d > tx) d = tx;
  const size_t smem = (size_t)waves * ((size_t)d * 1024 + (size_t)txp * 64 + (size_t)m * (txp * 256 + 16));
  if (smem > 160 * 1024) return -1;
#define AWQ_DW(DT_, W_, D_)                                                                                                        \
  if (waves == W_ && d == D_) {                                                                                                    \
    auto kern = gemv_dmaw_kernel<DT_, W_, D_>;                                                                                     \
    static LdsOptIn optin;                                                                                                         \
    if (smem > 64 * 1024) optin.ensure(reinterpret_cast<const void*>(kern));                                                       \
    hipLaunchKernelGGL(kern, dim3(n / 16), dim3(64 * W_), smem, st, (const uint16_t*)x, (const u32*)qw, (const u32*)szh,           \
                       (const uint16_t*)bias, (uint16_t*)out, m, n, k, tx);                                                        \
    return 0;                                                                                                                      \
  }
  if (dtype == 0) {
    AWQ_DW(F16, 8, 2) AWQ_DW(F16, 16, 1) AWQ_DW(F16, 4, 7) AWQ_DW(F16, 4, 4)
  } else {
    AWQ_DW(BF16, 8, 2) AWQ_DW(BF16, 16, 1) AWQ_DW(BF16, 4, 7) AWQ_DW(BF16, 4, 4)
  }
#undef AWQ_DW
  return -1;
}

}  // namespace awq

#endif  // AWQ_ENABLE_PROBES
