// Prefill GEMM v5 on cdna4-interleaved weights (bf16 and fp16, gfx950): "weights never touch LDS".
//
// v4 (awq_gemm_v4.hip) dequantises a 256 x 64 weight tile into LDS once per block and every wave reads x AND weight fragments
// back for v_mfma_f32_32x32x16: its matrix pipe is busy 71 % of the time, 4 % of it with dequant MFMAs, the rest of the gap is
// the per-K-tile block barrier / LDS round trip of the weight tile (profiles/r01_pmc_gemm_v4.txt).  Here a wave owns 32 output
// columns (two 16-row slabs of the weight matrix) for ALL 256 rows of the block's x tile: it streams its own cdna4 tiles from
// global memory straight into registers (like the decode kernel), dequantises them on the matrix core into the A operand of
// v_mfma_f32_16x16x32 (16 weight rows x 32 k -- exactly what Cdna4DequantT produces, no relayout), and multiplies against x
// fragments (32 k x 16 rows) read from the block's LDS tile.  No weight is dequantised twice inside a block, no weight tile is
// written to or read from LDS, and the only block-wide synchronisation is the x-tile hand-over, once per 128 k instead of per 64.
//   block  = 256 rows x 256 columns, 8 waves, wave w = columns [32 w, 32 w + 32)
//   K tile = 128 (one quantisation group): x tile 256 x 128 (64 KiB) by LDS-DMA, two stages (128 KiB)
//   per K tile and wave: 2 weight tiles (2 KiB) + 2 scale dwords from global; 4 k-steps x (2 dequant words, 16 x fragments read
//   in two batches of 8, 32 product MFMAs): 128 product MFMAs (2048 matrix-pipe cycles) per 16 dequant MFMAs (128 cycles)
//   accumulators: 16 row fragments x 2 slabs x 4 = 128 VGPRs
// LDS layout of an x stage: row r (256 B = 16 granules of 8 k) stores logical granule p at slot p ^ (r & 15): the 16 rows of a
// fragment read the same logical granule from 16 different slots (conflict free); the swizzle is applied to the DMA's SOURCE address.
// All LDS reads of the K loop are inline asm (hipcc would drain the DMA queue in front of LDS reads it can see) and the DMA
// waits are counted.  Numerics: the products and the fp32 accumulation order along K are v4's; results differ from v4 only in
// the association inside one 32-k MFMA (16x16x32 vs 2 x 32x32x16) -- held to the same bounds against the oracle.
#include <type_traits>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

namespace {
constexpr int V5_TN = 256, V5_TK = 128;
constexpr int kV5Pitch = 2 * V5_TN + 16;  // epilogue staging: bytes per output row (+16: consecutive rows start 4 banks apart)
}  // namespace

#define V5_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// MF = row fragments (of 16 rows) per wave: 16 -> 256-row blocks, 8 -> 128-row blocks (twice the blocks for the same matrix:
// prompts whose 256-row tiles would fill only half of the CUs)
// SPR: 1 = the DMA pieces of the next x tile are issued in three groups behind the first three k-steps instead of one burst
template <typename DT, int BITS, int MF, int SPR = 0>
__global__ __launch_bounds__(512) void gemm_cdna4_v5_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                            const u32* __restrict__ szp, const uint16_t* __restrict__ bias,
                                                            uint16_t* __restrict__ out, int M, int N, int K, int tiles_m, int tiles_n,
                                                            int n_begin, int n_end) {
  using vec8 = typename DT::vec8;
  constexpr int V5_TM = 16 * MF;
  constexpr int kV5Stage = V5_TM * V5_TK * 2;  // 64 KiB (MF 16) / 32 KiB (MF 8)
  constexpr int NQ = MF / 2;                   // DMA pieces (4 rows each) per wave per K tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int nit = K >> 7;

  // XCD-aware, two-row-band tile order (as awq_gemm_v4.hip)
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
  const int m0 = tm * V5_TM, n0 = n_begin + tn * V5_TN;  // rows >= M are clamped on the way in and not stored

  // ---- x tile DMA: NQ wave-instructions per wave per K tile; instruction q of wave w covers rows 4 NQ w + 4 q .. + 3 ----
  u32 xsrc[NQ];  // element offsets of this lane's granule for its pieces (K-tile 0)
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int row = 4 * NQ * wv + 4 * q + (lane >> 4), p = lane & 15;
    xsrc[q] = (u32)min(m0 + row, M - 1) * (u32)K + (u32)((p ^ (row & 15)) * 8);
  }
  auto issue_x_pieces = [&](int kt, int stage, int q0, int q1) {
    char* dst = smem + stage * kV5Stage + wv * (NQ * 1024);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q < q0 || q >= q1) continue;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x + (size_t)kt * V5_TK + xsrc[q]),
                                       V5_LDS_PTR(dst + q * 1024), 16, 0, 0);
    }
  };
  auto issue_x = [&](int kt, int stage) { issue_x_pieces(kt, stage, 0, NQ); };

  // ---- weights: slabs 2 wv, 2 wv + 1 of the block's 16 ----
  const int nslab = N >> 4, slab_end = min(nslab, n_end >> 4);
  constexpr int kTileWords = BITS == 4 ? 256 : 192, kLaneWords = BITS == 4 ? 4 : 3;
  u32 w_off[2], s_off[2];
  bool live[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int sl = (n0 >> 4) + 2 * wv + s;
    live[s] = sl < slab_end;
    const int slc = min(sl, slab_end - 1);
    w_off[s] = (u32)slc * (u32)nit * kTileWords + lane * kLaneWords;
    s_off[s] = (u32)slc * (u32)nit * 16 + i;
  }
  struct WG {
    u32x4 w[2];
    u32 sz[2];
  };
  auto load_w = [&](int grp) {
    WG r;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const u32* wp = qw + (size_t)grp * kTileWords + w_off[s];
      if (BITS == 4) {
        r.w[s] = *reinterpret_cast<const u32x4*>(wp);
      } else {
        typedef u32 u32x3 __attribute__((ext_vector_type(3)));
        const u32x3 w3 = *reinterpret_cast<const u32x3*>(wp);
        r.w[s] = u32x4{w3.x, w3.y, w3.z, 0u};
      }
      r.sz[s] = szp[(size_t)grp * 16 + s_off[s]];
    }
    return r;
  };
  Cdna4DequantT<DT> cd;
  cd.init(lane, BITS == 4 ? 0x000F000Fu : 0x00070007u);

  // ---- x fragment addresses: fragment f (rows 16 f .. 16 f + 15), k-step a: row 16 f + i, logical granule 4 a + g ----
  const u32 lds0 = (u32)(size_t)(__attribute__((address_space(3))) char*)smem;
  u32 xa[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) xa[a] = lds0 + i * 256 + (((4 * a + g) ^ i) << 4);

  f32x4 acc[MF][2];
#pragma unroll
  for (int f = 0; f < MF; ++f)
#pragma unroll
    for (int s = 0; s < 2; ++s) acc[f][s] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue_x(0, 0);
  WG cur = load_w(0);
  for (int t = 0; t < nit; ++t) {
    const int stage = t & 1;
    // tile t and this group's weights were requested a whole K tile ago: wait for all of it, then meet the other waves (every
    // wave's pieces of tile t are in LDS; every wave is done reading tile t - 1, whose stage the next DMA overwrites)
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    // hipcc does not see that wait: touching the group's registers HERE makes it place its own wait for them in front of the
    // next DMA (where nothing is outstanding) instead of behind it (where it would drain the DMA it just issued)
    asm volatile("" : "+v"(cur.w[0]), "+v"(cur.w[1]), "+v"(cur.sz[0]), "+v"(cur.sz[1]));
    __builtin_amdgcn_s_barrier();
    WG nxt = load_w(min(t + 1, nit - 1));
    if (t + 1 < nit) {
      if (SPR == 0) issue_x(t + 1, stage ^ 1);
      else issue_x_pieces(t + 1, stage ^ 1, 0, (3 * NQ) / 8);
    }
    // (expand w3c tiles once per group)
    u32x4 wt[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) wt[s] = BITS == 4 ? cur.w[s] : w3_expand(cur.w[s].x, cur.w[s].y, cur.w[s].z);
    u32 b01[2], b23[2];
    float cv[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const u32 sdup = __builtin_amdgcn_perm(cur.sz[s], cur.sz[s], 0x01000100u);
      b01[s] = sdup & cd.m01;
      b23[s] = sdup & cd.m23;
      cv[s] = DT::dq_offset(cur.sz[s]);
    }
    const u32 sbase = (u32)stage * (u32)kV5Stage;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      vec8 op[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const u32 word = a == 0 ? wt[s].x : (a == 1 ? wt[s].y : (a == 2 ? wt[s].z : wt[s].w));
        op[s] = cd.word(word, b01[s], b23[s], cv[s]);
      }
      const u32 addr = xa[a] + sbase;
      if (SPR != 0 && t + 1 < nit) {
        if (a == 1) issue_x_pieces(t + 1, stage ^ 1, (3 * NQ) / 8, (6 * NQ) / 8);
        if (a == 2) issue_x_pieces(t + 1, stage ^ 1, (6 * NQ) / 8, NQ);
      }
#pragma unroll
      for (int h = 0; h < MF / 8; ++h) {
        u32x4 xf[8];
#define V5_RD(j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[j]) : "v"(addr), "n"((8 * 0 + j) * 4096) : "memory")
#define V5_RD_HI(j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[j]) : "v"(addr), "n"((8 + j) * 4096) : "memory")
        if (h == 0) {
          V5_RD(0); V5_RD(1); V5_RD(2); V5_RD(3); V5_RD(4); V5_RD(5); V5_RD(6); V5_RD(7);
        } else {
          V5_RD_HI(0); V5_RD_HI(1); V5_RD_HI(2); V5_RD_HI(3); V5_RD_HI(4); V5_RD_HI(5); V5_RD_HI(6); V5_RD_HI(7);
        }
#undef V5_RD
#undef V5_RD_HI
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(xf[0]), "+v"(xf[1]), "+v"(xf[2]), "+v"(xf[3]), "+v"(xf[4]), "+v"(xf[5]), "+v"(xf[6]), "+v"(xf[7])
                     :
                     : "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
          for (int s = 0; s < 2; ++s) acc[8 * h + j][s] = DT::mfma(op[s], __builtin_bit_cast(vec8, xf[j]), acc[8 * h + j][s]);
      }
    }
    cur = nxt;
  }

  // ---- epilogue through LDS: acc[f][s][r] = C[n = n0 + 32 wv + 16 s + 4 g + r][m = m0 + 16 f + i].  The block's 16 MF x 256 output is
  // staged row-major (pitch 528 B) so that the global stores are whole 512-byte rows, 16 B per lane ----
  __builtin_amdgcn_s_barrier();  // every wave is done with the x stages
  {
    const u32 wbase = lds0 + i * kV5Pitch + (32 * wv + 4 * g) * 2;
#pragma unroll
    for (int f = 0; f < MF; ++f)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        u32x2 v;
        v.x = (u32)DT::from_float(acc[f][s][0]) | ((u32)DT::from_float(acc[f][s][1]) << 16);
        v.y = (u32)DT::from_float(acc[f][s][2]) | ((u32)DT::from_float(acc[f][s][3]) << 16);
        asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(wbase + f * (16 * kV5Pitch)), "v"(v), "n"(s * 32) : "memory");
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");
  {
    const int col = (lane & 31) * 8;          // 8 columns (16 B) per lane, two rows per wave-instruction
    const int nn = n0 + col;
    const bool ncol_ok = nn < n_end;
    u32x4 bv = {0u, 0u, 0u, 0u};
    if (bias != nullptr && ncol_ok) bv = *reinterpret_cast<const u32x4*>(bias + nn);
#pragma unroll
    for (int it = 0; it < MF; ++it) {
      const int row = 2 * MF * wv + 2 * it + (lane >> 5);
      u32x4 v;
      asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(lds0 + row * kV5Pitch + col * 2) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory");
      const int m = m0 + row;
      if (ncol_ok && m < M) {
        if (bias != nullptr) {  // `out + self.bias` in T (qmodule.py:221)
          auto add2 = [](u32 a, u32 b) {
            const float lo = DT::to_float((uint16_t)(a & 0xFFFFu)) + DT::to_float((uint16_t)(b & 0xFFFFu));
            const float hi = DT::to_float((uint16_t)(a >> 16)) + DT::to_float((uint16_t)(b >> 16));
            return (u32)DT::from_float(lo) | ((u32)DT::from_float(hi) << 16);
          };
          v = u32x4{add2(v.x, bv.x), add2(v.y, bv.y), add2(v.z, bv.z), add2(v.w, bv.w)};
        }
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out + (size_t)m * N + nn));  // streamed: keep x / weight panels in L2
      }
    }
  }
}

// weight rows [n_begin, n_end) with (16 mf) x 256 blocks, mf = 16 or 8; any m >= 1 (rows past m are clamped / not stored)
void launch_gemm_cdna4_v5(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int n_begin,
                          int n_end, int dtype, hipStream_t st, int bits, int mf) {
  const int tm = mf == 17 ? 256 : 16 * mf;
  const int stage = tm * V5_TK * 2, epi = tm * kV5Pitch;
  const int smem = 2 * stage > epi ? 2 * stage : epi;
  const int tiles_m = (m + tm - 1) / tm, tiles_n = (n_end - n_begin + V5_TN - 1) / V5_TN;
  using Kern = void (*)(const uint16_t*, const u32*, const u32*, const uint16_t*, uint16_t*, int, int, int, int, int, int, int);
  static const Kern spread[2] = {gemm_cdna4_v5_kernel<F16, 4, 16, 1>, gemm_cdna4_v5_kernel<BF16, 4, 16, 1>};
  static const Kern kerns[2][2][2] = {
      {{gemm_cdna4_v5_kernel<F16, 4, 16>, gemm_cdna4_v5_kernel<F16, 4, 8>}, {gemm_cdna4_v5_kernel<F16, 3, 16>, gemm_cdna4_v5_kernel<F16, 3, 8>}},
      {{gemm_cdna4_v5_kernel<BF16, 4, 16>, gemm_cdna4_v5_kernel<BF16, 4, 8>}, {gemm_cdna4_v5_kernel<BF16, 3, 16>, gemm_cdna4_v5_kernel<BF16, 3, 8>}}};
  const int a = dtype == 0 ? 0 : 1, b = bits == 3 ? 1 : 0, c = mf == 8 ? 1 : 0;
  const bool spr = mf == 17;  // experiments: 256-row blocks with the spread DMA
  const Kern kern = spr ? spread[a] : kerns[a][b][c];
  static LdsOptIn optin[2][2][2], optin_s[2];
  (spr ? optin_s[a] : optin[a][b][c]).ensure(reinterpret_cast<const void*>(kern), smem);
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), smem, st, (const uint16_t*)x, (const u32*)qw, (const u32*)szp,
                     (const uint16_t*)bias, (uint16_t*)out, m, n, k, tiles_m, tiles_n, n_begin, n_end);
}

}  // namespace awq
