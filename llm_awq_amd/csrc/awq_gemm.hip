// Prefill GEMM (M > 16) for W4A16, gfx950.
//
// Replaces gemm_w4a16_T1 / gemm_w4a16_T2 (reference awq/kernels/csrc/quantization_new/gemm/gemm_cuda.cu:312-1124).
// v1 structure ("gemm"): 128x128x64 block tile, 4 waves (2x2, 64x64 each),
// x tile and the DEQUANTISED weight tile both staged in XOR-swizzled LDS as 16-bit T, fragments
// read with ds_read_b128, v_mfma_f32_16x16x32 with fp32 accumulators.  The packed int4 tile is
// loaded once per block (16 B per thread per K-step), dequantised once (reference numerics:
// round_T(q*s+sz)) and shared by all waves, so the unpack cost is amortised over 128 rows of M.
// Global loads for step t+1 are issued before the MFMAs of step t (register staging).
#include <string.h>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

namespace {
constexpr int BM = 128, BN = 128, BK = 64;

// byte offset of 16-byte granule `gc` (8 elements along k) of row `row` in a [rows][64] 16-bit tile.
// 256-B LDS bank row = 2 tile rows; XOR with (row>>1)&7 makes the 16 rows of an MFMA fragment hit
// 16 distinct 16-B slots (conflict-free ds_read_b128 / ds_write_b128).
__device__ __forceinline__ int tile_off(int row, int gc) { return row * 128 + ((gc ^ ((row >> 1) & 7)) << 4); }
}  // namespace

// one 128 x 128 output tile (tm, tn) of out[M, N] = x[M, K] . Wdeq[N, K]^T; shared by the plain and the grouped kernel
template <typename DT, int LAYOUT>
__device__ __forceinline__ void gemm128_tile(char* smem, const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                             const uint16_t* __restrict__ scales, const uint16_t* __restrict__ zeros,
                                             uint16_t* __restrict__ out, int M, int N, int K, int tm, int tn) {
  using vec8 = typename DT::vec8;
  char* As = smem;                 // x tile      [128][64] T
  char* Bs = smem + BM * BK * 2;   // weight tile [128][64] T (dequantised)

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int wm = wv >> 1, wn = wv & 1;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging assignments ----
  // x: 1024 granules / 256 threads = 4 each; granule q -> row q/8, gc q%8
  const uint16_t* a_src[4];
  int a_dst[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int q = tid + 256 * p, row = q >> 3, gc = q & 7;
    a_src[p] = x + (size_t)min(m0 + row, M - 1) * K + gc * 8;
    a_dst[p] = tile_off(row, gc);
  }
  // weights, LAYOUT 0 (v2): thread -> (row nl, 32-k chunk c); 8 consecutive threads read one 128-B v2 block.
  // LAYOUT 1 (cdna4): wave wv dequantises slabs 2*wv, 2*wv+1 of the tile on the matrix core; per K-step (64 k)
  // a lane needs words {2h, 2h+1} (h = K-step parity) of its 16 bytes of each slab's 1-KiB tile.
  const int nl = LAYOUT == 0 ? (tid >> 3) * 4 + ((tid >> 1) & 3) : (2 * wv) * 16 + i;
  const int c = tid & 1;
  const int nit = K / kGroup;
  const int sl0 = min((n0 >> 4) + 2 * wv, (N >> 4) - 1), sl1 = min((n0 >> 4) + 2 * wv + 1, (N >> 4) - 1);
  const int nrow = LAYOUT == 0 ? min(n0 + nl, N - 1) : sl0 * 16 + i;
  const int nrow1 = sl1 * 16 + i;  // cdna4: the wave's second slab
  const u32* b_src = LAYOUT == 0 ? qw + v2_chunk_word(nrow, c, K)  // + kt*32 words per K-step (64 k)
                                 : qw + cdna4_tile_word(sl0, 0, nit) + lane * 4;
  const u32* b_src1 = qw + cdna4_tile_word(sl1, 0, nit) + lane * 4;
  const uint16_t* s_src = scales + nrow;
  const uint16_t* z_src = zeros + nrow;
  Cdna4DequantT<DT> cd;
  if (LAYOUT == 1) cd.init(lane);

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  u32x4 ra[4], rb;       // rb: v2 = one 16-byte chunk; cdna4 = {slab0 words 2h,2h+1, slab1 words 2h,2h+1}
  uint16_t rs, rz, rs1 = 0, rz1 = 0;
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const u32x4*>(a_src[p] + (size_t)kt * BK);
    const int grp = (kt * BK) / kGroup;
    if (LAYOUT == 0) {
      rb = *reinterpret_cast<const u32x4*>(b_src + (size_t)kt * 32);
    } else {
      const u32x2 lo = *reinterpret_cast<const u32x2*>(b_src + (size_t)grp * 256 + 2 * (kt & 1));
      const u32x2 hi = *reinterpret_cast<const u32x2*>(b_src1 + (size_t)grp * 256 + 2 * (kt & 1));
      rb = u32x4{lo.x, lo.y, hi.x, hi.y};
      rs1 = scales[(size_t)grp * N + nrow1];
      rz1 = zeros[(size_t)grp * N + nrow1];
    }
    rs = s_src[(size_t)grp * N];
    rz = z_src[(size_t)grp * N];
  };
  load_tile(0);

  for (int kt = 0; kt < nk; ++kt) {
    // registers -> LDS (weights are dequantised here, once per block)
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(As + a_dst[p]) = ra[p];
    if (LAYOUT == 0) {
      vec8 wop[4];
      dequant_chunk<DT>(rb, DT::make_sz(rs, rz), wop);
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<vec8*>(Bs + tile_off(nl, c * 4 + j)) = wop[j];
    } else {
      // lane (row i of the slab, octet g): word w covers k = 32w + 8g + 0..7 -> granule 4w + g
      const u32 sd0 = (u32)rs * 0x00010001u, sd1 = (u32)rs1 * 0x00010001u;
      const float c0 = DT::dq_offset((u32)rs | ((u32)rz << 16)), c1 = DT::dq_offset((u32)rs1 | ((u32)rz1 << 16));
      *reinterpret_cast<vec8*>(Bs + tile_off(nl, 0 + g)) = cd.word(rb.x, sd0 & cd.m01, sd0 & cd.m23, c0);
      *reinterpret_cast<vec8*>(Bs + tile_off(nl, 4 + g)) = cd.word(rb.y, sd0 & cd.m01, sd0 & cd.m23, c0);
      *reinterpret_cast<vec8*>(Bs + tile_off(nl + 16, 0 + g)) = cd.word(rb.z, sd1 & cd.m01, sd1 & cd.m23, c1);
      *reinterpret_cast<vec8*>(Bs + tile_off(nl + 16, 4 + g)) = cd.word(rb.w, sd1 & cd.m01, sd1 & cd.m23, c1);
    }
    __syncthreads();
    if (kt + 1 < nk) load_tile(kt + 1);

#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      vec8 af[4], bf[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        af[t] = *reinterpret_cast<const vec8*>(Bs + tile_off(wn * 64 + t * 16 + i, ks * 4 + g));  // weights = A operand
        bf[t] = *reinterpret_cast<const vec8*>(As + tile_off(wm * 64 + t * 16 + i, ks * 4 + g));  // x = B operand
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = DT::mfma(af[a], bf[b], acc[a][b]);
    }
    __syncthreads();
  }

  // epilogue: acc[a][b][r] = C[n = n0 + wn*64 + a*16 + 4g + r][m = m0 + wm*64 + b*16 + i]
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int m = m0 + wm * 64 + b * 16 + i;
    if (m >= M) continue;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int nn = n0 + wn * 64 + a * 16 + 4 * g;
      if (nn + 3 < N) {
        u32x2 v;
        v.x = (u32)DT::from_float(acc[a][b][0]) | ((u32)DT::from_float(acc[a][b][1]) << 16);
        v.y = (u32)DT::from_float(acc[a][b][2]) | ((u32)DT::from_float(acc[a][b][3]) << 16);
        *reinterpret_cast<u32x2*>(out + (size_t)m * N + nn) = v;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (nn + r < N) out[(size_t)m * N + nn + r] = DT::from_float(acc[a][b][r]);
      }
    }
  }
}

template <typename DT, int LAYOUT>
__global__ __launch_bounds__(256) void gemm_w4a16_128x128_kernel(const uint16_t* __restrict__ x,
                                                                 const u32* __restrict__ qw,
                                                                 const uint16_t* __restrict__ scales,
                                                                 const uint16_t* __restrict__ zeros,
                                                                 uint16_t* __restrict__ out, int M, int N, int K,
                                                                 int tiles_m) {
  __shared__ __attribute__((aligned(16))) char smem[2 * BM * BK * 2];
  // tile_m fastest: the blocks that share one weight panel run together (weights read once from HBM)
  gemm128_tile<DT, LAYOUT>(smem, x, qw, scales, zeros, out, M, N, K, blockIdx.x % tiles_m, blockIdx.x / tiles_m);
}

// ---------------------------------------------------------------------------------------------
// Grouped (per-expert) GEMM for MoE layers (BASELINE.json config 5, Mixtral-8x7B; no reference counterpart):
// tokens are sorted by expert, expert e owns rows [offsets[e], offsets[e+1]) of x / out and the e-th slice of the
// stacked packed weights.  The grid is sized for the worst case sum_e ceil(m_e / 128) <= T / 128 + E row tiles; a
// block finds its (expert, row tile) by walking the E + 1 offsets (E is small) and exits if there is none.
// ---------------------------------------------------------------------------------------------
template <typename DT, int LAYOUT>
__global__ __launch_bounds__(256) void moe_gemm_w4a16_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                             const uint16_t* __restrict__ scales,
                                                             const uint16_t* __restrict__ zeros,
                                                             const int* __restrict__ offsets, uint16_t* __restrict__ out,
                                                             int E, int N, int K, int gpad, int tiles_n) {
  __shared__ __attribute__((aligned(16))) char smem[2 * BM * BK * 2];
  const int tn = blockIdx.x % tiles_n;  // column tiles of one row tile run together (x tile shared through L2)
  int mt = blockIdx.x / tiles_n;
  int e = 0, m_e = 0, row0 = 0;
  for (; e < E; ++e) {
    row0 = offsets[e];
    m_e = offsets[e + 1] - row0;
    const int t = (m_e + BM - 1) / BM;
    if (mt < t) break;
    mt -= t;
  }
  if (e == E) return;
  gemm128_tile<DT, LAYOUT>(smem, x + (size_t)row0 * K, qw + (size_t)e * ((size_t)N * K / 8), scales + (size_t)e * gpad * N,
                           zeros + (size_t)e * gpad * N, out + (size_t)row0 * N, m_e, N, K, mt, tn);
}

// ---------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves (2 along M x 4 along N, 128 x 64 each), double-buffered LDS (2 x 64 KiB):
//   * x tile: global_load_lds_dwordx4 straight into LDS (no VGPR round trip).  The LDS image is
//     lane-linear, so the XOR swizzle is applied to the per-lane SOURCE address and to the ds_read.
//   * weight tile: 16 B of packed int4 per thread per K-tile, prefetched two tiles ahead into VGPRs,
//     dequantised (reference numerics) while the MFMAs of the current tile run, ds_write_b128 into the
//     other buffer.  One barrier per K-tile.
//   * epilogue: accumulators -> LDS (per-wave [128 m][64 n] image, 144-B rows) -> 16-byte row stores.
// ---------------------------------------------------------------------------------------------
namespace {
constexpr int TM = 256, TN = 256, TK = 64;
constexpr int kBufBytes = (TM + TN) * TK * 2;  // 64 KiB per stage
constexpr int kEpiRow = 144;                   // bytes per staged output row (64 n x 2 B + 16 pad)
constexpr int kSmem256 = 8 * 128 * kEpiRow;    // 147456 >= 2 * kBufBytes
}  // namespace

template <typename DT, int LAYOUT>
__global__ __launch_bounds__(512) void gemm_w4a16_256x256_kernel(const uint16_t* __restrict__ x,
                                                                 const u32* __restrict__ qw,
                                                                 const uint16_t* __restrict__ scales,
                                                                 const uint16_t* __restrict__ zeros,
                                                                 uint16_t* __restrict__ out, int M, int N, int K,
                                                                 int tiles_m, int tiles_n) {
  using vec8 = typename DT::vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int wm = wv >> 2, wn = wv & 3;

  // XCD-aware tile order: block b runs on XCD b % 8; give every XCD a contiguous range of tiles so that
  // the weight panels it touches stay in ITS L2 (bijective for any tile count).
  const int T = tiles_m * tiles_n;
  int tile;
  {
    const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
    const int q = T >> 3, r = T & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tm = tile % tiles_m, tn = tile / tiles_m;
  const int m0 = tm * TM, n0 = tn * TN;

  // ---- x staging (LDS-DMA): 4 granules per thread per K-tile ----
  const uint16_t* a_src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int p = q * 512 + tid, row = p >> 3, gcp = p & 7;
    const int gc = gcp ^ ((row >> 1) & 7);  // LDS slot gcp of this row receives source granule gc
    a_src[q] = x + (size_t)min(m0 + row, M - 1) * K + gc * 8;
  }
  // ---- weight staging.  LAYOUT 0: thread -> (row nl, 32-k chunk c).  LAYOUT 1 (cdna4): wave wv owns slabs
  //      2*wv, 2*wv+1 of the 16 in the tile and dequantises them on the matrix core ----
  const int nl = LAYOUT == 0 ? (tid >> 3) * 4 + ((tid >> 1) & 3) : (2 * wv) * 16 + i;
  const int c = tid & 1;
  const int nit = K / kGroup;
  const int sl0 = min((n0 >> 4) + 2 * wv, (N >> 4) - 1), sl1 = min((n0 >> 4) + 2 * wv + 1, (N >> 4) - 1);
  const int nrow = LAYOUT == 0 ? min(n0 + nl, N - 1) : sl0 * 16 + i;
  const int nrow1 = sl1 * 16 + i;
  const u32* b_src = LAYOUT == 0 ? qw + v2_chunk_word(nrow, c, K)  // + kt*32 words
                                 : qw + cdna4_tile_word(sl0, 0, nit) + lane * 4;
  const u32* b_src1 = qw + cdna4_tile_word(sl1, 0, nit) + lane * 4;
  const uint16_t* s_src = scales + nrow;
  const uint16_t* z_src = zeros + nrow;
  Cdna4DequantT<DT> cd;
  if (LAYOUT == 1) cd.init(lane);

  auto issue_a = [&](int kt, int buf) {
    char* dst = smem + buf * kBufBytes + wv * 1024;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_src[q] + (size_t)kt * TK),
                                       (__attribute__((address_space(3))) void*)(dst + q * 8192), 16, 0, 0);
  };
  struct BRegs {
    u32x4 w;
    uint16_t s, z, s1, z1;
  };
  auto write_b = [&](const BRegs& r, int buf) {
    char* Bs = smem + buf * kBufBytes + TM * TK * 2;
    if (LAYOUT == 0) {
      vec8 wop[4];
      dequant_chunk<DT>(r.w, DT::make_sz(r.s, r.z), wop);
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<vec8*>(Bs + tile_off(nl, c * 4 + j)) = wop[j];
    } else {
      const u32 sd0 = (u32)r.s * 0x00010001u, sd1 = (u32)r.s1 * 0x00010001u;
      const float c0 = DT::dq_offset((u32)r.s | ((u32)r.z << 16)), c1 = DT::dq_offset((u32)r.s1 | ((u32)r.z1 << 16));
      *reinterpret_cast<vec8*>(Bs + tile_off(nl, 0 + g)) = cd.word(r.w.x, sd0 & cd.m01, sd0 & cd.m23, c0);
      *reinterpret_cast<vec8*>(Bs + tile_off(nl, 4 + g)) = cd.word(r.w.y, sd0 & cd.m01, sd0 & cd.m23, c0);
      *reinterpret_cast<vec8*>(Bs + tile_off(nl + 16, 0 + g)) = cd.word(r.w.z, sd1 & cd.m01, sd1 & cd.m23, c1);
      *reinterpret_cast<vec8*>(Bs + tile_off(nl + 16, 4 + g)) = cd.word(r.w.w, sd1 & cd.m01, sd1 & cd.m23, c1);
    }
  };

  f32x4 acc[4][8];  // [n-frag][m-frag]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / TK;
  BRegs rb_cur, rb_nxt;
  auto load_b = [&](int kt, BRegs& r) {
    const int grp = (kt * TK) / kGroup;
    if (LAYOUT == 0) {
      r.w = *reinterpret_cast<const u32x4*>(b_src + (size_t)kt * 32);
      r.s1 = r.z1 = 0;
    } else {
      const u32x2 lo = *reinterpret_cast<const u32x2*>(b_src + (size_t)grp * 256 + 2 * (kt & 1));
      const u32x2 hi = *reinterpret_cast<const u32x2*>(b_src1 + (size_t)grp * 256 + 2 * (kt & 1));
      r.w = u32x4{lo.x, lo.y, hi.x, hi.y};
      r.s1 = scales[(size_t)grp * N + nrow1];
      r.z1 = zeros[(size_t)grp * N + nrow1];
    }
    r.s = s_src[(size_t)grp * N];
    r.z = z_src[(size_t)grp * N];
  };

  issue_a(0, 0);
  load_b(0, rb_cur);
  load_b(nk > 1 ? 1 : 0, rb_nxt);
  // Drain EVERYTHING before the loop: hipcc waits vmcnt(0) at any use of an ordinary load while an
  // LDS-DMA is in flight, so the loop is arranged such that ordinary loads are only consumed after a
  // barrier that already drained them (the scoreboard must be provably empty at the loop header).
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  write_b(rb_cur, 0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const char* As = smem + buf * kBufBytes;
    const char* Bs = As + TM * TK * 2;
    if (kt + 1 < nk) issue_a(kt + 1, buf ^ 1);
    rb_cur = rb_nxt;  // tile kt+1's packed weights (already in registers)
    if (kt + 2 < nk) load_b(kt + 2, rb_nxt);

#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      vec8 wf[4], xf[8];
#pragma unroll
      for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const vec8*>(Bs + tile_off(wn * 64 + t * 16 + i, ks * 4 + g));
#pragma unroll
      for (int t = 0; t < 8; ++t) xf[t] = *reinterpret_cast<const vec8*>(As + tile_off(wm * 128 + t * 16 + i, ks * 4 + g));
#pragma unroll
      for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a][b] = DT::mfma(wf[a], xf[b], acc[a][b]);
      if (ks == 0 && kt + 1 < nk) write_b(rb_cur, buf ^ 1);  // dequant work to hide under the MFMAs
    }
    __syncthreads();
  }

  // ---- epilogue through LDS: acc[a][b][r] = C[n = wn*64 + a*16 + 4g + r][m = wm*128 + b*16 + i] ----
  char* eb = smem + wv * (128 * kEpiRow);
#pragma unroll
  for (int b = 0; b < 8; ++b)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      u32x2 v;
      v.x = (u32)DT::from_float(acc[a][b][0]) | ((u32)DT::from_float(acc[a][b][1]) << 16);
      v.y = (u32)DT::from_float(acc[a][b][2]) | ((u32)DT::from_float(acc[a][b][3]) << 16);
      *reinterpret_cast<u32x2*>(eb + (b * 16 + i) * kEpiRow + a * 32 + g * 8) = v;
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

size_t gemm_workspace_bytes(int, int, int) { return 0; }

int launch_moe_gemm(const void* x, const void* qw, const void* s, const void* z, const void* offsets, void* out, int total_m,
                    int experts, int n, int k, int gpad, int dtype, int layout, hipStream_t st) {
  const int tiles_n = (n + BN - 1) / BN;
  const int row_tiles = total_m / BM + experts;  // >= sum_e ceil(m_e / BM)
  dim3 grid((unsigned)(row_tiles * tiles_n)), block(256);
#define AWQ_MOE(DT_, L_)                                                                                              \
  hipLaunchKernelGGL((moe_gemm_w4a16_kernel<DT_, L_>), grid, block, 0, st, (const uint16_t*)x, (const u32*)qw,      \
                     (const uint16_t*)s, (const uint16_t*)z, (const int*)offsets, (uint16_t*)out, experts, n, k, gpad, \
                     tiles_n)
  if (layout == 1 && dtype == 0) AWQ_MOE(F16, 1);
  else if (layout == 1) AWQ_MOE(BF16, 1);
  else if (dtype == 0) AWQ_MOE(F16, 0);
  else AWQ_MOE(BF16, 0);
#undef AWQ_MOE
  return 0;
}

namespace {
int g_gemm_variant = 0;  // 0 = auto, 1 = force 128x128, 2 = force 256x256, 3 = force 256x256 v3 (cdna4 layout)
}
int gemm_variant_get() { return g_gemm_variant; }
int gemm_tune_set(const char* key, int value) {
  if (!strcmp(key, "gemm_variant")) {
    g_gemm_variant = value;
    return 0;
  }
  return -1;
}

template <typename DT, int LAYOUT>
static int launch_gemm_t(const void* x, const void* qw, const void* s, const void* z, void* out, int m, int n, int k,
                         hipStream_t st) {
  const bool big = g_gemm_variant == 2 || (g_gemm_variant == 0 && m > 128);
  if (big) {
    const int tiles_m = (m + TM - 1) / TM, tiles_n = (n + TN - 1) / TN;
    auto kern = gemm_w4a16_256x256_kernel<DT, LAYOUT>;
    static LdsOptIn optin;  // per (kernel instantiation, device)
    optin.ensure(reinterpret_cast<const void*>(kern), kSmem256);
    hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(512), kSmem256, st, (const uint16_t*)x, (const u32*)qw,
                       (const uint16_t*)s, (const uint16_t*)z, (uint16_t*)out, m, n, k, tiles_m, tiles_n);
    return 0;
  }
  const int tiles_m = (m + BM - 1) / BM, tiles_n = (n + BN - 1) / BN;
  dim3 grid(tiles_m * tiles_n), block(256);
  hipLaunchKernelGGL((gemm_w4a16_128x128_kernel<DT, LAYOUT>), grid, block, 0, st, (const uint16_t*)x, (const u32*)qw,
                     (const uint16_t*)s, (const uint16_t*)z, (uint16_t*)out, m, n, k, tiles_m);
  return 0;
}

int launch_gemm(const void* x, const void* qw, const void* s, const void* z, const void* szp, void* out, int m, int n,
                int k, int dtype, int layout, void* ws, size_t ws_bytes, hipStream_t st) {
  if (layout == 0 && g_gemm_variant == 0 && m > 8 && m <= 255 &&
      launch_skinny_v2(x, qw, s, z, nullptr, out, m, n, k, k / kGroup, dtype, st) == 0)
    return 0;  // short prompts / batched decode on un-repacked (fp16) checkpoints
  if (m <= 16) return launch_gemv(x, qw, s, z, szp, out, m, n, k, dtype, layout, st);
  if (layout == 1) {
    // variant 3 / auto: v3 kernel with the tile width picked by chip fill; 4 = force 256 x 256; 5 = force 256 x 128
    if ((g_gemm_variant >= 3 || (g_gemm_variant == 0 && gemm_cdna4_v3_takes(m, k))) &&
        launch_gemm_cdna4_v3(x, qw, szp, nullptr, out, m, n, k, g_gemm_variant == 4 ? 256 : (g_gemm_variant == 5 ? 128 : 0), dtype, ws, ws_bytes, st) == 0)
      return 0;
    return dtype == 0 ? launch_gemm_t<F16, 1>(x, qw, s, z, out, m, n, k, st) : launch_gemm_t<BF16, 1>(x, qw, s, z, out, m, n, k, st);
  }
  return dtype == 0 ? launch_gemm_t<F16, 0>(x, qw, s, z, out, m, n, k, st)
                    : launch_gemm_t<BF16, 0>(x, qw, s, z, out, m, n, k, st);
}

}  // namespace awq
