// Prefill GEMM (M > 16) for W4A16, gfx950.
//
// Replaces gemm_w4a16_T1 / gemm_w4a16_T2 (reference awq/kernels/csrc/quantization_new/gemm/gemm_cuda.cu:312-1124).
// v1 structure ("gemm"): 128x128x64 block tile, 4 waves (2x2, 64x64 each),
// x tile and the DEQUANTISED weight tile both staged in XOR-swizzled LDS as 16-bit T, fragments
// read with ds_read_b128, v_mfma_f32_16x16x32 with fp32 accumulators.  The packed int4 tile is
// loaded once per block (16 B per thread per K-step), dequantised once (reference numerics:
// round_T(q*s+sz)) and shared by all waves, so the unpack cost is amortised over 128 rows of M.
// Global loads for step t+1 are issued before the MFMAs of step t (register staging).
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

template <typename DT>
__global__ __launch_bounds__(256) void gemm_w4a16_128x128_kernel(const uint16_t* __restrict__ x,
                                                                 const u32* __restrict__ qw,
                                                                 const uint16_t* __restrict__ scales,
                                                                 const uint16_t* __restrict__ zeros,
                                                                 uint16_t* __restrict__ out, int M, int N, int K,
                                                                 int tiles_m) {
  using vec8 = typename DT::vec8;
  __shared__ __attribute__((aligned(16))) char smem[2 * BM * BK * 2];
  char* As = smem;                 // x tile      [128][64] T
  char* Bs = smem + BM * BK * 2;   // weight tile [128][64] T (dequantised)

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int i = lane & 15, g = lane >> 4;
  const int wm = wv >> 1, wn = wv & 1;
  // tile_m fastest: the blocks that share one weight panel run together (weights read once from HBM)
  const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
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
  // weights: thread -> (row nl, 32-k chunk c) ; 8 consecutive threads read one 128-B v2 block
  const int nl = (tid >> 3) * 4 + ((tid >> 1) & 3), c = tid & 1;
  const int nrow = min(n0 + nl, N - 1);
  const u32* b_src = qw + v2_chunk_word(nrow, c, K);  // + kt*32 words per K-step (64 k)
  const uint16_t* s_src = scales + nrow;
  const uint16_t* z_src = zeros + nrow;

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  u32x4 ra[4], rb;
  uint16_t rs, rz;
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const u32x4*>(a_src[p] + (size_t)kt * BK);
    rb = *reinterpret_cast<const u32x4*>(b_src + (size_t)kt * 32);
    const int grp = (kt * BK) / kGroup;
    rs = s_src[(size_t)grp * N];
    rz = z_src[(size_t)grp * N];
  };
  load_tile(0);

  for (int kt = 0; kt < nk; ++kt) {
    // registers -> LDS (weights are dequantised here, once per block)
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(As + a_dst[p]) = ra[p];
    {
      vec8 wop[4];
      dequant_chunk<DT>(rb, DT::make_sz(rs, rz), wop);
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<vec8*>(Bs + tile_off(nl, c * 4 + j)) = wop[j];
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

size_t gemm_workspace_bytes(int, int, int) { return 0; }

int gemm_tune_set(const char*, int) { return -1; }

template <typename DT>
static int launch_gemm_t(const void* x, const void* qw, const void* s, const void* z, void* out, int m, int n, int k,
                         hipStream_t st) {
  const int tiles_m = (m + BM - 1) / BM, tiles_n = (n + BN - 1) / BN;
  dim3 grid(tiles_m * tiles_n), block(256);
  hipLaunchKernelGGL((gemm_w4a16_128x128_kernel<DT>), grid, block, 0, st, (const uint16_t*)x, (const u32*)qw,
                     (const uint16_t*)s, (const uint16_t*)z, (uint16_t*)out, m, n, k, tiles_m);
  return 0;
}

int launch_gemm(const void* x, const void* qw, const void* s, const void* z, void* out, int m, int n, int k, int dtype,
                void*, size_t, hipStream_t st) {
  if (m <= 16) return launch_gemv(x, qw, s, z, out, m, n, k, dtype, st);
  return dtype == 0 ? launch_gemm_t<F16>(x, qw, s, z, out, m, n, k, st)
                    : launch_gemm_t<BF16>(x, qw, s, z, out, m, n, k, st);
}

}  // namespace awq
