// W3 ("w3c") format kernels: pack / unpack / dequant / expand-to-W4.  The 3-bit format is this repository's
// (awq_device.hpp "W3 tiles"); the quantisation grid is the reference's pseudo_quantize_tensor with n_bit = 3
// (awq/quantize/quantizer.py:61-103).  unpack / dequant run the SAME device routines as the W3 GEMV, so their
// bit-exact agreement with the oracle pins the index arithmetic and the numerics of the matmul path.
#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

// logical (n, k) held by nibble p of logical word a of lane `lane` of a cdna4 tile (see awq_device.hpp)
__device__ __forceinline__ void cdna4_nibble_nk(int lane, int a, int p, int& n_in_slab, int& k_in_group) {
  const int g = lane >> 4, nq = (lane >> 2) & 3, r = lane & 3, i = p & 3, hi = p >> 2;
  n_in_slab = 4 * nq + 2 * (i & 1) + hi;
  k_in_group = 32 * a + 8 * g + 4 * (i >> 1) + r;
}

// one thread per (tile, lane): gather the 32 integers of the lane's four logical words, fold, store 3 words
__global__ void pack_w3_kernel(const uint8_t* __restrict__ q, u32* __restrict__ qw3, int N, int K) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nit = K >> 7;
  if (t >= (size_t)(N >> 4) * nit * 64) return;
  const int lane = (int)(t & 63);
  const size_t tile = t >> 6;
  const int nb = (int)(tile / nit), kg = (int)(tile % nit);
  u32 w[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    u32 v = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      int nn, kk;
      cdna4_nibble_nk(lane, a, p, nn, kk);
      v |= (u32)(q[(size_t)(nb * 16 + nn) * K + kg * 128 + kk] & 7u) << (4 * p);
    }
    w[a] = v;
  }
  u32 o[3];
  w3_fold(u32x4{w[0], w[1], w[2], w[3]}, o);
  u32* dst = qw3 + tile * 192 + lane * 3;
  dst[0] = o[0];
  dst[1] = o[1];
  dst[2] = o[2];
}

__global__ void unpack_w3_kernel(const u32* __restrict__ qw3, uint8_t* __restrict__ out, int N, int K) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int nit = K >> 7;
  if (t >= (size_t)(N >> 4) * nit * 64) return;
  const int lane = (int)(t & 63);
  const size_t tile = t >> 6;
  const int nb = (int)(tile / nit), kg = (int)(tile % nit);
  const u32* src = qw3 + tile * 192 + lane * 3;
  const u32x4 w = w3_expand(src[0], src[1], src[2]);
  const u32 ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      int nn, kk;
      cdna4_nibble_nk(lane, a, p, nn, kk);
      out[(size_t)(nb * 16 + nn) * K + kg * 128 + kk] = (uint8_t)((ws[a] >> (4 * p)) & 7u);
    }
}

// one wave per tile, through the matrix-core dequant of the GEMV
template <typename DT>
__global__ __launch_bounds__(64) void dequant_w3_kernel(const u32* __restrict__ qw3, const uint16_t* __restrict__ scales,
                                                         const uint16_t* __restrict__ zeros, uint16_t* __restrict__ out,
                                                         int N, int K) {
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  const int nit = K >> 7;
  const int nb = blockIdx.x / nit, kg = blockIdx.x % nit;
  using vec8 = typename DT::vec8;
  Cdna4DequantT<DT> cd;
  cd.init(lane, 0x00070007u);
  const u32* src = qw3 + w3_tile_word(nb, kg, nit) + lane * 3;
  const u32x4 w = w3_expand(src[0], src[1], src[2]);
  const int n = nb * 16 + c;
  vec8 op[4];
  cd.tile(w, scales[(size_t)kg * N + n], zeros[(size_t)kg * N + n], op);
#pragma unroll
  for (int a = 0; a < 4; ++a)
    *reinterpret_cast<vec8*>(out + (size_t)n * K + (size_t)kg * 128 + 32 * a + 8 * g) = op[a];
}

// w3c tiles -> cdna4 W4 tiles of the same integers (prefill: the W4 GEMM kernels then run unchanged)
__global__ void expand_w3_kernel(const u32* __restrict__ qw3, u32* __restrict__ qw4, size_t lanes) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= lanes) return;
  const u32* src = qw3 + t * 3;
  const u32x4 w = w3_expand(src[0], src[1], src[2]);
  *reinterpret_cast<u32x4*>(qw4 + t * 4) = u32x4{w.x & 0x77777777u, w.y & 0x77777777u, w.z & 0x77777777u, w.w};
}

static inline unsigned nblk3(size_t n, unsigned b) { return (unsigned)((n + b - 1) / b); }

int launch_pack_w3(const void* q_u8, void* qw3, int n, int k, hipStream_t st) {
  const size_t lanes = (size_t)(n / 16) * (k / 128) * 64;
  hipLaunchKernelGGL(pack_w3_kernel, dim3(nblk3(lanes, 256)), dim3(256), 0, st, (const uint8_t*)q_u8, (u32*)qw3, n, k);
  return 0;
}
int launch_unpack_w3(const void* qw3, void* out_u8, int n, int k, hipStream_t st) {
  const size_t lanes = (size_t)(n / 16) * (k / 128) * 64;
  hipLaunchKernelGGL(unpack_w3_kernel, dim3(nblk3(lanes, 256)), dim3(256), 0, st, (const u32*)qw3, (uint8_t*)out_u8, n, k);
  return 0;
}
int launch_dequant_w3(const void* qw3, const void* s, const void* z, void* out, int n, int k, int dtype, hipStream_t st) {
  auto kern = dtype == 0 ? dequant_w3_kernel<F16> : dequant_w3_kernel<BF16>;
  hipLaunchKernelGGL(kern, dim3((n / 16) * (k / 128)), dim3(64), 0, st, (const u32*)qw3, (const uint16_t*)s,
                     (const uint16_t*)z, (uint16_t*)out, n, k);
  return 0;
}
int launch_expand_w3_to_cdna4(const void* qw3, void* qw4, int n, int k, hipStream_t st) {
  const size_t lanes = (size_t)(n / 16) * (k / 128) * 64;
  hipLaunchKernelGGL(expand_w3_kernel, dim3(nblk3(lanes, 256)), dim3(256), 0, st, (const u32*)qw3, (u32*)qw4, lanes);
  return 0;
}

}  // namespace awq
