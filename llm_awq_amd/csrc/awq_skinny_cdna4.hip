// Skinny GEMM on cdna4-interleaved weights, 9 <= M <= 64 per pass, bf16 and fp16 (gfx950): short prompts / chunk prefill / batched decode --
// the M range between the decode GEMV (awq_gemv_cdna4.hip) and the tiled prefill GEMM (awq_gemm_plan.hip), which the
// reference serves with gemm_w4a16_T1's 16/32-row tiles + split-K (gemm_cuda.cu:1155-1193).
//
// Still weight-stream bound (every packed byte is read once), but the activations are no longer negligible: a block
// owns NS 16-row slabs that share every x operand, WAVES waves split K in interleaved 128-k steps, and per step a wave
//   * stages its x slice (16 CB rows x 128 k) through a wave-private, XOR-swizzled LDS region (conflict-free
//     ds_read_b128 for the 16 x rows of an MFMA operand), prefetched one step ahead through registers;
//   * dequantises NS 1-KiB tiles on the matrix core (Cdna4Dequant) -- their packed words are prefetched one step ahead;
//   * issues NS x CB x 4 v_mfma_f32_16x16x32_bf16 (weights = A operand, 16 x rows = B operand), fp32 accumulation.
// Split-K partials are reduced through LDS in fp32; one rounding; bias fused.  Numerics as everywhere else.
#include <string.h>

#include <atomic>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

// the block's work for slab group `nb` on M <= 16 CB rows: shared by the plain kernel and the grouped (per-expert) kernel
// DQ 1: szp is the decode side buffer "sz_half" (f16-mantissa dequant form, Cdna4DequantH); EPI 2: QuantLlamaMLP's 8 + 8 interleaved
// gate / up slabs, out[m, N / 2] = silu(gate) * up (batched decode of 5..8 rows arrives here from launch_gemv_dma)
template <typename DT, int WAVES, int NS, int CB, int DQ = 0, int EPI = 0, int BITS = 4, int XU = 4 * CB>
__device__ __forceinline__ void skinny_cdna4_body(char* smem, const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                  const u32* __restrict__ szp, const uint16_t* __restrict__ bias,
                                                  uint16_t* __restrict__ out, int M, int N, int K, int nb, int f32out = 0, int k0 = 0, int kn = -1) {
  using vec8 = typename DT::vec8;
  constexpr int XB = 4 * CB;            // staging pieces per step: 4 x rows (1 KiB) each
  constexpr int XBYTES = 16 * CB * 256; // wave-private x region: 16 CB rows x 256 B
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int nit = K >> 7, nslab = N >> 4;
  char* xs = smem + wv * XBYTES;

  static_assert(BITS == 4 || DQ == 0, "the w3c tiles have no sz_half form");
  constexpr u32 kTile = BITS == 4 ? 1024u : 768u;  // bytes per 16 x 128 weight tile (BITS 3: the w3c tile, three words per lane)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(qw), 0, nslab * nit * (int)kTile, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(szp), 0, nslab * nit * 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, M * K * 2, 0x00020000);
  u32 slab_tile[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) slab_tile[s] = (u32)min(nb * NS + s, nslab - 1) * (u32)nit;
  const u32 wlane_b = BITS == 4 ? lane * 16u : lane * 12u, ilane_b = (u32)i * 4u;
  // staging piece b: LDS row r = 4b + g, slot i  <-  source row min(r, M-1), granule i ^ (r & 15)
  u32 xsrc_b[XB];
#pragma unroll
  for (int b = 0; b < XB; ++b) {
    const int r = 4 * b + g;
    xsrc_b[b] = ((u32)min(r, M - 1) * (u32)K + (u32)((i ^ (r & 15)) * 8)) * 2u;
  }

  Cdna4DequantT<DT> cd;
  Cdna4DequantH<DT> ch;
  if (DQ == 0) cd.init(lane, BITS == 4 ? 0x000F000Fu : 0x00070007u);
  else ch.init(lane);
  // the block's K part: k-steps [k0, k0 + kn) (kn < 0: all of K).  f32out 2 = a split-K partial: fp32 sums stored write-through (sc1) for the
  // block that arrives last at the slab group's ticket, which may sit on another XCD (skinny_splitk_kernel below)
  if (kn < 0) kn = nit;
  const int cnt = (kn - wv + WAVES - 1) / WAVES;  // this wave's steps: kg = k0 + wv + WAVES * t

  f32x4 acc[NS][CB];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[s][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 w[NS], xr[XB];
  u32 sz[NS];
  auto load_step = [&](int t) {
    const int kg = k0 + min(wv + WAVES * t, kn - 1);
#pragma unroll
    for (int b = 0; b < XU; ++b) xr[b] = __builtin_amdgcn_raw_buffer_load_b128(rx, xsrc_b[b], (u32)kg * 256u, 0);  // (XU < XB: batched decode of <= 4 XU rows -- the pieces past them are never staged, their MFMA columns never stored)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const u32 tidx = slab_tile[s] + (u32)kg;
      if (BITS == 4) {
        w[s] = __builtin_amdgcn_raw_buffer_load_b128(rw, wlane_b, tidx * 1024u, 2);  // aux 2 = nt
      } else {
        typedef u32 u32x3 __attribute__((ext_vector_type(3)));
        const u32x3 w3 = __builtin_amdgcn_raw_buffer_load_b96(rw, wlane_b, tidx * 768u, 2);
        w[s] = u32x4{w3.x, w3.y, w3.z, 0u};
      }
      sz[s] = __builtin_amdgcn_raw_buffer_load_b32(rs, ilane_b, tidx * 64u, 0);
    }
  };
  if (cnt > 0) load_step(0);
  for (int t = 0; t < cnt; ++t) {
    // this step's operands: x slice -> LDS (the previous step's reads of the region are retired: same wave, in order)
#pragma unroll
    for (int b = 0; b < XU; ++b) *reinterpret_cast<u32x4*>(xs + b * 1024 + lane * 16) = xr[b];
    u32x4 wc[NS];
    u32 szc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      wc[s] = w[s];
      szc[s] = sz[s];
    }
    if (t + 1 < cnt) load_step(t + 1);  // next step's packed words and x slice stream in under this step's math
    // x operands of the step, shared by the block's NS slabs: xo[c][a] = rows 16c .. 16c+15, k = 32a + 8g .. +8
    vec8 xo[CB][4];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const char* xrow = xs + (c * 16 + i) * 256;
#pragma unroll
      for (int a = 0; a < 4; ++a) xo[c][a] = *reinterpret_cast<const vec8*>(xrow + (((4 * a + g) ^ i) << 4));
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      vec8 op[4];
      if (DQ == 0) cd.tile_packed(BITS == 4 ? wc[s] : w3_expand(wc[s].x, wc[s].y, wc[s].z), szc[s], op);
      else ch.tile(wc[s], szc[s], op);
#pragma unroll
      for (int a = 0; a < 4; ++a)  // a outer: consecutive MFMAs hit different accumulators
#pragma unroll
        for (int c = 0; c < CB; ++c) acc[s][c] = DT::mfma(op[a], xo[c][a], acc[s][c]);
    }
  }

  // ---- split-K reduction across the block's waves (fp32), x regions re-used.  acc[s][c][r] = C[n = 4g + r][m = 16c + i] ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);  // [wave][blk = s * CB + c][r][lane]
  constexpr int NBLK = NS * CB;
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int c = 0; c < CB; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[((wv * NBLK + s * CB + c) * 4 + r) * 64 + lane] = acc[s][c][r];
  __syncthreads();
  for (int blk = wv; blk < NBLK; blk += WAVES) {
    const int s = blk / CB, c = blk - s * CB;
    const int slab = nb * NS + s, m = 16 * c + i;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < WAVES; ++q) t += red[((q * NBLK + blk) * 4 + r) * 64 + lane];
      v[r] = t;
    }
    auto to_f = [](uint16_t b) { return DT::to_float(b); };
    if (EPI == 2) {
      // rows 0..7 of a slab are gate rows 8 slab .. + 7, rows 8..15 the matching up rows: lane (g < 2) pairs with lane + 32
      // (fused_mlp.py:79-82: c = F.silu(gate_output) * up_output, every op rounded to T -- as the decode kernel's EPI 2)
      float u[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < WAVES; ++q) t += red[((q * NBLK + blk) * 4 + r) * 64 + ((lane + 32) & 63)];
        u[r] = t;
      }
      if (slab < nslab && m < M && g < 2) {
        uint16_t o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float gt = to_f(DT::from_float(v[r])), up = to_f(DT::from_float(u[r]));
          const float sl = to_f(DT::from_float(silu_f32(gt)));
          o[r] = DT::from_float(sl * up);
        }
        *reinterpret_cast<u32x2*>(out + (size_t)m * (N >> 1) + slab * 8 + 4 * g) =
            u32x2{(u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16)};
      }
    } else if (f32out) {
      // K shard of a tensor-parallel row split (awq_w4a16_partial_cdna4): fp32 sums, unrounded, no bias, out = float [M, N]
      if (slab < nslab && m < M) {
        float* dst = reinterpret_cast<float*>(out) + (size_t)m * N + slab * 16 + 4 * g;
        const f32x4 pv = f32x4{v[0], v[1], v[2], v[3]};
        if (f32out == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(pv) : "memory");  // (s_nop: as in awq_gemm_v6.hip -- a > 64-bit store the compiler cannot see; its data registers must survive two more issue slots)
        else *reinterpret_cast<f32x4*>(dst) = pv;
      }
    } else if (slab < nslab && m < M) {
      const int nn = slab * 16 + 4 * g;
      uint16_t o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = DT::from_float(v[r]);
        if (bias != nullptr) o[r] = DT::from_float(to_f(o[r]) + to_f(bias[nn + r]));  // `out + self.bias` in T
      }
      *reinterpret_cast<u32x2*>(out + (size_t)m * N + nn) = u32x2{(u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16)};
    }
  }
}

template <typename DT, int WAVES, int NS, int CB, int DQ = 0, int EPI = 0, int BITS = 4, int XU = 4 * CB>
__global__ __launch_bounds__(64 * WAVES) void skinny_cdna4_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                                   const u32* __restrict__ szp,
                                                                   const uint16_t* __restrict__ bias,
                                                                   uint16_t* __restrict__ out, int M, int N, int K, int f32out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  skinny_cdna4_body<DT, WAVES, NS, CB, DQ, EPI, BITS, XU>(smem, x, qw, szp, bias, out, M, N, K, blockIdx.x, EPI == 0 ? f32out : 0);
}

// Grouped (per-expert) form for MoE batches between the grouped GEMV (<= 8 sorted rows) and the grouped prefill GEMM
// (>= 256): block = (expert, slab group); expert e owns rows [offsets[e], offsets[e+1]) of the sorted x / out and the e-th
// slice of the stacked cdna4 weights / packed scales; more than 16 CB rows run as further passes over the expert's slabs.
// tail > 0 (round 5): the TAIL PASS behind the grouped prefill GEMM (launch_moe_gemm_cdna4_v6 with the same threshold): only the rows of an expert's last
// partial 256-row tile, and only when their number lies in [tail_lo, tail) -- the tile launch leaves exactly the remainders below `tail` out.  The grid is a fixed
// number of blocks (what fits the chip at once) that stride over the work items (qualifying expert, slab group) found from the device-side offsets: a grid of
// experts x groups blocks of which most exit at once costs more in block dispatch than the tails cost to compute (7168 blocks for Mixtral's w1 / w3).
// EPI 2: the expert's w1 / w3 rows are interleaved 8 + 8 per slab (N = 2 x ffn), out [rows, N / 2] = silu(w1 x) * (w3 x)
constexpr int kMoeTailMaxExperts = 64;  // (the tail pass scans the offsets per work item: keep that loop short)
template <typename DT, int WAVES, int NS, int CB, int EPI = 0>
__global__ __launch_bounds__(64 * WAVES) void moe_skinny_cdna4_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                                       const u32* __restrict__ szp,
                                                                       const int* __restrict__ offsets,
                                                                       uint16_t* __restrict__ out, int N, int K, int groups, int experts, int tail_lo, int tail) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const size_t et = (size_t)(N >> 4) * (K >> 7);             // tiles per expert
  const int ncol = EPI == 2 ? (N >> 1) : N;
  if (tail > 0) {
    auto qualifies = [&](int e2) {
      const int rem = (offsets[e2 + 1] - offsets[e2]) & 255;
      return rem >= tail_lo && rem < tail;
    };
    int nq = 0;
    for (int e2 = 0; e2 < experts; ++e2) nq += qualifies(e2) ? 1 : 0;  // (block-uniform scalar loop over the offsets)
    for (int item = blockIdx.x; item < nq * groups; item += gridDim.x) {
      const int q = item / groups, nb = item - q * groups;
      int e = 0;
      for (int seen = 0; e < experts; ++e) {  // the q-th qualifying expert (no LDS list: the kernel's LDS budget is the staging region's, up to 128 KiB + opt-in)
        if (qualifies(e)) {
          if (seen == q) break;
          ++seen;
        }
      }
      const int lo = offsets[e], cnt = offsets[e + 1] - lo, rem = cnt & 255;
      skinny_cdna4_body<DT, WAVES, NS, CB, 0, EPI>(smem, x + (size_t)(lo + cnt - rem) * K, qw + (size_t)e * et * 256, szp + (size_t)e * et * 16, nullptr,
                                                   out + (size_t)(lo + cnt - rem) * ncol, rem, N, K, nb);
      __syncthreads();  // the item's reduction reads of the LDS region are done before the next item stages x into it
    }
    return;
  }
  const int e = blockIdx.x / groups, nb = blockIdx.x - e * groups;
  const int row0 = offsets[e], m_e = offsets[e + 1] - row0;  // block-uniform
  for (int r0 = 0; r0 < m_e; r0 += 16 * CB) {
    if (r0 > 0) __syncthreads();  // the previous pass's reduction reads of the LDS region are done
    skinny_cdna4_body<DT, WAVES, NS, CB, 0, EPI>(smem, x + (size_t)(row0 + r0) * K, qw + (size_t)e * et * 256, szp + (size_t)e * et * 16, nullptr,
                                                 out + (size_t)(row0 + r0) * ncol, min(m_e - r0, 16 * CB), N, K, nb);
  }
}

template <typename DT, int WAVES, int NS, int CB, int DQ = 0, int EPI = 0, int BITS = 4, int XU = 4 * CB>
static void launch_skinny(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                          hipStream_t st, int f32out = 0) {
  const size_t xbytes = (size_t)WAVES * 16 * CB * 256, rbytes = (size_t)WAVES * NS * CB * 1024;
  const size_t smem = xbytes > rbytes ? xbytes : rbytes;
  auto kern = skinny_cdna4_kernel<DT, WAVES, NS, CB, DQ, EPI, BITS, XU>;
  static LdsOptIn optin;  // per (kernel instantiation, device)
  if (smem > 64 * 1024) optin.ensure(reinterpret_cast<const void*>(kern));
  const int nslab = n / 16;
  hipLaunchKernelGGL(kern, dim3((nslab + NS - 1) / NS), dim3(64 * WAVES), smem, st, (const uint16_t*)x, (const u32*)qw,
                     (const u32*)szp, (const uint16_t*)bias, (uint16_t*)out, m, n, k, f32out);
}

// ---- K split ACROSS blocks for the launches whose slab groups leave half the chip idle (N = 4096 at 17..64 rows: 128 blocks of two slabs) ----
// grid = (slab groups, KS): block (nb, p) sums the k-steps [p nit / KS, (p + 1) nit / KS) of its group and stores the fp32 sums write-through into
// part p of the workspace ([KS][M][N] floats); after its stores have been acknowledged it takes a ticket of the group; the block that draws the
// last one adds the KS parts IN PART ORDER (the result does not depend on which block came last), rounds once, adds the bias and puts the
// ticket word back to 0 -- the role of the reference's split_k_iters + Semaphore (gemm_cuda.cu:546-619), without a second launch.  Ticket
// words: splitk_ticket_words (awq_midm_cdna4.hip) -- zero-initialised, library-owned, and private to ONE launch at a time (the lane of the launch's stream;
// words of its own for a launch recorded during a capture).
template <typename DT, int WAVES, int NS, int CB, int DQ = 0>
__global__ __launch_bounds__(64 * WAVES) void skinny_splitk_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                                   const u32* __restrict__ szp, const uint16_t* __restrict__ bias,
                                                                   uint16_t* __restrict__ out, float* __restrict__ parts, u32* __restrict__ tickets,
                                                                   int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nb = blockIdx.x, KS = gridDim.y, p = blockIdx.y, kn = (K >> 7) / KS;
  skinny_cdna4_body<DT, WAVES, NS, CB, DQ, 0>(smem, x, qw, szp, nullptr, reinterpret_cast<uint16_t*>(parts + (size_t)p * M * N), M, N, K, nb, 2, p * kn, kn);
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");  // this thread's write-through stores have been acknowledged
  __syncthreads();                                     // ... and every other thread's; the reduction's LDS reads are done as well
  u32* bc = reinterpret_cast<u32*>(smem);
  // (relaxed: the release is the acknowledged write-through stores above, the acquire the sc1 loads below -- an acq_rel atomic would add an L2
  // write-back and an L2 invalidate of the whole XCD around it)
  if (threadIdx.x == 0) bc[0] = __hip_atomic_fetch_add(tickets + nb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (bc[0] != (u32)(KS - 1)) return;
  if (threadIdx.x == 0) __hip_atomic_store(tickets + nb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // all KS tickets are drawn: ready for the next launch
  // the group's tile: M rows x NS * 16 columns, four columns (16 B) per thread and pass; parts read at agent scope (sc1: the writers may sit on other XCDs)
  constexpr int QC = NS * 4;  // column quads per row
  const int nslab = N >> 4;
  for (int idx = threadIdx.x; idx < M * QC; idx += 64 * WAVES) {
    const int m = idx / QC, q = idx - m * QC, col = nb * NS * 16 + 4 * q;
    if (col >= nslab * 16) continue;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < KS; p0 += 4) {  // up to four parts in flight, summed in part order
      // (every load unconditional, the part index clamped: a load under a branch would make hipcc merge its destination with the untaken
      // path's value BEFORE the wait below -- a copy of registers the load has not written yet)
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* src = parts + ((size_t)min(p0 + j, KS - 1) * M + m) * N + col;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[j]) : "v"(src) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (p0 + j < KS) acc += v[j];
    }
    uint16_t o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[r] = DT::from_float(acc[r]);
      if (bias != nullptr) o[r] = DT::from_float(DT::to_float(o[r]) + DT::to_float(bias[col + r]));  // `out + self.bias` in T
    }
    *reinterpret_cast<u32x2*>(out + (size_t)m * N + col) = u32x2{(u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16)};
  }
}

namespace {
constexpr int kTicketGroups = 512;  // slab groups of a split launch (its ticket words: splitk_ticket_words, awq_midm_cdna4.hip -- private to the launch)
int g_skinny_ks = -1;
int g_skinny_deep64 = 32;  // knob skinny_deep64: the k-step count from which a launch of at most ONE slab per CU takes the 16-wave shape (0 = only from 96 steps, as rounds 3 - 5) for batched decode (<= 8 rows; at 12 / 16 rows it measured 0.6-0.9 % slower): o_proj -2 ... -3.5 %, Llama-2-7B down_proj (86 steps) 8.5 -> 6.6 us
int g_skinny_xu = 1;  // knob skinny_xu: 0 = batched decode stages all sixteen x rows per step as rounds 3 - 5 did
// seven slabs per block make ONE round of one block per CU (Llama-3-8B gate/up: 1792 slabs -> 256 blocks)?  Then the x slices are read by that many blocks
// instead of 1.75 - 3.5 times as many (profiles/r05_skinny_splitk.txt)
bool seven_slab_round(int nslab) {
  const int groups = (nslab + 6) / 7, cus = device_cu_count();
  return groups <= cus + 16 && groups >= cus * 9 / 10;
}  // knob skinny_splitk: -1 = by shape, 0 = off, 2 / 4 = force that many K parts where the shape allows it
}  // namespace

int skinny_tune_set(const char* key, int value) {
  if (!strcmp(key, "skinny_splitk")) g_skinny_ks = value;
  else if (!strcmp(key, "skinny_xu")) g_skinny_xu = value;
  else if (!strcmp(key, "skinny_deep64")) g_skinny_deep64 = value;
  else return -1;
  return 0;
}

// K parts of the skinny launch for (m rows per pass, n, k): 1 = unsplit.  Only the two-slab configurations of launch_skinny_64 (17..64 rows, fewer
// than 512 slabs) split: their (n / 32) blocks of eight waves hold one CU each
int skinny_splitk_parts(int m, int n, int k) {
  const int nslab = n / 16, nit = k / 128, cus = device_cu_count(), groups = (nslab + 1) / 2;
  // (qkv, 384 slabs: three slabs per block x two parts = 256 blocks measured 6 % SLOWER than its 192 unsplit two-slab blocks: not built in)
  if (g_skinny_ks == 0 || m <= 16 || m > 64 || nslab >= 512 || groups > kTicketGroups) return 1;
  // by shape (profiles/r05_skinny_splitk.txt): 33..64 rows, where the unsplit launch runs two slabs per block and fills half the chip, from K = 4096
  // (o_proj -12 / -16 % at 48 / 64 rows, down_proj -30 %); 17..32 rows, where the unsplit launch runs ONE slab per block on every CU, only against a
  // long K (down_proj -15 ... -18 %; o_proj loses 8 %: half the x traffic does not pay for the ticket round trip there).  The knob forces a part count
  int ks = 1;
  if (g_skinny_ks > 0) ks = g_skinny_ks;
  else if (groups * 2 <= cus + 16 && (m > 32 ? nit >= 32 : nit >= 64)) ks = 2;
  while (ks > 1 && ((nit % ks) != 0 || nit / ks < 8)) ks >>= 1;  // every wave of a part keeps at least one k-step
  return ks < 1 ? 1 : ks;
}
// fp32 scratch of the split launch ([parts][rows of a pass][n]); 0 = the launch does not split
size_t skinny_splitk_workspace_bytes(int m, int n, int k) {
  if (m < 9 || m > 255) return 0;
  const int chunks = (m + 63) / 64, rows = (m + chunks - 1) / chunks;
  const int ks = skinny_splitk_parts(rows, n, k);
  return ks > 1 ? (size_t)ks * rows * n * sizeof(float) : 0;
}

template <typename DT, int WAVES, int NS, int CB>
static void launch_skinny_splitk(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int ks, float* parts,
                                 u32* tickets, hipStream_t st) {
  const size_t xbytes = (size_t)WAVES * 16 * CB * 256, rbytes = (size_t)WAVES * NS * CB * 1024;
  const size_t smem = xbytes > rbytes ? xbytes : rbytes;
  auto kern = skinny_splitk_kernel<DT, WAVES, NS, CB>;
  static LdsOptIn optin;
  if (smem > 64 * 1024) optin.ensure(reinterpret_cast<const void*>(kern));
  const int nslab = n / 16;
  hipLaunchKernelGGL(kern, dim3((nslab + NS - 1) / NS, ks), dim3(64 * WAVES), smem, st, (const uint16_t*)x, (const u32*)qw, (const u32*)szp,
                     (const uint16_t*)bias, (uint16_t*)out, parts, tickets, m, n, k);
}

template <typename DT, int EPI = 0>
static int launch_skinny_64(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                            hipStream_t st, int f32out, void* ws, size_t ws_bytes);

// 9 <= m <= 255, cdna4 layout, packed sz required.  Returns -1 if unsupported.  65 <= m <= 255 (below the 256-row tile of
// the prefill GEMM) runs as row chunks of <= 64: the weights are re-streamed per chunk, which still beats the 128 x 128
// kernel's long serial K loop on one wave of tiles (measured: profiles/r01_skinny_sweep.txt) except for very wide N.
int launch_skinny_cdna4(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                        int dtype, hipStream_t st, int f32out, void* ws, size_t ws_bytes) {
  if (!szp || m < 1 || m > 255 || (n % 16) != 0 || (k % 128) != 0 || (size_t)m * (size_t)k >= (1ull << 31) || (f32out && bias)) return -1;
  if (m > 128 && n >= 16384) return -1;
  const int chunks = (m + 63) / 64, rows = (m + chunks - 1) / chunks;
  for (int r0 = 0; r0 < m; r0 += rows) {
    const int mr = m - r0 < rows ? m - r0 : rows;
    const uint16_t* xr = static_cast<const uint16_t*>(x) + (size_t)r0 * k;
    void* orow = static_cast<char*>(out) + (size_t)r0 * n * (f32out ? 4 : 2);
    // (the row chunks run one after the other on the stream: they share the optional split-K scratch)
    if (dtype == 0) launch_skinny_64<F16>(xr, qw, szp, bias, orow, mr, n, k, st, f32out, ws, ws_bytes);
    else launch_skinny_64<BF16>(xr, qw, szp, bias, orow, mr, n, k, st, f32out, ws, ws_bytes);
  }
  return 0;
}

template <typename DT, int EPI>
static int launch_skinny_64(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                            hipStream_t st, int f32out, void* ws, size_t ws_bytes) {
  const int nslab = n / 16;
  // half-empty chip (N = 4096 at 17..64 rows): the K split across blocks, when the caller brought the scratch and the ticket array exists
  const int ks = skinny_splitk_parts(m, n, k);
  if (EPI == 0 && ks > 1 && !f32out && ws != nullptr && (reinterpret_cast<uintptr_t>(ws) & 15) == 0 && ws_bytes >= (size_t)ks * m * n * sizeof(float)) {
    u32* tk = splitk_ticket_words(st, (nslab + 1) / 2);
    if (tk != nullptr) {
      float* parts = static_cast<float*>(ws);
      if constexpr (EPI == 0) {
        if (m <= 32) launch_skinny_splitk<DT, 8, 2, 2>(x, qw, szp, bias, out, m, n, k, ks, parts, tk, st);
        else if (m <= 48) launch_skinny_splitk<DT, 8, 2, 3>(x, qw, szp, bias, out, m, n, k, ks, parts, tk, st);
        else launch_skinny_splitk<DT, 8, 2, 4>(x, qw, szp, bias, out, m, n, k, ks, parts, tk, st);
        return 0;
      }
    }
  }
  if (m <= 16) {
    if (nslab >= 1024 && seven_slab_round(nslab)) launch_skinny<DT, 8, 7, 1, 0, EPI>(x, qw, szp, bias, out, m, n, k, st, f32out);  // (-2 % on the 7-row decode leg, -4 % at 16 rows)
    else if (nslab >= 1024) launch_skinny<DT, 8, 2, 1, 0, EPI>(x, qw, szp, bias, out, m, n, k, st, f32out);
    else if (k / 128 >= 96) launch_skinny<DT, 16, 1, 1, 0, EPI>(x, qw, szp, bias, out, m, n, k, st, f32out);
    else launch_skinny<DT, 8, 1, 1, 0, EPI>(x, qw, szp, bias, out, m, n, k, st, f32out);
  } else if (m <= 32) {
    if (nslab >= 512) launch_skinny<DT, 8, 2, 2, 0, EPI>(x, qw, szp, bias, out, m, n, k, st, f32out);
    else launch_skinny<DT, 8, 1, 2, 0, EPI>(x, qw, szp, bias, out, m, n, k, st, f32out);
  } else if (nslab >= 512 && seven_slab_round(nslab)) {
    // seven slabs per block where that is ONE round of one block per CU (Llama-3-8B gate/up: 1792 slabs -> 256 blocks): the x slices are read by
    // 256 blocks instead of 448 -- -7 % at 40 / 48 rows, -13 % at 64 and 128 (profiles/r05_skinny_splitk.txt); a two-deep prefetch of the packed words
    // for its single wave per SIMD measured 3 % slower again
    if (m <= 48) launch_skinny<DT, 4, 7, 3, 0, EPI>(x, qw, szp, bias, out, m, n, k, st, f32out);
    else launch_skinny<DT, 4, 7, 4, 0, EPI>(x, qw, szp, bias, out, m, n, k, st, f32out);
  } else if (m <= 48) {
    if (nslab >= 512) launch_skinny<DT, 4, 4, 3, 0, EPI>(x, qw, szp, bias, out, m, n, k, st, f32out);
    else launch_skinny<DT, 8, 2, 3, 0, EPI>(x, qw, szp, bias, out, m, n, k, st, f32out);
  } else {
    if (nslab >= 512) launch_skinny<DT, 4, 4, 4, 0, EPI>(x, qw, szp, bias, out, m, n, k, st, f32out);
    else launch_skinny<DT, 8, 2, 4, 0, EPI>(x, qw, szp, bias, out, m, n, k, st, f32out);
  }
  return 0;
}

// QuantLlamaMLP's interleaved gate / up stack (EPI 2: out [m, n2 / 2] = silu(gate) * up) for 9 <= m <= 64 rows: ONE weight pass with the x slices in
// registers, instead of a 256-row tile of the prefill GEMM masked down to m rows (awq_w4a16_mlp_gate_up_forward_cdna4).  Returns -1 if unsupported.
int launch_skinny_gate_up(const void* x, const void* qw, const void* szp, void* out, int m, int n2, int k, int dtype, hipStream_t st) {
  if (!szp || m < 9 || m > 64 || (n2 % 32) != 0 || (k % 128) != 0 || (size_t)m * (size_t)k >= (1ull << 31)) return -1;
  if (dtype == 0) return launch_skinny_64<F16, 2>(x, qw, szp, nullptr, out, m, n2, k, st, 0, nullptr, 0);
  return launch_skinny_64<BF16, 2>(x, qw, szp, nullptr, out, m, n2, k, st, 0, nullptr, 0);
}

// The same kernel on w3c tiles (BITS 3: three packed words per lane, the fourth rebuilt with six VALU operations, awq_device.hpp w3_expand) for 9 <= m <= 64
// rows of a 3-bit layer -- prompts / batched decode that would otherwise pay a 256-row tile of the prefill GEMM masked down to m rows.  epi 0 (bias fused)
// or 2 (QuantLlamaMLP's interleaved pair, out [m, n / 2]).  A smaller set of block shapes than W4's (no K split across blocks).  Returns -1 if unsupported.
template <typename DT, int EPI>
static void launch_skinny_w3_dt(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, hipStream_t st, int f32out) {
  const int nslab = n / 16;
  const bool wide = nslab >= 512;
  if (m <= 16) {
    if (wide) launch_skinny<DT, 8, 2, 1, 0, EPI, 3>(x, qw, szp, bias, out, m, n, k, st, f32out);
    else launch_skinny<DT, 8, 1, 1, 0, EPI, 3>(x, qw, szp, bias, out, m, n, k, st, f32out);
  } else if (m <= 32) {
    if (wide) launch_skinny<DT, 8, 2, 2, 0, EPI, 3>(x, qw, szp, bias, out, m, n, k, st, f32out);
    else launch_skinny<DT, 8, 1, 2, 0, EPI, 3>(x, qw, szp, bias, out, m, n, k, st, f32out);
  } else {  // 33 .. 64 rows: four column blocks (the rows past m are masked)
    if (wide) launch_skinny<DT, 4, 4, 4, 0, EPI, 3>(x, qw, szp, bias, out, m, n, k, st, f32out);
    else launch_skinny<DT, 8, 2, 4, 0, EPI, 3>(x, qw, szp, bias, out, m, n, k, st, f32out);
  }
}
// f32out (epi 0 only, no bias): out = float [m, n], the unrounded sums of a tensor-parallel row split's K shard (awq_w3a16_partial)
int launch_skinny_w3(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int epi, int dtype, hipStream_t st,
                     int f32out) {
  if (!szp || m < 9 || m > 64 || (n % (epi == 2 ? 32 : 16)) != 0 || (k % 128) != 0 || (epi != 0 && epi != 2) || (epi == 2 && bias) ||
      (f32out && (epi != 0 || bias)) || (size_t)m * (size_t)k >= (1ull << 31))
    return -1;
  if (dtype == 0) {
    if (epi == 2) launch_skinny_w3_dt<F16, 2>(x, qw, szp, nullptr, out, m, n, k, st, 0);
    else launch_skinny_w3_dt<F16, 0>(x, qw, szp, bias, out, m, n, k, st, f32out);
  } else {
    if (epi == 2) launch_skinny_w3_dt<BF16, 2>(x, qw, szp, nullptr, out, m, n, k, st, 0);
    else launch_skinny_w3_dt<BF16, 0>(x, qw, szp, bias, out, m, n, k, st, f32out);
  }
  return 0;
}

// Batched decode (launch_gemv_dma hands over the row counts where one weight pass with x in registers beats its LDS-DMA staging of
// m x K x 2 bytes per slab, profiles/r03_decode_m_sweep.txt): m <= 16 rows, one 16-row x block; szfmt 1 = szp is "sz_half";
// epi 0 (bias fused) or 2 (8 + 8 interleaved gate / up pair -> silu(gate) * up, out [m, n / 2]).  Returns -1 if unsupported.
int launch_skinny_decode(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int epi,
                         int dtype, int szfmt, hipStream_t st, int f32out) {
  if (!szp || m < 1 || m > 16 || (n % 16) != 0 || (k % 128) != 0 || (epi != 0 && epi != 2) || (epi == 2 && bias) || (f32out && (epi || bias))) return -1;
  const bool wide = n / 16 >= 1024, deep = !wide && (k / 128 >= 96 || (g_skinny_deep64 && m <= 8 && k / 128 >= g_skinny_deep64 && n / 16 <= device_cu_count()));  // deep: 16 waves split a long K (down_proj: 112 steps; knob skinny_deep64 = the step count from which that also holds where a CU holds one slab)
#define AWQ_SD(DT_, DQ_, EPI_)                                                                    \
  {                                                                                               \
    if (m <= 4 && g_skinny_xu) {  /* (Llama-3-70B's long-K launches hand over from two / three rows: one piece per step) */ \
      if (wide && seven_slab_round(n / 16)) launch_skinny<DT_, 8, 7, 1, DQ_, EPI_, 4, 1>(x, qw, szp, bias, out, m, n, k, st, f32out); \
      else if (wide) launch_skinny<DT_, 8, 2, 1, DQ_, EPI_, 4, 1>(x, qw, szp, bias, out, m, n, k, st, f32out);       \
      else if (deep) launch_skinny<DT_, 16, 1, 1, DQ_, EPI_, 4, 1>(x, qw, szp, bias, out, m, n, k, st, f32out); \
      else launch_skinny<DT_, 8, 1, 1, DQ_, EPI_, 4, 1>(x, qw, szp, bias, out, m, n, k, st, f32out);            \
      return 0;                                                                                   \
    }                                                                                             \
    if (m <= 8 && g_skinny_xu) {  /* batched decode: rows 8 .. 15 of the x block are never staged (two of the four pieces per step) */ \
      if (wide && seven_slab_round(n / 16)) launch_skinny<DT_, 8, 7, 1, DQ_, EPI_, 4, 2>(x, qw, szp, bias, out, m, n, k, st, f32out); \
      else if (wide) launch_skinny<DT_, 8, 2, 1, DQ_, EPI_, 4, 2>(x, qw, szp, bias, out, m, n, k, st, f32out);       \
      else if (deep) launch_skinny<DT_, 16, 1, 1, DQ_, EPI_, 4, 2>(x, qw, szp, bias, out, m, n, k, st, f32out); \
      else launch_skinny<DT_, 8, 1, 1, DQ_, EPI_, 4, 2>(x, qw, szp, bias, out, m, n, k, st, f32out);            \
      return 0;                                                                                   \
    }                                                                                             \
    if (m <= 12 && g_skinny_xu) {  /* 9 .. 12 rows: three of the four pieces */ \
      if (wide && seven_slab_round(n / 16)) launch_skinny<DT_, 8, 7, 1, DQ_, EPI_, 4, 3>(x, qw, szp, bias, out, m, n, k, st, f32out); \
      else if (wide) launch_skinny<DT_, 8, 2, 1, DQ_, EPI_, 4, 3>(x, qw, szp, bias, out, m, n, k, st, f32out);       \
      else if (deep) launch_skinny<DT_, 16, 1, 1, DQ_, EPI_, 4, 3>(x, qw, szp, bias, out, m, n, k, st, f32out); \
      else launch_skinny<DT_, 8, 1, 1, DQ_, EPI_, 4, 3>(x, qw, szp, bias, out, m, n, k, st, f32out);            \
      return 0;                                                                                   \
    }                                                                                             \
    if (wide && seven_slab_round(n / 16)) launch_skinny<DT_, 8, 7, 1, DQ_, EPI_>(x, qw, szp, bias, out, m, n, k, st, f32out); \
    else if (wide) launch_skinny<DT_, 8, 2, 1, DQ_, EPI_>(x, qw, szp, bias, out, m, n, k, st, f32out);       \
    else if (deep) launch_skinny<DT_, 16, 1, 1, DQ_, EPI_>(x, qw, szp, bias, out, m, n, k, st, f32out); \
    else launch_skinny<DT_, 8, 1, 1, DQ_, EPI_>(x, qw, szp, bias, out, m, n, k, st, f32out);            \
    return 0;                                                                                     \
  }
#define AWQ_SD_DT(DT_)                          \
  if (szfmt == 1) {                             \
    if (epi == 2) AWQ_SD(DT_, 1, 2)             \
    AWQ_SD(DT_, 1, 0)                           \
  }                                             \
  if (epi == 2) AWQ_SD(DT_, 0, 2)               \
  AWQ_SD(DT_, 0, 0)
  if (dtype == 0) {
    AWQ_SD_DT(F16)
  }
  AWQ_SD_DT(BF16)
#undef AWQ_SD_DT
#undef AWQ_SD
}

template <typename DT, int WAVES, int NS, int CB, int EPI = 0>
static void launch_moe_skinny(const void* x, const void* qw, const void* szp, const void* offsets, void* out, int experts, int n, int k,
                              hipStream_t st, int tail_lo = 0, int tail = 0) {
  const size_t xbytes = (size_t)WAVES * 16 * CB * 256, rbytes = (size_t)WAVES * NS * CB * 1024;
  const size_t smem = xbytes > rbytes ? xbytes : rbytes;
  auto kern = moe_skinny_cdna4_kernel<DT, WAVES, NS, CB, EPI>;
  static LdsOptIn optin;  // per (kernel instantiation, device)
  if (smem > 64 * 1024) optin.ensure(reinterpret_cast<const void*>(kern));
  const int groups = (n / 16 + NS - 1) / NS;
  int blocks = experts * groups;
  if (tail > 0) {  // tail pass: what the chip holds at once (LDS-limited), striding over the qualifying (expert, slab group) items
    const int per_cu = smem <= 32 * 1024 ? 4 : (smem <= 64 * 1024 ? 2 : 1);
    const int cap = device_cu_count() * per_cu;
    if (blocks > cap) blocks = cap;
  }
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WAVES), smem, st, (const uint16_t*)x, (const u32*)qw, (const u32*)szp,
                     (const int*)offsets, (uint16_t*)out, n, k, groups, experts, tail_lo, tail);
}

// the tail pass of the grouped prefill GEMM: every expert's last partial row tile of fewer than `tail` (<= 64) rows, one weight stream of that expert
// (blocks of experts without such a tail exit at once); epi 0 / 2 as the tile launch.  Returns -1 if unsupported.
int launch_moe_skinny_tail_cdna4(const void* x, const void* qw, const void* szp, const void* offsets, void* out, int experts, int n, int k,
                                 int dtype, hipStream_t st, int epi, int tail) {
  if (!szp || experts < 1 || experts > kMoeTailMaxExperts || (n % 16) != 0 || (k % 128) != 0 || tail < 1 || tail > 64 || (epi != 0 && epi != 2) ||
      (epi == 2 && (n % 32) != 0))
    return -1;
  if ((size_t)n * (size_t)k / 8 >= (1ull << 31)) return -1;
  // three launches by remainder size -- 1..16 rows: one 16-row x block (32 KiB of LDS, five blocks per CU); 17..32: two; 33..tail-1: four (128 KiB, one block
  // per CU) -- each a grid of (expert, slab group) blocks of which only the experts with such a remainder do any work
#define AWQ_MST(DT_, EPI_)                                                                                              \
  {                                                                                                                     \
    launch_moe_skinny<DT_, 8, 2, 1, EPI_>(x, qw, szp, offsets, out, experts, n, k, st, 1, tail < 17 ? tail : 17);       \
    if (tail > 17) launch_moe_skinny<DT_, 8, 2, 2, EPI_>(x, qw, szp, offsets, out, experts, n, k, st, 17, tail < 33 ? tail : 33); \
    if (tail > 33) launch_moe_skinny<DT_, 8, 2, 4, EPI_>(x, qw, szp, offsets, out, experts, n, k, st, 33, tail);        \
  }
  if (dtype == 0) {
    if (epi == 2) AWQ_MST(F16, 2) else AWQ_MST(F16, 0)
  } else {
    if (epi == 2) AWQ_MST(BF16, 2) else AWQ_MST(BF16, 0)
  }
#undef AWQ_MST
  return 0;
}

// 9 <= total_rows <= 255 sorted rows over `experts` experts (stacked cdna4 weights + packed sz).  The column-block count is
// sized for 1.5x the mean rows per expert; an expert with more rows takes extra passes.  Returns -1 if unsupported.
int launch_moe_skinny_cdna4(const void* x, const void* qw, const void* szp, const void* offsets, void* out, int total_rows,
                            int experts, int n, int k, int dtype, hipStream_t st) {
  if (!szp || total_rows < 1 || total_rows > 255 || experts < 1 || (n % 16) != 0 || (k % 128) != 0) return -1;
  if ((size_t)n * (size_t)k / 8 >= (1ull << 31) || (size_t)total_rows * (size_t)k >= (1ull << 31)) return -1;
  const int want = (3 * total_rows + 2 * experts - 1) / (2 * experts);  // 1.5 x mean
  const int cb = want <= 16 ? 1 : (want <= 32 ? 2 : (want <= 48 ? 3 : 4));
  const bool wide = n / 16 >= 512;
#define AWQ_MS(DT_)                                                                                             \
  if (cb == 1) {                                                                                                \
    if (wide) launch_moe_skinny<DT_, 8, 2, 1>(x, qw, szp, offsets, out, experts, n, k, st);                     \
    else launch_moe_skinny<DT_, 8, 1, 1>(x, qw, szp, offsets, out, experts, n, k, st);                          \
  } else if (cb == 2) {                                                                                         \
    if (wide) launch_moe_skinny<DT_, 8, 2, 2>(x, qw, szp, offsets, out, experts, n, k, st);                     \
    else launch_moe_skinny<DT_, 8, 1, 2>(x, qw, szp, offsets, out, experts, n, k, st);                          \
  } else if (cb == 3) {                                                                                         \
    launch_moe_skinny<DT_, 8, 2, 3>(x, qw, szp, offsets, out, experts, n, k, st);                               \
  } else {                                                                                                      \
    launch_moe_skinny<DT_, 8, 2, 4>(x, qw, szp, offsets, out, experts, n, k, st);                               \
  }
  if (dtype == 0) {
    AWQ_MS(F16)
  } else {
    AWQ_MS(BF16)
  }
#undef AWQ_MS
  return 0;
}

}  // namespace awq
