// Decode GEMV on cdna4-interleaved weights, 1 <= M <= 8, bf16 and fp16 (gfx950).  The fast path of
// WQLinear.forward for decode (replaces gemv_kernel, awq/kernels/csrc/quantization_new/gemv/gemv_cuda.cu:74-229)
// and, with EPI = 1, of QuantLlamaMLP's gate/up pair + SiLU*mul (tinychat/modules/fused_mlp.py:36-83).
//
// Structure (DESIGN.md "Decode GEMV"; measurements under profiles/r01_gemv*):
//   * block = one 16-row slab (EPI 1: the gate slab and the matching up slab), WAVES waves split K in interleaved
//     128-k steps.  A wave walks its steps in chunks of S (1 or 2) through a ring of PIPE = 2 register sets: the loads of
//     the next chunk (S x 1 KiB of packed weights, S dwords of packed {scale | scaled_zero}, its own x slices) are in
//     flight while the current chunk is dequantised.  No load sits in a branch (indices past the end are clamped), so
//     hipcc emits counted vmcnt waits.  (PIPE = 0, kept for the knob experiments: all loads of a chunk up front.)
//   * x slices are staged through a WAVE-PRIVATE LDS region (no block barrier before the final reduction).
//   * weights are dequantised on the matrix core (Cdna4DequantT, awq_device.hpp: exact q*s+sz in fp32, one
//     v_cvt_pk = the reference's rounding) and fed as the A operand of v_mfma_f32_16x16x32 against the activation rows;
//     fp32 accumulation; split-K partials reduced through LDS; one rounding; bias / SiLU*mul in the epilogue.
//   * every global load is a raw buffer load with a scalar tile offset: the per-step address arithmetic is on the SALU.
// NORM = 1 fuses the RMSNorm in front of the linear (FTLlamaRMSNorm -> WQLinear, tinychat/modules/fused_norm.py:7-21 +
// awq/kernels/csrc/layernorm/layernorm.cu:39-61): out = W . T((x * rsqrt(mean(x^2) + eps)) * gamma).
// The block's waves already load disjoint k-slices of x, so each wave sums the squares of its own slices, the partials
// meet in LDS (one barrier, while the first weight tiles are in flight), and the slices are normalised into the same
// wave-private staging region the plain kernel uses -- the 8 KiB activation row never makes a round trip through HBM.
#include <string.h>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

constexpr int kNormSteps = 8;  // NORM: k-steps of x a wave may own (K <= 128 * 8 * WAVES)

// the whole block's work for slab `nb`: shared by the plain kernel, the norm-fused kernel and the grouped (per-expert) kernel

template <typename DT, int WAVES, int S, int MB, int EPI, int BITS, int PIPE, int NORM = 0>
__device__ __forceinline__ void gemv_cdna4_body(char* smem, const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                const u32* __restrict__ szp, const uint16_t* __restrict__ bias,
                                                uint16_t* __restrict__ out, int M, int N, int K, int nb,
                                                const uint16_t* __restrict__ gamma = nullptr, float eps = 0.f) {
  constexpr int NS = EPI == 1 ? 2 : 1;  // slabs per block
  // the wave index is wave-uniform: taking it through readfirstlane puts every tile / step address computation on
  // the scalar unit (the kernel is VALU-issue bound: ~5 cycles per VALU instruction per SIMD, DESIGN.md "gemv")
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int nit = K >> 7;
  const int xstep = M * 256;  // bytes of x per 128-k step (M rows)
  float(*red)[4][64] = reinterpret_cast<float(*)[4][64]>(smem);  // [NS * WAVES][4][64]
  // x staging: S steps per wave (re-written chunk by chunk); NORM: every step the wave owns, normalised once up front
  const int xsteps = NORM ? (nit + WAVES - 1) / WAVES : S;
  char* xs = smem + NS * WAVES * 1024 + wv * (xsteps * xstep);

  // Every global load is a raw buffer load: SGPR descriptor + loop-invariant VGPR lane offset + SGPR tile offset, so
  // the per-step address arithmetic runs on the scalar unit and costs no VALU issue slot (out-of-range reads, which
  // cannot happen here, would return 0).  All three tensors are < 4 GiB.
  const int tile_bytes = BITS == 4 ? 1024 : 768;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(qw), 0, (N >> 4) * nit * tile_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(szp), 0, (N >> 4) * nit * 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, M * K * 2, 0x00020000);
  u32 slab_tile[NS];  // first tile index of the block's slab(s)
#pragma unroll
  for (int s = 0; s < NS; ++s) slab_tile[s] = ((u32)nb + (u32)s * (u32)(N >> 5)) * (u32)nit;  // EPI 1: up slab = gate slab + (N/2)/16
  const u32 wlane_b = BITS == 4 ? lane * 16u : lane * 12u;
  const u32 ilane_b = (u32)i * 4u;
  using vec8 = typename DT::vec8;
  Cdna4DequantT<DT> cd;
  cd.init(lane, BITS == 4 ? 0x000F000Fu : 0x00070007u);
  const int mrow = min(i, M - 1);
  const int cnt = (nit - wv + WAVES - 1) / WAVES;  // this wave's steps: kg = wv + WAVES * t

  f32x4 acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Regs {  // everything a wave needs from global memory for one chunk of S steps
    u32x4 xr[S][MB];
    u32x4 w[NS][S];
    u32 sz[NS][S];
  };
  auto load_chunk = [&](int c0, Regs& R) {
    // x slices first (vmcnt retires in order: their wait must not sit behind the weight stream)
#pragma unroll
    for (int t = 0; t < (NORM ? 0 : S); ++t) {
      const int kg = min(wv + WAVES * (c0 + t), nit - 1);
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        // LDS slot i of row r receives granule i ^ (r & 15): the M rows of an MFMA operand are 256 B apart (one full
        // bank sweep), the XOR spreads their equal granule indices over different banks
        const int r = min(4 * b + g, M - 1);
        const u32 xoff_b = ((u32)r * (u32)K + (u32)((i ^ (r & 15)) * 8)) * 2u;  // M * K * 2 < 2^32
        R.xr[t][b] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff_b, (u32)kg * 256u, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < S; ++t) {
      const int kg = min(wv + WAVES * (c0 + t), nit - 1);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const u32 tidx = slab_tile[s] + (u32)kg;
        if (BITS == 4) {
          R.w[s][t] = __builtin_amdgcn_raw_buffer_load_b128(rw, wlane_b, tidx * 1024u, 2);  // aux 2 = nt (streamed once)
        } else {
          typedef u32 u32x3 __attribute__((ext_vector_type(3)));
          const u32x3 w3 = __builtin_amdgcn_raw_buffer_load_b96(rw, wlane_b, tidx * 768u, 2);
          R.w[s][t] = u32x4{w3.x, w3.y, w3.z, 0u};
        }
        R.sz[s][t] = __builtin_amdgcn_raw_buffer_load_b32(rs, ilane_b, tidx * 64u, 0);
      }
    }
  };
  auto compute_chunk = [&](int c0, const Regs& R) {
#pragma unroll
    for (int t = 0; t < (NORM ? 0 : S); ++t)
#pragma unroll
      for (int b = 0; b < MB; ++b)
        // unconditional: lanes past the last row hold a copy of row M - 1 and write it to that row's slot (same bytes).
        // A branch here lets hipcc sink the x loads behind the weight stream, whose in-order vmcnt then stalls the staging
        *reinterpret_cast<u32x4*>(xs + t * xstep + min(4 * b + g, M - 1) * 256 + i * 16) = R.xr[t][b];
#pragma unroll
    for (int t = 0; t < S; ++t) {
      if (wv + WAVES * (c0 + t) >= nit) continue;  // ragged tail: wave-uniform skip (the loads were clamped)
      const u32x4* xrow = reinterpret_cast<const u32x4*>(xs + (NORM ? c0 + t : t) * xstep + mrow * 256);
      vec8 xop[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) xop[a] = __builtin_bit_cast(vec8, xrow[(4 * a + g) ^ (mrow & 15)]);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const u32 szv = R.sz[s][t];
        vec8 op[4];
        const u32x4 wt = BITS == 4 ? R.w[s][t] : w3_expand(R.w[s][t].x, R.w[s][t].y, R.w[s][t].z);
        cd.tile_packed(wt, szv, op);
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[s] = DT::mfma(op[a], xop[a], acc[s]);
      }
    }
  };
  // NORM: every x / gamma slice the wave owns goes to registers FIRST (vmcnt retires in order: these small L2-resident loads
  // must not queue behind the weight stream), then the ring's first weight chunk is requested, then the norm is finished
  static_assert(!NORM || MB == 1, "the fused RMSNorm path serves M <= 4");
  u32x4 xv[NORM ? kNormSteps : 1], gv[NORM ? kNormSteps : 1];
  const int nr = min(g, M - 1);                        // (MB == 1) row of this lane group, clamped
  const u32 ngran_b = (u32)((i ^ (nr & 15)) * 8) * 2u;  // the granule that lands in LDS slot i of row nr
  auto norm_issue = [&]() {
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(gamma), 0, K * 2, 0x00020000);
    const u32 xoff_b = (u32)nr * (u32)K * 2u + ngran_b;
#pragma unroll
    for (int t = 0; t < kNormSteps; ++t) {
      const int kg = min(wv + WAVES * t, nit - 1);  // clamped: no load in a branch; steps past the end are masked below
      xv[t] = __builtin_amdgcn_raw_buffer_load_b128(rx, xoff_b, (u32)kg * 256u, 0);
      gv[t] = __builtin_amdgcn_raw_buffer_load_b128(rg, ngran_b, (u32)kg * 256u, 0);
    }
  };
  auto norm_finish = [&]() {
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < kNormSteps; ++t) {
      if (t < cnt) {
        const u32 w4[4] = {xv[t].x, xv[t].y, xv[t].z, xv[t].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float lo = DT::to_float((uint16_t)(w4[e] & 0xFFFFu)), hi = DT::to_float((uint16_t)(w4[e] >> 16));
          ss = __builtin_fmaf(lo, lo, ss);
          ss = __builtin_fmaf(hi, hi, ss);
        }
      }
    }
    // the 16 lanes of a row group hold different granules of the same row
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) ss += __shfl_xor(ss, d, 64);
    float* psum = reinterpret_cast<float*>(smem + NS * WAVES * 1024 + WAVES * (xsteps * xstep));  // [WAVES][4]
    if (i == 0) psum[wv * 4 + g] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < WAVES; ++q) tot += psum[q * 4 + g];
    const float rstd = rsqrtf(tot / (float)K + eps);  // layernorm.cu:55: rsqrtf(variance / n + eps)
#pragma unroll
    for (int t = 0; t < kNormSteps; ++t) {
      if (t < cnt) {
        const u32 w4[4] = {xv[t].x, xv[t].y, xv[t].z, xv[t].w}, g4[4] = {gv[t].x, gv[t].y, gv[t].z, gv[t].w};
        u32 o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // layernorm.cu:60: T((float(x) * s_variance) * float(gamma))
          const float lo = (DT::to_float((uint16_t)(w4[e] & 0xFFFFu)) * rstd) * DT::to_float((uint16_t)(g4[e] & 0xFFFFu));
          const float hi = (DT::to_float((uint16_t)(w4[e] >> 16)) * rstd) * DT::to_float((uint16_t)(g4[e] >> 16));
          o4[e] = (u32)DT::from_float(lo) | ((u32)DT::from_float(hi) << 16);
        }
        *reinterpret_cast<u32x4*>(xs + t * xstep + nr * 256 + i * 16) = u32x4{o4[0], o4[1], o4[2], o4[3]};
      }
    }
  };
  if (PIPE == 0) {
    // every load of a chunk up front, then its math
    for (int c0 = 0; c0 < cnt; c0 += S) {
      Regs R;
      load_chunk(c0, R);
      compute_chunk(c0, R);
    }
  } else {
    // software pipeline over a ring of PIPE register sets: the loads of the next PIPE - 1 chunks are in flight while a
    // chunk is dequantised.  A wave keeps at most PIPE * S tiles in flight and -- with ~7 waves per SIMD -- the memory
    // system serves the waves breadth first, so all of them compute at the same time instead of one after the other
    // (loads past the end are clamped: L2 hits on the slab's last tile)
    constexpr int D = PIPE > 0 ? PIPE : 1;  // (PIPE == 0 never gets here; keeps the array non-empty)
    Regs R[D];
    int cbeg = 0;
    if constexpr (NORM != 0) {
      // both register sets are requested before the norm is finished (two chunks per wave keep the stream going through the
      // reduce / barrier / normalise chain); the first ring iteration is peeled accordingly
      static_assert(!NORM || D == 2, "");
      norm_issue();
      load_chunk(0, R[0]);
      load_chunk(S, R[1]);
      norm_finish();
      compute_chunk(0, R[0]);
      load_chunk(2 * S, R[0]);
      compute_chunk(S, R[1]);
      cbeg = 2 * S;
    } else {
#pragma unroll
      for (int j = 0; j + 1 < D; ++j) load_chunk(j * S, R[j]);
    }
    for (int c0 = cbeg; c0 < cnt; c0 += D * S) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        load_chunk(c0 + (j + D - 1) * S, R[(j + D - 1) % D]);
        compute_chunk(c0 + j * S, R[j]);
      }
    }
  }

  // ---- split-K reduction across the block's waves (fp32).  acc[r] = C[n = 4g + r][m = i] ----
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[s * WAVES + wv][r][lane] = acc[s][r];
  __syncthreads();
  if (EPI == 2) {
    // QuantLlamaMLP's interleaved pair (fused_mlp.py:79-82): rows 0..7 of the slab are gate rows 8 nb .. 8 nb + 7, rows 8..15 the matching up
    // rows, so lane (g < 2) pairs with lane + 32; out[m, N / 2] = T(T(silu(T(gate))) * T(up))
    if (wv < 4 && i < M && g < 2) {
      const int r = wv;
      float gsum = 0.f, usum = 0.f;
#pragma unroll
      for (int q = 0; q < WAVES; ++q) {
        gsum += red[q][r][lane];
        usum += red[q][r][lane + 32];
      }
      const float gt = DT::to_float(DT::from_float(gsum)), up = DT::to_float(DT::from_float(usum));
      const float sl = DT::to_float(DT::from_float(silu_f32(gt)));
      out[(size_t)i * (N >> 1) + nb * 8 + 4 * g + r] = DT::from_float(sl * up);
    }
    return;
  }
  if (wv < 4 && i < M) {
    const int r = wv;
    float v[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < WAVES; ++q) t += red[s * WAVES + q][r][lane];
      v[s] = t;
    }
    const int nn = nb * 16 + 4 * g + r;
    auto to_f = [](uint16_t b) { return DT::to_float(b); };
    if (EPI == 0) {
      uint16_t o = DT::from_float(v[0]);
      if (bias != nullptr) o = DT::from_float(to_f(o) + to_f(bias[nn]));  // `out + self.bias` in T (qmodule.py:221)
      out[(size_t)i * N + nn] = o;
    } else if (EPI == 3) {
      // row-split shard: the fp32 sum leaves unrounded, the caller reduces it across ranks before the one rounding
      reinterpret_cast<float*>(out)[(size_t)i * N + nn] = v[0];
    } else {
      // fused_mlp.py:79-82: c = F.silu(gate_output) * up_output, every op rounded to T
      const float gt = to_f(DT::from_float(v[0])), up = to_f(DT::from_float(v[NS - 1]));
      const float sl = to_f(DT::from_float(silu_f32(gt)));
      out[(size_t)i * (N >> 1) + nn] = DT::from_float(sl * up);
    }
  }
}

template <typename DT, int WAVES, int S, int EPI>
__global__ __launch_bounds__(64 * WAVES) void gemv_cdna4_norm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ gamma,
                                                                      const u32* __restrict__ qw, const u32* __restrict__ szp,
                                                                      const uint16_t* __restrict__ bias, uint16_t* __restrict__ out,
                                                                      int M, int N, int K, float eps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemv_cdna4_body<DT, WAVES, S, 1, EPI, 4, 2, 1>(smem, x, qw, szp, bias, out, M, N, K, blockIdx.x, gamma, eps);
}

template <typename DT, int WAVES, int S, int MB, int EPI, int BITS, int PIPE>
__global__ __launch_bounds__(64 * WAVES) void gemv_cdna4_kernel(const uint16_t* __restrict__ x,
                                                                 const u32* __restrict__ qw,
                                                                 const u32* __restrict__ szp,
                                                                 const uint16_t* __restrict__ bias,
                                                                 uint16_t* __restrict__ out, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemv_cdna4_body<DT, WAVES, S, MB, EPI, BITS, PIPE>(smem, x, qw, szp, bias, out, M, N, K, blockIdx.x);
}

// Grouped (per-expert) decode GEMV for MoE layers: block = (expert, slab); expert e owns rows [offsets[e], offsets[e+1]) of
// the sorted x / out (at most 4 MB rows: the host only routes here when the TOTAL row count is that small) and the
// e-th slice of the stacked cdna4 weights / packed scales.  Experts without tokens cost one offsets read per block.
template <typename DT, int WAVES, int S, int MB>
__global__ __launch_bounds__(64 * WAVES) void moe_gemv_cdna4_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                                     const u32* __restrict__ szp,
                                                                     const int* __restrict__ offsets,
                                                                     uint16_t* __restrict__ out, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nslab = N >> 4, nit = K >> 7;
  const int e = blockIdx.x / nslab, nb = blockIdx.x - e * nslab;
  const int row0 = offsets[e], m_e = offsets[e + 1] - row0;
  if (m_e <= 0) return;
  gemv_cdna4_body<DT, WAVES, S, MB, 0, 4, 0>(smem, x + (size_t)row0 * K, qw + (size_t)e * nslab * nit * 256,
                                      szp + (size_t)e * nslab * nit * 16, nullptr, out + (size_t)row0 * N, min(m_e, 4 * MB), N, K, nb);
}

namespace {
struct Cfg {
  int waves, s;
};
// choose the K split (waves per slab) and, for the chunk mode, the chunk length
Cfg pick_cfg(int m, int n_rows, int k, int ns, int force_waves, int force_s, int bits = 4) {
  // measured on MI355X (tools/gemvc_sweep.py, profiles/r01_gemvc_sweep.txt): short chunks (2..4 steps, 7 when the
  // step count is a multiple of 7) and ~7 k waves in flight chip-wide; a chunk longer than the wave's step count
  // only adds clamped loads
  const int nit = k / kGroup, slabs = n_rows / 16 / ns;
  int waves = slabs >= 768 ? 4 : (slabs >= 384 ? 8 : (nit >= 64 ? 8 : 16));
  // w3c tiles: eight waves up to ~1000 slabs (Llama-2-7B qkv, 768 slabs: 6.7 vs 7.1 us), four above (the fused gate / up stack, 1376 slabs:
  // 11.2-11.7 vs 11.8-12.2 us): profiles/r05_w3_fused_mlp.txt
  if (bits == 3 && slabs >= 768 && slabs < 1024) waves = 8;
  if (ns == 2 && waves > 4 && slabs >= 256) waves >>= 1;  // two slabs per block: half the waves give the same bytes in flight
  while (waves > 4 && waves * 2 > nit) waves >>= 1;
  if (force_waves) waves = force_waves;
  const int per = (nit + waves - 1) / waves;
  int s = (per >= 7 && per % 7 == 0) ? 7 : (per >= 4 ? 4 : 2);
  if (ns == 2) s = 2;  // registers: two weight streams
  if (force_s) s = force_s;
  while (s > 2 && (size_t)waves * s * m * 256 > 48 * 1024) s = s == 8 ? 7 : (s == 7 ? 4 : 2);  // LDS for the x slices
  return {waves, s};
}
int g_force_waves = 0, g_force_s = 0;
// g_pipe: -1 = default (ring of 2, the measured optimum: profiles/r01_gemvc_ring_sweep.txt), 0 = all-loads-up-front chunks,
// >= 2 = ring depth of the software pipeline; g_pipe_s: steps per chunk (0 = by steps per wave)
int g_pipe = -1, g_pipe_s = 0;
int g_use_dma = 1;  // 1 = the LDS-DMA streaming kernel (awq_gemv_dma.hip) serves W4 decode; 0 = the register-ring kernel below
}  // namespace

int gemv_cdna4_tune_set(const char* key, int value) {
  if (!strcmp(key, "gemvc_waves")) g_force_waves = value;
  else if (!strcmp(key, "gemvc_s")) g_force_s = value;
  else if (!strcmp(key, "gemvc_pipe")) g_pipe = value;
  else if (!strcmp(key, "gemvc_pipe_s")) g_pipe_s = value;
  else if (!strcmp(key, "gemv_dma")) g_use_dma = value;
  else return gemv_dma_tune_set(key, value);
  return 0;
}

template <typename DT, int WAVES, int S, int MB, int EPI, int BITS, int PIPE>
static void launch_cfg(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                       hipStream_t st) {
  constexpr int NS = EPI == 1 ? 2 : 1;
  const size_t smem = (size_t)NS * WAVES * 1024 + (size_t)WAVES * S * m * 256;
  auto kern = gemv_cdna4_kernel<DT, WAVES, S, MB, EPI, BITS, PIPE>;
  static LdsOptIn optin;  // per (kernel instantiation, device)
  if (smem > 64 * 1024) optin.ensure(reinterpret_cast<const void*>(kern));
  hipLaunchKernelGGL(kern, dim3(n / 16 / NS), dim3(64 * WAVES), smem, st, (const uint16_t*)x, (const u32*)qw,
                     (const u32*)szp, (const uint16_t*)bias, (uint16_t*)out, m, n, k);
}

template <typename DT, int MB, int EPI, int BITS>
static int launch_mb(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                     hipStream_t st) {
  const Cfg c = pick_cfg(m, n, k, EPI == 1 ? 2 : 1, g_force_waves, g_force_s, BITS);
  const int pipe = g_pipe < 0 ? 2 : g_pipe;
  if (pipe >= 2) {  // software-pipelined variant: ring of `pipe` chunks of `ps` steps
    const int per = (k / kGroup + c.waves - 1) / c.waves;
    // steps per chunk: about 16-28 KiB of weight tiles in flight per CU (measured optimum, profiles/r01_gemvc_ring_sweep.txt
    // and r01_gemvc_ring2.txt: deeper rings and longer chunks only lengthen the queue in front of the last waves' first
    // data -- a CU does not pull more than ~25 GB/s whatever is outstanding, profiles/r01_gemv_trace.txt)
    const double waves_per_cu = (double)(n / 16 / (EPI == 1 ? 2 : 1)) * c.waves / 256.0;
    const int ps = g_pipe_s ? g_pipe_s : ((per >= 2 && waves_per_cu * (EPI == 1 ? 2 : 1) <= 10.0) ? 2 : 1);
#define AWQ_PCASE(W_, S_, D_)                                                  \
  if (c.waves == W_ && ps == S_ && pipe == D_ && (size_t)W_ * S_ * m * 256 <= 96 * 1024) { \
    launch_cfg<DT, W_, S_, MB, EPI, BITS, D_>(x, qw, szp, bias, out, m, n, k, st); \
    return 0;                                                                  \
  }
    AWQ_PCASE(4, 1, 2) AWQ_PCASE(4, 2, 2) AWQ_PCASE(8, 1, 2) AWQ_PCASE(8, 2, 2) AWQ_PCASE(16, 1, 2) AWQ_PCASE(16, 2, 2)
#ifdef AWQ_ENABLE_PROBES
    if constexpr (BITS == 4 && EPI == 0 && DT::id == 1) {
      AWQ_PCASE(4, 1, 3) AWQ_PCASE(8, 1, 3) AWQ_PCASE(16, 1, 3)
    }
#endif
#undef AWQ_PCASE
  }
#ifndef AWQ_ENABLE_PROBES
  return -1;  // (the all-loads-up-front chunk variants, some of which spill, exist in AWQ_PROBES builds only)
#else
  if constexpr (DT::id != 1) return -1;  // the chunk-mode variants below exist for the knob experiments: bf16 only
#define AWQ_CASE(W_, S_)                                                      \
  if (c.waves == W_ && c.s == S_) {                                           \
    launch_cfg<DT, W_, S_, MB, EPI, BITS, 0>(x, qw, szp, bias, out, m, n, k, st); \
    return 0;                                                                 \
  }
  if constexpr (DT::id == 1) {
    AWQ_CASE(4, 2) AWQ_CASE(4, 4) AWQ_CASE(4, 7) AWQ_CASE(4, 8) AWQ_CASE(8, 2) AWQ_CASE(8, 4) AWQ_CASE(8, 7) AWQ_CASE(8, 8)
    AWQ_CASE(16, 2) AWQ_CASE(16, 4) AWQ_CASE(16, 7) AWQ_CASE(16, 8)
  }
#undef AWQ_CASE
  return -1;
#endif
}

// epi 0: out[m, n] (+ bias);  epi 1: qw holds [gate; up] stacked along N (n = 2 * ffn rows), out[m, n/2] = silu(gate) * up
// bits 4: cdna4 W4 tiles; bits 3: w3c tiles (epi 0; epi 2 = gate / up rows interleaved 8 + 8 per slab, out[m, n/2]; epi 3 = fp32 partial sums out[m, n] with no bias, for row-split shards)
int launch_gemv_cdna4(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                      int epi, int bits, int dtype, hipStream_t st) {
  if (m < 1 || m > 8) return -1;
  if (bits == 4 && g_use_dma && launch_gemv_dma(x, qw, szp, bias, out, m, n, k, epi, dtype, 0, st) == 0) return 0;
  if (epi == 2 && bits != 3) return -1;  // interleaved gate / up rows of a W4 layer: only the streaming kernel pairs them
  if (bits == 3) {
    if (epi == 3) {
      if (dtype == 0) return m <= 4 ? launch_mb<F16, 1, 3, 3>(x, qw, szp, nullptr, out, m, n, k, st) : launch_mb<F16, 2, 3, 3>(x, qw, szp, nullptr, out, m, n, k, st);
      return m <= 4 ? launch_mb<BF16, 1, 3, 3>(x, qw, szp, nullptr, out, m, n, k, st) : launch_mb<BF16, 2, 3, 3>(x, qw, szp, nullptr, out, m, n, k, st);
    }
    if (epi == 2) {  // the interleaved gate / up stack of a 3-bit QuantLlamaMLP
      if (dtype == 0) return m <= 4 ? launch_mb<F16, 1, 2, 3>(x, qw, szp, nullptr, out, m, n, k, st) : launch_mb<F16, 2, 2, 3>(x, qw, szp, nullptr, out, m, n, k, st);
      return m <= 4 ? launch_mb<BF16, 1, 2, 3>(x, qw, szp, nullptr, out, m, n, k, st) : launch_mb<BF16, 2, 2, 3>(x, qw, szp, nullptr, out, m, n, k, st);
    }
    if (epi != 0) return -1;
    if (dtype == 0) return m <= 4 ? launch_mb<F16, 1, 0, 3>(x, qw, szp, bias, out, m, n, k, st) : launch_mb<F16, 2, 0, 3>(x, qw, szp, bias, out, m, n, k, st);
    return m <= 4 ? launch_mb<BF16, 1, 0, 3>(x, qw, szp, bias, out, m, n, k, st) : launch_mb<BF16, 2, 0, 3>(x, qw, szp, bias, out, m, n, k, st);
  }
#define AWQ_DT(DT_)                                                                                                     \
  if (epi == 1)                                                                                                         \
    return m <= 4 ? launch_mb<DT_, 1, 1, 4>(x, qw, szp, bias, out, m, n, k, st) : launch_mb<DT_, 2, 1, 4>(x, qw, szp, bias, out, m, n, k, st); \
  return m <= 4 ? launch_mb<DT_, 1, 0, 4>(x, qw, szp, bias, out, m, n, k, st) : launch_mb<DT_, 2, 0, 4>(x, qw, szp, bias, out, m, n, k, st);
  if (dtype == 0) {
    AWQ_DT(F16)
  }
  AWQ_DT(BF16)
#undef AWQ_DT
}

// RMSNorm fused in front of the linear (see the header): x is the UN-normalised activation, gamma the norm weight [k].
// 1 <= m <= 4, k <= 128 * 8 * waves; epi as above.  Returns -1 if unsupported (the caller runs norm and linear separately).
int launch_gemv_cdna4_norm(const void* x, const void* gamma, float eps, const void* qw, const void* szp, const void* bias, void* out,
                           int m, int n, int k, int epi, int dtype, hipStream_t st) {
  if (m < 1 || m > 4 || !gamma || !szp || (n % (epi == 1 ? 32 : 16)) != 0 || (k % 128) != 0) return -1;
  const int ns = epi == 1 ? 2 : 1;
  const Cfg c = pick_cfg(m, n, k, ns, g_force_waves, 0);
  const int nit = k / kGroup, per = (nit + c.waves - 1) / c.waves;
  if (per > kNormSteps) return -1;
  const double waves_per_cu = (double)(n / 16 / ns) * c.waves / 256.0;
  const int ps = g_pipe_s ? g_pipe_s : ((per >= 2 && waves_per_cu * ns <= 10.0) ? 2 : 1);
  const size_t smem = (size_t)ns * c.waves * 1024 + (size_t)c.waves * per * m * 256 + (size_t)c.waves * 16;
  if (smem > 160 * 1024) return -1;
#define AWQ_NCASE(DT_, W_, S_, E_)                                                                                       \
  if (c.waves == W_ && ps == S_ && epi == E_) {                                                                          \
    auto kern = gemv_cdna4_norm_kernel<DT_, W_, S_, E_>;                                                                 \
    static LdsOptIn optin;                                                                                                \
    if (smem > 64 * 1024) optin.ensure(reinterpret_cast<const void*>(kern));                                             \
    hipLaunchKernelGGL(kern, dim3(n / 16 / ns), dim3(64 * W_), smem, st, (const uint16_t*)x, (const uint16_t*)gamma, (const u32*)qw, \
                       (const u32*)szp, (const uint16_t*)bias, (uint16_t*)out, m, n, k, eps);                            \
    return 0;                                                                                                            \
  }
#define AWQ_NDT(DT_)                                                                                                     \
  AWQ_NCASE(DT_, 4, 1, 0) AWQ_NCASE(DT_, 4, 2, 0) AWQ_NCASE(DT_, 8, 1, 0) AWQ_NCASE(DT_, 8, 2, 0) AWQ_NCASE(DT_, 16, 1, 0)   \
  AWQ_NCASE(DT_, 16, 2, 0) AWQ_NCASE(DT_, 4, 1, 1) AWQ_NCASE(DT_, 4, 2, 1) AWQ_NCASE(DT_, 8, 1, 1) AWQ_NCASE(DT_, 8, 2, 1)
  if (dtype == 0) {
    AWQ_NDT(F16)
  } else {
    AWQ_NDT(BF16)
  }
#undef AWQ_NDT
#undef AWQ_NCASE
  return -1;
}

// grouped decode GEMV: total_rows <= 8 (so every expert has <= 8 rows), cdna4 layout, stacked packed sz [E][N/16][K/128][16]
int launch_moe_gemv_cdna4(const void* x, const void* qw, const void* szp, const void* offsets, void* out, int total_rows,
                          int experts, int n, int k, int dtype, hipStream_t st) {
  if (total_rows < 1 || total_rows > 8 || (n % 16) != 0 || (k % 128) != 0) return -1;
  const int nit = k / kGroup;
  constexpr int WAVES = 8, S = 2;
  if (nit < WAVES) return -1;
  const int grid = experts * (n / 16);
  if (total_rows <= 4) {
    const size_t smem = (size_t)WAVES * 1024 + (size_t)WAVES * S * 4 * 256;
    auto kern = dtype == 0 ? moe_gemv_cdna4_kernel<F16, WAVES, S, 1> : moe_gemv_cdna4_kernel<BF16, WAVES, S, 1>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), smem, st, (const uint16_t*)x,
                       (const u32*)qw, (const u32*)szp, (const int*)offsets, (uint16_t*)out, n, k);
  } else {
    const size_t smem = (size_t)WAVES * 1024 + (size_t)WAVES * S * 8 * 256;
    auto kern = dtype == 0 ? moe_gemv_cdna4_kernel<F16, WAVES, S, 2> : moe_gemv_cdna4_kernel<BF16, WAVES, S, 2>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), smem, st, (const uint16_t*)x,
                       (const u32*)qw, (const u32*)szp, (const int*)offsets, (uint16_t*)out, n, k);
  }
  return 0;
}

}  // namespace awq
