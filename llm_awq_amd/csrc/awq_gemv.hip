// Decode GEMV / skinny GEMM (1 <= M <= 16) for W4A16, gfx950.
//
// Replaces gemv_kernel<NPerBlock,Batch,BlockSize,GroupSize,T> (reference
// awq/kernels/csrc/quantization_new/gemv/gemv_cuda.cu:74-229).  Design ("gemv"):
//   * HBM-bound: every packed weight byte is read exactly once with 16-byte non-temporal loads,
//     kept PF deep in flight per wave in a register ring.  The steady-state loop has NO conditional
//     loads, so hipcc emits counted s_waitcnt vmcnt(N) (a load inside a branch degrades every wait
//     to vmcnt(0) and serialises the stream -- measured 2x).
//   * one wave owns 16 output rows x a K slice (interleaved 128-k steps across the WAVES waves of a
//     block); x (<= 16 rows) and the slab's scales / scaled_zeros are staged once per block in LDS, so
//     the only global loads in the loop are the packed weights.
//   * two weight layouts:
//       LAYOUT 0  reference v2 interleave: 16 B per lane = 32 k of one row; unpack + dequant on the VALU
//                 (round_T(q*s+sz), bit-exact with the reference's __hfma2), fp16 and bf16.
//       LAYOUT 1  "cdna4" interleave (bf16): one contiguous 1-KiB tile per wave-load, dequantised ON THE
//                 MATRIX CORE by two v_mfma_f32_4x4x4_16B_bf16 per word (awq_device.hpp) -- ~2.5x fewer
//                 VALU instructions per weight.
//   * the dequantised weights are the A operand of v_mfma_f32_16x16x32, the activation rows the B
//     operand: one MFMA per 512 weights whatever M is.  Split-K partials are reduced through LDS in fp32;
//     one rounding to T at the end.
#include <string.h>

#include "awq_device.hpp"
#include "awq_kernels.hpp"

namespace awq {

// One ring slot = everything a wave needs from global memory for one 128-k step.
struct GemvSlot {
  u32x4 w;   // 16 B of packed weights per lane
  u32 s, z;  // scale / scaled_zero bits of the lane's row for this group (SZP: both packed in s)
};

// SZP: scales come from the packed {s | z << 16} array (one dword load per step) instead of two ushort loads
template <typename DT, int PF, int WAVES, int LAYOUT, bool SZP, int PROBE>
__global__ __launch_bounds__(64 * WAVES) void gemv_w4a16_kernel(const uint16_t* __restrict__ x,
                                                                 const u32* __restrict__ qw,
                                                                 const uint16_t* __restrict__ scales,
                                                                 const uint16_t* __restrict__ zeros,
                                                                 const u32* __restrict__ szp,
                                                                 uint16_t* __restrict__ out, int M, int N, int K,
                                                                 int seg_steps) {
  using vec8 = typename DT::vec8;
  constexpr int NT = 64 * WAVES;
  constexpr int XS = 8;  // x granules (16 B) a thread can stage through registers ahead of the weight stream
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  const int i = lane & 15;  // weight row inside the 16-row slab  /  activation row
  const int g = lane >> 4;  // which 32-k chunk of the 128-k step (v2) / which 8-k octet (cdna4)
  const int nb = blockIdx.x;
  const int n0 = nb * 16;
  const int n = min(n0 + i, N - 1);
  const int nit = K / kGroup;  // 128-k steps == quantisation groups

  const u32* wp;  // per-lane pointer of step 0; + it * WSTEP words
  constexpr int WSTEP = LAYOUT == 0 ? 64 : 256;
  if (LAYOUT == 0)
    wp = qw + v2_chunk_word(n, g, K);
  else
    wp = qw + cdna4_tile_word(nb, 0, nit) + lane * 4;
  const u32* szl = szp + (size_t)nb * nit * 16 + i;  // packed {s,z}: + it * 16 (SZP only)
  const int mrow = min(i, M - 1);

  // ---- LDS carve: [reduce WAVES KiB][x segment rows (padded)] ----
  float(*red)[4][64] = reinterpret_cast<float(*)[4][64]>(smem);
  char* xs = smem + WAVES * 1024;
  const int xrow_bytes = 2 * seg_steps * kGroup + 16;

  Cdna4DequantT<DT> cd;
  if (LAYOUT == 1) cd.init(lane);

  auto load_slot = [&](int it) {
    GemvSlot r;
    r.w = ldg_nt_u32x4(wp + (size_t)it * WSTEP);
    if (SZP) {
      r.s = szl[(size_t)it * 16];
      r.z = 0;
    } else {
      r.s = scales[(size_t)it * N + n];
      r.z = zeros[(size_t)it * N + n];
    }
    return r;
  };

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  u32 sink = 0;

  auto compute = [&](const GemvSlot& sl, int it, int seg0) {
    if (PROBE) {
      sink ^= sl.w.x ^ sl.w.y ^ sl.w.z ^ sl.w.w ^ sl.s ^ sl.z;
      return;
    }
    const uint16_t sb = (uint16_t)(sl.s & 0xFFFFu), zb = SZP ? (uint16_t)(sl.s >> 16) : (uint16_t)sl.z;
    if (LAYOUT == 0) {
      const u32x4* xv = reinterpret_cast<const u32x4*>(xs + mrow * xrow_bytes + ((it - seg0) * 128 + g * 32) * 2);
      vec8 wop[4];
      dequant_chunk<DT>(sl.w, DT::make_sz(sb, zb), wop);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = DT::mfma(wop[j], __builtin_bit_cast(vec8, xv[j]), acc);
    } else {
      // operand a covers k = 32a + 8g + 0..7 of the step: x granule index 4a + g
      const u32x4* xr = reinterpret_cast<const u32x4*>(xs + mrow * xrow_bytes + (it - seg0) * 256);
      vec8 op[4];
      cd.tile(sl.w, sb, zb, op);
#pragma unroll
      for (int a = 0; a < 4; ++a)
        acc = DT::mfma(op[a], __builtin_bit_cast(vec8, xr[4 * a + g]), acc);
    }
  };

  for (int seg0 = 0; seg0 < nit; seg0 += seg_steps) {
    const int seg1 = min(nit, seg0 + seg_steps);
    // this wave's steps inside the segment: it(t) = seg0 + wv + WAVES * t, t in [0, cnt)
    const int cnt = max(0, (seg1 - seg0 - wv + WAVES - 1) / WAVES);
    const int groups = cnt / PF, rem = cnt - groups * PF;
    auto step = [&](int t) { return seg0 + wv + WAVES * t; };

    // ---- (1) x granules of this segment -> registers, issued BEFORE the weight stream: vmcnt retires in
    //          order, so the wait in (3) must not sit behind the PF weight loads of (2) ----
    const int per_row = (seg1 - seg0) * 16;  // 16-byte granules per x row in this segment
    const int gran = PROBE ? 0 : M * per_row;
    u32x4 xreg[XS];
#pragma unroll
    for (int e = 0; e < XS; ++e) {
      const int q = min((int)threadIdx.x + e * NT, max(gran - 1, 0));
      const int r = q / per_row, c = q - r * per_row;
      xreg[e] = *reinterpret_cast<const u32x4*>(x + (size_t)r * K + (size_t)seg0 * kGroup + c * 8);
    }
    // ---- (2) start the weight stream (addresses clamped: no branches around loads) ----
    GemvSlot ring[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) ring[u] = load_slot(min(step(min(u, max(cnt - 1, 0))), nit - 1));

    // ---- (3) x -> LDS ----
    if (seg0 > 0) __syncthreads();  // previous segment's x fully consumed
#pragma unroll
    for (int e = 0; e < XS; ++e) {
      const int q = (int)threadIdx.x + e * NT;
      if (q < gran) {
        const int r = q / per_row, c = q - r * per_row;
        *reinterpret_cast<u32x4*>(xs + r * xrow_bytes + c * 16) = xreg[e];
      }
    }
    for (int q = threadIdx.x + XS * NT; q < gran; q += NT) {  // large M*K only (drains the ring once)
      const int r = q / per_row, c = q - r * per_row;
      *reinterpret_cast<u32x4*>(xs + r * xrow_bytes + c * 16) =
          *reinterpret_cast<const u32x4*>(x + (size_t)r * K + (size_t)seg0 * kGroup + c * 8);
    }
    __syncthreads();

    // ---- (4) steady state: consume slot u, THEN refill the same registers PF steps ahead.  (Issuing the
    //          refill before the last use makes hipcc allocate a second register set + copies at the loop
    //          latch, whose waits drain the ring; a load inside a branch degrades every wait to vmcnt(0).) ----
    if (groups > 0) {
      for (int grp = 0; grp + 1 < groups; ++grp) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int t = grp * PF + u;
          compute(ring[u], step(t), seg0);
          ring[u] = load_slot(step(t + PF));
        }
      }
      // last full group: refill only the slots the remainder will use
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int t = (groups - 1) * PF + u;
        compute(ring[u], step(t), seg0);
        if (u < rem) ring[u] = load_slot(step(t + PF));
      }
    }
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (u < rem) compute(ring[u], step(groups * PF + u), seg0);
  }
  if (PROBE) acc[0] = __builtin_bit_cast(float, sink & 0x3fffffffu);

  // cross-wave (split-K) reduction in fp32.  acc[r] = C[n = 4g + r][m = i]
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wv][r][lane] = acc[r];
  __syncthreads();
  if (wv < 4) {
    const int r = wv;  // thread (lane, wv) finalises register r of lane
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s += red[w][r][lane];
    const int m = i;
    const int nn = n0 + 4 * g + r;
    if (m < M && nn < N) out[(size_t)m * N + nn] = DT::from_float(s);
  }
}

// ---- calibration probes (experiments only; compiled with -DAWQ_ENABLE_PROBES = AWQ_PROBES=1 python -m llm_awq_amd.build):
//      ideal linear 16-byte streaming read, and an empty kernel.  A default build has no knob that changes results ----
#ifdef AWQ_ENABLE_PROBES
constexpr bool kProbes = true;
#else
constexpr bool kProbes = false;
#endif
__global__ __launch_bounds__(256) void probe_linear_read_kernel(const u32x4* __restrict__ p, size_t n16, u32* out) {
  u32 acc = 0;
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; idx + 3 * stride < n16; idx += 4 * stride) {
    u32x4 a = __builtin_nontemporal_load(p + idx), b = __builtin_nontemporal_load(p + idx + stride);
    u32x4 c = __builtin_nontemporal_load(p + idx + 2 * stride), d = __builtin_nontemporal_load(p + idx + 3 * stride);
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
  }
  for (; idx < n16; idx += stride) {
    u32x4 a = __builtin_nontemporal_load(p + idx);
    acc ^= a.x ^ a.y ^ a.z ^ a.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void probe_null_kernel(u32* out) {
  if (out == nullptr) out[1] = 0;
}

namespace {
struct GemvTune {
  int waves = 0;  // 0 = auto
  int pf = 0;
  int probe = 0;  // 1 = stream-only in the real access pattern, 2 = linear read, 3 = null kernel
  int probe_blocks = 2048;
  int x_budget_kib = 64;
  int v2fast = 1;
} g_tune;
}  // namespace

int gemv_tune_set(const char* key, int value) {
  if (!strcmp(key, "gemv_waves")) g_tune.waves = value;
  else if (!strcmp(key, "gemv_pf")) g_tune.pf = value;
  else if (kProbes && !strcmp(key, "gemv_probe")) g_tune.probe = value;
  else if (kProbes && !strcmp(key, "gemv_probe_blocks")) g_tune.probe_blocks = value;
  else if (!strcmp(key, "gemv_x_budget_kib")) g_tune.x_budget_kib = value;
  else if (!strcmp(key, "gemv_v2fast")) g_tune.v2fast = value;
  else return -1;
  return 0;
}

bool gemv_v2fast_enabled() { return g_tune.v2fast && g_tune.probe == 0 && g_tune.waves == 0 && g_tune.pf == 0; }

template <typename DT, int PF, int WAVES, int LAYOUT, bool SZP, int PROBE>
static void launch_one(const void* x, const void* qw, const void* s, const void* z, const void* szp, void* out, int m,
                       int n, int k, hipStream_t st) {
  const int nit = k / kGroup;
  // x segment: as many 128-k steps (a multiple of WAVES) as fit the LDS budget
  const long budget = (long)g_tune.x_budget_kib * 1024;
  long seg = ((budget / m) - 16) / (2 * kGroup);
  seg = seg / WAVES * WAVES;
  if (seg < WAVES) seg = WAVES;
  if (seg > nit) seg = (nit + WAVES - 1) / WAVES * WAVES;
  const size_t smem = (size_t)WAVES * 1024 + (size_t)m * (2 * seg * kGroup + 16);
  dim3 grid((n + 15) / 16), block(64 * WAVES);
  auto kern = gemv_w4a16_kernel<DT, PF, WAVES, LAYOUT, SZP, PROBE>;
  static LdsOptIn optin;  // per (kernel instantiation, device)
  if (smem > 64 * 1024) optin.ensure(reinterpret_cast<const void*>(kern));
  hipLaunchKernelGGL(kern, grid, block, smem, st, (const uint16_t*)x, (const u32*)qw, (const uint16_t*)s,
                     (const uint16_t*)z, (const u32*)szp, (uint16_t*)out, m, n, k, (int)seg);
}

template <typename DT, int LAYOUT>
static int launch_gemv_t(const void* x, const void* qw, const void* s, const void* z, const void* szp, void* out, int m,
                         int n, int k, hipStream_t st) {
  const int nit = k / kGroup;
  const int slabs = (n + 15) / 16;
  if (kProbes && g_tune.probe == 2) {
    hipLaunchKernelGGL(probe_linear_read_kernel, dim3(g_tune.probe_blocks), dim3(256), 0, st, (const u32x4*)qw,
                       (size_t)n * k / 32, (u32*)out);
    return 0;
  }
  if (kProbes && g_tune.probe == 3) {
    hipLaunchKernelGGL(probe_null_kernel, dim3(256), dim3(256), 0, st, (u32*)out);
    return 0;
  }
  int waves = g_tune.waves;
  if (waves == 0) waves = slabs >= 768 ? 4 : (slabs >= 320 ? 8 : 16);  // ~16 resident waves per CU
  while (waves > 4 && waves > nit) waves >>= 1;
  int pf = g_tune.pf;
  if (pf == 0) {
    const int per = nit / waves;
    pf = per >= 8 ? 8 : (per >= 4 ? 4 : 2);
  }
  const bool probe = kProbes && g_tune.probe == 1;
#define AWQ_GEMV_CASE(W_, P_)                                                                 \
  if (waves == W_ && pf == P_) {                                                              \
    if (probe) launch_one<DT, P_, W_, LAYOUT, false, kProbes ? 1 : 0>(x, qw, s, z, szp, out, m, n, k, st);  \
    else if (LAYOUT == 1 && szp) launch_one<DT, P_, W_, LAYOUT, LAYOUT == 1, 0>(x, qw, s, z, szp, out, m, n, k, st); \
    else launch_one<DT, P_, W_, LAYOUT, false, 0>(x, qw, s, z, szp, out, m, n, k, st);        \
    return 0;                                                                                 \
  }
  AWQ_GEMV_CASE(4, 4) AWQ_GEMV_CASE(4, 8) AWQ_GEMV_CASE(8, 4) AWQ_GEMV_CASE(8, 8) AWQ_GEMV_CASE(16, 4)
  AWQ_GEMV_CASE(16, 8) AWQ_GEMV_CASE(4, 2) AWQ_GEMV_CASE(8, 2) AWQ_GEMV_CASE(16, 2)
#undef AWQ_GEMV_CASE
  launch_one<DT, 4, 4, LAYOUT, false, 0>(x, qw, s, z, szp, out, m, n, k, st);
  return 0;
}

int launch_gemv(const void* x, const void* qw, const void* s, const void* z, const void* szp, void* out, int m, int n,
                int k, int dtype, int layout, hipStream_t st) {
  if (layout == 1)
    return dtype == 0 ? launch_gemv_t<F16, 1>(x, qw, s, z, szp, out, m, n, k, st) : launch_gemv_t<BF16, 1>(x, qw, s, z, szp, out, m, n, k, st);
  // reference layout, m <= 8: the pipelined kernel of awq_gemv_v2fast.hip unless a knob of this file's kernel is set
  if (gemv_v2fast_enabled() && launch_gemv_v2fast(x, qw, s, z, nullptr, out, m, n, k, k / kGroup, dtype, st) == 0)
    return 0;
  return dtype == 0 ? launch_gemv_t<F16, 0>(x, qw, s, z, nullptr, out, m, n, k, st)
                    : launch_gemv_t<BF16, 0>(x, qw, s, z, nullptr, out, m, n, k, st);
}

}  // namespace awq
