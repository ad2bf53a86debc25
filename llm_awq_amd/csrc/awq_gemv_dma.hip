// Decode GEMV on cdna4-interleaved weights, LDS-DMA streaming form, 1 <= M <= 8, bf16 and fp16 (gfx950).
// Replaces gemv_kernel (awq/kernels/csrc/quantization_new/gemv/gemv_cuda.cu:74-229) behind WQLinear.forward for decode and, with
// EPI = 1 / 2, QuantLlamaMLP's gate/up pair + SiLU*mul (tinychat/modules/fused_mlp.py:36-83).
//
// Why this form (DESIGN.md "Decode GEMV"; profiles/r02_*): the register-ring kernel (awq_gemv_cdna4.hip) keeps ONE chunk per wave in
// flight -- ~28 KiB per CU -- and its slowest waves receive their first bytes last and then pay one full queue round trip per
// chunk (profiles/r01_gemv_trace.txt: the median wave of the gate/up launch ends at 9.8 us, the last at 15.7 us).  Here every wave
// requests its weight tiles up front with `buffer_load_dwordx4 ... lds` (LDS-DMA: no VGPRs, one instruction per KiB), up to D tiles
// deep into a wave-private LDS ring, so a CU has ~100 KiB in flight from the first microsecond and the memory system is never
// waiting for a dependent request; the math trails the stream and reads the tiles back from LDS (one ds_read_b128 per tile).
//   * block = one 16-row slab (EPI 1: the gate slab and its up slab, NS = 2), WAVES waves split K into CONTIGUOUS ranges of TX
//     128-k steps (the slab's K extent is one contiguous stream in the cdna4 layout);
//   * up front per wave: its x slices (M rows x TX x 256 B) and its packed {scale | zero} dwords (NS x TX x 64 B), also by LDS-DMA
//     (older in the vmcnt order than every weight tile, so the first tile wait covers them);
//   * ring of D steps (NS KiB each): step t waits with a COUNTED vmcnt ((D - 1) NS tiles may stay in flight), reads its tile(s), and
//     re-issues the slot for step t + D; the last D steps count down.  All LDS reads of the loop are inline asm: hipcc would
//     otherwise drain the DMA queue (vmcnt(0)) in front of every LDS read that may alias an in-flight DMA;
//   * dequant on the matrix core (Cdna4DequantT: exact q*s+sz, one v_cvt_pk = the reference's rounding), product on
//     v_mfma_f32_16x16x32, fp32 accumulate, split-K partials reduced through LDS, bias / SiLU*mul fused.
#include <string.h>

#include <type_traits>
#include <utility>

#include "../../include/awq_cdna4.h"
#include "awq_device.hpp"
#include "awq_dma.hpp"
#include "awq_kernels.hpp"

namespace awq {

// timing probes (builds with AWQ_PROBES=1 only; wrong results): bit 0 = no math (stream only), bit 1 = no weight DMA and no
// waits for it (math only, on whatever the ring holds), bit 2 = no x staging DMA, bit 3 = no scale (sz) staging DMA (tools/gemvd_xprobe.py)
#ifdef AWQ_ENABLE_PROBES
#define DMA_PROBE(p) ((p) & 0xFF)
#else
#define DMA_PROBE(p) 0
#endif
// bit 8 of the same kernel argument (every build): EPI 0 stores its fp32 sums to `out` as float [M, N] instead of rounding them to T
constexpr int kDmaF32Out = 0x100;
// bit 9: the waves split K in INTERLEAVED 128-k steps (wave w: steps w, w + WAVES, ...) instead of contiguous ranges: the tiles a block has in flight at any
// moment are one contiguous run of its slab's stream (launches whose ring is shallower than a wave's K range: down_proj, qkv, o_proj)
constexpr int kDmaInterleave = 0x200;
// EPI 0: out[m, n] (+ bias);  EPI 1: qw = [gate; up] stacked along N, out[m, n/2] = silu(gate) * up (two slabs per block);
// EPI 2: gate / up rows interleaved 8 + 8 inside every 16-row slab (fused_mlp.QuantLlamaMLP stacks them that way), out[m, n/2]
template <typename DT, int WAVES, int D, int DQ, int EPI>
__device__ __forceinline__ void gemv_dma_body(char* smem, const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                              const u32* __restrict__ szp, const uint16_t* __restrict__ bias,
                                              uint16_t* __restrict__ out, int M, int N, int K, int TX, int probe_, int nb) {
  constexpr int NS = EPI == 1 ? 2 : 1;
  const int probe = DMA_PROBE(probe_);
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int nit = K >> 7;
  const int TXp = (TX + 3) & ~3;                 // steps covered by the 4-step DMA pieces of x / sz
  const int xrow = TXp * 256 + 16;               // bytes per staged x row (+16: the M rows of an operand land in different banks)
  const int wave_bytes = D * NS * 1024 + NS * TXp * 64 + M * xrow;
  char* wbase = smem + wv * wave_bytes;
  char* ring = wbase;                            // [D][NS] tiles of 1 KiB
  char* szs = wbase + D * NS * 1024;             // [NS][TXp] x 64 B
  char* xs = szs + NS * TXp * 64;                // [M][xrow]
  const bool il = (probe_ & kDmaInterleave) != 0;
  const int s0 = il ? wv : wv * TX;              // this wave's first k-step
  const int sk = il ? WAVES : 1;                 // ... and the distance to its next one

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(qw), 0, (N >> 4) * nit * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(szp), 0, (N >> 4) * nit * 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, M * K * 2, 0x00020000);
  u32 slab_tile[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) slab_tile[s] = ((u32)nb + (u32)s * (u32)(N >> 5)) * (u32)nit;

  const u32 lane16 = lane * 16u;
  auto issue = [&](int t, int slot) {  // weight tile(s) of local step t into ring slot `slot`
    const u32 kg = (u32)min(s0 + sk * t, nit - 1);  // steps past the end (ragged K split) re-read the last tile; their math is skipped
    if (probe & 2) return;
#pragma unroll
    for (int s = 0; s < NS; ++s)
      dma_to_lds<16, 2>(rw, ring + (slot * NS + s) * 1024, lane16, (slab_tile[s] + kg) * 1024u);
  };
  // ---- up front.  The first weight tile goes out FIRST (it has the longest way to come), then the packed scales and the x
  // slices (out-of-range pieces read 0 through the buffer descriptor), then the rest of the ring: step 0's counted wait
  // (D - 1 tiles may stay in flight) covers everything older than tile 1 ----
  issue(0, 0);
  // a staging piece covers four of the wave's steps: lane / 16 picks the step (sk steps apart in the source), lane % 16 its 4 / 16 bytes
  const u32 sz_voff = (u32)(lane >> 4) * (u32)(sk * 64) + (u32)(lane & 15) * 4u, x_voff = (u32)(lane >> 4) * (u32)(sk * 256) + (u32)(lane & 15) * 16u;
  if (!(probe & 8)) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
      for (int q = 0; q < TXp; q += 4)
        dma_to_lds<4, 0>(rs, szs + (s * TXp + q) * 64, sz_voff, (slab_tile[s] + (u32)(s0 + sk * q)) * 64u);
  }
  if (!(probe & 4)) {
    for (int r = 0; r < M; ++r)
      for (int q = 0; q < TXp; q += 4)
        dma_to_lds<16, 0>(rx, xs + r * xrow + q * 256, x_voff, ((u32)r * (u32)K + (u32)(s0 + sk * q) * 128u) * 2u);
  }
#pragma unroll
  for (int d = 1; d < D; ++d) issue(d, d);
  using vec8 = typename DT::vec8;
  Cdna4DequantT<DT> cd;   // DQ 0: sz_packed in T
  Cdna4DequantH<DT> ch;   // DQ 1: sz_half, f16-mantissa extraction
  if (DQ == 0) cd.init(lane);
  else ch.init(lane);
  const int mrow = min(i, M - 1);
  const u32 lds0 = (u32)(size_t)(__attribute__((address_space(3))) char*)wbase;
  const u32 ring_lane = lds0 + lane16;
  const u32 sz_lane = lds0 + D * NS * 1024 + i * 4;
  const u32 x_lane = lds0 + D * NS * 1024 + NS * TXp * 64 + mrow * xrow + g * 16;

  f32x4 acc[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};

  int slot = 0;
  auto step = [&](int t, auto vm_, auto reissue_) {
    constexpr int VM = decltype(vm_)::value;
    constexpr bool REISSUE = decltype(reissue_)::value;
    u32x4 w[NS], xo[4];
    u32 sz[NS];
    const u32 ra = ring_lane + slot * (NS * 1024), sa = sz_lane + t * 64, xa = x_lane + t * 256;
    if (!(probe & 2)) dma_wait_vm<VM>();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (s == 0) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(w[s]) : "v"(ra) : "memory");
      else asm volatile("ds_read_b128 %0, %1 offset:1024" : "=v"(w[s]) : "v"(ra) : "memory");
      asm volatile("ds_read_b32 %0, %1" : "=v"(sz[s]) : "v"(sa + s * TXp * 64) : "memory");
    }
    asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(xo[0]) : "v"(xa) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:64" : "=v"(xo[1]) : "v"(xa) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:128" : "=v"(xo[2]) : "v"(xa) : "memory");
    asm volatile("ds_read_b128 %0, %1 offset:192" : "=v"(xo[3]) : "v"(xa) : "memory");
    if (NS == 1)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w[0]), "+v"(sz[0]), "+v"(xo[0]), "+v"(xo[1]), "+v"(xo[2]), "+v"(xo[3]) : : "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(w[0]), "+v"(w[NS - 1]), "+v"(sz[0]), "+v"(sz[NS - 1]), "+v"(xo[0]), "+v"(xo[1]), "+v"(xo[2]), "+v"(xo[3])
                   :
                   : "memory");
    if (REISSUE) issue(t + D, slot);  // the slot's bytes are in registers: refill it for step t + D
    if (s0 + sk * t < nit && !(probe & 1)) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        vec8 op[4];
        if (DQ == 0) cd.tile_packed(w[s], sz[s], op);
        else ch.tile(w[s], sz[s], op);
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[s] = DT::mfma(op[a], __builtin_bit_cast(vec8, xo[a]), acc[s]);
      }
    }
    slot = slot + 1 == D ? 0 : slot + 1;
  };
  int t = 0;
  for (; t < TX - D; ++t) step(t, std::integral_constant<int, (D - 1) * NS>{}, std::true_type{});
  // the last D steps: nothing left to issue, the waits count down
  static_for<0, D>([&](auto j_) {
    constexpr int J = decltype(j_)::value;
    step(t + J, std::integral_constant<int, (D - 1 - J) * NS>{}, std::false_type{});
  });

  // ---- split-K reduction across the block's waves (fp32) through each wave's own (now idle) ring slots ----
  // acc[r] = C[n = 4g + r][m = i];  wave q's partial of slab s lives at smem + q * wave_bytes + s * 1024
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int r = 0; r < 4; ++r) reinterpret_cast<float*>(wbase + s * 1024)[r * 64 + lane] = acc[s][r];
  __syncthreads();
  auto to_f = [](uint16_t b) { return DT::to_float(b); };
  if (EPI != 2) {
    if (wv < 4 && i < M) {
      const int r = wv;
      float v[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        float tsum = 0.f;
#pragma unroll
        for (int q = 0; q < WAVES; ++q) tsum += reinterpret_cast<const float*>(smem + q * wave_bytes + s * 1024)[r * 64 + lane];
        v[s] = tsum;
      }
      const int nn = nb * 16 + 4 * g + r;
      if (EPI == 0 && (probe_ & kDmaF32Out)) {
        // K shard of a tensor-parallel row split (awq_w4a16_partial_cdna4): the fp32 sum goes out unrounded, no bias
        reinterpret_cast<float*>(out)[(size_t)i * N + nn] = v[0];
      } else if (EPI == 0) {
        uint16_t o = DT::from_float(v[0]);
        if (bias != nullptr) o = DT::from_float(to_f(o) + to_f(bias[nn]));  // `out + self.bias` in T (qmodule.py:221)
        out[(size_t)i * N + nn] = o;
      } else {
        // fused_mlp.py:79-82: c = F.silu(gate_output) * up_output, every op rounded to T
        const float gt = to_f(DT::from_float(v[0])), up = to_f(DT::from_float(v[NS - 1]));
        const float sl = to_f(DT::from_float(silu_f32(gt)));
        out[(size_t)i * (N >> 1) + nn] = DT::from_float(sl * up);
      }
    }
  } else {
    // rows 0..7 of the slab are gate rows 8 nb .. 8 nb + 7, rows 8..15 the matching up rows: lane (g < 2) pairs with lane + 32
    auto h_of = [&](int r) {  // T(T(silu(gate)) * up) of accumulator register r: the block's fp32 partials summed over its waves
      float gsum = 0.f, usum = 0.f;
#pragma unroll
      for (int q = 0; q < WAVES; ++q) {
        const float* p = reinterpret_cast<const float*>(smem + q * wave_bytes);
        gsum += p[r * 64 + lane];
        usum += p[r * 64 + lane + 32];
      }
      const float gt = to_f(DT::from_float(gsum)), up = to_f(DT::from_float(usum));
      const float sl = to_f(DT::from_float(silu_f32(gt)));
      return DT::from_float(sl * up);
    };
    if (wv < 4 && i < M && g < 2) {
      out[(size_t)i * (N >> 1) + nb * 8 + 4 * g + wv] = h_of(wv);
    }
  }
}

template <typename DT, int WAVES, int D, int DQ, int EPI>
__global__ __launch_bounds__(64 * WAVES) void gemv_dma_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw,
                                                               const u32* __restrict__ szp, const uint16_t* __restrict__ bias,
                                                               uint16_t* __restrict__ out, int M, int N, int K, int TX, int probe_) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemv_dma_body<DT, WAVES, D, DQ, EPI>(smem, x, qw, szp, bias, out, M, N, K, TX, probe_, blockIdx.x);
}

namespace {
struct DmaCfg {
  int waves, tx, d;
  size_t smem;
};
int g_dma_skinny_from = 0;  // knob decode_skinny_from: 0 = by shape (skinny_takes below), 1..8 = from that row count, 9 = never
int g_dma_waves = 0, g_dma_d = 0, g_dma_probe = 0, g_dma_four = 1, g_dma_want = 0, g_dma_wide8 = 2, g_dma_skinny_small = 1, g_dma_skinny_m1 = 1, g_dma_il = 2;  // decode_skinny_m1 (one row): bit 0 = launches with >= 1.5 slabs per CU (qkv: -4 % per launch, 0 ... -1.2 % per token over three boxes), bit 1 = the long-K 16-wave launches (down_proj: neutral, off)  // gemvd_il: 0 = contiguous K ranges per wave everywhere, 1 = interleaved everywhere, 2 (default) = interleaved for eight-wave blocks whose ring is shallower than a wave's K range (qkv -1.2 %, o_proj -3.2 % at one row; 16-wave down_proj no different: profiles/r06_decode_cfg.txt (7))  // gemvd_four: four ring-7 blocks per CU where there are > 3 slabs per CU (rounds 4 + 3 instead of 3 + 3 + 1: +0.6 % decode tok/s, profiles/r03_gemvps.txt)

size_t dma_smem(int waves, int d, int ns, int tx, int m) {
  const int txp = (tx + 3) & ~3;
  return (size_t)waves * ((size_t)d * ns * 1024 + (size_t)ns * txp * 64 + (size_t)m * (txp * 256 + 16));
}

// Batched decode: this kernel stages m x K x 2 bytes of x per SLAB by LDS-DMA -- at m = 4 as many bytes as the slab's weights -- and the
// staging region crowds the ring out of LDS (gate/up at m = 5: ring depth 1) or forces row chunks that re-stream the weights (K = 14336 at
// m >= 6).  The skinny kernel (awq_skinny_cdna4.hip: x through registers, shared by a block's slabs, one weight pass for up to 16 rows)
// costs the same for every m <= 8 and takes over where that staging exceeds ~128 KiB per CU (profiles/r03_decode_m_sweep.txt: gate/up
// and down_proj from 5 rows, qkv at 8, o_proj never; a Llama-3-8B layer at m = 7: 56.0 -> 38.6 us).
bool skinny_takes(int m, int n_rows, int k, int epi) {
  if (epi == 1) return false;  // (the stacked [gate; up] form has no skinny epilogue)
  if (g_dma_skinny_from) return m >= g_dma_skinny_from;
  const double blocks_per_cu = (double)(n_rows / 16) / 256.0;
  // more than three slabs per CU (the fused gate/up pair; eight-wave blocks from two rows, pick_dma): three co-resident blocks stage 3 m K 2 bytes --
  // K = 4096: the skinny kernel from five rows (15.97 vs 16.99 us), K = 8192 (70B): from three (47.5 vs 51.0 us; at two rows 46.4 vs 47.1 the other way)
  if (blocks_per_cu > 3.0 && k >= 32 * 128 && k < 96 * 128 && g_dma_wide8 != 0) return (size_t)m * (size_t)k * 2 * 3 >= (size_t)(k >= 64 * 128 ? 96 : 112) * 1024;
  // Since the skinny kernel stages only the x pieces that hold rows (round 6, third session: one 4-row piece per k-step up to four rows instead of four) it is AHEAD of this
  // kernel from ONE row wherever a CU holds at least 1.5 slabs or the K loop is long enough for its 16-wave shape -- qkv 5.0 vs 5.35 us, down_proj 8.2 vs 8.7 at one row, 7.85 vs
  // 9.2 at four; Llama-3-70B qkv / o / down -8 ... -9 % at one row -- while ONE slab per CU against a short K keeps this kernel's deeper ring (o_proj 4.2 vs 4.5, Llama-2-7B
  // down_proj 7.8 vs 9.1): profiles/r06_decode_cfg.txt (9).  Knob decode_skinny_small: 0 = the rule of the first two sessions below, 1 (default) = from two rows, 2 = from one row.
  // (ONE row stays here: in the token's chain of dependent launches the skinny kernel's isolated -6 % does not show -- bench.py A/B: 0.9766-0.9801 vs 0.9704-0.9772 ms per step,
  // drop-in leg -2 % -- while four rows gain 5 %: 1.012-1.019 vs 1.066-1.070 ms)
  // ... except where a CU holds two or more slabs (Llama-3-70B qkv / o / down: its four-launch decode layer 93.9 -> 91.2-92.7 us at one row)
  if (g_dma_skinny_small && (m >= 2 || g_dma_skinny_small == 2 || blocks_per_cu >= 2.0 || ((g_dma_skinny_m1 & 1) && blocks_per_cu >= 1.5) || ((g_dma_skinny_m1 & 2) && k >= 96 * 128)) && n_rows / 16 < 1024 &&
      (blocks_per_cu >= 1.5 || k >= 96 * 128 || (blocks_per_cu <= 1.0 && k >= 32 * 128)))  // (one slab per CU: the skinny kernel's 16-wave shape, o_proj 4.15-4.44 vs 4.26-4.6 us at 2 .. 7 rows)
    return true;
  const int want = blocks_per_cu <= 1.0 ? 1 : (blocks_per_cu <= 2.0 ? 2 : (blocks_per_cu <= 3.0 ? 3 : 4));
  // (from five rows -- or from the row count at which ONE block's staging reaches 100 KiB: Llama-3-70B's down_proj, K = 28672, two slabs per CU, 56 KiB of x per
  // row: 39 / 64 / 75 us on the streaming kernel at 2 / 3 / 4 rows (one eight-wave block per CU, then row chunks that re-stream the weights) against 31.5 - 35 on
  // the skinny kernel; Llama-3-8B's down_proj at four rows -- 112 KiB, ONE slab per CU -- stays: 9.4 vs 9.9 us.  profiles/r06_decode_cfg.txt (6))
  return (m >= 5 || (size_t)m * (size_t)k * 2 >= 100 * 1024) && (size_t)m * (size_t)k * 2 * want >= 128 * 1024;
}

// K split and ring depth: as many tiles in flight per CU as LDS allows (<= ~150 KiB per CU over the blocks that share it),
// every wave at least 2 steps
bool pick_dma(int m, int n_rows, int k, int ns, DmaCfg& c) {
  // measured on MI355X (tools/gemvd_sweep.py, tools/gemvd_probe.py; profiles/r02_gemvd_*): few waves with a deep ring when many
  // slabs share a CU (gate/up: 4 waves x 8 tiles), 8 waves x 2..4 tiles for 1..2 slabs per CU, 16 waves x 1..2 for a long K
  const int nit = k / kGroup, slabs = n_rows / 16 / ns;
  const double blocks_per_cu = (double)slabs / 256.0;
  // > 3 slabs per CU (the fused gate/up pair): eight-wave blocks whose ring holds a wave's whole K range (K = 4096: four tiles, nothing re-issued inside the
  // loop), three co-resident per CU.  The x staging is m x K x 2 bytes per BLOCK, so half as many blocks stage half as much and leave the ring its depth:
  // gate/up -7 ... -13 % at 2 .. 4 rows, -2.5 % at one row against rounds 2 - 5's four-wave blocks with a ring of seven (profiles/r06_decode_cfg.txt;
  // knob gemvd_wide8: 0 = the four-wave blocks at every m, 1 = eight waves from two rows only, 2 (default) = at every m)
  bool wide8 = blocks_per_cu > 3.0 && nit >= 32 && nit < 96 && m >= (g_dma_wide8 == 2 ? 1 : 2) && g_dma_wide8 != 0;  // (measured at K = 4096 and 8192; the short K ranges of tensor-parallel shards keep the four-wave blocks)
  int waves = nit >= 96 ? 16 : (blocks_per_cu > 3.0 && !wide8 ? 4 : 8);
  if (g_dma_waves) waves = g_dma_waves;
  while (waves > 4 && waves > nit) waves >>= 1;  // (waves beyond the step count idle: their steps are clamped and skipped)
  int want = blocks_per_cu <= 1.0 ? 1 : (blocks_per_cu <= 2.0 ? 2 : (blocks_per_cu <= 3.0 ? 3 : 4));
  if (g_dma_want) want = g_dma_want;  // knob gemvd_want (experiments): blocks per CU the LDS budget is sized for
  if (waves == 4 && want > 3 && !g_dma_four) want = 3;  // (three ring-8 blocks beat four ring-4 ones: 14.8 vs 15.1 us; four ring-7 blocks beat both)
  // x staging is m * k * 2 bytes per block whatever the wave count: when it crowds out the ring, fewer, longer waves
  while (waves > 4 && dma_smem(waves, 1, ns, (nit + waves - 1) / waves, m) > 150 * 1024) waves >>= 1;
  if (waves != 8) wide8 = false;
  const int tx = (nit + waves - 1) / waves;
  int d = tx < 8 ? tx : 8;
  if (waves == 16 && !g_dma_d) d = 1;            // 16 waves per block already keep 16+ KiB per CU in flight (8.2 vs 8.7 us at d = 2)
  if (waves == 8 && !g_dma_d && d > 2 && !wide8 && (m < 3 || blocks_per_cu <= 1.0)) d = 2;  // (two blocks per CU from three rows: a ring of four, qkv -2 %; o_proj keeps two)
  if (wide8 && !g_dma_want) want = tx > 4 ? 2 : 3;  // three eight-wave blocks per CU (two when a wave's K range is eight tiles: K = 8192); the LDS budget below picks the depth (4 / 4 / 2 / 2 at 1 .. 4 rows against K = 4096)
  if (g_dma_d) d = g_dma_d < tx ? g_dma_d : tx;
  // LDS: blocks that want to be co-resident on a CU must fit in 160 KiB
  while (d > 1 && dma_smem(waves, d, ns, tx, m) * want > 156 * 1024) --d;
  // compiled ring depths: 1, 2, 4, 8 (7 with 16 waves: K = 14336)
  if (d == 3) d = 2;
  if (d == 5 || d == 6 || (d == 7 && waves == 8)) d = 4;
  if (d == 8 && waves == 16) d = 7;
  c = {waves, tx, d, dma_smem(waves, d, ns, tx, m)};
  return c.smem <= 160 * 1024 && tx >= d;
}
}  // namespace

int gemv_dma_tune_set(const char* key, int value) {
  if (!strcmp(key, "gemvd_waves")) g_dma_waves = value;
  else if (!strcmp(key, "gemvd_d")) g_dma_d = value;
  else if (!strcmp(key, "gemvd_probe")) g_dma_probe = value;
  else if (!strcmp(key, "gemvd_four")) g_dma_four = value;
  else if (!strcmp(key, "gemvd_want")) g_dma_want = value;
  else if (!strcmp(key, "gemvd_wide8")) g_dma_wide8 = value;
  else if (!strcmp(key, "gemvd_il")) g_dma_il = value;
  else if (!strcmp(key, "decode_skinny_small")) g_dma_skinny_small = value;
  else if (!strcmp(key, "decode_skinny_m1")) g_dma_skinny_m1 = value;
  else if (!strcmp(key, "decode_skinny_from")) g_dma_skinny_from = value;
  else return -1;
  return 0;
}

template <typename DT, int WAVES, int D, int DQ, int EPI>
static void launch_dma_cfg(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                           const DmaCfg& c, hipStream_t st, int f32out) {
  constexpr int NS = EPI == 1 ? 2 : 1;
  auto kern = gemv_dma_kernel<DT, WAVES, D, DQ, EPI>;
  static LdsOptIn optin;
  if (c.smem > 64 * 1024) optin.ensure(reinterpret_cast<const void*>(kern));
  hipLaunchKernelGGL(kern, dim3(n / 16 / NS), dim3(64 * WAVES), c.smem, st, (const uint16_t*)x, (const u32*)qw, (const u32*)szp,
                     (const uint16_t*)bias, (uint16_t*)out, m, n, k, c.tx, (g_dma_probe & 0xFF) | (f32out ? kDmaF32Out : 0) | ((g_dma_il == 1 || (g_dma_il == 2 && c.d < c.tx && c.waves == 8)) ? kDmaInterleave : 0));
}

template <typename DT, int EPI, int DQ>
static int launch_dma_dt(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                         hipStream_t st, int f32out) {
  DmaCfg c;
  if (!pick_dma(m, n, k, EPI == 1 ? 2 : 1, c)) return -1;
#define AWQ_DCASE(W_, D_)                                                           \
  if (c.waves == W_ && c.d == D_) {                                                 \
    launch_dma_cfg<DT, W_, D_, DQ, EPI>(x, qw, szp, bias, out, m, n, k, c, st, f32out); \
    return 0;                                                                       \
  }
  AWQ_DCASE(8, 1) AWQ_DCASE(8, 2) AWQ_DCASE(8, 4) AWQ_DCASE(8, 8)
  AWQ_DCASE(16, 1) AWQ_DCASE(16, 2) AWQ_DCASE(16, 4) AWQ_DCASE(16, 7)
  AWQ_DCASE(4, 1) AWQ_DCASE(4, 2) AWQ_DCASE(4, 4) AWQ_DCASE(4, 7) AWQ_DCASE(4, 8)
#undef AWQ_DCASE
  return -1;
}

// host-side: how launch_gemv_dma serves the call (awq_w4a16_decode_cdna4_plan): weight passes, *kernel 0 streaming / 1 skinny
int gemv_dma_plan(int m, int n, int k, int epi, int* kernel) {
  if (m < 1 || m > 8 || k < 128 || (k % 128) != 0 || epi < 0 || epi > 2 || n < (epi == 1 ? 32 : 16) || (n % (epi == 1 ? 32 : 16)) != 0) return 0;
  if (skinny_takes(m, n, k, epi)) {
    if (kernel) *kernel = 1;
    return 1;
  }
  if (kernel) *kernel = 0;
  DmaCfg c;
  int mc = m;
  while (mc > 1 && !pick_dma(mc, n, k, epi == 1 ? 2 : 1, c)) --mc;
  if (!pick_dma(mc, n, k, epi == 1 ? 2 : 1, c)) return 0;
  return (m + mc - 1) / mc;
}

// epi as in the kernel header; returns -1 if the shape is not served.  The kernel stages every row's x slice in LDS up front
// (m * k * 2 bytes per block): when m rows do not fit (m * k > ~50 k elements: batched decode against K >= 8 k) the rows are
// served in chunks of as many rows as do fit, each chunk re-streaming the weights -- as the reference's GEMV does per row
// (gemv_cuda.cu:187-208 loops over the batch inside one weight pass; here the LDS budget decides).
int launch_gemv_dma(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int epi,
                    int dtype, int szfmt, hipStream_t st, int f32out) {
  if (m < 1 || m > 8 || (k % 128) != 0 || (n % (epi == 1 ? 32 : 16)) != 0 || (f32out && (epi != 0 || bias != nullptr))) return -1;
  if (skinny_takes(m, n, k, epi) && launch_skinny_decode(x, qw, szp, bias, out, m, n, k, epi, dtype, szfmt, st, f32out) == 0) return 0;
  DmaCfg probe_cfg;
  int mc = m;
  while (mc > 1 && !pick_dma(mc, n, k, epi == 1 ? 2 : 1, probe_cfg)) --mc;
  if (mc < m) {
    if (!pick_dma(mc, n, k, epi == 1 ? 2 : 1, probe_cfg)) return -1;
    const size_t ncols = epi ? (size_t)n / 2 : (size_t)n;
    for (int r = 0; r < m; r += mc) {
      const int rows = m - r < mc ? m - r : mc;
      const int rc = launch_gemv_dma((const char*)x + (size_t)r * k * 2, qw, szp, bias, (char*)out + (size_t)r * ncols * (f32out ? 4 : 2), rows, n, k,
                                     epi, dtype, szfmt, st, f32out);
      if (rc != 0) return rc;
    }
    return 0;
  }
#define AWQ_DDT(DT_, DQ_)                                                                   \
  if (epi == 0) return launch_dma_dt<DT_, 0, DQ_>(x, qw, szp, bias, out, m, n, k, st, f32out); \
  if (epi == 1) return launch_dma_dt<DT_, 1, DQ_>(x, qw, szp, bias, out, m, n, k, st, 0);      \
  return launch_dma_dt<DT_, 2, DQ_>(x, qw, szp, bias, out, m, n, k, st, 0);
  if (szfmt == 1) {
    if (dtype == 0) {
      AWQ_DDT(F16, 1)
    }
    AWQ_DDT(BF16, 1)
  }
  if (dtype == 0) {
    AWQ_DDT(F16, 0)
  }
  AWQ_DDT(BF16, 0)
#undef AWQ_DDT
}

}  // namespace awq
