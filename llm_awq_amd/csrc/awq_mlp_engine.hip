// QuantLlamaMLP.forward at decode (one row) as ONE persistent launch: gate/up GEMV + SiLU * mul + down_proj GEMV (gfx950).
// Replaces, for m = 1, the sequence of tinychat/modules/fused_mlp.py:33-83 (two gemv_forward_cuda_new calls, F.silu, a multiply, down_proj's
// gemv_forward_cuda_new -- gemv_kernel, awq/kernels/csrc/quantization_new/gemv/gemv_cuda.cu:74-229) and this library's own two launches.
//
// Structure (DESIGN.md "Decode: the persistent MLP engine"; MI355X_MICROARCH.md price list rows prefetch-credit / ldsdma-fill / nt-weights / allgather):
//   * grid = ONE 16-wave workgroup per CU (256 blocks, 160 KiB of LDS each: co-resident by construction).  Block b owns the gate/up slabs
//     [slab0, slab0 + S_b) (16 rows each: 8 gate + 8 up rows interleaved, so it produces 8 S_b values of h) and down_proj's slab b (16 output rows);
//   * every wave is loader AND consumer of its own tile stream: wave w takes the k-steps {2 w, 2 w + 1} of every gate/up slab (hidden = 4096 = 16 waves
//     x 2 x 128) followed by its n_w k-steps of down_proj's slab -- ONE sequence of 2 S_b + n_w one-KiB tiles through a wave-private ring of 7 slots
//     filled by LDS-DMA (`buffer_load_dwordx4 ... lds`, nt).  The ring knows nothing of the op boundary: down_proj's tiles are requested as soon as gate/up
//     slots drain, so the weights of the dependent op stream while the gate/up tail, the reduction and the hand-over of h run (the "prefetch-credit" a
//     per-slab grid cannot collect: profiles/r04_mlp_one_launch.txt);
//   * a slot is refilled AFTER the math on its tile, not before (a DMA instruction the memory pipeline cannot accept yet holds its wave: with the refill in
//     front, a wave's math waits behind its own request -- profiles/r03_gemvps.txt, "deeper is slower");
//   * the wave's slice of x is loop invariant: it lives in 32 registers as the B operands of v_mfma_f32_16x16x32 (no staging per slab); the per-tile
//     sz_half dwords are fetched once, up front, into registers too (oldest in the vmcnt order);
//   * no workgroup barrier inside the gate/up phase: per slab a wave leaves its 16 fp32 partials in LDS; ONE barrier, then wave 0 sums them in wave (= k)
//     order, applies T(T(silu(T(gate))) * T(up)) and publishes h as 8-byte {2 x T, epoch} granules, one `sc1` (write-through) store each;
//   * every wave gathers the granules of ITS OWN down_proj k range (<= 7 `global_load_dwordx2 sc1` per lane, re-swept until every tag carries the launch's
//     epoch, bounded), stages the data halves in LDS and multiplies; vmcnt is in order, so the gather also covers the wave's down_proj tiles;
//   * a second barrier, wave 0 sums the 16 partials of the block's 16 output rows, bias, store.
// state (shared with the host API): int32 [0] = epoch of the last finished launch (tags are epoch + 1: one buffer serves every call, graph replays included),
// [1] = blocks finished, [2] = sticky error word (a wave gave up waiting: bounded spin; its block's outputs are NaN); granules at byte
// AWQ_MLP_DECODE_COUNTER_BYTES: [ffn / 2] x 8 B.
#include <string.h>

#include "../../include/awq_cdna4.h"
#include "awq_device.hpp"
#include "awq_dma.hpp"
#include "awq_kernels.hpp"

namespace awq {
namespace {

constexpr int kEngWaves = 16, kEngRing = 7, kEngMaxSlabs = 7, kEngHidden = 4096, kEngNout = 4096, kEngBlocks = 256;
constexpr int kEngRingBytes = kEngWaves * kEngRing * 1024;          // 112 KiB
constexpr int kEngPartA = kEngRingBytes;                            // [7 slabs][16 waves][16 floats]
constexpr int kEngH = kEngPartA + kEngMaxSlabs * kEngWaves * 64;    // h staged for down_proj: [k-steps][256 B]
constexpr int kEngPartB = kEngH + kEngMaxSlabs * kEngWaves * 256;   // [16 waves][16 floats]
constexpr int kEngSmem = kEngPartB + kEngWaves * 64;
static_assert(kEngSmem <= 160 * 1024, "LDS");

__device__ __forceinline__ void eng_wait_dyn(int n) {  // s_waitcnt vmcnt(min(n, 6)), n wave-uniform
  if (n >= 6) {  // the steady state: one compare, one branch
    dma_wait_vm<6>();
    return;
  }
  switch (n) {
    case 0: dma_wait_vm<0>(); break;
    case 1: dma_wait_vm<1>(); break;
    case 2: dma_wait_vm<2>(); break;
    case 3: dma_wait_vm<3>(); break;
    case 4: dma_wait_vm<4>(); break;
    case 5: dma_wait_vm<5>(); break;
    default: dma_wait_vm<6>(); break;
  }
}
__device__ __forceinline__ void eng_stamp(unsigned long long* stamps, int slot) {
  if (stamps != nullptr) {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    if ((threadIdx.x & 63) == 0) stamps[((size_t)blockIdx.x * kEngWaves + (threadIdx.x >> 6)) * 8 + slot] = t;
  }
}

}  // namespace

// timing probes (knob mlp_engine_probe; AWQ_PROBES=1 builds only, wrong results): bit 0 = no math, bit 1 = no weight DMA (profiles/r05_mlp_engine.txt)
#ifdef AWQ_ENABLE_PROBES
#define ENG_PROBE(p) (p)
#else
#define ENG_PROBE(p) 0
#endif

template <typename DT>
__global__ __launch_bounds__(64 * kEngWaves) void mlp_engine_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw_gu,
                                                                     const u32* __restrict__ szh_gu, const u32* __restrict__ qw_d,
                                                                     const u32* __restrict__ szh_d, const uint16_t* __restrict__ bias_d,
                                                                     uint16_t* __restrict__ out, int ffn, int* state, unsigned long long* stamps,
                                                                     int probe_) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using vec8 = typename DT::vec8;
  const int probe = ENG_PROBE(probe_);
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int b = blockIdx.x;
  constexpr int nitA = kEngHidden / 128;  // 32
  const int nitB = ffn >> 7;
  // this block's gate/up slabs, this wave's down_proj k-steps
  const int slabsA = ffn >> 3;  // 2 ffn rows / 16
  const int baseA = slabsA / kEngBlocks, remA = slabsA % kEngBlocks;
  const int slab0 = b * baseA + min(b, remA), S_b = baseA + (b < remA ? 1 : 0);
  const int baseB = nitB / kEngWaves, remB = nitB % kEngWaves;
  const int ks0 = wv * baseB + min(wv, remB), n_w = baseB + (wv < remB ? 1 : 0);
  const int T_A = 2 * S_b, T = T_A + n_w;
  eng_stamp(stamps, 0);

  // the epoch word (stable: only the LAST block to leave advances it).  Inline asm like every load of the prologue: hipcc would otherwise put the wait for it
  // in front of its FIRST USE -- behind the gate/up phase, as a vmcnt(0) that also waits for the wave's down_proj tiles; here the first tile wait covers it
  u32 epoch_m1;
  asm volatile("global_load_dword %0, %1, off sc1" : "=v"(epoch_m1) : "v"(state) : "memory");
  uint64_t* gran = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(state) + AWQ_MLP_DECODE_COUNTER_BYTES);

  const __amdgpu_buffer_rsrc_t rwA = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(qw_gu), 0, slabsA * nitA * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rwD = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(qw_d), 0, (kEngNout >> 4) * nitB * 1024, 0x00020000);
  char* ring = smem + wv * (kEngRing * 1024);
  const u32 lane16 = lane * 16u;
  auto issue = [&](int tt, int slot) {  // tile tt of this wave's sequence into ring slot `slot`
    if (probe & 2) return;
    if (tt < T_A) {
      const u32 tile = (u32)(slab0 + (tt >> 1)) * (u32)nitA + (u32)(2 * wv + (tt & 1));
      dma_to_lds<16, 2>(rwA, ring + slot * 1024, lane16, tile * 1024u);
    } else {
      const u32 tile = (u32)b * (u32)nitB + (u32)(ks0 + (tt - T_A));
      dma_to_lds<16, 2>(rwD, ring + slot * 1024, lane16, tile * 1024u);
    }
  };

  // ---- up front: x (B operands: 16 bytes at k = 128 ks + 32 a + 8 g), the sz_half dword of every tile of the sequence (lane's row = i), then the ring.
  // Unconditional loads (clamped indices): a load issued on one side of a branch would have its result register copied at the join before the data is there.
  u32x4 xo[2][4];
  u32 szA[2 * kEngMaxSlabs], szB[kEngMaxSlabs];
  {
    const char* xb = reinterpret_cast<const char*>(x) + (size_t)(2 * wv) * 256 + g * 16;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const char* p = xb + kk * 256 + a * 64;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(xo[kk][a]) : "v"(p) : "memory");
      }
#pragma unroll
    for (int t = 0; t < 2 * kEngMaxSlabs; ++t) {
      const int tc = min(t, max(T_A - 1, 0));
      const u32* p = szh_gu + ((size_t)min(slab0 + (tc >> 1), slabsA - 1) * nitA + (size_t)(2 * wv + (tc & 1))) * 16 + i;
      asm volatile("global_load_dword %0, %1, off" : "=v"(szA[t]) : "v"(p) : "memory");
    }
#pragma unroll
    for (int j = 0; j < kEngMaxSlabs; ++j) {
      const int jc = min(j, max(n_w - 1, 0));
      const u32* p = szh_d + ((size_t)b * nitB + (size_t)min(ks0 + jc, nitB - 1)) * 16 + i;
      asm volatile("global_load_dword %0, %1, off" : "=v"(szB[j]) : "v"(p) : "memory");
    }
  }
#pragma unroll
  for (int d = 0; d < kEngRing; ++d)
    if (d < T) issue(d, d);
  // everything older than the ring has arrived once the first tile wait passes; tie the registers to a wait so no use is scheduled above it
  {
    const int first = max(min(T, kEngRing) - 1, 0);  // tiles that may stay in flight behind tile 0
    eng_wait_dyn(probe & 2 ? 0 : first);
    asm volatile("" : "+v"(xo[0][0]), "+v"(xo[0][1]), "+v"(xo[0][2]), "+v"(xo[0][3]), "+v"(xo[1][0]), "+v"(xo[1][1]), "+v"(xo[1][2]), "+v"(xo[1][3]) : : "memory");
    asm volatile("" : "+v"(szA[0]), "+v"(szA[1]), "+v"(szA[2]), "+v"(szA[3]), "+v"(szA[4]), "+v"(szA[5]), "+v"(szA[6]), "+v"(szA[7]), "+v"(szA[8]), "+v"(szA[9]),
                      "+v"(szA[10]), "+v"(szA[11]), "+v"(szA[12]), "+v"(szA[13])
                 :
                 : "memory");
    asm volatile("" : "+v"(szB[0]), "+v"(szB[1]), "+v"(szB[2]), "+v"(szB[3]), "+v"(szB[4]), "+v"(szB[5]), "+v"(szB[6]), "+v"(epoch_m1) : : "memory");
  }
  const u32 epoch = (u32)__builtin_amdgcn_readfirstlane((int)epoch_m1) + 1u;
  eng_stamp(stamps, 1);

  Cdna4DequantH<DT> ch;
  ch.init(lane);
  const u32 lds0 = (u32)(size_t)(__attribute__((address_space(3))) char*)smem;
  const u32 ring_lane = lds0 + wv * (kEngRing * 1024) + lane16;

  // =========================== phase A: gate/up, S_b slabs x this wave's two k-steps ===========================
  static_for<0, kEngMaxSlabs>([&](auto s_) {
    constexpr int S = decltype(s_)::value;
    if (S < S_b) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      static_for<0, 2>([&](auto k_) {
        constexpr int KK = decltype(k_)::value;
        constexpr int t = 2 * S + KK;
        constexpr int slot = t % kEngRing;
        if (!(probe & 2) && t > 0) eng_wait_dyn(T - 1 - t);  // (t = 0: waited above)
        u32x4 w;
        asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(ring_lane), "n"(slot * 1024) : "memory");
        if (!(probe & 1)) {
          vec8 op[4];
          ch.tile(w, szA[t], op);
#pragma unroll
          for (int a = 0; a < 4; ++a) acc = DT::mfma(op[a], __builtin_bit_cast(vec8, xo[KK][a]), acc);
        }
        if (t + kEngRing < T) issue(t + kEngRing, slot);  // the slot's bytes are in registers and its math is done: refill it
      });
      // acc[r] = the slab's row 4 g + r (every column i holds the same sum: one activation row); lanes i == 0 leave it for the reduction
      if (i == 0) {
        const u32 pa = lds0 + kEngPartA + (S * kEngWaves + wv) * 64 + g * 16;
        asm volatile("ds_write_b128 %0, %1" : : "v"(pa), "v"(acc) : "memory");
      }
    }
  });
  eng_stamp(stamps, 2);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");

  // =========================== the edge: wave 0 reduces, applies the tail and publishes h; every wave gathers its own k range ===========================
  if (wv == 0) {
    // lane l < 4 S_b: slab l / 4, rows 4 (l % 4) .. + 3 of it summed over the 16 waves in wave (= k) order
    // rows 0..7 of a slab are gate rows, 8..15 the matching up rows: lane (sl, rq < 2) sums gate rows 4 rq .. + 3 AND up rows 8 + 4 rq .. + 3 itself
    // (no cross-lane op here: hipcc puts `s_waitcnt vmcnt(0)` in front of any DS instruction it can see while LDS-DMA is in flight, and this
    // wave's down_proj tiles are -- the publish must not wait for them)
    const int sl = lane >> 1, rq = lane & 1;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f}, up = {0.f, 0.f, 0.f, 0.f};
    if (sl < S_b) {
      const u32 pa = lds0 + kEngPartA + sl * (kEngWaves * 64) + rq * 16;
      static_for<0, kEngWaves>([&](auto q_) {
        constexpr int Q = decltype(q_)::value;
        f32x4 p, q;
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(p), "=&v"(q)
                     : "v"(pa), "n"(Q * 64), "n"(Q * 64 + 32)
                     : "memory");
        sum += p;
        up += q;
      });
    }
    if (sl < S_b) {
      auto to_f = [](uint16_t v) { return DT::to_float(v); };
      u32 hv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // fused_mlp.py:79-82: c = F.silu(gate_output) * up_output, every op rounded to T
        const float gt = to_f(DT::from_float(sum[r])), uu = to_f(DT::from_float(up[r]));
        const float sv = to_f(DT::from_float(silu_f32(gt)));
        hv[r] = (u32)DT::from_float(sv * uu);
      }
      // h index = 8 (slab0 + sl) + 4 rq + r -> granule (h index) / 2
      uint64_t* dst = gran + (size_t)(slab0 + sl) * 4 + rq * 2;
      const u32x2 g0 = {hv[0] | (hv[1] << 16), epoch}, g1 = {hv[2] | (hv[3] << 16), epoch};
      asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(dst), "v"(g0) : "memory");
      asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(dst + 1), "v"(g1) : "memory");
    }
  }
  eng_stamp(stamps, 3);

  bool gave_up = false;
  {
    // this wave's granule range: k-steps [ks0, ks0 + n_w) = granules [64 ks0, 64 (ks0 + n_w)); lane takes granule 64 j + lane of k-step j
    const int g0 = ks0 * 64, g1 = (ks0 + n_w) * 64;
    if (n_w > 0) {
      const u32 hb = lds0 + kEngH + ks0 * 256 + lane * 4;
      int spins = 0;
      for (;;) {
        u32x2 gv[kEngMaxSlabs];
#pragma unroll
        for (int j = 0; j < kEngMaxSlabs; ++j) {
          const uint64_t* gp = gran + min(g0 + j * 64 + lane, g1 - 1);
          asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(gv[j]) : "v"(gp) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(gv[0]), "+v"(gv[1]), "+v"(gv[2]), "+v"(gv[3]), "+v"(gv[4]), "+v"(gv[5]), "+v"(gv[6]) : : "memory");
        bool bad = false;
#pragma unroll
        for (int j = 0; j < kEngMaxSlabs; ++j) bad = bad || gv[j].y != epoch;
        if (__builtin_amdgcn_ballot_w64(bad) == 0ull) {
#pragma unroll
          for (int j = 0; j < kEngMaxSlabs; ++j)
            if (j < n_w) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(hb), "v"(gv[j].x), "n"(j * 256) : "memory");
          break;
        }
        if (++spins > 200000) {  // a lost producer: a flagged error and NaN outputs, not a hung queue or a plausible number
          if (lane == 0) __hip_atomic_store(state + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          gave_up = true;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : : : "memory");  // h is staged; vmcnt is in order, so the wave's down_proj tiles have landed too
  }
  eng_stamp(stamps, 4);

  // =========================== phase B: down_proj, this wave's n_w k-steps of slab b ===========================
  {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (gave_up) {
      const float nanv = __builtin_nanf("");
      acc = f32x4{nanv, nanv, nanv, nanv};
    }
    const u32 hl = lds0 + kEngH + ks0 * 256 + g * 16;
    static_for<0, kEngMaxSlabs>([&](auto j_) {
      constexpr int J = decltype(j_)::value;
      if (J < n_w) {
        const int slot = (T_A + J) % kEngRing;
        u32x4 w, xb[4];
        asm volatile("ds_read_b128 %0, %1" : "=v"(w) : "v"(ring_lane + (u32)slot * 1024u) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xb[0]) : "v"(hl), "n"(J * 256) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xb[1]) : "v"(hl), "n"(J * 256 + 64) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xb[2]) : "v"(hl), "n"(J * 256 + 128) : "memory");
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xb[3]) : "v"(hl), "n"(J * 256 + 192) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w), "+v"(xb[0]), "+v"(xb[1]), "+v"(xb[2]), "+v"(xb[3]) : : "memory");
        if (!(probe & 1)) {
          vec8 op[4];
          ch.tile(w, szB[J], op);
#pragma unroll
          for (int a = 0; a < 4; ++a) acc = DT::mfma(op[a], __builtin_bit_cast(vec8, xb[a]), acc);
        }
      }
    });
    if (i == 0) {
      const u32 pb = lds0 + kEngPartB + wv * 64 + g * 16;
      asm volatile("ds_write_b128 %0, %1" : : "v"(pb), "v"(acc) : "memory");
    }
  }
  eng_stamp(stamps, 5);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : : : "memory");

  if (wv == 0) {
    if (lane < 4) {  // rows 4 lane .. + 3 of the block's 16 outputs, the 16 waves' partials in wave (= k) order
      const u32 pb = lds0 + kEngPartB + lane * 16;
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
      static_for<0, kEngWaves>([&](auto q_) {
        constexpr int Q = decltype(q_)::value;
        f32x4 p;
        asm volatile("ds_read_b128 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(p) : "v"(pb), "n"(Q * 64) : "memory");
        sum += p;
      });
      const int n0 = b * 16 + lane * 4;
      uint16_t o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        o[r] = DT::from_float(sum[r]);
        if (bias_d != nullptr) o[r] = DT::from_float(DT::to_float(o[r]) + DT::to_float(bias_d[n0 + r]));  // `out + self.bias` in T (qmodule.py:221)
      }
      *reinterpret_cast<u32x2*>(out + n0) = u32x2{(u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16)};
    }
    if (lane == 0 && __hip_atomic_fetch_add(state + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kEngBlocks - 1) {
      // last block: every block has gathered its h; the next launch on this state starts after this one ends and tags with the next epoch
      __hip_atomic_store(state + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(state, (int)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  eng_stamp(stamps, 6);
}

namespace {
unsigned long long* g_eng_stamps = nullptr;
int g_eng_probe = 0;
}  // namespace
void mlp_engine_set_stamps(void* device_u64) { g_eng_stamps = reinterpret_cast<unsigned long long*>(device_u64); }
void mlp_engine_set_probe(int v) { g_eng_probe = v; }

size_t mlp_decode_state_bytes(int m, int ffn) { return (size_t)AWQ_MLP_DECODE_COUNTER_BYTES + (size_t)m * ffn * 4; }

// host-side: is (m, hidden, ffn, n_out) served?  One row; hidden = 4096 (16 waves x 2 k-steps: x lives in registers) and n_out = 4096 (one down_proj
// slab per CU of the 256); ffn a multiple of 128 with at most 7 gate/up slabs per block and 7 down_proj k-steps per wave (the ring and the sz_half
// registers are sized for them): 128 <= ffn <= 14336 -- Llama-3-8B (14336) and Llama-2-7B (11008) MLPs.  The 256 blocks must be co-resident: the
// whole MI355X (a partitioned or CU-masked device runs the two launches).
int mlp_decode_plan(int m, int hidden, int ffn, int n_out) {
  if (m != 1 || hidden != kEngHidden || n_out != kEngNout || ffn < 128 || (ffn % 128) != 0) return 0;
  const int slabsA = ffn / 8, nitB = ffn / 128;
  if ((slabsA + kEngBlocks - 1) / kEngBlocks > kEngMaxSlabs || (nitB + kEngWaves - 1) / kEngWaves > kEngMaxSlabs) return 0;
  return device_cu_count() >= kEngBlocks ? 1 : 0;
}

int launch_mlp_decode(const void* x, const void* qw_gu, const void* szh_gu, const void* qw_d, const void* szh_d, const void* bias_d,
                      void* out, int m, int hidden, int ffn, int n_out, int dtype, int* state, hipStream_t st) {
  if (!mlp_decode_plan(m, hidden, ffn, n_out)) return -1;
#define AWQ_ENG(DT_)                                                                                                                      \
  {                                                                                                                                       \
    auto kern = mlp_engine_kernel<DT_>;                                                                                                   \
    static LdsOptIn optin;                                                                                                                \
    optin.ensure(reinterpret_cast<const void*>(kern), kEngSmem);                                                                          \
    hipLaunchKernelGGL(kern, dim3(kEngBlocks), dim3(64 * kEngWaves), kEngSmem, st, (const uint16_t*)x, (const u32*)qw_gu,                 \
                       (const u32*)szh_gu, (const u32*)qw_d, (const u32*)szh_d, (const uint16_t*)bias_d, (uint16_t*)out, ffn, state,      \
                       g_eng_stamps, g_eng_probe);                                                                                        \
  }
  if (dtype == 0) AWQ_ENG(F16) else AWQ_ENG(BF16)
#undef AWQ_ENG
  return 0;
}

}  // namespace awq
