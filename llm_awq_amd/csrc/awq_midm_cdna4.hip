// Mid-M GEMM on cdna4-interleaved weights, up to 128 rows per pass, bf16 and fp16 (gfx950): prompts of 65 .. 192 rows (and 17 .. 32 rows of very wide
// layers) -- the row range the reference serves with gemm_w4a16_T1's 16 / 32 / 64-row tiles + split-K
// (awq/kernels/csrc/quantization_new/gemm/gemm_cuda.cu:1155-1206, split-K epilogue :546-619, semaphore.h:44-103).
//
// Shape of the work (DESIGN.md "Mid-M"): every packed byte is read once, and the activations are NOT negligible: a block that owns S slabs moves
// M x 256 B of x per k-step for S KiB of weights.  So
//   * the x tile of a k-step (16 CB rows x 128 k) lands ONCE per block in LDS -- LDS-DMA (`buffer_load_dwordx4 ... lds`, no VGPRs), a ring of DX
//     stages, XOR-swizzled on the SOURCE side so that the 16 x rows of an MFMA operand are one conflict-free ds_read_b128;
//   * the block's WAVES waves split N: wave w streams its own NS 16-row slabs (1-KiB tiles + a 256-byte piece of scale dwords by LDS-DMA into a
//     wave-private ring of DX + 1 slots), dequantises them on the matrix core (Cdna4DequantT / H: exact q s + sz, one v_cvt_pk = the reference's single
//     rounding) and multiplies them against the SHARED x tile: NS x CB x 4 v_mfma_f32_16x16x32 per k-step, fp32 accumulation; the NEXT k-step's tile is
//     dequantised word by word between this k-step's product MFMAs;
//   * the chip is filled by a K split ACROSS blocks (grid.y parts of whole quantisation groups): every block stores its fp32 sums write-through, draws
//     a ticket of its slab group, and the block that draws the last one adds the parts IN PART ORDER (deterministic), rounds once, adds the bias
//     (or applies QuantLlamaMLP's SiLU * mul) -- the reference's split_k_iters + Semaphore without a second launch and without a waiting block.
// One s_barrier per k-step; a wave's VMEM queue per k-step is a fixed group {XP x pieces, NS tiles, NS scale pieces} (steps past the block's K range are
// issued with an out-of-bounds buffer offset: no traffic, same count), so the wait for a stage is a COUNTED vmcnt and nothing ever drains the queue.
// vmcnt is ONE in-order counter: the x ring is as deep as the weight ring -- the DX - 1 groups in flight are what the CU has outstanding.
// All LDS reads of the loop are inline asm (hipcc would put vmcnt(0) in front of every LDS read that may alias an in-flight DMA).
#include <string.h>

#include <atomic>
#include <mutex>
#include <type_traits>

#include "awq_device.hpp"
#include "awq_dma.hpp"
#include "awq_kernels.hpp"

#ifdef AWQ_ENABLE_PROBES
// (AWQ_PROBES builds only; tools/midm_stamps.py) per-block time stamps of a launch: {start, loop start, loop end, end} x {s_memrealtime (100 MHz), s_memtime (shader clock)}
__device__ unsigned long long g_midm_stamps[2048 * 8];
extern "C" __attribute__((visibility("default"))) int awq_dev_midm_stamps(void* host_dst, int blocks) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_midm_stamps), (size_t)blocks * 64, 0, hipMemcpyDeviceToHost);
}
#define MIDM_STAMP(j)                                                                \
  if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 2048) {                    \
    g_midm_stamps[blockIdx.x * 8 + 2 * (j)] = __builtin_amdgcn_s_memrealtime();      \
    g_midm_stamps[blockIdx.x * 8 + 2 * (j) + 1] = __builtin_readcyclecounter();      \
  }
#else
#define MIDM_STAMP(j)
#endif

namespace awq {

namespace {
constexpr u32 kOob = 0x80000000u;  // added to a buffer voffset: beyond num_records (< 2^31 for every buffer of this file) -> the load reads 0, no memory traffic
}

// the x fragments of one 32-k slice: rows 16 C + i of the stage (asm: the compiler must not see an LDS read behind the in-flight DMAs)
template <int C, int CB>
__device__ __forceinline__ void midm_read_x(u32x4 (&dst)[CB], u32 addr) {
  if constexpr (C < CB) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst[C]) : "v"(addr), "n"(C * 4096) : "memory");
    midm_read_x<C + 1, CB>(dst, addr);
  }
}

// slab S's tile words + scale dword of a ring slot -> registers
template <int S>
__device__ __forceinline__ void midm_read_w(u32x4& w, u32& sz, u32 waddr, u32 saddr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w) : "v"(waddr), "n"(S * 1280) : "memory");
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(sz) : "v"(saddr), "n"(S * 1280) : "memory");
}

// ---- the K split's hand-over (both mid-M kernels): the block has stored its fp32 part write-through; it draws a ticket of its slab group, and the block that
// draws the last one adds the KS parts IN PART ORDER, rounds once and applies the epilogue.  Group tile: M rows x NSB * 16 columns from column nb * NSB * 16 ----
template <typename DT, int THREADS, int NSB>
__device__ __forceinline__ void midm_reduce_parts(char* smem, const float* __restrict__ parts, u32* __restrict__ tickets, const uint16_t* __restrict__ bias,
                                                  uint16_t* __restrict__ out, int M, int N, int KS, int nb, int epi, int f32out) {
  const int nslab = N >> 4;
  auto to_f = [](uint16_t b) { return DT::to_float(b); };
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");  // this thread's write-through stores have been acknowledged
  __syncthreads();                                     // ... and every other thread's
  u32* bc = reinterpret_cast<u32*>(smem);
  // (relaxed: the release is the acknowledged write-through stores above, the acquire the sc1 loads below)
  if (threadIdx.x == 0) bc[0] = __hip_atomic_fetch_add(tickets + nb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (bc[0] != (u32)(KS - 1)) return;
  if (threadIdx.x == 0) __hip_atomic_store(tickets + nb, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // all KS tickets are drawn: the word is ready for its next launch
  // the group's tile: M rows x NSB * 16 columns.  Work item = (row, column quad); epi 2: (row, slab, half) = a gate quad and its up quad
  const int col0 = nb * NSB * 16;
  auto sum_quad = [&](int m, int col) {
    f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int p0 = 0; p0 < KS; p0 += 4) {  // up to four parts in flight, added in part order
      // (every load unconditional, the part index clamped: a load under a branch would make hipcc merge its destination with the untaken
      // path's value BEFORE the wait below)
      f32x4 v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* src = parts + ((size_t)min(p0 + j, KS - 1) * M + m) * N + col;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[j]) : "v"(src) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : : "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (p0 + j < KS) a += v[j];
    }
    return a;
  };
  if (epi == 2) {
    constexpr int QC = NSB * 2;
    for (int idx = threadIdx.x; idx < M * QC; idx += THREADS) {
      const int m = idx / QC, q = idx - m * QC, sl = q >> 1, h = q & 1;
      const int slab = nb * NSB + sl;
      if (slab >= nslab) continue;
      const f32x4 gt4 = sum_quad(m, col0 + sl * 16 + 4 * h), up4 = sum_quad(m, col0 + sl * 16 + 8 + 4 * h);
      uint16_t o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float gt = to_f(DT::from_float(gt4[r])), u = to_f(DT::from_float(up4[r]));
        const float sl2 = to_f(DT::from_float(silu_f32(gt)));
        o[r] = DT::from_float(sl2 * u);
      }
      *reinterpret_cast<u32x2*>(out + (size_t)m * (N >> 1) + slab * 8 + 4 * h) = u32x2{(u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16)};
    }
  } else {
    constexpr int QC = NSB * 4;
    for (int idx = threadIdx.x; idx < M * QC; idx += THREADS) {
      const int m = idx / QC, q = idx - m * QC, col = col0 + 4 * q;
      if (col >= nslab * 16) continue;
      const f32x4 a = sum_quad(m, col);
      if (f32out) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (size_t)m * N + col) = a;
      } else {
        uint16_t o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          o[r] = DT::from_float(a[r]);
          if (bias != nullptr) o[r] = DT::from_float(to_f(o[r]) + to_f(bias[col + r]));
        }
        *reinterpret_cast<u32x2*>(out + (size_t)m * N + col) = u32x2{(u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16)};
      }
    }
  }
}

// epi 0: out[m, n] (+ bias);  epi 2: 8 + 8 interleaved gate / up slabs, out[m, n / 2] = silu(gate) * up (fused_mlp.py:79-82)
// f32out (epi 0, no bias): out = float [m, n], the unrounded sums (K shard of a tensor-parallel row split)
template <typename DT, int WAVES, int NS, int CB, int DX, int DW, int DQ, int PROBE>
__global__ __launch_bounds__(64 * WAVES) void midm_kernel(const uint16_t* __restrict__ x, const u32* __restrict__ qw, const u32* __restrict__ szp,
                                                        const uint16_t* __restrict__ bias, uint16_t* __restrict__ out, float* __restrict__ parts,
                                                        u32* __restrict__ tickets, int M, int N, int K, int epi, int f32out, int probe_) {
  MIDM_STAMP(0)
  using vec8 = typename DT::vec8;
  const int probe = PROBE ? probe_ : 0;  // timing probes (AWQ_PROBES builds only; wrong results): bit 0 no x traffic, 1 no weight traffic, 2 no LDS reads / math, 3 no barrier, 4 no vmcnt wait, 5 no waits for the x fragments
  static_assert(DX >= 2 && DW >= DX + 1, "the packed words of step t + 1 must have landed when step t's x stage has");
  constexpr int NSB = WAVES * NS;
  constexpr int XPT = 4 * CB, XP = (XPT + WAVES - 1) / WAVES;   // 1-KiB staging pieces per stage / per wave
  constexpr int XSTAGE = XP * WAVES * 1024;                     // bytes per x stage (16 CB rows x 256 B, padded to whole rounds of pieces: surplus pieces land in the padding)
  constexpr int OPS = XP + 2 * NS;                              // VMEM operations of one issue group
  constexpr int WSLOT = 1024 + 256;                             // ring slot of one slab: the tile + its scale dwords (64 B used; the piece is 256 B)
  constexpr int WWAVE = DW * NS * WSLOT;                        // a wave's ring
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [DX x stages][WAVES rings]
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = lane & 15, g = lane >> 4;
  const int nit = K >> 7, nslab = N >> 4;
  const int KS = gridDim.y, p = blockIdx.y, nb = blockIdx.x;
  const int k0 = (int)(((long)p * nit) / KS), kn = (int)(((long)(p + 1) * nit) / KS) - k0;  // this block's k-steps [k0, k0 + kn)

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(qw), 0, nslab * nit * 1024, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32*>(szp), 0, nslab * nit * 64, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(x), 0, M * K * 2, 0x00020000);

  // LDS addresses are kept as integers (the M0 values of the DMAs / the VGPR addresses of the reads): no pointer casts inside the loop
  const u32 lds0 = (u32)(size_t)(__attribute__((address_space(3))) char*)smem;
  const u32 xpiece0 = lds0 + (u32)wv * 1024u;                          // this wave's piece q of stage 0: + q * WAVES KiB
  const u32 wring0 = lds0 + (u32)(DX * XSTAGE) + (u32)wv * (u32)WWAVE;  // this wave's ring, slot 0

  // ---- this wave's slabs (slabs past the matrix are never fetched and never stored) ----
  u32 wsoff[NS], wvoff[NS], svoff[NS];  // soffset of the slab's tile of step k0 (bytes); lane offsets of the tile / scale-piece loads
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int slab = nb * NSB + wv * NS + s;
    const u32 dead = (slab < nslab && !(probe & 2)) ? 0u : kOob;  // (probe bit 1, experiments: no weight traffic)
    wsoff[s] = ((u32)min(slab, nslab - 1) * (u32)nit + (u32)k0) * 1024u;
    wvoff[s] = lane * 16u + dead;
    svoff[s] = lane * 4u + dead;  // (lanes 16 .. 63 fetch the dwords of the next three k-steps: one 256-byte piece, 64 B used)
  }
  // ---- this wave's x staging pieces: piece b = wv + WAVES q covers LDS rows 4b .. 4b + 3 of a stage; lane -> row 4b + lane / 16, slot lane % 16,
  //      which receives source granule slot ^ (row & 15) of x row min(row, M - 1); pieces past the stage's rows fetch nothing ----
  u32 xvoff[XP];
#pragma unroll
  for (int q = 0; q < XP; ++q) {
    const int b = wv + WAVES * q, r = 4 * b + g;
    xvoff[q] = (b < XPT && !(probe & 1)) ? ((u32)min(r, M - 1) * (u32)K + (u32)((i ^ (r & 15)) * 8)) * 2u : kOob;
  }
  // the issue group of a k-step: x pieces of step tx -> the stage at xdst, packed words + scale dwords of step tw -> the ring slot at wdst.
  // skip = kOob for steps past the block's K range: issued out of bounds -- no traffic, same operation count
  auto issue_x = [&](u32 xdst, u32 tx, u32 skip) {
    const u32 soff = ((u32)k0 + tx) * 256u;
#pragma unroll
    for (int q = 0; q < XP; ++q) dma_to_lds_at<16, 0>(rx, xdst + (u32)(q * WAVES * 1024), xvoff[q] | skip, soff);
  };
  auto issue_w = [&](u32 wdst, u32 tw, u32 skip) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const u32 so = wsoff[s] + tw * 1024u;
      dma_to_lds_at<16, 2>(rw, wdst + (u32)(s * WSLOT), wvoff[s] | skip, so);  // aux 2 = nt: every packed byte is read once
      dma_to_lds_at<4, 0>(rs, wdst + (u32)(s * WSLOT + 1024), svoff[s] | skip, so >> 4);
    }
  };

  Cdna4DequantT<DT> cd;
  Cdna4DequantH<DT> ch;
  if (DQ == 0) cd.init(lane);
  else ch.init(lane);

  f32x4 acc[NS][CB];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int c = 0; c < CB; ++c) acc[s][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32 xrd[4];  // x fragment addresses in stage 0: rows 16c + i, granule (4a + g) ^ i
#pragma unroll
  for (int a = 0; a < 4; ++a) xrd[a] = lds0 + (u32)(i * 256 + (((4 * a + g) ^ i) << 4));
  const u32 wrd = wring0 + lane * 16u;          // this lane's 16 B of slab 0's tile in ring slot 0
  const u32 srd = wring0 + 1024u + (u32)i * 4u;  // ... and its scale dword

  // ---- prologue: the groups of steps -(DW - 1) .. -1: packed words of steps 0 .. DW - 2, x stages of steps 0 .. DX - 2 (the youngest DX - 1 groups are full) ----
  static_for<0, DW - 1>([&](auto j_) {
    constexpr int J = decltype(j_)::value;  // group J - (DW - 1)
    asm volatile("" ::: "memory");
    if constexpr (J - (DW - 1) + (DX - 1) >= 0) {
      constexpr int TX = J - (DW - 1) + (DX - 1);
      issue_x(xpiece0 + (u32)(TX * XSTAGE), (u32)TX, TX < kn ? 0u : kOob);
    }
    issue_w(wring0 + (u32)(J * NS * WSLOT), (u32)J, J < kn ? 0u : kOob);
    asm volatile("" ::: "memory");
  });

  // tile registers of one k-step, and their dequantisation word by word
  u32x4 wn[NS];
  u32 szn[NS];
  auto read_w = [&](u32 woff) {  // woff: byte offset of the ring slot
    const u32 wa = wrd + woff, sa = srd + woff;
    static_for<0, NS>([&](auto s_) {
      constexpr int S = decltype(s_)::value;
      midm_read_w<S>(wn[S], szn[S], wa, sa);
    });
  };
  vec8 op[NS][4];
  typename Cdna4DequantT<DT>::Prep pt[NS];
  typename Cdna4DequantH<DT>::Prep ph[NS];
  auto prep_all = [&]() {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if (DQ == 0) pt[s] = cd.prep(szn[s]);
      else ph[s] = ch.prep(szn[s]);
    }
  };
  auto dq_word = [&](int s, u32 w) { return DQ == 0 ? cd.word(w, pt[s]) : ch.word(w, ph[s]); };

  // step 0's tile: dequantised up front
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"((DX - 2) * OPS) : "memory");
  read_w(0u);
  {
    asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
#pragma unroll
    for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(wn[s]), "+v"(szn[s]));
    prep_all();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      op[s][0] = dq_word(s, wn[s].x);
      op[s][1] = dq_word(s, wn[s].y);
      op[s][2] = dq_word(s, wn[s].z);
      op[s][3] = dq_word(s, wn[s].w);
    }
  }

  // loop state (scalar): byte offsets of the x stage / ring slot of step t and of step t - 1 (= where step t's group lands: stage (t + DX - 1) % DX, slot (t + DW - 1) % DW)
  u32 xs = 0, xprev = (u32)((DX - 1) * XSTAGE), ws = 0, wprev = (u32)((DW - 1) * NS * WSLOT);
  int t = 0;
  auto step = [&](auto tail_) {
    constexpr bool TAIL = decltype(tail_)::value;
    // this wave's pieces of stage t (and its packed words of steps t, t + 1) have landed: the DX - 2 youngest groups may stay in flight;
    // behind the barrier every wave's pieces have, and every wave is done reading stage t - 1
    if (!(probe & 16)) asm volatile("s_waitcnt vmcnt(%0)" : : "n"((DX - 2) * OPS) : "memory");
    if (!(probe & 8)) asm volatile("s_barrier" : : : "memory");
    issue_x(xpiece0 + xprev, (u32)(t + DX - 1), (!TAIL || t + DX - 1 < kn) ? 0u : kOob);
    issue_w(wring0 + wprev, (u32)(t + DW - 1), (!TAIL || t + DW - 1 < kn) ? 0u : kOob);
    asm volatile("" ::: "memory");
    const u32 wnext = ws + (u32)(NS * WSLOT) == (u32)WWAVE ? 0u : ws + (u32)(NS * WSLOT);
    if (!(probe & 4)) {
      // the NEXT step's tile -> registers (its dequant is spread over this step's product MFMAs), then the x fragments, 32-k slice by slice
      read_w(wnext);
      u32x4 xo[2][CB];
      midm_read_x<0, CB>(xo[0], xrd[0] + xs);
      static_for<0, 4>([&](auto a_) {
        constexpr int A = decltype(a_)::value;
        if constexpr (A < 3) midm_read_x<0, CB>(xo[(A + 1) & 1], xrd[A + 1] + xs);
        // LDS operations return in order: [next tile (2 NS)] [slice 0 (CB)] [slice 1 (CB)] ...: slice A + 1's CB reads may stay outstanding
        if (!(probe & 32)) {  // (probe bit 5, experiments: no waits for the x fragments -- timing only)
          if constexpr (A < 3) asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(CB) : "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory");
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) asm volatile("" : "+v"(xo[A & 1][c]));
        if constexpr (A == 0) {
#pragma unroll
          for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(wn[s]), "+v"(szn[s]));
          prep_all();
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
          for (int c = 0; c < CB; ++c) acc[s][c] = DT::mfma(op[s][A], __builtin_bit_cast(vec8, xo[A & 1][c]), acc[s][c]);
          // word A of the next tile takes the operand's place (independent of the products above and of slice A + 1's)
          op[s][A] = dq_word(s, A == 0 ? wn[s].x : (A == 1 ? wn[s].y : (A == 2 ? wn[s].z : wn[s].w)));
        }
      });
    }
    xprev = xs;
    xs = xs + (u32)XSTAGE == (u32)(DX * XSTAGE) ? 0u : xs + (u32)XSTAGE;
    wprev = ws;
    ws = wnext;
    ++t;
  };
  MIDM_STAMP(1)
  while (t + DW - 1 < kn) step(std::false_type{});  // every group in range
  while (t < kn) step(std::true_type{});            // the last DW - 1 steps: groups past the K range are issued out of bounds
  MIDM_STAMP(2)
  asm volatile("s_waitcnt vmcnt(0)" : : : "memory");  // the out-of-range tail groups (no traffic) have retired: LDS may be re-used
  // the K split stores the accumulators with inline-asm (write-through) stores: the compiler's hazard recogniser does not see a VMEM read of the last
  // MFMAs' destination registers there -- keep the required wait states (ISA: XDL write VGPR -> VMEM read) between them by hand
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int c = 0; c < CB; ++c) asm volatile("" : "+v"(acc[s][c]));
  asm volatile("s_nop 15\n\ts_nop 7" : : : "memory");

  auto to_f = [](uint16_t b) { return DT::to_float(b); };
  // ---- unsplit launch: acc[s][c][r] = C[n = 16 slab + 4g + r][m = 16c + i] straight to the output ----
  if (KS == 1) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int slab = nb * NSB + wv * NS + s;
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const int m = 16 * c + i;
        if (epi == 2) {
          // rows 0..7 of a slab are gate rows 8 slab .. + 7, rows 8..15 the matching up rows: lane (g < 2) pairs with lane + 32
          f32x4 up;
#pragma unroll
          for (int r = 0; r < 4; ++r) up[r] = __shfl(acc[s][c][r], (lane + 32) & 63, 64);
          if (slab < nslab && m < M && g < 2) {
            uint16_t o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float gt = to_f(DT::from_float(acc[s][c][r])), u = to_f(DT::from_float(up[r]));
              const float sl = to_f(DT::from_float(silu_f32(gt)));
              o[r] = DT::from_float(sl * u);
            }
            *reinterpret_cast<u32x2*>(out + (size_t)m * (N >> 1) + slab * 8 + 4 * g) =
                u32x2{(u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16)};
          }
        } else if (slab < nslab && m < M) {
          const int nn = slab * 16 + 4 * g;
          if (f32out) {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(out) + (size_t)m * N + nn) = acc[s][c];
          } else {
            uint16_t o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              o[r] = DT::from_float(acc[s][c][r]);
              if (bias != nullptr) o[r] = DT::from_float(to_f(o[r]) + to_f(bias[nn + r]));  // `out + self.bias` in T (qmodule.py:221)
            }
            *reinterpret_cast<u32x2*>(out + (size_t)m * N + nn) = u32x2{(u32)o[0] | ((u32)o[1] << 16), (u32)o[2] | ((u32)o[3] << 16)};
          }
        }
      }
    }
#ifdef AWQ_ENABLE_PROBES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    MIDM_STAMP(3)
    return;
  }

  // ---- K split across blocks: fp32 sums of this part, write-through (the reducer may sit on another XCD) ----
  {
    float* mine = parts + (size_t)p * M * N;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int slab = nb * NSB + wv * NS + s;
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        const int m = 16 * c + i;
        if (slab < nslab && m < M) {
          float* dst = mine + (size_t)m * N + slab * 16 + 4 * g;
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(acc[s][c]) : "memory");  // (s_nop: as in awq_gemm_v6.hip -- a > 64-bit store the compiler cannot see; its data registers must survive two more issue slots)
        }
      }
    }
  }
  midm_reduce_parts<DT, 64 * WAVES, NSB>(smem, parts, tickets, bias, out, M, N, KS, nb, epi, f32out);
}

// =============================================================================================================================================
// host side
// =============================================================================================================================================
namespace {
// ---- ticket words of the K split.  A word belongs to ONE slab group of ONE launch at a time:
//   * eager launches take the lane of their STREAM (launches of a stream are ordered; a lane is zero outside a launch that uses it);
//   * a launch recorded during a stream capture gets words of its own from a bump region that is never handed out twice: two graphs -- or a graph and
//     the eager work of the stream it was captured on -- never share a word, whatever streams they are replayed on (a graph never runs concurrently
//     with itself).  When the region is exhausted, or no lane is free, the launch runs unsplit.
constexpr int kStreamLanes = 64, kLaneWords = 1024;   // 64 streams x 1024 slab groups
constexpr int kCaptureWords = 1 << 20;                // 4 MiB of words for captured launches
struct TicketPool {
  u32* base = nullptr;  // [kStreamLanes * kLaneWords | kCaptureWords]
  hipStream_t lane_stream[kStreamLanes] = {};
  bool lane_used[kStreamLanes] = {};
  int capture_next = 0;
};
std::mutex g_ticket_mu;
TicketPool g_pools[64];
}  // namespace

// (shared with the skinny kernel's K split, awq_skinny_cdna4.hip)

unsigned* splitk_ticket_words(hipStream_t st, int groups) {
  int dev = 0;
  if (groups > kLaneWords || hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (st != nullptr && hipStreamIsCapturing(st, &cs) != hipSuccess) return nullptr;  // (the null stream cannot be captured; the query errors on it while another stream is)
  const bool capturing = cs != hipStreamCaptureStatusNone;
  std::lock_guard<std::mutex> lock(g_ticket_mu);
  TicketPool& tp = g_pools[dev];
  if (tp.base == nullptr) {
    if (capturing) return nullptr;  // no allocation inside a capture: that launch runs unsplit (awq_midm_init avoids it)
    u32* fresh = nullptr;
    const size_t bytes = ((size_t)kStreamLanes * kLaneWords + kCaptureWords) * sizeof(u32);
    if (hipMalloc(reinterpret_cast<void**>(&fresh), bytes) != hipSuccess) return nullptr;
    // (the fill is ordered on the null stream and may return before it has run: wait for it, so that launches on ANY stream after this point see zeros)
    if (hipMemset(fresh, 0, bytes) != hipSuccess || hipStreamSynchronize(nullptr) != hipSuccess) {
      (void)hipFree(fresh);
      return nullptr;
    }
    tp.base = fresh;
  }
  if (capturing) {
    if (tp.capture_next + groups > kCaptureWords) return nullptr;
    u32* w = tp.base + (size_t)kStreamLanes * kLaneWords + tp.capture_next;
    tp.capture_next += groups;
    return w;
  }
  int free_lane = -1;
  for (int l = 0; l < kStreamLanes; ++l) {
    if (tp.lane_used[l] && tp.lane_stream[l] == st) return tp.base + (size_t)l * kLaneWords;
    if (!tp.lane_used[l] && free_lane < 0) free_lane = l;
  }
  if (free_lane < 0) return nullptr;  // more than 64 distinct streams have split launches: the rest run unsplit
  tp.lane_used[free_lane] = true;
  tp.lane_stream[free_lane] = st;
  return tp.base + (size_t)free_lane * kLaneWords;
}

namespace {

int g_midm = 1;                                     // knob midm: 0 = off (the skinny / masked-tile kernels of rounds 1-5 serve 9 .. 255 rows)
int g_midm_min = 65, g_midm_max = 128;              // knobs midm_min / midm_max: the row counts the forward entries hand to this kernel (tests: 9 .. 255); the defaults = the by-shape rules of midm_takes
int g_midm_waves = 0, g_midm_ns = 0, g_midm_ks = 0;  // knobs midm_waves / midm_ns / midm_ks: force a block shape / part count (experiments, tests)
int g_midm_probe = 0;                               // knob midm_probe: timing probes of AWQ_PROBES builds (see the kernel)
}  // namespace

int midm_tune_set(const char* key, int value) {
  if (!strcmp(key, "midm")) g_midm = value;
  else if (!strcmp(key, "midm_min")) g_midm_min = value;
  else if (!strcmp(key, "midm_max")) g_midm_max = value;
  else if (!strcmp(key, "midm_waves")) g_midm_waves = value;
  else if (!strcmp(key, "midm_ns")) g_midm_ns = value;
  else if (!strcmp(key, "midm_ks")) g_midm_ks = value;
  else if (!strcmp(key, "midm_probe")) g_midm_probe = value;
  else return -1;
  return 0;
}

namespace {
struct MidmCfg {
  int waves, ns, cb, ks;
};
// LDS of a block: DX x stages of 16 cb rows (padded to whole rounds of pieces), per wave a ring of DX + 1 slots of ns (tile + scale piece)
constexpr int midm_xstage(int cb, int waves) { return (4 * cb + waves - 1) / waves * waves * 1024; }
constexpr int midm_lds(int cb, int waves, int ns, int dx) { return dx * midm_xstage(cb, waves) + waves * ns * (dx + 1) * 1280; }
// ring depth: the DX - 1 groups in flight are what the CU has outstanding -- as deep as ~150 KiB of LDS allows, 3 .. 8 stages
constexpr int midm_dx(int cb, int waves, int ns) {
  int dx = 8;
  while (dx > 3 && midm_lds(cb, waves, ns, dx) > 150 * 1024) --dx;
  return dx;
}
int ks_for(int groups, int nit, int cus) {
  int ks = 1;
  while (groups * ks * 2 <= cus && nit / (ks * 2) >= 2) ks *= 2;
  return ks;
}
// Block shape and K parts for a pass of m <= 128 rows (profiles/r06_midm_sweep.txt).  One slab per wave; eight waves (two per SIMD) unless the K split
// that fills the chip would then leave a block fewer than eight k-steps (o_proj: 4096 x 4096) -- four waves halve the slab group instead; ks fills the
// chip with blocks, every part keeps at least two k-steps.
bool midm_pick(int m, int n, int k, bool may_split, MidmCfg& c) {
  const int nslab = n / 16, nit = k / 128, cus = device_cu_count();
  c.cb = (m + 15) / 16;
  c.ns = g_midm_ns ? g_midm_ns : 1;
  if (g_midm_waves) c.waves = g_midm_waves;
  else {
    const int g8 = (nslab + 8 * c.ns - 1) / (8 * c.ns);
    c.waves = nit / ks_for(g8, nit, cus) >= 8 ? 8 : 4;
  }
  const int groups = (nslab + c.waves * c.ns - 1) / (c.waves * c.ns);
  int ks = g_midm_ks > 0 ? g_midm_ks : ks_for(groups, nit, cus);
  if (ks > nit / 2) ks = nit / 2 > 0 ? nit / 2 : 1;
  if (!may_split || ks < 1) ks = 1;
  c.ks = ks;
  return c.cb >= 1 && c.cb <= 8 && ((c.waves == 8 && c.ns == 1) || (c.waves == 4 && (c.ns == 1 || c.ns == 2))) &&
         midm_lds(c.cb, c.waves, c.ns, midm_dx(c.cb, c.waves, c.ns)) <= 160 * 1024;
}
// the rows of a call are served in passes of at most 128 (129 .. 255: two equal passes, each re-streaming the weights)
void midm_passes(int m, int& chunks, int& rows) {
  chunks = (m + 127) / 128;
  rows = (m + chunks - 1) / chunks;
}
}  // namespace

// Which calls the forward entries hand to this kernel.  Measured against the round-5 kernels on the Llama-3-8B shapes (profiles/r06_midm_sweep.txt) and, third session,
// on the Llama-3-70B / Llama-2-7B shapes and with the true round-5 baseline above 128 rows (profiles/r06_midm_routing.txt):
//   * 65 .. 128 rows always (one pass: 0.60 - 0.94 x the time of the skinny kernel's two chunks / the masked 256-row tile on every shape measured);
//   * 129 .. 192 rows NOT any more: two passes, each re-streaming the weights, lose to the masked tile + split-K wherever K is long (down_proj 14336 -> 4096: 1.22 - 1.32 x,
//     Llama-3-70B qkv / o: 1.2 - 1.36 x) and tie on the K = 4096 projections (0.96 - 1.13 x) -- the second session's figure for down_proj came from a baseline run with a
//     scratch sized for another plan;
//   * below 65 rows the skinny kernel (x through registers, up to seven slabs per block, no barrier) is as fast or faster on the Llama-3-8B shapes, but its block shapes leave
//     other shapes under-filled -- this kernel's K split across blocks takes: 17 .. 32 rows against n >= 16384 (0.81 - 0.88 x); every row count against a very long K and a
//     narrow n (Llama-3-70B down_proj, 28672 -> 8192: 0.57 - 0.87 x at 16 .. 64 rows); 33 .. 64 rows against n = 8192-class matrices (the skinny kernel's four-slab blocks fill
//     half the chip: Llama-3-70B o_proj 0.76 x); 49 .. 64 rows against wide pairs whose slab count is not one round of seven-slab blocks (gate/up of Llama-3-70B / Llama-2-7B: 0.84 - 0.89 x).
namespace {
bool midm_small_takes(int m, int n, int k) {
  const int nslab = n / 16, nit = k / 128, cus = device_cu_count();
  const int g7 = (nslab + 6) / 7;
  const bool seven_round = g7 <= cus + 16 && g7 >= cus * 9 / 10;  // (awq_skinny_cdna4.hip: seven slabs per block = one round of one block per CU)
  if (m >= 17 && m <= 32 && n >= 16384) return true;
  if (nit >= 192 && nslab <= 512) return true;
  if (m >= 33 && nslab >= 512 && nslab < 640) return true;
  if (m >= 49 && n >= 16384 && !seven_round) return true;
  return false;
}
}  // namespace
bool midm_takes(int m, int n, int k) {
  if (!g_midm || m < 9 || m > 255 || (n % 16) != 0 || (k % 128) != 0 || k < 256) return false;
  if (g_midm_min == 65 && g_midm_max == 128) {  // the product's rules
    if (m > 128 || (m < 65 && !midm_small_takes(m, n, k))) return false;
  } else {  // knobs (tests, sweeps): a plain row range; 129 .. 192 rows against a wide n stay on the tiles as in the first two sessions
    if (m < g_midm_min || m > g_midm_max) return false;
    if (m > 128 && n >= 16384 && g_midm_max <= 192) return false;
  }
  if ((size_t)n * (size_t)k / 2 >= (1ull << 31) || (size_t)m * (size_t)k * 2 >= (1ull << 31)) return false;
  int chunks, rows;
  midm_passes(m, chunks, rows);
  MidmCfg c;
  return midm_pick(rows, n, k, true, c);
}
// K parts of a pass of m rows (1 = unsplit) and the fp32 scratch of a call: [parts][rows of a pass][n] (the passes share it: they run one after the other)
int midm_parts(int m, int n, int k) {
  MidmCfg c;
  if (m < 1 || m > 128 || !midm_pick(m, n, k, true, c)) return 1;
  return c.ks;
}
size_t midm_workspace_bytes(int m, int n, int k) {
  if (!midm_takes(m, n, k)) return 0;
  int chunks, rows;
  midm_passes(m, chunks, rows);
  const int ks = midm_parts(rows, n, k);
  return ks > 1 ? (size_t)ks * rows * n * sizeof(float) : 0;
}

template <typename DT, int WAVES, int NS, int CB, int DQ>
static void launch_midm_cfg(const void* x, const void* qw, const void* szp, const void* bias, void* out, float* parts, u32* tickets, int m, int n, int k,
                            int ks, int epi, int f32out, hipStream_t st) {
#ifdef AWQ_ENABLE_PROBES
  constexpr int kProbe = 1;
#else
  constexpr int kProbe = 0;
#endif
  constexpr int DX = midm_dx(CB, WAVES, NS);
  auto kern = midm_kernel<DT, WAVES, NS, CB, DX, DX + 1, DQ, kProbe>;
  constexpr size_t smem = (size_t)midm_lds(CB, WAVES, NS, DX);
  static LdsOptIn optin;
  if (smem > 64 * 1024) optin.ensure(reinterpret_cast<const void*>(kern));
  const int nslab = n / 16, groups = (nslab + WAVES * NS - 1) / (WAVES * NS);
  hipLaunchKernelGGL(kern, dim3(groups, ks), dim3(64 * WAVES), smem, st, (const uint16_t*)x, (const u32*)qw, (const u32*)szp, (const uint16_t*)bias,
                     (uint16_t*)out, parts, tickets, m, n, k, epi, f32out, g_midm_probe);
}

template <typename DT, int DQ>
static int launch_midm_dt(const void* x, const void* qw, const void* szp, const void* bias, void* out, float* parts, u32* tickets, int m, int n, int k,
                          const MidmCfg& c, int epi, int f32out, hipStream_t st) {
#define AWQ_MM(W_, NS_, CB_)                                                                                       \
  if (c.waves == W_ && c.ns == NS_ && c.cb == CB_) {                                                              \
    launch_midm_cfg<DT, W_, NS_, CB_, DQ>(x, qw, szp, bias, out, parts, tickets, m, n, k, c.ks, epi, f32out, st); \
    return 0;                                                                                                     \
  }
#define AWQ_MM_CB(W_, NS_) AWQ_MM(W_, NS_, 1) AWQ_MM(W_, NS_, 2) AWQ_MM(W_, NS_, 3) AWQ_MM(W_, NS_, 4) AWQ_MM(W_, NS_, 5) AWQ_MM(W_, NS_, 6) AWQ_MM(W_, NS_, 7) AWQ_MM(W_, NS_, 8)
  AWQ_MM_CB(8, 1)
  AWQ_MM_CB(4, 1)
  AWQ_MM_CB(4, 2)  // (two slabs per wave: never the plan's choice on the measured shapes -- kept compiled for the knob and the tests of NS > 1)
#undef AWQ_MM_CB
#undef AWQ_MM
  return -1;
}

// one pass of m <= 128 rows
static int launch_midm_pass(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int epi, int dtype, int szfmt,
                            int f32out, void* ws, size_t ws_bytes, hipStream_t st) {
  MidmCfg c;
  const bool ws_ok = ws != nullptr && (reinterpret_cast<uintptr_t>(ws) & 15) == 0;
  if (!midm_pick(m, n, k, ws_ok, c)) return -1;
  u32* tk = nullptr;
  if (c.ks > 1) {
    const int groups = (n / 16 + c.waves * c.ns - 1) / (c.waves * c.ns);
    if (ws_bytes < (size_t)c.ks * m * n * sizeof(float) || (tk = splitk_ticket_words(st, groups)) == nullptr) c.ks = 1;
  }
  float* parts = c.ks > 1 ? static_cast<float*>(ws) : nullptr;
  if (dtype == 0) return szfmt == 1 ? launch_midm_dt<F16, 1>(x, qw, szp, bias, out, parts, tk, m, n, k, c, epi, f32out, st)
                                    : launch_midm_dt<F16, 0>(x, qw, szp, bias, out, parts, tk, m, n, k, c, epi, f32out, st);
  return szfmt == 1 ? launch_midm_dt<BF16, 1>(x, qw, szp, bias, out, parts, tk, m, n, k, c, epi, f32out, st)
                    : launch_midm_dt<BF16, 0>(x, qw, szp, bias, out, parts, tk, m, n, k, c, epi, f32out, st);
}

// The forward entries' call: the row counts of midm_takes, cdna4 layout.  szfmt 0: szp = sz_packed, 1: sz_half.  epi 0 (bias fused; f32out: float [m, n]
// unrounded, no bias) or 2 (out [m, n / 2] = silu(gate) * up).  ws: optional fp32 scratch of the K split (midm_workspace_bytes).  -1 if not served.
int launch_midm_cdna4(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int epi, int dtype, int szfmt,
                      int bits, int f32out, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!szp || bits != 4 || !midm_takes(m, n, k) || (epi != 0 && epi != 2) || (epi == 2 && ((n % 32) != 0 || bias || f32out)) || (f32out && bias)) return -1;
  int chunks, rows;
  midm_passes(m, chunks, rows);
  const size_t ocols = epi == 2 ? (size_t)n / 2 : (size_t)n;
  for (int r0 = 0; r0 < m; r0 += rows) {
    const int mr = m - r0 < rows ? m - r0 : rows;
    const int rc = launch_midm_pass(static_cast<const uint16_t*>(x) + (size_t)r0 * k, qw, szp, bias, static_cast<char*>(out) + (size_t)r0 * ocols * (f32out ? 4 : 2),
                                    mr, n, k, epi, dtype, szfmt, f32out, ws, ws_bytes, st);
    if (rc != 0) return rc;
  }
  return 0;
}

// allocate the ticket words of the current device outside any capture / forward call (optional: the first split launch does it otherwise)
int midm_init() { return splitk_ticket_words(nullptr, 1) != nullptr ? 0 : -1; }

}  // namespace awq
