// One-shot all-reduce (sum) for latency-class messages over peer-mapped exchange buffers (gfx950, xGMI).
//
// Why (DESIGN.md "Multi-GPU"): K-sharded decode ends every o_proj / down_proj with an all-reduce of M * N * 2
// bytes = 8 .. 16 KiB per token.  A ring pays 2 (W - 1) link hops for such a message; on MI355X's fully connected xGMI mesh every
// rank can instead WRITE its partial into every peer directly (W - 1 stores per 16 bytes, all links in parallel), raise one flag
// per peer, and reduce locally once the W flags of this round have arrived -- one hop of latency, no ordering between ranks.
// The reference has no multi-GPU path; the numerics are the single-device oracle's: fp32 sum in RANK ORDER
// (deterministic, identical on every rank), one rounding to T.
//
// Exchange buffer of a rank (allocated fine-grained, exported to the peers with hipIpc by the Python side, llm_awq_amd/oneshot.py):
//   data  [2 halves][W slots][max_bytes]   half = round & 1, slot q holds rank q's partial of this round
//   flags [2 halves][W]  u32                flag (half, q) == round  <=>  slot q of that half is complete
// Round e on rank r:  (1) store `in` into slot r of half e & 1 of EVERY rank's buffer (its own included);  (2) system-scope
// release, then store e into flag (e & 1, r) of every rank;  (3) wait until the W local flags of half e & 1 equal e (bounded
// spin: `status` receives 1 on a timeout instead of hanging the queue);  (4) out = T(sum_q fp32(slot q)).
// fp32 form (awq_oneshot_allreduce_f32, IN32): the message is the K shard's UNROUNDED fp32 partial (awq_w4a16_partial_cdna4), count * 4 bytes;
// out = T(sum_q slot q) (+ bias in T): the partials are rounded ONCE, after the sum, like the single-device kernel rounds its own accumulator
// (T-rounded partials put a bf16 row-parallel output 2.6-2.9e-3 from the single-device result; fp32 partials 2e-6).
// Re-use: half e & 1 is written again in round e + 2, which a rank can only enter after it saw every peer's flag of round e + 1,
// and a peer raises that flag after it finished reading round e (stream order) -- no second barrier is needed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/awq_cdna4.h"
#include "awq_device.hpp"

namespace awq {

constexpr int kOneShotMaxWorld = 8;

struct OneShotPeers {
  char* data[kOneShotMaxWorld];
  u32* flags[kOneShotMaxWorld];
};

template <typename DT, bool IN32>
__global__ __launch_bounds__(1024) void oneshot_allreduce_kernel(OneShotPeers peers, const void* __restrict__ in_v, uint16_t* __restrict__ out_,
                                                                  const uint16_t* __restrict__ bias, int bias_n, int count, int rank0, int world,
                                                                  u32 round, int max_bytes, u32 spin_limit, int* __restrict__ status) {
  // one block = one rank.  A real call launches ONE block (rank0 = this process' rank); the single-GPU self-test launches `world`
  // co-resident blocks that play the ranks against each other (awq_oneshot_allreduce_selftest)
  constexpr int EB = IN32 ? 4 : 2;   // bytes per element of the exchanged message
  const int rank = rank0 + (int)blockIdx.x;
  const char* in = static_cast<const char*>(in_v) + (size_t)blockIdx.x * count * EB;
  uint16_t* out = out_ + (size_t)blockIdx.x * count;
  const int out_chunks = (count * 2 + 15) / 16;
  auto poison = [&]() {
    // never hand back a sum of stale slots: the output is poisoned with NaNs (0x7FFF is a NaN in fp16 and in bf16) and the sticky
    // status word is set -- OneShotAllReduce.check() raises, and a caller that does not check sees NaNs, not a plausible number
    for (int c = threadIdx.x; c < out_chunks; c += blockDim.x)
      *reinterpret_cast<u32x4*>(out + (size_t)c * 8) = u32x4{0x7FFF7FFFu, 0x7FFF7FFFu, 0x7FFF7FFFu, 0x7FFF7FFFu};
  };
  // a communicator that lost a round is dead until it is rebuilt: its epoch was not advanced, so late flags of the lost round would match
  // the next call's round and stale slots would be summed into a plausible number -- every later call poisons its output instead
  if (status != nullptr && *reinterpret_cast<volatile int*>(status) != 0) {
    poison();
    return;
  }
  // round == 0: the epoch lives in this rank's own buffer (u32 [32] of the flag area) and advances by one per call -- every rank issues
  // the same sequence of calls, so the counters agree, and a REPLAYED hipGraph (whose kernel arguments are frozen) still sees a new
  // round every time.  One mode per communicator: explicit rounds and device epochs must not be mixed on the same buffers.
  u32* epoch = peers.flags[rank] + 32;
  const bool dev_epoch = round == 0u;
  if (dev_epoch) round = *reinterpret_cast<volatile u32*>(epoch) + 1u;  // (written by this rank's previous call only: stream order)
  const int half = (int)(round & 1u);
  __shared__ int timed_out;
  if (threadIdx.x == 0) timed_out = 0;
  const int chunks = (count * EB + 15) / 16;  // 16-byte chunks of the message (count % 8 == 0)
  const size_t slot_off = ((size_t)half * world + rank) * (size_t)max_bytes;
  // (1) my partial -> slot `rank` of every rank's buffer (write-through system-scope stores: they must leave this GPU's L2)
  for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(in + (size_t)c * 16);
    for (int p = 0; p < world; ++p) {
      u32* dst = reinterpret_cast<u32*>(peers.data[p] + slot_off + (size_t)c * 16);
      __hip_atomic_store(dst + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(dst + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(dst + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(dst + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // (2) all of this block's stores are visible system-wide before any flag is
  __threadfence_system();
  __syncthreads();
  if ((int)threadIdx.x < world) {
    __hip_atomic_store(peers.flags[threadIdx.x] + half * world + rank, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // (3) wait for rank threadIdx.x's flag in MY buffer
    const u32* f = peers.flags[rank] + half * world + threadIdx.x;
    u32 spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != round) {
      if (++spins > spin_limit) {
        if (status) atomicExch(status, 1);
        timed_out = 1;  // (shared) a slot of this round never arrived: the sum below would mix rounds
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  if (timed_out) {
    poison();
    return;  // (the device epoch is NOT advanced: the communicator is dead until it is rebuilt -- see the entry check)
  }
  // (4) fixed-order fp32 sum of the W slots of my buffer (system-scope loads: the bytes were written by other GPUs)
  const char* mine = peers.data[rank] + (size_t)half * world * (size_t)max_bytes;
  for (int c = threadIdx.x; c < out_chunks; c += blockDim.x) {  // one 16-byte chunk of the OUTPUT (8 elements) per thread and step
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < world; ++q) {
      if constexpr (IN32) {
        const u32* src = reinterpret_cast<const u32*>(mine + (size_t)q * max_bytes + (size_t)c * 32);
        u32 w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += __uint_as_float(w[e]);
      } else {
        const u32* src = reinterpret_cast<const u32*>(mine + (size_t)q * max_bytes + (size_t)c * 16);
        u32 w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[2 * e] += DT::to_float((uint16_t)(w[e] & 0xFFFFu));
          acc[2 * e + 1] += DT::to_float((uint16_t)(w[e] >> 16));
        }
      }
    }
    uint16_t o16[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o16[e] = DT::from_float(acc[e]);
    if (bias != nullptr) {  // `out + self.bias` in T, once, after the sum (qmodule.py:221)
      const u32x4 bv = *reinterpret_cast<const u32x4*>(bias + ((size_t)c * 8) % (size_t)bias_n);
      const u32 bw[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) o16[e] = DT::from_float(DT::to_float(o16[e]) + DT::to_float((uint16_t)((bw[e >> 1] >> (16 * (e & 1))) & 0xFFFFu)));
    }
    u32x4 o;
    o.x = (u32)o16[0] | ((u32)o16[1] << 16);
    o.y = (u32)o16[2] | ((u32)o16[3] << 16);
    o.z = (u32)o16[4] | ((u32)o16[5] << 16);
    o.w = (u32)o16[6] | ((u32)o16[7] << 16);
    *reinterpret_cast<u32x4*>(out + (size_t)c * 8) = o;
  }
  if (dev_epoch) {
    __syncthreads();  // every thread has read the old value
    if (threadIdx.x == 0) *reinterpret_cast<volatile u32*>(epoch) = round;
  }
}

}  // namespace awq

extern "C" {

size_t awq_oneshot_buffer_bytes(int world, int max_bytes) {
  if (world < 1 || world > awq::kOneShotMaxWorld || max_bytes <= 0 || (max_bytes % 16) != 0) return 0;
  return (size_t)2 * world * max_bytes + 256;  // data, then the flags in the last 256 bytes (2 * 8 u32 = 64 B used)
}

int awq_oneshot_alloc(void** buffer, int world, int max_bytes) {
  if (!buffer) return AWQ_ERR_NULL;
  const size_t bytes = awq_oneshot_buffer_bytes(world, max_bytes);
  if (!bytes) return AWQ_ERR_SHAPE;
  // fine-grained (uncached across agents) device memory: peers' stores and this GPU's polling loads meet in memory, not in an L2
  // (NO fallback to plain hipMalloc: in coarse-grained memory a peer's xGMI stores need not become visible to a kernel that is
  // already polling, and the round would time out -- the caller falls back to RCCL instead)
  if (hipExtMallocWithFlags(buffer, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    *buffer = nullptr;
    return AWQ_ERR_LAUNCH;
  }
  if (hipMemset(*buffer, 0, bytes) != hipSuccess) return AWQ_ERR_LAUNCH;
  return hipDeviceSynchronize() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int awq_oneshot_free(void* buffer) { return hipFree(buffer) == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH; }

int awq_oneshot_ipc_export(void* buffer, void* handle64) {
  if (!buffer || !handle64) return AWQ_ERR_NULL;
  static_assert(sizeof(hipIpcMemHandle_t) <= 64, "handle does not fit the 64-byte slot");
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, buffer) != hipSuccess) return AWQ_ERR_LAUNCH;
  memcpy(handle64, &h, sizeof(h));
  return AWQ_OK;
}

int awq_oneshot_ipc_open(const void* handle64, void** buffer) {
  if (!buffer || !handle64) return AWQ_ERR_NULL;
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof(h));
  return hipIpcOpenMemHandle(buffer, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int awq_oneshot_ipc_close(void* buffer) { return hipIpcCloseMemHandle(buffer) == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH; }

namespace {
// spin bound of the flag wait.  One poll is a system-scope load + s_sleep: ~1 us, so the default lets a peer be ~40 s late (a lazy code-object
// load, a host sync, GC on another rank are benign skews and must not kill the communicator; a dead peer still ends the wait instead of hanging
// the queue).  awq_oneshot_set_spin_limit / AWQ_ONESHOT_SPIN_LIMIT (read by llm_awq_amd/oneshot.py) change it; tests use a small one.
unsigned g_spin_limit = 40000000u;
}  // namespace

int awq_oneshot_set_spin_limit(unsigned spins) {
  if (spins == 0) return AWQ_ERR_SHAPE;
  g_spin_limit = spins;
  return AWQ_OK;
}

static int oneshot_launch(void* const* peer_buffers, const void* in, const void* bias, int bias_n, void* out, int count, int dtype, int in32, int rank,
                          int world, unsigned round, int max_bytes, int* status_dev, void* stream, int blocks) {
  if (!peer_buffers || !in || !out) return AWQ_ERR_NULL;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (world < 1 || world > awq::kOneShotMaxWorld || rank < 0 || rank >= world || count <= 0 || (count % 8) != 0 ||
      (long long)count * (in32 ? 4 : 2) > (long long)max_bytes || (max_bytes % 16) != 0)
    return AWQ_ERR_SHAPE;  // (round == 0 selects the device-resident epoch)
  if (bias && (bias_n <= 0 || (bias_n % 8) != 0 || (count % bias_n) != 0)) return AWQ_ERR_SHAPE;
  if ((reinterpret_cast<uintptr_t>(in) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u) || (reinterpret_cast<uintptr_t>(bias) & 15u)) return AWQ_ERR_ALIGN;
  awq::OneShotPeers peers;
  for (int p = 0; p < awq::kOneShotMaxWorld; ++p) {
    char* b = (char*)peer_buffers[p < world ? p : 0];
    if (!b) return AWQ_ERR_NULL;
    peers.data[p] = b;
    peers.flags[p] = reinterpret_cast<awq::u32*>(b + (size_t)2 * world * max_bytes);
  }
  const int chunks = count / 8;
  const int threads = chunks >= 1024 ? 1024 : (chunks < 64 ? 64 : ((chunks + 63) / 64) * 64);
  using Kern = void (*)(awq::OneShotPeers, const void*, uint16_t*, const uint16_t*, int, int, int, int, awq::u32, int, awq::u32, int*);
  static const Kern kerns[2][2] = {{awq::oneshot_allreduce_kernel<awq::F16, false>, awq::oneshot_allreduce_kernel<awq::F16, true>},
                                   {awq::oneshot_allreduce_kernel<awq::BF16, false>, awq::oneshot_allreduce_kernel<awq::BF16, true>}};
  hipLaunchKernelGGL(kerns[dtype == AWQ_F16 ? 0 : 1][in32 ? 1 : 0], dim3(blocks), dim3(threads), 0, (hipStream_t)stream, peers, in, (uint16_t*)out,
                     (const uint16_t*)bias, bias_n, count, rank, world, (awq::u32)round, max_bytes, g_spin_limit, status_dev);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

int awq_oneshot_allreduce(void* const* peer_buffers, const void* in, void* out, int count, int dtype, int rank, int world,
                          unsigned round, int max_bytes, int* status_dev, void* stream) {
  return oneshot_launch(peer_buffers, in, nullptr, 0, out, count, dtype, 0, rank, world, round, max_bytes, status_dev, stream, 1);
}

int awq_oneshot_allreduce_f32(void* const* peer_buffers, const float* in_f32, const void* bias, int bias_n, void* out, int count, int dtype, int rank,
                              int world, unsigned round, int max_bytes, int* status_dev, void* stream) {
  return oneshot_launch(peer_buffers, in_f32, bias, bias_n, out, count, dtype, 1, rank, world, round, max_bytes, status_dev, stream, 1);
}

int awq_oneshot_allreduce_selftest(void* const* peer_buffers, const void* in_all, void* out_all, int count, int dtype, int world,
                                   unsigned round, int max_bytes, int* status_dev, void* stream) {
  return oneshot_launch(peer_buffers, in_all, nullptr, 0, out_all, count, dtype, 0, 0, world, round, max_bytes, status_dev, stream, world);
}

int awq_oneshot_allreduce_f32_selftest(void* const* peer_buffers, const float* in_all_f32, const void* bias, int bias_n, void* out_all, int count,
                                       int dtype, int world, unsigned round, int max_bytes, int* status_dev, void* stream) {
  return oneshot_launch(peer_buffers, in_all_f32, bias, bias_n, out_all, count, dtype, 1, 0, world, round, max_bytes, status_dev, stream, world);
}

}  // extern "C"
