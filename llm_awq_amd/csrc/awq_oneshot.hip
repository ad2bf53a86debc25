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

template <typename DT>
__global__ __launch_bounds__(1024) void oneshot_allreduce_kernel(OneShotPeers peers, const uint16_t* __restrict__ in_, uint16_t* __restrict__ out_,
                                                                  int count, int rank0, int world, u32 round, int max_bytes, u32 spin_limit,
                                                                  int* __restrict__ status) {
  // one block = one rank.  A real call launches ONE block (rank0 = this process' rank); the single-GPU self-test launches `world`
  // co-resident blocks that play the ranks against each other (awq_oneshot_allreduce_selftest)
  const int rank = rank0 + (int)blockIdx.x;
  const uint16_t* in = in_ + (size_t)blockIdx.x * count;
  uint16_t* out = out_ + (size_t)blockIdx.x * count;
  // round == 0: the epoch lives in this rank's own buffer (u32 [32] of the flag area) and advances by one per call -- every rank issues
  // the same sequence of calls, so the counters agree, and a REPLAYED hipGraph (whose kernel arguments are frozen) still sees a new
  // round every time.  One mode per communicator: explicit rounds and device epochs must not be mixed on the same buffers.
  u32* epoch = peers.flags[rank] + 32;
  const bool dev_epoch = round == 0u;
  if (dev_epoch) round = *reinterpret_cast<volatile u32*>(epoch) + 1u;  // (written by this rank's previous call only: stream order)
  const int half = (int)(round & 1u);
  __shared__ int timed_out;
  if (threadIdx.x == 0) timed_out = 0;
  const int chunks = (count * 2 + 15) / 16;  // 16-byte chunks of the message (count % 8 == 0)
  const size_t slot_off = ((size_t)half * world + rank) * (size_t)max_bytes;
  // (1) my partial -> slot `rank` of every rank's buffer (write-through system-scope stores: they must leave this GPU's L2)
  for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(in + (size_t)c * 8);
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
    // never hand back a sum of stale slots: the output is poisoned with NaNs (0x7FFF is a NaN in fp16 and in bf16) and the sticky
    // status word is set -- OneShotAllReduce.check() raises, and a caller that does not check sees NaNs, not a plausible number
    for (int c = threadIdx.x; c < chunks; c += blockDim.x)
      *reinterpret_cast<u32x4*>(out + (size_t)c * 8) = u32x4{0x7FFF7FFFu, 0x7FFF7FFFu, 0x7FFF7FFFu, 0x7FFF7FFFu};
    return;  // (the device epoch is NOT advanced: the communicator is dead until it is rebuilt)
  }
  // (4) fixed-order fp32 sum of the W slots of my buffer (system-scope loads: the bytes were written by other GPUs)
  const char* mine = peers.data[rank] + (size_t)half * world * (size_t)max_bytes;
  for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int q = 0; q < world; ++q) {
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
    u32x4 o;
    o.x = (u32)DT::from_float(acc[0]) | ((u32)DT::from_float(acc[1]) << 16);
    o.y = (u32)DT::from_float(acc[2]) | ((u32)DT::from_float(acc[3]) << 16);
    o.z = (u32)DT::from_float(acc[4]) | ((u32)DT::from_float(acc[5]) << 16);
    o.w = (u32)DT::from_float(acc[6]) | ((u32)DT::from_float(acc[7]) << 16);
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

static int oneshot_launch(void* const* peer_buffers, const void* in, void* out, int count, int dtype, int rank, int world, unsigned round,
                          int max_bytes, int* status_dev, void* stream, int blocks);

int awq_oneshot_allreduce(void* const* peer_buffers, const void* in, void* out, int count, int dtype, int rank, int world,
                          unsigned round, int max_bytes, int* status_dev, void* stream) {
  return oneshot_launch(peer_buffers, in, out, count, dtype, rank, world, round, max_bytes, status_dev, stream, 1);
}

int awq_oneshot_allreduce_selftest(void* const* peer_buffers, const void* in_all, void* out_all, int count, int dtype, int world,
                                   unsigned round, int max_bytes, int* status_dev, void* stream) {
  return oneshot_launch(peer_buffers, in_all, out_all, count, dtype, 0, world, round, max_bytes, status_dev, stream, world);
}

static int oneshot_launch(void* const* peer_buffers, const void* in, void* out, int count, int dtype, int rank, int world, unsigned round,
                          int max_bytes, int* status_dev, void* stream, int blocks) {
  if (!peer_buffers || !in || !out) return AWQ_ERR_NULL;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (world < 1 || world > awq::kOneShotMaxWorld || rank < 0 || rank >= world || count <= 0 || (count % 8) != 0 || count * 2 > max_bytes ||
      (max_bytes % 16) != 0)
    return AWQ_ERR_SHAPE;  // (round == 0 selects the device-resident epoch)
  if ((reinterpret_cast<uintptr_t>(in) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u)) return AWQ_ERR_ALIGN;
  awq::OneShotPeers peers;
  for (int p = 0; p < awq::kOneShotMaxWorld; ++p) {
    char* b = (char*)peer_buffers[p < world ? p : 0];
    if (!b) return AWQ_ERR_NULL;
    peers.data[p] = b;
    peers.flags[p] = reinterpret_cast<awq::u32*>(b + (size_t)2 * world * max_bytes);
  }
  const int chunks = count / 8;
  const int threads = chunks >= 1024 ? 1024 : (chunks < 64 ? 64 : ((chunks + 63) / 64) * 64);
  auto kern = dtype == AWQ_F16 ? awq::oneshot_allreduce_kernel<awq::F16> : awq::oneshot_allreduce_kernel<awq::BF16>;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, peers, (const uint16_t*)in, (uint16_t*)out, count, rank, world,
                     (awq::u32)round, max_bytes, 4000000u, status_dev);
  return hipGetLastError() == hipSuccess ? AWQ_OK : AWQ_ERR_LAUNCH;
}

}  // extern "C"
