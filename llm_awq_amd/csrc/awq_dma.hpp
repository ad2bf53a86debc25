// LDS-DMA (`buffer_load ... lds`) helpers shared by the streaming decode kernels (gfx950).
#pragma once
#include <type_traits>
#include <utility>

#include "awq_device.hpp"

namespace awq {

#define DMA_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// `buffer_load_dword(x4) ... lds`: SIZE bytes per lane from rsrc[voff + soff] to LDS at (wave-uniform) dst + lane * SIZE; AUX 2 = nt.
// (The builtin only exists in the device pass; un-guarded, the host pass silently drops the kernel's launch stub.)
template <int SIZE, int AUX>
__device__ __forceinline__ void dma_to_lds(const __amdgpu_buffer_rsrc_t& rsrc, char* dst, u32 voff, u32 soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (SIZE == 16 && AUX == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, DMA_LDS_PTR(dst), 16, voff, soff, 0, 2);
  if constexpr (SIZE == 16 && AUX == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, DMA_LDS_PTR(dst), 16, voff, soff, 0, 0);
  if constexpr (SIZE == 4 && AUX == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, DMA_LDS_PTR(dst), 4, voff, soff, 0, 0);
  if constexpr (SIZE == 16 && AUX == 17) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, DMA_LDS_PTR(dst), 16, voff, soff, 0, 17);  // sc0 sc1
#endif
}
// the same with the LDS destination given as its 32-bit address (the M0 value): kernels that keep their LDS bookkeeping in integers avoid a
// generic -> local pointer cast (and its null check) per DMA
template <int SIZE, int AUX>
__device__ __forceinline__ void dma_to_lds_at(const __amdgpu_buffer_rsrc_t& rsrc, u32 lds_addr, u32 voff, u32 soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(size_t)lds_addr;
  if constexpr (SIZE == 16 && AUX == 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, 2);
  if constexpr (SIZE == 16 && AUX == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 16, voff, soff, 0, 0);
  if constexpr (SIZE == 4 && AUX == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst, 4, voff, soff, 0, 0);
#endif
}
template <int N_>
__device__ __forceinline__ void dma_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N_) : "memory");
}
template <int J, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (J < E) {
    f(std::integral_constant<int, J>{});
    static_for<J + 1, E>(f);
  }
}

}  // namespace awq
