// Internal launcher declarations (host side).  The public surface is include/awq_cdna4.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

namespace awq {
// Kernels that use more than 64 KiB of dynamic LDS must be opted in with hipFuncSetAttribute, and the attribute belongs to the
// (kernel, DEVICE) pair: a single-process multi-GPU caller (the reference's accelerate layer placement, awq/entry.py:167-186)
// launches the same kernel on several devices.  One LdsOptIn per kernel instantiation remembers, per device ordinal, that the
// opt-in has been made; racing threads may both make the (idempotent) call.
struct LdsOptIn {
  std::atomic<uint64_t> done[4] = {};  // device ordinals 0..255
  void ensure(const void* kern, int bytes = 160 * 1024) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    std::atomic<uint64_t>& w = done[(dev >> 6) & 3];
    if (w.load(std::memory_order_acquire) & bit) return;
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    w.fetch_or(bit, std::memory_order_release);
  }
};
// Compute units of the current device (cached per ordinal).  The plans that assume ONE residency round on the whole MI355X (the block-pair
// K split: 256 blocks, pair partners on one XCD) ask it: a partitioned (CPX / NPS), CU-masked or other part falls back to plans that do
// not depend on co-residency.  Without a device (host-side plan queries in CPU tests) the answer is the MI355X's 256.
inline int device_cu_count() {
  static std::atomic<int> cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  std::atomic<int>& c = cached[dev & 63];
  int v = c.load(std::memory_order_relaxed);
  if (v == 0) {
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    c.store(v, std::memory_order_relaxed);
  }
  return v;
}
// layout: 0 = reference v2 interleave, 1 = cdna4 interleave (bf16 only)
// szp: optional packed {scale | scaled_zero << 16} u32 [N/16][K/128][16] (cdna4 layout only), else nullptr
int launch_gemv(const void* x, const void* qw, const void* s, const void* z, const void* szp, void* out, int m, int n,
                int k, int dtype, int layout, hipStream_t st);
// fast decode path on cdna4 buffers (1 <= m <= 8, bf16 / fp16 (W3: bf16), packed sz required).  epi 0: out[m,n] (+bias);
// epi 1: qw = [gate; up] stacked (n = 2*ffn rows), out[m, n/2] = silu(gate) * up.  Returns -1 if unsupported.
int launch_gemv_cdna4(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                      int epi, int bits, int dtype, hipStream_t st);
// RMSNorm fused in front of the decode GEMV (awq_gemv_cdna4.hip, NORM = 1): 1 <= m <= 4.  Returns -1 if unsupported.
int launch_gemv_cdna4_norm(const void* x, const void* gamma, float eps, const void* qw, const void* szp, const void* bias, void* out,
                           int m, int n, int k, int epi, int dtype, hipStream_t st);
// reference (v2) layout fast decode path (awq_gemv_v2fast.hip): 1 <= m <= 8, n % 16 == 0, fp16 / bf16, bias fused.
// gpad = rows of scales / zeros that may be read (>= k / 128).  Returns -1 if unsupported.
bool gemv_v2fast_enabled();  // false while a knob of the older kernel (awq_tune_set gemv_*) is active
int launch_gemv_v2fast(const void* x, const void* qw, const void* s, const void* z, const void* bias, void* out, int m, int n, int k,
                       int gpad, int dtype, hipStream_t st);
// reference (v2) layout skinny GEMM (awq_skinny_v2.hip): 9 <= m <= 255, n % 16 == 0, fp16 / bf16, bias fused.  -1 if unsupported.
int launch_skinny_v2(const void* x, const void* qw, const void* s, const void* z, const void* bias, void* out, int m, int n, int k,
                     int gpad, int dtype, hipStream_t st);
// W3 ("w3c") format helpers (awq_w3.hip)
int launch_pack_w3(const void* q_u8, void* qw3, int n, int k, hipStream_t st);
int launch_unpack_w3(const void* qw3, void* out_u8, int n, int k, hipStream_t st);
int launch_dequant_w3(const void* qw3, const void* s, const void* z, void* out, int n, int k, int dtype, hipStream_t st);
int launch_expand_w3_to_cdna4(const void* qw3, void* qw4, int n, int k, hipStream_t st);
int gemv_cdna4_tune_set(const char* key, int value);
// LDS-DMA streaming decode GEMV (awq_gemv_dma.hip): 1 <= m <= 8, cdna4 layout + packed sz.  epi 0: out[m,n] (+bias); epi 1: stacked
// [gate; up]; epi 2: gate / up rows interleaved 8 + 8 per slab; both out[m, n/2] = silu(gate) * up.  -1 if the shape is not served.
// szfmt 0: szp = sz_packed {s | sz << 16} in T;  szfmt 1: szp = sz_half (f16-mantissa dequant, awq_pack_szh_cdna4)
// f32out (epi 0, no bias): out is float [m, n], the fp32 sums unrounded -- the K-shard partial of a tensor-parallel row split
int launch_gemv_dma(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int epi,
                    int dtype, int szfmt, hipStream_t st, int f32out = 0);
int gemv_dma_tune_set(const char* key, int value);
int launch_moe_gemv_cdna4(const void* x, const void* qw, const void* szp, const void* offsets, void* out, int total_rows,
                          int experts, int n, int k, int dtype, hipStream_t st);
// tail pass of the grouped prefill GEMM (awq_skinny_cdna4.hip): the rows of every expert's last partial 256-row tile when there are fewer than `tail` of them
int launch_moe_skinny_tail_cdna4(const void* x, const void* qw, const void* szp, const void* offsets, void* out, int experts, int n, int k,
                                 int dtype, hipStream_t st, int epi, int tail);
void moe_v6_set_tail(int v);  // knob moe_tail (0 .. 64; default 64)
// grouped skinny kernel for 9..255 sorted rows (awq_skinny_cdna4.hip)
int launch_moe_skinny_cdna4(const void* x, const void* qw, const void* szp, const void* offsets, void* out, int total_rows,
                            int experts, int n, int k, int dtype, hipStream_t st);
int launch_pack_sz_cdna4(const void* s, const void* z, void* szp, int n, int k, hipStream_t st);
// "sz_half" form for the decode kernels: {f16(s') | f16(sz) << 16}; *inexact (device int, caller-zeroed) is set if a value is not exact
int launch_pack_szh_cdna4(const void* s, const void* z, void* szh, int* inexact, int n, int k, int dtype, hipStream_t st);
int launch_gemm(const void* x, const void* qw, const void* s, const void* z, const void* szp, void* out, int m, int n,
                int k, int dtype, int layout, void* ws, size_t ws_bytes, hipStream_t st);
int launch_repack_v2_cdna4(const void* src, void* dst, int n, int k, int to_cdna4, hipStream_t st);
int launch_unpack_cdna4(const void* qw, void* out_u8, int n, int k, hipStream_t st);
int launch_dequant_cdna4(const void* qw, const void* s, const void* z, void* out, int n, int k, int dtype, hipStream_t st);
size_t gemm_workspace_bytes(int m, int n, int k);
// bias may be nullptr; when given it is added in the epilogue (`out + bias` in T)
// bits 4: cdna4 W4 tiles; bits 3: w3c tiles (read natively by the v4 / v4n weight producers)
int launch_gemm_cdna4_v3(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                         int tile_n, int dtype, void* ws, size_t ws_bytes, hipStream_t st, int bits = 4, int epi = 0, const void* szh = nullptr);
// epi 2: qw holds QuantLlamaMLP's 8 + 8 row-interleaved gate / up pair, out[m, n/2] = silu(gate) * up fused into the tile epilogue
// epi 3: out is float [m, n]: the fp32 accumulators unrounded, no bias (K-shard partial of a tensor-parallel row split)
int gemm_variant_get();
// skinny GEMM, 9 <= m <= 255 (row chunks of <= 64), cdna4 layout + packed sz (awq_skinny_cdna4.hip); bias may be nullptr; -1 if unsupported
int launch_skinny_cdna4(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                        int dtype, hipStream_t st, int f32out = 0, void* ws = nullptr, size_t ws_bytes = 0);
// the skinny launch's K split across blocks (awq_skinny_cdna4.hip): K parts for a pass of m rows (1 = unsplit), its optional fp32 scratch, knob
int launch_skinny_w3(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int epi, int dtype, hipStream_t st,
                     int f32out = 0);
int launch_skinny_gate_up(const void* x, const void* qw, const void* szp, void* out, int m, int n2, int k, int dtype, hipStream_t st);
int skinny_splitk_parts(int m, int n, int k);
size_t skinny_splitk_workspace_bytes(int m, int n, int k);
int skinny_tune_set(const char* key, int value);
// mid-M GEMM (awq_midm_cdna4.hip): 9 <= m <= 255, x tile shared by the block through LDS-DMA, waves split N, K split across blocks (ticket + fp32 parts);
// szfmt 0: szp = sz_packed, 1: sz_half; epi 0 / 2; f32out: float [m, n] unrounded; ws: the optional scratch of the K split.  -1 if the shape is not served
int launch_midm_cdna4(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int epi, int dtype, int szfmt,
                      int bits, int f32out, void* ws, size_t ws_bytes, hipStream_t st);
bool midm_takes(int m, int n, int k);
size_t midm_workspace_bytes(int m, int n, int k);
int midm_parts(int m, int n, int k);
int midm_tune_set(const char* key, int value);
int midm_init();
// ticket words of an in-launch K split (zero outside a launch that uses them): `groups` words that belong to ONE launch at a time -- the lane of the stream for
// eager launches, words of their own for launches recorded during a capture; nullptr = none available (the caller runs unsplit).  awq_midm_cdna4.hip
unsigned* splitk_ticket_words(hipStream_t st, int groups);
// batched decode on the same kernel (m <= 16; szfmt 1: szp = sz_half; epi 0 / 2 as launch_gemv_dma); -1 if unsupported
int launch_skinny_decode(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int epi,
                         int dtype, int szfmt, hipStream_t st, int f32out = 0);
int launch_moe_gemm(const void* x, const void* qw, const void* s, const void* z, const void* offsets, void* out, int total_m,
                    int experts, int n, int k, int gpad, int dtype, int layout, hipStream_t st);
int gemv_tune_set(const char* key, int value);
int gemm_tune_set(const char* key, int value);
int gemm_v3_tune_set(const char* key, int value);  // tile-plan / dispatch knobs (awq_gemm_plan.hip)
// 256 x 128 tiles with the same hand-scheduled K loop (awq_gemm_v4n.hip)
void launch_gemm_cdna4_v4n(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k,
                           int n_begin, int n_end, int dtype, void* ws, size_t ws_bytes, hipStream_t st, int bits = 4, int epi = 0);
size_t gemm_v4n_workspace_bytes(int m, int n_cols, int k);
extern int g_v4n_ksplit_force;  // knob gemm_splitk > 1
extern int g_v4n_ksplit_cap;    // knob gemm_splitk_cap
bool gemm_cdna4_v3_takes(int m, int k);  // m >= 256, or a shorter prompt the 256-row tile still beats the skinny kernel on
size_t gemm_cdna4_v3_workspace_bytes(int m, int n, int k);
int gemm_cdna4_v3_plan(int m, int n, int bits, int* mode, int* cols_main);
int gemm_cdna4_v3_narrow_kernel(int m, int n_cols, int k, int bits, int has_workspace, int epi);  // 1 v6 (NS = 2), 0 v4n unsplit, >= 2 v4n split-K ranges
int gemv_dma_plan(int m, int n, int k, int epi, int* kernel);  // weight passes of the decode entry (0: not served); *kernel 0 streaming, 1 skinny
size_t gemm_cdna4_v3_workspace_bytes_w3(int m, int n, int k);  // same rule for w3c tiles (every m > 8 takes the tile kernels)
bool moe_v4_enabled();  // knob moe_v4 (default on): the grouped skinny / tile kernels; 0 = the 128 x 128 grouped kernel for every batch above 8 rows
// the same grouped GEMM on the v6 tile (awq_gemm_v6.hip: one pipelined wave per SIMD, weights in registers); total >= 256
// epi 2: per-expert w1 / w3 pair interleaved 8 + 8 per slab (n = 2 x ffn), out [total, n / 2] = silu(w1 x) * (w3 x) fused into the tile epilogue
// szh: the experts' stacked sz_half side buffers (optional): the tile launch then dequantises in the f16-mantissa form; the tail pass keeps sz_packed
int launch_moe_gemm_cdna4_v6(const void* x, const void* qw, const void* szp, const void* offsets, void* out, int total, int experts,
                             int n, int k, int dtype, hipStream_t st, int epi = 0, const void* szh = nullptr);
// out[t, 8 j + c] = T(T(silu(in[t, 16 j + c])) * in[t, 16 j + 8 + c]): the SiLU * mul tail on an [m, n2] result of the 8 + 8 interleaved pair
int launch_silu_mul_interleaved(const void* in, void* out, int m, int n2, int dtype, hipStream_t st);
int launch_silu_mul(const void* gate, const void* up, void* out, size_t count, int dtype, hipStream_t st);  // out = T(T(silu(gate)) * up), count % 8 == 0
bool moe_v6_enabled();  // knob moe_v6 (default on)
// awq_gemm_v6.hip: 256 x 256 blocks of four software-pipelined waves (256 x 64 per wave, weights in registers, x through ds_write)
void launch_gemm_cdna4_v6(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int n_begin,
                          int n_end, int dtype, hipStream_t st, int bits = 4, int epi = 0, int szfmt = 0, int tile_n = 256);
void gemm_v6_set_probe(int v);
// K split over pairs of 256 x 256 blocks inside one launch (tiles that fill at most half the chip, long K: down_proj at 2048 rows); -1 = not served
bool gemm_v6_pair_takes(int m, int n, int k);
int gemm_v6_pair_lost(unsigned* count);  // awq_gemm_v6.hip: pair blocks of the current device that gave up waiting for their partner (outputs NaN) since load
int gemm_cdna4_v3_pair_plan(int m, int n, int k);  // awq_gemm_plan.hip: 1 = the prefill call (with its workspace) takes the block-pair K split
size_t gemm_v6_pair_workspace_bytes(int m, int n, int k);
int launch_gemm_cdna4_v6_pair(const void* x, const void* qw, const void* szp, const void* bias, void* out, int m, int n, int k, int n_begin, int n_end,
                              int dtype, void* ws, size_t ws_bytes, hipStream_t st, int bits = 4, int epi = 0, int szfmt = 0);
void gemm_v6_set_pair_lead(int v);
void gemm_v6_set_pair_min_nit(int v);
int launch_bias_add(void* out, const void* bias, int m, int n, int dtype, hipStream_t st);
// out[m, n] = T(in_f32) (+ bias in T); n % 8 == 0 (awq_util.hip)
int launch_round_bias_f32(const void* in_f32, const void* bias, void* out, int m, int n, int dtype, hipStream_t st);
// RMSNorm of m rows of k (awq_util.hip; layernorm.cu:39-61's arithmetic); -1 if k % 8 != 0
int launch_rmsnorm(const void* x, const void* gamma, float eps, void* out, int m, int k, int dtype, hipStream_t st);
int launch_unpack_v2(const void* qw, void* out_u8, int n, int k, hipStream_t st);
int launch_dequant_v2(const void* qw, const void* s, const void* z, void* out, int n, int k, int dtype, hipStream_t st);
int launch_pack_v2(const void* q_u8, void* qw, int n, int k, hipStream_t st);
int launch_repack_v1_to_v2(const void* qw1, const void* s1, const void* qz1, void* qw2, void* s2, void* sz2, int n, int k,
                           int gpad, int dtype, hipStream_t st);
}  // namespace awq
