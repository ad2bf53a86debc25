// extern "C" entry points declared in include/awq_cdna4.h: argument validation + dispatch.
// No torch types; errors are returned as codes (the Python/pybind layer turns them into
// RuntimeError with the reference's messages).
#include "../../include/awq_cdna4.h"

#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include "awq_kernels.hpp"

namespace {
thread_local char g_last_hip_error[256] = "";

int finish_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    strncpy(g_last_hip_error, hipGetErrorString(e), sizeof(g_last_hip_error) - 1);
    return AWQ_ERR_LAUNCH;
  }
  return AWQ_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_common(const void* x, const void* qweight, const void* scales, const void* zeros, const void* out, int m, int n,
                 int k, int group_size, int dtype) {
  if (!x || !qweight || !scales || !zeros || !out) return AWQ_ERR_NULL;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (group_size != 128) return AWQ_ERR_GROUP;
  if (m < 1 || n < 8 || k < 128 || (n % 8) != 0 || (k % 128) != 0) return AWQ_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(qweight) || !aligned16(out) || (reinterpret_cast<uintptr_t>(scales) & 1u) ||
      (reinterpret_cast<uintptr_t>(zeros) & 1u))
    return AWQ_ERR_ALIGN;
  return AWQ_OK;
}
}  // namespace

extern "C" {

int awq_abi_version(void) { return AWQ_ABI_VERSION; }

const char* awq_last_hip_error(void) { return g_last_hip_error; }

const char* awq_status_string(int status) {
  switch (status) {
    case AWQ_OK: return "ok";
    case AWQ_ERR_BATCH: return "Unsupported batch size for gemv kernel.";
    case AWQ_ERR_GROUP: return "Unsupported group size for gemv kernel.";
    case AWQ_ERR_DTYPE: return "Unsupported dtype: expected float16 or bfloat16.";
    case AWQ_ERR_SHAPE: return "Unsupported shape: need n % 8 == 0, k % 128 == 0, m >= 1.";
    case AWQ_ERR_ALIGN: return "Pointer is not 16-byte aligned (tensors must be contiguous).";
    case AWQ_ERR_NULL: return "NULL pointer argument.";
    case AWQ_ERR_WORKSPACE: return "Workspace too small.";
    case AWQ_ERR_LAUNCH: return "HIP launch failed.";
    case AWQ_ERR_BITS: return "Only 4-bit (and the repo's 3-bit extension) are supported.";
    default: return "unknown status";
  }
}

int awq_w4a16_gemv(const void* x, const void* qweight, const void* scales, const void* scaled_zeros, void* out, int m,
                   int n, int k, int group_size, int dtype, void* stream) {
  if (group_size != 128) return AWQ_ERR_GROUP;
  if (m < 1 || m > 16) return AWQ_ERR_BATCH;
  int st = check_common(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype);
  if (st != AWQ_OK) return st;
  awq::launch_gemv(x, qweight, scales, scaled_zeros, nullptr, out, m, n, k, dtype, 0, (hipStream_t)stream);
  return finish_launch();
}

size_t awq_w4a16_gemm_workspace_bytes(int m, int n, int k) { return awq::gemm_workspace_bytes(m, n, k); }

int awq_w4a16_gemm(const void* x, const void* qweight, const void* scales, const void* scaled_zeros, void* out, int m,
                   int n, int k, int group_size, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  int st = check_common(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype);
  if (st != AWQ_OK) return st;
  const size_t need = awq::gemm_workspace_bytes(m, n, k);
  if (need > 0 && (!workspace || workspace_bytes < need)) return AWQ_ERR_WORKSPACE;
  awq::launch_gemm(x, qweight, scales, scaled_zeros, nullptr, out, m, n, k, dtype, 0, workspace, workspace_bytes, (hipStream_t)stream);
  return finish_launch();
}

int awq_w4a16_forward(const void* x, const void* qweight, const void* scales, const void* scaled_zeros, const void* bias,
                      void* out, int m, int n, int k, int group_size, int dtype, void* workspace, size_t workspace_bytes,
                      void* stream) {
  int st;
  if (m < 8 && bias && group_size == 128 &&
      check_common(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype) == AWQ_OK && awq::gemv_v2fast_enabled() &&
      awq::launch_gemv_v2fast(x, qweight, scales, scaled_zeros, bias, out, m, n, k, k / 128, dtype, (hipStream_t)stream) == 0)
    return finish_launch();  // bias fused into the decode kernel's epilogue
  if (m > 8 && m <= 255 && bias && group_size == 128 && awq::gemm_variant_get() == 0 &&
      check_common(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype) == AWQ_OK &&
      awq::launch_skinny_v2(x, qweight, scales, scaled_zeros, bias, out, m, n, k, k / 128, dtype, (hipStream_t)stream) == 0)
    return finish_launch();  // bias fused into the skinny kernel's epilogue
  if (m < 8)
    st = awq_w4a16_gemv(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype, stream);
  else
    st = awq_w4a16_gemm(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype, workspace, workspace_bytes,
                        stream);
  if (st != AWQ_OK || !bias) return st;
  awq::launch_bias_add(out, bias, m, n, dtype, (hipStream_t)stream);
  return finish_launch();
}

int awq_unpack_v2(const void* qweight, void* out_u8, int n, int k, void* stream) {
  if (!qweight || !out_u8) return AWQ_ERR_NULL;
  if (n < 4 || (n % 4) != 0 || k < 64 || (k % 64) != 0) return AWQ_ERR_SHAPE;
  awq::launch_unpack_v2(qweight, out_u8, n, k, (hipStream_t)stream);
  return finish_launch();
}

int awq_dequant_v2(const void* qweight, const void* scales, const void* scaled_zeros, void* out, int n, int k,
                   int group_size, int dtype, void* stream) {
  if (!qweight || !scales || !scaled_zeros || !out) return AWQ_ERR_NULL;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (group_size != 128) return AWQ_ERR_GROUP;
  if (n < 4 || (n % 4) != 0 || k < 128 || (k % 128) != 0) return AWQ_ERR_SHAPE;
  awq::launch_dequant_v2(qweight, scales, scaled_zeros, out, n, k, dtype, (hipStream_t)stream);
  return finish_launch();
}

int awq_pack_v2(const void* q_u8, void* qweight, int n, int k, void* stream) {
  if (!q_u8 || !qweight) return AWQ_ERR_NULL;
  if (n < 4 || (n % 4) != 0 || k < 64 || (k % 64) != 0) return AWQ_ERR_SHAPE;
  awq::launch_pack_v2(q_u8, qweight, n, k, (hipStream_t)stream);
  return finish_launch();
}

int awq_repack_v1_to_v2(const void* qweight_v1, const void* scales_v1, const void* qzeros_v1, void* qweight_v2,
                        void* scales_v2, void* scaled_zeros_v2, int n, int k, int gpad, int dtype, void* stream) {
  if (!qweight_v1 || !scales_v1 || !qzeros_v1 || !qweight_v2 || !scales_v2 || !scaled_zeros_v2) return AWQ_ERR_NULL;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (n < 4 || (n % 4) != 0 || k < 64 || (k % 64) != 0 || gpad < 8 || (gpad % 8) != 0) return AWQ_ERR_SHAPE;
  awq::launch_repack_v1_to_v2(qweight_v1, scales_v1, qzeros_v1, qweight_v2, scales_v2, scaled_zeros_v2, n, k, gpad, dtype,
                              (hipStream_t)stream);
  return finish_launch();
}

// ---- cdna4 interleave (bf16) ----
int awq_repack_v2_to_cdna4(const void* qweight_v2, void* qweight_cdna4, int n, int k, void* stream) {
  if (!qweight_v2 || !qweight_cdna4) return AWQ_ERR_NULL;
  if (qweight_v2 == qweight_cdna4) return AWQ_ERR_ALIGN;  // not an in-place permutation
  if (n < 16 || (n % 16) != 0 || k < 128 || (k % 128) != 0) return AWQ_ERR_SHAPE;
  awq::launch_repack_v2_cdna4(qweight_v2, qweight_cdna4, n, k, 1, (hipStream_t)stream);
  return finish_launch();
}

int awq_repack_cdna4_to_v2(const void* qweight_cdna4, void* qweight_v2, int n, int k, void* stream) {
  if (!qweight_v2 || !qweight_cdna4) return AWQ_ERR_NULL;
  if (qweight_v2 == qweight_cdna4) return AWQ_ERR_ALIGN;
  if (n < 16 || (n % 16) != 0 || k < 128 || (k % 128) != 0) return AWQ_ERR_SHAPE;
  awq::launch_repack_v2_cdna4(qweight_cdna4, qweight_v2, n, k, 0, (hipStream_t)stream);
  return finish_launch();
}

int awq_unpack_cdna4(const void* qweight, void* out_u8, int n, int k, void* stream) {
  if (!qweight || !out_u8) return AWQ_ERR_NULL;
  if (n < 16 || (n % 16) != 0 || k < 128 || (k % 128) != 0) return AWQ_ERR_SHAPE;
  awq::launch_unpack_cdna4(qweight, out_u8, n, k, (hipStream_t)stream);
  return finish_launch();
}

int awq_dequant_cdna4(const void* qweight, const void* scales, const void* scaled_zeros, void* out, int n, int k,
                      int group_size, int dtype, void* stream) {
  if (!qweight || !scales || !scaled_zeros || !out) return AWQ_ERR_NULL;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (group_size != 128) return AWQ_ERR_GROUP;
  if (n < 16 || (n % 16) != 0 || k < 128 || (k % 128) != 0) return AWQ_ERR_SHAPE;
  awq::launch_dequant_cdna4(qweight, scales, scaled_zeros, out, n, k, dtype, (hipStream_t)stream);
  return finish_launch();
}

int awq_pack_sz_cdna4(const void* scales, const void* scaled_zeros, void* sz_packed, int n, int k, void* stream) {
  if (!scales || !scaled_zeros || !sz_packed) return AWQ_ERR_NULL;
  if (n < 16 || (n % 16) != 0 || k < 128 || (k % 128) != 0) return AWQ_ERR_SHAPE;
  awq::launch_pack_sz_cdna4(scales, scaled_zeros, sz_packed, n, k, (hipStream_t)stream);
  return finish_launch();
}

int awq_pack_szh_cdna4(const void* scales, const void* scaled_zeros, void* sz_half, int* inexact_dev, int n, int k, int dtype,
                       void* stream) {
  if (!scales || !scaled_zeros || !sz_half || !inexact_dev) return AWQ_ERR_NULL;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (n < 16 || (n % 16) != 0 || k < 128 || (k % 128) != 0) return AWQ_ERR_SHAPE;
  awq::launch_pack_szh_cdna4(scales, scaled_zeros, sz_half, inexact_dev, n, k, dtype, (hipStream_t)stream);
  return finish_launch();
}

int awq_w4a16_decode_cdna4(const void* x, const void* qweight, const void* sz_half, const void* bias, void* out, int m, int n, int k,
                           int group_size, int dtype, int epilogue, void* stream) {
  if (!x || !qweight || !sz_half || !out) return AWQ_ERR_NULL;
  if (group_size != 128) return AWQ_ERR_GROUP;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (m < 1 || m > 8) return AWQ_ERR_BATCH;
  if (epilogue < 0 || epilogue > 2 || (epilogue != 0 && bias)) return AWQ_ERR_SHAPE;
  const int mult = epilogue == 1 ? 32 : 16;
  if (n < mult || (n % mult) != 0 || k < 128 || (k % 128) != 0) return AWQ_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(qweight) || !aligned16(out) || !aligned16(sz_half)) return AWQ_ERR_ALIGN;
  if (awq::launch_gemv_dma(x, qweight, sz_half, bias, out, m, n, k, epilogue, dtype, 1, (hipStream_t)stream) != 0) return AWQ_ERR_SHAPE;
  return finish_launch();
}

int awq_w4a16_gemv_cdna4(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                         const void* sz_packed, void* out, int m, int n, int k, int group_size, int dtype, void* stream) {
  if (group_size != 128) return AWQ_ERR_GROUP;
  if (m < 1 || m > 16) return AWQ_ERR_BATCH;
  int st = check_common(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype);
  if (st != AWQ_OK) return st;
  if ((n % 16) != 0) return AWQ_ERR_SHAPE;
  if (!(sz_packed && m <= 8 && awq::launch_gemv_cdna4(x, qweight, sz_packed, nullptr, out, m, n, k, 0, 4, dtype, (hipStream_t)stream) == 0))
    awq::launch_gemv(x, qweight, scales, scaled_zeros, sz_packed, out, m, n, k, dtype, 1, (hipStream_t)stream);
  return finish_launch();
}

int awq_w4a16_mlp_gate_up_cdna4(const void* x, const void* qweight_gate_up, const void* sz_packed, void* out, int m,
                                int n2, int k, int group_size, int dtype, void* stream) {
  if (!x || !qweight_gate_up || !sz_packed || !out) return AWQ_ERR_NULL;
  if (group_size != 128) return AWQ_ERR_GROUP;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (m < 1 || m > 8) return AWQ_ERR_BATCH;
  if (n2 < 32 || (n2 % 32) != 0 || k < 128 || (k % 128) != 0) return AWQ_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(qweight_gate_up) || !aligned16(out) || !aligned16(sz_packed)) return AWQ_ERR_ALIGN;
  if (awq::launch_gemv_cdna4(x, qweight_gate_up, sz_packed, nullptr, out, m, n2, k, 1, 4, dtype, (hipStream_t)stream) != 0)
    return AWQ_ERR_SHAPE;
  return finish_launch();
}

static int g_skinny16_szh = 1;     // knob skinny16_szh: 0 = 9 .. 16 rows on the T-typed sz_packed form as in rounds 1 - 5
static int g_mlp_skinny_max = 64;  // knob mlp_skinny_max: row counts up to this go to the skinny kernel's fused epilogue (8 = never)
int awq_w4a16_mlp_gate_up_forward_cdna4(const void* x, const void* qweight_interleaved, const void* sz_packed, const void* sz_half,
                                        void* out, int m, int n2, int k, int group_size, int dtype, void* stream) {
  return awq_w4a16_mlp_gate_up_forward_cdna4_ws(x, qweight_interleaved, sz_packed, sz_half, out, m, n2, k, group_size, dtype, nullptr, 0, stream);
}

size_t awq_w4a16_mlp_gate_up_forward_cdna4_workspace_bytes(int m, int n2, int k) {
  if (m <= 8) return 0;
  const size_t a = awq::gemm_cdna4_v3_workspace_bytes(m, n2, k), b = awq::midm_workspace_bytes(m, n2, k);
  return a > b ? a : b;
}

int awq_w4a16_mlp_gate_up_forward_cdna4_ws(const void* x, const void* qweight_interleaved, const void* sz_packed, const void* sz_half,
                                           void* out, int m, int n2, int k, int group_size, int dtype, void* workspace, size_t workspace_bytes,
                                           void* stream) {
  if (!x || !qweight_interleaved || !sz_packed || !out) return AWQ_ERR_NULL;
  if (group_size != 128) return AWQ_ERR_GROUP;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (m < 1) return AWQ_ERR_BATCH;
  if (n2 < 16 || (n2 % 16) != 0 || k < 128 || (k % 128) != 0 || (size_t)m * (size_t)k >= (1ull << 31)) return AWQ_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(qweight_interleaved) || !aligned16(out) || !aligned16(sz_packed) || (sz_half && !aligned16(sz_half)))
    return AWQ_ERR_ALIGN;
  if (m <= 8) {  // decode: one streaming launch, rows r and r + 8 of a slab paired in the epilogue
    if (awq::launch_gemv_dma(x, qweight_interleaved, sz_half ? sz_half : sz_packed, nullptr, out, m, n2, k, 2, dtype, sz_half ? 1 : 0,
                             (hipStream_t)stream) != 0)
      return AWQ_ERR_SHAPE;
    return finish_launch();
  }
  // 9 .. 255 rows: the mid-M kernel (x tile shared by the block, waves split N, K split across blocks), rows r and r + 8 of a slab paired in its epilogue
  if (awq::midm_takes(m, n2, k) &&
      awq::launch_midm_cdna4(x, qweight_interleaved, sz_half ? sz_half : sz_packed, nullptr, out, m, n2, k, 2, dtype, sz_half ? 1 : 0, 4, 0, workspace, workspace_bytes,
                             (hipStream_t)stream) == 0)
    return finish_launch();
  // 9 .. 16 rows with the layer's sz_half side buffer: the skinny kernel's one-column-block shapes in the f16-mantissa dequant form (the batched-decode instantiations:
  // the same block shapes as launch_skinny_gate_up picks at <= 16 rows, ~10 VALU fewer per tile -- round 6, third session)
  if (m <= 16 && sz_half && g_skinny16_szh &&
      awq::launch_skinny_decode(x, qweight_interleaved, sz_half, nullptr, out, m, n2, k, 2, dtype, 1, (hipStream_t)stream, 0) == 0)
    return finish_launch();
  // (knob midm = 0) 9 .. 64 rows: one weight pass on the skinny kernel, rows r and r + 8 of a slab paired in its epilogue (a 256-row tile masked down to m rows costs
  // the same for every m: 46 vs 31 us at 64 rows on Llama-3-8B's pair, profiles/r05_skinny_splitk.txt)
  if (m <= g_mlp_skinny_max && awq::launch_skinny_gate_up(x, qweight_interleaved, sz_packed, out, m, n2, k, dtype, (hipStream_t)stream) == 0)
    return finish_launch();
  // prefill / batched decode: the tile kernels with the SiLU * mul tail fused into their epilogue (out is [m, n2 / 2]: the
  // [m, n2] intermediate of the reference's two GEMMs + F.silu + multiply never exists)
  if (workspace && ((reinterpret_cast<uintptr_t>(workspace) & 63) != 0 || workspace_bytes < awq::gemm_cdna4_v3_workspace_bytes(m, n2, k))) {
    workspace = nullptr;  // (optional scratch: without it the columns behind the full rounds run as 256 x 128 blocks)
    workspace_bytes = 0;
  }
  if (awq::launch_gemm_cdna4_v3(x, qweight_interleaved, sz_packed, nullptr, out, m, n2, k, 0, dtype, workspace, workspace_bytes, (hipStream_t)stream, 4, 2,
                                sz_half) != 0)
    return AWQ_ERR_SHAPE;
  return finish_launch();
}

int awq_w4a16_rmsnorm_forward_cdna4(const void* x, const void* gamma, float eps, const void* qweight, const void* sz_packed,
                                    const void* bias, void* out, int m, int n, int k, int group_size, int dtype,
                                    int fused_gate_up, void* stream) {
  if (!x || !gamma || !qweight || !sz_packed || !out) return AWQ_ERR_NULL;
  if (group_size != 128) return AWQ_ERR_GROUP;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (m < 1 || m > 4) return AWQ_ERR_BATCH;
  const int mult = fused_gate_up ? 32 : 16;
  if (n < mult || (n % mult) != 0 || k < 128 || (k % 128) != 0 || (fused_gate_up && bias)) return AWQ_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(gamma) || !aligned16(qweight) || !aligned16(out) || !aligned16(sz_packed)) return AWQ_ERR_ALIGN;
  if (awq::launch_gemv_cdna4_norm(x, gamma, eps, qweight, sz_packed, bias, out, m, n, k, fused_gate_up ? 1 : 0, dtype,
                                  (hipStream_t)stream) != 0)
    return AWQ_ERR_SHAPE;
  return finish_launch();
}

int awq_rmsnorm(const void* x, const void* gamma, float eps, void* out, int m, int k, int dtype, void* stream) {
  if (!x || !gamma || !out) return AWQ_ERR_NULL;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (m < 1) return AWQ_ERR_BATCH;
  if (k < 8 || (k % 8) != 0) return AWQ_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(gamma) || !aligned16(out)) return AWQ_ERR_ALIGN;
  if (awq::launch_rmsnorm(x, gamma, eps, out, m, k, dtype, (hipStream_t)stream) != 0) return AWQ_ERR_SHAPE;
  return finish_launch();
}

size_t awq_w4a16_forward_cdna4_workspace_bytes(int m, int n, int k) {
  if (awq::midm_takes(m, n, k)) return awq::midm_workspace_bytes(m, n, k);  // 9 .. 255 rows: the fp32 parts of the mid-M kernel's K split
  if (m > 8 && m < 256 && !awq::gemm_cdna4_v3_takes(m, k)) return awq::skinny_splitk_workspace_bytes(m, n, k);  // (knob midm = 0) the skinny launch's K split
  return awq::gemm_cdna4_v3_workspace_bytes(m, n, k);
}
int awq_midm_init(void) { return awq::midm_init() == 0 ? AWQ_OK : AWQ_ERR_LAUNCH; }
int awq_w4a16_gemm_cdna4_pair_plan(int m, int n, int k) { return awq::gemm_cdna4_v3_pair_plan(m, n, k); }
int awq_w4a16_gemm_cdna4_pair_lost(unsigned int* count) { return awq::gemm_v6_pair_lost(count) == 0 ? AWQ_OK : AWQ_ERR_LAUNCH; }
int awq_w4a16_gemm_cdna4_plan(int m, int n, int bits, int* mode, int* cols_main) { return awq::gemm_cdna4_v3_plan(m, n, bits, mode, cols_main); }
int awq_w4a16_gemm_cdna4_narrow_kernel(int m, int n_cols, int k, int bits, int has_workspace, int epilogue) {
  return awq::gemm_cdna4_v3_narrow_kernel(m, n_cols, k, bits, has_workspace, epilogue);
}
int awq_w4a16_decode_cdna4_plan(int m, int n, int k, int epilogue, int* kernel) { return awq::gemv_dma_plan(m, n, k, epilogue, kernel); }

int awq_w4a16_gemm_cdna4(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                         const void* sz_packed, void* out, int m, int n, int k, int group_size, int dtype, void* workspace,
                         size_t workspace_bytes, void* stream) {
  int st = check_common(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype);
  if (st != AWQ_OK) return st;
  if ((n % 16) != 0) return AWQ_ERR_SHAPE;
  const bool skinny = sz_packed && m > 8 && m < 256 && awq::gemm_variant_get() == 0 && !awq::gemm_cdna4_v3_takes(m, k) &&
                      awq::launch_skinny_cdna4(x, qweight, sz_packed, nullptr, out, m, n, k, dtype, (hipStream_t)stream, 0, workspace, workspace_bytes) == 0;
  if (!skinny && !(sz_packed && m <= 8 && awq::launch_gemv_cdna4(x, qweight, sz_packed, nullptr, out, m, n, k, 0, 4, dtype, (hipStream_t)stream) == 0))
    awq::launch_gemm(x, qweight, scales, scaled_zeros, sz_packed, out, m, n, k, dtype, 1, workspace, workspace_bytes, (hipStream_t)stream);
  return finish_launch();
}

int awq_w4a16_forward_cdna4_szh(const void* x, const void* qweight, const void* scales, const void* scaled_zeros, const void* sz_packed,
                                const void* sz_half, const void* bias, void* out, int m, int n, int k, int group_size, int dtype, void* workspace,
                                size_t workspace_bytes, void* stream) {
  if (sz_half && sz_packed && group_size == 128 && aligned16(sz_half) && awq::midm_takes(m, n, k)) {
    // 9 .. 255 rows with the layer's sz_half side buffer: the mid-M kernel in the f16-mantissa dequant form
    int st0 = check_common(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype);
    if (st0 != AWQ_OK) return st0;
    if (bias && !aligned16(bias)) return AWQ_ERR_ALIGN;
    if (awq::launch_midm_cdna4(x, qweight, sz_half, bias, out, m, n, k, 0, dtype, 1, 4, 0, workspace, workspace_bytes, (hipStream_t)stream) == 0) return finish_launch();
  }
  if (sz_half && sz_packed && m >= 9 && m <= 16 && group_size == 128 && g_skinny16_szh && awq::gemm_variant_get() == 0 && aligned16(sz_half) && (n % 16) == 0) {
    // 9 .. 16 rows with the layer's sz_half side buffer: the skinny kernel's one-column-block shapes in the f16-mantissa dequant form
    int st0 = check_common(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype);
    if (st0 != AWQ_OK) return st0;
    if (bias && !aligned16(bias)) return AWQ_ERR_ALIGN;
    if (awq::launch_skinny_decode(x, qweight, sz_half, bias, out, m, n, k, 0, dtype, 1, (hipStream_t)stream, 0) == 0) return finish_launch();
  }
  if (sz_half && sz_packed && m >= 256 && group_size == 128 && awq::gemm_variant_get() == 0 && aligned16(sz_half)) {
    // prefill with the layer's sz_half side buffer: the tile kernels dequantise in the f16-mantissa form (every block width, the block pairs included)
    int st0 = check_common(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype);
    if (st0 != AWQ_OK) return st0;
    if ((n % 16) != 0 || (bias && !aligned16(bias))) return (n % 16) ? AWQ_ERR_SHAPE : AWQ_ERR_ALIGN;
    if (awq::launch_gemm_cdna4_v3(x, qweight, sz_packed, bias, out, m, n, k, 0, dtype, workspace, workspace_bytes, (hipStream_t)stream, 4, 0, sz_half) == 0)
      return finish_launch();
  }
  return awq_w4a16_forward_cdna4(x, qweight, scales, scaled_zeros, sz_packed, bias, out, m, n, k, group_size, dtype, workspace, workspace_bytes, stream);
}

int awq_w4a16_forward_cdna4(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                            const void* sz_packed, const void* bias, void* out, int m, int n, int k, int group_size,
                            int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (m <= 8 && sz_packed && group_size == 128) {  // decode: bias fused into the GEMV epilogue
    int st0 = check_common(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype);
    if (st0 != AWQ_OK) return st0;
    if ((n % 16) != 0) return AWQ_ERR_SHAPE;
    if (awq::launch_gemv_cdna4(x, qweight, sz_packed, bias, out, m, n, k, 0, 4, dtype, (hipStream_t)stream) == 0) return finish_launch();
  }
  if (sz_packed && group_size == 128 && awq::gemm_variant_get() == 0 && awq::midm_takes(m, n, k)) {
    // 9 .. 255 rows: the mid-M kernel, bias fused
    int st0 = check_common(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype);
    if (st0 != AWQ_OK) return st0;
    if (bias && !aligned16(bias)) return AWQ_ERR_ALIGN;
    if (awq::launch_midm_cdna4(x, qweight, sz_packed, bias, out, m, n, k, 0, dtype, 0, 4, 0, workspace, workspace_bytes, (hipStream_t)stream) == 0) return finish_launch();
  }
  if (m > 8 && m < 256 && sz_packed && group_size == 128 && awq::gemm_variant_get() == 0 && !awq::gemm_cdna4_v3_takes(m, k)) {
    // (knob midm = 0) short prompts / batched decode: skinny kernel, bias fused
    int st0 = check_common(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype);
    if (st0 != AWQ_OK) return st0;
    if ((n % 16) != 0) return AWQ_ERR_SHAPE;
    if (awq::launch_skinny_cdna4(x, qweight, sz_packed, bias, out, m, n, k, dtype, (hipStream_t)stream, 0, workspace, workspace_bytes) == 0) return finish_launch();
  }
  if (bias && awq::gemm_cdna4_v3_takes(m, k) && sz_packed && group_size == 128 && awq::gemm_variant_get() == 0) {
    // prefill: bias fused into the prefill GEMM epilogue (awq_gemm_v4.hip / awq_gemm_v4n.hip) (no second kernel)
    int st0 = check_common(x, qweight, scales, scaled_zeros, out, m, n, k, group_size, dtype);
    if (st0 != AWQ_OK) return st0;
    if ((n % 16) != 0 || !aligned16(bias)) return (n % 16) ? AWQ_ERR_SHAPE : AWQ_ERR_ALIGN;
    if (awq::launch_gemm_cdna4_v3(x, qweight, sz_packed, bias, out, m, n, k, 0, dtype, workspace, workspace_bytes, (hipStream_t)stream) == 0)
      return finish_launch();
  }
  int st = awq_w4a16_gemm_cdna4(x, qweight, scales, scaled_zeros, sz_packed, out, m, n, k, group_size, dtype, workspace,
                                workspace_bytes, stream);  // m <= 16 is routed to the GEMV inside
  if (st != AWQ_OK || !bias) return st;
  awq::launch_bias_add(out, bias, m, n, dtype, (hipStream_t)stream);
  return finish_launch();
}

// ---- tensor-parallel row split: the K shard's fp32 partial + the single rounding after the sum ----
int awq_w4a16_partial_cdna4(const void* x, const void* qweight, const void* sz_packed, const void* sz_half, float* out_f32, int m, int n, int k,
                            int group_size, int dtype, void* stream) {
  if (!x || !qweight || !sz_packed || !out_f32) return AWQ_ERR_NULL;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (group_size != 128) return AWQ_ERR_GROUP;
  if (m < 1 || n < 16 || k < 128 || (n % 16) != 0 || (k % 128) != 0) return AWQ_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(qweight) || !aligned16(out_f32) || !aligned16(sz_packed) || !aligned16(sz_half)) return AWQ_ERR_ALIGN;
  const hipStream_t st = (hipStream_t)stream;
  if (m <= 8) {  // decode: the streaming kernel (or the skinny kernel where it takes the row count), f16-mantissa dequant when the layer allows it
    if (awq::launch_gemv_dma(x, qweight, sz_half ? sz_half : sz_packed, nullptr, out_f32, m, n, k, 0, dtype, sz_half ? 1 : 0, st, 1) == 0) return finish_launch();
    if (awq::launch_skinny_decode(x, qweight, sz_packed, nullptr, out_f32, m, n, k, 0, dtype, 0, st, 1) == 0) return finish_launch();
    return AWQ_ERR_SHAPE;
  }
  if (awq::midm_takes(m, n, k) &&
      awq::launch_midm_cdna4(x, qweight, sz_half ? sz_half : sz_packed, nullptr, out_f32, m, n, k, 0, dtype, sz_half ? 1 : 0, 4, 1, nullptr, 0, st) == 0)
    return finish_launch();
  if (m < 256 && !awq::gemm_cdna4_v3_takes(m, k)) {
    if (awq::launch_skinny_cdna4(x, qweight, sz_packed, nullptr, out_f32, m, n, k, dtype, st, 1) == 0) return finish_launch();
  }
  if (awq::launch_gemm_cdna4_v3(x, qweight, sz_packed, nullptr, out_f32, m, n, k, 0, dtype, nullptr, 0, st, 4, 3) != 0) return AWQ_ERR_SHAPE;
  return finish_launch();
}

int awq_round_bias_f32(const float* in_f32, const void* bias, void* out, int m, int n, int dtype, void* stream) {
  if (!in_f32 || !out) return AWQ_ERR_NULL;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (!aligned16(in_f32) || !aligned16(out) || !aligned16(bias)) return AWQ_ERR_ALIGN;
  if (awq::launch_round_bias_f32(in_f32, bias, out, m, n, dtype, (hipStream_t)stream) != 0) return AWQ_ERR_SHAPE;
  return finish_launch();
}

int awq_w4a16_moe_gemm(const void* x_sorted, const void* qweight, const void* scales, const void* scaled_zeros,
                       const void* expert_offsets, void* out, int total_tokens, int num_experts, int n, int k, int gpad,
                       int group_size, int dtype, int layout, void* stream) {
  if (!expert_offsets) return AWQ_ERR_NULL;
  if (num_experts < 1 || gpad * 128 < k || total_tokens < 0) return AWQ_ERR_SHAPE;
  if (layout != 0 && layout != 1) return AWQ_ERR_SHAPE;
  if (layout == 1 && (n % 16) != 0) return AWQ_ERR_SHAPE;
  if (total_tokens == 0) return AWQ_OK;
  int st = check_common(x_sorted, qweight, scales, scaled_zeros, out, total_tokens, n, k, group_size, dtype);
  if (st != AWQ_OK) return st;
  awq::launch_moe_gemm(x_sorted, qweight, scales, scaled_zeros, expert_offsets, out, total_tokens, num_experts, n, k, gpad,
                       dtype, layout, (hipStream_t)stream);
  return finish_launch();
}

int awq_w4a16_moe_forward_cdna4(const void* x_sorted, const void* qweight, const void* scales, const void* scaled_zeros,
                                const void* sz_packed, const void* expert_offsets, void* out, int total_tokens,
                                int num_experts, int n, int k, int gpad, int group_size, int dtype, void* stream) {
  return awq_w4a16_moe_forward_cdna4_szh(x_sorted, qweight, scales, scaled_zeros, sz_packed, nullptr, expert_offsets, out, total_tokens, num_experts, n, k, gpad,
                                         group_size, dtype, stream);
}

int awq_w4a16_moe_forward_cdna4_szh(const void* x_sorted, const void* qweight, const void* scales, const void* scaled_zeros, const void* sz_packed,
                                    const void* sz_half, const void* expert_offsets, void* out, int total_tokens, int num_experts, int n, int k,
                                    int gpad, int group_size, int dtype, void* stream) {
  if (!expert_offsets || !sz_packed) return AWQ_ERR_NULL;
  if (sz_half && !aligned16(sz_half)) return AWQ_ERR_ALIGN;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (num_experts < 1 || gpad * 128 < k || total_tokens < 0 || (n % 16) != 0) return AWQ_ERR_SHAPE;
  if (total_tokens == 0) return AWQ_OK;
  int st = check_common(x_sorted, qweight, scales, scaled_zeros, out, total_tokens, n, k, group_size, dtype);
  if (st != AWQ_OK) return st;
  if (total_tokens <= 8 && awq::launch_moe_gemv_cdna4(x_sorted, qweight, sz_packed, expert_offsets, out, total_tokens, num_experts, n,
                                                      k, dtype, (hipStream_t)stream) == 0)
    return finish_launch();
  if (total_tokens > 8 && total_tokens < 256 && awq::moe_v4_enabled() &&
      awq::launch_moe_skinny_cdna4(x_sorted, qweight, sz_packed, expert_offsets, out, total_tokens, num_experts, n, k, dtype,
                                   (hipStream_t)stream) == 0)
    return finish_launch();
  if (total_tokens >= 256 && awq::moe_v6_enabled() && awq::moe_v4_enabled() &&
      awq::launch_moe_gemm_cdna4_v6(x_sorted, qweight, sz_packed, expert_offsets, out, total_tokens, num_experts, n, k, dtype,
                                    (hipStream_t)stream, 0, sz_half) == 0)
    return finish_launch();
  awq::launch_moe_gemm(x_sorted, qweight, scales, scaled_zeros, expert_offsets, out, total_tokens, num_experts, n, k, gpad, dtype, 1,
                       (hipStream_t)stream);
  return finish_launch();
}

int awq_silu_mul(const void* gate, const void* up, void* out, size_t count, int dtype, void* stream) {
  if (!gate || !up || !out) return AWQ_ERR_NULL;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (!aligned16(gate) || !aligned16(up) || !aligned16(out)) return AWQ_ERR_ALIGN;
  if (count == 0) return AWQ_OK;
  if (awq::launch_silu_mul(gate, up, out, count, dtype, (hipStream_t)stream) != 0) return AWQ_ERR_SHAPE;
  return finish_launch();
}

int awq_w4a16_moe_mlp_gate_up_cdna4(const void* x_sorted, const void* qweight_interleaved, const void* scales, const void* scaled_zeros,
                                    const void* sz_packed, const void* expert_offsets, void* out, void* scratch, size_t scratch_bytes,
                                    int total_tokens, int num_experts, int n2, int k, int gpad, int group_size, int dtype, void* stream) {
  return awq_w4a16_moe_mlp_gate_up_cdna4_szh(x_sorted, qweight_interleaved, scales, scaled_zeros, sz_packed, nullptr, expert_offsets, out, scratch, scratch_bytes,
                                             total_tokens, num_experts, n2, k, gpad, group_size, dtype, stream);
}

int awq_w4a16_moe_mlp_gate_up_cdna4_szh(const void* x_sorted, const void* qweight_interleaved, const void* scales, const void* scaled_zeros,
                                        const void* sz_packed, const void* sz_half, const void* expert_offsets, void* out, void* scratch,
                                        size_t scratch_bytes, int total_tokens, int num_experts, int n2, int k, int gpad, int group_size, int dtype,
                                        void* stream) {
  if (!expert_offsets || !sz_packed || !x_sorted || !qweight_interleaved || !out) return AWQ_ERR_NULL;
  if (sz_half && !aligned16(sz_half)) return AWQ_ERR_ALIGN;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (group_size != 128) return AWQ_ERR_GROUP;
  if (num_experts < 1 || total_tokens < 0 || n2 < 32 || (n2 % 32) != 0 || k < 128 || (k % 128) != 0) return AWQ_ERR_SHAPE;
  if (total_tokens == 0) return AWQ_OK;
  if (!aligned16(x_sorted) || !aligned16(qweight_interleaved) || !aligned16(out) || !aligned16(sz_packed) || !aligned16(scratch)) return AWQ_ERR_ALIGN;
  const hipStream_t st = (hipStream_t)stream;
  if (total_tokens >= 256 && awq::moe_v6_enabled() &&
      awq::launch_moe_gemm_cdna4_v6(x_sorted, qweight_interleaved, sz_packed, expert_offsets, out, total_tokens, num_experts, n2, k, dtype, st, 2, sz_half) == 0)
    return finish_launch();
  // fewer than 256 sorted rows (grouped GEMV / grouped skinny kernel) -- or the v6 tile switched off: the pair's [total, n2] product goes
  // through the caller's scratch, then the SiLU * mul tail as its own launch
  if (!scratch || scratch_bytes < (size_t)total_tokens * n2 * 2) return AWQ_ERR_WORKSPACE;
  const int rc = awq_w4a16_moe_forward_cdna4(x_sorted, qweight_interleaved, scales, scaled_zeros, sz_packed, expert_offsets, scratch, total_tokens,
                                             num_experts, n2, k, gpad, group_size, dtype, stream);
  if (rc != AWQ_OK) return rc;
  if (awq::launch_silu_mul_interleaved(scratch, out, total_tokens, n2, dtype, st) != 0) return AWQ_ERR_SHAPE;
  return finish_launch();
}

// ---- W3 ("w3c") : the repository's 3-bit format (bf16 only; no reference counterpart) ----
static int check_w3_shape(int n, int k) { return (n < 16 || (n % 16) != 0 || k < 128 || (k % 128) != 0) ? AWQ_ERR_SHAPE : AWQ_OK; }

int awq_pack_w3(const void* q_u8, void* qweight_w3, int n, int k, void* stream) {
  if (!q_u8 || !qweight_w3) return AWQ_ERR_NULL;
  if (check_w3_shape(n, k)) return AWQ_ERR_SHAPE;
  awq::launch_pack_w3(q_u8, qweight_w3, n, k, (hipStream_t)stream);
  return finish_launch();
}

int awq_unpack_w3(const void* qweight_w3, void* out_u8, int n, int k, void* stream) {
  if (!qweight_w3 || !out_u8) return AWQ_ERR_NULL;
  if (check_w3_shape(n, k)) return AWQ_ERR_SHAPE;
  awq::launch_unpack_w3(qweight_w3, out_u8, n, k, (hipStream_t)stream);
  return finish_launch();
}

int awq_dequant_w3(const void* qweight_w3, const void* scales, const void* scaled_zeros, void* out, int n, int k,
                   int group_size, int dtype, void* stream) {
  if (!qweight_w3 || !scales || !scaled_zeros || !out) return AWQ_ERR_NULL;
  if (dtype != AWQ_BF16 && dtype != AWQ_F16) return AWQ_ERR_DTYPE;
  if (group_size != 128) return AWQ_ERR_GROUP;
  if (check_w3_shape(n, k)) return AWQ_ERR_SHAPE;
  awq::launch_dequant_w3(qweight_w3, scales, scaled_zeros, out, n, k, dtype, (hipStream_t)stream);
  return finish_launch();
}

static int g_w3_skinny_max = 64;  // knob w3_skinny_max: 3-bit row counts up to this go to the skinny kernel (8 = never: the masked tile of the prefill GEMM)
// w3c tiles are read natively by every kernel: the only workspace is the OPTIONAL split-K scratch of short prompts (as for W4)
size_t awq_w3a16_forward_workspace_bytes(int m, int n, int k) { return m <= 8 ? 0 : awq::gemm_cdna4_v3_workspace_bytes_w3(m, n, k); }

int awq_w3a16_forward(const void* x, const void* qweight_w3, const void* scales, const void* scaled_zeros,
                      const void* sz_packed, const void* bias, void* out, int m, int n, int k, int group_size, int dtype,
                      void* workspace, size_t workspace_bytes, void* stream) {
  if (!sz_packed) return AWQ_ERR_NULL;
  int st = check_common(x, qweight_w3, scales, scaled_zeros, out, m, n, k, group_size, dtype);
  if (st != AWQ_OK) return st;
  if (check_w3_shape(n, k)) return AWQ_ERR_SHAPE;
  if (m <= 8) {
    // decode: the register-ring kernel (the LDS-DMA streaming kernel on w3c tiles measured equal to 4 % slower: profiles/r05_w3_streaming.txt)
    if (awq::launch_gemv_cdna4(x, qweight_w3, sz_packed, bias, out, m, n, k, 0, 3, dtype, (hipStream_t)stream) != 0)
      return AWQ_ERR_SHAPE;
    return finish_launch();
  }
  if (bias && !aligned16(bias)) return AWQ_ERR_ALIGN;
  // 9 .. 64 rows: one weight pass on the skinny kernel (w3c tiles), instead of a 256-row tile masked down to m rows
  if (m <= g_w3_skinny_max && awq::launch_skinny_w3(x, qweight_w3, sz_packed, bias, out, m, n, k, 0, dtype, (hipStream_t)stream) == 0) return finish_launch();
  // prefill / batched decode: the v4 / v4n weight producers read the 768-byte tiles directly (three words per lane, the fourth
  // rebuilt with six VALU operations per group); bias fused into the epilogue
  if (workspace && (!aligned16(workspace) || workspace_bytes < awq_w3a16_forward_workspace_bytes(m, n, k))) {
    workspace = nullptr;
    workspace_bytes = 0;
  }
  if (awq::launch_gemm_cdna4_v3(x, qweight_w3, sz_packed, bias, out, m, n, k, 0, dtype, workspace, workspace_bytes, (hipStream_t)stream, 3) != 0)
    return AWQ_ERR_SHAPE;
  return finish_launch();
}

// QuantLlamaMLP on 3-bit gate / up projections (tinychat/modules/fused_mlp.py:33-83 with this repository's w_bit = 3): the two projections'
// rows interleaved 8 + 8 per 16-row slab (integer rows, then packed into w3c tiles), out[m, n2 / 2] = silu(gate) * up in the kernels' epilogue
size_t awq_w3a16_mlp_gate_up_forward_workspace_bytes(int m, int n2, int k) { return m > 8 ? awq::gemm_cdna4_v3_workspace_bytes_w3(m, n2, k) : 0; }

int awq_w3a16_mlp_gate_up_forward(const void* x, const void* qweight_w3_interleaved, const void* sz_packed, void* out, int m, int n2, int k,
                                  int group_size, int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (!x || !qweight_w3_interleaved || !sz_packed || !out) return AWQ_ERR_NULL;
  if (group_size != 128) return AWQ_ERR_GROUP;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (m < 1) return AWQ_ERR_BATCH;
  if (n2 < 32 || (n2 % 32) != 0 || k < 128 || (k % 128) != 0 || (size_t)m * (size_t)k >= (1ull << 31)) return AWQ_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(qweight_w3_interleaved) || !aligned16(out) || !aligned16(sz_packed)) return AWQ_ERR_ALIGN;
  const hipStream_t st = (hipStream_t)stream;
  if (m <= 8) {
    if (awq::launch_gemv_cdna4(x, qweight_w3_interleaved, sz_packed, nullptr, out, m, n2, k, 2, 3, dtype, st) != 0) return AWQ_ERR_SHAPE;
    return finish_launch();
  }
  if (m <= g_w3_skinny_max && awq::launch_skinny_w3(x, qweight_w3_interleaved, sz_packed, nullptr, out, m, n2, k, 2, dtype, st) == 0) return finish_launch();
  if (workspace && ((reinterpret_cast<uintptr_t>(workspace) & 63) != 0 || workspace_bytes < awq_w3a16_mlp_gate_up_forward_workspace_bytes(m, n2, k))) {
    workspace = nullptr;
    workspace_bytes = 0;
  }
  if (awq::launch_gemm_cdna4_v3(x, qweight_w3_interleaved, sz_packed, nullptr, out, m, n2, k, 0, dtype, workspace, workspace_bytes, st, 3, 2) != 0)
    return AWQ_ERR_SHAPE;
  return finish_launch();
}

// tensor-parallel row split of a 3-bit layer: the K shard's product as fp32 [m, n], unrounded, no bias (the W3 twin of
// awq_w4a16_partial_cdna4; awq_round_bias_f32 rounds the reduced sum once)
int awq_w3a16_partial(const void* x, const void* qweight_w3, const void* sz_packed, float* out_f32, int m, int n, int k,
                      int group_size, int dtype, void* stream) {
  if (!x || !qweight_w3 || !sz_packed || !out_f32) return AWQ_ERR_NULL;
  if (dtype != AWQ_F16 && dtype != AWQ_BF16) return AWQ_ERR_DTYPE;
  if (group_size != 128) return AWQ_ERR_GROUP;
  if (m < 1 || check_w3_shape(n, k)) return AWQ_ERR_SHAPE;
  if (!aligned16(x) || !aligned16(qweight_w3) || !aligned16(out_f32) || !aligned16(sz_packed)) return AWQ_ERR_ALIGN;
  const hipStream_t st = (hipStream_t)stream;
  if (m <= 8) {
    if (awq::launch_gemv_cdna4(x, qweight_w3, sz_packed, nullptr, out_f32, m, n, k, 3, 3, dtype, st) != 0) return AWQ_ERR_SHAPE;
    return finish_launch();
  }
  if (m <= g_w3_skinny_max && awq::launch_skinny_w3(x, qweight_w3, sz_packed, nullptr, out_f32, m, n, k, 0, dtype, st, 1) == 0) return finish_launch();
  if (awq::launch_gemm_cdna4_v3(x, qweight_w3, sz_packed, nullptr, out_f32, m, n, k, 0, dtype, nullptr, 0, st, 3, 3) != 0) return AWQ_ERR_SHAPE;
  return finish_launch();
}

int awq_tune_set(const char* key, int value) {
  if (!key) return AWQ_ERR_NULL;
  // the knobs select between shipped code paths (tests force each of them) or, in AWQ_PROBES builds, timing probes; all of them
  // can change the summation order of results, so a default process cannot reach them: AWQ_TUNING=1 in the environment opts in
  const char* gate = getenv("AWQ_TUNING");
  if (!gate || gate[0] != '1') return AWQ_ERR_SHAPE;
  if (awq::gemv_tune_set(key, value) == 0) return AWQ_OK;
  if (awq::gemm_tune_set(key, value) == 0) return AWQ_OK;
  if (awq::gemv_cdna4_tune_set(key, value) == 0) return AWQ_OK;
  if (awq::skinny_tune_set(key, value) == 0) return AWQ_OK;
  if (awq::midm_tune_set(key, value) == 0) return AWQ_OK;
  if (awq::gemm_v3_tune_set(key, value) == 0) return AWQ_OK;
  if (!strcmp(key, "w3_skinny_max")) {
    g_w3_skinny_max = value;
    return AWQ_OK;
  }
  if (!strcmp(key, "mlp_skinny_max")) {
    g_mlp_skinny_max = value;
    return AWQ_OK;
  }
  if (!strcmp(key, "skinny16_szh")) {
    g_skinny16_szh = value;
    return AWQ_OK;
  }
  return AWQ_ERR_SHAPE;
}

}  // extern "C"
