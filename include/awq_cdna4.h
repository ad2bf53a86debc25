/*
 * awq_cdna4.h -- C ABI of the MI355X-native (gfx950 / CDNA4) W4A16 fused dequant+matmul path.
 *
 * This is the drop-in boundary for the ONE hot path of mit-han-lab/llm-awq that this repository
 * accelerates: the two extension entry points behind `awq.quantize.qmodule.WQLinear.forward`
 *     awq_inference_engine.gemv_forward_cuda_new   (awq/kernels/csrc/quantization_new/gemv/gemv_cuda.h:4-12,
 *                                                   gemv_cuda.cu:245-338, bound at awq/kernels/csrc/pybind.cpp:23)
 *     awq_inference_engine.gemm_forward_cuda_new   (awq/kernels/csrc/quantization_new/gemm/gemm_cuda.h:3,
 *                                                   gemm_cuda.cu:1126-1236, bound at awq/kernels/csrc/pybind.cpp:22)
 * plus the data-format helpers either side of it (tinychat/offline-weight-repacker.py).
 *
 * Plain pointers and sizes only -- no torch types.  All pointers are DEVICE pointers unless
 * stated; `stream` is a hipStream_t passed as void* (NULL = the null stream).  Every function
 * returns AWQ_OK (0) or a negative AWQ_ERR_* code and never throws; kernels are enqueued
 * asynchronously on `stream`.  Tensors follow the reference's v2 contract:
 *     qweight       int16 [N/4, K]      (awq/quantize/qmodule.py:26-65, 98-108)
 *     scales        T     [Gpad, N]     (qmodule.py:109-119), Gpad = 8*ceil(K/G/8) rows, only K/G used
 *     scaled_zeros  T     [Gpad, N]     (qmodule.py:120-130), = -(scales * zeros)
 *     x             T     [M, K] row-major contiguous,  out  T [M, N]
 * with T = fp16 or bf16 (`dtype`), G = group_size = 128.
 *
 * Numerics: each weight is materialised as round_T(q*s + sz) exactly as the
 * reference's __hfma2 does, products are accumulated in fp32 (MFMA), the result is rounded once to T.
 */
#ifndef AWQ_CDNA4_H_
#define AWQ_CDNA4_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AWQ_ABI_VERSION 1

/* status codes */
#define AWQ_OK 0
#define AWQ_ERR_BATCH (-1)      /* gemv: m outside [1,16] ("Unsupported batch size for gemv kernel", gemv_cuda.cu:328-329) */
#define AWQ_ERR_GROUP (-2)      /* group_size != 128 ("Unsupported group size for gemv kernel", gemv_cuda.cu:332-335) */
#define AWQ_ERR_DTYPE (-3)      /* dtype is neither AWQ_F16 nor AWQ_BF16 (dispatch_utils.cuh:7-18) */
#define AWQ_ERR_SHAPE (-4)      /* n % 8 != 0, k % 128 != 0, non-positive sizes */
#define AWQ_ERR_ALIGN (-5)      /* a pointer is not 16-byte aligned */
#define AWQ_ERR_NULL (-6)       /* NULL pointer */
#define AWQ_ERR_WORKSPACE (-7)  /* workspace too small */
#define AWQ_ERR_LAUNCH (-8)     /* hipLaunch / runtime error (hipGetLastError text via awq_last_hip_error) */
#define AWQ_ERR_BITS (-9)       /* unsupported w_bit */

/* activation / scale element type */
#define AWQ_F16 0
#define AWQ_BF16 1

int awq_abi_version(void);
const char* awq_status_string(int status);
const char* awq_last_hip_error(void);

/* Replaces gemv_forward_cuda_new (gemv_cuda.cu:245-338): out[m,n] = x[m,k] . Wdeq[n,k]^T for
 * 1 <= m <= 16 (the reference accepts 1..7; the Python binding keeps that limit). */
int awq_w4a16_gemv(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                   void* out, int m, int n, int k, int group_size, int dtype, void* stream);

/* Replaces gemm_forward_cuda_new (gemm_cuda.cu:1126-1236) for any m >= 1.  `workspace` may be
 * NULL when awq_w4a16_gemm_workspace_bytes(m,n,k) == 0. It is zero-initialised by the callee when
 * used (the reference's uninitialised semaphore tensor, gemm_cuda.cu:28, is not reproduced). */
size_t awq_w4a16_gemm_workspace_bytes(int m, int n, int k);
int awq_w4a16_gemm(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                   void* out, int m, int n, int k, int group_size, int dtype,
                   void* workspace, size_t workspace_bytes, void* stream);

/* WQLinear.forward's dispatch (qmodule.py:201-224): m < 8 -> gemv, else gemm; optional bias[n]
 * (may be NULL) added in T after the matmul result was rounded to T, like `out + self.bias`. */
int awq_w4a16_forward(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                      const void* bias, void* out, int m, int n, int k, int group_size, int dtype,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Parity / format helpers running the SAME device unpack + dequant code as the matmul kernels. */
/* out_u8[n, k] = logical 4-bit integer Q[n,k]  (inverse of pack_intweight, qmodule.py:26-65) */
int awq_unpack_v2(const void* qweight, void* out_u8, int n, int k, void* stream);
/* out[n, k] = round_T(Q[n,k] * scales[k/G, n] + scaled_zeros[k/G, n])  (gemv_cuda.cu:159-166) */
int awq_dequant_v2(const void* qweight, const void* scales, const void* scaled_zeros, void* out,
                   int n, int k, int group_size, int dtype, void* stream);
/* qweight_v2[n/4, k] from logical Q u8 [n, k]  (pack_intweight / packing_v2_from_unpacked) */
int awq_pack_v2(const void* q_u8, void* qweight, int n, int k, void* stream);
/* v1 checkpoint tensors -> v2 (tinychat/offline-weight-repacker.py:111-152):
 *   qweight_v1 int32 [n, k/8] -> qweight_v2 int16 [n/4, k]
 *   scales_v1 T [n, gpad], qzeros_v1 int32 [n, gpad/8] -> scales_v2 T [gpad, n], scaled_zeros_v2 T [gpad, n] */
int awq_repack_v1_to_v2(const void* qweight_v1, const void* scales_v1, const void* qzeros_v1,
                        void* qweight_v2, void* scales_v2, void* scaled_zeros_v2,
                        int n, int k, int gpad, int dtype, void* stream);

/* ---- "cdna4" interleave: this repository's MI355X-native int4 layout (what the rewritten
 * tinychat/offline-weight-repacker.py equivalent emits; DESIGN.md "cdna4 interleave").  Same bytes and
 * shape as v2 (int16 [N/4, K]); a pure nibble permutation that makes one 16-row x 128-k tile a contiguous
 * 1-KiB wave-load and lets the weights be dequantised on the matrix core.  bf16 and fp16; n % 16 == 0.
 * scales / scaled_zeros keep the v2 contract. ---- */
int awq_repack_v2_to_cdna4(const void* qweight_v2, void* qweight_cdna4, int n, int k, void* stream);
int awq_repack_cdna4_to_v2(const void* qweight_cdna4, void* qweight_v2, int n, int k, void* stream);
int awq_unpack_cdna4(const void* qweight_cdna4, void* out_u8, int n, int k, void* stream);
int awq_dequant_cdna4(const void* qweight_cdna4, const void* scales, const void* scaled_zeros, void* out,
                      int n, int k, int group_size, int dtype, void* stream);
/* sz_packed u32 [n/16][k/128][16] = {scale | scaled_zero << 16}: the scales re-laid next to the tiles so a lane
 * fetches both with one dword load per step.  Optional for the matmul entry points (NULL = read scales/zeros). */
int awq_pack_sz_cdna4(const void* scales, const void* scaled_zeros, void* sz_packed, int n, int k, void* stream);
/* sz_half u32 [n/16][k/128][16] = {f16(s') | f16(scaled_zero) << 16}, s' = scale for rows n % 4 < 2 and scale / 16 for the
 * others: the side buffer of the decode kernels' "f16-mantissa" dequant (csrc/awq_device.hpp Cdna4DequantH: two of the four
 * nibble extractions of a word need no shift when the dequant MFMA runs in its f16 form; exact, same single rounding).
 * *inexact_dev (device int, zeroed by the caller) is set when a scale or scaled zero is not exactly representable as a
 * normal f16 number: such a layer must keep using sz_packed.  dtype = type of scales / scaled_zeros. */
int awq_pack_szh_cdna4(const void* scales, const void* scaled_zeros, void* sz_half, int* inexact_dev, int n, int k, int dtype,
                       void* stream);
/* Decode (1 <= m <= 8) on cdna4 weights + sz_half: the fast path behind WQLinear.forward for m < 8 (replaces gemv_forward_cuda_new,
 * awq/kernels/csrc/quantization_new/gemv/gemv_cuda.cu:245-338) and QuantLlamaMLP's gate/up pair (tinychat/modules/fused_mlp.py:36-83).
 *   epilogue 0: out[m, n] = x . W^T (+ bias, in T)
 *   epilogue 1: qweight = [gate; up] stacked along N (n = 2 ffn), out[m, n/2] = silu(x . Wg^T) * (x . Wu^T), bias must be NULL
 *   epilogue 2: as 1 with gate / up rows interleaved 8 + 8 inside every 16-row slab (rows 16 j .. 16 j + 7 = gate rows 8 j ..,
 *               rows 16 j + 8 .. = the matching up rows): twice the blocks, every block one tile stream */
int awq_w4a16_decode_cdna4(const void* x, const void* qweight_cdna4, const void* sz_half, const void* bias, void* out, int m, int n,
                           int k, int group_size, int dtype, int epilogue, void* stream);
/* gemv on cdna4-interleaved weights (same contract as awq_w4a16_gemv otherwise) */
int awq_w4a16_gemv_cdna4(const void* x, const void* qweight_cdna4, const void* scales, const void* scaled_zeros,
                         const void* sz_packed, void* out, int m, int n, int k, int group_size, int dtype, void* stream);

/* QuantLlamaMLP's gate/up pair + SiLU*mul in ONE launch (tinychat/modules/fused_mlp.py:36-83 issues two
 * gemv_forward_cuda_new calls, F.silu and a multiply): qweight_gate_up = the gate and up projections' cdna4 buffers
 * stacked along N (n2 = 2 * intermediate rows, exactly what torch.cat([gate.qweight, up.qweight], 0) gives),
 * sz_packed built from the equally concatenated scales / scaled_zeros; out[m, n2/2] = silu(x.Wg^T) * (x.Wu^T), every
 * intermediate rounded to T like the reference's separate ops.  1 <= m <= 8, bf16 / fp16. */
int awq_w4a16_mlp_gate_up_cdna4(const void* x, const void* qweight_gate_up, const void* sz_packed, void* out, int m,
                                int n2, int k, int group_size, int dtype, void* stream);

/* QuantLlamaMLP.our_llama_mlp for ANY row count (tinychat/modules/fused_mlp.py:36-83: decode = two gemv_forward_cuda_new + F.silu +
 * multiply, prefill = two gemm_forward_cuda_new + F.silu + multiply) on the pair as llm_awq_amd.fused_mlp stacks it: gate and up
 * rows interleaved 8 + 8 inside every 16-row slab (n2 = 2 * intermediate rows).  out[m, n2/2] = T(T(silu(x.Wg^T)) * (x.Wu^T)).
 * m <= 8: one streaming launch (sz_half if given, else sz_packed); m > 8: the prefill tile kernels with the SiLU * mul tail fused
 * into their epilogue -- the [m, n2] intermediate is never written.  sz_half may be NULL. */
int awq_w4a16_mlp_gate_up_forward_cdna4(const void* x, const void* qweight_interleaved, const void* sz_packed, const void* sz_half,
                                        void* out, int m, int n2, int k, int group_size, int dtype, void* stream);
/* the same with an optional scratch buffer (awq_w4a16_mlp_gate_up_forward_cdna4_workspace_bytes, 64-byte aligned; NULL / too small: ignored): with it
 * the columns a prompt's full rounds of 256-wide tiles leave over run as block pairs (K split inside the launch, csrc/awq_gemm_v6.hip) instead of 256 x 128 blocks */
size_t awq_w4a16_mlp_gate_up_forward_cdna4_workspace_bytes(int m, int n2, int k);
int awq_w4a16_mlp_gate_up_forward_cdna4_ws(const void* x, const void* qweight_interleaved, const void* sz_packed, const void* sz_half, void* out, int m,
                                           int n2, int k, int group_size, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* gemm / WQLinear.forward dispatch on cdna4-interleaved weights (any m >= 1; m <= 16 runs the GEMV) */
/* RMSNorm fused in front of the quantised linear: replaces the FTLlamaRMSNorm launch
 * (tinychat/modules/fused_norm.py:7-21 -> awq/kernels/csrc/layernorm/layernorm.cu:39-61, layernorm_forward_cuda) followed by
 * WQLinear.forward / QuantLlamaMLP's gate/up pair.  x is the UN-normalised activation [m, k], gamma the norm weight [k]:
 *   xn = T((float(x) * rsqrtf(mean(x^2) + eps)) * float(gamma))          (layernorm.cu:55,60)
 *   fused_gate_up == 0: out[m, n]   = xn . W^T (+ bias)                  (qmodule.py:201-224)
 *   fused_gate_up != 0: out[m, n/2] = silu(xn . Wg^T) * (xn . Wu^T)      (fused_mlp.py:36-83; qweight = [gate; up], n = 2 * ffn)
 * Decode rows only: 1 <= m <= 4, k <= 16384; returns AWQ_ERR_BATCH / AWQ_ERR_SHAPE otherwise (run norm and linear separately). */
int awq_w4a16_rmsnorm_forward_cdna4(const void* x, const void* gamma, float eps, const void* qweight, const void* sz_packed,
                                    const void* bias, void* out, int m, int n, int k, int group_size, int dtype,
                                    int fused_gate_up, void* stream);

/* The norm alone, any row count (the prefill side of the same step; replaces layernorm_forward_cuda, awq/kernels/csrc/layernorm/
 * layernorm.cu:39-89 as tinychat's FTLlamaRMSNorm calls it, fused_norm.py:7-21): out[m, k] = T((float(x) * rsqrtf(mean(x^2) + eps)) *
 * float(gamma)), fp32 sum of squares, one rounding.  k % 8 == 0; x, gamma, out 16-byte aligned. */
int awq_rmsnorm(const void* x, const void* gamma, float eps, void* out, int m, int k, int dtype, void* stream);

/* OPTIONAL fp32 workspace of awq_w4a16_gemm_cdna4 / awq_w4a16_forward_cdna4 (0 = none useful).  Prompts of 256 .. ~1 k tokens
 * against narrow projections produce too few output tiles to fill the 256 CUs; given this many bytes (16-byte aligned) the
 * K loop is split over blocks and a second kernel adds the partial tiles in a fixed order -- the role of the reference's
 * split_k_iters + semaphore (gemm_cuda.cu:546-619); for awq_w4a16_gemm_cdna4_pair_plan's shapes the two halves of K meet inside
 * ONE launch (64-byte aligned workspace, any contents; one launch at a time per workspace).  Short prompts on the skinny kernel
 * (17 .. 146 rows) against n = 4096-class projections split K over two blocks per slab group the same way inside one
 * launch: fp32 parts in the workspace, added in part order by the block that draws the group's last ticket (the ticket words live in a
 * small zero-initialised per-device array the library allocates at the first such call outside a stream capture).  Without a workspace
 * the call runs unsplit (slower, same contract). */
size_t awq_w4a16_forward_cdna4_workspace_bytes(int m, int n, int k);
/* 9 .. 255 rows (round 6): the mid-M kernel (csrc/awq_midm_cdna4.hip) -- the row range of the reference's 16 / 32 / 64-row tiles + split_k_iters,
 * gemm_cuda.cu:1155-1206, :546-619.  Its K split across blocks keeps the fp32 parts in the caller's workspace and the ticket words in a library-owned,
 * zero-initialised per-device array in which a word belongs to ONE launch at a time: eager launches use the lane of their stream, a launch recorded
 * during a stream capture gets words of its own that are never handed out again (so graphs may be replayed on any streams); no lane / region left, or
 * no workspace: the call runs unsplit.  awq_midm_init allocates that array for the current device ahead of the first call (optional; the first
 * split launch outside a capture does it otherwise).  Returns AWQ_OK or AWQ_ERR_LAUNCH. */
int awq_midm_init(void);
/* host-side query (no GPU work): the tiles the prefill GEMM launches for an [m, n] output of a 3- or 4-bit matrix.  *mode: 0 = 256 x 256
 * blocks, 1 = 256 x 128, 2 = 256 x 256 for the first *cols_main column tiles and 256 x 128 for the rest, 3 = 256 x 192; returns the number
 * of thread blocks, 0 if the GEMM does not take this m (decode / skinny kernels do).  The counterpart of the reference's tile table,
 * gemm_cuda.cu:1155-1232. */
int awq_w4a16_gemm_cdna4_plan(int m, int n, int bits, int* mode, int* cols_main);
/* host-side query: 1 if a W4 prefill call of this shape, GIVEN its workspace, runs as pairs of 256 x 256 blocks that each sum half of K and
 * combine inside the launch (tiles that fill at most half the chip and K >= 8192: down_proj of Llama-3-8B at 1536 .. 2048 rows) -- the role
 * of the reference's split_k_iters + Semaphore, gemm_cuda.cu:546-619, without a second kernel; 0 = the tiles awq_w4a16_gemm_cdna4_plan names.
 * The two halves' fp32 sums are added once (lower K range + upper K range): another association than the unsplit kernels', same products. */
int awq_w4a16_gemm_cdna4_pair_plan(int m, int n, int k);
/* The two blocks of a pair wait for each other's partial sums with a BOUNDED spin; a block whose partner never arrives (a launch that cannot own
 * the chip: CU masks, another tenant holding every CU) writes NaN to its outputs instead of hanging the queue and counts itself in a library-owned
 * per-device word.  This query copies that word to the host (it synchronises with the device): *count = pair blocks of the CURRENT device that
 * gave up since the library was loaded; 0 on a healthy run.  The reference's Semaphore (semaphore.h:44-103) spins without a bound.  Returns
 * AWQ_OK or AWQ_ERR_LAUNCH. */
int awq_w4a16_gemm_cdna4_pair_lost(unsigned int* count);
/* host-side query: which kernel runs the 256 x 128 blocks of such a launch over n_cols weight rows.  Returns 1 = awq_gemm_v6.hip (one wave
 * per SIMD, two slabs per wave: every unsplit launch of W4 tiles at m >= 256), 0 = awq_gemm_v4n.hip unsplit (m < 256: masked single row
 * tile; W3 tiles), ks >= 2 = awq_gemm_v4n.hip with the K loop split into ks ranges (needs the workspace and no fused SiLU*mul tail). */
int awq_w4a16_gemm_cdna4_narrow_kernel(int m, int n_cols, int k, int bits, int has_workspace, int epilogue);
/* host-side query: how awq_w4a16_decode_cdna4 / awq_w4a16_mlp_gate_up_forward_cdna4 serve m <= 8 rows of an [n, k] matrix.  *kernel: 0 =
 * the LDS-DMA streaming kernel (awq_gemv_dma.hip; x staged per slab), 1 = the skinny kernel (awq_skinny_cdna4.hip; x through registers,
 * one weight pass -- where the streaming kernel's staging of m x k x 2 bytes per slab would crowd its ring out of LDS, and, since round 6, from two rows on
 * every launch that is not a wide fused pair or between one and 1.5 slabs per CU: csrc/awq_gemv_dma.hip skinny_takes).  Returns the number
 * of weight passes (1; more when the streaming kernel serves the rows in chunks), 0 if the shape is not served.  The reference's GEMV
 * handles its batch inside one pass, gemv_cuda.cu:187-208, 291-329. */
int awq_w4a16_decode_cdna4_plan(int m, int n, int k, int epilogue, int* kernel);
int awq_w4a16_gemm_cdna4(const void* x, const void* qweight_cdna4, const void* scales, const void* scaled_zeros,
                         const void* sz_packed, void* out, int m, int n, int k, int group_size, int dtype,
                         void* workspace, size_t workspace_bytes, void* stream);
int awq_w4a16_forward_cdna4(const void* x, const void* qweight_cdna4, const void* scales, const void* scaled_zeros,
                            const void* sz_packed, const void* bias, void* out, int m, int n, int k, int group_size,
                            int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* the same with the layer's sz_half side buffer (awq_pack_szh_cdna4 reported exact; NULL = awq_w4a16_forward_cdna4): prompts of >= 256 rows dequantise
 * in the f16-mantissa form the decode kernels use (fewer VALU instructions per weight beside the MFMAs) -- identical results, sz_packed still serves every
 * other row count */
int awq_w4a16_forward_cdna4_szh(const void* x, const void* qweight_cdna4, const void* scales, const void* scaled_zeros, const void* sz_packed,
                                const void* sz_half, const void* bias, void* out, int m, int n, int k, int group_size, int dtype, void* workspace,
                                size_t workspace_bytes, void* stream);

/* ---- grouped (per-expert) W4A16 GEMM for MoE layers: BASELINE.json's Mixtral-8x7B configuration.  New capability
 * (the reference has no MoE path).  Tokens are sorted by expert: expert e owns rows
 * [expert_offsets[e], expert_offsets[e+1]) of x_sorted [T, k] / out [T, n]; expert_offsets is a DEVICE int32 [E+1]
 * array (offsets[0] = 0, offsets[E] = T), so routing never synchronises the host.  Weights are stacked per expert:
 * qweight int16 [E, n/4, k], scales / scaled_zeros T [E, gpad, n].  layout 0 = v2, 1 = cdna4 (bf16). ---- */
int awq_w4a16_moe_gemm(const void* x_sorted, const void* qweight, const void* scales, const void* scaled_zeros,
                       const void* expert_offsets, void* out, int total_tokens, int num_experts, int n, int k, int gpad,
                       int group_size, int dtype, int layout, void* stream);

/* the same on cdna4 buffers with the stacked packed scales (int32 [E, n/16, k/128, 16]): decode batches (total_tokens <= 8,
 * hence at most 8 rows per expert) stream each expert's tiles once with the GEMV structure (block = (expert, slab));
 * larger batches run the grouped GEMM.  bf16 and fp16. */
int awq_w4a16_moe_forward_cdna4(const void* x_sorted, const void* qweight, const void* scales, const void* scaled_zeros,
                                const void* sz_packed, const void* expert_offsets, void* out, int total_tokens,
                                int num_experts, int n, int k, int gpad, int group_size, int dtype, void* stream);
/* the same with the experts' stacked sz_half side buffers (int32 [E, N/16, K/128, 16], every expert reported exact by awq_pack_szh_cdna4; NULL = the call
 * above): the grouped tile launch (>= 256 sorted rows) dequantises in the f16-mantissa form, identical results */
int awq_w4a16_moe_forward_cdna4_szh(const void* x_sorted, const void* qweight, const void* scales, const void* scaled_zeros, const void* sz_packed,
                                    const void* sz_half, const void* expert_offsets, void* out, int total_tokens, int num_experts, int n, int k,
                                    int gpad, int group_size, int dtype, void* stream);

/* out = T(T(silu(gate)) * up) elementwise over `count` values (count % 8 == 0): the activation step between the projections where it is not
 * fused into a kernel epilogue (tinychat/modules/fused_mlp.py:79-82: c = F.silu(gate_output) * up_output, every op rounded to T). */
int awq_silu_mul(const void* gate, const void* up, void* out, size_t count, int dtype, void* stream);
/* The expert MLP's first half in ONE grouped launch: every expert's w1 (gate) and w3 (up) rows interleaved 8 + 8 inside each 16-row slab
 * (n2 = 2 x ffn rows per expert: qweight int16 [E, n2/4, k], sz_packed int32 [E, n2/16, k/128, 16], scales / scaled_zeros T [E, gpad, n2] of
 * the same interleaved rows), out[T, n2/2] = T(T(silu(x . W1^T)) * T(x . W3^T)) -- the dense pair's epilogue (tinychat/modules/
 * fused_mlp.py:36-83) generalised to sorted tokens; the reference has no MoE path.  >= 256 sorted rows: fused into the grouped tile's
 * epilogue, the [T, n2] intermediate is never written (scratch may be NULL).  Fewer rows: the grouped GEMV / skinny kernels write the pair's
 * product to `scratch` (>= total_tokens * n2 * 2 bytes, 16-byte aligned; AWQ_ERR_WORKSPACE otherwise) and the SiLU * mul tail runs as its own
 * launch.  scales / scaled_zeros are only read by the fallback kernels of that small-batch path. */
int awq_w4a16_moe_mlp_gate_up_cdna4(const void* x_sorted, const void* qweight_interleaved, const void* scales, const void* scaled_zeros,
                                    const void* sz_packed, const void* expert_offsets, void* out, void* scratch, size_t scratch_bytes,
                                    int total_tokens, int num_experts, int n2, int k, int gpad, int group_size, int dtype, void* stream);
int awq_w4a16_moe_mlp_gate_up_cdna4_szh(const void* x_sorted, const void* qweight_interleaved, const void* scales, const void* scaled_zeros,
                                        const void* sz_packed, const void* sz_half, const void* expert_offsets, void* out, void* scratch,
                                        size_t scratch_bytes, int total_tokens, int num_experts, int n2, int k, int gpad, int group_size, int dtype,
                                        void* stream);

/* ---- W3A16 ("w3c" tiles): BASELINE.json's INT3 configuration.  The reference has NO packed 3-bit format
 * (awq/quantize/qmodule.py:82-83 raises for w_bit != 4; INT3 exists only as pseudo-quantisation,
 * awq/quantize/quantizer.py:61-103 with n_bit = 3), so the format is this repository's: per 16-row x 128-k tile
 * 64 lanes x 3 words (768 B) = the cdna4 W4 tile of the same integers (0..7) with its fourth word folded into the
 * free bit 3 of every nibble of the other three.  qweight_w3 is int16 [N/4, 3K/4] (N*K*3/8 bytes); scales /
 * scaled_zeros / sz_packed keep the W4 contract.  bf16 and fp16, n % 16 == 0, k % 128 == 0. ---- */
int awq_pack_w3(const void* q_u8 /* u8 [n, k], values 0..7 */, void* qweight_w3, int n, int k, void* stream);
int awq_unpack_w3(const void* qweight_w3, void* out_u8, int n, int k, void* stream);
int awq_dequant_w3(const void* qweight_w3, const void* scales, const void* scaled_zeros, void* out, int n, int k,
                   int group_size, int dtype, void* stream);
/* WQLinear.forward for w_bit = 3.  Every kernel reads the 3-bit tiles natively: m <= 8 streams them through the decode GEMV;
 * larger m runs the prefill tile kernels whose weight producer loads three words per lane and rebuilds the fourth (no expanded
 * copy).  `workspace` is OPTIONAL (NULL / 0 is always accepted): awq_w3a16_forward_workspace_bytes is the fp32 split-K
 * scratch that lets short prompts on narrow projections fill the chip, 0 for m <= 8 and for launches that already do. */
size_t awq_w3a16_forward_workspace_bytes(int m, int n, int k);
int awq_w3a16_forward(const void* x, const void* qweight_w3, const void* scales, const void* scaled_zeros,
                      const void* sz_packed, const void* bias, void* out, int m, int n, int k, int group_size, int dtype,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- Tensor-parallel row split (K-sharded WQLinear; SURVEY.md 8(e): new capability, the reference has no multi-GPU path for
 * awq/quantize/qmodule.py:201-224).  Rank r holds the k range [k0, k1) of the cdna4 buffers and computes
 *     out_f32[m, n] = x[:, k0:k1] . W[:, k0:k1]^T        fp32 accumulators, UNROUNDED, no bias
 * with the same kernels as the unsharded forward (decode streaming / skinny / prefill tiles by m); the ranks' partials are summed in fp32
 * (awq_oneshot_allreduce_f32 for the latency-class messages, RCCL on the float tensor above that) and rounded to T ONCE, then the bias
 * is added in T -- what the single-device kernel does with its one accumulator, so the sharded output stays within the 1e-3 budget of
 * the single-device result in bf16 as well (T-rounded partials: 2.6-2.9e-3).  sz_half is optional (NULL = read sz_packed). ---- */
int awq_w4a16_partial_cdna4(const void* x, const void* qweight_cdna4, const void* sz_packed, const void* sz_half, float* out_f32, int m,
                            int n, int k, int group_size, int dtype, void* stream);
/* QuantLlamaMLP's gate / up pair + SiLU * mul (tinychat/modules/fused_mlp.py:33-83) for 3-bit projections: qweight_w3_interleaved holds the
 * two projections' INTEGER rows interleaved 8 + 8 per 16-row slab (gate rows 8 j .. 8 j + 7, then the matching up rows) packed into w3c
 * tiles, sz_packed the same interleave of their scales / zeros; out[m, n2 / 2] = T(T(silu(T(gate))) * T(up)).  m <= 8: the register-ring
 * decode kernel pairs the rows in its epilogue; 9 .. 64 rows: the skinny kernel's paired epilogue; above that the prefill tiles' fused tail.  workspace: optional split-K scratch (NULL / 0 ok). */
size_t awq_w3a16_mlp_gate_up_forward_workspace_bytes(int m, int n2, int k);
int awq_w3a16_mlp_gate_up_forward(const void* x, const void* qweight_w3_interleaved, const void* sz_packed, void* out, int m, int n2, int k,
                                  int group_size, int dtype, void* workspace, size_t workspace_bytes, void* stream);
/* The same for a 3-bit layer (w3c tiles, awq_pack_w3_from_v1; the reference's w_bit = 3 is NotImplemented at
 * awq/quantize/qmodule.py:95-96, so the sharded form is new as well): m <= 8 the register-ring decode kernel with an fp32
 * epilogue, above that the prefill tiles' fp32 epilogue. */
int awq_w3a16_partial(const void* x, const void* qweight_w3, const void* sz_packed, float* out_f32, int m, int n, int k,
                      int group_size, int dtype, void* stream);
/* out[m, n] = T(in_f32[m, n]) (+ bias[n] in T; may be NULL): the single rounding after an RCCL sum of the partials.  n % 8 == 0. */
int awq_round_bias_f32(const float* in_f32, const void* bias, void* out, int m, int n, int dtype, void* stream);

/* ---- One-shot all-reduce (sum, fp32 accumulation in rank order, one rounding to T) for the latency-class messages of
 * K-sharded decode: M * N * 2 bytes = 8 .. 16 KiB after o_proj / down_proj (the reference has no multi-GPU path).
 * Every rank owns one exchange buffer (awq_oneshot_alloc: fine-grained device memory, awq_oneshot_buffer_bytes(world, max_bytes)
 * bytes), exports it with awq_oneshot_ipc_export (a 64-byte hipIpc handle) and opens its peers' with awq_oneshot_ipc_open;
 * peer_buffers[q] is rank q's buffer as mapped in THIS process (peer_buffers[rank] = the local allocation).  A call stores
 * `in` (count elements, count % 8 == 0, count * 2 <= max_bytes) into every rank's buffer over xGMI, raises one flag per peer and
 * reduces locally when the `world` flags of this round have arrived.  `round` must be 1, 2, 3, ... in call order and equal on
 * all ranks, or 0 on every call of a communicator: the epoch then lives in the rank's own buffer and advances by one per call, which
 * is what a captured and replayed hipGraph needs (its kernel arguments are frozen); the two modes must not be mixed on one buffer set.  *status_dev (optional device int) is set to 1 if a peer's flag did not arrive within the spin bound (the kernel
 * returns instead of hanging the queue).  world <= 8.  Messages above max_bytes belong to RCCL (bandwidth-bound). ---- */
size_t awq_oneshot_buffer_bytes(int world, int max_bytes);
int awq_oneshot_alloc(void** buffer, int world, int max_bytes);
int awq_oneshot_free(void* buffer);
int awq_oneshot_ipc_export(void* buffer, void* handle64);
int awq_oneshot_ipc_open(const void* handle64, void** buffer);
int awq_oneshot_ipc_close(void* buffer);
int awq_oneshot_allreduce(void* const* peer_buffers, const void* in, void* out, int count, int dtype, int rank, int world,
                          unsigned round, int max_bytes, int* status_dev, void* stream);
/* The same exchange on fp32 partials (count floats, count * 4 <= max_bytes): out = T(sum over ranks, fp32, rank order) (+ bias in T, bias_n =
 * its length, count % bias_n == 0; NULL = none).  A communicator serves both forms; a call after *status_dev was set poisons its output. */
int awq_oneshot_allreduce_f32(void* const* peer_buffers, const float* in_f32, const void* bias, int bias_n, void* out, int count, int dtype,
                              int rank, int world, unsigned round, int max_bytes, int* status_dev, void* stream);
/* Spin bound of the flag wait (polls of ~1 us; default 40 M: a peer may be tens of seconds late before the round is declared lost). */
int awq_oneshot_set_spin_limit(unsigned spins);
/* Single-GPU self-test of the device protocol: ONE launch of `world` co-resident blocks, block r playing rank r against the
 * others on `world` exchange buffers of the same device (in_all / out_all: [world][count]).  Tests only. */
int awq_oneshot_allreduce_selftest(void* const* peer_buffers, const void* in_all, void* out_all, int count, int dtype, int world,
                                   unsigned round, int max_bytes, int* status_dev, void* stream);

int awq_oneshot_allreduce_f32_selftest(void* const* peer_buffers, const float* in_all_f32, const void* bias, int bias_n, void* out_all, int count,
                                       int dtype, int world, unsigned round, int max_bytes, int* status_dev, void* stream);

/* Tuning hook for tests, experiments and benchmarks (not part of the reference surface): integer knobs that force one of the
 * shipped code paths ("gemm_variant", "gemm_splitk", "gemv_dma", "gemvd_waves", ...) so that tests can cover each of them; 0
 * restores the default heuristic.  A default process cannot reach it: unless AWQ_TUNING=1 is set in the environment every
 * call returns AWQ_ERR_SHAPE and changes nothing.  Timing probes and experiment-only kernel instantiations exist only in
 * builds made with AWQ_PROBES=1.  Returns AWQ_OK, or AWQ_ERR_SHAPE for an unknown key.  Process-global, not thread-safe. */
int awq_tune_set(const char* key, int value);

#ifdef __cplusplus
}
#endif
#endif /* AWQ_CDNA4_H_ */
