// Python module `awq_inference_engine` for PyTorch-ROCm: the reference's two hot-path exports
//     gemv_forward_cuda_new(in_feats, kernel, scaling_factors, zeros, m, n, k, group_size) -> Tensor
//     gemm_forward_cuda_new(in_feats, kernel, scales, zeros) -> Tensor
// (awq/kernels/csrc/pybind.cpp:22-23; gemv_cuda.cu:245-338; gemm_cuda.cu:1126-1236) re-implemented as a
// thin shim over the C ABI in include/awq_cdna4.h.  Same names, argument meaning, output
// allocation and error behaviour (RuntimeError); plus what the reference lacks: device guard,
// launch on the CURRENT stream, contiguity / shape checks.
// Torch is plumbing here (tensor memory, streams); all compute lives behind the C ABI.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include <stdexcept>
#include <string>

#include "../../include/awq_cdna4.h"

namespace {

int dtype_code(const torch::Tensor& t) {
  if (t.scalar_type() == at::kHalf) return AWQ_F16;
  if (t.scalar_type() == at::kBFloat16) return AWQ_BF16;
  TORCH_CHECK(false, "awq_inference_engine: only float16 / bfloat16 activations are supported, got ", t.scalar_type());
}

void raise_on(int st) {
  if (st == AWQ_OK) return;
  std::string msg = awq_status_string(st);
  if (st == AWQ_ERR_LAUNCH) msg += std::string(" ") + awq_last_hip_error();
  if (st == AWQ_ERR_BATCH || st == AWQ_ERR_GROUP) throw std::runtime_error(msg + "\n");  // gemv_cuda.cu:329,334
  TORCH_CHECK(false, "awq_inference_engine: ", msg);
}

void check_inputs(const torch::Tensor& x, const torch::Tensor& kernel, const torch::Tensor& scales,
                  const torch::Tensor& zeros) {
  TORCH_CHECK(x.is_cuda() && kernel.is_cuda() && scales.is_cuda() && zeros.is_cuda(),
              "awq_inference_engine: all tensors must live on the GPU (this build has no CPU path)");
  TORCH_CHECK(scales.scalar_type() == x.scalar_type());  // gemv_cuda.cu:260 / gemm_cuda.cu:1145
  TORCH_CHECK(zeros.scalar_type() == x.scalar_type());   // gemv_cuda.cu:261 / gemm_cuda.cu:1146
  TORCH_CHECK(kernel.scalar_type() == at::kShort, "qweight must be int16 [N/4, K]");
  TORCH_CHECK(x.is_contiguous() && kernel.is_contiguous() && scales.is_contiguous() && zeros.is_contiguous(),
              "awq_inference_engine: tensors must be contiguous");
  TORCH_CHECK(kernel.device() == x.device() && scales.device() == x.device() && zeros.device() == x.device(),
              "awq_inference_engine: tensors must be on the same device");
}

torch::Tensor gemv_forward_cuda_new(torch::Tensor in_feats, torch::Tensor kernel, torch::Tensor scaling_factors,
                                    torch::Tensor zeros, int m, int n, int k, int group_size) {
  check_inputs(in_feats, kernel, scaling_factors, zeros);
  if (group_size != 128) raise_on(AWQ_ERR_GROUP);
  if (m < 1 || m > 7) raise_on(AWQ_ERR_BATCH);  // the reference's switch(m) covers 1..7 (gemv_cuda.cu:291-329)
  TORCH_CHECK(in_feats.size(-1) == k && in_feats.numel() == (int64_t)m * k, "in_feats must be [m, k]");
  TORCH_CHECK(kernel.numel() == (int64_t)n / 4 * k, "kernel must be int16 [n/4, k]");
  TORCH_CHECK(scaling_factors.size(-1) == n && zeros.size(-1) == n && scaling_factors.size(0) * 128 >= k &&
                  zeros.size(0) * 128 >= k,
              "scales / scaled_zeros must be [Gpad, n]");
  std::vector<int64_t> shape = in_feats.sizes().vec();
  shape.back() = n;
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(in_feats.device());
  at::Tensor out = torch::empty(shape, in_feats.options());
  auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  raise_on(awq_w4a16_gemv(in_feats.data_ptr(), kernel.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(),
                          out.data_ptr(), m, n, k, group_size, dtype_code(in_feats), (void*)stream));
  return out;
}

torch::Tensor gemm_forward_cuda_new(torch::Tensor in_feats, torch::Tensor kernel, torch::Tensor scales,
                                    torch::Tensor zeros) {
  check_inputs(in_feats, kernel, scales, zeros);
  const int64_t n = kernel.size(0) * 4;  // gemm_cuda.cu:1133
  const int64_t k = in_feats.size(-1);   // gemm_cuda.cu:1135
  TORCH_CHECK(k > 0 && in_feats.numel() % k == 0);
  const int64_t m = in_feats.numel() / k;  // gemm_cuda.cu:1134
  TORCH_CHECK(kernel.numel() == n / 4 * k, "kernel must be int16 [n/4, k]");
  TORCH_CHECK(scales.size(-1) == n && zeros.size(-1) == n && scales.size(0) * 128 >= k && zeros.size(0) * 128 >= k,
              "scales / scaled_zeros must be [Gpad, n]");
  std::vector<int64_t> shape = in_feats.sizes().vec();
  shape.back() = n;
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(in_feats.device());
  at::Tensor out = torch::empty(shape, in_feats.options());
  if (m == 0) return out;
  auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  const size_t ws_bytes = awq_w4a16_gemm_workspace_bytes((int)m, (int)n, (int)k);
  at::Tensor ws;
  void* wsp = nullptr;
  if (ws_bytes) {
    ws = torch::empty({(int64_t)ws_bytes}, in_feats.options().dtype(at::kByte));
    wsp = ws.data_ptr();
  }
  raise_on(awq_w4a16_gemm(in_feats.data_ptr(), kernel.data_ptr(), scales.data_ptr(), zeros.data_ptr(), out.data_ptr(),
                          (int)m, (int)n, (int)k, 128, dtype_code(in_feats), wsp, ws_bytes, (void*)stream));
  return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "MI355X-native W4A16 kernels behind llm-awq's awq_inference_engine API (hot path only)";
  m.def("gemm_forward_cuda_new", &gemm_forward_cuda_new, "New quantized GEMM kernel.");
  m.def("gemv_forward_cuda_new", &gemv_forward_cuda_new, "New quantized GEMV kernel.");
  m.def("abi_version", []() { return awq_abi_version(); });
}
