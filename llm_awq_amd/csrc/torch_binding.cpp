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

#include <cstdlib>
#include <mutex>
#include <tuple>
#include <stdexcept>
#include <string>
#include <unordered_map>

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

// ---------------------------------------------------------------------------------------------------------------
// Lazy v2 -> cdna4 cache.  tinychat hands the extension RAW reference-layout buffers (fused_mlp.py:40-77 passes
// gate_proj_qweight etc. directly, make_quant_attn concatenates v2 buffers after load), so the drop-in entry points
// cannot assume the repacker ran.  The first call with a given (qweight, scales, zeros) triple repacks
// them once on the GPU (awq_repack_v2_to_cdna4 + awq_pack_sz_cdna4, tens of microseconds) and later calls run the
// cdna4 kernels.  An entry is tied to the IDENTITY of the three tensors (weak TensorImpl references, so a freed
// tensor whose address is re-used can never hit) and to their version counters (an in-place update re-packs).
// Cost: a second packed copy of the weights (N*K/2 + N*K/32 bytes).  AWQ_CDNA4_AUTOCACHE=0 disables it.  While the stream is
// being captured into a hipGraph nothing is built (no allocations inside a capture): entries made by a warm-up call are used.
// ---------------------------------------------------------------------------------------------------------------
// One entry per qweight tensor: the permuted weights (the expensive part: N*K/2 bytes) are tied to the qweight's identity and
// version ONLY.  The packed scale buffers (3 % of that) hang off the entry in two slots keyed by the (scales, zeros) pair they
// were built from: tinychat's QuantLlamaMLP passes the module's scaled_zeros on its decode branch and a FRESH temporary
// (`scaled_zeros - 8 * scales`, fused_mlp.py:69,76) on every prefill call -- the temporary re-packs 1-2 MB of scales into the
// least recently used slot, never the weights, and the decode pair keeps hitting its own slot.
inline uint32_t tensor_version(const torch::Tensor& t) { return t.is_inference() ? 0u : (uint32_t)t._version(); }
struct SzSlot {
  c10::weak_intrusive_ptr<c10::TensorImpl> s, z;
  uint32_t vs = 0, vz = 0;
  at::Tensor szp, szh;  // sz_packed (T) and sz_half (decode; undefined when the layer's scales are not f16-exact)
  uint64_t stamp = 0;
  hipEvent_t built = nullptr;  // recorded behind the pack kernels; a hit from another stream waits for it until it has completed
  hipStream_t build_stream = nullptr;
  bool settled = false;
  bool expired() const { return !szp.defined() || s.expired() || z.expired(); }
  SzSlot() : s(c10::weak_intrusive_ptr<c10::TensorImpl>(c10::intrusive_ptr<c10::TensorImpl>())), z(s) {}
};
struct CacheEntry {
  c10::weak_intrusive_ptr<c10::TensorImpl> w;
  uint32_t vw;
  at::Tensor c4;
  SzSlot slot[2];
  hipEvent_t built = nullptr;  // recorded on the building stream; other streams wait for it until it has completed
  hipStream_t build_stream = nullptr;
  bool settled = false;
  // AWQ_CDNA4_INPLACE=1: the caller's qweight storage is converted where it lies (no second copy).  The entry does NOT own that storage -- a cache must
  // never extend the lifetime of the model's weights (`del model` has to give the memory back) -- it keeps a WEAK reference to it, the
  // view's geometry, and the version counter the module's tensor shares with its `.detach()` / `.data` / view aliases
  bool inplace = false;
  c10::weak_intrusive_ptr<c10::StorageImpl> ws{c10::intrusive_ptr<c10::StorageImpl>()};
  int64_t ws_offset = 0, ws_n4 = 0, ws_k = 0;
  c10::Device ws_dev{c10::DeviceType::CPU};
  bool has_vc = false;
  c10::VariableVersion vc{c10::VariableVersion::DISABLED};
  uint32_t vc4 = 0;  // the counter's value at conversion: a later value means the caller wrote fresh v2 bytes over the converted ones (copy_, load_state_dict)
  bool inplace_alive() const { return inplace && !ws.expired(); }
  // the converted bytes are still what the cache put there (false: storage gone, or overwritten since -- nothing to restore, nothing to follow)
  bool inplace_intact() const { return inplace_alive() && (!has_vc || (uint32_t)vc.current_version() == vc4); }
  bool holds(const torch::Tensor& t) const {  // is `t` a tensor over the converted storage?
    auto l = ws.lock();
    return l && l.get() == t.storage().unsafeGetStorageImpl();
  }
  // a tensor over the converted bytes (keeps the storage alive for the call); undefined if the storage is gone
  at::Tensor inplace_tensor() const {
    auto l = ws.lock();
    if (!l) return at::Tensor();
    at::Tensor t = at::empty({0}, at::TensorOptions().dtype(at::kShort).device(ws_dev));
    t.set_(c10::Storage(std::move(l)), ws_offset, {ws_n4, ws_k}, {ws_k, 1});
    return t;
  }
  explicit CacheEntry(const torch::Tensor& tw) : w(tw.getIntrusivePtr()), vw(tensor_version(tw)) {}
};
std::mutex g_cache_mu;
std::unordered_map<const void*, CacheEntry> g_cache;
int g_cache_enabled = -1, g_cache_inplace = -1;
int64_t g_cache_hits = 0, g_cache_builds = 0, g_cache_sz_builds = 0, g_cache_bytes = 0, g_cache_max_bytes = -1;
uint64_t g_cache_clock = 0;

bool cache_enabled() {
  if (g_cache_enabled < 0) {
    const char* e = std::getenv("AWQ_CDNA4_AUTOCACHE");
    g_cache_enabled = (e && e[0] == '0') ? 0 : 1;
  }
  if (g_cache_max_bytes < 0) {  // AWQ_CDNA4_AUTOCACHE_MAX_GB bounds the second copy of the weights (0 / unset = unbounded)
    const char* e = std::getenv("AWQ_CDNA4_AUTOCACHE_MAX_GB");
    g_cache_max_bytes = e ? (int64_t)(std::atof(e) * (double)(1ull << 30)) : 0;
  }
  if (g_cache_inplace < 0) {
    // AWQ_CDNA4_INPLACE=1: the first call through the reference entry points converts the qweight WHERE IT LIES (via one layer-sized
    // temporary that is freed at once) instead of keeping a permuted second copy: no extra weight memory (35 GB on Llama-3-70B), but
    // the module's `qweight` then holds the cdna4 interleave -- save checkpoints with llm_awq_amd.repacker / cdna4_restore first.
    const char* e = std::getenv("AWQ_CDNA4_INPLACE");
    g_cache_inplace = (e && e[0] == '1') ? 1 : 0;
  }
  return g_cache_enabled == 1;
}

// an in-place entry whose qweight is still alive: put the reference (v2) interleave back before the entry is forgotten, or the next
// call would take the permuted bytes for v2 data and permute them again
void restore_inplace(CacheEntry& e) {
  if (!e.inplace) return;
  at::Tensor t = e.inplace_intact() ? e.inplace_tensor() : at::Tensor();
  e.inplace = false;
  if (!t.defined()) return;  // storage gone, or overwritten with fresh v2 bytes since the conversion: running the cdna4 -> v2 permutation over them would corrupt them
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(t.device());
  hipStream_t st = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
  if (e.built && st != e.build_stream) (void)hipStreamWaitEvent(st, e.built, 0);
  at::Tensor tmp = torch::empty_like(t);
  const int n = (int)t.size(0) * 4, k = (int)t.size(1);
  if (awq_repack_cdna4_to_v2(t.data_ptr(), tmp.data_ptr(), n, k, (void*)st) == AWQ_OK)
    (void)hipMemcpyAsync(t.data_ptr(), tmp.data_ptr(), t.nbytes(), hipMemcpyDeviceToDevice, st);
}

void drop_entry(std::unordered_map<const void*, CacheEntry>::iterator it) {
  if (it->second.c4.defined() && !it->second.inplace) g_cache_bytes -= (int64_t)it->second.c4.nbytes();
  if (it->second.built) (void)hipEventDestroy(it->second.built);
  for (SzSlot& sl : it->second.slot)
    if (sl.built) (void)hipEventDestroy(sl.built);
  g_cache.erase(it);
}

// is this exact buffer (data pointer incl. storage offset) currently held converted in place?  (g_cache_mu held by the caller)
bool inplace_converted_locked(const torch::Tensor& kernel) {
  auto it = g_cache.find(kernel.data_ptr());
  return it != g_cache.end() && it->second.inplace_intact() && it->second.holds(kernel);
}
// the reference-layout kernels must never read a buffer the cache converted where it lies: callers that cannot be served by the cdna4 path
// (cache switched off, fp32 scales, a failed scale pack) get an error that names the way out instead of silently wrong products
void refuse_if_converted(const torch::Tensor& kernel, const char* why) {
  TORCH_CHECK(!inplace_converted_locked(kernel), "awq_inference_engine: this qweight was converted to the cdna4 interleave IN PLACE (AWQ_CDNA4_INPLACE=1) and ",
              why, "; call awq_inference_engine.cdna4_restore(qweight) first");
}

// returns true and fills (c4, szp, szh) when the cdna4 kernels can serve this call (szh stays undefined if the scales are not f16-exact)
bool cdna4_view(const torch::Tensor& kernel, const torch::Tensor& scales, const torch::Tensor& zeros, int64_t n, int64_t k,
                hipStream_t stream, at::Tensor& c4, at::Tensor& szp, at::Tensor* szh = nullptr) {
  if (!cache_enabled() || kernel.scalar_type() != at::kShort ||
      (scales.scalar_type() != at::kBFloat16 && scales.scalar_type() != at::kHalf)) {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    refuse_if_converted(kernel, "this call cannot run on the cdna4 kernels (cache disabled, or scales that are neither fp16 nor bf16)");
    return false;
  }
  if (n % 16 != 0 || k % 128 != 0 || kernel.numel() != n / 4 * k) return false;
  // inside a hipGraph capture the weights are never re-packed (an entry built by an earlier warm-up call is used, a first call
  // falls back to the reference-layout kernels); the small scale buffers may be (torch's allocator is capture-safe)
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
  std::lock_guard<std::mutex> lock(g_cache_mu);
  const void* key = kernel.data_ptr();
  auto it = g_cache.find(key);
  if (it != g_cache.end()) {
    auto lw = it->second.w.lock();
    // an in-place entry whose storage was overwritten since the conversion (the shared version counter moved) is stale whichever alias asks
    const bool gone = it->second.inplace && !it->second.inplace_alive();  // its storage was freed and the address handed to another tensor
    const bool overwritten = it->second.inplace && it->second.inplace_alive() && !it->second.inplace_intact();
    if (gone || overwritten || lw.get() != kernel.unsafeGetTensorImpl() || it->second.vw != tensor_version(kernel)) {  // address re-used or edited in place
      CacheEntry& old = it->second;
      const bool same_bytes = old.inplace_intact() && old.holds(kernel);
      if (same_bytes && lw.get() != kernel.unsafeGetTensorImpl()) {
        // ANOTHER tensor over the converted bytes (a view, .detach(), .data, load_state_dict(assign=True) of an alias, a compile wrapper):
        // the storage already holds the cdna4 interleave -- converting "again" would permute it twice.  The entry follows the caller
        // (host-side only: legal inside a capture); a tensor whose version moved under the SAME impl was overwritten with new v2 data.
        old.w = c10::weak_intrusive_ptr<c10::TensorImpl>(kernel.getIntrusivePtr());
        old.vw = tensor_version(kernel);
      } else {
        TORCH_CHECK(!(capturing && (same_bytes || overwritten)), "awq_inference_engine: an in-place converted qweight was modified during a graph capture");
        if (capturing) return false;
        // (same impl, new version: the caller wrote fresh v2 bytes over the buffer -- nothing to restore; a foreign tensor at a re-used
        // address: the old entry's storage is not ours to touch either)
        drop_entry(it);
        it = g_cache.end();
      }
    }
  }
  if (it == g_cache.end()) {
    if (capturing) return false;
    // drop entries whose qweight died or was re-pointed (keeps the map from growing when models are reloaded)
    for (auto i2 = g_cache.begin(); i2 != g_cache.end();) {
      auto cur = i2++;
      auto lw = cur->second.w.lock();
      // (the key is the tensor's data pointer INCLUDING its storage offset: sharded / flattened parameter buffers are views)
      if (cur->second.inplace) {
        // an in-place entry lives exactly as long as the storage it converted: the tensor it last followed may have been a temporary alias, and
        // forgetting the entry while the module's own tensor is alive would have the next call permute the bytes a second time
        if (cur->second.inplace_alive()) continue;
        drop_entry(cur);  // the weights are gone (the cache never held them): nothing to restore
        continue;
      }
      if (!lw || lw->data() != cur->first) drop_entry(cur);
    }
    const int64_t need = (int64_t)kernel.nbytes();
    if (g_cache_inplace != 1 && g_cache_max_bytes > 0 && g_cache_bytes + need > g_cache_max_bytes) return false;  // over budget: reference-layout kernels
    CacheEntry e(kernel);
    if (g_cache_inplace == 1) {
      at::Tensor tmp = torch::empty_like(kernel);  // stream-ordered allocation: returned to the pool when it goes out of scope
      if (awq_repack_v2_to_cdna4(kernel.data_ptr(), tmp.data_ptr(), (int)n, (int)k, (void*)stream) != AWQ_OK) return false;
      if (hipMemcpyAsync(kernel.data_ptr(), tmp.data_ptr(), kernel.nbytes(), hipMemcpyDeviceToDevice, stream) != hipSuccess) return false;
      // (a raw copy: the tensor's version counter does not move, so the entry stays valid)
      e.inplace = true;
      e.ws = c10::weak_intrusive_ptr<c10::StorageImpl>(c10::intrusive_ptr<c10::StorageImpl>::reclaim_copy(kernel.storage().unsafeGetStorageImpl()));
      e.ws_offset = kernel.storage_offset();
      e.ws_n4 = n / 4;
      e.ws_k = k;
      e.ws_dev = kernel.device();
      e.has_vc = !kernel.is_inference();
      if (e.has_vc) e.vc = kernel.unsafeGetTensorImpl()->version_counter();
      e.vc4 = tensor_version(kernel);
    } else {
      e.c4 = torch::empty_like(kernel);
      if (awq_repack_v2_to_cdna4(kernel.data_ptr(), e.c4.data_ptr(), (int)n, (int)k, (void*)stream) != AWQ_OK) return false;
      g_cache_bytes += need;
    }
    if (hipEventCreateWithFlags(&e.built, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(e.built, stream);
    e.build_stream = stream;
    ++g_cache_builds;
    it = g_cache.emplace(key, std::move(e)).first;
  }
  CacheEntry& e = it->second;
  if (!e.settled && e.built) {
    if (stream != e.build_stream && !capturing) (void)hipStreamWaitEvent(stream, e.built, 0);  // the copy was made on another stream
    if (!capturing && hipEventQuery(e.built) == hipSuccess) e.settled = true;
  }
  // ---- the scale side buffers ----
  SzSlot* hit = nullptr;
  for (SzSlot& sl : e.slot) {
    auto ls = sl.s.lock(), lz = sl.z.lock();
    if (sl.szp.defined() && ls.get() == scales.unsafeGetTensorImpl() && lz.get() == zeros.unsafeGetTensorImpl() &&
        sl.vs == tensor_version(scales) && sl.vz == tensor_version(zeros))
      hit = &sl;
  }
  if (hit == nullptr) {
    // victim: a slot whose (scales, zeros) pair is gone (a prefill call's `scaled_zeros - 8 * scales` temporary) before a live one --
    // two temporaries in a row must not evict the persistent decode pair; among equals the least recently used
    SzSlot* victim = &e.slot[0];
    if (e.slot[0].expired() != e.slot[1].expired()) victim = e.slot[0].expired() ? &e.slot[0] : &e.slot[1];
    else if (e.slot[1].stamp < e.slot[0].stamp) victim = &e.slot[1];
    SzSlot& sl = *victim;
    at::Tensor nszp = torch::empty({n / 16, k / 128, 16}, scales.options().dtype(at::kInt));
    if (awq_pack_sz_cdna4(scales.data_ptr(), zeros.data_ptr(), nszp.data_ptr(), (int)n, (int)k, (void*)stream) != AWQ_OK) {
      refuse_if_converted(kernel, "its scales could not be packed for the cdna4 kernels");
      return false;
    }
    if (capturing) {
      // built while a graph is being captured: the pack kernel is only RECORDED, so the buffer must not be published (an eager call
      // could hit it before the first replay); it lives for this call's launch alone, decode uses the T-typed sz_packed
      c4 = e.inplace ? kernel : e.c4;
      szp = nszp;
      if (szh) *szh = at::Tensor();
      return true;
    }
    at::Tensor nszh;
    if (!capturing) {  // the exactness flag is read back once: not inside a capture (decode then uses sz_packed)
      at::Tensor h = torch::empty({n / 16, k / 128, 16}, scales.options().dtype(at::kInt));
      at::Tensor flag = torch::zeros({1}, scales.options().dtype(at::kInt));
      if (awq_pack_szh_cdna4(scales.data_ptr(), zeros.data_ptr(), h.data_ptr(), (int*)flag.data_ptr(), (int)n, (int)k,
                             scales.scalar_type() == at::kHalf ? AWQ_F16 : AWQ_BF16, (void*)stream) == AWQ_OK &&
          flag.item<int>() == 0)
        nszh = h;
    }
    sl.s = c10::weak_intrusive_ptr<c10::TensorImpl>(scales.getIntrusivePtr());
    sl.z = c10::weak_intrusive_ptr<c10::TensorImpl>(zeros.getIntrusivePtr());
    sl.vs = tensor_version(scales);
    sl.vz = tensor_version(zeros);
    sl.szp = nszp;
    sl.szh = nszh;
    if (!sl.built && hipEventCreateWithFlags(&sl.built, hipEventDisableTiming) != hipSuccess) sl.built = nullptr;
    if (sl.built) (void)hipEventRecord(sl.built, stream);
    sl.build_stream = stream;
    sl.settled = false;
    hit = &sl;
    ++g_cache_sz_builds;
  } else {
    ++g_cache_hits;
    if (!hit->settled && hit->built) {  // packed on another stream: order this one behind it until the event has completed
      if (stream != hit->build_stream && !capturing) (void)hipStreamWaitEvent(stream, hit->built, 0);
      if (!capturing && hipEventQuery(hit->built) == hipSuccess) hit->settled = true;
    }
  }
  hit->stamp = ++g_cache_clock;
  c4 = e.inplace ? kernel : e.c4;  // (in place: the caller's tensor IS the converted buffer)
  szp = hit->szp;
  if (szh) *szh = hit->szh;
  return true;
}

// optional split-K workspace of the cdna4 prefill path (include/awq_cdna4.h); comes from torch's caching allocator, so it is
// stream-ordered and legal inside a graph capture
static at::Tensor cdna4_workspace(const at::Tensor& like, int64_t m, int64_t n, int64_t k, void*& ptr, size_t& bytes) {
  at::Tensor ws;
  ptr = nullptr;
  bytes = awq_w4a16_forward_cdna4_workspace_bytes((int)m, (int)n, (int)k);
  if (bytes) {
    ws = torch::empty({(int64_t)bytes}, like.options().dtype(at::kByte));
    ptr = ws.data_ptr();
  }
  return ws;
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
  at::Tensor c4, szp, szh;
  if (cdna4_view(kernel, scaling_factors, zeros, n, k, stream, c4, szp, &szh)) {
    if (szh.defined())
      raise_on(awq_w4a16_decode_cdna4(in_feats.data_ptr(), c4.data_ptr(), szh.data_ptr(), nullptr, out.data_ptr(), m, n, k, group_size,
                                      dtype_code(in_feats), 0, (void*)stream));
    else
      raise_on(awq_w4a16_forward_cdna4(in_feats.data_ptr(), c4.data_ptr(), scaling_factors.data_ptr(), zeros.data_ptr(), szp.data_ptr(),
                                       nullptr, out.data_ptr(), m, n, k, group_size, dtype_code(in_feats), nullptr, 0, (void*)stream));
    return out;
  }
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
  {
    at::Tensor c4, szp, szh;
    if (cdna4_view(kernel, scales, zeros, n, k, stream, c4, szp, &szh)) {
      void* wsp;
      size_t wsb;
      at::Tensor ws = cdna4_workspace(in_feats, m, n, k, wsp, wsb);
      // (round 6: the cached sz_half side buffer -- built for the decode entry when the layer's scales are f16-exact -- serves the prompts too: the tile kernels, the
      // mid-M kernel and the 9 .. 16-row skinny launches dequantise in the f16-mantissa form, as the native modules' calls do; same bits)
      raise_on(awq_w4a16_forward_cdna4_szh(in_feats.data_ptr(), c4.data_ptr(), scales.data_ptr(), zeros.data_ptr(), szp.data_ptr(),
                                           szh.defined() ? szh.data_ptr() : nullptr, nullptr, out.data_ptr(), (int)m, (int)n, (int)k, 128, dtype_code(in_feats), wsp,
                                           wsb, (void*)stream));
      return out;
    }
  }
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

// ---- MI355X-native "cdna4" interleave (bf16): same tensors / shapes, permuted qweight + packed scales ----
torch::Tensor repack_v2_to_cdna4(torch::Tensor kernel) {
  TORCH_CHECK(kernel.is_cuda() && kernel.is_contiguous() && kernel.scalar_type() == at::kShort && kernel.dim() == 2);
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(kernel.device());
  at::Tensor out = torch::empty_like(kernel);
  raise_on(awq_repack_v2_to_cdna4(kernel.data_ptr(), out.data_ptr(), (int)kernel.size(0) * 4, (int)kernel.size(1),
                                  (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()));
  return out;
}

torch::Tensor repack_cdna4_to_v2(torch::Tensor kernel) {
  TORCH_CHECK(kernel.is_cuda() && kernel.is_contiguous() && kernel.scalar_type() == at::kShort && kernel.dim() == 2);
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(kernel.device());
  at::Tensor out = torch::empty_like(kernel);
  raise_on(awq_repack_cdna4_to_v2(kernel.data_ptr(), out.data_ptr(), (int)kernel.size(0) * 4, (int)kernel.size(1),
                                  (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()));
  return out;
}

torch::Tensor pack_sz_cdna4(torch::Tensor scales, torch::Tensor zeros, int k) {
  TORCH_CHECK(scales.is_cuda() && zeros.is_cuda() && scales.is_contiguous() && zeros.is_contiguous());
  TORCH_CHECK(scales.scalar_type() == zeros.scalar_type() && scales.sizes() == zeros.sizes() && scales.dim() == 2);
  const int64_t n = scales.size(1);
  TORCH_CHECK(n % 16 == 0 && k % 128 == 0 && scales.size(0) * 128 >= k);
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(scales.device());
  at::Tensor out = torch::empty({n / 16, k / 128, 16}, scales.options().dtype(at::kInt));
  raise_on(awq_pack_sz_cdna4(scales.data_ptr(), zeros.data_ptr(), out.data_ptr(), (int)n, k,
                             (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()));
  return out;
}

// WQLinear.forward on cdna4 buffers: any number of rows (<= 16 -> GEMV kernel), bias optional
torch::Tensor forward_cdna4(torch::Tensor in_feats, torch::Tensor kernel, torch::Tensor scales, torch::Tensor zeros,
                            torch::Tensor sz_packed, c10::optional<torch::Tensor> bias, c10::optional<torch::Tensor> sz_half) {
  check_inputs(in_feats, kernel, scales, zeros);
  TORCH_CHECK(in_feats.scalar_type() == at::kBFloat16 || in_feats.scalar_type() == at::kHalf,
              "the cdna4 interleave is defined for bfloat16 / float16");
  TORCH_CHECK(sz_packed.is_cuda() && sz_packed.is_contiguous() && sz_packed.scalar_type() == at::kInt);
  const int64_t n = kernel.size(0) * 4, k = in_feats.size(-1);
  TORCH_CHECK(k > 0 && in_feats.numel() % k == 0 && kernel.numel() == n / 4 * k);
  TORCH_CHECK(sz_packed.numel() == n * (k / 128), "sz_packed must be int32 [n/16, k/128, 16]");
  const int64_t m = in_feats.numel() / k;
  std::vector<int64_t> shape = in_feats.sizes().vec();
  shape.back() = n;
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(in_feats.device());
  at::Tensor out = torch::empty(shape, in_feats.options());
  if (m == 0) return out;
  const void* bp = nullptr;
  if (bias.has_value() && bias->defined()) {
    TORCH_CHECK(bias->is_cuda() && bias->is_contiguous() && bias->scalar_type() == in_feats.scalar_type() && bias->numel() == n);
    bp = bias->data_ptr();
  }
  void* wsp;
  size_t wsb;
  at::Tensor ws = cdna4_workspace(in_feats, m, n, k, wsp, wsb);
  const void* hp = nullptr;  // the layer's sz_half side buffer (pack_szh_cdna4 reported exact): prompts dequantise in the f16-mantissa form
  if (sz_half.has_value() && sz_half->defined()) {
    TORCH_CHECK(sz_half->is_cuda() && sz_half->is_contiguous() && sz_half->scalar_type() == at::kInt && sz_half->numel() == sz_packed.numel());
    hp = sz_half->data_ptr();
  }
  raise_on(awq_w4a16_forward_cdna4_szh(in_feats.data_ptr(), kernel.data_ptr(), scales.data_ptr(), zeros.data_ptr(), sz_packed.data_ptr(), hp, bp,
                                       out.data_ptr(), (int)m, (int)n, (int)k, 128, dtype_code(in_feats), wsp, wsb,
                                       (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()));
  return out;
}

// grouped (per-expert) GEMM for MoE layers: tokens sorted by expert, stacked expert weights
torch::Tensor moe_gemm_forward(torch::Tensor x_sorted, torch::Tensor kernel, torch::Tensor scales, torch::Tensor zeros,
                               torch::Tensor expert_offsets, bool cdna4) {
  check_inputs(x_sorted, kernel, scales, zeros);
  TORCH_CHECK(expert_offsets.is_cuda() && expert_offsets.is_contiguous() && expert_offsets.scalar_type() == at::kInt);
  TORCH_CHECK(kernel.dim() == 3 && scales.dim() == 3 && zeros.dim() == 3 && x_sorted.dim() == 2);
  const int64_t e = kernel.size(0), n = kernel.size(1) * 4, k = kernel.size(2), t = x_sorted.size(0);
  TORCH_CHECK(x_sorted.size(1) == k && scales.size(0) == e && scales.size(2) == n && expert_offsets.numel() == e + 1);
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x_sorted.device());
  at::Tensor out = torch::empty({t, n}, x_sorted.options());
  raise_on(awq_w4a16_moe_gemm(x_sorted.data_ptr(), kernel.data_ptr(), scales.data_ptr(), zeros.data_ptr(),
                              expert_offsets.data_ptr(), out.data_ptr(), (int)t, (int)e, (int)n, (int)k, (int)scales.size(1), 128,
                              dtype_code(x_sorted), cdna4 ? 1 : 0,
                              (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()));
  return out;
}

// grouped forward on cdna4 buffers: decode batches (<= 8 sorted rows) take the grouped GEMV
torch::Tensor moe_forward_cdna4(torch::Tensor x_sorted, torch::Tensor kernel, torch::Tensor scales, torch::Tensor zeros,
                                torch::Tensor sz_packed, torch::Tensor expert_offsets) {
  check_inputs(x_sorted, kernel, scales, zeros);
  TORCH_CHECK(x_sorted.scalar_type() == at::kBFloat16 || x_sorted.scalar_type() == at::kHalf,
              "the cdna4 interleave is defined for bfloat16 / float16");
  TORCH_CHECK(expert_offsets.is_cuda() && expert_offsets.is_contiguous() && expert_offsets.scalar_type() == at::kInt);
  TORCH_CHECK(sz_packed.is_cuda() && sz_packed.is_contiguous() && sz_packed.scalar_type() == at::kInt);
  TORCH_CHECK(kernel.dim() == 3 && scales.dim() == 3 && zeros.dim() == 3 && x_sorted.dim() == 2);
  const int64_t e = kernel.size(0), n = kernel.size(1) * 4, k = kernel.size(2), t = x_sorted.size(0);
  TORCH_CHECK(x_sorted.size(1) == k && scales.size(0) == e && scales.size(2) == n && expert_offsets.numel() == e + 1);
  TORCH_CHECK(sz_packed.numel() == e * n * (k / 128), "sz_packed must be int32 [E, n/16, k/128, 16]");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(x_sorted.device());
  at::Tensor out = torch::empty({t, n}, x_sorted.options());
  raise_on(awq_w4a16_moe_forward_cdna4(x_sorted.data_ptr(), kernel.data_ptr(), scales.data_ptr(), zeros.data_ptr(),
                                       sz_packed.data_ptr(), expert_offsets.data_ptr(), out.data_ptr(), (int)t, (int)e, (int)n,
                                       (int)k, (int)scales.size(1), 128, dtype_code(x_sorted),
                                       (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()));
  return out;
}

// ---- W3 ("w3c" tiles, bf16): pack from logical integers, and WQLinear.forward for w_bit = 3 ----
torch::Tensor pack_w3(torch::Tensor q_u8) {
  TORCH_CHECK(q_u8.is_cuda() && q_u8.is_contiguous() && q_u8.scalar_type() == at::kByte && q_u8.dim() == 2);
  const int64_t n = q_u8.size(0), k = q_u8.size(1);
  TORCH_CHECK(n % 16 == 0 && k % 128 == 0, "w3c tiles need n % 16 == 0 and k % 128 == 0");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(q_u8.device());
  at::Tensor out = torch::empty({n / 4, k * 3 / 4}, q_u8.options().dtype(at::kShort));
  raise_on(awq_pack_w3(q_u8.data_ptr(), out.data_ptr(), (int)n, (int)k,
                       (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()));
  return out;
}

torch::Tensor forward_w3(torch::Tensor in_feats, torch::Tensor kernel, torch::Tensor scales, torch::Tensor zeros,
                         torch::Tensor sz_packed, c10::optional<torch::Tensor> bias) {
  check_inputs(in_feats, kernel, scales, zeros);
  TORCH_CHECK(in_feats.scalar_type() == at::kBFloat16 || in_feats.scalar_type() == at::kHalf, "the W3 path is defined for bfloat16 / float16");
  TORCH_CHECK(sz_packed.is_cuda() && sz_packed.is_contiguous() && sz_packed.scalar_type() == at::kInt);
  const int64_t n = kernel.size(0) * 4, k = in_feats.size(-1);
  TORCH_CHECK(k > 0 && in_feats.numel() % k == 0 && kernel.numel() == n / 4 * (k * 3 / 4), "qweight must be int16 [n/4, 3k/4]");
  TORCH_CHECK(sz_packed.numel() == n * (k / 128), "sz_packed must be int32 [n/16, k/128, 16]");
  const int64_t m = in_feats.numel() / k;
  std::vector<int64_t> shape = in_feats.sizes().vec();
  shape.back() = n;
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(in_feats.device());
  at::Tensor out = torch::empty(shape, in_feats.options());
  if (m == 0) return out;
  const void* bp = nullptr;
  if (bias.has_value() && bias->defined()) {
    TORCH_CHECK(bias->is_cuda() && bias->is_contiguous() && bias->scalar_type() == in_feats.scalar_type() && bias->numel() == n);
    bp = bias->data_ptr();
  }
  const size_t wsb = awq_w3a16_forward_workspace_bytes((int)m, (int)n, (int)k);
  at::Tensor ws;
  if (wsb) ws = torch::empty({(int64_t)wsb}, in_feats.options().dtype(at::kByte));
  raise_on(awq_w3a16_forward(in_feats.data_ptr(), kernel.data_ptr(), scales.data_ptr(), zeros.data_ptr(), sz_packed.data_ptr(),
                             bp, out.data_ptr(), (int)m, (int)n, (int)k, 128, dtype_code(in_feats), wsb ? ws.data_ptr() : nullptr, wsb,
                             (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()));
  return out;
}

// QuantLlamaMLP's gate/up + SiLU*mul in one launch (tinychat/modules/fused_mlp.py:36-83), decode rows only
torch::Tensor mlp_gate_up_cdna4(torch::Tensor in_feats, torch::Tensor kernel_gate_up, torch::Tensor sz_packed) {
  TORCH_CHECK(in_feats.is_cuda() && kernel_gate_up.is_cuda() && sz_packed.is_cuda());
  TORCH_CHECK(in_feats.is_contiguous() && kernel_gate_up.is_contiguous() && sz_packed.is_contiguous());
  TORCH_CHECK((in_feats.scalar_type() == at::kBFloat16 || in_feats.scalar_type() == at::kHalf) &&
              kernel_gate_up.scalar_type() == at::kShort && sz_packed.scalar_type() == at::kInt);
  const int64_t n2 = kernel_gate_up.size(0) * 4, k = in_feats.size(-1);
  TORCH_CHECK(k > 0 && in_feats.numel() % k == 0 && kernel_gate_up.numel() == n2 / 4 * k);
  TORCH_CHECK(sz_packed.numel() == n2 * (k / 128), "sz_packed must be int32 [n2/16, k/128, 16]");
  const int64_t m = in_feats.numel() / k;
  std::vector<int64_t> shape = in_feats.sizes().vec();
  shape.back() = n2 / 2;
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(in_feats.device());
  at::Tensor out = torch::empty(shape, in_feats.options());
  raise_on(awq_w4a16_mlp_gate_up_cdna4(in_feats.data_ptr(), kernel_gate_up.data_ptr(), sz_packed.data_ptr(), out.data_ptr(),
                                       (int)m, (int)n2, (int)k, 128, dtype_code(in_feats),
                                       (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()));
  return out;
}

// FTLlamaRMSNorm (tinychat/modules/fused_norm.py:7-21 -> layernorm.cu:39-61) fused in front of the decode GEMV: x is the
// UN-normalised activation, gamma the norm weight [k]; <= 4 rows.  fused_gate_up: kernel = stacked [gate; up], out [.., n/2].
// FTLlamaRMSNorm.forward for any row count (tinychat/modules/fused_norm.py:16-21 -> layernorm_forward_cuda): one launch
torch::Tensor rmsnorm(torch::Tensor in_feats, torch::Tensor gamma, double eps) {
  TORCH_CHECK(in_feats.is_cuda() && gamma.is_cuda() && in_feats.is_contiguous() && gamma.is_contiguous());
  TORCH_CHECK((in_feats.scalar_type() == at::kBFloat16 || in_feats.scalar_type() == at::kHalf) && gamma.scalar_type() == in_feats.scalar_type());
  const int64_t k = in_feats.size(-1);
  TORCH_CHECK(k > 0 && gamma.numel() == k);
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(in_feats.device());
  at::Tensor out = torch::empty_like(in_feats);
  if (in_feats.numel() == 0) return out;
  raise_on(awq_rmsnorm(in_feats.data_ptr(), gamma.data_ptr(), (float)eps, out.data_ptr(), (int)(in_feats.numel() / k), (int)k, dtype_code(in_feats),
                       (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()));
  return out;
}

torch::Tensor rmsnorm_forward_cdna4(torch::Tensor in_feats, torch::Tensor gamma, double eps, torch::Tensor kernel, torch::Tensor sz_packed,
                                    c10::optional<torch::Tensor> bias, bool fused_gate_up) {
  TORCH_CHECK(in_feats.is_cuda() && gamma.is_cuda() && kernel.is_cuda() && sz_packed.is_cuda());
  TORCH_CHECK(in_feats.is_contiguous() && gamma.is_contiguous() && kernel.is_contiguous() && sz_packed.is_contiguous());
  TORCH_CHECK((in_feats.scalar_type() == at::kBFloat16 || in_feats.scalar_type() == at::kHalf) && gamma.scalar_type() == in_feats.scalar_type() &&
              kernel.scalar_type() == at::kShort && sz_packed.scalar_type() == at::kInt);
  const int64_t n = kernel.size(0) * 4, k = in_feats.size(-1);
  TORCH_CHECK(k > 0 && in_feats.numel() % k == 0 && kernel.numel() == n / 4 * k && gamma.numel() == k);
  TORCH_CHECK(sz_packed.numel() == n * (k / 128), "sz_packed must be int32 [n/16, k/128, 16]");
  const int64_t m = in_feats.numel() / k;
  std::vector<int64_t> shape = in_feats.sizes().vec();
  shape.back() = fused_gate_up ? n / 2 : n;
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(in_feats.device());
  at::Tensor out = torch::empty(shape, in_feats.options());
  const void* bp = nullptr;
  if (bias.has_value() && bias->defined()) {
    TORCH_CHECK(bias->is_cuda() && bias->is_contiguous() && bias->scalar_type() == in_feats.scalar_type() && bias->numel() == n);
    bp = bias->data_ptr();
  }
  raise_on(awq_w4a16_rmsnorm_forward_cdna4(in_feats.data_ptr(), gamma.data_ptr(), (float)eps, kernel.data_ptr(), sz_packed.data_ptr(), bp,
                                           out.data_ptr(), (int)m, (int)n, (int)k, 128, dtype_code(in_feats), fused_gate_up ? 1 : 0,
                                           (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()));
  return out;
}


// QuantLlamaMLP.our_llama_mlp for any row count on the 8 + 8 interleaved gate / up pair: out [.., n2 / 2]
torch::Tensor mlp_gate_up_forward_cdna4(torch::Tensor in_feats, torch::Tensor kernel, torch::Tensor sz_packed, c10::optional<torch::Tensor> sz_half) {
  TORCH_CHECK(in_feats.is_cuda() && kernel.is_cuda() && sz_packed.is_cuda());
  TORCH_CHECK(in_feats.is_contiguous() && kernel.is_contiguous() && sz_packed.is_contiguous());
  TORCH_CHECK((in_feats.scalar_type() == at::kBFloat16 || in_feats.scalar_type() == at::kHalf) && kernel.scalar_type() == at::kShort &&
              sz_packed.scalar_type() == at::kInt);
  const int64_t n2 = kernel.size(0) * 4, k = in_feats.size(-1);
  TORCH_CHECK(k > 0 && in_feats.numel() % k == 0 && kernel.numel() == n2 / 4 * k);
  TORCH_CHECK(sz_packed.numel() == n2 * (k / 128), "sz_packed must be int32 [n2/16, k/128, 16]");
  const void* hp = nullptr;
  if (sz_half.has_value() && sz_half->defined()) {
    TORCH_CHECK(sz_half->is_cuda() && sz_half->is_contiguous() && sz_half->scalar_type() == at::kInt && sz_half->numel() == sz_packed.numel());
    hp = sz_half->data_ptr();
  }
  const int64_t m = in_feats.numel() / k;
  std::vector<int64_t> shape = in_feats.sizes().vec();
  shape.back() = n2 / 2;
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(in_feats.device());
  at::Tensor out = torch::empty(shape, in_feats.options());
  if (m == 0) return out;
  // optional scratch of the prefill plan (block pairs for the columns behind the full rounds): torch's caching allocator, stream-ordered, capture-safe
  at::Tensor ws;
  void* wsp = nullptr;
  const size_t wsb = awq_w4a16_mlp_gate_up_forward_cdna4_workspace_bytes((int)m, (int)n2, (int)k);
  if (wsb) {
    ws = torch::empty({(int64_t)wsb}, in_feats.options().dtype(at::kByte));
    wsp = ws.data_ptr();
  }
  raise_on(awq_w4a16_mlp_gate_up_forward_cdna4_ws(in_feats.data_ptr(), kernel.data_ptr(), sz_packed.data_ptr(), hp, out.data_ptr(), (int)m, (int)n2,
                                                  (int)k, 128, dtype_code(in_feats), wsp, wsb,
                                                  (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()));
  return out;
}

// "sz_half" side buffer of the decode kernels; returns (int32 [n/16, k/128, 16], exact).  Synchronises once (reads the flag).
std::tuple<torch::Tensor, bool> pack_szh_cdna4(torch::Tensor scales, torch::Tensor zeros, int k) {
  TORCH_CHECK(scales.is_cuda() && zeros.is_cuda() && scales.is_contiguous() && zeros.is_contiguous());
  TORCH_CHECK(scales.scalar_type() == zeros.scalar_type() && scales.sizes() == zeros.sizes() && scales.dim() == 2);
  TORCH_CHECK(scales.scalar_type() == at::kBFloat16 || scales.scalar_type() == at::kHalf);
  const int64_t n = scales.size(1);
  TORCH_CHECK(n % 16 == 0 && k % 128 == 0 && scales.size(0) * 128 >= k);
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(scales.device());
  at::Tensor out = torch::empty({n / 16, k / 128, 16}, scales.options().dtype(at::kInt));
  at::Tensor flag = torch::zeros({1}, scales.options().dtype(at::kInt));
  raise_on(awq_pack_szh_cdna4(scales.data_ptr(), zeros.data_ptr(), out.data_ptr(), (int*)flag.data_ptr(), (int)n, k, dtype_code(scales),
                              (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()));
  return {out, flag.item<int>() == 0};
}

// decode (<= 8 rows) on cdna4 weights + sz_half; epilogue 0: x.W^T (+bias), 1: stacked [gate; up] -> silu(gate)*up, 2: interleaved 8 + 8
torch::Tensor decode_cdna4(torch::Tensor in_feats, torch::Tensor kernel, torch::Tensor sz_half, c10::optional<torch::Tensor> bias,
                           int epilogue) {
  TORCH_CHECK(in_feats.is_cuda() && kernel.is_cuda() && sz_half.is_cuda());
  TORCH_CHECK(in_feats.is_contiguous() && kernel.is_contiguous() && sz_half.is_contiguous());
  TORCH_CHECK((in_feats.scalar_type() == at::kBFloat16 || in_feats.scalar_type() == at::kHalf) && kernel.scalar_type() == at::kShort &&
              sz_half.scalar_type() == at::kInt);
  const int64_t n = kernel.size(0) * 4, k = in_feats.size(-1);
  TORCH_CHECK(k > 0 && in_feats.numel() % k == 0 && kernel.numel() == n / 4 * k);
  TORCH_CHECK(sz_half.numel() == n * (k / 128), "sz_half must be int32 [n/16, k/128, 16]");
  const int64_t m = in_feats.numel() / k;
  std::vector<int64_t> shape = in_feats.sizes().vec();
  shape.back() = epilogue ? n / 2 : n;
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(in_feats.device());
  at::Tensor out = torch::empty(shape, in_feats.options());
  const void* bp = nullptr;
  if (bias.has_value() && bias->defined()) {
    TORCH_CHECK(bias->is_cuda() && bias->is_contiguous() && bias->scalar_type() == in_feats.scalar_type() && bias->numel() == n);
    bp = bias->data_ptr();
  }
  raise_on(awq_w4a16_decode_cdna4(in_feats.data_ptr(), kernel.data_ptr(), sz_half.data_ptr(), bp, out.data_ptr(), (int)m, (int)n, (int)k,
                                  128, dtype_code(in_feats), epilogue, (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()));
  return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "MI355X-native W4A16 kernels behind llm-awq's awq_inference_engine API (hot path only)";
  m.def("gemm_forward_cuda_new", &gemm_forward_cuda_new, "New quantized GEMM kernel.");
  m.def("gemv_forward_cuda_new", &gemv_forward_cuda_new, "New quantized GEMV kernel.");
  m.def("abi_version", []() { return awq_abi_version(); });
  m.def("cdna4_cache_info", []() {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    return py::dict(py::arg("enabled") = cache_enabled(), py::arg("entries") = (int64_t)g_cache.size(), py::arg("hits") = g_cache_hits,
                    py::arg("builds") = g_cache_builds, py::arg("sz_builds") = g_cache_sz_builds, py::arg("bytes") = g_cache_bytes,
                    py::arg("inplace") = g_cache_inplace == 1);
  }, "state of the lazy v2 -> cdna4 weight cache behind gemv/gemm_forward_cuda_new");
  auto clear_all = []() {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    while (!g_cache.empty()) {
      restore_inplace(g_cache.begin()->second);  // (in-place entries: the qweight gets its v2 interleave back)
      drop_entry(g_cache.begin());
    }
  };
  m.def("cdna4_cache_clear", clear_all);
  m.def("cdna4_cache_enable", [clear_all](bool on) {
    if (!on) clear_all();  // with the cache off the reference-layout kernels read the qweights: none may stay converted in place
    g_cache_enabled = on ? 1 : 0;
  });
  m.def("cdna4_cache_inplace", [clear_all](bool on) {
    cache_enabled();
    clear_all();
    g_cache_inplace = on ? 1 : 0;
  }, "AWQ_CDNA4_INPLACE at run time: convert qweights where they lie (no second copy) on their first call through the reference entry points");
  m.def("cdna4_restore", [](torch::Tensor kernel) {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    auto it = g_cache.find(kernel.data_ptr());
    if (it == g_cache.end()) return false;
    const bool was = it->second.inplace_intact();  // (overwritten since the conversion: the buffer holds v2 data, nothing is restored)
    restore_inplace(it->second);
    drop_entry(it);
    return was;
  }, "undo an in-place conversion of this qweight (e.g. before saving a reference-layout checkpoint); true if it was converted");
  m.def("decode_cdna4_plan", [](int m, int n, int k, int epilogue) {
    int kernel = 0;
    const int passes = awq_w4a16_decode_cdna4_plan(m, n, k, epilogue, &kernel);
    return py::make_tuple(passes, kernel);
  }, "host-side: (weight passes, kernel) of decode_cdna4 for this shape; passes == 0: the shape is not served (use forward_cdna4)");
  m.def("cdna4_is_converted", [](torch::Tensor kernel) {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    return kernel.is_cuda() && inplace_converted_locked(kernel);
  }, "true while this qweight's storage holds the cdna4 interleave because the cache converted it in place (AWQ_CDNA4_INPLACE): the module's "
     "`layout` attribute still says v2 -- format tools must cdna4_restore() it before they slice, repack or save it");
  // extras of the MI355X build (not part of the reference module)
  m.def("repack_v2_to_cdna4", &repack_v2_to_cdna4, "qweight v2 -> cdna4 interleave (same shape)");
  m.def("repack_cdna4_to_v2", &repack_cdna4_to_v2, "qweight cdna4 -> v2 interleave (same shape)");
  m.def("pack_sz_cdna4", &pack_sz_cdna4, "scales/scaled_zeros [Gpad,N] -> packed int32 [N/16, K/128, 16]");
  m.def("forward_cdna4", &forward_cdna4, "WQLinear forward on cdna4-interleaved buffers", py::arg("in_feats"),
        py::arg("kernel"), py::arg("scales"), py::arg("zeros"), py::arg("sz_packed"), py::arg("bias") = py::none(), py::arg("sz_half") = py::none());
  m.def("moe_gemm_forward", &moe_gemm_forward, "grouped per-expert W4A16 GEMM (tokens sorted by expert)", py::arg("x_sorted"),
        py::arg("kernel"), py::arg("scales"), py::arg("zeros"), py::arg("expert_offsets"), py::arg("cdna4") = false);
  m.def("moe_forward_cdna4", &moe_forward_cdna4, "grouped per-expert forward on cdna4 buffers (GEMV for <= 8 rows, GEMM otherwise)");
  m.def("pack_w3", &pack_w3, "logical uint8 [N, K] (0..7) -> w3c tiles int16 [N/4, 3K/4]");
  m.def("forward_w3", &forward_w3, "WQLinear forward for w_bit = 3 (w3c tiles)", py::arg("in_feats"), py::arg("kernel"),
        py::arg("scales"), py::arg("zeros"), py::arg("sz_packed"), py::arg("bias") = py::none());
  m.def("pack_szh_cdna4", &pack_szh_cdna4, "scales/scaled_zeros [Gpad,N] -> (sz_half int32 [N/16, K/128, 16], exact)");
  m.def("decode_cdna4", &decode_cdna4, "<= 8 rows on cdna4 weights + sz_half (LDS-DMA streaming kernel)", py::arg("in_feats"),
        py::arg("kernel"), py::arg("sz_half"), py::arg("bias") = py::none(), py::arg("epilogue") = 0);
  m.def("rmsnorm", &rmsnorm, "RMSNorm of every row (layernorm_forward_cuda's arithmetic), one launch", py::arg("in_feats"), py::arg("gamma"), py::arg("eps"));
  m.def("rmsnorm_forward_cdna4", &rmsnorm_forward_cdna4, "RMSNorm fused in front of the decode GEMV (<= 4 rows)", py::arg("in_feats"),
        py::arg("gamma"), py::arg("eps"), py::arg("kernel"), py::arg("sz_packed"), py::arg("bias") = py::none(),
        py::arg("fused_gate_up") = false);
  m.def("mlp_gate_up_forward_cdna4", &mlp_gate_up_forward_cdna4, "silu(x Wg^T) * (x Wu^T) for any row count on the 8 + 8 interleaved gate/up pair",
        py::arg("in_feats"), py::arg("kernel"), py::arg("sz_packed"), py::arg("sz_half") = py::none());
  m.def("mlp_gate_up_cdna4", &mlp_gate_up_cdna4, "silu(x Wg^T) * (x Wu^T) on stacked cdna4 gate/up buffers, <= 8 rows");
}
