// mi355_ops_api.cpp -- see mi355_ops_api.h. Host C++ only: tensor checks, output allocation, stream lookup, and
// one C-ABI call per operator. No kernels here.
#include "mi355_ops_api.h"

// ROCm builds of torch present HIP devices as "cuda": use the masquerading guard / stream accessors
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <ATen/hip/HIPGeneratorImpl.h>

#include <ATen/hip/HIPEvent.h>
#include <c10/hip/HIPGraphsC10Utils.h>

#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../include/xllm_mi355.h"

namespace xllm::kernel::mi355 {
namespace {

void* cur_stream() { return static_cast<void*>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()); }
using DeviceGuard = c10::hip::OptionalHIPGuardMasqueradingAsCUDA;

int dt(const torch::Tensor& t) {
  switch (t.scalar_type()) {
    case torch::kFloat32: return XM_F32;
    case torch::kBFloat16: return XM_BF16;
    case torch::kFloat16: return XM_F16;
    default: TORCH_CHECK(false, "xllm_mi355: unsupported dtype ", t.scalar_type());
  }
}
int dt(torch::ScalarType s) {
  TORCH_CHECK(s == torch::kBFloat16 || s == torch::kFloat16, "output dtype must be half or bfloat16");
  return s == torch::kBFloat16 ? XM_BF16 : XM_F16;
}
// the library behind the C ABI must be the one this file was compiled against (same argument lists behind the same names)
const bool kAbiChecked = [] {
  TORCH_CHECK(xllm_mi355_abi_version() == XLLM_MI355_ABI_VERSION, "libxllm_mi355.so reports ABI version ",
              xllm_mi355_abi_version(), ", the shim was compiled against ", XLLM_MI355_ABI_VERSION);
  return true;
}();
void check(int rc, const char* what) { TORCH_CHECK(rc == 0, what, ": ", xllm_mi355_strerror(rc)); }
void* p(const torch::Tensor& t) { return t.data_ptr(); }
void* p(const std::optional<torch::Tensor>& t) { return t.has_value() && t->defined() ? t->data_ptr() : nullptr; }
int act_code(const std::string& m) {
  if (m == "silu") return XM_ACT_SILU;
  if (m == "gelu") return XM_ACT_GELU;
  if (m == "gelu_tanh") return XM_ACT_GELU_TANH;
  TORCH_CHECK(false, "Unsupported act mode: ", m, ", only support silu, gelu, gelu_tanh");
}

}  // namespace

void rotary_embedding(torch::Tensor& positions, torch::Tensor& query, std::optional<torch::Tensor> key,
                      torch::Tensor& cos_sin_cache, bool is_neox) {
  DeviceGuard guard(query.device());
  TORCH_CHECK(positions.scalar_type() == torch::kInt64, "positions must be int64");
  const int64_t T = positions.numel();
  const int64_t rot = cos_sin_cache.size(-1);
  // head_size: the reference derives it from the tensor shape (rope.cu:176-200); q is [T, nq*hs] or [T,nq,hs]
  const int64_t hs = query.dim() == 3 ? query.size(-1) : rot;
  const int64_t nq = query.numel() / T / hs;
  const int64_t nk = key.has_value() ? key->numel() / T / hs : 0;
  check(xllm_mi355_rotary_embedding(positions.data_ptr<int64_t>(), p(query), p(key), p(cos_sin_cache), T, nq, nk, hs,
                                    rot, query.stride(0), key.has_value() ? key->stride(0) : 0, hs, is_neox ? 1 : 0,
                                    dt(query), cur_stream()),
        "rotary_embedding");
}

void act_and_mul(torch::Tensor out, torch::Tensor input, const std::string& act_mode) {
  DeviceGuard guard(input.device());
  const int64_t d = input.size(-1) / 2;
  check(xllm_mi355_act_and_mul(p(out), p(input), input.numel() / (2 * d), d, act_code(act_mode), dt(input),
                               cur_stream()),
        "act_and_mul");
}

void reshape_paged_cache(torch::Tensor slot_ids, torch::Tensor keys, torch::Tensor values, torch::Tensor key_cache,
                         torch::Tensor value_cache) {
  DeviceGuard guard(keys.device());
  TORCH_CHECK(keys.stride(-1) == 1 && keys.stride(-2) == keys.size(-1));      // reshape_paged_cache.cu:73
  TORCH_CHECK(values.stride(-1) == 1 && values.stride(-2) == values.size(-1));  // :74
  check(xllm_mi355_reshape_paged_cache(slot_ids.data_ptr<int32_t>(), p(keys), p(values), p(key_cache), p(value_cache),
                                       keys.size(-3), keys.size(-2), keys.size(-1), key_cache.size(-3),
                                       key_cache.size(0), keys.stride(-3), values.stride(-3),
                                       (int)keys.element_size(), cur_stream()),
        "reshape_paged_cache");
}

torch::Tensor bmm_heads(const torch::Tensor& x, const torch::Tensor& w_nk, std::optional<torch::Tensor> out) {
  TORCH_CHECK(x.dim() == 3 && w_nk.dim() == 3 && x.size(1) == w_nk.size(0) && x.size(2) == w_nk.size(2),
              "bmm_heads: x [tokens, heads, K], w [heads, N, K]");
  TORCH_CHECK(x.stride(2) == 1 && w_nk.stride(2) == 1 && x.scalar_type() == w_nk.scalar_type());
  DeviceGuard guard(x.device());
  torch::Tensor o = out.has_value() ? *out : torch::empty({x.size(0), x.size(1), w_nk.size(1)}, x.options());
  TORCH_CHECK(o.stride(2) == 1);
  check(xllm_mi355_bmm_heads(p(x), x.stride(0), x.stride(1), p(w_nk), w_nk.stride(0), w_nk.stride(1), p(o), o.stride(0),
                             o.stride(1), x.size(0), x.size(1), w_nk.size(1), x.size(2), dt(x), cur_stream()),
        "bmm_heads");
  return o;
}

void block_copy(torch::Tensor key_cache_ptrs, torch::Tensor value_cache_ptrs, torch::Tensor src_block_indices,
                torch::Tensor dst_block_indices, torch::Tensor cum_sum, int64_t numel_per_block,
                torch::ScalarType cache_dtype) {
  if (src_block_indices.numel() == 0) return;                                     // block_copy.cu:128-130
  for (const torch::Tensor* t : {&key_cache_ptrs, &value_cache_ptrs, &src_block_indices, &dst_block_indices, &cum_sum})
    TORCH_CHECK(t->is_cuda() && t->dim() == 1 && t->is_contiguous());            // :136-155
  TORCH_CHECK(key_cache_ptrs.scalar_type() == torch::kInt64 && value_cache_ptrs.scalar_type() == torch::kInt64);
  TORCH_CHECK(src_block_indices.scalar_type() == torch::kInt32 && dst_block_indices.scalar_type() == torch::kInt32 &&
              cum_sum.scalar_type() == torch::kInt32);
  TORCH_CHECK(key_cache_ptrs.size(0) == value_cache_ptrs.size(0));               // :156
  TORCH_CHECK(src_block_indices.size(0) == cum_sum.size(0));                     // :157
  TORCH_CHECK(numel_per_block > 0);                                              // :158
  DeviceGuard guard(key_cache_ptrs.device());
  check(xllm_mi355_block_copy(key_cache_ptrs.data_ptr<int64_t>(), value_cache_ptrs.data_ptr<int64_t>(),
                              src_block_indices.data_ptr<int32_t>(), dst_block_indices.data_ptr<int32_t>(),
                              cum_sum.data_ptr<int32_t>(), key_cache_ptrs.size(0), src_block_indices.size(0),
                              dst_block_indices.size(0), numel_per_block * (int64_t)c10::elementSize(cache_dtype),
                              cur_stream()),
        "block_copy");
}

void rms_norm(torch::Tensor output, torch::Tensor input, torch::Tensor weight, double eps) {
  DeviceGuard guard(input.device());
  const int64_t H = input.size(-1);
  auto x = input.view({-1, H});
  check(xllm_mi355_rms_norm(p(output), p(x), p(weight), (float)eps, x.size(0), H, x.stride(0), dt(input), cur_stream()),
        "rms_norm");
}

void fused_add_rms_norm(torch::Tensor& input, torch::Tensor& residual, torch::Tensor& weight, double epsilon) {
  DeviceGuard guard(input.device());
  const int64_t H = input.size(-1);
  check(xllm_mi355_fused_add_rms_norm(p(input), p(residual), p(weight), (float)epsilon, input.numel() / H, H, H,
                                      dt(input), cur_stream()),
        "fused_add_rms_norm");
}

torch::Tensor matmul(torch::Tensor a, torch::Tensor b, std::optional<torch::Tensor> bias) {
  DeviceGuard guard(a.device());
  const int64_t K = a.size(-1), N = b.size(0);
  auto a2 = a.reshape({-1, K}).contiguous();
  auto out = torch::empty({a2.size(0), N}, a.options());
  check(xllm_mi355_matmul(p(a2), p(b.contiguous()), p(bias), p(out), a2.size(0), N, K, dt(a), cur_stream()), "matmul");
  auto shape = a.sizes().vec();
  shape.back() = N;
  return out.view(shape);
}

void static_scaled_fp8_quant(torch::Tensor& out, torch::Tensor const& input, torch::Tensor const& scale) {
  DeviceGuard guard(input.device());
  check(xllm_mi355_static_scaled_fp8_quant(static_cast<uint8_t*>(p(out)), p(input.contiguous()),
                                           scale.data_ptr<float>(), input.numel(), dt(input), cur_stream()),
        "static_scaled_fp8_quant");
}

std::tuple<torch::Tensor, torch::Tensor> fp8_scaled_quantize(const torch::Tensor& input,
                                                             const std::optional<torch::Tensor>& output,
                                                             const std::optional<torch::Tensor>& scale) {
  DeviceGuard guard(input.device());
  torch::Tensor q = (output.has_value() && output->defined())
                        ? *output
                        : torch::empty_like(input, input.options().dtype(torch::kFloat8_e4m3fn));
  const bool is_static = scale.has_value() && scale->defined();
  torch::Tensor s = is_static ? *scale : torch::empty({1}, input.options().dtype(torch::kFloat32));
  auto x = input.contiguous();
  if (is_static) {
    check(xllm_mi355_fp8_scaled_quantize(static_cast<uint8_t*>(p(q)), p(x), s.data_ptr<float>(), nullptr, x.numel(), dt(x),
                                         cur_stream()),
          "fp8_scaled_quantize");
  } else {
    // dynamic scale: per-block maxima in a transient buffer of this call, then fold + quantise (two launches, no memset node)
    torch::Tensor ws = torch::empty({(int64_t)xllm_mi355_fp8_scaled_quantize_workspace_bytes()},
                                    input.options().dtype(torch::kUInt8));
    check(xllm_mi355_fp8_scaled_quantize_ws(static_cast<uint8_t*>(p(q)), p(x), s.data_ptr<float>(), x.numel(), dt(x), p(ws),
                                            (size_t)ws.numel(), cur_stream()),
          "fp8_scaled_quantize");
  }
  return {q, s};
}

void rms_norm_static_fp8_quant(torch::Tensor& out, torch::Tensor& input, torch::Tensor& weight, torch::Tensor& scale,
                               double epsilon) {
  DeviceGuard guard(input.device());
  const int64_t H = input.size(-1);
  auto x = input.view({-1, H});
  check(xllm_mi355_rms_norm_static_fp8_quant(static_cast<uint8_t*>(p(out)), p(x), nullptr, p(weight),
                                             scale.data_ptr<float>(), (float)epsilon, x.size(0), H, x.stride(0),
                                             dt(input), cur_stream()),
        "rms_norm_static_fp8_quant");
}

void fused_add_rms_norm_static_fp8_quant(torch::Tensor& out, torch::Tensor& input, torch::Tensor& residual,
                                         torch::Tensor& weight, torch::Tensor& scale, double epsilon) {
  DeviceGuard guard(input.device());
  const int64_t H = input.size(-1);
  auto x = input.view({-1, H});
  check(xllm_mi355_rms_norm_static_fp8_quant(static_cast<uint8_t*>(p(out)), p(x), p(residual), p(weight),
                                             scale.data_ptr<float>(), (float)epsilon, x.size(0), H, x.stride(0),
                                             dt(input), cur_stream()),
        "fused_add_rms_norm_static_fp8_quant");
}

torch::Tensor fp8_scaled_matmul(const torch::Tensor& a, const torch::Tensor& b, const torch::Tensor& a_scale,
                                const torch::Tensor& b_scale, torch::ScalarType output_dtype,
                                const std::optional<torch::Tensor>& bias, const std::optional<torch::Tensor>& output) {
  DeviceGuard guard(a.device());
  TORCH_CHECK(a.dim() == 2 && b.dim() == 2 && a.size(1) == b.size(1), "a [M,K], b [N,K]");
  const int64_t M = a.size(0), K = a.size(1), N = b.size(0);
  torch::Tensor out = output.has_value() ? *output : torch::empty({M, N}, a.options().dtype(output_dtype));
  check(xllm_mi355_fp8_scaled_matmul(static_cast<const uint8_t*>(p(a)), static_cast<const uint8_t*>(p(b)),
                                     a_scale.data_ptr<float>(), a_scale.numel(), b_scale.data_ptr<float>(),
                                     b_scale.numel(), p(bias), p(out), M, N, K, dt(output_dtype), cur_stream()),
        "fp8_scaled_matmul");
  return out;
}

void fused_qk_norm_rope(torch::Tensor& qkv, int64_t num_heads_q, int64_t num_heads_k, int64_t num_heads_v,
                        int64_t head_dim, double eps, const torch::Tensor& q_weight, const torch::Tensor& k_weight,
                        const torch::Tensor& cos_sin_cache, bool interleaved, const torch::Tensor& position_ids) {
  DeviceGuard guard(qkv.device());
  check(xllm_mi355_fused_qk_norm_rope(p(qkv), qkv.size(0), num_heads_q, num_heads_k, num_heads_v, head_dim, (float)eps,
                                      p(q_weight), p(k_weight), p(cos_sin_cache), dt(cos_sin_cache),
                                      interleaved ? 1 : 0, position_ids.data_ptr<int64_t>(), dt(qkv), cur_stream()),
        "fused_qk_norm_rope");
}

std::tuple<torch::Tensor, torch::Tensor> scaled_quantize(
    const torch::Tensor& x, const torch::Tensor& smooth, const std::optional<torch::Tensor>& zero,
    const std::optional<torch::Tensor>& token_count, const std::optional<torch::Tensor>& gather_index,
    const std::optional<torch::Tensor>& gather_index_start_position, const std::optional<torch::Tensor>& output,
    const std::optional<torch::Tensor>& output_scale, const std::string& act_mode, double /*active_coef*/,
    bool is_gated, torch::ScalarType quant_type) {
  DeviceGuard guard(x.device());
  // same restrictions as the DCU implementation (kernels/dcu/scaled_quantize.hip:411-447)
  TORCH_CHECK(!smooth.defined() || smooth.numel() == 0, "mi355 scaled_quantize: smooth factor not supported");
  TORCH_CHECK(!zero.has_value() && !token_count.has_value() && !gather_index.has_value() &&
                  !gather_index_start_position.has_value(),
              "mi355 scaled_quantize: only plain per-token int8 quantization is supported");
  TORCH_CHECK(quant_type == torch::kInt8 || quant_type == torch::kChar, "quant_type must be int8");
  TORCH_CHECK(x.dim() == 2 && x.is_contiguous(), "x must be a contiguous [M,K] tensor");
  const int64_t M = x.size(0), K = x.size(1);
  if (is_gated) {  // N1 fusion: act(gate)*up then quantize (ScaledQuantizeParams.act_mode/is_gated, param.h:805-815)
    const int64_t d = K / 2;
    torch::Tensor q = output.has_value() ? *output : torch::empty({M, d}, x.options().dtype(torch::kInt8));
    torch::Tensor s = output_scale.has_value() ? *output_scale : torch::empty({M}, x.options().dtype(torch::kFloat32));
    check(xllm_mi355_act_and_mul_dynamic_int8_quant(q.data_ptr<int8_t>(), s.data_ptr<float>(), p(x), M, d,
                                                    act_code(act_mode), dt(x), cur_stream()),
          "scaled_quantize(is_gated)");
    return {q, s};
  }
  torch::Tensor q = output.has_value() ? *output : torch::empty({M, K}, x.options().dtype(torch::kInt8));
  torch::Tensor s = output_scale.has_value() ? *output_scale : torch::empty({M}, x.options().dtype(torch::kFloat32));
  check(xllm_mi355_scaled_quantize(p(x), q.data_ptr<int8_t>(), s.data_ptr<float>(), M, K, dt(x), cur_stream()),
        "scaled_quantize");
  return {q, s};
}

namespace {
// Decode-shaped W8A8 GEMMs stream the weight in MFMA-fragment order (xllm_mi355_pack_weight_i8 + xllm_mi355_scaled_matmul_packed).
// The reference's operator has no "packed weight" argument, so the shim keeps one packed copy per weight tensor. The intended
// use is EXPLICIT: the linear layer packs at weight-load time (pack_w8a8_weight below, called from the USE_MI355 branch of the
// loader -- INTEGRATION.md) and the copy lives as long as the weight. The on-demand path (first decode-shaped call) stays as a
// convenience for callers that never registered. Round-2 advisor findings, all addressed here:
//   * identity: entries are keyed on the TensorImpl and hold a WEAK reference to it plus its version counter -- a weight that was
//     freed (the weak pointer is expired) or updated in place through torch (version moved) is re-packed; a recycled address can
//     no longer alias a dead entry. Expired entries are dropped on every insertion. (Raw-pointer writes into a weight bump no
//     version: call invalidate_packed_weight / clear_packed_weight_cache after them.)
//   * ordering: packing records an event on the packing stream; a call on another stream waits for it first;
//   * lifetime: look-ups return the tensor BY VALUE (its own reference), not a pointer into the map;
//   * scratch: K-slice slabs are per (device, STREAM), not per device; nothing is allocated inside a graph capture (a missing
//     buffer or an unpacked weight makes the call fall back to the row-major kernel there).
// The reference runs one worker thread per device in one process (runtime/dist_manager.cpp:82-84): all maps are mutex-guarded.
struct PackedEntry {
  c10::weak_intrusive_ptr<c10::TensorImpl, c10::UndefinedTensorImpl> owner;
  torch::Tensor packed;
  int64_t n, k;
  uint32_t version;
  const void* data;
  std::shared_ptr<at::cuda::CUDAEvent> ready;
  void* pack_stream;
};
std::mutex g_pack_mu;
std::unordered_map<const c10::TensorImpl*, PackedEntry> g_packed;
struct StreamKey {
  int device;
  void* stream;
  bool operator==(const StreamKey& o) const { return device == o.device && stream == o.stream; }
};
struct StreamKeyHash {
  size_t operator()(const StreamKey& k) const { return std::hash<void*>()(k.stream) ^ (size_t)k.device * 0x9e3779b97f4a7c15ull; }
};
std::unordered_map<StreamKey, torch::Tensor, StreamKeyHash> g_slab_ws;      // K-slice slabs of the packed kernels
std::unordered_map<StreamKey, torch::Tensor, StreamKeyHash> g_splitk_ws;    // zero-at-rest split-K scratch of the row-major kernels
constexpr int64_t kSlabBytes = 64ll << 20, kSplitKBytes = 64ll << 20;

bool capturing() { return c10::hip::currentStreamCaptureStatusMayInitCtx() != c10::hip::CaptureStatus::None; }

bool prefer_packed(int64_t M, int64_t N, int64_t K) {   // the measured policy of xllm_amd/ops.py::_prefer_packed (round 3)
  return N % 16 == 0 && K % 128 == 0 && K / 128 >= 4 && M <= 512;
}

void drop_expired_locked() {
  for (auto it = g_packed.begin(); it != g_packed.end();)
    it = it->second.owner.expired() ? g_packed.erase(it) : std::next(it);
}

// packs `b` on the current stream and registers the copy (caller holds no lock)
std::optional<torch::Tensor> pack_and_register(const torch::Tensor& b) {
  torch::Tensor packed = torch::empty_like(b);
  if (xllm_mi355_pack_weight_i8(b.data_ptr<int8_t>(), packed.data_ptr<int8_t>(), b.size(0), b.size(1), cur_stream()) != 0)
    return std::nullopt;
  auto ev = std::make_shared<at::cuda::CUDAEvent>();
  ev->record(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA());
  std::lock_guard<std::mutex> lock(g_pack_mu);
  drop_expired_locked();
  c10::TensorImpl* impl = b.unsafeGetTensorImpl();
  g_packed.erase(impl);
  g_packed.emplace(impl, PackedEntry{c10::weak_intrusive_ptr<c10::TensorImpl, c10::UndefinedTensorImpl>(b.getIntrusivePtr()),
                                     packed, b.size(0), b.size(1), impl->version_counter().current_version(), b.data_ptr(), ev,
                                     cur_stream()});
  return packed;
}

std::optional<torch::Tensor> packed_weight_for(const torch::Tensor& b) {
  c10::TensorImpl* impl = b.unsafeGetTensorImpl();
  const uint32_t ver = impl->version_counter().current_version();
  {
    std::lock_guard<std::mutex> lock(g_pack_mu);
    auto it = g_packed.find(impl);
    if (it != g_packed.end()) {
      const PackedEntry& e = it->second;
      if (!e.owner.expired() && e.n == b.size(0) && e.k == b.size(1) && e.version == ver && e.data == b.data_ptr()) {
        if (e.pack_stream != cur_stream()) e.ready->block(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA());
        return e.packed;
      }
      g_packed.erase(it);   // a dead or changed weight behind a recycled TensorImpl address
    }
  }
  if (capturing()) return std::nullopt;   // no allocation / packing inside a capture: the row-major kernel serves the call
  return pack_and_register(b);
}

// scratch of one (device, stream); created lazily OUTSIDE a capture, never replaced afterwards (graphs bake the address in)
std::optional<torch::Tensor> stream_scratch(std::unordered_map<StreamKey, torch::Tensor, StreamKeyHash>& map,
                                            const torch::Tensor& like, int64_t bytes, bool zero_at_rest) {
  const StreamKey key{(int)like.device().index(), cur_stream()};
  // ONE critical section for look-up, allocation, registration with the C side and insertion: two threads that miss on the same
  // (device, stream) must not both register a buffer of which one is then destroyed (round-3 advisor finding: the loser's
  // pointer stayed registered as the split-K workspace). torch::empty under the lock is fine: first use per stream only.
  std::lock_guard<std::mutex> lock(g_pack_mu);
  auto it = map.find(key);
  if (it != map.end()) return it->second;
  if (capturing()) return std::nullopt;
  torch::Tensor ws = torch::empty({bytes}, like.options().dtype(torch::kUInt8));
  if (zero_at_rest)   // the C side zeroes it (a memset on the null stream) and keys it on the stream
    check(xllm_mi355_set_gemm_workspace_for_stream(cur_stream(), ws.data_ptr(), (size_t)ws.numel()), "set_gemm_workspace_for_stream");
  return map.emplace(key, ws).first->second;
}
}  // namespace

torch::Tensor pack_w8a8_weight(const torch::Tensor& b) {
  DeviceGuard guard(b.device());
  TORCH_CHECK(b.dim() == 2 && b.scalar_type() == torch::kInt8 && b.is_contiguous(), "pack_w8a8_weight: [N, K] int8, contiguous");
  TORCH_CHECK(b.size(0) % 16 == 0 && b.size(1) % 128 == 0, "pack_w8a8_weight: N % 16 == 0 and K % 128 == 0");
  auto packed = pack_and_register(b);
  TORCH_CHECK(packed.has_value(), "pack_w8a8_weight: xllm_mi355_pack_weight_i8 failed");
  return *packed;
}

void invalidate_packed_weight(const torch::Tensor& b) {
  std::lock_guard<std::mutex> lock(g_pack_mu);
  g_packed.erase(b.unsafeGetTensorImpl());
}

int64_t packed_weight_cache_size() {
  std::lock_guard<std::mutex> lock(g_pack_mu);
  drop_expired_locked();
  return (int64_t)g_packed.size();
}

void clear_packed_weight_cache() {
  std::lock_guard<std::mutex> lock(g_pack_mu);
  g_packed.clear();
}

torch::Tensor scaled_matmul(const torch::Tensor& a, const torch::Tensor& b, const std::optional<torch::Tensor>& a_scale,
                            const torch::Tensor& b_scale, torch::ScalarType output_dtype,
                            const std::optional<torch::Tensor>& bias, const std::optional<torch::Tensor>& c,
                            const std::string& /*act_mode*/, int64_t quant_bit_size, double alpha, double beta,
                            bool /*use_hp_active*/, int64_t a_quant_bit_size,
                            const std::optional<torch::Tensor>& /*a_calib*/,
                            const std::optional<torch::Tensor>& /*b_calib*/,
                            const std::optional<torch::Tensor>& output) {
  DeviceGuard guard(a.device());
  TORCH_CHECK(quant_bit_size == 8 && a_quant_bit_size == 8, "scaled_matmul only supports w8a8 quantization");
  TORCH_CHECK(a.dim() == 2 && b.dim() == 2 && a.scalar_type() == torch::kInt8 && b.scalar_type() == torch::kInt8 &&
                  a.size(1) == b.size(1) && a.is_contiguous() && b.is_contiguous(),
              "scaled_matmul: a [M,K] int8, b [N,K] int8, contiguous");
  TORCH_CHECK(a_scale.has_value() && a_scale->defined(), "a_scale is required for scaled_matmul");
  const int64_t M = a.size(0), K = a.size(1), N = b.size(0);
  torch::Tensor out = output.has_value() ? *output : torch::empty({M, N}, a.options().dtype(output_dtype));
  auto as = a_scale->reshape({-1}).contiguous();
  auto bs = b_scale.reshape({-1}).contiguous();
  // ScaledMatmulParams::c (param.h:852-866): honoured for alpha = beta = 1 -- the reference's DCU backend drops c silently
  // (kernels/dcu/scaled_matmul.cpp:112-113); any other alpha / beta is refused rather than ignored
  const bool has_c = c.has_value() && c->defined();
  if (has_c) {
    TORCH_CHECK(alpha == 1.0 && beta == 1.0, "mi355 scaled_matmul: c is supported with alpha = beta = 1 only");
    TORCH_CHECK(c->dim() == 2 && c->size(0) == M && c->size(1) == N && c->scalar_type() == output_dtype && c->is_contiguous(),
                "scaled_matmul: c must be a contiguous [M, N] tensor of the output dtype");
    const int rc = xllm_mi355_scaled_matmul_add(a.data_ptr<int8_t>(), b.data_ptr<int8_t>(), as.data_ptr<float>(),
                                                bs.data_ptr<float>(), p(bias), c->data_ptr(), p(out), M, N, K,
                                                dt(output_dtype), cur_stream());
    if (rc == 0) return out;
    TORCH_CHECK(rc == XM_ERR_UNSUPPORTED, "scaled_matmul (c): ", xllm_mi355_strerror(rc));
    // no epilogue takes the addend at this shape: the plain product into a temporary, then one 16-bit add
    torch::Tensor y = scaled_matmul(a, b, a_scale, b_scale, output_dtype, bias, std::nullopt, "none", quant_bit_size, 1.0, 1.0,
                                    false, a_quant_bit_size, std::nullopt, std::nullopt, std::nullopt);
    check(xllm_mi355_add16(out.data_ptr(), y.data_ptr(), c->data_ptr(), M * N, dt(output_dtype), cur_stream()), "add16");
    return out;
  }
  if (prefer_packed(M, N, K)) {   // decode shapes: the weight-stream kernel on the packed copy; declines fall through
    if (auto wp = packed_weight_for(b)) {
      auto ws = stream_scratch(g_slab_ws, a, kSlabBytes, false);   // (none inside a capture that was not warmed up: unsliced)
      const int rc = xllm_mi355_scaled_matmul_packed(a.data_ptr<int8_t>(), wp->data_ptr<int8_t>(), as.data_ptr<float>(),
                                                     bs.data_ptr<float>(), p(bias), p(out), nullptr, M, N, K,
                                                     dt(output_dtype), ws ? ws->data_ptr() : nullptr,
                                                     ws ? (size_t)ws->numel() : 0, cur_stream());
      if (rc == 0) return out;
      TORCH_CHECK(rc == XM_ERR_UNSUPPORTED || rc == XM_ERR_WORKSPACE, "scaled_matmul (packed): ", xllm_mi355_strerror(rc));
    }
  }
  // row-major kernels: decode shapes split K through a zero-at-rest scratch registered for THIS (device, stream) -- the
  // registration the Python mirror does in ops._ensure_gemm_workspace and the shim used to skip (round-2 review, missing #6)
  if (M <= 512) (void)stream_scratch(g_splitk_ws, a, kSplitKBytes, true);
  check(xllm_mi355_scaled_matmul(a.data_ptr<int8_t>(), b.data_ptr<int8_t>(), as.data_ptr<float>(),
                                 bs.data_ptr<float>(), p(bias), p(out), nullptr, M, N, K, dt(output_dtype),
                                 cur_stream()),
        "scaled_matmul");
  return out;
}

torch::Tensor group_gemm(const torch::Tensor& input, const torch::Tensor& weight, const torch::Tensor& token_count,
                         std::optional<torch::Tensor> output) {
  DeviceGuard guard(input.device());
  TORCH_CHECK(input.dim() == 2 && weight.dim() == 3 && input.size(1) == weight.size(2), "group_gemm shapes");
  const int64_t E = weight.size(0), N = weight.size(1), K = weight.size(2);
  torch::Tensor out = output.has_value() ? *output : torch::empty({input.size(0), N}, input.options());
  check(xllm_mi355_group_gemm(p(input), p(weight), token_count.data_ptr<int32_t>(), p(out), input.size(0), E, N, K,
                              dt(input), cur_stream()),
        "group_gemm");
  return out;
}

namespace {
// MoE scratch (chunk counts of the index build + the grouped-GEMM tile table): the C ABI never allocates, so the shim owns it the
// way the reference's backends own their workspaces -- ONE PER DEVICE (the C side keys the registration on the device that owns
// the pointer), mutex-guarded, grown only outside a graph capture, and an outgrown buffer is RETIRED, not freed: a captured
// graph may still launch kernels that write it (round-2 review, weak #9: this was one static tensor, replaced in place and
// thrashing whenever two devices alternated)
std::mutex g_moe_mu;
std::unordered_map<int, torch::Tensor> g_moe_ws;
std::vector<torch::Tensor> g_moe_retired;
void ensure_moe_scratch(const torch::Tensor& like, int64_t bytes) {
  std::lock_guard<std::mutex> lock(g_moe_mu);
  auto& ws = g_moe_ws[(int)like.device().index()];
  if (ws.defined() && ws.numel() >= bytes) return;
  TORCH_CHECK(!capturing(), "MoE scratch too small inside a graph capture: run one eager step of this shape first");
  if (ws.defined()) g_moe_retired.push_back(ws);
  ws = torch::empty({std::max<int64_t>(bytes, 4 << 20)}, like.options().dtype(torch::kUInt8));
  check(xllm_mi355_set_moe_workspace(ws.data_ptr(), (size_t)ws.numel()), "set_moe_workspace");
}
}  // namespace

std::tuple<torch::Tensor, torch::Tensor> moe_fused_topk(const torch::Tensor& gating_output, int64_t topk, bool renormalize,
                                                        const std::optional<torch::Tensor>& correction_bias,
                                                        const std::string& scoring_func) {
  TORCH_CHECK(scoring_func == "softmax" || scoring_func == "sigmoid", "Unsupported scoring function for moe topk: ",
              scoring_func);
  TORCH_CHECK(gating_output.dim() == 2, "gating_output must be [num_tokens, num_experts]");
  DeviceGuard guard(gating_output.device());
  const torch::Tensor g = gating_output.contiguous();
  const int64_t T = g.size(0), E = g.size(1);
  auto w = torch::empty({T, topk}, g.options().dtype(torch::kFloat32));
  auto ids = torch::empty({T, topk}, g.options().dtype(torch::kInt32));
  torch::Tensor bias;
  if (correction_bias.has_value() && correction_bias->defined() && scoring_func == "sigmoid")
    bias = correction_bias->to(torch::kFloat32).contiguous();
  check(xllm_mi355_moe_fused_topk(p(g), dt(g), T, E, topk, renormalize ? 1 : 0, bias.defined() ? bias.data_ptr<float>() : nullptr,
                                  scoring_func == "softmax" ? 0 : 1, w.data_ptr<float>(), ids.data_ptr<int32_t>(), cur_stream()),
        "moe_fused_topk");
  return {w, ids};
}

std::tuple<torch::Tensor, torch::Tensor> moe_grouped_topk(const torch::Tensor& gating_output, int64_t topk,
                                                          int64_t num_expert_group, int64_t topk_group, bool renormalize,
                                                          const std::optional<torch::Tensor>& correction_bias,
                                                          const std::string& scoring_func, double routed_scaling_factor) {
  TORCH_CHECK(gating_output.defined() && gating_output.dim() == 2, "moe_grouped_topk: input must be [num_tokens, num_experts]");
  TORCH_CHECK(topk > 0, "moe_grouped_topk: topk must be positive");
  TORCH_CHECK(num_expert_group > 1, "moe_grouped_topk requires num_expert_group > 1");
  TORCH_CHECK(topk_group > 0 && topk_group <= num_expert_group, "moe_grouped_topk: 0 < topk_group <= num_expert_group");
  TORCH_CHECK(scoring_func == "softmax" || scoring_func == "sigmoid", "moe_grouped_topk: unsupported scoring function ",
              scoring_func);
  const bool has_bias = correction_bias.has_value() && correction_bias->defined();
  TORCH_CHECK(!has_bias || scoring_func == "sigmoid", "moe_grouped_topk: correction bias is supported only for sigmoid scoring");
  DeviceGuard guard(gating_output.device());
  const torch::Tensor g = gating_output.contiguous();
  const int64_t T = g.size(0), E = g.size(1);
  auto w = torch::empty({T, topk}, g.options().dtype(torch::kFloat32));
  auto ids = torch::empty({T, topk}, g.options().dtype(torch::kInt32));
  torch::Tensor bias;
  if (has_bias) bias = correction_bias->to(torch::kFloat32).contiguous();
  check(xllm_mi355_moe_grouped_topk(p(g), dt(g), T, E, topk, num_expert_group, topk_group, renormalize ? 1 : 0,
                                    bias.defined() ? bias.data_ptr<float>() : nullptr, scoring_func == "softmax" ? 0 : 1,
                                    static_cast<float>(routed_scaling_factor), w.data_ptr<float>(), ids.data_ptr<int32_t>(),
                                    cur_stream()),
        "moe_grouped_topk");
  return {w, ids};
}

std::tuple<torch::Tensor, torch::Tensor> moe_active_topk(const torch::Tensor& gating_output, int64_t topk,
                                                         int64_t num_expert_group, int64_t topk_group, bool renormalize,
                                                         const std::optional<torch::Tensor>& correction_bias,
                                                         const std::string& scoring_func, double routed_scaling_factor) {
  if (num_expert_group > 1)
    return moe_grouped_topk(gating_output, topk, num_expert_group, topk_group, renormalize, correction_bias, scoring_func,
                            routed_scaling_factor);
  return moe_fused_topk(gating_output, topk, renormalize, correction_bias, scoring_func);
}

std::vector<torch::Tensor> moe_gen_idx(const torch::Tensor& expert_id, int64_t expert_num) {
  TORCH_CHECK(expert_id.dim() == 2 && expert_id.scalar_type() == torch::kInt32, "expert_id must be int32 [num_tokens, topk]");
  DeviceGuard guard(expert_id.device());
  const torch::Tensor ids = expert_id.contiguous();
  const int64_t T = ids.size(0), topk = ids.size(1), n = T * topk;
  ensure_moe_scratch(ids, 4 * ((n + 1023) / 1024 + 1) * expert_num + 16 * (n / 256 + expert_num) + 64);
  auto src_dst = torch::empty({n}, ids.options()), dst_src = torch::empty({n}, ids.options());
  auto sizes = torch::empty({expert_num}, ids.options());
  check(xllm_mi355_moe_compute_index(ids.data_ptr<int32_t>(), T, topk, expert_num, src_dst.data_ptr<int32_t>(),
                                     dst_src.data_ptr<int32_t>(), sizes.data_ptr<int32_t>(), cur_stream()),
        "moe_gen_idx");
  return {src_dst, dst_src, sizes};
}

// the CUDA-header spellings the reference's USE_DCU branches of ops_api.cpp call (kernels/cuda/cuda_ops_api.h:
// moe_compute_index -> (src_dst, dst_src, expert_sizes); moe_combine_result(input, weights, N, topk)); with
// patches/xllm-use-mi355.patch those branches run on this backend through `namespace cuda = mi355`
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> moe_compute_index(const torch::Tensor& expert_id, int64_t expert_num) {
  auto v = moe_gen_idx(expert_id, expert_num);
  return {v[0], v[1], v[2]};
}

torch::Tensor moe_combine_result(const torch::Tensor& input, const torch::Tensor& reduce_weight, int64_t num_tokens,
                                 int32_t topk) {
  TORCH_CHECK(reduce_weight.size(0) == num_tokens && reduce_weight.size(1) == topk, "moe_combine_result: weights [N, topk]");
  return moe_combine_result(input, reduce_weight);
}

torch::Tensor moe_combine_result(const torch::Tensor& input, const torch::Tensor& reduce_weight) {
  TORCH_CHECK(input.dim() == 2 && reduce_weight.dim() == 2 && reduce_weight.numel() == input.size(0) &&
                  reduce_weight.scalar_type() == torch::kFloat32,
              "moe_combine_result: input [T*topk, H], reduce_weight float32 [T, topk]");
  DeviceGuard guard(input.device());
  const torch::Tensor x = input.contiguous(), w = reduce_weight.contiguous();
  const int64_t T = w.size(0), topk = w.size(1), H = x.size(1);
  auto out = torch::empty({T, H}, x.options());
  check(xllm_mi355_moe_combine(p(out), p(x), w.data_ptr<float>(), T, topk, H, dt(x), cur_stream()), "moe_combine_result");
  return out;
}

torch::Tensor moe_combine_result_sorted(const torch::Tensor& input_sorted, const torch::Tensor& reduce_weight,
                                        const torch::Tensor& gather_ids,
                                        const std::optional<torch::Tensor>& local_expert_sizes) {
  TORCH_CHECK(input_sorted.dim() == 2 && reduce_weight.dim() == 2 && reduce_weight.numel() == input_sorted.size(0) &&
                  gather_ids.numel() == input_sorted.size(0) && gather_ids.scalar_type() == torch::kInt32 &&
                  reduce_weight.scalar_type() == torch::kFloat32,
              "moe_combine_result_sorted: input [T*topk, H], reduce_weight float32 [T, topk], gather_ids int32 [T*topk]");
  DeviceGuard guard(input_sorted.device());
  const torch::Tensor x = input_sorted.contiguous(), w = reduce_weight.contiguous(), g = gather_ids.contiguous();
  const int64_t T = w.size(0), topk = w.size(1), H = x.size(1);
  auto out = torch::empty({T, H}, x.options());
  if (local_expert_sizes.has_value() && local_expert_sizes->defined()) {
    TORCH_CHECK(local_expert_sizes->scalar_type() == torch::kInt32, "local_expert_sizes must be int32");
    const torch::Tensor ls = local_expert_sizes->contiguous();
    check(xllm_mi355_moe_combine_sorted_local(p(out), p(x), g.data_ptr<int32_t>(), w.data_ptr<float>(), ls.data_ptr<int32_t>(),
                                              ls.numel(), T, topk, H, dt(x), cur_stream()),
          "moe_combine_result_sorted");
    return out;
  }
  check(xllm_mi355_moe_combine_sorted(p(out), p(x), g.data_ptr<int32_t>(), w.data_ptr<float>(), T, topk, H, dt(x), cur_stream()),
        "moe_combine_result_sorted");
  return out;
}

torch::Tensor group_gemm_gather(const torch::Tensor& input, const torch::Tensor& row_index, int64_t index_div,
                                const torch::Tensor& weight, const torch::Tensor& token_count) {
  DeviceGuard guard(input.device());
  TORCH_CHECK(input.dim() == 2 && weight.dim() == 3 && input.size(1) == weight.size(2) &&
                  row_index.scalar_type() == torch::kInt32 && token_count.scalar_type() == torch::kInt32,
              "group_gemm_gather shapes");
  const torch::Tensor x = input.contiguous(), w = weight.contiguous(), idx = row_index.contiguous();
  const int64_t E = w.size(0), N = w.size(1), K = w.size(2), rows = idx.numel();
  ensure_moe_scratch(x, 16 * (rows / 256 + E) + 64);
  auto out = torch::empty({rows, N}, x.options());
  const int rc = xllm_mi355_group_gemm_gather(p(x), x.size(0), idx.data_ptr<int32_t>(), index_div, p(w),
                                              token_count.data_ptr<int32_t>(), p(out), rows, E, N, K, dt(x), cur_stream());
  if (rc == XM_ERR_UNSUPPORTED) return torch::Tensor();
  check(rc, "group_gemm_gather");
  return out;
}

torch::Tensor group_gemm_w8a8(const torch::Tensor& input, const torch::Tensor& a_scale, const torch::Tensor& weight,
                              const torch::Tensor& b_scale, const torch::Tensor& token_count, torch::ScalarType out_dtype,
                              const std::optional<torch::Tensor>& gather_index, int64_t index_div) {
  DeviceGuard guard(input.device());
  TORCH_CHECK(input.dim() == 2 && weight.dim() == 3 && input.size(1) == weight.size(2) &&
                  input.scalar_type() == torch::kInt8 && weight.scalar_type() == torch::kInt8 &&
                  a_scale.scalar_type() == torch::kFloat32 && b_scale.scalar_type() == torch::kFloat32 &&
                  token_count.scalar_type() == torch::kInt32,
              "group_gemm_w8a8: int8 input [rows, K] / weight [E, N, K], float32 scales, int32 token_count");
  const bool gather = gather_index.has_value() && gather_index->defined();
  const torch::Tensor x = input.contiguous(), w = weight.contiguous(), as = a_scale.contiguous(), bs = b_scale.contiguous();
  torch::Tensor idx;
  if (gather) idx = gather_index->contiguous();
  const int64_t E = w.size(0), N = w.size(1), K = w.size(2), rows = gather ? idx.numel() : x.size(0);
  TORCH_CHECK(as.numel() == x.size(0) && bs.numel() == E * N, "group_gemm_w8a8: scale shapes");
  ensure_moe_scratch(x, 16 * (rows / 256 + E) + 64);
  auto out = torch::empty({rows, N}, x.options().dtype(out_dtype));
  check(xllm_mi355_group_gemm_w8a8(x.data_ptr<int8_t>(), x.size(0), as.data_ptr<float>(),
                                   gather ? idx.data_ptr<int32_t>() : nullptr, index_div, w.data_ptr<int8_t>(),
                                   bs.data_ptr<float>(), token_count.data_ptr<int32_t>(), p(out), rows, E, N, K, dt(out_dtype),
                                   cur_stream()),
        "group_gemm_w8a8");
  return out;
}

torch::Tensor mla_decode(const torch::Tensor& q, const torch::Tensor& k_cache, const torch::Tensor& seqlens_k,
                         const torch::Tensor& block_table, int64_t head_size_v, double softmax_scale, int64_t max_kv_len) {
  TORCH_CHECK(q.dim() == 3 && k_cache.dim() == 4 && k_cache.size(2) == 1 && k_cache.size(3) == q.size(2),
              "mla_decode: q [B, H, D], k_cache [n_blocks, block, 1, D]");
  TORCH_CHECK(seqlens_k.scalar_type() == torch::kInt32 && block_table.scalar_type() == torch::kInt32, "int32 metadata");
  DeviceGuard guard(q.device());
  const torch::Tensor qc = q.contiguous(), bt = block_table.contiguous();
  const int64_t B = qc.size(0), H = qc.size(1), D = qc.size(2);
  auto out = torch::empty({B, H, head_size_v}, qc.options());
  // split-KV partials: (head_size_v + 2) floats per (entry, head, split), at most 32 splits
  auto ws = torch::empty({B * H * 32 * (head_size_v + 2)}, qc.options().dtype(torch::kFloat32));
  check(xllm_mi355_mla_decode(p(qc), p(k_cache), p(out), seqlens_k.data_ptr<int32_t>(), bt.data_ptr<int32_t>(), bt.size(1),
                              B, H, D, head_size_v, k_cache.size(1), k_cache.size(0), max_kv_len, (float)softmax_scale,
                              dt(qc), ws.data_ptr(), (size_t)ws.numel() * 4, cur_stream()),
        "mla_decode");
  return out;
}

namespace flash_mla {
namespace {
// the input contract of flash_mla_adapter.cpp:48-101 (CHECKs there; TORCH_CHECK here), minus the page == 64 restriction
void check_dense_inputs(const DenseDecodeParams& p) {
  TORCH_CHECK(p.q_nope.defined() && p.q_pe.defined() && p.k_cache.defined() && p.seqlens_k.defined() && p.block_table.defined(),
              "flash_mla: q_nope, q_pe, k_cache, seqlens_k and block_table must be defined");
  const auto dev = p.k_cache.device();
  TORCH_CHECK(p.q_nope.device() == dev && p.q_pe.device() == dev && p.seqlens_k.device() == dev && p.block_table.device() == dev,
              "flash_mla: every tensor must be on the k_cache's device");
  TORCH_CHECK(p.q_nope.dim() == 4 && p.q_pe.dim() == 4, "flash_mla: q_nope / q_pe must be 4D [B,S_q,H_q,D]");
  TORCH_CHECK(p.k_cache.dim() == 4 && p.k_cache.size(2) == 1, "flash_mla: k_cache must be 4D [blocks,page,1,D] with one KV head");
  TORCH_CHECK(p.seqlens_k.dim() == 1 && p.block_table.dim() == 2, "flash_mla: seqlens_k [B], block_table [B,max_blocks]");
  TORCH_CHECK(p.k_cache.size(3) == p.q_nope.size(3) + p.q_pe.size(3), "flash_mla: k_cache last dim must equal q_nope.dim + q_pe.dim");
  TORCH_CHECK(p.q_nope.size(0) == p.q_pe.size(0) && p.q_nope.size(1) == p.q_pe.size(1) && p.q_nope.size(2) == p.q_pe.size(2),
              "flash_mla: q_nope / q_pe batch, seq and head counts must match");
  TORCH_CHECK(p.seqlens_k.size(0) == p.q_nope.size(0) && p.block_table.size(0) == p.q_nope.size(0), "flash_mla: batch mismatch");
  TORCH_CHECK(p.head_size_v > 0, "flash_mla: head_size_v must be positive");
  TORCH_CHECK(p.softmax_scale > 0.0F, "flash_mla: softmax_scale must be set by the MLA attention layer");
}
torch::Tensor as_i32(const torch::Tensor& t) { return (t.scalar_type() == torch::kInt32 ? t : t.to(torch::kInt32)).contiguous(); }
// scratch of the MLA kernels: split-KV partials ((head_size_v + 2) floats per (entry, head, split), at most 32 splits) and, for the
// prefill form, the per-query (sequence, visible keys) expansion in front of them
torch::Tensor mla_workspace(const torch::Tensor& like, int64_t entries, int64_t heads, int64_t head_size_v, int64_t q_tokens) {
  const int64_t idx_bytes = ((q_tokens * 2 * 4) + 255) / 256 * 256;
  const int64_t part_bytes = entries * heads * 32 * (head_size_v + 2) * 4;
  return torch::empty({idx_bytes + part_bytes}, like.options().dtype(torch::kUInt8));
}
}  // namespace

torch::Tensor dense_decode(DenseDecodeParams& params) {
  TORCH_CHECK(params.kind == DenseDecodeKind::kQNopePe, "flash_mla: only DenseDecodeKind::kQNopePe (BF16/FP16 split-q) is supported");
  check_dense_inputs(params);
  DeviceGuard guard(params.k_cache.device());
  const int64_t B = params.q_nope.size(0), Sq = params.q_nope.size(1), H = params.q_nope.size(2);
  const int64_t D = params.k_cache.size(3), page = params.k_cache.size(1);
  // the kernels take the query as one [.., D] row (nope || pe): one small copy (B * S_q * H * D elements) next to the cache read
  torch::Tensor q = torch::cat({params.q_nope, params.q_pe}, /*dim=*/-1).contiguous();   // [B, S_q, H, D]
  torch::Tensor lens = as_i32(params.seqlens_k), bt = as_i32(params.block_table);
  const int64_t max_kv = bt.size(1) * page;           // upper bound of every kv length: sizes the split-KV plan, never read back
  auto out = torch::empty({B, Sq, H, params.head_size_v}, q.options());
  if (B == 0 || Sq == 0) return out;
  if (Sq == 1) {
    auto ws = mla_workspace(q, B, H, params.head_size_v, 0);
    check(xllm_mi355_mla_decode(p(q), p(params.k_cache), p(out), lens.data_ptr<int32_t>(), bt.data_ptr<int32_t>(), bt.size(1), B, H,
                                D, params.head_size_v, page, params.k_cache.size(0), max_kv, params.softmax_scale, dt(q),
                                ws.data_ptr(), (size_t)ws.numel(), cur_stream()),
          "flash_mla::dense_decode");
    return out;
  }
  // S_q > 1: every sequence brings S_q query tokens, packed in batch order
  auto cu_q = torch::arange(0, (B + 1) * Sq, Sq, lens.options());
  auto ws = mla_workspace(q, B * Sq, H, params.head_size_v, B * Sq);
  check(xllm_mi355_mla_prefill(p(q), p(params.k_cache), p(out), cu_q.data_ptr<int32_t>(), lens.data_ptr<int32_t>(),
                               bt.data_ptr<int32_t>(), bt.size(1), B, B * Sq, H, D, params.head_size_v, page, params.k_cache.size(0),
                               max_kv, params.softmax_scale, params.is_causal ? 1 : 0, dt(q), ws.data_ptr(), (size_t)ws.numel(),
                               cur_stream()),
        "flash_mla::dense_decode (S_q > 1)");
  return out;
}

torch::Tensor prefill_paged(const torch::Tensor& q_nope, const torch::Tensor& q_pe, const torch::Tensor& k_cache,
                            const torch::Tensor& q_cu_seq_lens, const torch::Tensor& kv_seq_lens, const torch::Tensor& block_table,
                            int64_t head_size_v, double softmax_scale, bool is_causal) {
  TORCH_CHECK(q_nope.dim() == 3 && q_pe.dim() == 3 && q_nope.size(0) == q_pe.size(0) && q_nope.size(1) == q_pe.size(1),
              "flash_mla::prefill_paged: q_nope [T, H, kv_lora], q_pe [T, H, rope]");
  TORCH_CHECK(k_cache.dim() == 4 && k_cache.size(2) == 1 && k_cache.size(3) == q_nope.size(2) + q_pe.size(2),
              "flash_mla::prefill_paged: k_cache [blocks, page, 1, kv_lora + rope]");
  TORCH_CHECK(q_cu_seq_lens.dim() == 1 && kv_seq_lens.dim() == 1 && q_cu_seq_lens.size(0) == kv_seq_lens.size(0) + 1 &&
                  block_table.dim() == 2 && block_table.size(0) == kv_seq_lens.size(0),
              "flash_mla::prefill_paged: q_cu_seq_lens [B + 1], kv_seq_lens [B], block_table [B, max_blocks]");
  TORCH_CHECK(head_size_v > 0 && softmax_scale > 0.0, "flash_mla::prefill_paged: head_size_v and softmax_scale must be positive");
  DeviceGuard guard(k_cache.device());
  const int64_t T = q_nope.size(0), H = q_nope.size(1), B = kv_seq_lens.size(0), D = k_cache.size(3), page = k_cache.size(1);
  torch::Tensor q = torch::cat({q_nope, q_pe}, /*dim=*/-1).contiguous();
  torch::Tensor cu = as_i32(q_cu_seq_lens), lens = as_i32(kv_seq_lens), bt = as_i32(block_table);
  auto out = torch::empty({T, H, head_size_v}, q.options());
  if (T == 0 || B == 0) return out;
  auto ws = mla_workspace(q, T, H, head_size_v, T);
  check(xllm_mi355_mla_prefill(p(q), p(k_cache), p(out), cu.data_ptr<int32_t>(), lens.data_ptr<int32_t>(), bt.data_ptr<int32_t>(),
                               bt.size(1), B, T, H, D, head_size_v, page, k_cache.size(0), bt.size(1) * page, (float)softmax_scale,
                               is_causal ? 1 : 0, dt(q), ws.data_ptr(), (size_t)ws.numel(), cur_stream()),
        "flash_mla::prefill_paged");
  return out;
}

void store_latent_cache(const torch::Tensor& latent_cache, const torch::Tensor& slot_mapping, const torch::Tensor& k_cache) {
  TORCH_CHECK(latent_cache.dim() == 2 && k_cache.dim() == 4 && k_cache.size(2) == 1 && k_cache.size(3) == latent_cache.size(1) &&
                  slot_mapping.dim() == 1 && slot_mapping.size(0) == latent_cache.size(0),
              "flash_mla::store_latent_cache: latent [T, D], slots [T], k_cache [blocks, page, 1, D]");
  DeviceGuard guard(k_cache.device());
  torch::Tensor rows = latent_cache.contiguous(), slots = as_i32(slot_mapping);
  const int64_t T = rows.size(0), D = rows.size(1);
  check(xllm_mi355_reshape_paged_cache(slots.data_ptr<int32_t>(), p(rows), nullptr, p(k_cache), nullptr, T, /*n_kv_heads=*/1,
                                       /*head_dim=*/D, /*block_size=*/k_cache.size(1), /*n_blocks=*/k_cache.size(0),
                                       /*k_stride=*/D, /*v_stride=*/0, (int)rows.element_size(), cur_stream()),
        "flash_mla::store_latent_cache");
}
}  // namespace flash_mla

torch::Tensor build_block_table_from_paged_kv(const torch::Tensor& indptr, const torch::Tensor& indices) {
  DeviceGuard guard(indptr.device());
  const int64_t B = indptr.size(0) - 1, total = indices.size(0);
  auto table = torch::empty({B, total}, indptr.options().dtype(torch::kInt32));
  check(xllm_mi355_build_block_table_from_paged_kv(indptr.data_ptr<int32_t>(), indices.data_ptr<int32_t>(), (int32_t)B,
                                                   (int32_t)total, table.data_ptr<int32_t>(), cur_stream()),
        "build_block_table_from_paged_kv");
  return table;
}

torch::Tensor random_sample(const torch::Tensor& probs) {
  TORCH_CHECK(probs.dim() == 2 || probs.dim() == 3, "probs must be a 2D or 3D tensor");      // random_sample.hip:241
  TORCH_CHECK(probs.scalar_type() == torch::kFloat32, "probs must be float32");               // :253
  DeviceGuard guard(probs.device());
  torch::Tensor flat = probs.reshape({-1, probs.size(-1)}).contiguous();
  const int64_t batch = flat.size(0);
  auto out = torch::empty({batch}, flat.options().dtype(torch::kInt32));
  // Philox (seed, offset) of the device's default generator, advanced like get_seed_and_offset (random_sample.hip:58-74:
  // the offset is bumped by the rounded-up increment and the NEW offset seeds the kernel)
  uint64_t seed = 0, offset = 0;
  {
    at::Generator gen = at::cuda::detail::getDefaultCUDAGenerator(probs.device().index());
    std::lock_guard<std::mutex> lock(gen.mutex());
    auto* impl = at::check_generator<at::CUDAGeneratorImpl>(gen);
    seed = impl->current_seed();
    offset = impl->get_offset() + (uint64_t)((batch + 3) / 4 * 4);
    impl->set_offset(offset);
  }
  check(xllm_mi355_random_sample(flat.data_ptr<float>(), out.data_ptr<int32_t>(), batch, flat.size(1), nullptr, seed, offset,
                                 cur_stream()),
        "random_sample");
  if (probs.dim() == 3) return out.view({probs.size(0), probs.size(1)});
  return out;
}

torch::Tensor rejection_sample(const torch::Tensor& draft_token_ids, const torch::Tensor& num_draft_tokens,
                               const torch::Tensor& cu_num_draft_tokens, const std::optional<torch::Tensor>& draft_probs,
                               const torch::Tensor& target_probs, const torch::Tensor& bonus_token_ids,
                               const torch::Tensor& uniform_rand, const torch::Tensor& uniform_probs,
                               int64_t max_spec_len) {
  (void)max_spec_len;
  TORCH_CHECK(draft_probs.has_value(), "rejection_sample requires dense draft_probs");        // rejection_sample.hip:151
  DeviceGuard guard(target_probs.device());
  auto f32 = [](const torch::Tensor& t) { return t.to(torch::kFloat32).contiguous(); };
  torch::Tensor dp = f32(*draft_probs), tp = f32(target_probs), ur = f32(uniform_rand), up = f32(uniform_probs);
  TORCH_CHECK(dp.dim() == 2 && dp.sizes() == tp.sizes() && up.sizes() == tp.sizes(), "probs must be [drafts, vocab]");
  const int64_t batch = num_draft_tokens.size(0);
  TORCH_CHECK(cu_num_draft_tokens.numel() == batch && bonus_token_ids.numel() == batch);
  TORCH_CHECK(ur.numel() == draft_token_ids.numel() && dp.size(0) == draft_token_ids.numel());
  auto out = torch::empty({batch + draft_token_ids.size(0)}, draft_token_ids.options().dtype(torch::kInt32));
  if (batch == 0) return out;
  check(xllm_mi355_rejection_sample(draft_token_ids.contiguous().data_ptr<int32_t>(), num_draft_tokens.data_ptr<int32_t>(),
                                    cu_num_draft_tokens.data_ptr<int32_t>(), dp.data_ptr<float>(), tp.data_ptr<float>(),
                                    bonus_token_ids.data_ptr<int32_t>(), ur.data_ptr<float>(), up.data_ptr<float>(), batch,
                                    tp.size(1), out.data_ptr<int32_t>(), cur_stream()),
        "rejection_sample");
  return out;
}

void update_llm_decode_metadata(const LlmDecodeMetadataUpdateParams& params, void* stream) {
  check(xllm_mi355_decode_metadata_update(&params, stream), "update_llm_decode_metadata");
}

torch::Tensor prefill_attention(const torch::Tensor& q, const torch::Tensor& k, const torch::Tensor& v,
                                const torch::Tensor& cu_q, const torch::Tensor& cu_k, int64_t max_q_len, double scale,
                                bool is_causal, int64_t window_left, std::optional<torch::Tensor> out) {
  DeviceGuard guard(q.device());
  const int64_t Tq = q.size(0), nq = q.size(1), d = q.size(2), nkv = k.size(1);
  torch::Tensor o = out.has_value() ? *out : torch::empty({Tq, nq * d}, q.options());
  check(xllm_mi355_prefill_attention(p(q), p(k), p(v), p(o), cu_q.data_ptr<int32_t>(), cu_k.data_ptr<int32_t>(),
                                     cu_q.numel() - 1, nq, nkv, d, q.stride(0), k.stride(0), v.stride(0), max_q_len,
                                     (float)scale, is_causal ? 1 : 0, window_left, dt(q), cur_stream()),
        "prefill_attention");
  return o;
}

torch::Tensor paged_attention(const torch::Tensor& q, const torch::Tensor& k_cache, const torch::Tensor& v_cache,
                              const std::optional<torch::Tensor>& cu_q, const torch::Tensor& kv_seq_lens,
                              const torch::Tensor& block_table, int64_t max_q_len, int64_t max_kv_len, double scale,
                              bool is_causal, int64_t window_left, std::optional<torch::Tensor> out) {
  DeviceGuard guard(q.device());
  const int64_t Tq = q.size(0), nq = q.size(1), d = q.size(2);
  const int64_t n_blocks = k_cache.size(0), bs = k_cache.size(1), nkv = k_cache.size(2), B = kv_seq_lens.numel();
  torch::Tensor o = out.has_value() ? *out : torch::empty({Tq, nq * d}, q.options());
  const size_t ws_bytes = xllm_mi355_paged_attention_workspace_bytes(B, nq, d, max_q_len, Tq);
  torch::Tensor ws = torch::empty({(int64_t)std::max<size_t>(ws_bytes, 1)}, q.options().dtype(torch::kUInt8));
  auto bt = block_table.contiguous();
  check(xllm_mi355_paged_attention(p(q), p(k_cache), p(v_cache), p(o),
                                   cu_q.has_value() ? cu_q->data_ptr<int32_t>() : nullptr,
                                   kv_seq_lens.data_ptr<int32_t>(), bt.data_ptr<int32_t>(), bt.size(1), B, Tq, nq, nkv,
                                   d, bs, n_blocks, q.stride(0), max_q_len, max_kv_len, (float)scale,
                                   is_causal ? 1 : 0, window_left, dt(q), p(ws), ws_bytes, cur_stream()),
        "paged_attention");
  return o;
}

}  // namespace xllm::kernel::mi355
