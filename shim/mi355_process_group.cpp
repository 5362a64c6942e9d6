// mi355_process_group.cpp -- see mi355_process_group.h. Host C++ only; the kernels are csrc/allreduce.hip behind the C ABI.
#include "mi355_process_group.h"

#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include <chrono>
#include <set>
#include <sstream>
#include <torch/csrc/distributed/c10d/TCPStore.hpp>

#include "../include/xllm_mi355.h"

namespace xllm {
namespace mi355 {
namespace {

void* cur_stream() { return static_cast<void*>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()); }
using DeviceGuard = c10::hip::OptionalHIPGuardMasqueradingAsCUDA;

int xm_dtype(const torch::Tensor& t) {
  switch (t.scalar_type()) {
    case torch::kFloat32: return XM_F32;
    case torch::kBFloat16: return XM_BF16;
    case torch::kFloat16: return XM_F16;
    default: return -1;
  }
}
void* p(const torch::Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }

void put(const c10::intrusive_ptr<c10d::Store>& s, const std::string& key, const std::string& v) {
  s->set(key, std::vector<uint8_t>(v.begin(), v.end()));
}
std::string get(const c10::intrusive_ptr<c10d::Store>& s, const std::string& key) {  // blocks until the key exists (store timeout)
  const std::vector<uint8_t> v = s->get(key);
  return std::string(v.begin(), v.end());
}

// identity of the physical GPU behind a torch device: ranks that report the same string share one GPU
std::string device_identity(int index) {
  char host[256] = {0};
  gethostname(host, sizeof(host) - 1);
  std::ostringstream o;
  o << host << ":";
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, index) == hipSuccess) {
    o << prop.pciDomainID << "." << prop.pciBusID << "." << prop.pciDeviceID;
  } else {
    o << "dev" << index;
  }
  return o.str();
}

}  // namespace

c10::intrusive_ptr<c10d::Store> create_tcp_store(const std::string& host, int port, int rank, int world) {
  c10d::TCPStoreOptions o;
  o.isServer = (rank == 0);
  o.port = static_cast<uint16_t>(port);
  o.numWorkers = world;
  o.waitWorkers = false;
  return c10::make_intrusive<c10d::TCPStore>(host, o);
}

// every rank publishes its verdict for this round and reads everybody's: *mine becomes "" iff all ranks said "", else the joined
// reasons (identical on every rank). get() blocks until the key exists, so the exchange is a barrier as well.
void OneShotAllReduce::agree(const std::string& round, std::string* mine) {
  const std::string base = prefix_ + "/" + round + "/";
  put(store_, base + std::to_string(rank_), *mine);
  std::set<std::string> bad;
  for (int r = 0; r < world_; ++r) {
    const std::string v = get(store_, base + std::to_string(r));
    if (!v.empty()) bad.insert("rank " + std::to_string(r) + ": " + v);
  }
  std::string all;
  for (const auto& b : bad) all += (all.empty() ? "" : "; ") + b;
  *mine = all;
}

std::shared_ptr<OneShotAllReduce> OneShotAllReduce::create(const c10::intrusive_ptr<c10d::Store>& store,
                                                           const std::string& prefix, int rank, int world,
                                                           const torch::Device& device, const Options& opt, std::string* note) {
  auto set_note = [&](const std::string& s) { if (note) *note = s; };
  if (world <= 1) { set_note("world size 1: nothing to reduce"); return nullptr; }
  std::shared_ptr<OneShotAllReduce> ar(new OneShotAllReduce());
  ar->store_ = store;
  ar->prefix_ = prefix;
  ar->rank_ = rank;
  ar->world_ = world;
  ar->device_ = device;
  ar->opt_ = opt;
  ar->peers_.assign(world, nullptr);
  // Every exchange below runs UNCONDITIONALLY on every rank: a rank-local failure is recorded in `err` and travels with the next
  // exchange; it never makes this rank skip a step its peers sit in (a rank that threw here would leave them blocked on the store).
  std::string err;
  std::string handle(XLLM_MI355_IPC_HANDLE_BYTES, '\0');
  const size_t total = xllm_mi355_oneshot_allreduce_buffer_bytes(opt.max_bytes);
  try {
    DeviceGuard guard(device);
    bool exported = false;
    for (int first_kind = 0; first_kind <= 2 && !exported; ++first_kind) {  // fine-grained, uncached, plain: the first exportable kind
      void* ptr = nullptr;
      int kind = first_kind;
      if (xllm_mi355_ipc_alloc(total, &ptr, &kind) != 0) { err = "ipc_alloc failed"; break; }
      if (xllm_mi355_ipc_get_handle(ptr, handle.data()) == 0) {
        ar->own_ = ptr;
        ar->kind_ = kind;
        exported = true;
      } else {
        xllm_mi355_ipc_free(ptr);
        if (kind >= 2) break;
        first_kind = kind;  // the allocator may already have fallen through to a later kind
      }
    }
    if (!exported && err.empty()) err = "hipIpcGetMemHandle failed for every memory kind";
  } catch (const std::exception& e) {
    err = std::string("buffer set-up failed: ") + e.what();
  }
  // exchange 1: (handle, memory kind, GPU identity, error) of every rank
  const std::string ident = device_identity(device.has_index() ? device.index() : 0);
  {
    std::ostringstream rec;
    rec << ar->kind_ << "\n" << ident << "\n" << err << "\n";
    put(store, prefix + "/rec/" + std::to_string(rank), rec.str() + handle);
  }
  std::vector<std::string> handles(world), idents(world);
  std::vector<int> kinds(world, -1);
  std::string refused;
  for (int r = 0; r < world; ++r) {
    const std::string v = get(store, prefix + "/rec/" + std::to_string(r));
    std::istringstream in(v);
    std::string k, id, e;
    std::getline(in, k);
    std::getline(in, id);
    std::getline(in, e);
    const size_t hdr = k.size() + id.size() + e.size() + 3;
    kinds[r] = std::atoi(k.c_str());
    idents[r] = id;
    handles[r] = v.size() >= hdr + XLLM_MI355_IPC_HANDLE_BYTES ? v.substr(hdr, XLLM_MI355_IPC_HANDLE_BYTES) : std::string();
    if (!e.empty() || kinds[r] < 0) refused += (refused.empty() ? "" : "; ") + ("rank " + std::to_string(r) + ": " + (e.empty() ? "no exportable buffer" : e));
  }
  std::string local;
  if (refused.empty()) {
    try {
      DeviceGuard guard(device);
      for (int r = 0; r < world; ++r) {
        if (r == rank) { ar->peers_[r] = ar->own_; continue; }
        void* q = nullptr;
        if (xllm_mi355_ipc_open_handle(handles[r].data(), &q) != 0) { local = "ipc_open_handle(rank " + std::to_string(r) + ") failed"; break; }
        ar->peers_[r] = q;
        ar->opened_.push_back(q);
      }
    } catch (const std::exception& e) {
      local = std::string("peer mapping failed: ") + e.what();
    }
    const std::set<std::string> distinct(idents.begin(), idents.end());
    // one rank per GPU: the fused kernel may use 256 blocks (one row per block for a 256-row decode message); ranks that share a
    // GPU keep 64 -- every block waits for its peers' blocks, so all grids must be co-resident. `idents` is the same list on every
    // rank, so is the choice.
    if ((int)distinct.size() == world) ar->grid_limit_ = 256;
    bool any_plain = false;
    for (int k : kinds) any_plain = any_plain || k >= 2;
    if (any_plain && distinct.size() != 1 && local.empty())
      local = "only plain hipMalloc memory could be exported on some rank: not guaranteed visible to a peer GPU while the kernel runs";
    try {
      const auto i32 = torch::TensorOptions().dtype(torch::kInt32).device(device);
      ar->state_ = torch::zeros({2}, i32);   // epoch, blocks-done counter
      ar->status_ = torch::zeros({1}, i32);
      c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().synchronize();
    } catch (const std::exception& e) {
      if (local.empty()) local = std::string("state allocation failed: ") + e.what();
    }
  } else {
    local = refused;   // identical on every rank already; the agreement round below still runs
  }
  // exchange 2: either every rank has every buffer mapped before the first launch, or nobody launches
  ar->agree("mapped", &local);
  if (local.empty() && opt.self_test) {
    local = ar->self_test() ? "" : "self-test failed (a checked message came back wrong or a wait timed out)";
    ar->agree("selftest", &local);
  }
  if (!local.empty()) {
    set_note(local);
    ar->close();     // every rank takes this branch together
    return nullptr;
  }
  set_note("ok");
  return ar;
}

void OneShotAllReduce::unmap() {
  for (void* q : opened_) xllm_mi355_ipc_close_handle(q);
  opened_.clear();
  if (own_) { xllm_mi355_ipc_free(own_); own_ = nullptr; }
}

void OneShotAllReduce::close() {
  if (!store_) return;
  if (state_.defined()) {
    DeviceGuard guard(device_);
    (void)hipDeviceSynchronize();
  }
  std::string none;
  agree("close" + std::to_string(round_++), &none);
  unmap();
  store_.reset();
}

OneShotAllReduce::~OneShotAllReduce() { unmap(); }

bool OneShotAllReduce::takes(const torch::Tensor& x) const {
  if (suspended_ > 0 || !own_ || !x.defined() || !x.is_cuda() || x.device() != device_) return false;
  const int64_t n = x.numel() * (int64_t)x.element_size();
  return xm_dtype(x) >= 0 && n > 0 && (size_t)n <= opt_.max_bytes && n % 16 == 0;
}

void OneShotAllReduce::allreduce(torch::Tensor& x) {
  TORCH_CHECK(takes(x), "OneShotAllReduce::allreduce: the message is outside the kernel's envelope (ask takes() first)");
  DeviceGuard guard(device_);
  torch::Tensor t = x;
  const bool staged = !x.is_contiguous() || reinterpret_cast<uintptr_t>(x.data_ptr()) % 16 != 0;
  if (staged) t = x.contiguous().clone();   // fresh allocations are 256-byte aligned
  const int rc = xllm_mi355_oneshot_allreduce(t.data_ptr(), t.numel(), xm_dtype(t), peers_.data(), rank_, world_, opt_.max_bytes,
                                              reinterpret_cast<uint32_t*>(state_.data_ptr()), status_.data_ptr<int>(),
                                              opt_.timeout_s, cur_stream());
  TORCH_CHECK(rc == 0, "oneshot_allreduce: ", xllm_mi355_strerror(rc));
  if (staged) x.copy_(t);
}

std::optional<std::pair<torch::Tensor, torch::Tensor>> OneShotAllReduce::allreduce_add_rms_norm(
    const torch::Tensor& partial, torch::Tensor& residual, const torch::Tensor& weight, double eps, bool quantize,
    torch::Tensor* sum_out) {
  if (partial.dim() != 2 || !takes(partial)) return std::nullopt;
  const int64_t M = partial.size(0), H = partial.size(1);
  if (H % 8 != 0 || H > 16384) return std::nullopt;
  TORCH_CHECK(residual.sizes() == partial.sizes() && residual.is_contiguous() && residual.scalar_type() == partial.scalar_type(),
              "allreduce_add_rms_norm: residual [M, H] contiguous, same dtype as the partial sums");
  DeviceGuard guard(device_);
  const torch::Tensor pc = (partial.is_contiguous() && reinterpret_cast<uintptr_t>(partial.data_ptr()) % 16 == 0)
                               ? partial : partial.contiguous().clone();
  torch::Tensor q, qs, n16, ysum;
  const auto like = partial.options();
  if (quantize) {
    q = torch::empty({M, H}, like.dtype(torch::kInt8));
    qs = torch::empty({M}, like.dtype(torch::kFloat32));
  } else {
    n16 = torch::empty({M, H}, like);
  }
  if (sum_out) ysum = torch::empty({M, H}, like);
  const int rc = xllm_mi355_oneshot_allreduce_add_rms_norm(
      pc.data_ptr(), residual.data_ptr(), weight.data_ptr(), (float)eps, p(n16), quantize ? q.data_ptr<int8_t>() : nullptr,
      quantize ? qs.data_ptr<float>() : nullptr, p(ysum), M, H, xm_dtype(partial), peers_.data(), rank_, world_, opt_.max_bytes,
      reinterpret_cast<uint32_t*>(state_.data_ptr()), status_.data_ptr<int>(), opt_.timeout_s, grid_limit_, cur_stream());
  TORCH_CHECK(rc == 0, "oneshot_allreduce_add_rms_norm: ", xllm_mi355_strerror(rc));
  if (sum_out) *sum_out = ysum;
  return quantize ? std::make_pair(q, qs) : std::make_pair(n16, torch::Tensor());
}

std::optional<std::pair<torch::Tensor, torch::Tensor>> OneShotAllReduce::matmul_allreduce_add_rms_norm(
    const torch::Tensor& a_q, const torch::Tensor& a_scale, const torch::Tensor& w_packed, const torch::Tensor& w_scale,
    const std::optional<torch::Tensor>& bias, torch::Tensor& residual, const torch::Tensor& weight, double eps, bool quantize,
    torch::Tensor* sum_out) {
  if (suspended_ > 0 || !own_ || a_q.dim() != 2) return std::nullopt;
  const int64_t M = a_q.size(0), K = a_q.size(1), N = w_scale.numel();
  TORCH_CHECK(residual.dim() == 2 && residual.size(0) == M && residual.size(1) == N && residual.is_contiguous() && a_q.is_contiguous(),
              "matmul_allreduce_add_rms_norm: a [M, K] and residual [M, N] contiguous");
  // the envelope is decided by M, N, K alone (the same on every rank), BEFORE anything is launched
  if ((size_t)(M * N * (int64_t)residual.element_size()) > opt_.max_bytes || M > 512 || N % 8 != 0 || N > 16384) return std::nullopt;
  DeviceGuard guard(device_);
  const size_t need = (size_t)8 * M * N * 4;   // up to eight K-slice slabs of exact int32 sums
  if (!slab_ws_.defined() || (size_t)slab_ws_.numel() < need)
    slab_ws_ = torch::empty({(int64_t)std::max(need, (size_t)32 << 20)}, a_q.options().dtype(torch::kUInt8));
  torch::Tensor q, qs, n16, ysum;
  const auto like = residual.options();
  if (quantize) {
    q = torch::empty({M, N}, like.dtype(torch::kInt8));
    qs = torch::empty({M}, like.dtype(torch::kFloat32));
  } else {
    n16 = torch::empty({M, N}, like);
  }
  if (sum_out) ysum = torch::empty({M, N}, like);
  const int rc = xllm_mi355_scaled_matmul_oneshot_allreduce_add_rms_norm(
      a_q.data_ptr<int8_t>(), reinterpret_cast<const int8_t*>(w_packed.data_ptr()), a_scale.data_ptr<float>(),
      w_scale.data_ptr<float>(), bias.has_value() && bias->defined() ? bias->data_ptr() : nullptr, residual.data_ptr(),
      weight.data_ptr(), (float)eps, p(n16), quantize ? q.data_ptr<int8_t>() : nullptr, quantize ? qs.data_ptr<float>() : nullptr,
      p(ysum), M, N, K, xm_dtype(residual), slab_ws_.data_ptr(), (size_t)slab_ws_.numel(), peers_.data(), rank_, world_,
      opt_.max_bytes, reinterpret_cast<uint32_t*>(state_.data_ptr()), status_.data_ptr<int>(), opt_.timeout_s, grid_limit_,
      cur_stream());
  if (rc == XM_ERR_UNSUPPORTED || rc == XM_ERR_WORKSPACE) return std::nullopt;   // the packed GEMM declines the shape: nothing written
  TORCH_CHECK(rc == 0, "scaled_matmul_oneshot_allreduce_add_rms_norm: ", xllm_mi355_strerror(rc));
  if (sum_out) *sum_out = ysum;
  return quantize ? std::make_pair(q, qs) : std::make_pair(n16, torch::Tensor());
}

bool OneShotAllReduce::healthy() {
  if (!status_.defined()) return false;
  return status_.item<int>() == 0;   // (synchronises)
}

// one checked message per size class through every kernel form: rank r sends (r + 1) * pattern, the sum must be
// pattern * W (W + 1) / 2 exactly (small integers: exact in bf16). Catches its own exceptions; the caller agrees on the verdict.
bool OneShotAllReduce::self_test() {
  const double saved = opt_.timeout_s;
  opt_.timeout_s = std::min(saved, 5.0);
  bool ok = true;
  try {
    DeviceGuard guard(device_);
    const float tri = (float)(world_ * (world_ + 1) / 2);
    const auto bf = torch::TensorOptions().dtype(torch::kBFloat16).device(device_);
    const auto i64 = torch::TensorOptions().dtype(torch::kInt64).device(device_);
    for (int64_t n : {(int64_t)8, (int64_t)4096, (int64_t)256 * 3584, (int64_t)(opt_.max_bytes / 2)}) {
      if ((size_t)n * 2 > opt_.max_bytes || n <= 0) continue;
      const torch::Tensor pat = torch::arange(n, i64).remainder(7).to(torch::kBFloat16);
      torch::Tensor x = pat * (float)(rank_ + 1);
      allreduce(x);
      ok = ok && torch::equal(x, pat * tri);
    }
    auto fused = [&](int64_t M, int64_t H) {
      const torch::Tensor pat = torch::arange(M * H, i64).view({M, H}).remainder(5).to(torch::kBFloat16);
      torch::Tensor res = torch::ones({M, H}, bf), ysum;
      const torch::Tensor w = torch::ones({H}, bf);
      const auto got = allreduce_add_rms_norm(pat * (float)(rank_ + 1), res, w, 1e-6, false, &ysum);
      ok = ok && got.has_value() && torch::equal(ysum, pat * tri) && torch::equal(res, pat * tri + 1.0f) &&
           torch::isfinite(got->first.to(torch::kFloat32)).all().item<bool>();
    };
    fused(8, 512);
    // one row per block (grid_limit = 256, every rank on a GPU of its own): production decode sends M = 256 rows, which uses flag
    // rows 64 .. 255 and relies on 256 co-resident blocks per rank -- the 8-row message above touches neither. (A buffer too small
    // for the message is a rank-invariant condition: skipped by all.)
    if (grid_limit_ > 64 && (size_t)256 * 512 * 2 <= opt_.max_bytes) fused(256, 512);
    {  // the GEMM-fed form: every rank multiplies ones by ones (sum = K) with a_scale = (rank + 1) / K -> partial = rank + 1
      const int64_t M = 8, K = 512, N = 512;
      const auto i8 = torch::TensorOptions().dtype(torch::kInt8).device(device_);
      const auto f32 = torch::TensorOptions().dtype(torch::kFloat32).device(device_);
      const torch::Tensor a = torch::ones({M, K}, i8), wrow = torch::ones({N, K}, i8);
      torch::Tensor wp = torch::empty({N, K}, i8);
      if (xllm_mi355_pack_weight_i8(wrow.data_ptr<int8_t>(), wp.data_ptr<int8_t>(), N, K, cur_stream()) == 0) {
        const torch::Tensor a_s = torch::full({M}, (float)(rank_ + 1) / (float)K, f32), w_s = torch::ones({N}, f32);
        torch::Tensor res = torch::ones({M, N}, bf), ysum;
        const torch::Tensor w = torch::ones({N}, bf);
        const auto got = matmul_allreduce_add_rms_norm(a, a_s, wp, w_s, std::nullopt, res, w, 1e-6, false, &ysum);
        if (got.has_value())   // (nullopt: the packed GEMM declined the shape -- the same on every rank)
          ok = ok && (ysum == tri).all().item<bool>() && (res == tri + 1.0f).all().item<bool>();
      }
    }
    ok = ok && healthy();
  } catch (const std::exception&) {
    ok = false;
  }
  opt_.timeout_s = saved;
  return ok;
}

void ProcessGroupMi355::allreduce(torch::Tensor& input) {
  if (world_ <= 1) return;   // parallel_state.cpp:184-186
  if (oneshot_ && oneshot_->takes(input)) {
    oneshot_->allreduce(input);   // a plain kernel on the current stream: also inside a graph capture
    return;
  }
  TORCH_CHECK(static_cast<bool>(fallback_), "ProcessGroupMi355::allreduce: the message is outside the one-shot envelope and no backend is attached");
  fallback_(input);
}

}  // namespace mi355
}  // namespace xllm
