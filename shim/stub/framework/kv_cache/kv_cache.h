// STUB of xllm/core/framework/kv_cache/kv_cache.h:29-70 (the two accessors the attention class uses; the real class owns a
// KVCacheImpl and is move-only). See shim/stub/layers/common/attention_metadata.h for how the stubs are used.
#pragma once
#include <torch/torch.h>

namespace xllm {

class KVCache final {
 public:
  KVCache() = default;
  KVCache(torch::Tensor k, torch::Tensor v) : k_(std::move(k)), v_(std::move(v)) {}
  KVCache(const KVCache&) = delete;
  KVCache& operator=(const KVCache&) = delete;
  KVCache(KVCache&&) noexcept = default;
  KVCache& operator=(KVCache&&) noexcept = default;
  torch::Tensor get_k_cache() const { return k_; }
  torch::Tensor get_v_cache() const { return v_; }

 private:
  torch::Tensor k_, v_;
};

}  // namespace xllm
