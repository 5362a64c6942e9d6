// layers/mi355/attention.h of a USE_MI355 build: xllm::layer::AttentionImpl with the declaration of
// xllm/core/layers/dcu/attention.h:31-51 (same constructor, same forward), implemented on xllm::kernel::mi355 with the
// dispatch of layers/dcu/flash_attention.cpp:291-376. It is written against the reference's OWN types: inside xLLM the two
// includes below resolve to the real headers; outside (this repository's build and tests) shim/stub/ provides field-for-field
// stand-ins, and mi355_attention.cpp static_asserts the members it uses against whichever header it sees.
#pragma once
#include <torch/torch.h>

#include <optional>
#include <tuple>

#include "framework/kv_cache/kv_cache.h"
#include "layers/common/attention_metadata.h"

namespace xllm {
namespace layer {

class AttentionImpl final : public torch::nn::Module {
 public:
  AttentionImpl() = default;
  AttentionImpl(int64_t num_heads, int64_t head_size, float scale, int64_t num_kv_heads, int64_t sliding_window);

  std::tuple<torch::Tensor, std::optional<torch::Tensor>> forward(const AttentionMetadata& attn_metadata,
                                                                  torch::Tensor& query, torch::Tensor& key,
                                                                  torch::Tensor& value, KVCache& kv_cache);

 private:
  int64_t num_heads_ = 0, head_size_ = 0;
  float scale_ = 1.0f;
  int64_t num_kv_heads_ = 0, window_left_ = -1;
};
TORCH_MODULE(Attention);

}  // namespace layer
}  // namespace xllm
