// layers/mi355/attention.h of a USE_MI355 build: the xllm::layer::AttentionImpl contract of
// xllm/core/layers/dcu/attention.h:31-51 implemented on xllm::kernel::mi355 (dispatch as
// layers/dcu/flash_attention.cpp:291-376). Self-contained mirror types are used here because the reference's
// AttentionMetadata / KVCache headers pull in the whole framework; field names are the reference's.
#pragma once
#include <torch/torch.h>

#include <optional>
#include <tuple>

namespace xllm::layer::mi355 {

struct AttentionMetadata {  // layers/common/attention_metadata.h:73-186 (subset used on CUDA/DCU)
  torch::Tensor q_cu_seq_lens, kv_cu_seq_lens, kv_seq_lens, slot_mapping, block_table;
  int64_t max_query_len = 1, max_seq_len = 0;
  bool is_prefill = false, is_chunked_prefill = false;
  bool is_causal() const { return is_prefill || is_chunked_prefill; }  // attention_metadata_builder.cpp:240-241
};

struct KVCache {  // framework/kv_cache: [n_blocks, block_size, n_kv_heads_local, head_dim]
  torch::Tensor k_cache, v_cache;
  torch::Tensor get_k_cache() const { return k_cache; }
  torch::Tensor get_v_cache() const { return v_cache; }
};

class AttentionImpl {
 public:
  AttentionImpl(int64_t num_heads, int64_t head_size, float scale, int64_t num_kv_heads, int64_t sliding_window)
      : num_heads_(num_heads), head_size_(head_size), scale_(scale), num_kv_heads_(num_kv_heads),
        window_left_(sliding_window > 0 ? sliding_window : -1) {}  // flash_attention.cpp:257

  std::tuple<torch::Tensor, std::optional<torch::Tensor>> forward(const AttentionMetadata& md, torch::Tensor& query,
                                                                  torch::Tensor& key, torch::Tensor& value,
                                                                  KVCache& kv_cache);

 private:
  int64_t num_heads_, head_size_;
  float scale_;
  int64_t num_kv_heads_, window_left_;
};

}  // namespace xllm::layer::mi355
