#include "mi355_attention.h"

#include "mi355_ops_api.h"

namespace xllm::layer::mi355 {

std::tuple<torch::Tensor, std::optional<torch::Tensor>> AttentionImpl::forward(const AttentionMetadata& md,
                                                                               torch::Tensor& query, torch::Tensor& key,
                                                                               torch::Tensor& value, KVCache& kv_cache) {
  namespace k = xllm::kernel::mi355;
  auto q = query.unflatten(-1, {num_heads_, head_size_});
  auto kk = key.unflatten(-1, {num_kv_heads_, head_size_});
  auto vv = value.unflatten(-1, {num_kv_heads_, head_size_});
  auto kc = kv_cache.get_k_cache(), vc = kv_cache.get_v_cache();
  k::reshape_paged_cache(md.slot_mapping, kk, vv, kc, vc);  // flash_attention.cpp:310-318
  torch::Tensor out;
  if (md.is_prefill) {
    out = k::prefill_attention(q, kk, vv, md.q_cu_seq_lens, md.kv_cu_seq_lens, md.max_query_len, scale_, true,
                               window_left_);
  } else if (md.is_chunked_prefill) {
    out = k::paged_attention(q, kc, vc, md.q_cu_seq_lens, md.kv_seq_lens, md.block_table, md.max_query_len,
                             md.max_seq_len, scale_, true, window_left_);
  } else {
    out = k::paged_attention(q, kc, vc, std::nullopt, md.kv_seq_lens, md.block_table, 1, md.max_seq_len, scale_, false,
                             window_left_);
  }
  return {out, std::nullopt};
}

}  // namespace xllm::layer::mi355
