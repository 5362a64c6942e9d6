// inside the xLLM tree this file is xllm/core/layers/mi355/attention.cpp next to attention.h (= mi355_attention.h, INTEGRATION.md
// section 3); in this repository the header keeps its own name
#if __has_include("layers/mi355/attention.h")
#include "layers/mi355/attention.h"
#else
#include "mi355_attention.h"
#endif

#include <type_traits>

// the reference's host code for piecewise HIP-graph capture (kernels/dcu/attention_runner.{h,cpp}, compiled into a USE_MI355
// build unchanged); outside the xLLM tree shim/stub/kernels/dcu/ stands in
#include "kernels/dcu/attention_runner.h"
#if __has_include("kernels/mi355/mi355_ops_api.h")
#include "kernels/mi355/mi355_ops_api.h"
#else
#include "mi355_ops_api.h"
#endif

namespace xllm {
namespace layer {

// every member of the reference's types this file touches, with the type it is used as
// (layers/common/attention_metadata.h:73-186, framework/kv_cache/kv_cache.h:46-47)
#define MI355_CHECK_FIELD(S, F, T) static_assert(std::is_same_v<decltype(S::F), T>, #S "::" #F " is not a " #T)
MI355_CHECK_FIELD(AttentionMetadata, q_cu_seq_lens, torch::Tensor);
MI355_CHECK_FIELD(AttentionMetadata, kv_cu_seq_lens, torch::Tensor);
MI355_CHECK_FIELD(AttentionMetadata, kv_seq_lens, torch::Tensor);
MI355_CHECK_FIELD(AttentionMetadata, block_table, torch::Tensor);
MI355_CHECK_FIELD(AttentionMetadata, slot_mapping, torch::Tensor);
MI355_CHECK_FIELD(AttentionMetadata, max_query_len, int64_t);
MI355_CHECK_FIELD(AttentionMetadata, max_seq_len, int64_t);
MI355_CHECK_FIELD(AttentionMetadata, is_prefill, bool);
MI355_CHECK_FIELD(AttentionMetadata, is_chunked_prefill, bool);
MI355_CHECK_FIELD(AttentionMetadata, is_causal, bool);
#undef MI355_CHECK_FIELD
static_assert(std::is_same_v<decltype(std::declval<const KVCache&>().get_k_cache()), torch::Tensor> &&
                  std::is_same_v<decltype(std::declval<const KVCache&>().get_v_cache()), torch::Tensor>,
              "KVCache accessors");

AttentionImpl::AttentionImpl(int64_t num_heads, int64_t head_size, float scale, int64_t num_kv_heads,
                             int64_t sliding_window)
    : num_heads_(num_heads), head_size_(head_size), scale_(scale), num_kv_heads_(num_kv_heads),
      window_left_(sliding_window > 0 ? sliding_window : -1) {}  // flash_attention.cpp:257

std::tuple<torch::Tensor, std::optional<torch::Tensor>> AttentionImpl::forward(const AttentionMetadata& md,
                                                                               torch::Tensor& query, torch::Tensor& key,
                                                                               torch::Tensor& value, KVCache& kv_cache) {
  namespace k = xllm::kernel::mi355;
  // the output is allocated here so that a captured graph segment after the attention records its address
  // (layers/dcu/attention.cpp:46-47)
  torch::Tensor output = torch::empty_like(query);
  // an empty / dummy data-parallel step: nothing to write, nothing to attend, and no runner registered under capture
  // (layers/dcu/flash_attention.cpp:299-302)
  if (md.max_seq_len == 0) return {output, std::nullopt};
  auto q = query.unflatten(-1, {num_heads_, head_size_});
  auto kk = key.unflatten(-1, {num_kv_heads_, head_size_});
  auto vv = value.unflatten(-1, {num_kv_heads_, head_size_});
  auto out3 = output.unflatten(-1, {num_heads_, head_size_});
  auto kc = kv_cache.get_k_cache(), vc = kv_cache.get_v_cache();
  if (kc.defined() && kc.dim() >= 2) k::reshape_paged_cache(md.slot_mapping, kk, vv, kc, vc);  // flash_attention.cpp:310-318
  if (md.is_prefill) {
    // prefill attention is the piece a piecewise-captured step leaves OUT of its graphs (flash_attention.cpp:325-365,
    // runtime/dcu_graph_executor_impl.cpp:722-800): while the executor captures, the closure is registered and the placeholder
    // returned; at replay the executor calls it with the step's real metadata. Eager steps run it here and now.
    const double scale = scale_;
    const int64_t window = window_left_;
    AttentionMetadata md_copy = md;
    return ::xllm::kernel::dcu::prefill_with_optional_piecewise_capture(
        [md_copy, q, kk, vv, out3, output, scale, window](const ::xllm::kernel::dcu::AttentionReplayParams& params) mutable
        -> std::tuple<torch::Tensor, std::optional<torch::Tensor>> {
          const AttentionMetadata& m = params.attn_metadata ? *params.attn_metadata : md_copy;
          // causal by construction: attention_metadata_builder.cpp:240-241 sets is_causal with is_prefill
          xllm::kernel::mi355::prefill_attention(q, kk, vv, m.q_cu_seq_lens, m.kv_cu_seq_lens, m.max_query_len, scale,
                                                 m.is_causal, window, out3);
          return {output, std::nullopt};
        },
        output);
  }
  if (md.is_chunked_prefill) {
    k::paged_attention(q, kc, vc, md.q_cu_seq_lens, md.kv_seq_lens, md.block_table, md.max_query_len, md.max_seq_len, scale_,
                       md.is_causal, window_left_, out3);
  } else {                      // decode: one query per sequence, nothing to mask (flash_attention.cpp:220-288)
    k::paged_attention(q, kc, vc, std::nullopt, md.kv_seq_lens, md.block_table, 1, md.max_seq_len, scale_, false,
                       window_left_, out3);
  }
  return {output, std::nullopt};
}

}  // namespace layer
}  // namespace xllm
