// STUB of xllm/core/layers/common/attention_metadata.h:73-186 for building shim/mi355_attention.cpp outside the xLLM tree.
// In a USE_MI355 build of xLLM the REAL header is on the include path instead (same include string), and the static_asserts in
// mi355_attention.cpp check, against whichever header is seen, every member the attention class touches. Members are listed
// in the reference's order with the reference's types; the blocks the reference guards with USE_CUDA / USE_MUSA / USE_NPU
// (plan_info ..., *_host tensors) do not exist in a USE_MI355 build and are left out, like in its USE_DCU build.
#pragma once
#include <torch/torch.h>

#include <cstdint>
#include <memory>
#include <optional>
#include <string>
#include <vector>

namespace xllm::layer {

struct DSAMetadata;  // layers/common/dsa_metadata.h (DeepSeek sparse attention; unused on this path)

// the members, as (type, name, default initialiser) -- one list, expanded into the struct below and readable by
// tests/test_host_abi.py, which compares it entry by entry with the reference header whenever that is present
#define XLLM_MI355_ATTENTION_METADATA_FIELDS(F) \
  F(torch::Tensor, q_cu_seq_lens, )                                 \
  F(torch::Tensor, kv_cu_seq_lens, )                                \
  F(torch::Tensor, kv_seq_lens, )                                   \
  F(torch::Tensor, q_seq_lens, )                                    \
  F(std::vector<int32_t>, kv_seq_lens_vec, )                        \
  F(std::vector<int32_t>, q_seq_lens_vec, )                         \
  F(torch::Tensor, block_table, )                                   \
  F(torch::Tensor, slot_mapping, )                                  \
  F(int64_t, max_query_len, )                                       \
  F(int64_t, max_seq_len, )                                         \
  F(int64_t, total_kv_len, = 0)                                     \
  F(std::string, compute_dtype, )                                   \
  F(bool, is_prefill, )                                             \
  F(bool, is_chunked_prefill, )                                     \
  F(bool, is_dummy, )                                               \
  F(bool, is_causal, = true)                                        \
  F(bool, use_expanded_decode_for_spec_verify_attention, = false)   \
  F(torch::Tensor, expanded_kv_seq_lens, )                          \
  F(torch::Tensor, expanded_block_table, )                          \
  F(torch::Tensor, expanded_paged_attention_tiling_data, )          \
  F(torch::Tensor, expanded_kv_seq_lens_host, )                     \
  F(torch::Tensor, mrope_cos, )                                     \
  F(torch::Tensor, mrope_sin, )                                     \
  F(torch::Tensor, paged_kv_indptr, )                               \
  F(torch::Tensor, paged_kv_indices, )                              \
  F(torch::Tensor, paged_kv_last_page_len, )                        \
  F(std::optional<torch::Tensor>, qo_indptr, )                      \
  F(bool, enable_cuda_graph, = false)                               \
  F(torch::Tensor, full_k_cache, )                                  \
  F(torch::Tensor, full_v_cache, )                                  \
  F(torch::Tensor, unshared_k_cache, )                              \
  F(torch::Tensor, unshared_v_cache, )                              \
  F(torch::Tensor, step_tensor, )                                   \
  F(torch::Tensor, chunk_indices, )                                 \
  F(torch::Tensor, batch, )                                         \
  F(torch::Tensor, token_block_offset, )                            \
  F(torch::Tensor, has_initial_states, )                            \
  F(int32_t, tot, = 0)                                              \
  F(torch::Tensor, attn_mask, )                                     \
  F(std::shared_ptr<DSAMetadata>, dsa_metadata, )                  

struct AttentionMetadata {
#define XLLM_MI355_FIELD(type_, name_, init_) type_ name_ init_;
  XLLM_MI355_ATTENTION_METADATA_FIELDS(XLLM_MI355_FIELD)
#undef XLLM_MI355_FIELD
};

}  // namespace xllm::layer
