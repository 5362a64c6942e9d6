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

struct AttentionMetadata {
  torch::Tensor q_cu_seq_lens;
  torch::Tensor kv_cu_seq_lens;
  torch::Tensor kv_seq_lens;
  torch::Tensor q_seq_lens;
  std::vector<int32_t> kv_seq_lens_vec;
  std::vector<int32_t> q_seq_lens_vec;
  torch::Tensor block_table;
  torch::Tensor slot_mapping;
  int64_t max_query_len;
  int64_t max_seq_len;
  int64_t total_kv_len = 0;
  std::string compute_dtype;
  bool is_prefill;
  bool is_chunked_prefill;
  bool is_dummy;
  bool is_causal = true;
  bool use_expanded_decode_for_spec_verify_attention = false;
  torch::Tensor expanded_kv_seq_lens;
  torch::Tensor expanded_block_table;
  torch::Tensor expanded_paged_attention_tiling_data;
  torch::Tensor expanded_kv_seq_lens_host;
  torch::Tensor mrope_cos;
  torch::Tensor mrope_sin;
  torch::Tensor paged_kv_indptr;
  torch::Tensor paged_kv_indices;
  torch::Tensor paged_kv_last_page_len;
  std::optional<torch::Tensor> qo_indptr;
  bool enable_cuda_graph = false;
  torch::Tensor full_k_cache;
  torch::Tensor full_v_cache;
  torch::Tensor unshared_k_cache;
  torch::Tensor unshared_v_cache;
  torch::Tensor step_tensor;
  torch::Tensor chunk_indices;
  torch::Tensor batch;
  torch::Tensor token_block_offset;
  torch::Tensor has_initial_states;
  int32_t tot = 0;
  torch::Tensor attn_mask;
  std::shared_ptr<DSAMetadata> dsa_metadata;
};

}  // namespace xllm::layer
