// mi355_ops_api.h -- xllm::kernel::mi355::*: the per-backend operator header a USE_MI355 build of xLLM includes
// from xllm/core/kernels/ops_api.cpp, next to kernels/cuda/cuda_ops_api.h and kernels/dcu/dcu_ops_api.h.
// Argument lists are IDENTICAL to the CUDA/DCU headers (cited per function) so ops_api.cpp gains one
// `#elif defined(USE_MI355)` line per operator (see INTEGRATION.md / patches/xllm-use-mi355.patch).
// Every function forwards to one C-ABI symbol of include/xllm_mi355.h on c10::hip::getCurrentHIPStream();
// outputs the reference allocates with torch::empty are allocated here the same way; a non-zero C return
// becomes TORCH_CHECK(false, xllm_mi355_strerror(rc)) (reference: CHECK / TORCH_CHECK).
#pragma once
#include <torch/torch.h>

#include <vector>

#include "../include/xllm_mi355.h"

#include <optional>
#include <string>
#include <tuple>

namespace xllm::kernel::mi355 {

// MLA weight absorption (round 5): what DeepseekV2AttentionImpl computes with torch::bmm(x.transpose(0, 1), w).transpose(0, 1)
// (layers/dcu/deepseek_v2_attention.cpp:180-187 project_output, :310-311 q_nope x W_kc), on this backend's kernel and without the
// transposes: x [tokens, heads, K] (any token / head strides), w_nk [heads, N, K] (K contiguous per output column: kv_b_proj's own
// slice for W_vc -- skip the transpose of :336-338 --, W_kc transposed once at load time) -> [tokens, heads, N]
torch::Tensor bmm_heads(const torch::Tensor& x, const torch::Tensor& w_nk, std::optional<torch::Tensor> out = std::nullopt);

// kernels/cuda/cuda_ops_api.h:50-56 (block_copy.cu:120-205): whole-block KV copies for beam-search / prefix forks, called by
// WorkerImpl::execute_cuda_block_copy_kernel (runtime/worker_impl.cpp:1071-1082) under USE_CUDA || USE_DCU (|| USE_MI355)
void block_copy(torch::Tensor key_cache_ptrs, torch::Tensor value_cache_ptrs, torch::Tensor src_block_indices,
                torch::Tensor dst_block_indices, torch::Tensor cum_sum, int64_t numel_per_block,
                torch::ScalarType cache_dtype);

// kernels/cuda/cuda_ops_api.h:31-36
void rotary_embedding(torch::Tensor& positions, torch::Tensor& query, std::optional<torch::Tensor> key,
                      torch::Tensor& cos_sin_cache, bool is_neox);
// cuda_ops_api.h:38-40
void act_and_mul(torch::Tensor out, torch::Tensor input, const std::string& act_mode);
// cuda_ops_api.h:42-47
void reshape_paged_cache(torch::Tensor slot_ids, torch::Tensor keys, torch::Tensor values, torch::Tensor key_cache,
                         torch::Tensor value_cache);
// cuda_ops_api.h:146-155
void rms_norm(torch::Tensor output, torch::Tensor input, torch::Tensor weight, double eps);
void fused_add_rms_norm(torch::Tensor& input, torch::Tensor& residual, torch::Tensor& weight, double epsilon);
// cuda_ops_api.h:157-159 == kernels/dcu/dcu_ops_api.h:30-32
torch::Tensor matmul(torch::Tensor a, torch::Tensor b, std::optional<torch::Tensor> bias);
// cuda_ops_api.h:184-186, 190-193, 201-216, 220-227
void static_scaled_fp8_quant(torch::Tensor& out, torch::Tensor const& input, torch::Tensor const& scale);
std::tuple<torch::Tensor, torch::Tensor> fp8_scaled_quantize(const torch::Tensor& input,
                                                             const std::optional<torch::Tensor>& output = std::nullopt,
                                                             const std::optional<torch::Tensor>& scale = std::nullopt);
void rms_norm_static_fp8_quant(torch::Tensor& out, torch::Tensor& input, torch::Tensor& weight, torch::Tensor& scale,
                               double epsilon);
void fused_add_rms_norm_static_fp8_quant(torch::Tensor& out, torch::Tensor& input, torch::Tensor& residual,
                                         torch::Tensor& weight, torch::Tensor& scale, double epsilon);
torch::Tensor fp8_scaled_matmul(const torch::Tensor& a, const torch::Tensor& b, const torch::Tensor& a_scale,
                                const torch::Tensor& b_scale, torch::ScalarType output_dtype,
                                const std::optional<torch::Tensor>& bias = std::nullopt,
                                const std::optional<torch::Tensor>& output = std::nullopt);
// cuda_ops_api.h:235-249
void fused_qk_norm_rope(torch::Tensor& qkv, int64_t num_heads_q, int64_t num_heads_k, int64_t num_heads_v,
                        int64_t head_dim, double eps, const torch::Tensor& q_weight, const torch::Tensor& k_weight,
                        const torch::Tensor& cos_sin_cache, bool interleaved, const torch::Tensor& position_ids);
// kernels/dcu/dcu_ops_api.h:96-109 (only the no-smooth per-token int8 mode exists on DCU; same here)
std::tuple<torch::Tensor, torch::Tensor> scaled_quantize(
    const torch::Tensor& x, const torch::Tensor& smooth, const std::optional<torch::Tensor>& zero,
    const std::optional<torch::Tensor>& token_count, const std::optional<torch::Tensor>& gather_index,
    const std::optional<torch::Tensor>& gather_index_start_position, const std::optional<torch::Tensor>& output,
    const std::optional<torch::Tensor>& output_scale, const std::string& act_mode, double active_coef, bool is_gated,
    torch::ScalarType quant_type);
// dcu_ops_api.h:113-129
torch::Tensor scaled_matmul(const torch::Tensor& a, const torch::Tensor& b, const std::optional<torch::Tensor>& a_scale,
                            const torch::Tensor& b_scale, torch::ScalarType output_dtype,
                            const std::optional<torch::Tensor>& bias, const std::optional<torch::Tensor>& c,
                            const std::string& act_mode, int64_t quant_bit_size, double alpha, double beta,
                            bool use_hp_active, int64_t a_quant_bit_size, const std::optional<torch::Tensor>& a_calib,
                            const std::optional<torch::Tensor>& b_calib, const std::optional<torch::Tensor>& output);
// dcu_ops_api.h:48-51, 71-73
// decode-shaped scaled_matmul calls run on a packed copy of the weight kept by the shim (one per weight tensor); these two
// expose the cache to tests and to code that frees or replaces weights
int64_t packed_weight_cache_size();
void clear_packed_weight_cache();
// weight-load-time packing of a W8A8 weight [N, K] int8 (N % 16 == 0, K % 128 == 0) into MFMA-fragment order: what the
// USE_MI355 branch of the linear layer's loader calls once per weight; scaled_matmul then finds the copy.
// Returns the packed tensor (the cache holds a reference too, for as long as `b` lives).
torch::Tensor pack_w8a8_weight(const torch::Tensor& b);
// forget the packed copy of `b` (after writing into the weight through a raw pointer, which bumps no version counter)
void invalidate_packed_weight(const torch::Tensor& b);
torch::Tensor group_gemm(const torch::Tensor& input, const torch::Tensor& weight, const torch::Tensor& token_count,
                         std::optional<torch::Tensor> output = std::nullopt);
// kernel::moe_active_topk / cuda::moe_fused_topk (ops_api.h:70; kernels/cuda/moe/moe_fused_topk.cu:31-61):
// (topk_weights float32 [T, topk], topk_ids int32 [T, topk]); scoring_func "softmax" | "sigmoid" (anything else throws,
// like the reference's LOG(FATAL)); the correction bias only takes part in the sigmoid selection.
std::tuple<torch::Tensor, torch::Tensor> moe_fused_topk(const torch::Tensor& gating_output, int64_t topk, bool renormalize,
                                                        const std::optional<torch::Tensor>& correction_bias,
                                                        const std::string& scoring_func);
// dcu::moe_grouped_topk / dcu::moe_active_topk (kernels/dcu/dcu_ops_api.h, kernels/dcu/topk_gate.cpp:59-146): the
// DeepSeek grouped gate (aiter grouped_topk / biased_grouped_topk in the reference) and the dispatcher the layer calls:
// num_expert_group > 1 -> grouped, else moe_fused_topk (which ignores routed_scaling_factor, as in the reference)
std::tuple<torch::Tensor, torch::Tensor> moe_grouped_topk(const torch::Tensor& gating_output, int64_t topk,
                                                          int64_t num_expert_group, int64_t topk_group, bool renormalize,
                                                          const std::optional<torch::Tensor>& correction_bias,
                                                          const std::string& scoring_func, double routed_scaling_factor);
std::tuple<torch::Tensor, torch::Tensor> moe_active_topk(const torch::Tensor& gating_output, int64_t topk,
                                                         int64_t num_expert_group, int64_t topk_group, bool renormalize,
                                                         const std::optional<torch::Tensor>& correction_bias,
                                                         const std::string& scoring_func, double routed_scaling_factor);
// kernel::moe_gen_idx (ops_api.h:73) -> {src_dst, dst_src, expert_sizes} (int32); stable inside an expert
std::vector<torch::Tensor> moe_gen_idx(const torch::Tensor& expert_id, int64_t expert_num);
// kernel::moe_combine_result (ops_api.h:77; MoeCombineResultParams param.h:575-...): input [T*topk, H] in TOKEN order
// (after the caller's index_copy_), reduce_weight [T, topk] float32
torch::Tensor moe_combine_result(const torch::Tensor& input, const torch::Tensor& reduce_weight);
// the spellings of kernels/cuda/cuda_ops_api.h that the USE_DCU branches of ops_api.cpp (:629-633, :679-689) call
std::tuple<torch::Tensor, torch::Tensor, torch::Tensor> moe_compute_index(const torch::Tensor& expert_id, int64_t expert_num);
torch::Tensor moe_combine_result(const torch::Tensor& input, const torch::Tensor& reduce_weight, int64_t num_tokens,
                                 int32_t topk);
// the same with MoeCombineResultParams::gather_ids honoured: input stays in EXPERT order (the second grouped GEMM's
// output) and row gather_ids[t*topk+k] is read for (t, k) -- index_copy_ + moe_combine_result in one pass
// local_expert_sizes (int32 [E_local], EP rank whose experts were sorted to the front): rows at or past their sum are the
// zero rows of the reference's gemm2_full (fused_moe.cpp:291-297) and are skipped -- no host read of the sizes
torch::Tensor moe_combine_result_sorted(const torch::Tensor& input_sorted, const torch::Tensor& reduce_weight,
                                        const torch::Tensor& gather_ids,
                                        const std::optional<torch::Tensor>& local_expert_sizes = std::nullopt);
// index_select(hidden, dst_src / topk) + group_gemm without the expanded copy (fused_moe.cpp:195-197, 250-262);
// returns an undefined tensor when the 256x256 kernel cannot take the shape (the caller keeps the two reference calls)
torch::Tensor group_gemm_gather(const torch::Tensor& input, const torch::Tensor& row_index, int64_t index_div,
                                const torch::Tensor& weight, const torch::Tensor& token_count);
// kernel::group_gemm with GroupGemmParams::a_scale / b_scale (kernels/param.h:374-394): W8A8 experts. input int8
// [rows, K] sorted by expert with a_scale [rows]; or, with gather_index defined, the un-expanded [T, K] / [T] gathered
// as gather_index[r] / index_div. weight int8 [E, N, K], b_scale float32 [E, N]. Returns [rows, N] in out_dtype.
torch::Tensor group_gemm_w8a8(const torch::Tensor& input, const torch::Tensor& a_scale, const torch::Tensor& weight,
                              const torch::Tensor& b_scale, const torch::Tensor& token_count, torch::ScalarType out_dtype,
                              const std::optional<torch::Tensor>& gather_index, int64_t index_div);
// flash_mla::dense_decode (kernels/dcu/flash_mla_adapter.h:40-50): q [B, H, 576] (nope || pe), k_cache
// [n_blocks, block, 1, 576], values = the first head_size_v dims of the latent
torch::Tensor mla_decode(const torch::Tensor& q, const torch::Tensor& k_cache, const torch::Tensor& seqlens_k,
                         const torch::Tensor& block_table, int64_t head_size_v, double softmax_scale, int64_t max_kv_len);

// ---- flash_mla adapter: kernels/dcu/flash_mla_adapter.h:33-53, the interface layers/dcu/deepseek_v2_attention.cpp:189-210 binds.
// Under USE_MI355 that layer file includes this header instead of kernels/dcu/flash_mla_adapter.h and reaches it through
// `namespace dcu = mi355` (patches/xllm-use-mi355.patch), so `kernel::dcu::flash_mla::dense_decode(params)` compiles unchanged.
namespace flash_mla {
enum class DenseDecodeKind {
  kQNopePe,
};
// q_nope [B, S_q, H_q, kv_lora_rank], q_pe [B, S_q, H_q, qk_rope_head_dim], k_cache [num_blocks, page, 1, kv_lora_rank + rope],
// seqlens_k [B] int32, block_table [B, max_blocks] int32 (flash_mla_adapter.h:33-50, member for member)
struct DenseDecodeParams {
  torch::Tensor q_nope;
  torch::Tensor q_pe;
  torch::Tensor k_cache;
  torch::Tensor seqlens_k;
  torch::Tensor block_table;
  int64_t head_size_v = 0;
  float softmax_scale = -1.0F;
  bool is_causal = false;
  DenseDecodeKind kind = DenseDecodeKind::kQNopePe;
};
// Returns [B, S_q, H_q, head_size_v]. S_q == 1: xllm_mi355_mla_decode; S_q > 1 (multi-token decode): xllm_mi355_mla_prefill over
// the cache with q_cu = S_q * arange(B + 1) -- query i of a sequence sees seqlens_k - S_q + i + 1 keys when is_causal, all of
// them otherwise (the flash kernels' bottom-right alignment). Pages of any size (the closed flash_mla.so requires 64,
// flash_mla_adapter.cpp:79-82; 64-multiples take the LDS-DMA kernels here). No host sync: graph-capturable; the launch plan is
// sized from block_table.size(1) * page, an upper bound of every sequence length.
torch::Tensor dense_decode(DenseDecodeParams& params);
// DeepseekV2AttentionImpl::prefill_sdpa (deepseek_v2_attention.cpp:212-262) without the host loop over sequences: the latent rows
// were written to the paged cache by store_latent_cache just before (:300-303), so prefill -- and chunked prefill, which the
// reference CHECK-fails on (:270-271) -- read them back through the block table. q_nope [T, H, kv_lora], q_pe [T, H, rope],
// q_cu_seq_lens [B + 1], kv_seq_lens [B], block_table [B, max_blocks]; returns [T, H, head_size_v].
torch::Tensor prefill_paged(const torch::Tensor& q_nope, const torch::Tensor& q_pe, const torch::Tensor& k_cache,
                            const torch::Tensor& q_cu_seq_lens, const torch::Tensor& kv_seq_lens,
                            const torch::Tensor& block_table, int64_t head_size_v, double softmax_scale, bool is_causal);
// DeepseekV2AttentionImpl::store_latent_cache (:170-178) as one kernel (index_copy_ with int64 slots in the reference): rows with
// slot < 0 (graph padding) are skipped
void store_latent_cache(const torch::Tensor& latent_cache, const torch::Tensor& slot_mapping, const torch::Tensor& k_cache);
}  // namespace flash_mla

torch::Tensor build_block_table_from_paged_kv(const torch::Tensor& paged_kv_indptr,
                                              const torch::Tensor& paged_kv_indices);

// sampler (N3): dcu_ops_api.h:71-81
torch::Tensor random_sample(const torch::Tensor& probs);
torch::Tensor rejection_sample(const torch::Tensor& draft_token_ids, const torch::Tensor& num_draft_tokens,
                               const torch::Tensor& cu_num_draft_tokens, const std::optional<torch::Tensor>& draft_probs,
                               const torch::Tensor& target_probs, const torch::Tensor& bonus_token_ids,
                               const torch::Tensor& uniform_rand, const torch::Tensor& uniform_probs,
                               int64_t max_spec_len);

// graph-mode decode: kernels/cuda/llm_decode_metadata_update.h:34-57 (same field names; the struct IS the C-ABI one,
// whose first 19 members are the reference's LlmDecodeMetadataUpdateParams in order, followed by the optional
// dense-block-table extension)
using LlmDecodeMetadataUpdateParams = xllm_mi355_decode_metadata_t;
void update_llm_decode_metadata(const LlmDecodeMetadataUpdateParams& params, void* stream);

// attention entry points bound by layers/mi355/attention.cpp (arg sets of prefix_prefill_varlen_fwd /
// prefix_decode_varlen_fwd, layers/dcu/flash_attention.cpp:45-94)
torch::Tensor prefill_attention(const torch::Tensor& q, const torch::Tensor& k, const torch::Tensor& v,
                                const torch::Tensor& cu_seqlens_q, const torch::Tensor& cu_seqlens_k, int64_t max_q_len,
                                double scale, bool is_causal, int64_t window_left,
                                std::optional<torch::Tensor> out = std::nullopt);
torch::Tensor paged_attention(const torch::Tensor& q, const torch::Tensor& k_cache, const torch::Tensor& v_cache,
                              const std::optional<torch::Tensor>& cu_seqlens_q, const torch::Tensor& kv_seq_lens,
                              const torch::Tensor& block_table, int64_t max_q_len, int64_t max_kv_len, double scale,
                              bool is_causal, int64_t window_left, std::optional<torch::Tensor> out = std::nullopt);

}  // namespace xllm::kernel::mi355
