// test-only binding so the pytest suite can call the C++ shim (xllm::kernel::mi355::*) exactly as ops_api.cpp would
#include <torch/extension.h>

#include "kernels/dcu/attention_runner.h"
#include "mi355_attention.h"
#include "mi355_ops_api.h"
#include "mi355_process_group.h"

namespace k = xllm::kernel::mi355;

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("rms_norm", &k::rms_norm);
  m.def("fused_add_rms_norm", [](torch::Tensor x, torch::Tensor r, torch::Tensor w, double eps) { k::fused_add_rms_norm(x, r, w, eps); });
  m.def("act_and_mul", &k::act_and_mul);
  m.def("reshape_paged_cache", &k::reshape_paged_cache);
  m.def("block_copy", &k::block_copy);
  m.def("bmm_heads", [](const torch::Tensor& x, const torch::Tensor& w) { return k::bmm_heads(x, w); });
  m.def("rotary_embedding", [](torch::Tensor pos, torch::Tensor q, std::optional<torch::Tensor> kk, torch::Tensor cache, bool neox) { k::rotary_embedding(pos, q, kk, cache, neox); });
  m.def("matmul", &k::matmul);
  m.def("random_sample", &k::random_sample);
  m.def("moe_fused_topk", &k::moe_fused_topk);
  m.def("moe_grouped_topk", &k::moe_grouped_topk);
  m.def("moe_active_topk", &k::moe_active_topk);
  m.def("moe_gen_idx", &k::moe_gen_idx);
  m.def("moe_combine_result", [](const torch::Tensor& x, const torch::Tensor& w) { return k::moe_combine_result(x, w); });
  m.def("moe_combine_result4", [](const torch::Tensor& x, const torch::Tensor& w, int64_t n, int64_t topk) { return k::moe_combine_result(x, w, n, (int32_t)topk); });
  m.def("moe_compute_index", &k::moe_compute_index);
  m.def("moe_combine_result_sorted", [](const torch::Tensor& x, const torch::Tensor& w, const torch::Tensor& g, std::optional<torch::Tensor> ls) { return k::moe_combine_result_sorted(x, w, g, ls); }, pybind11::arg("input_sorted"), pybind11::arg("reduce_weight"), pybind11::arg("gather_ids"), pybind11::arg("local_expert_sizes") = std::nullopt);
  m.def("group_gemm", [](const torch::Tensor& x, const torch::Tensor& w, const torch::Tensor& c) { return k::group_gemm(x, w, c, std::nullopt); });
  m.def("group_gemm_gather", &k::group_gemm_gather);
  m.def("group_gemm_w8a8", [](const torch::Tensor& x, const torch::Tensor& as, const torch::Tensor& w, const torch::Tensor& bs, const torch::Tensor& c, std::optional<torch::Tensor> idx, int64_t div) { return k::group_gemm_w8a8(x, as, w, bs, c, torch::kBFloat16, idx, div); });
  m.def("mla_decode", &k::mla_decode);
  // kernel::dcu::flash_mla::dense_decode(DenseDecodeParams&) exactly as DeepseekV2AttentionImpl::decode_flash_mla fills the struct
  // (layers/dcu/deepseek_v2_attention.cpp:189-210): q views [B, 1, H, .], k_cache, kv_seq_lens, block_table, kv_lora, scale, causal
  m.def("flash_mla_dense_decode", [](const torch::Tensor& q_nope, const torch::Tensor& q_pe, const torch::Tensor& k_cache,
                                     const torch::Tensor& seqlens_k, const torch::Tensor& block_table, int64_t head_size_v,
                                     double scale, bool is_causal) {
    k::flash_mla::DenseDecodeParams params;
    params.q_nope = q_nope;
    params.q_pe = q_pe;
    params.k_cache = k_cache;
    params.seqlens_k = seqlens_k;
    params.block_table = block_table;
    params.head_size_v = head_size_v;
    params.softmax_scale = (float)scale;
    params.is_causal = is_causal;
    params.kind = k::flash_mla::DenseDecodeKind::kQNopePe;
    return k::flash_mla::dense_decode(params);
  });
  m.def("flash_mla_prefill_paged", &k::flash_mla::prefill_paged);
  m.def("flash_mla_store_latent_cache", &k::flash_mla::store_latent_cache);
  m.def("rejection_sample", &k::rejection_sample);
  m.def("scaled_quantize", [](const torch::Tensor& x) {
    return k::scaled_quantize(x, torch::Tensor(), std::nullopt, std::nullopt, std::nullopt, std::nullopt, std::nullopt, std::nullopt, "none", 1.0, false, torch::kInt8);
  });
  m.def("scaled_matmul", [](const torch::Tensor& a, const torch::Tensor& b, const torch::Tensor& as, const torch::Tensor& bs, std::optional<torch::Tensor> bias) {
    return k::scaled_matmul(a, b, as, bs, torch::kBFloat16, bias, std::nullopt, "none", 8, 1.0, 0.0, false, 8, std::nullopt, std::nullopt, std::nullopt);
  });
  m.def("scaled_matmul_c", [](const torch::Tensor& a, const torch::Tensor& b, const torch::Tensor& as, const torch::Tensor& bs,
                              std::optional<torch::Tensor> bias, const torch::Tensor& c, double alpha, double beta,
                              std::optional<torch::Tensor> output) {
    return k::scaled_matmul(a, b, as, bs, c.scalar_type(), bias, c, "none", 8, alpha, beta, false, 8, std::nullopt, std::nullopt, output);
  });
  m.def("packed_weight_cache_size", &k::packed_weight_cache_size);
  m.def("pack_w8a8_weight", &k::pack_w8a8_weight);
  m.def("invalidate_packed_weight", &k::invalidate_packed_weight);
  m.def("fused_layernorm", [](torch::Tensor x, torch::Tensor w, double eps, std::optional<torch::Tensor> residual) {
    // the reference's decoder layer calls kernel::fused_layernorm with a residual (qwen2_decoder_layer.cpp:66-85)
    if (residual.has_value()) { k::fused_add_rms_norm(x, *residual, w, eps); return x; }
    torch::Tensor out = torch::empty_like(x);
    k::rms_norm(out, x, w, eps);
    return out;
  });
  m.def("clear_packed_weight_cache", &k::clear_packed_weight_cache);
  m.def("fp8_scaled_quantize", [](const torch::Tensor& x) { return k::fp8_scaled_quantize(x); });
  m.def("paged_attention", [](const torch::Tensor& q, const torch::Tensor& kc, const torch::Tensor& vc, const torch::Tensor& kv_lens, const torch::Tensor& bt, int64_t max_kv, double scale) {
    return k::paged_attention(q, kc, vc, std::nullopt, kv_lens, bt, 1, max_kv, scale, false, -1);
  });
  m.def("attention_forward", [](torch::Tensor q, torch::Tensor kk, torch::Tensor v, torch::Tensor kc, torch::Tensor vc, torch::Tensor slots, torch::Tensor kv_lens, torch::Tensor bt, int64_t nq, int64_t nkv, int64_t d, int64_t max_kv) {
    xllm::layer::Attention attn(nq, d, 1.0f / std::sqrt((float)d), nkv, -1);   // TORCH_MODULE holder, as the layers use it
    xllm::layer::AttentionMetadata md{};
    md.kv_seq_lens = kv_lens; md.block_table = bt; md.slot_mapping = slots; md.max_seq_len = max_kv;
    md.max_query_len = 1; md.is_prefill = false; md.is_chunked_prefill = false; md.is_dummy = false; md.is_causal = false;
    xllm::KVCache cache(kc, vc);
    return std::get<0>(attn->forward(md, q, kk, v, cache));
  });
  // AttentionImpl::forward on a prefill step, eagerly or the way a piecewise-captured step of the DCU graph executor drives it
  // (runtime/dcu_graph_executor_impl.cpp:722-800): capture = true registers the attention closure and returns the untouched
  // placeholder; piecewise_replay() then runs the registered closures with the metadata of the step being replayed
  m.def("attention_prefill_forward", [](torch::Tensor q, torch::Tensor kk, torch::Tensor v, torch::Tensor kc, torch::Tensor vc,
                                        torch::Tensor slots, torch::Tensor q_cu, torch::Tensor kv_cu, int64_t max_q, int64_t nq,
                                        int64_t nkv, int64_t d, bool capture) {
    xllm::layer::Attention attn(nq, d, 1.0f / std::sqrt((float)d), nkv, -1);
    xllm::layer::AttentionMetadata md{};
    md.q_cu_seq_lens = q_cu; md.kv_cu_seq_lens = kv_cu; md.slot_mapping = slots; md.max_query_len = max_q; md.max_seq_len = max_q;
    md.is_prefill = true; md.is_chunked_prefill = false; md.is_dummy = false; md.is_causal = true;
    xllm::KVCache cache(kc, vc);
    if (capture) xllm::kernel::dcu::stub_begin_piecewise_capture();
    torch::Tensor out = std::get<0>(attn->forward(md, q, kk, v, cache));
    const int64_t n = capture ? xllm::kernel::dcu::stub_end_piecewise_capture() : 0;
    return std::make_tuple(out, n);
  });
  m.def("piecewise_replay", [](std::optional<torch::Tensor> q_cu, std::optional<torch::Tensor> kv_cu, int64_t max_q, int64_t n_tokens) {
    xllm::kernel::dcu::AttentionReplayParams params;
    params.actual_num_tokens = (uint32_t)n_tokens;
    if (q_cu.has_value()) {
      auto md = std::make_shared<xllm::layer::AttentionMetadata>();
      md->q_cu_seq_lens = *q_cu; md->kv_cu_seq_lens = *kv_cu; md->max_query_len = max_q; md->max_seq_len = max_q;
      md->is_prefill = true; md->is_causal = true;
      params.attn_metadata = md;
    }
    xllm::kernel::dcu::stub_replay_runners(params);
  });
  m.def("abi_version", [] { return (int64_t)xllm_mi355_abi_version(); });

  // ---- the C++ host side of the one-shot all-reduce (shim/mi355_process_group.{h,cpp}), as ProcessGroupImpl would drive it
  // (cuda_process_group.h:25-53: one TCPStore per group, rank 0 serves): tests/test_shim.py runs two processes through this
  namespace pg = xllm::mi355;
  struct OneShotGroup {
    c10::intrusive_ptr<c10d::Store> store;
    std::shared_ptr<pg::OneShotAllReduce> ar;
    std::unique_ptr<pg::ProcessGroupMi355> group;
    std::string note;
    int rank, world;
  };
  pybind11::class_<OneShotGroup>(m, "OneShotGroup")
      .def(pybind11::init([](const std::string& host, int port, int rank, int world, int device_index, int64_t max_bytes,
                             bool self_test, const std::string& prefix, pybind11::object fallback) {
             auto g = std::make_unique<OneShotGroup>();
             g->rank = rank;
             g->world = world;
             g->store = pg::create_tcp_store(host, port, rank, world);
             pg::OneShotAllReduce::Options opt;
             opt.max_bytes = (size_t)max_bytes;
             opt.self_test = self_test;
             {
               pybind11::gil_scoped_release nogil;   // the set-up blocks on the store until every rank has arrived
               g->ar = pg::OneShotAllReduce::create(g->store, prefix, rank, world, torch::Device(torch::kCUDA, device_index), opt, &g->note);
             }
             std::function<void(torch::Tensor&)> fb;
             if (!fallback.is_none()) fb = [fallback](torch::Tensor& t) { pybind11::gil_scoped_acquire gil; fallback(t); };
             g->group = std::make_unique<pg::ProcessGroupMi355>(rank, world, g->ar, fb);
             return g;
           }),
           pybind11::arg("host"), pybind11::arg("port"), pybind11::arg("rank"), pybind11::arg("world"), pybind11::arg("device_index"),
           pybind11::arg("max_bytes") = (int64_t)(8 << 20), pybind11::arg("self_test") = true, pybind11::arg("prefix") = "tp0",
           pybind11::arg("fallback") = pybind11::none())
      .def("active", [](OneShotGroup& g) { return g.ar != nullptr; })
      .def("note", [](OneShotGroup& g) { return g.note; })
      .def("allreduce_kind", [](OneShotGroup& g) { return g.group->allreduce_kind(); })
      .def("grid_limit", [](OneShotGroup& g) { return g.ar ? g.ar->grid_limit() : -1; })
      .def("set_grid_limit", [](OneShotGroup& g, int v) { if (g.ar) g.ar->set_grid_limit(v); })
      .def("memory_kind", [](OneShotGroup& g) { return g.ar ? g.ar->memory_kind() : -1; })
      .def("takes", [](OneShotGroup& g, const torch::Tensor& x) { return g.ar && g.ar->takes(x); })
      .def("allreduce", [](OneShotGroup& g, torch::Tensor x) { g.group->allreduce(x); return x; })
      .def("allreduce_add_rms_norm", [](OneShotGroup& g, const torch::Tensor& partial, torch::Tensor residual, const torch::Tensor& w,
                                        double eps, bool quantize) -> pybind11::object {
        TORCH_CHECK(g.ar, "one-shot path is off: ", g.note);
        torch::Tensor sum;
        auto got = g.ar->allreduce_add_rms_norm(partial, residual, w, eps, quantize, &sum);
        if (!got.has_value()) return pybind11::none();
        return pybind11::cast(std::make_tuple(got->first, got->second, sum));
      })
      .def("matmul_allreduce_add_rms_norm", [](OneShotGroup& g, const torch::Tensor& a, const torch::Tensor& as, const torch::Tensor& wp,
                                               const torch::Tensor& ws, std::optional<torch::Tensor> bias, torch::Tensor residual,
                                               const torch::Tensor& w, double eps, bool quantize) -> pybind11::object {
        TORCH_CHECK(g.ar, "one-shot path is off: ", g.note);
        torch::Tensor sum;
        auto got = g.ar->matmul_allreduce_add_rms_norm(a, as, wp, ws, bias, residual, w, eps, quantize, &sum);
        if (!got.has_value()) return pybind11::none();
        return pybind11::cast(std::make_tuple(got->first, got->second, sum));
      })
      .def("healthy", [](OneShotGroup& g) { return g.ar && g.ar->healthy(); })
      .def("close", [](OneShotGroup& g) {
        if (g.ar) { pybind11::gil_scoped_release nogil; g.ar->close(); }
        g.ar.reset();
      });
}
