// Stand-in for xllm/core/kernels/dcu/attention_runner.h (the reference's HOST code for piecewise HIP-graph capture of prefill
// steps), used only to build shim/mi355_attention.cpp OUTSIDE the xLLM tree. Inside xLLM the real header is found first
// (kernels/CMakeLists.txt of a USE_MI355 build compiles dcu/{attention_runner,piecewise_graphs,global_capture_instance}.cpp).
// Interface restated from attention_runner.h:25-63: the replay parameters, the runner's RunFn and the entry point the
// attention layer calls (layers/dcu/flash_attention.cpp:334-365). The out-of-tree implementation (attention_runner_stub.cpp)
// keeps one thread-local list of captured runners that the test binding can replay.
#pragma once
#include <torch/torch.h>

#include <cstdint>
#include <functional>
#include <memory>
#include <optional>
#include <tuple>

namespace xllm::layer {
struct AttentionMetadata;
}

namespace xllm::kernel::dcu {

struct AttentionReplayParams {
  uint32_t actual_num_tokens = 0;
  std::shared_ptr<layer::AttentionMetadata> attn_metadata;
};

class AttentionRunner final {
 public:
  using RunFn = std::function<std::tuple<torch::Tensor, std::optional<torch::Tensor>>(const AttentionReplayParams&)>;
  AttentionRunner() = default;
  AttentionRunner(const AttentionRunner&) = delete;
  AttentionRunner& operator=(const AttentionRunner&) = delete;
  AttentionRunner(AttentionRunner&&) noexcept = default;
  AttentionRunner& operator=(AttentionRunner&&) noexcept = default;
  void run_capture(RunFn run_fn);
  void run_replay(const AttentionReplayParams& params);

 private:
  RunFn run_fn_;
};

std::tuple<torch::Tensor, std::optional<torch::Tensor>> prefill_with_optional_piecewise_capture(
    AttentionRunner::RunFn run_fn, const torch::Tensor& output);

// ---- out-of-tree test hooks (NOT part of the reference interface): what GlobalCaptureInstance does for the graph executor
void stub_begin_piecewise_capture();
int64_t stub_end_piecewise_capture();                            // number of attention runners registered
void stub_replay_runners(const AttentionReplayParams& params);   // PiecewiseGraphs::replay's attention part

}  // namespace xllm::kernel::dcu
