// Out-of-tree implementation of shim/stub/kernels/dcu/attention_runner.h: the control flow of the reference's
// attention_runner.cpp:28-70 (capturing -> register the runner and return the output placeholder without running attention;
// otherwise run it now) over a thread-local runner list instead of GlobalCaptureInstance / PiecewiseGraphs.
#include "kernels/dcu/attention_runner.h"

#include <c10/core/InferenceMode.h>

#include <vector>

namespace xllm::kernel::dcu {
namespace {
thread_local bool g_capturing = false;
thread_local std::vector<AttentionRunner> g_runners;
}  // namespace

void AttentionRunner::run_capture(RunFn run_fn) {
  run_fn_ = std::move(run_fn);
  TORCH_CHECK(run_fn_ != nullptr);
}

void AttentionRunner::run_replay(const AttentionReplayParams& params) {
  TORCH_CHECK(run_fn_ != nullptr);
  c10::InferenceMode guard(true);
  run_fn_(params);
}

std::tuple<torch::Tensor, std::optional<torch::Tensor>> prefill_with_optional_piecewise_capture(
    AttentionRunner::RunFn run_fn, const torch::Tensor& output) {
  if (g_capturing) {
    AttentionRunner runner;
    runner.run_capture(std::move(run_fn));
    g_runners.push_back(std::move(runner));
    return {output, std::nullopt};
  }
  AttentionReplayParams params;
  params.actual_num_tokens = static_cast<uint32_t>(output.size(0));
  return run_fn(params);
}

void stub_begin_piecewise_capture() {
  g_runners.clear();
  g_capturing = true;
}
int64_t stub_end_piecewise_capture() {
  g_capturing = false;
  return static_cast<int64_t>(g_runners.size());
}
void stub_replay_runners(const AttentionReplayParams& params) {
  for (auto& r : g_runners) r.run_replay(params);
}

}  // namespace xllm::kernel::dcu
