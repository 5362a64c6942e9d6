// mi355_process_group.h -- C++ host side of the one-shot xGMI all-reduce (round 6; until round 5 this set-up existed only in
// Python, xllm_amd/parallel.py::OneShotAllReduce).
//
// Reference: xllm::ProcessGroup binds its collectives in C++ over c10d::ProcessGroupNCCL (= RCCL on ROCm):
//   framework/parallel_state/process_group.h:42-125, process_group.cpp:98-110 (allreduce / allreduce_async),
//   cuda_process_group.h:25-53 (ProcessGroupImpl: one TCPStore per group, then the NCCL backend),
//   parallel_state.cpp:183-192 (parallel_state::reduce = in-place SUM all-reduce after a row-parallel linear).
// A USE_MI355 build keeps all of that (rule H of patches/xllm-use-mi355.patch: the DCU branch is taken) and may OPT IN to the
// one-shot kernel of csrc/allreduce.hip for the per-layer sums of a decode step (<= a few MiB): one xGMI hop instead of the
// ring's 2 (W - 1), a plain kernel that is captured INTO the step's HIP graph, fused with the residual add + RMSNorm + int8
// quantisation that follows it. This file is that opt-in, compiled and tested (tests/test_shim.py, two processes):
//
//   OneShotAllReduce::create(store, prefix, rank, world, device, opt, &note)   COLLECTIVE set-up over the group's c10d::Store:
//       IPC buffer (fine-grained -> uncached -> plain), handle exchange, peer mapping, grid_limit choice (one rank per GPU:
//       256 blocks, ranks sharing a GPU: 64), SELF-TEST (one checked message per size class through every kernel form,
//       including a 256-row fused message when grid_limit = 256), and an AGREED verdict: either every rank of the group gets
//       an object or every rank gets nullptr + the reason -- a rank-local failure never makes a rank skip an exchange its
//       peers sit in.
//   ProcessGroupMi355: what process_group.cpp:98-101 becomes -- allreduce(x) takes the one-shot kernel when it accepts the
//       message (a rank-invariant predicate) and the group's own backend otherwise.
#pragma once

#include <torch/torch.h>

#include <functional>
#include <memory>
#include <optional>
#include <string>
#include <torch/csrc/distributed/c10d/Store.hpp>
#include <utility>
#include <vector>

namespace xllm {
namespace mi355 {

class OneShotAllReduce {
 public:
  struct Options {
    size_t max_bytes = size_t(8) << 20;  // largest message the shared slots hold (larger ones: the group's own all-reduce)
    double timeout_s = 2.0;              // bounded wait for a peer's flag: a timeout raises the status word, never hangs the queue
    bool self_test = true;
  };

  // COLLECTIVE over `store` (every rank of the group calls it with the same prefix / world / max_bytes). Returns nullptr on EVERY
  // rank when any rank could not set up or failed the self-test; *note then says why (the same text on every rank).
  static std::shared_ptr<OneShotAllReduce> create(const c10::intrusive_ptr<c10d::Store>& store, const std::string& prefix,
                                                  int rank, int world, const torch::Device& device, const Options& opt,
                                                  std::string* note = nullptr);
  ~OneShotAllReduce();  // unmaps; call close() first while peers may still launch
  // COLLECTIVE: drains this rank's stream, meets the peers at the store, then unmaps (nobody unmaps while a peer may still read)
  void close();

  // rank-invariant: dtype, byte count, the size limit, being on this group's GPU -- never contiguity / alignment (a tensor that is
  // unaligned on SOME rank is staged through an aligned copy instead of sending that rank alone down another path)
  bool takes(const torch::Tensor& x) const;
  // in-place SUM, fp32 accumulation in rank order, one rounding: bit-identical on every rank (parallel_state::reduce)
  void allreduce(torch::Tensor& x);
  // SUM all-reduce of `partial` [M, H] -> residual <- rT(sum + residual) -> RMSNorm (-> per-token int8 quant):
  // {q int8 [M, H], scale f32 [M]} when quantize, {norm [M, H], undefined} otherwise; *sum_out (optional) receives the sum.
  // nullopt = outside the fused kernel's envelope (H % 8, H <= 16384, M * H * 2 <= max_bytes): the caller runs the operators
  std::optional<std::pair<torch::Tensor, torch::Tensor>> allreduce_add_rms_norm(const torch::Tensor& partial,
                                                                                torch::Tensor& residual,
                                                                                const torch::Tensor& weight, double eps,
                                                                                bool quantize, torch::Tensor* sum_out = nullptr);
  // the same fed by the row-parallel W8A8 GEMM's int32 K-slice sums (a [M, K] int8, w_packed = pack_weight_i8 of the [N, K] shard)
  std::optional<std::pair<torch::Tensor, torch::Tensor>> matmul_allreduce_add_rms_norm(
      const torch::Tensor& a_q, const torch::Tensor& a_scale, const torch::Tensor& w_packed, const torch::Tensor& w_scale,
      const std::optional<torch::Tensor>& bias, torch::Tensor& residual, const torch::Tensor& weight, double eps, bool quantize,
      torch::Tensor* sum_out = nullptr);

  // host-side look at the status word (synchronises): false = some launch gave up waiting for a peer, its result was undefined
  bool healthy();
  int grid_limit() const { return grid_limit_; }
  void set_grid_limit(int g) { grid_limit_ = g; }  // tests: the one-row-per-block regime on a shared GPU at row counts that fit
  int memory_kind() const { return kind_; }        // 0 fine-grained, 1 uncached, 2 plain hipMalloc
  int rank() const { return rank_; }
  int world_size() const { return world_; }
  size_t max_bytes() const { return opt_.max_bytes; }
  // every launch inside declines the kernel (callers that issue collectives from SEVERAL streams of one rank: one epoch / flag /
  // slot state per rank)
  struct Suspend {
    explicit Suspend(OneShotAllReduce& o) : o_(o) { ++o_.suspended_; }
    ~Suspend() { --o_.suspended_; }
    OneShotAllReduce& o_;
  };

 private:
  OneShotAllReduce() = default;
  bool self_test();
  void agree(const std::string& round, std::string* mine);  // *mine <- "" iff every rank reported ""
  void unmap();

  c10::intrusive_ptr<c10d::Store> store_;
  std::string prefix_;
  int rank_ = 0, world_ = 1, kind_ = -1, grid_limit_ = 0, suspended_ = 0, round_ = 0;
  torch::Device device_{torch::kCPU};
  Options opt_;
  void* own_ = nullptr;
  std::vector<void*> peers_, opened_;
  torch::Tensor state_, status_, slab_ws_;
};

// ProcessGroup::allreduce with the opt-in (process_group.cpp:98-101): `fallback` = the group's own all-reduce
// (`allreduce_async(input)->wait()` over c10d::ProcessGroupNCCL in an xLLM build)
class ProcessGroupMi355 {
 public:
  ProcessGroupMi355(int rank, int world, std::shared_ptr<OneShotAllReduce> oneshot, std::function<void(torch::Tensor&)> fallback)
      : rank_(rank), world_(world), oneshot_(std::move(oneshot)), fallback_(std::move(fallback)) {}
  int rank() const { return rank_; }
  int world_size() const { return world_; }
  const std::shared_ptr<OneShotAllReduce>& oneshot() const { return oneshot_; }
  // "oneshot-xgmi" | "backend" | "none" (world 1)
  std::string allreduce_kind() const { return world_ <= 1 ? "none" : (oneshot_ ? "oneshot-xgmi" : "backend"); }
  void allreduce(torch::Tensor& input);

 private:
  int rank_, world_;
  std::shared_ptr<OneShotAllReduce> oneshot_;
  std::function<void(torch::Tensor&)> fallback_;
};

// create_tcp_store (process_group.cpp:86-93): the store a group rendezvouses over, rank 0 serves
c10::intrusive_ptr<c10d::Store> create_tcp_store(const std::string& host, int port, int rank, int world);

}  // namespace mi355
}  // namespace xllm
