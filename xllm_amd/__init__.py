"""xllm_amd -- MI355X (gfx950) kernel backend for xLLM's decode/prefill hot path.

Layout:
  csrc/            hand-written HIP kernels + the C ABI declared in include/xllm_mi355.h
  lib/             libxllm_mi355.so (built by csrc/Makefile / __graft_entry__.build()); libxllm_mi355_tuning.so = the -DXM_TUNING
                   flavour for tools/ only (make -C csrc tuning)
  _lib.py          ctypes binding of the C ABI (fails loudly when the library is missing: no CPU fallback)
  ops.py           host-side mirror of xllm::kernel::* (xllm/core/kernels/ops_api.h) over torch tensors, incl. the sampler's logits
                   processors (framework/sampling/logits_utils.cpp)
  attention.py     AttentionMetadata + AttentionImpl mirror (xllm/core/layers/dcu/attention.h:31-51), host batch builder
  parallel.py      ProcessGroup mirror: tensor-parallel SUM all-reduce as a one-shot kernel over peer-mapped xGMI buffers IN the
                   compute stream (fused with residual add + RMSNorm + quant; self-tested at set-up, RCCL otherwise), RCCL all-gather /
                   all-to-all, launch_reduce / finish_reduce (RCCL on its own stream), piecewise HIP graphs, sharded greedy argmax
  layers.py        Qwen2 decoder layer call order (xllm/core/layers/qwen2_decoder_layer.cpp:87-110), FusedMoE, DeepseekV2Attention,
                   the dual micro-batch executor
  engine.py        step-level decode harness (LLMEngine::step shape: host builder -> H2D -> metadata refresh -> graph replay -> sampler)
  forward_input.py ForwardInput / contiguous input buffer mirror (runtime/forward_params.h)
"""
__all__ = ["ops", "attention", "parallel", "layers", "engine", "forward_input"]
