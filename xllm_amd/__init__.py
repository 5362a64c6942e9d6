"""xllm_amd -- MI355X (gfx950) kernel backend for xLLM's decode/prefill hot path.

Layout:
  csrc/        hand-written HIP kernels + the C ABI declared in include/xllm_mi355.h
  lib/         libxllm_mi355.so (built by csrc/Makefile / __graft_entry__.build())
  _lib.py      ctypes binding of the C ABI (fails loudly when the library is missing)
  ops.py       host-side mirror of xllm::kernel::* (xllm/core/kernels/ops_api.h) over torch tensors
  attention.py AttentionMetadata + AttentionImpl mirror (xllm/core/layers/dcu/attention.h:31-51)
  parallel.py  ProcessGroup mirror: TP all-reduce / all-gather over RCCL on a side stream
  layers.py    Qwen2 decoder layer call order (xllm/core/layers/qwen2_decoder_layer.cpp:87-110)
"""
__all__ = ["ops", "attention", "parallel", "layers"]
