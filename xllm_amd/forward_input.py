"""ForwardInput: what one engine step hands to the model, and how it reaches the device in ONE copy.

Reference: `ForwardInput` (runtime/forward_params.h) = token_ids + positions + `ModelInputParams` (its `AttentionInput`:
`AttentionHostInput` vectors and `AttentionDeviceInput` tensors, framework/model/model_input_params.h:346-432) + sampling
parameters, and the *contiguous input buffer* of forward_params.h:87-175: every host tensor of a step is laid out in one pinned
byte buffer (`ForwardInputBufferPlan`: entries in insertion order, each at the next multiple of 16 bytes, tails zero-filled),
copied to the device once, and the device tensors are VIEWS into that one device buffer (`bind_device_views`). This module is
the hot-path subset of that contract for the CUDA / DCU branch (the fields `build_attention_metadata`,
layers/common/attention_metadata_builder.cpp:45-244, reads), built on the product's own host batch builder:

    fi = forward_input_from_batch(batch_input, token_ids)         # host tensors, reference field names
    dev = fi.to(device)                                            # ONE pinned buffer, ONE H2D copy, views
    md = dev.attention_metadata(is_prefill=..., is_chunked_prefill=...)

Not here (control plane, out of scope): the protobuf / shared-memory transport of ForwardInput between the engine and its
workers (runtime/forward_shared_memory_manager.cpp), multimodal / rec / DiT payloads, beam search state.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Tuple

import torch

from .attention import AttentionMetadata, BatchInput

kForwardInputBufferAlignment = 16          # forward_params.h:48


def align_up(value: int, alignment: int) -> int:
    """forward_params.h:50-55"""
    return value if alignment == 0 else ((value + alignment - 1) // alignment) * alignment


class ForwardInputBufferPlan:
    """forward_params.h:94-175. `add` takes a host tensor and a setter that will receive the device view."""

    def __init__(self):
        self.entries: List[dict] = []

    def add(self, tensor: Optional[torch.Tensor], target: Optional[Callable[[torch.Tensor], None]]) -> bool:
        if tensor is None:                  # an undefined tensor takes no space (:98-100)
            return True
        if tensor.device.type != "cpu":     # (:101-103) the plan only packs host tensors
            return False
        self.entries.append({"host": tensor.contiguous(), "target": target, "offset": 0, "aligned_bytes": 0})
        return True

    def _c_entries(self):
        import ctypes as C
        from . import _lib
        arr = (_lib.HostBufferEntry * max(len(self.entries), 1))()
        for i, e in enumerate(self.entries):
            h = e["host"]
            arr[i].data = h.data_ptr() if h.numel() else None
            arr[i].bytes = h.numel() * h.element_size()
            arr[i].offset, arr[i].aligned_bytes = e["offset"], e["aligned_bytes"]
        return arr

    def prepare_layout(self) -> int:
        """offsets and padded sizes of every entry (the library's host code: xllm_mi355_host_plan_input_buffer)"""
        import ctypes as C
        from . import _lib
        arr, total = self._c_entries(), C.c_uint64(0)
        _lib.check(_lib.lib().xllm_mi355_host_plan_input_buffer(arr, len(self.entries), kForwardInputBufferAlignment,
                                                               C.byref(total)), "host_plan_input_buffer")
        for i, e in enumerate(self.entries):
            e["offset"], e["aligned_bytes"] = int(arr[i].offset), int(arr[i].aligned_bytes)
        return int(total.value)

    def build_host_buffer(self, total_bytes: int, pin: bool = True) -> torch.Tensor:
        """one (pinned) byte buffer holding every entry, tails zero-filled (xllm_mi355_host_pack_input_buffer)"""
        from . import _lib
        buf = torch.empty(max(total_bytes, 1), dtype=torch.uint8)
        if pin and torch.cuda.is_available():
            buf = buf.pin_memory()
        _lib.check(_lib.lib().xllm_mi355_host_pack_input_buffer(self._c_entries(), len(self.entries), buf.data_ptr(), total_bytes),
                   "host_pack_input_buffer")
        return buf

    def bind_device_views(self, device_buffer: torch.Tensor) -> None:
        """every target becomes a typed view of the ONE device buffer (get_tensor_from_blob, :144-160)"""
        for e in self.entries:
            if e["target"] is None:
                continue
            h = e["host"]
            nbytes = h.numel() * h.element_size()
            view = device_buffer[e["offset"]:e["offset"] + nbytes].view(h.dtype).view(h.shape)
            e["target"](view)


@dataclass
class AttentionDeviceInput:
    """framework/model/model_input_params.h:358-381 (the fields of the CUDA / DCU path)"""
    q_seq_lens: Optional[torch.Tensor] = None
    kv_seq_lens: Optional[torch.Tensor] = None
    q_cu_seq_lens: Optional[torch.Tensor] = None
    new_cache_slots: Optional[torch.Tensor] = None
    block_tables: Optional[torch.Tensor] = None
    paged_kv_indptr: Optional[torch.Tensor] = None
    paged_kv_indices: Optional[torch.Tensor] = None
    paged_kv_last_page_len: Optional[torch.Tensor] = None


# the order add_attention_to_plan packs them in (forward_params.h:222-252)
_ATTENTION_PLAN_ORDER = ("q_seq_lens", "kv_seq_lens", "q_cu_seq_lens", "new_cache_slots", "block_tables", "paged_kv_indptr",
                         "paged_kv_indices", "paged_kv_last_page_len")


@dataclass
class SamplingParameters:
    """framework/sampling/sampling_params.h, the tensors add_sampling_to_plan packs (forward_params.h:177-195) that the
    decode path uses"""
    selected_token_idxes: Optional[torch.Tensor] = None
    temperatures: Optional[torch.Tensor] = None
    top_p: Optional[torch.Tensor] = None
    top_k: Optional[torch.Tensor] = None
    sample_idxes: Optional[torch.Tensor] = None
    do_sample: Optional[torch.Tensor] = None


_SAMPLING_PLAN_ORDER = ("selected_token_idxes", "temperatures", "top_p", "top_k", "sample_idxes", "do_sample")


@dataclass
class ModelInputParams:
    """the scalars + AttentionInput of framework/model/model_input_params.h that the attention metadata builder reads"""
    num_sequences: int = 0
    q_max_seq_len: int = 0
    kv_max_seq_len: int = 0
    kv_cu_seq_lens_host: List[int] = field(default_factory=list)     # AttentionHostInput.kv_cu_seq_lens (stays on the host)
    attention: AttentionDeviceInput = field(default_factory=AttentionDeviceInput)


@dataclass
class ForwardInput:
    token_ids: Optional[torch.Tensor] = None
    positions: Optional[torch.Tensor] = None
    input_params: ModelInputParams = field(default_factory=ModelInputParams)
    sampling_params: SamplingParameters = field(default_factory=SamplingParameters)
    input_host_buffer: Optional[torch.Tensor] = None        # the packed step (host), kept so that a replay can re-send it
    input_device_buffer: Optional[torch.Tensor] = None
    device_tensors_ready: bool = False

    def plan(self, target: "ForwardInput") -> Tuple[ForwardInputBufferPlan, bool]:
        """token_ids, positions, attention tensors, sampling tensors -- in that order"""
        p = ForwardInputBufferPlan()
        ok = p.add(self.token_ids, lambda t: setattr(target, "token_ids", t))
        ok = ok and p.add(self.positions, lambda t: setattr(target, "positions", t))
        for name in _ATTENTION_PLAN_ORDER:
            ok = ok and p.add(getattr(self.input_params.attention, name),
                              (lambda n: lambda t: setattr(target.input_params.attention, n, t))(name))
        for name in _SAMPLING_PLAN_ORDER:
            ok = ok and p.add(getattr(self.sampling_params, name),
                              (lambda n: lambda t: setattr(target.sampling_params, n, t))(name))
        return p, ok

    def to(self, device, non_blocking: bool = True) -> "ForwardInput":
        """ForwardInput::to(device) over the contiguous input buffer (forward_params.h:421-470): ONE H2D copy; positions
        become int64 on the device as on the reference's CUDA branch (normalize_positions_for_device, :197-204)"""
        device = torch.device(device)
        out = ForwardInput(input_params=ModelInputParams(
            num_sequences=self.input_params.num_sequences, q_max_seq_len=self.input_params.q_max_seq_len,
            kv_max_seq_len=self.input_params.kv_max_seq_len, kv_cu_seq_lens_host=list(self.input_params.kv_cu_seq_lens_host)))
        plan, ok = self.plan(out)
        if not ok:
            raise ValueError("ForwardInput.to: every tensor of the step must be a host tensor")
        total = plan.prepare_layout()
        host = plan.build_host_buffer(total, pin=device.type == "cuda")
        dev = host.to(device, non_blocking=non_blocking) if device.type != "cpu" else host
        plan.bind_device_views(dev)
        if out.positions is not None and device.type == "cuda" and out.positions.dtype != torch.int64:
            out.positions = out.positions.to(torch.int64)
        out.input_host_buffer, out.input_device_buffer, out.device_tensors_ready = host, dev, True
        return out

    def attention_metadata(self, is_prefill: bool, is_chunked_prefill: bool) -> AttentionMetadata:
        """build_attention_metadata (layers/common/attention_metadata_builder.cpp:45-244) from the device tensors"""
        a, p = self.input_params.attention, self.input_params
        dev = a.kv_seq_lens.device
        kv_cu = torch.tensor(p.kv_cu_seq_lens_host, dtype=torch.int32, device=dev) if p.kv_cu_seq_lens_host else None
        return AttentionMetadata(q_cu_seq_lens=a.q_cu_seq_lens, kv_cu_seq_lens=kv_cu, kv_seq_lens=a.kv_seq_lens,
                                 slot_mapping=a.new_cache_slots, block_table=a.block_tables, max_query_len=p.q_max_seq_len,
                                 max_seq_len=p.kv_max_seq_len, is_prefill=is_prefill, is_chunked_prefill=is_chunked_prefill,
                                 paged_kv_indptr=a.paged_kv_indptr, paged_kv_indices=a.paged_kv_indices,
                                 paged_kv_last_page_len=a.paged_kv_last_page_len)


def forward_input_from_batch(bi: BatchInput, token_ids: torch.Tensor, temperatures: Optional[torch.Tensor] = None,
                             selected_token_idxes: Optional[torch.Tensor] = None) -> ForwardInput:
    """the host-side ForwardInput of one step from the product's batch builder (attention.build_batch_input =
    BatchInputBuilder::setup_kv_cache_info + finalisation, framework/batch/batch_input_builder.cpp:739-938)"""
    att = AttentionDeviceInput(q_seq_lens=bi.q_seq_lens, kv_seq_lens=bi.kv_seq_lens, q_cu_seq_lens=bi.q_cu_seq_lens,
                               new_cache_slots=bi.new_cache_slots, block_tables=bi.block_tables,
                               paged_kv_indptr=bi.paged_kv_indptr, paged_kv_indices=bi.paged_kv_indices,
                               paged_kv_last_page_len=bi.paged_kv_last_page_len)
    params = ModelInputParams(num_sequences=int(bi.q_seq_lens.numel()), q_max_seq_len=int(bi.q_max_seq_len),
                              kv_max_seq_len=int(bi.kv_max_seq_len), kv_cu_seq_lens_host=bi.kv_cu_seq_lens.tolist(),
                              attention=att)
    if selected_token_idxes is None:        # the last token of every sequence (llm_model_base.h:193-204)
        selected_token_idxes = (bi.q_cu_seq_lens[1:] - 1).to(torch.int32)
    samp = SamplingParameters(selected_token_idxes=selected_token_idxes, temperatures=temperatures)
    return ForwardInput(token_ids=token_ids, positions=bi.positions, input_params=params, sampling_params=samp)
