"""AttentionMetadata + AttentionImpl mirror for the MI355 backend.

Reference: xllm/core/layers/common/attention_metadata.h:73-186 (fields), attention_metadata_builder.cpp:45-300
(how they are filled from BatchInputBuilder outputs), and the per-backend AttentionImpl contract
`forward(attn_metadata, q, k, v, kv_cache) -> (out, lse?)` of layers/dcu/attention.h:31-51 with the dispatch of
layers/dcu/flash_attention.cpp:291-376: write K/V at slot_mapping, then prefill / chunked prefill / decode.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib, ops


@dataclass
class AttentionMetadata:
    """subset of layers/common/attention_metadata.h:73-186 used on the CUDA/DCU path"""
    q_cu_seq_lens: torch.Tensor          # int32 [B+1], leading 0
    kv_cu_seq_lens: Optional[torch.Tensor]  # int32 [B+1]
    kv_seq_lens: torch.Tensor            # int32 [B]
    slot_mapping: torch.Tensor           # int32 [T]  (new_cache_slots)
    block_table: torch.Tensor            # int32 [B, max_blocks], 0-padded
    max_query_len: int
    max_seq_len: int
    is_prefill: bool = False
    is_chunked_prefill: bool = False
    paged_kv_indptr: Optional[torch.Tensor] = None
    paged_kv_indices: Optional[torch.Tensor] = None
    paged_kv_last_page_len: Optional[torch.Tensor] = None

    @property
    def is_causal(self) -> bool:  # attention_metadata_builder.cpp:240-241
        return self.is_prefill or self.is_chunked_prefill


@dataclass
class BatchInput:
    """The per-step indexing data of a batch: what BatchInputBuilder leaves in ModelInputParams.attention
    (framework/batch/batch_input_builder.cpp:900-938, CUDA / DCU branch). Host int32 tensors."""
    new_cache_slots: torch.Tensor        # [sum q]
    positions: torch.Tensor              # [sum q]
    paged_kv_indptr: torch.Tensor        # [B+1]
    paged_kv_indices: torch.Tensor       # [sum blocks]
    paged_kv_last_page_len: torch.Tensor  # [B]
    block_tables: torch.Tensor           # [B, max_blocks], padded with 0
    q_cu_seq_lens: torch.Tensor          # [B+1] cumulative, leading 0 (ModelInputParams calls it q_seq_lens)
    kv_cu_seq_lens: torch.Tensor         # [B+1] cumulative, leading 0 (ModelInputParams calls it kv_seq_lens)
    q_seq_lens: torch.Tensor             # [B]
    kv_seq_lens: torch.Tensor            # [B]
    q_max_seq_len: int
    kv_max_seq_len: int
    total_kv_len: int


def cache_slots(block_ids, block_size: int, pos_start: int, pos_end: int) -> torch.Tensor:
    """KVCacheState::cache_slots (framework/request/sequence_kv_state.cpp:86-104)"""
    blocks = torch.as_tensor(block_ids, dtype=torch.int32).contiguous()
    out = torch.empty(max(pos_end - pos_start, 0), dtype=torch.int32)
    _lib.check(_lib.lib().xllm_mi355_host_cache_slots(blocks.data_ptr(), blocks.numel(), block_size, pos_start, pos_end,
                                                      out.data_ptr()), "host_cache_slots")
    return out


def build_batch_input(n_kv_cache_tokens, seq_lens, block_ids_per_seq, block_size: int) -> BatchInput:
    """BatchInputBuilder::setup_kv_cache_info + finalisation (batch_input_builder.cpp:525-537, 739-830, 900-938): sequence
    b has n_kv_cache_tokens[b] tokens in the cache, seq_lens[b] after this step and owns block_ids_per_seq[b]."""
    B = len(seq_lens)
    cached = torch.as_tensor(n_kv_cache_tokens, dtype=torch.int32).contiguous()
    lens = torch.as_tensor(seq_lens, dtype=torch.int32).contiguous()
    indptr = torch.zeros(B + 1, dtype=torch.int32)
    if B:
        indptr[1:] = torch.cumsum(torch.tensor([len(b) for b in block_ids_per_seq], dtype=torch.int64), 0).to(torch.int32)
    flat = torch.tensor([int(x) for b in block_ids_per_seq for x in b], dtype=torch.int32)
    hb = _lib.HostBatch()  # first call with zero capacities: counts only
    rc = _lib.lib().xllm_mi355_host_build_batch(cached.data_ptr(), lens.data_ptr(), indptr.data_ptr(), flat.data_ptr(), B,
                                                block_size, hb)
    if rc not in (0, -4):  # -4 = XM_ERR_WORKSPACE: the counts are valid, the buffers are not sized yet
        _lib.check(rc, "host_build_batch")
    i32 = lambda n: torch.empty(int(n), dtype=torch.int32)
    bufs = dict(new_cache_slots=i32(hb.n_tokens), positions=i32(hb.n_tokens), paged_kv_indptr=i32(B + 1),
                paged_kv_indices=i32(hb.n_indices), paged_kv_last_page_len=i32(B),
                block_tables=torch.empty(B, int(hb.max_blocks), dtype=torch.int32), q_cu_seq_lens=i32(B + 1),
                kv_cu_seq_lens=i32(B + 1), q_seq_lens=i32(B), kv_seq_lens=i32(B))
    hb.cap_tokens, hb.cap_indices, hb.cap_sequences, hb.cap_block_table = hb.n_tokens, hb.n_indices, B, B * hb.max_blocks
    for k, t in bufs.items():
        setattr(hb, k, t.data_ptr())
    _lib.check(_lib.lib().xllm_mi355_host_build_batch(cached.data_ptr(), lens.data_ptr(), indptr.data_ptr(),
                                                      flat.data_ptr(), B, block_size, hb), "host_build_batch")
    return BatchInput(q_max_seq_len=int(hb.q_max_seq_len), kv_max_seq_len=int(hb.kv_max_seq_len),
                      total_kv_len=int(hb.total_kv_len), **bufs)


class BatchBuilder:
    """build_batch_input for a FIXED set of sequences and pages, called every step (the engine's decode loop): the block
    lists are flattened once and the output buffers are reused, so a step costs one C call and no allocation. The
    returned BatchInput aliases the builder's buffers (valid until the next build)."""

    def __init__(self, block_ids_per_seq, block_size: int, max_tokens: int):
        self.B, self.block_size = len(block_ids_per_seq), block_size
        B = self.B
        self._indptr = torch.zeros(B + 1, dtype=torch.int32)
        if B:
            self._indptr[1:] = torch.cumsum(torch.tensor([len(b) for b in block_ids_per_seq], dtype=torch.int64), 0).to(torch.int32)
        self._flat = torch.tensor([int(x) for b in block_ids_per_seq for x in b], dtype=torch.int32)
        n_idx, max_blocks = self._flat.numel(), max((len(b) for b in block_ids_per_seq), default=0)
        i32 = lambda n: torch.empty(int(n), dtype=torch.int32)
        self._bufs = dict(new_cache_slots=i32(max_tokens), positions=i32(max_tokens), paged_kv_indptr=i32(B + 1),
                          paged_kv_indices=i32(n_idx), paged_kv_last_page_len=i32(B),
                          block_tables=torch.empty(B, max_blocks, dtype=torch.int32), q_cu_seq_lens=i32(B + 1),
                          kv_cu_seq_lens=i32(B + 1), q_seq_lens=i32(B), kv_seq_lens=i32(B))
        self._hb = _lib.HostBatch()
        self._hb.cap_tokens, self._hb.cap_indices, self._hb.cap_sequences = max_tokens, n_idx, B
        self._hb.cap_block_table = B * max_blocks
        for k, t in self._bufs.items():
            setattr(self._hb, k, t.data_ptr())
        self._cached, self._lens = i32(B), i32(B)

    def build(self, n_kv_cache_tokens, seq_lens) -> BatchInput:
        self._cached.copy_(torch.as_tensor(n_kv_cache_tokens, dtype=torch.int32))
        self._lens.copy_(torch.as_tensor(seq_lens, dtype=torch.int32))
        hb = self._hb
        _lib.check(_lib.lib().xllm_mi355_host_build_batch(self._cached.data_ptr(), self._lens.data_ptr(),
                                                          self._indptr.data_ptr(), self._flat.data_ptr(), self.B,
                                                          self.block_size, hb), "host_build_batch")
        b = dict(self._bufs)
        b["new_cache_slots"], b["positions"] = b["new_cache_slots"][:hb.n_tokens], b["positions"][:hb.n_tokens]
        return BatchInput(q_max_seq_len=int(hb.q_max_seq_len), kv_max_seq_len=int(hb.kv_max_seq_len),
                          total_kv_len=int(hb.total_kv_len), **b)


def build_attention_metadata(batch: BatchInput, is_prefill: bool, is_chunked_prefill: bool, device) -> "AttentionMetadata":
    """build_attention_metadata, DCU branch (layers/common/attention_metadata_builder.cpp:45-244): cumulative lengths
    straight from the batch, per-sequence lengths = their differences, block table for every phase but pure prefill,
    the dummy batch of :214-232 when no sequence has a query token."""
    to = lambda t: t.to(device, non_blocking=True)
    md = AttentionMetadata(
        q_cu_seq_lens=to(batch.q_cu_seq_lens), kv_cu_seq_lens=to(batch.kv_cu_seq_lens), kv_seq_lens=to(batch.kv_seq_lens),
        slot_mapping=to(batch.new_cache_slots), block_table=to(batch.block_tables),
        max_query_len=batch.q_max_seq_len, max_seq_len=batch.kv_max_seq_len, is_prefill=is_prefill,
        is_chunked_prefill=is_chunked_prefill, paged_kv_indptr=to(batch.paged_kv_indptr),
        paged_kv_indices=to(batch.paged_kv_indices), paged_kv_last_page_len=to(batch.paged_kv_last_page_len))
    if batch.q_max_seq_len == 0:  # is_dummy
        one = torch.ones(1, dtype=torch.int32, device=device)
        md.slot_mapping, md.kv_seq_lens = one, one.clone()
        md.q_cu_seq_lens = torch.tensor([0, 1], dtype=torch.int32, device=device)
        md.max_query_len, md.max_seq_len = 1, max(md.max_seq_len, 1)
    return md


class KVCache:
    """framework/kv_cache: k/v caches [n_blocks, block_size, n_kv_heads_local, head_dim] (kv_cache_shape.cpp:239-267)"""

    def __init__(self, k_cache: torch.Tensor, v_cache: torch.Tensor):
        self.k_cache, self.v_cache = k_cache, v_cache

    def get_k_cache(self):
        return self.k_cache

    def get_v_cache(self):
        return self.v_cache


class AttentionImpl:
    """xllm::layer::AttentionImpl for USE_MI355 (constructor args of layers/dcu/attention.h:33-38)."""

    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: int, sliding_window: int = -1):
        self.num_heads, self.head_size, self.scale = num_heads, head_size, scale
        self.num_kv_heads = num_kv_heads
        # DCU passes sliding_window_ > 0 ? sliding_window_ : -1 (flash_attention.cpp:257)
        self.window_left = sliding_window if sliding_window and sliding_window > 0 else -1

    def forward(self, md: AttentionMetadata, query, key, value, kv_cache: KVCache,
                output: Optional[torch.Tensor] = None, kv_written: bool = False) -> Tuple[torch.Tensor, None]:
        """kv_written: the caller already stored this step's K / V (ops.rotary_embedding_and_cache, the N1 fusion of RoPE
        with the KV write) -- skip the reshape_paged_cache below"""
        T = query.size(0)
        q = query.view(T, self.num_heads, self.head_size) if query.dim() == 2 and query.is_contiguous() \
            else query.unflatten(-1, (self.num_heads, self.head_size))
        k = key.unflatten(-1, (self.num_kv_heads, self.head_size)) if key.dim() == 2 else key
        v = value.unflatten(-1, (self.num_kv_heads, self.head_size)) if value.dim() == 2 else value
        kc, vc = kv_cache.get_k_cache(), kv_cache.get_v_cache()
        if not kv_written:
            ops.reshape_paged_cache(md.slot_mapping, k, v, kc, vc)  # flash_attention.cpp:310-318
        if md.is_prefill:
            out = ops.prefill_attention(q, k, v, md.q_cu_seq_lens, md.kv_cu_seq_lens, md.max_query_len, self.scale,
                                        True, self.window_left, out=output)
        elif md.is_chunked_prefill:
            out = ops.paged_attention(q, kc, vc, md.q_cu_seq_lens, md.kv_seq_lens, md.block_table, md.max_query_len,
                                      md.max_seq_len, self.scale, True, self.window_left, out=output)
        else:
            out = ops.paged_attention(q, kc, vc, None, md.kv_seq_lens, md.block_table, 1, md.max_seq_len, self.scale,
                                      False, self.window_left, out=output)
        return out, None
