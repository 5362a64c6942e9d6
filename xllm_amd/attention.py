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

from . import ops


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
                output: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, None]:
        T = query.size(0)
        q = query.view(T, self.num_heads, self.head_size) if query.dim() == 2 and query.is_contiguous() \
            else query.unflatten(-1, (self.num_heads, self.head_size))
        k = key.unflatten(-1, (self.num_kv_heads, self.head_size)) if key.dim() == 2 else key
        v = value.unflatten(-1, (self.num_kv_heads, self.head_size)) if value.dim() == 2 else value
        kc, vc = kv_cache.get_k_cache(), kv_cache.get_v_cache()
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
