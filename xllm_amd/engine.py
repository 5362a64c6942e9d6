"""Step-level decode harness: the shape contract of LLMEngine::step for the decode phase (SURVEY 8f N2 + N3).

Reference: one engine step = BatchInputBuilder (host) -> ForwardInput H2D -> HIP-graph replay of the model with the
persistent metadata buffers refreshed on the device (runtime/dcu_graph_executor_impl.{h,cpp},
kernels/cuda/llm_decode_metadata_update.cu:27-60) -> sampler (framework/sampling/, kernels/dcu/random_sample.hip) ->
next tokens back to the host, where the sequences are extended (framework/request/sequence_kv_state.cpp).

What is here: exactly that loop for a fixed set of sequences whose pages are pre-assigned (the scheduler, block manager
and stop-condition handling of the reference are out of scope): every step
  1. builds the batch's indexing data on the HOST with the product's own builder (attention.build_batch_input),
  2. copies it into device staging buffers (one pinned buffer, one H2D copy),
  3. refreshes the graph's persistent metadata on the DEVICE (ops.decode_metadata_update: tokens, positions, slots,
     cumulative lengths, paged-KV triple, dense block table + per-sequence lengths),
  4. replays ONE captured HIP graph: embedding -> L decoder layers (KV write included) -> lm_head -> sampler,
  5. reads the sampled tokens back (the only host sync of the step).
Sampling: greedy argmax, or temperature (+ top-k / top-p: ops.apply_top_k_top_p, the logits processors of
framework/sampling/logits_utils.cpp as kernels) sampling through ops.softmax_random_sample with the step's uniforms drawn OUTSIDE the
graph (ops.philox_uniform with offset = step), so a replay never repeats random numbers.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from . import attention, ops
from .attention import AttentionMetadata


class DecodeEngine:
    def __init__(self, model, kv_caches, block_size: int, seq_lens: Sequence[int], block_ids_per_seq: Sequence[Sequence[int]],
                 last_tokens: torch.Tensor, max_seq_len: int, temperature: float = 0.0, seed: int = 0,
                 use_graph: bool = True, top_k: Optional[int] = None, top_p: Optional[float] = None):
        """seq_lens[b] tokens of sequence b are already in the cache (the prompt was prefilled); last_tokens[b] is the
        token the next step feeds; block_ids_per_seq[b] are the sequence's pre-assigned pages; max_seq_len bounds every
        sequence's length over the engine's life (the attention launch plan is frozen in the graph)."""
        self.model, self.kv_caches, self.block_size = model, kv_caches, block_size
        self.dev = last_tokens.device
        self.B = len(seq_lens)
        self.seq_lens_init: List[int] = [int(x) for x in seq_lens]
        self.blocks = [list(map(int, b)) for b in block_ids_per_seq]
        self.max_seq_len, self.temperature, self.seed, self.step_no = int(max_seq_len), float(temperature), int(seed), 0
        B = self.B
        self.max_blocks = max(len(b) for b in self.blocks)
        n_idx = sum(len(b) for b in self.blocks)
        i32 = lambda *s: torch.zeros(*s, dtype=torch.int32, device=self.dev)
        # device staging (what the H2D copy fills) and the graph's persistent buffers
        self._layout = [("tokens", B), ("positions", B), ("new_cache_slots", B), ("kv_seq_lens", B + 1),
                        ("paged_kv_indptr", B + 1), ("paged_kv_indices", n_idx), ("paged_kv_last_page_len", B)]
        total = sum(n for _, n in self._layout)
        self._host = torch.zeros(total, dtype=torch.int32).pin_memory() if torch.cuda.is_available() else torch.zeros(total, dtype=torch.int32)
        self._stage = i32(total)
        self.src, off = {}, 0
        for name, n in self._layout:
            self.src[name] = self._stage[off:off + n]
            off += n
        self.dst = {name: i32(n) for name, n in self._layout}
        self.dst.update(kv_seq_lens_delta=i32(B), block_table=i32(B, self.max_blocks), kv_lens=i32(B))
        self.n_idx = n_idx
        self._tokens64 = torch.zeros(B, dtype=torch.int64, device=self.dev)
        self._pos64 = torch.zeros(B, dtype=torch.int64, device=self.dev)
        self._uniform = torch.zeros(B, dtype=torch.float32, device=self.dev)
        self._uniform_for = -1                     # the step whose uniforms self._uniform holds
        # per-sequence sampling parameters (SamplingParameters: temperatures / top_k / top_p tensors, sampling_params.cpp:33-110)
        self._temps = torch.full((B,), float(temperature), dtype=torch.float32, device=self.dev)
        self._top_k = None if top_k is None else torch.full((B,), int(top_k), dtype=torch.int64, device=self.dev)
        self._top_p = None if top_p is None else torch.full((B,), float(top_p), dtype=torch.float32, device=self.dev)
        self._next_host = last_tokens.to(torch.int32).cpu()
        self.md = AttentionMetadata(
            q_cu_seq_lens=torch.arange(B + 1, dtype=torch.int32, device=self.dev), kv_cu_seq_lens=self.dst["kv_seq_lens"],
            kv_seq_lens=self.dst["kv_lens"], slot_mapping=self.dst["new_cache_slots"], block_table=self.dst["block_table"],
            max_query_len=1, max_seq_len=self.max_seq_len, is_prefill=False, is_chunked_prefill=False)
        self._builder = attention.BatchBuilder(self.blocks, block_size, max_tokens=B)
        self._lens_t = torch.tensor(self.seq_lens_init, dtype=torch.int32)
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._use_graph = use_graph
        self._out = None

    # ---- the captured part: everything between the metadata refresh and the sampled tokens
    def _device_step(self):
        self._tokens64.copy_(self.dst["tokens"])
        self._pos64.copy_(self.dst["positions"])
        hidden = self.model.forward(self._tokens64, self._pos64, self.md, self.kv_caches)
        if self.temperature <= 0.0:
            return self.model.greedy_tokens(hidden).to(torch.int32)   # lm_head + argmax in one pass, no [B, V] logits
        # Sampler::forward's random path (sampler.cpp:100-137): temperatures -> top-k -> top-p (apply_top_k_top_p, one sort-free
        # kernel, in place on the lm_head's own 16-bit logits as in the reference) -> softmax in fp32 + random_sample in ONE launch
        # (ops.softmax_random_sample, round 6): no fp32 copy of the logits, no [B, V] probabilities, no vendor operator
        logits = self.model.logits(hidden)
        return ops.sample_top_k_top_p(logits, self._temps, self._top_k, self._top_p, uniform=self._uniform)

    @property
    def seq_lens(self) -> List[int]:
        return self._lens_t.tolist()

    def _host_batch(self) -> None:
        """BatchInputBuilder for a decode step: one new token per sequence, the rest of it cached"""
        bi = self._builder.build(self._lens_t, self._lens_t + 1)
        parts = {"tokens": self._next_host, "positions": bi.positions, "new_cache_slots": bi.new_cache_slots,
                 "kv_seq_lens": bi.kv_cu_seq_lens, "paged_kv_indptr": bi.paged_kv_indptr,
                 "paged_kv_indices": bi.paged_kv_indices, "paged_kv_last_page_len": bi.paged_kv_last_page_len}
        off = 0
        for name, n in self._layout:
            self._host[off:off + n].copy_(parts[name])
            off += n

    def step(self) -> torch.Tensor:
        """one engine step; returns the sampled tokens (host int32 [B]) and extends every sequence by one token"""
        if int(self._lens_t.max()) + 1 > self.max_seq_len:  # (the builder itself rejects a sequence that outgrows its pages)
            raise ValueError("a sequence would outgrow the planned max_seq_len")
        self._host_batch()
        self._stage.copy_(self._host, non_blocking=True)
        ops.decode_metadata_update(self.src, self.dst, self.B, self.B, self.B, self.n_idx, self.B)
        if self.temperature > 0.0 and self._uniform_for != self.step_no:   # (only the first step: see below)
            ops.philox_uniform(self.B, self.seed, self.step_no, device=self.dev, out=self._uniform)
            self._uniform_for = self.step_no
        if not self._use_graph:
            self._out = self._device_step()
        elif self._graph is None:
            self._device_step()                       # warm-up outside the capture (workspaces, lazy init)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._out = self._device_step()
            self._graph = g
            g.replay()
        else:
            self._graph.replay()
        if self.temperature > 0.0:
            # the NEXT step's uniforms, drawn behind this step in stream order (offset = step: a replay never repeats random
            # numbers) -- the launch's host time hides under the step the device is still running
            ops.philox_uniform(self.B, self.seed, self.step_no + 1, device=self.dev, out=self._uniform)
            self._uniform_for = self.step_no + 1
        out = self._out.cpu()                          # the step's host sync: the scheduler needs the tokens
        tp = getattr(self.model, "tp", None)
        if tp is not None:
            tp.check()                                 # one-shot all-reduce: a wait that timed out voids the step (raises)
        self._next_host = out
        self._lens_t += 1
        self.step_no += 1
        return out
