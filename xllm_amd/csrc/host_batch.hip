// host_batch.hip -- HOST side of SURVEY 8 row a1: the per-step indexing data of a batch, built on the CPU exactly as the
// reference's BatchInputBuilder does (no device code in this file; it is part of the same library so that the engine
// binds ONE shared object).
//
// Reference: KVCacheState::cache_slots (xllm/core/framework/request/sequence_kv_state.cpp:86-104),
// BatchInputBuilder::setup_kv_cache_info / finalisation (framework/batch/batch_input_builder.cpp:525-537, 739-830,
// 900-938; CUDA / DCU branch: cumulative seq lens with a leading 0, block table padded with 0) and
// build_attention_metadata (layers/common/attention_metadata_builder.cpp:45-244; DCU branch: per-sequence lengths =
// diff of the cumulative ones, is_causal = prefill || chunked, the dummy batch of :214-232).
#include <stdint.h>
#include <string.h>

#include "../../include/xllm_mi355.h"

extern "C" int xllm_mi355_host_cache_slots(const int32_t* block_ids, int64_t n_blocks, int64_t block_size,
                                           int64_t pos_start, int64_t pos_end, int32_t* slots) {
  if (!block_ids || !slots || block_size <= 0 || pos_start < 0 || pos_end < pos_start) return XM_ERR_INVALID;
  if (n_blocks <= 0) return XM_ERR_INVALID;  // reference: CHECK "no cache blocks available"
  if (pos_end > n_blocks * block_size) return XM_ERR_INVALID;
  for (int64_t i = pos_start; i < pos_end; ++i)
    slots[i - pos_start] = (int32_t)(block_ids[i / block_size] * block_size + i % block_size);
  return XM_OK;
}

extern "C" int xllm_mi355_host_build_batch(const int32_t* n_kv_cache_tokens, const int32_t* seq_lens,
                                           const int32_t* block_indptr, const int32_t* block_ids, int64_t num_sequences,
                                           int64_t block_size, xllm_mi355_host_batch_t* out) {
  if (!out || num_sequences < 0 || block_size <= 0) return XM_ERR_INVALID;
  if (num_sequences > 0 && (!n_kv_cache_tokens || !seq_lens || !block_indptr || !block_ids)) return XM_ERR_INVALID;
  const int64_t B = num_sequences;
  // pass 1: sizes
  int64_t n_tokens = 0, n_indices = 0, max_blocks = 0;
  for (int64_t b = 0; b < B; ++b) {
    const int64_t q = (int64_t)seq_lens[b] - n_kv_cache_tokens[b], nb = (int64_t)block_indptr[b + 1] - block_indptr[b];
    if (q < 0 || n_kv_cache_tokens[b] < 0 || nb <= 0 || (int64_t)seq_lens[b] > nb * block_size) return XM_ERR_INVALID;
    n_tokens += q;
    n_indices += nb;
    max_blocks = nb > max_blocks ? nb : max_blocks;
  }
  out->num_sequences = (int32_t)B;
  out->n_tokens = n_tokens;
  out->n_indices = n_indices;
  out->max_blocks = max_blocks;
  if (n_tokens > out->cap_tokens || n_indices > out->cap_indices || B > out->cap_sequences ||
      (out->block_tables && B * max_blocks > out->cap_block_table))
    return XM_ERR_WORKSPACE;  // the caller sizes the buffers from the counts above and calls again
  int32_t q_max = 0, kv_max = 0;
  int64_t t = 0, idx = 0;
  if (out->q_cu_seq_lens) out->q_cu_seq_lens[0] = 0;
  if (out->kv_cu_seq_lens) out->kv_cu_seq_lens[0] = 0;
  if (out->paged_kv_indptr) out->paged_kv_indptr[0] = 0;
  if (out->block_tables) memset(out->block_tables, 0, sizeof(int32_t) * B * max_blocks);  // pad value 0 (:934)
  int64_t q_cu = 0, kv_cu = 0;
  for (int64_t b = 0; b < B; ++b) {
    const int32_t cached = n_kv_cache_tokens[b], len = seq_lens[b], q = len - cached;
    const int32_t* blocks = block_ids + block_indptr[b];
    const int64_t nb = (int64_t)block_indptr[b + 1] - block_indptr[b];
    for (int32_t p = cached; p < len; ++p, ++t) {
      if (out->new_cache_slots) out->new_cache_slots[t] = (int32_t)(blocks[p / block_size] * block_size + p % block_size);
      if (out->positions) out->positions[t] = p;
    }
    for (int64_t j = 0; j < nb; ++j, ++idx) {
      if (out->paged_kv_indices) out->paged_kv_indices[idx] = blocks[j];
      if (out->block_tables) out->block_tables[b * max_blocks + j] = blocks[j];
    }
    if (out->paged_kv_indptr) out->paged_kv_indptr[b + 1] = (int32_t)idx;
    if (out->paged_kv_last_page_len)
      out->paged_kv_last_page_len[b] = (int32_t)(len % block_size == 0 ? block_size : len % block_size);
    q_cu += q;
    kv_cu += len;
    if (out->q_cu_seq_lens) out->q_cu_seq_lens[b + 1] = (int32_t)q_cu;
    if (out->kv_cu_seq_lens) out->kv_cu_seq_lens[b + 1] = (int32_t)kv_cu;
    if (out->q_seq_lens) out->q_seq_lens[b] = q;
    if (out->kv_seq_lens) out->kv_seq_lens[b] = len;
    q_max = q > q_max ? q : q_max;
    kv_max = len > kv_max ? len : kv_max;
  }
  out->q_max_seq_len = q_max;
  out->kv_max_seq_len = kv_max;
  out->total_kv_len = kv_cu;
  return XM_OK;
}

// ---- the contiguous input buffer of a step (runtime/forward_params.h:94-143: ForwardInputBufferPlan::prepare_layout /
// build_host_buffer): every host tensor of a ForwardInput laid out in ONE byte buffer, entries in insertion order, each at
// the next multiple of `alignment` bytes, tails zero-filled, so that the step reaches the device in one H2D copy.
extern "C" int xllm_mi355_host_plan_input_buffer(xllm_mi355_host_buffer_entry_t* entries, int64_t n_entries,
                                                 uint64_t alignment, uint64_t* total_bytes) {
  if ((!entries && n_entries > 0) || n_entries < 0 || !total_bytes) return XM_ERR_INVALID;
  auto align_up = [&](uint64_t v) { return alignment == 0 ? v : (v + alignment - 1) / alignment * alignment; };
  uint64_t total = 0;
  for (int64_t i = 0; i < n_entries; ++i) {
    total = align_up(total);
    entries[i].offset = total;
    entries[i].aligned_bytes = align_up(entries[i].bytes);
    total += entries[i].aligned_bytes;
  }
  *total_bytes = total;
  return XM_OK;
}

extern "C" int xllm_mi355_host_pack_input_buffer(const xllm_mi355_host_buffer_entry_t* entries, int64_t n_entries,
                                                 void* buffer, uint64_t buffer_bytes) {
  if ((!entries && n_entries > 0) || n_entries < 0 || (!buffer && buffer_bytes > 0)) return XM_ERR_INVALID;
  char* base = static_cast<char*>(buffer);
  for (int64_t i = 0; i < n_entries; ++i) {
    const xllm_mi355_host_buffer_entry_t& e = entries[i];
    if (e.offset + e.aligned_bytes > buffer_bytes || e.aligned_bytes < e.bytes) return XM_ERR_WORKSPACE;
    if (e.bytes == 0) continue;
    if (!e.data) return XM_ERR_INVALID;
    __builtin_memcpy(base + e.offset, e.data, e.bytes);
    if (e.aligned_bytes > e.bytes) __builtin_memset(base + e.offset + e.bytes, 0, e.aligned_bytes - e.bytes);
  }
  return XM_OK;
}
