// attention_api.hip -- C-ABI entry points for the attention modes (dispatch only; kernels live in
// attention_decode.hip / attention_prefill.hip / attention_mla.hip).
#include <stdlib.h>

#include "common.h"

namespace xm {

template <typename T, int D>
int launch_paged_decode(const void* q, const void* kc, const void* vc, void* out, const int32_t* cu_q,
                        const int32_t* kv_lens, const int32_t* block_table, int64_t max_blocks, int64_t batch,
                        int64_t nq, int64_t nkv, int64_t block_size, int64_t q_stride, int64_t max_kv_len,
                        float scale, int64_t window_left, void* workspace, size_t ws_bytes, hipStream_t s,
                        int8_t* out_q, float* out_scale);

template <typename T, int D, bool PAGED>
int launch_flash_prefill(const void* q, const void* k, const void* v, void* out, const int32_t* cu_q,
                         const int32_t* cu_k, const int32_t* kv_lens, const int32_t* block_table, int64_t max_blocks,
                         int64_t batch, int64_t nq, int64_t nkv, int64_t block_size, int64_t q_stride,
                         int64_t k_stride, int64_t v_stride, int64_t max_q_len, float scale, int causal,
                         int64_t window_left, hipStream_t s);

// XLLM_MI355_DECODE_SPLITS (product switch, read once): force the grid-level split-KV count -- the parity tests use it to cover
// the split counts the planner picks at other batch sizes. Tuning arms: 3-stage prefetch (round-1 A/B: 5.87 vs 5.93 TB/s with
// 2 stages), forced heads per workgroup, exclusive-CU LDS padding.
static int g_split_override = -2;  // -2: env not read yet; -1: no override
XM_TUNE_VAR(g_deep, "XLLM_MI355_DECODE_DEEP", 0);
XM_TUNE_VAR(g_hpw_override, "XLLM_MI355_DECODE_HPW", -1);
XM_TUNE_VAR(g_excl, "XLLM_MI355_DECODE_EXCL", 0);

// grid-level split-KV count of the decode kernel. Target: ~256 workgroups = ONE per CU (4 waves each, 16-32 KiB
// in flight per wave): the round-1 sweep (tools/attn_bench.py, profiles/r01_attn_split_sweep.txt) shows that
// fewer, longer token streams beat more resident waves (no split at B*nkv/hpw >= 256: 5.90 TB/s vs 5.62 TB/s
// with 2 splits; B=64: 4 splits best), every wave keeping >= 8 tiles of 32 tokens.
int decode_num_splits(int64_t batch, int64_t nkv, int hpw, int64_t max_kv_len) {
  if (g_split_override == -2) g_split_override = xm_switch("XLLM_MI355_DECODE_SPLITS", -1);
  const int nsub = 4 / hpw;
  const int64_t base = batch * (nkv / hpw);
  const int64_t tiles = (max_kv_len + 31) / 32;
  int64_t by_len = tiles / (8 * nsub);
  if (by_len < 1) by_len = 1;
  int64_t n;
  if (g_split_override > 0) n = g_split_override;
  else {
    n = (256 + base - 1) / base;
    if (n > by_len) n = by_len;
  }
  if (n > 32) n = 32;
  if (n < 1) n = 1;
  return (int)n;
}

bool decode_deep_prefetch() { return g_deep != 0; }

// heads per workgroup of the decode kernel (4 waves = hpw kv heads x 4/hpw sub-ranges of the token range, merged in LDS).
// All kv heads of a token in one workgroup (hpw = 4) read whole token rows and allow the fused int8 epilogue, but give only
// batch * nkv / 4 workgroups; below ~one workgroup per CU the parallelism has to come from somewhere, and sub-ranges inside
// the workgroup are free (LDS merge) while grid-level splits pay partial writes + a merge launch. So: the largest hpw that
// still yields >= 192 workgroups, else hpw = 1 and the rest through grid-level splits.
int decode_heads_per_wg(int64_t batch, int64_t nkv) {
  if (g_hpw_override > 0 && 4 % g_hpw_override == 0 && nkv % g_hpw_override == 0) return g_hpw_override;
  for (int hpw = 4; hpw > 1; hpw >>= 1)
    if (nkv % hpw == 0 && batch * (nkv / hpw) >= 192) return hpw;
  return 1;
}

int decode_exclusive_cu() { return g_excl; }

}  // namespace xm

using namespace xm;

#ifdef XM_TUNING
// analysis (tools/step_ab.py idle=...): ONE wave that does nothing for `us` microseconds (s_sleep loop on the 100-MHz wall clock): a
// controlled idle gap in front of / behind a kernel of the step, to separate what a kernel costs from the state its predecessor leaves
__global__ void debug_idle_kernel(long long ticks) {
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
extern "C" __attribute__((visibility("default"))) int xllm_mi355_debug_idle(double us, void* stream) {
  if (us <= 0) return 0;
  hipLaunchKernelGGL(debug_idle_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long)(us * 100.0));
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

// tuning (tools/step_ab.py): the decode kernel's launch plan for the following calls; < 0 = leave as it is, 0 = planner / default
extern "C" __attribute__((visibility("default"))) void xllm_mi355_debug_decode_plan(int splits, int hpw, int deep, int excl) {
  (void)decode_num_splits(1, 4, 4, 4096);  // env first
  if (splits >= 0) g_split_override = splits > 0 ? splits : -1;
  if (hpw >= 0) g_hpw_override = hpw > 0 ? hpw : -1;
  if (deep >= 0) g_deep = deep;
  if (excl >= 0) g_excl = excl;
}
#endif

extern "C" {

size_t xllm_mi355_paged_attention_workspace_bytes(int64_t batch, int64_t n_q_heads, int64_t head_dim_v,
                                                  int64_t max_q_len, int64_t total_q_tokens) {
  (void)total_q_tokens;
  if (max_q_len > 1) return 0;  // chunked prefill needs no workspace
  return (size_t)batch * n_q_heads * 32 * (head_dim_v + 2) * sizeof(float);  // 32 = max split-KV count
}

int xllm_mi355_paged_attention(const void* q, const void* k_cache, const void* v_cache, void* out,
                               const int32_t* cu_q, const int32_t* kv_lens, const int32_t* block_table,
                               int64_t max_blocks, int64_t batch, int64_t total_q_tokens, int64_t n_q_heads,
                               int64_t n_kv_heads, int64_t head_dim, int64_t block_size, int64_t n_blocks,
                               int64_t q_stride, int64_t max_q_len, int64_t max_kv_len, float scale, int causal,
                               int64_t window_left, int dtype, void* workspace, size_t workspace_bytes,
                               void* stream) {
  (void)n_blocks;
  if (!q || !k_cache || !v_cache || !out || !kv_lens || !block_table) return XM_ERR_INVALID;
  if (batch < 0 || n_q_heads <= 0 || n_kv_heads <= 0 || n_q_heads % n_kv_heads || block_size <= 0 || max_blocks <= 0)
    return XM_ERR_INVALID;
  if (batch == 0 || total_q_tokens == 0) return XM_OK;
  if ((uintptr_t)q % 16 || (uintptr_t)k_cache % 16 || (uintptr_t)v_cache % 16 || q_stride % 8) return XM_ERR_UNSUPPORTED;
  if (n_q_heads / n_kv_heads > 16) return XM_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (max_q_len <= 1) {
    // decode: one query token per sequence; the causal flag is irrelevant (the token sees all kv_len keys)
    size_t ws = workspace ? workspace_bytes : 0;
#define XM_DECODE(T, DD)                                                                                        \
  return launch_paged_decode<T, DD>(q, k_cache, v_cache, out, cu_q, kv_lens, block_table, max_blocks, batch,    \
                                    n_q_heads, n_kv_heads, block_size, q_stride, max_kv_len, scale, window_left, \
                                    workspace, ws, s, nullptr, nullptr)
    if (dtype == XM_BF16 && head_dim == 128) XM_DECODE(bf16_t, 128);
    if (dtype == XM_BF16 && head_dim == 64) XM_DECODE(bf16_t, 64);
    if (dtype == XM_F16 && head_dim == 128) XM_DECODE(f16_t, 128);
    if (dtype == XM_F16 && head_dim == 64) XM_DECODE(f16_t, 64);
#undef XM_DECODE
    return XM_ERR_UNSUPPORTED;
  }
  if (!cu_q) return XM_ERR_INVALID;
#define XM_CHUNKED(T, DD)                                                                                         \
  return launch_flash_prefill<T, DD, true>(q, k_cache, v_cache, out, cu_q, nullptr, kv_lens, block_table,         \
                                           max_blocks, batch, n_q_heads, n_kv_heads, block_size, q_stride, 0, 0, \
                                           max_q_len, scale, causal, window_left, s)
  if (dtype == XM_BF16 && head_dim == 128) XM_CHUNKED(bf16_t, 128);
  if (dtype == XM_BF16 && head_dim == 64) XM_CHUNKED(bf16_t, 64);
  if (dtype == XM_F16 && head_dim == 128) XM_CHUNKED(f16_t, 128);
  if (dtype == XM_F16 && head_dim == 64) XM_CHUNKED(f16_t, 64);
#undef XM_CHUNKED
  return XM_ERR_UNSUPPORTED;
}

int xllm_mi355_paged_decode_attention_int8(const void* q, const void* k_cache, const void* v_cache, void* out,
                                           int8_t* out_q, float* out_scale, const int32_t* kv_lens,
                                           const int32_t* block_table, int64_t max_blocks, int64_t batch,
                                           int64_t n_q_heads, int64_t n_kv_heads, int64_t head_dim, int64_t block_size,
                                           int64_t q_stride, int64_t max_kv_len, float scale, int64_t window_left,
                                           int dtype, void* stream) {
  return xllm_mi355_paged_decode_attention_int8_ws(q, k_cache, v_cache, out, out_q, out_scale, kv_lens, block_table, max_blocks,
                                                   batch, n_q_heads, n_kv_heads, head_dim, block_size, q_stride, max_kv_len,
                                                   scale, window_left, dtype, nullptr, 0, stream);
}

int xllm_mi355_paged_decode_attention_int8_ws(const void* q, const void* k_cache, const void* v_cache, void* out,
                                              int8_t* out_q, float* out_scale, const int32_t* kv_lens,
                                              const int32_t* block_table, int64_t max_blocks, int64_t batch,
                                              int64_t n_q_heads, int64_t n_kv_heads, int64_t head_dim, int64_t block_size,
                                              int64_t q_stride, int64_t max_kv_len, float scale, int64_t window_left,
                                              int dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (!q || !k_cache || !v_cache || !out_q || !out_scale || !kv_lens || !block_table) return XM_ERR_INVALID;
  if (batch < 0 || n_q_heads <= 0 || n_kv_heads <= 0 || n_q_heads % n_kv_heads || block_size <= 0 || max_blocks <= 0)
    return XM_ERR_INVALID;
  if (batch == 0) return XM_OK;
  if ((uintptr_t)q % 16 || (uintptr_t)k_cache % 16 || (uintptr_t)v_cache % 16 || q_stride % 8) return XM_ERR_UNSUPPORTED;
  if (n_q_heads / n_kv_heads > 16) return XM_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
#define XM_DECODEQ(T, DD)                                                                                            \
  return launch_paged_decode<T, DD>(q, k_cache, v_cache, out, nullptr, kv_lens, block_table, max_blocks, batch,      \
                                    n_q_heads, n_kv_heads, block_size, q_stride, max_kv_len, scale, window_left,     \
                                    workspace, workspace ? workspace_bytes : 0, s, out_q, out_scale)
  if (dtype == XM_BF16 && head_dim == 128) XM_DECODEQ(bf16_t, 128);
  if (dtype == XM_BF16 && head_dim == 64) XM_DECODEQ(bf16_t, 64);
  if (dtype == XM_F16 && head_dim == 128) XM_DECODEQ(f16_t, 128);
  if (dtype == XM_F16 && head_dim == 64) XM_DECODEQ(f16_t, 64);
#undef XM_DECODEQ
  return XM_ERR_UNSUPPORTED;
}

int xllm_mi355_prefill_attention(const void* q, const void* k, const void* v, void* out, const int32_t* cu_q,
                                 const int32_t* cu_k, int64_t batch, int64_t n_q_heads, int64_t n_kv_heads,
                                 int64_t head_dim, int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                 int64_t max_q_len, float scale, int causal, int64_t window_left, int dtype,
                                 void* stream) {
  if (!q || !k || !v || !out || !cu_q || !cu_k) return XM_ERR_INVALID;
  if (batch < 0 || n_q_heads <= 0 || n_kv_heads <= 0 || n_q_heads % n_kv_heads) return XM_ERR_INVALID;
  if (batch == 0 || max_q_len == 0) return XM_OK;
  if ((uintptr_t)q % 16 || (uintptr_t)k % 16 || (uintptr_t)v % 16 || q_stride % 8 || k_stride % 8 || v_stride % 8)
    return XM_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
#define XM_PREFILL(T, DD)                                                                                       \
  return launch_flash_prefill<T, DD, false>(q, k, v, out, cu_q, cu_k, nullptr, nullptr, 0, batch, n_q_heads,    \
                                            n_kv_heads, 1, q_stride, k_stride, v_stride, max_q_len, scale, causal, \
                                            window_left, s)
  if (dtype == XM_BF16 && head_dim == 128) XM_PREFILL(bf16_t, 128);
  if (dtype == XM_BF16 && head_dim == 64) XM_PREFILL(bf16_t, 64);
  if (dtype == XM_F16 && head_dim == 128) XM_PREFILL(f16_t, 128);
  if (dtype == XM_F16 && head_dim == 64) XM_PREFILL(f16_t, 64);
#undef XM_PREFILL
  return XM_ERR_UNSUPPORTED;
}

}  // extern "C"
