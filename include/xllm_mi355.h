/*
 * xllm_mi355.h -- C ABI of the MI355X (gfx950 / CDNA4) kernel backend for xLLM's
 * decode/prefill hot path.
 *
 * Every entry point replaces one operator of the reference's kernel boundary
 * (xllm/core/kernels/ops_api.h, per-backend headers kernels/cuda/cuda_ops_api.h and
 * kernels/dcu/dcu_ops_api.h) or one attention mode of the per-backend AttentionImpl
 * (xllm/core/layers/dcu/attention.h:31-51).  The reference interface each one binds to is
 * cited on the declaration.  INTEGRATION.md shows the `#elif defined(USE_MI355)` shim a
 * maintainer adds to ops_api.cpp (shim/mi355_ops_api.{h,cpp} in this repo is that shim).
 *
 * Conventions
 *   - plain pointers and sizes; all pointers are DEVICE pointers unless stated; nothing is
 *     allocated inside; `workspace` (where present) is caller provided, size from the
 *     matching *_workspace_bytes() call.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), performs no host
 *     synchronisation and reads no device data on the host => HIP-graph capturable
 *     (the reference's DCU path is not: kernels/dcu/group_gemm.cpp:45).
 *   - return 0 on success, <0 on error (XM_ERR_*); xllm_mi355_strerror() maps codes to text.
 *     The C++ shim turns non-zero into TORCH_CHECK(false, ...) (reference: CHECK/TORCH_CHECK).
 *   - dtype codes XM_F32/XM_BF16/XM_F16 select the 16/32-bit "scalar_t" of the reference's
 *     DISPATCH_FLOATING_TYPES.
 */
#ifndef XLLM_MI355_H_
#define XLLM_MI355_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XM_API __attribute__((visibility("default")))

enum { XM_F32 = 0, XM_BF16 = 1, XM_F16 = 2 };
enum {
  XM_OK = 0,
  XM_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, misalignment) */
  XM_ERR_UNSUPPORTED = -2, /* shape / dtype outside what the kernels implement */
  XM_ERR_HIP = -3,         /* a HIP runtime call failed (launch error) */
  XM_ERR_WORKSPACE = -4    /* workspace too small */
};
enum { XM_ACT_SILU = 0, XM_ACT_GELU = 1, XM_ACT_GELU_TANH = 2 };

XM_API const char* xllm_mi355_strerror(int code);
/* The ABI version this header describes; xllm_mi355_abi_version() returns the one the library was built from. A binding
 * (ctypes, the libtorch shim, cgo ...) written against this header must refuse a library that reports another number.
 * 2: `grid_limit` inserted before `stream` in the two oneshot_allreduce_add_rms_norm entry points; block_copy, build_digest. */
#define XLLM_MI355_ABI_VERSION 2
XM_API int xllm_mi355_abi_version(void);
/* sha256 (hex) over the library's sources (xllm_amd/csrc/{*.hip,*.h,Makefile} + this header) at build time -- the value
 * `python tools/source_digest.py --lib` prints for the tree: a prebuilt .so that does not match its sources is detectable. */
XM_API const char* xllm_mi355_build_digest(void);

/* ---- KV write ------------------------------------------------------------------------------
 * kernel::reshape_paged_cache (ops_api.h:31) -> cuda::reshape_paged_cache
 * (kernels/cuda/reshape_paged_cache.cu:65-100).  k/v [T, nkv, d] with token strides (elements),
 * caches [n_blocks, block_size, nkv, d]; v and v_cache may both be NULL (K-only cache: the MLA latent cache,
 * DeepseekV2AttentionImpl::store_latent_cache, layers/dcu/deepseek_v2_attention.cpp:170-178); slot<0 skipped; slot/block_size >= n_blocks is silently
 * skipped as out of range (the reference would write out of bounds).  elt_bytes 2 or 4. */
XM_API int xllm_mi355_reshape_paged_cache(const int32_t* slot_ids, const void* k, const void* v,
                                          void* k_cache, void* v_cache, int64_t n_tokens,
                                          int64_t n_kv_heads, int64_t head_dim, int64_t block_size,
                                          int64_t n_blocks, int64_t k_stride, int64_t v_stride,
                                          int elt_bytes, void* stream);

/* cuda::block_copy (kernels/cuda/cuda_ops_api.h:50-56, kernels/cuda/block_copy.cu:56-205), called by
 * WorkerImpl::execute_cuda_block_copy_kernel (runtime/worker_impl.cpp:1071-1082) for beam-search / prefix forks: destination j
 * (0 <= j < num_dst_blocks) belongs to source group g = the first g with j < cum_sum[g]; for every layer l,
 * K_l[dst[j]] <- K_l[src[g]] and V_l[dst[j]] <- V_l[src[g]] (whole cache blocks of bytes_per_block bytes).
 * k_cache_ptrs / v_cache_ptrs: DEVICE arrays [num_layers] of cache base addresses as int64 (the reference's layout);
 * v_cache_ptrs may be NULL (K-only caches: the MLA latent cache). A destination that is also a source of the same launch
 * is undefined, as in the reference. Byte copy: any cache dtype.  Cache base addresses that are not a multiple of 16 bytes
 * (a view at an odd offset; allocator-returned tensors never are) are served by a byte-wise walk instead of vector accesses. */
XM_API int xllm_mi355_block_copy(const int64_t* k_cache_ptrs, const int64_t* v_cache_ptrs,
                                 const int32_t* src_block_indices, const int32_t* dst_block_indices,
                                 const int32_t* cum_sum, int64_t num_layers, int64_t num_groups,
                                 int64_t num_dst_blocks, int64_t bytes_per_block, void* stream);

/* dcu::build_block_table_from_paged_kv_cuda (kernels/dcu/build_block_table_from_paged_kv.hip:74-110):
 * CSR (indptr[B+1], indices[total_pages]) -> dense [B, total_pages] int32, -1 padded. */
XM_API int xllm_mi355_build_block_table_from_paged_kv(const int32_t* indptr, const int32_t* indices,
                                                      int32_t batch, int32_t total_pages,
                                                      int32_t* block_table, void* stream);

/* N1 fusion across the GEMM boundary (decode shapes, M <= 512): the row-parallel W8A8 linear followed by the
 * residual add + RMSNorm (+ per-token int8 quant) of the next half-layer -- kernel::scaled_matmul (ops_api.h:98) then
 * kernel::fused_layernorm with residual (ops_api.h:43; MLU's fused_layernorm(dynamic_quant) for the quantised form) --
 * in two launches instead of four: the GEMM leaves exact int32 sums in the registered workspace
 * (xllm_mi355_set_gemm_workspace, >= 4*M*N bytes, XM_ERR_WORKSPACE otherwise) and one row kernel dequantises them
 * (r16(acc*a_scale[m]*w_scale[n] + bias[n])), adds `residual` in place (residual <- r16(y + residual)), normalises and
 * writes either the 16-bit norm (out_norm) or its int8 quantisation (out_q, out_q_scale) -- exactly one of the two.
 * Bit-identical to xllm_mi355_scaled_matmul -> xllm_mi355_fused_add_rms_norm / _rms_norm_dynamic_int8_quant.
 * Not applicable when a TP all-reduce sits between the linear and the norm. */
XM_API int xllm_mi355_scaled_matmul_add_rms_norm(const int8_t* a, const int8_t* w, const float* a_scale,
                                                 const float* w_scale, const void* bias, void* residual,
                                                 const void* norm_weight, float eps, void* out_norm,
                                                 int8_t* out_q, float* out_q_scale, int64_t M, int64_t N,
                                                 int64_t K, int dtype, void* stream);

/* N2: device-side refresh of the persistent decode metadata a replayed HIP graph reads.
 * cuda::update_llm_decode_metadata (kernels/cuda/llm_decode_metadata_update.cu:27-60, params struct
 * llm_decode_metadata_update.h:34-54; caller runtime/cuda_graph_executor_impl.cpp:218-258): copies tokens /
 * positions / new cache slots (tokens and slots of the padded tail are zeroed, positions are left alone),
 * kv_seq_lens and paged_kv_indptr (B+1 entries, cumulative form), kv_seq_lens_delta[b] = kv_seq_lens[b+1] -
 * kv_seq_lens[b], paged_kv_last_page_len and paged_kv_indices from the step's staging buffers into the
 * persistent ones.  MI355X extension, same launch: when dst_block_table is non-null the dense 0-padded block
 * table [actual_batch_size, max_blocks_per_seq] that xllm_mi355_paged_attention reads is rebuilt from the CSR
 * arrays (the DCU path needs a second kernel for that, build_block_table_from_paged_kv.hip), rows of the
 * padded batch tail [actual_batch_size, padded_batch_size) are zeroed and dst_kv_lens (per-sequence lengths
 * = the deltas, 0 for the padded tail) is written.  Any pointer pair may be null (skipped). */
typedef struct xllm_mi355_decode_metadata {
  const int32_t* src_tokens;
  const int32_t* src_positions;
  const int32_t* src_new_cache_slots;
  const int32_t* src_kv_seq_lens;
  const int32_t* src_paged_kv_indptr;
  const int32_t* src_paged_kv_indices;
  const int32_t* src_paged_kv_last_page_len;
  int32_t* dst_tokens;
  int32_t* dst_positions;
  int32_t* dst_new_cache_slots;
  int32_t* dst_kv_seq_lens;
  int32_t* dst_kv_seq_lens_delta;
  int32_t* dst_paged_kv_indptr;
  int32_t* dst_paged_kv_indices;
  int32_t* dst_paged_kv_last_page_len;
  int64_t actual_num_tokens;
  int64_t padded_num_tokens;
  int64_t actual_batch_size;
  int64_t actual_indices_size;
  /* MI355X extension (all optional) */
  int32_t* dst_block_table;
  int32_t* dst_kv_lens;
  int64_t max_blocks_per_seq;
  int64_t padded_batch_size;
} xllm_mi355_decode_metadata_t;
XM_API int xllm_mi355_decode_metadata_update(const xllm_mi355_decode_metadata_t* params, void* stream);

/* ---- RMSNorm family ------------------------------------------------------------------------
 * kernel::fused_layernorm (ops_api.h:43) -> cuda::rms_norm / cuda::fused_add_rms_norm
 * (kernels/cuda/norm.cu:430-512).  in_stride = token stride of `input` in elements. */
XM_API int xllm_mi355_rms_norm(void* out, const void* input, const void* weight, float eps,
                               int64_t n_tokens, int64_t hidden, int64_t in_stride, int dtype,
                               void* stream);
XM_API int xllm_mi355_fused_add_rms_norm(void* input, void* residual, const void* weight, float eps,
                                         int64_t n_tokens, int64_t hidden, int64_t in_stride,
                                         int dtype, void* stream);
/* kernel::rms_norm_static_fp8_quant / fused_add_rms_norm_static_fp8_quant (ops_api.h:168,172)
 * -> kernels/cuda/norm.cu:517-640.  residual may be NULL (no add).  out is e4m3fn bytes. */
XM_API int xllm_mi355_rms_norm_static_fp8_quant(uint8_t* out, const void* input, void* residual,
                                                const void* weight, const float* scale, float eps,
                                                int64_t n_tokens, int64_t hidden, int64_t in_stride,
                                                int dtype, void* stream);
/* "next" N1 fusion (MLU fused_layernorm(dynamic_quant) param.h:243-277): norm (+residual) then
 * per-token int8 quant of the normalised row in one pass: out_q [T,H] int8, out_scale [T] f32.
 * Arithmetic == rms_norm followed by scaled_quantize on its 16-bit output. residual may be NULL. */
XM_API int xllm_mi355_rms_norm_dynamic_int8_quant(int8_t* out_q, float* out_scale, const void* input,
                                                  void* residual, const void* weight, float eps,
                                                  int64_t n_tokens, int64_t hidden, int64_t in_stride,
                                                  int dtype, void* stream);

/* ---- RoPE ----------------------------------------------------------------------------------
 * kernel::apply_rotary (ops_api.h:27) -> cuda::rotary_embedding (kernels/cuda/rope.cu:156-250).
 * positions int64 [T]; q [T, nq*head_size], k optional (NULL) [T, nk*head_size], token strides in
 * elements, head_stride = head_size; cos_sin_cache [max_pos, rot_dim] = [cos(rot/2) || sin(rot/2)]. */
XM_API int xllm_mi355_rotary_embedding(const int64_t* positions, void* q, void* k,
                                       const void* cos_sin_cache, int64_t n_tokens, int64_t n_q_heads,
                                       int64_t n_k_heads, int64_t head_size, int64_t rot_dim,
                                       int64_t q_stride, int64_t k_stride, int64_t head_stride,
                                       int is_neox, int dtype, void* stream);
/* cuda::fused_qk_norm_rope (kernels/cuda/cuda_ops_api.h:235-249, fused_qknorm_rope.cu:388-...):
 * per-head RMSNorm(q),(k) + RoPE inside packed qkv [T,(nq+nk+nv)*d]; cache dtype = cache_dtype. */
XM_API int xllm_mi355_fused_qk_norm_rope(void* qkv, int64_t n_tokens, int64_t n_q, int64_t n_k,
                                         int64_t n_v, int64_t head_dim, float eps, const void* q_weight,
                                         const void* k_weight, const void* cos_sin_cache,
                                         int cache_dtype, int interleaved, const int64_t* positions,
                                         int dtype, void* stream);

/* ---- activation ----------------------------------------------------------------------------
 * kernel::active (ops_api.h:29) -> cuda::act_and_mul (kernels/cuda/activation.cu:143-185).
 * input [T, 2*d] contiguous, out [T, d]. */
XM_API int xllm_mi355_act_and_mul(void* out, const void* input, int64_t n_tokens, int64_t d,
                                  int act_mode, int dtype, void* stream);
/* "next" N1 fusion (ScaledQuantizeParams.act_mode/is_gated, param.h:805-815): silu(gate)*up then
 * per-token int8 quant; == act_and_mul followed by scaled_quantize. */
XM_API int xllm_mi355_act_and_mul_dynamic_int8_quant(int8_t* out_q, float* out_scale, const void* input,
                                                     int64_t n_tokens, int64_t d, int act_mode,
                                                     int dtype, void* stream);
/* the same operator on the sorted rows of one expert-parallel rank (FusedMoEImpl::forward_experts with EP, layers/dcu/
 * fused_moe.cpp:236-315): only the first sum(live_sizes[0 .. n_sizes)) rows exist (the rank's own experts sort to the front);
 * the count is read on the device; a row past it is not read, its scale is written as 0 and its quantised bytes are left as
 * they are (the same on every kernel path). live_sizes == NULL: every row. */
XM_API int xllm_mi355_act_and_mul_dynamic_int8_quant_live(int8_t* out_q, float* out_scale, const void* input,
                                                          int64_t n_tokens, int64_t d, int act_mode, int dtype,
                                                          const int32_t* live_sizes, int64_t n_sizes, void* stream);

/* ---- int8 W8A8 -----------------------------------------------------------------------------
 * kernel::scaled_quantize (ops_api.h:95) -> dcu::scaled_quantize (kernels/dcu/scaled_quantize.hip:411-...)
 * per-token symmetric int8: x [M,K] (dtype) -> q [M,K] int8, scale [M] f32. */
XM_API int xllm_mi355_scaled_quantize(const void* x, int8_t* out, float* out_scale, int64_t M,
                                      int64_t K, int dtype, void* stream);
/* kernel::scaled_matmul (ops_api.h:98) -> dcu::scaled_matmul (kernels/dcu/scaled_matmul.cpp:103-300):
 * out[m,n] = r16( int32(sum_k a[m,k]*w[n,k]) * a_scale[m] * w_scale[n] + bias[n] ); a [M,K] int8,
 * w [N,K] int8 row-major, bias (out dtype) may be NULL; out dtype XM_BF16 / XM_F16. K % 128 == 0
 * (the K step of every GEMM kernel is 128 bytes of each operand row).
 * acc_out (optional, may be NULL): raw int32 accumulators [M,N] (parity tests). */
XM_API int xllm_mi355_scaled_matmul(const int8_t* a, const int8_t* w, const float* a_scale,
                                    const float* w_scale, const void* bias, void* out, int32_t* acc_out,
                                    int64_t M, int64_t N, int64_t K, int out_dtype, void* stream);
/* kernel::scaled_matmul with ScaledMatmulParams::c, alpha = beta = 1 (kernels/param.h:852-866: "Result: alpha * (a @ b) +
 * beta * c"; the DCU backend of the reference drops c, alpha and beta -- kernels/dcu/scaled_matmul.cpp:112-113, 264-265 -- the
 * MLU backend honours them). out[m,n] = r16( y + c[m,n] ) with y = the 16-bit result of xllm_mi355_scaled_matmul: the GEMM's
 * rounding first, then a 16-bit add, so that out is bit-identical to scaled_matmul followed by the residual add of
 * fused_add_rms_norm (fused_layernorm with residual, ops_api.h:43). c [M,N] in the out dtype; out may alias c (in-place
 * residual update: the row-parallel o_proj / down_proj of a prefill chunk, whose norm pass then reads one tensor instead of two).
 * Served by the 8-phase kernel's dequant epilogue (prefill shapes); XM_ERR_UNSUPPORTED where a split-K or decode-shaped kernel
 * would take the problem -- the caller then runs scaled_matmul and adds in a second pass (xllm_amd/ops.py does). */
XM_API int xllm_mi355_scaled_matmul_add(const int8_t* a, const int8_t* w, const float* a_scale, const float* w_scale,
                                        const void* bias, const void* c, void* out, int64_t M, int64_t N, int64_t K,
                                        int out_dtype, void* stream);
/* out[i] = r16(a[i] + b[i]) over n 16-bit elements (16-byte aligned; out may alias a or b): the second pass of the addend form
 * above where no GEMM epilogue takes it (torch's `add` on 16-bit tensors: f32 add, one rounding). */
XM_API int xllm_mi355_add16(void* out, const void* a, const void* b, int64_t n, int dtype, void* stream);

/* ---- pre-packed int8 weights (decode-shaped GEMMs, M <= 512) -------------------------------------------------
 * A decode GEMM streams every weight byte once; the weight-stream kernel (xllm_amd/csrc/gemm_ws.hip) wants them in
 * MFMA-fragment order so that a fragment is 1 KiB contiguous in HBM and in the LDS:
 *   packed[((g * K/128 + kt) * 2 + ks) * 1024 + lane * 16 + j] = w[g*16 + (lane & 15)][kt*128 + ks*64 + (lane >> 4)*16 + j]
 * Packed once at weight-load time -- where the reference's loaders re-lay weights for a backend, e.g.
 * layers/common/linear.cpp:572-583 (fp8 scale handling at load) and the NPU / MLU weight formats. N % 16 == 0,
 * K % 128 == 0; `packed` has N*K bytes and must not alias `w`. */
XM_API int xllm_mi355_pack_weight_i8(const int8_t* w, int8_t* packed, int64_t N, int64_t K, void* stream);
/* kernel::scaled_matmul on packed weights. Same arithmetic and results as xllm_mi355_scaled_matmul (exact int32 sums,
 * the same dequant expression: bit-identical outputs). The scratch is EXPLICIT and caller-owned: `workspace` /
 * `ws_bytes` may be NULL / 0 (then K is never sliced); when given, K slices write their exact partial sums to
 * separate M*N int32 slabs with plain stores -- nothing has to be zero beforehand, nothing is left to clean up, and two
 * calls may share one buffer as long as they are ordered on a stream. XM_ERR_UNSUPPORTED outside the envelope
 * (M > 512, N % 16, K % 128, K < 512): callers fall back to xllm_mi355_scaled_matmul on the row-major weights. */
XM_API int xllm_mi355_scaled_matmul_packed(const int8_t* a, const int8_t* w_packed, const float* a_scale,
                                           const float* w_scale, const void* bias, void* out, int32_t* acc_out,
                                           int64_t M, int64_t N, int64_t K, int out_dtype, void* workspace,
                                           size_t ws_bytes, void* stream);
/* Planner hint for the packed-weight GEMMs launched AFTERWARDS BY THE CALLING THREAD (thread-local: a worker thread's hint never
 * touches another worker's launches; the analogue of an algo preference in a BLAS library -- the reference's hipBLASLt call picks
 * its algorithm by heuristic, kernels/dcu/scaled_matmul.cpp:242-262). ng = 16-column groups per wave (tile width), slices = K
 * slices, tile_rows = 128 | 256 for 128 < M <= 512; 0 = leave to the planner. Every combination gives bit-identical int8
 * results; the parity tests use the hint to cover the tile shapes the planner picks at other problem sizes. */
XM_API void xllm_mi355_gemm_plan_hint(int ng, int slices, int tile_rows);
/* xllm_mi355_scaled_matmul_add_rms_norm on packed weights with the explicit workspace (>= 4*M*N bytes required,
 * XM_ERR_WORKSPACE otherwise; more lets the planner slice K). Bit-identical to the unfused operator sequence. */
XM_API int xllm_mi355_scaled_matmul_add_rms_norm_packed(const int8_t* a, const int8_t* w_packed, const float* a_scale,
                                                        const float* w_scale, const void* bias, void* residual,
                                                        const void* norm_weight, float eps, void* out_norm,
                                                        int8_t* out_q, float* out_q_scale, int64_t M, int64_t N,
                                                        int64_t K, int dtype, void* workspace, size_t ws_bytes,
                                                        void* stream);
/* N1 fusion across the GEMM boundary ("quantized GEMM with fused dequant / RoPE"): the W8A8 qkv projection on packed weights,
 * its dequant epilogue (acc * a_scale[m] * w_scale[n] + bias -> 16 bit), RoPE of q and k and the KV write in TWO launches
 * (GEMM leaving exact int32 K-slice slabs in `workspace`, then one pass over each token's row): bit-identical to
 * scaled_matmul -> apply_rotary -> reshape_paged_cache (linear.cpp:481-507, qwen2_attention.cpp:150-176,
 * layers/dcu/attention.cpp:68-86). qkv [M, N] (N = (n_q_heads + 2 n_kv_heads) * head_size) receives the packed row with q and
 * k rotated; k / v go to the caches at slot_ids (a slot < 0 or past n_blocks is skipped). cos_sin_cache in the out dtype.
 * XM_ERR_UNSUPPORTED (no side effect) outside the envelope; XM_ERR_WORKSPACE below M * N * 4 bytes of scratch. */
XM_API int xllm_mi355_scaled_matmul_rope_cache_packed(const int8_t* a, const int8_t* w_packed, const float* a_scale,
                                                      const float* w_scale, const void* bias, void* qkv, int64_t M,
                                                      int64_t N, int64_t K, int dtype, const int64_t* positions,
                                                      const void* cos_sin_cache, const int32_t* slot_ids, void* k_cache,
                                                      void* v_cache, int64_t n_q_heads, int64_t n_kv_heads,
                                                      int64_t head_size, int64_t rot_dim, int64_t block_size,
                                                      int64_t n_blocks, int is_neox, void* workspace, size_t ws_bytes,
                                                      void* stream);

/* optional scratch for the int8 split-K path of scaled_matmul (>= M*N*4 bytes; the reference operator
 * has no workspace argument, so it is registered: ONE default buffer PER DEVICE -- the device is the one that owns the
 * registered pointer, so the reference's one-worker-thread-per-device processes (dist_manager.cpp:82-84) register one each --
 * NULL disables split-K on the current device). Registrations and look-ups are mutex-guarded (xllm_amd/csrc/workspace.hip). */
XM_API int xllm_mi355_set_gemm_workspace(void* workspace, size_t bytes);
/* Per-stream split-K workspace (same invariant: all-zero between calls). GEMMs launched on `stream` use it instead of
 * the global one, so that two micro-batches running concurrently on two streams (the reference's
 * enable_multi_stream_parallel / micro_batch_num, framework/config/parallel_config.h:83-85) never share partial sums.
 * ws == NULL unregisters the stream. At most 64 streams (all devices together). */
XM_API int xllm_mi355_set_gemm_workspace_for_stream(void* stream, void* ws, size_t bytes);

/* N1 fusion across the GEMM boundary (round 3): DenseMLP's gate_up_proj with its SiLU.mul fused into the GEMM epilogue, for the
 * W8A8 path (dense_mlp.cpp:97-116: gate_up_proj -> act_fn(gate) * up; linear.cpp:481-507: scaled_quantize -> scaled_matmul).
 * w [N = 2 I, K] int8 row-major, rows [0, I) = gate, [I, 2 I) = up (the reference's merged weight); w_packed = its
 * xllm_mi355_pack_weight_i8 copy (either may be NULL: packed serves M <= 512, row-major any M). The kernel computes the gate and
 * the up columns of the same act columns in one workgroup and writes act[m, i] = rT(rT(silu(g)) * u), g / u = the 16-bit
 * scaled_matmul outputs rT(acc * a_scale[m] * w_scale[n] + bias[n]), to act_out [M, I] -- bit-identical to scaled_matmul ->
 * act_and_mul -- and folds every row's |max| into row_amax [M] (f32; atomic max, MUST be zero on entry). Then
 * xllm_mi355_quantize_with_row_amax(act_out, row_amax, q, scale): q = rint(act * 127 / amax), scale = amax / 127
 * (scaled_quantize's expression) in one pass, and row_amax is zero again. The gate_up output (2 I columns of 16-bit values) is
 * never written. N % 256 == 0, K % 128 == 0; `workspace` is unused by this mode (K is never sliced) and may be NULL. */
XM_API int xllm_mi355_scaled_matmul_gate_up_act(const int8_t* a, const int8_t* w, const int8_t* w_packed, const float* a_scale,
                                                const float* w_scale, const void* bias, void* act_out, float* row_amax,
                                                int64_t M, int64_t N, int64_t K, int dtype, void* workspace, size_t ws_bytes,
                                                void* stream);
XM_API int xllm_mi355_quantize_with_row_amax(const void* act, float* row_amax, int8_t* out_q, float* out_scale,
                                             int64_t n_tokens, int64_t d, int dtype, void* stream);

/* ---- fp8 (OCP e4m3fn) ------------------------------------------------------------------------
 * kernel::static_scaled_fp8_quant (ops_api.h:160) -> kernels/cuda/fp8_quant.cu:115-155 */
XM_API int xllm_mi355_static_scaled_fp8_quant(uint8_t* out, const void* input, const float* scale,
                                              int64_t numel, int dtype, void* stream);
/* kernel::fp8_scaled_quantize (ops_api.h:151) -> kernels/cuda/fp8_scaled_quantize.cpp:20-50:
 * dynamic per-tensor scale = max(amax/448, 1e-12) written to scale_out[1] (device), then quant.
 * scale_in != NULL => static (scale_out ignored). */
XM_API int xllm_mi355_fp8_scaled_quantize(uint8_t* out, const void* input, const float* scale_in,
                                          float* scale_out, int64_t numel, int dtype, void* stream);
/* The dynamic form in TWO launches instead of four graph nodes (memset, amax with atomics, scale, quantise): one maximum per block
   into `workspace` (>= xllm_mi355_fp8_scaled_quantize_workspace_bytes(), caller-owned, private to the launching stream), folded
   by every block of the quantising launch. Same bits as xllm_mi355_fp8_scaled_quantize(out, input, NULL, scale_out, ...). */
XM_API size_t xllm_mi355_fp8_scaled_quantize_workspace_bytes(void);
XM_API int xllm_mi355_fp8_scaled_quantize_ws(uint8_t* out, const void* input, float* scale_out, int64_t numel, int dtype,
                                             void* workspace, size_t ws_bytes, void* stream);
/* kernel::fp8_scaled_matmul (ops_api.h:156) -> cutlass_scaled_mm (cutlass_w8a8/scaled_mm_entry.cu:55-116):
 * out = r16( a_scale * (w_scale * sum_fp32 a*w) + bias ); a [M,K] e4m3, w [N,K] e4m3 row-major
 * (the reference passes b.t() of it); a_scale numel 1 or M, w_scale numel 1 or N. K % 128 == 0. */
XM_API int xllm_mi355_fp8_scaled_matmul(const uint8_t* a, const uint8_t* w, const float* a_scale,
                                        int64_t a_scale_numel, const float* w_scale,
                                        int64_t w_scale_numel, const void* bias, void* out, int64_t M,
                                        int64_t N, int64_t K, int out_dtype, void* stream);

/* kernel::fp8_scaled_matmul on PRE-PACKED e4m3 weights (decode-shaped problems, M <= 512): the weight-stream kernel of
 * xllm_amd/csrc/gemm_ws.hip on v_mfma_f32_16x16x128_f8f6f4. The packed layout is byte-for-byte the int8 one
 * (xllm_mi355_pack_weight_i8 above; xllm_mi355_pack_weight_fp8 is the same permutation under the fp8 name): the two 16-byte
 * k-step fragments of a lane form the 32-byte MFMA operand. Same semantics as xllm_mi355_fp8_scaled_matmul
 * (linear.cpp:137-182, scaled_mm_entry.cu:55-116: per-tensor or per-token a_scale, per-tensor or per-channel w_scale, fp32
 * accumulation, out = r16(a_scale * (w_scale * acc) + bias)); the fp32 summation ORDER differs from the row-major kernel
 * (K tiles in order, K slices summed in slice order: deterministic), so results agree to fp32 rounding, not bit for bit.
 * `workspace` / `ws_bytes` explicit and caller-owned as for xllm_mi355_scaled_matmul_packed (fp32 slabs; NULL / 0 = K is
 * never sliced). XM_ERR_UNSUPPORTED outside the envelope (M > 512, N % 16, K % 128, K < 512). */
XM_API int xllm_mi355_pack_weight_fp8(const uint8_t* w, uint8_t* packed, int64_t N, int64_t K, void* stream);
XM_API int xllm_mi355_fp8_scaled_matmul_packed(const uint8_t* a, const uint8_t* w_packed, const float* a_scale,
                                               int64_t a_scale_numel, const float* w_scale, int64_t w_scale_numel,
                                               const void* bias, void* out, int64_t M, int64_t N, int64_t K,
                                               int out_dtype, void* workspace, size_t ws_bytes, void* stream);

/* kernel::matmul (F::linear, kernels/dcu/matmul.cpp:20-25) on PRE-PACKED 16-bit weights for decode-shaped problems (M <= 512):
 * the weight-stream kernel of gemm_ws.hip on v_mfma_f32_16x16x32_{bf16,f16}. The packed layout is the 8-bit kinds' byte
 * permutation applied to rows of 2 K bytes: packed[((g * KT + kt) * 2 + ks) * 1024 + lane * 16 + j] = byte
 * kt*128 + ks*64 + (lane >> 4)*16 + j of row g*16 + (lane & 15) (KT = 2 K / 128), i.e. a lane's 16 bytes are 8 consecutive
 * elements -- one MFMA operand. out = r16(sum_fp32 + bias); the fp32 summation order differs from xllm_mi355_matmul (K tiles in
 * order, K slices in slice order: deterministic), so results agree to fp32 rounding. N % 16 == 0, K % 64 == 0, K >= 256;
 * explicit caller-owned `workspace, ws_bytes` (fp32 slabs; NULL / 0: K is never sliced). XM_ERR_UNSUPPORTED outside the envelope. */
XM_API int xllm_mi355_pack_weight_16(const void* w, void* packed, int64_t N, int64_t K, void* stream);
XM_API int xllm_mi355_matmul_packed(const void* a, const void* w_packed, const void* bias, void* out, int64_t M, int64_t N,
                                    int64_t K, int dtype, void* workspace, size_t ws_bytes, void* stream);
/* lm_head + greedy sampling in one pass (round 4): out_idx[m] = argmax_n r16(a[m] . w[n] + bias[n]) -- the token
 * Sampler::greedy_sample picks from the logits of the (never quantised) lm_head (framework/sampling/sampler.cpp:160-168: argmax(-1);
 * layers/common/linear.cpp:512-520) -- without writing the [M, N] logits: every wave of the packed 16-bit GEMM reduces the 16-bit
 * ROUNDED values of its columns to a (max, first index) pair per row and a finishing launch reduces the pairs (argmax commutes
 * with the column tiling; torch.argmax order: NaN above every number, the first index among equals). out_val (optional, may be
 * NULL) receives the winning logit as float -- with a column-sharded lm_head the ranks then exchange [B] (value, index) pairs
 * instead of all-gathering [B, V / tp] logits (linear.cpp:712-714). The token ids equal xllm_mi355_greedy_argmax of
 * xllm_mi355_matmul_packed's output from the same launch plan bit for bit (the same fp32 sums, the same rounding).
 * workspace >= xllm_mi355_matmul_argmax_workspace_bytes(M, N) (= 8 * M * max(N / 16, 1024) bytes), XM_ERR_WORKSPACE otherwise; envelope of
 * xllm_mi355_matmul_packed (M <= 512, N % 16 == 0, K % 64 == 0), XM_ERR_UNSUPPORTED outside it. */
XM_API size_t xllm_mi355_matmul_argmax_workspace_bytes(int64_t M, int64_t N);
XM_API int xllm_mi355_matmul_argmax_packed(const void* a, const void* w_packed, const void* bias, int64_t* out_idx, float* out_val,
                                           int64_t M, int64_t N, int64_t K, int dtype, void* workspace, size_t ws_bytes,
                                           void* stream);
/* The gate_up linear of the dense MLP on packed 16-bit weights with SiLU * mul in its epilogue (dense_mlp.cpp:97-116 with an
 * unquantised layer: gate_up_proj -> kernel::act_and_mul, kernels/cuda/activation.cu:49-120): w_packed = pack_weight_16 of the
 * [N = 2 I, K] weight (gate rows first), act_out [M, I] = r16(r16(silu(gate)) * up) with gate / up = r16(sum_fp32 + bias) --
 * the expression of xllm_mi355_matmul_packed followed by xllm_mi355_act_and_mul on fp32 sums in this launch's own (fixed,
 * deterministic) K order, so the two agree to fp32 rounding like any two tile plans of the packed kernel (the fused form never
 * slices K). N % 32 == 0; otherwise the envelope of xllm_mi355_matmul_packed; XM_ERR_UNSUPPORTED outside it. */
XM_API int xllm_mi355_matmul_gate_up_act(const void* a, const void* w_packed, const void* bias, void* act_out, int64_t M,
                                         int64_t N, int64_t K, int dtype, void* workspace, size_t ws_bytes, void* stream);

/* kernel::matmul (ops_api.h:48) -> dcu::matmul == F::linear (kernels/dcu/matmul.cpp:20-25):
 * out = r16(a @ w^T + bias); a [M,K], w [N,K], dtype bf16/f16. K % 64 == 0.
 * Decode-shaped problems with a long K and few columns (M <= 512, N % 4 == 0) split K when a GEMM workspace is
 * registered (xllm_mi355_set_gemm_workspace): fp32 partial slabs [slices][M][N] in the workspace, reduced in slice order
 * (bit-reproducible) and re-zeroed, so the workspace stays zero at rest for the int8 split-K path. */
XM_API int xllm_mi355_matmul(const void* a, const void* w, const void* bias, void* out, int64_t M,
                             int64_t N, int64_t K, int dtype, void* stream);

/* ---- attention -----------------------------------------------------------------------------
 * AttentionImpl::forward modes (layers/dcu/flash_attention.cpp:167-288, arg set of
 * prefix_prefill_varlen_fwd / prefix_decode_varlen_fwd :45-94).
 *
 * prefill: packed q [Tq,nq,d] (token stride q_stride), k/v [Tk,nkv,d] (k_stride/v_stride),
 * cu_q/cu_k int32 [B+1], out [Tq, nq*d] contiguous; causal = bottom-right aligned;
 * window_left < 0 => unbounded. */
XM_API int xllm_mi355_prefill_attention(const void* q, const void* k, const void* v, void* out,
                                        const int32_t* cu_q, const int32_t* cu_k, int64_t batch,
                                        int64_t n_q_heads, int64_t n_kv_heads, int64_t head_dim,
                                        int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                        int64_t max_q_len, float scale, int causal,
                                        int64_t window_left, int dtype, void* stream);
/* paged (decode when max_q_len == 1, chunked prefill otherwise): q packed by cu_q [B+1];
 * caches [n_blocks, block_size, nkv, d]; kv_lens int32 [B]; block_table int32 [B, max_blocks]
 * (padding entries never read -- kv_lens decides; 0-padded or -1-padded both fine);
 * workspace: xllm_mi355_paged_attention_workspace_bytes(). */
XM_API size_t xllm_mi355_paged_attention_workspace_bytes(int64_t batch, int64_t n_q_heads,
                                                         int64_t head_dim_v, int64_t max_q_len,
                                                         int64_t total_q_tokens);
XM_API int xllm_mi355_paged_attention(const void* q, const void* k_cache, const void* v_cache,
                                      void* out, const int32_t* cu_q, const int32_t* kv_lens,
                                      const int32_t* block_table, int64_t max_blocks, int64_t batch,
                                      int64_t total_q_tokens, int64_t n_q_heads, int64_t n_kv_heads,
                                      int64_t head_dim, int64_t block_size, int64_t n_blocks,
                                      int64_t q_stride, int64_t max_q_len, int64_t max_kv_len,
                                      float scale, int causal, int64_t window_left, int dtype,
                                      void* workspace, size_t workspace_bytes, void* stream);
/* N1 fusion (decode): paged attention whose epilogue also emits the per-token int8 quantisation of its 16-bit
 * output -- the operand of the w8a8-dynamic o_proj (linear.cpp:481-507 would call scaled_quantize next).
 * out (16-bit, may be NULL), out_q [B, nq*d] int8, out_scale [B] f32; bit-identical to paged_attention followed by
 * scaled_quantize. Returns XM_ERR_UNSUPPORTED when the launch would need split-KV (small batches): the caller
 * then uses the two-operator sequence. No workspace. */
XM_API int xllm_mi355_paged_decode_attention_int8(const void* q, const void* k_cache, const void* v_cache,
                                                  void* out, int8_t* out_q, float* out_scale,
                                                  const int32_t* kv_lens, const int32_t* block_table,
                                                  int64_t max_blocks, int64_t batch, int64_t n_q_heads,
                                                  int64_t n_kv_heads, int64_t head_dim, int64_t block_size,
                                                  int64_t q_stride, int64_t max_kv_len, float scale,
                                                  int64_t window_left, int dtype, void* stream);
/* The same fusion for every launch plan: with a workspace (xllm_mi355_paged_attention_workspace_bytes) the plans whose
 * workgroups do not hold a whole token (grid-level split-KV at small batches, fewer than all kv heads per workgroup) leave
 * their (o, m, l) partials there and ONE finishing launch merges and quantises them -- still bit-identical to paged_attention
 * followed by scaled_quantize. workspace == NULL: the behaviour of the entry point above. */
XM_API int xllm_mi355_paged_decode_attention_int8_ws(const void* q, const void* k_cache, const void* v_cache,
                                                     void* out, int8_t* out_q, float* out_scale,
                                                     const int32_t* kv_lens, const int32_t* block_table,
                                                     int64_t max_blocks, int64_t batch, int64_t n_q_heads,
                                                     int64_t n_kv_heads, int64_t head_dim, int64_t block_size,
                                                     int64_t q_stride, int64_t max_kv_len, float scale,
                                                     int64_t window_left, int dtype, void* workspace,
                                                     size_t workspace_bytes, void* stream);
/* N1 fusion: RoPE (apply_rotary) + KV write (reshape_paged_cache) in one pass over the packed qkv row:
 * q and k are rotated in place, the rotated k and v are scattered to the caches at slot_ids. Bit-identical to the
 * two-operator sequence (cf. the MLA fused_mla_kv of param.h:1105-1178). */
XM_API int xllm_mi355_rotary_embedding_and_cache(const int64_t* positions, void* q, void* k, const void* v,
                                                 const void* cos_sin_cache, const int32_t* slot_ids, void* k_cache,
                                                 void* v_cache, int64_t n_tokens, int64_t n_q_heads,
                                                 int64_t n_kv_heads, int64_t head_size, int64_t rot_dim,
                                                 int64_t q_stride, int64_t k_stride, int64_t v_stride,
                                                 int64_t block_size, int64_t n_blocks, int is_neox, int dtype,
                                                 void* stream);

/* ---- logits processors in front of the sampler (N3; framework/sampling/logits_utils.cpp:24-155, Sampler::forward
 * sampler.cpp:33-116). logits [batch, vocab] with `row_stride` elements between rows, dtype XM_F32 / XM_BF16 / XM_F16, modified
 * IN PLACE like the reference's functions. None of them reads a device value on the host: graph-capturable.
 *
 * xllm_mi355_apply_penalties: apply_frequency_presence_penalties (:24-36) when frequency_penalties / presence_penalties [batch]
 * are given (both or neither), then apply_repetition_penalties (:38-52) when repetition_penalties [batch] is given.
 * unique_token_ids int64 [batch, n_unique] / unique_token_counts int32 [batch, n_unique] = the padded per-sequence tables of
 * SamplingParameters (sampling_params.cpp:127-134). Every (row, u) is computed from the row as it was before the call (the
 * reference's gather-then-scatter), so repeated padding ids are harmless; ids outside [0, vocab) are skipped.
 * workspace >= 4 * batch * n_unique bytes (xllm_mi355_apply_penalties_workspace_bytes). */
XM_API size_t xllm_mi355_apply_penalties_workspace_bytes(int64_t batch, int64_t n_unique);
XM_API int xllm_mi355_apply_penalties(void* logits, int64_t batch, int64_t vocab, int64_t row_stride, int dtype,
                                      const int64_t* unique_token_ids, const int32_t* unique_token_counts, int64_t n_unique,
                                      const float* frequency_penalties, const float* presence_penalties,
                                      const float* repetition_penalties, void* workspace, size_t workspace_bytes, void* stream);
/* apply_temperatures (:54-64): logits[b, :] /= (temperatures[b] == 0 ? 1 : temperatures[b]). */
XM_API int xllm_mi355_apply_temperatures(void* logits, int64_t batch, int64_t vocab, int64_t row_stride, int dtype,
                                         const float* temperatures, void* stream);
/* apply_top_k_top_p (:92-155): temperatures (optional, applied first), then top-k and / or top-p masking with -inf; top_k int64
 * [batch], top_p float [batch], either may be NULL.
 *   one of them given ("else" branch, :121-153 -- what a CUDA / DCU build of the reference runs): top_k <= 0 disables the row's
 *     top-k; top-p masks sorted rank i when cumsum(softmax(sorted))[i] - prob[i] > top_p (rank 0 always survives);
 *   both given: apply_top_k_top_p_torch_impl (:66-90): k = clamp(top_k, 1, vocab), top-p masks rank i > 0 when cumsum[i] > top_p.
 *     (On a CUDA / DCU build the reference's both-given case falls through an NPU / MLU-only #if, :107-120, and applies nothing;
 *     this backend applies the torch_impl rule the file ships for it.)
 * No sort: radix selection on an order-preserving key finds the k-th logit, the same descent over fixed-point probability mass
 * finds where the cumulative probability crosses p; ties at a boundary rank by column index (a stable descending sort). The
 * result equals the reference's wherever p is not within fp32 summation error of a cumulative-probability step. */
XM_API int xllm_mi355_apply_top_k_top_p(void* logits, int64_t batch, int64_t vocab, int64_t row_stride, int dtype,
                                        const float* temperatures, const int64_t* top_k, const float* top_p, void* stream);

/* flash_mla::dense_decode (kernels/dcu/flash_mla_adapter.h:40-50): q [B, H, 576] = [q_nope*W_kc || q_pe],
 * k_cache [n_blocks, block_size, 1, 576]; out [B, H, head_size_v] = softmax(scale q k^T) k[:, :512]. */
XM_API int xllm_mi355_mla_decode(const void* q, const void* k_cache, void* out,
                                 const int32_t* seqlens_k, const int32_t* block_table,
                                 int64_t max_blocks, int64_t batch, int64_t n_heads, int64_t head_dim,
                                 int64_t head_dim_v, int64_t block_size, int64_t n_blocks,
                                 int64_t max_kv_len, float scale, int dtype, void* workspace,
                                 size_t workspace_bytes, void* stream);

/* MLA prefill / chunked prefill: DeepseekV2AttentionImpl::prefill_sdpa (layers/dcu/deepseek_v2_attention.cpp:212-262;
 * the reference loops over sequences on the HOST and calls torch SDPA per sequence) in the same absorbed form as the
 * decode: q [T, H, 576] = [q_nope*W_kc || q_pe] of a ragged batch (cu_q [B+1]), keys = the paged latent cache rows
 * of each sequence (kv_lens [B], block_table [B, max_blocks]) -- call xllm_mi355_reshape_paged_cache with v = NULL
 * (store_latent_cache, :170-178) first -- out [T, H, 512].  causal != 0: query i of a sequence sees its first
 * kv_len - (q_len - 1 - i) keys (bottom-right alignment; identical to torch's is_causal when q_len == kv_len).
 * block_size % 64 == 0 and >= 128 workgroups of (4 query tokens x 16 heads): the four tokens of a workgroup share every
 * DMA'd KV tile (no workspace beyond the index arrays); otherwise every query token streams its keys like a decode
 * entry with split-KV.  workspace >= 8*T bytes (+ split-KV partials, see _mla_decode). */
XM_API int xllm_mi355_mla_prefill(const void* q, const void* k_cache, void* out, const int32_t* cu_q,
                                  const int32_t* kv_lens, const int32_t* block_table, int64_t max_blocks,
                                  int64_t batch, int64_t total_q_tokens, int64_t n_heads, int64_t head_dim,
                                  int64_t head_dim_v, int64_t block_size, int64_t n_blocks, int64_t max_kv_len,
                                  float scale, int causal, int dtype, void* workspace, size_t workspace_bytes,
                                  void* stream);

/* N4: cuda::moe_fused_topk (kernels/cuda/moe/moe_fused_topk.cu:31-61; DCU falls back to it from moe_active_topk,
 * kernels/dcu/topk_gate.cpp:127-146): gating [T, E] (f32 / bf16 / f16) -> topk_weights [T, topk] f32, topk_ids
 * [T, topk] int32.  scoring 0 = softmax, 1 = sigmoid (+ optional fp32 correction_bias [E]: added for the selection,
 * removed from the returned weight); ties go to the lower expert index; renormalize divides by the selected sum.
 * E <= 512. */
XM_API int xllm_mi355_moe_fused_topk(const void* gating, int dtype, int64_t n_tokens, int64_t n_experts,
                                     int64_t topk, int renormalize, const float* correction_bias, int scoring,
                                     float* topk_weights, int32_t* topk_ids, void* stream);

/* dcu::moe_grouped_topk (kernels/dcu/topk_gate.cpp:59-125; what moe_active_topk :127-146 calls when num_expert_group > 1)
 * -> aiter::native::grouped_topk / biased_grouped_topk (external library, not in the reference tree; the published
 * DeepSeek-V2 / V3 gate is restated): score s = softmax | sigmoid; choice c = s (+ correction_bias, sigmoid only); group
 * value = max c (no bias) | sum of the two largest c (bias); the topk_group best groups stay; topk experts by c among
 * them (ties: lower index); weight = s (unbiased), / selected sum when renormalize, * routed_scaling_factor.
 * Argument checks follow :67-98 (num_expert_group > 1, 0 < topk_group <= num_expert_group, bias needs sigmoid).
 * E <= 512, E % num_expert_group == 0, num_expert_group <= 64, topk <= min(64, topk_group * E / num_expert_group). */
XM_API int xllm_mi355_moe_grouped_topk(const void* gating, int dtype, int64_t n_tokens, int64_t n_experts,
                                       int64_t topk, int64_t num_expert_group, int64_t topk_group, int renormalize,
                                       const float* correction_bias, int scoring, float routed_scaling_factor,
                                       float* topk_weights, int32_t* topk_ids, void* stream);

/* ---- N3: sampler kernels of the decode step ---------------------------------------------------
 * dcu::random_sample (kernels/dcu/random_sample.hip:88-270; ops_api.h random_sample): probs [batch, vocab] fp32
 * -> one token id per row by CDF inversion: the first index with p > 0 whose inclusive prefix sum exceeds u; rows whose
 * total never exceeds u return their last index with p > 0 (0 if none).  u: `uniform` [batch] when non-null, else
 * the first hiprand_uniform() of hiprand_init(philox_seed, subsequence = row, philox_offset) (Philox4x32-10, the
 * reference's generator; xllm_mi355_philox_uniform exposes that stream).  Prefix sums are fp32 in a fixed order, so
 * the result is deterministic; it can differ from another summation order only when u is within fp32 rounding of a
 * CDF step. */
XM_API int xllm_mi355_random_sample(const float* probs, int32_t* out, int64_t batch, int64_t vocab,
                                    const float* uniform, uint64_t philox_seed, uint64_t philox_offset,
                                    void* stream);
XM_API int xllm_mi355_philox_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);
/* Sampler::forward's tail in ONE launch (framework/sampling/sampler.cpp:118-137: probs = softmax(sample_logits, -1, fp32);
 * samples = random_sample(probs) | greedy_sample(probs) | where(do_sample, random, greedy)): logits [batch, vocab] (XM_F32 /
 * XM_BF16 / XM_F16, row pitch `row_stride` elements; already temperature-scaled and top-k / top-p masked, e.g. by
 * xllm_mi355_apply_top_k_top_p) -> out[batch] int32.  The [batch, vocab] fp32 probabilities are never materialised.
 * Rows with do_sample[b] == 0 (do_sample may be null = sample every row) take the FIRST column of the row maximum
 * (= argmax of the probabilities).  u and the tie / fallback rules: as xllm_mi355_random_sample; prefix sums run over
 * exp(x - max) in fp32 against u * sum, so the index can differ from softmax -> random_sample only when u lies within fp32
 * rounding of a CDF step.  -inf and NaN logits carry no mass; a row without a finite maximum returns 0 (+inf: its first column). */
XM_API int xllm_mi355_softmax_random_sample(const void* logits, int32_t* out, int64_t batch, int64_t vocab, int64_t row_stride,
                                            int dtype, const float* uniform, uint64_t philox_seed, uint64_t philox_offset,
                                            const uint8_t* do_sample, void* stream);
/* The greedy branch of the sampler: Sampler::greedy_sample = argmax over the last dim (framework/sampling/sampler.cpp:160-168).
 * logits [batch, vocab] (XM_F32 / XM_BF16 / XM_F16) -> out[batch] int64 = the FIRST index of each row's maximum; a NaN counts
 * as larger than every number (torch.argmax). One workgroup per row, one pass, 16-byte loads. */
XM_API int xllm_mi355_greedy_argmax(const void* logits, int64_t* out, int64_t batch, int64_t vocab, int dtype, void* stream);

/* dcu::rejection_sample (kernels/dcu/rejection_sample.hip:33-215): speculative-decoding verification.  For sequence
 * s with n = num_draft_tokens[s] drafts ending at cu_num_draft_tokens[s] (inclusive prefix sums): output slots
 * [start + s, start + s + n] are set to -1, draft i is accepted while uniform_rand[row] < target[row, tok] /
 * draft[row, tok]; the first rejected position gets argmax_t max(target - draft, 0)[t] / max(uniform_probs[row, t],
 * FLT_MIN) (lowest index on ties) and the sequence stops; if all drafts pass, slot n gets bonus_token_ids[s].
 * output has batch + total_drafts entries.  Integer output, bit-exact. */
XM_API int xllm_mi355_rejection_sample(const int32_t* draft_token_ids, const int32_t* num_draft_tokens,
                                       const int32_t* cu_num_draft_tokens, const float* draft_probs,
                                       const float* target_probs, const int32_t* bonus_token_ids,
                                       const float* uniform_rand, const float* uniform_probs, int64_t batch,
                                       int64_t vocab, int32_t* output, void* stream);

/* ---- MoE -----------------------------------------------------------------------------------
 * kernel::moe_gen_idx (ops_api.h:73) -> cuda::moe_compute_index (kernels/cuda/moe/moe_compute_index.cu:111-160)
 * expert_id [T,topk] int32 -> src_dst[T*topk], dst_src[T*topk], expert_sizes[E]; DETERMINISTIC
 * (stable by expanded row index) unlike the reference's atomics order. workspace >= 4*(E+1)*... see .hip */
/* scratch for moe_compute_index: >= 4 * ceil(T*topk/1024) * n_experts bytes, registered once; group_gemm keeps the
 * tile table of its 256x256 kernel (16 * (rows / 256 + n_experts) bytes) in the tail of the same scratch and falls back
 * to the 128x128 kernel when the scratch is absent or too small */
XM_API int xllm_mi355_set_moe_workspace(void* workspace, size_t bytes);
/* the same scratch for launches on one particular stream (two MoE layers on two streams of one device); one default per device
 * otherwise, keyed by the device that owns the registered pointer. ws == NULL unregisters the stream. */
XM_API int xllm_mi355_set_moe_workspace_for_stream(void* stream, void* ws, size_t bytes);
XM_API int xllm_mi355_moe_compute_index(const int32_t* expert_id, int64_t n_tokens, int64_t topk,
                                        int64_t n_experts, int32_t* src_dst, int32_t* dst_src,
                                        int32_t* expert_sizes, void* stream);
/* kernel::moe_combine_result (ops_api.h:77) -> cuda::moe_combine_result (moe/moe_combine.cu:64-...) */
XM_API int xllm_mi355_moe_combine(void* out, const void* gemm2, const float* weights, int64_t n_tokens,
                                  int64_t topk, int64_t hidden, int dtype, void* stream);
/* fusion of the reference's `index_copy_` + moe_combine_result (layers/dcu/fused_moe.cpp:296-303): the rows of the second
 * grouped GEMM stay in expert order and are gathered through src_dst (from moe_compute_index) while they are combined:
 * out[t] = sum_k weights[t,k] * gemm2_sorted[src_dst[t*topk+k]] -- the same fp32 sum in the same order as the two
 * operators (bit-identical). 16-bit dtypes, topk <= 16, hidden % 8 == 0. */
XM_API int xllm_mi355_moe_combine_sorted(void* out, const void* gemm2_sorted, const int32_t* src_dst,
                                         const float* weights, int64_t n_tokens, int64_t topk, int64_t hidden,
                                         int dtype, void* stream);
/* Expert-parallel form of the same fusion (FusedMoEImpl::forward_experts on an EP rank, layers/dcu/fused_moe.cpp:236-303:
 * only the rank's experts are computed, gemm2_full is ZERO elsewhere, the EP all-reduce adds the ranks): the caller sorts
 * with expert ids rotated so that its own experts come first ((id - start_expert_id) mod E), runs the grouped GEMMs over
 * local_expert_sizes = expert_sizes[0 .. n_local_experts), and this combine skips every sorted row at or past
 * sum(local_expert_sizes) -- the reference's zero rows -- without a host read of the sizes (graph-capturable). */
XM_API int xllm_mi355_moe_combine_sorted_local(void* out, const void* gemm2_sorted, const int32_t* src_dst,
                                               const float* weights, const int32_t* local_expert_sizes,
                                               int64_t n_local_experts, int64_t n_tokens, int64_t topk, int64_t hidden,
                                               int dtype, void* stream);
/* Per-head batched GEMM of MLA's weight absorption (round 5): out[t, h, n] = r16(sum_k x[t, h, k] * w[h, n, k]), fp32 accumulation.
 * Replaces the two torch::bmm calls (rocBLAS) + transposes of DeepseekV2AttentionImpl (layers/dcu/deepseek_v2_attention.cpp:
 * 310-311: q_nope x W_kc, K = 128, N = 512; :180-187 project_output: attn x W_vc, K = 512, N = 128). x and out are the reference's
 * token-major tensors, addressed through their (token, head) strides in ELEMENTS (x may be a slice of the packed q tensor); w holds
 * each head's matrix with K contiguous per output column, (head, column) strides in elements -- for W_vc that is kv_b_proj's own
 * weight slice [h, v, kv_lora] before the reference transposes it for bmm (:336-338), W_kc is transposed once at load time.
 * K % 32 == 0; strides of x and w multiples of 8 elements, of out multiples of 4; XM_ERR_UNSUPPORTED otherwise. bf16 / f16. */
XM_API int xllm_mi355_bmm_heads(const void* x, int64_t x_stride_t, int64_t x_stride_h, const void* w, int64_t w_stride_h,
                                int64_t w_stride_n, void* out, int64_t out_stride_t, int64_t out_stride_h, int64_t n_tokens,
                                int64_t n_heads, int64_t N, int64_t K, int dtype, void* stream);
/* kernel::group_gemm (ops_api.h:57) -> dcu::group_gemm (kernels/dcu/group_gemm.cpp:25-74):
 * rows of `a` sorted by expert; out[off_e:off_e+M_e] = a[...] @ w[e]^T, w [E,N,K]; token_count is a
 * DEVICE int32 [E] (no host read). max_rows = a's row count. */
XM_API int xllm_mi355_group_gemm(const void* a, const void* w, const int32_t* token_count, void* out,
                                 int64_t max_rows, int64_t n_experts, int64_t N, int64_t K, int dtype,
                                 void* stream);
/* W8A8 grouped GEMM (GroupGemmParams with a_scale / b_scale, kernels/param.h:374-394; the reference implements it on
 * MLU / NPU only, its DCU path is 16-bit): out[off_e + m, n] = r16( i32(sum_k a[.., k] w[e, n, k]) * a_scale[..] * w_scale[e, n] ).
 * a int8 rows sorted by expert ([max_rows, K], a_scale [max_rows]) -- or, with row_index != NULL, the UN-expanded
 * activations [a_rows, K] / a_scale [a_rows] gathered as row row_index[r] / index_div (the expand of
 * layers/dcu/fused_moe.cpp:195-197 fused in, so each token is quantised once, not topk times). w [E, N, K] int8, w_scale
 * [E, N] float32 (16-byte aligned), token_count DEVICE int32 [E]. K % 128 == 0, N % 8 == 0. Needs the MoE scratch
 * (xllm_mi355_set_moe_workspace) for the device-built tile table: XM_ERR_WORKSPACE otherwise. int32 accumulators exact. */
XM_API int xllm_mi355_group_gemm_w8a8(const int8_t* a, int64_t a_rows, const float* a_scale, const int32_t* row_index,
                                      int64_t index_div, const int8_t* w, const float* w_scale,
                                      const int32_t* token_count, void* out, int64_t max_rows, int64_t n_experts,
                                      int64_t N, int64_t K, int out_dtype, void* stream);
/* group_gemm with the reference's expand step fused in (layers/dcu/fused_moe.cpp:195-197: `index_select(hidden,
 * dst_src / topk)` then group_gemm): sorted row r of the grouped problem is row row_index[r] / index_div of `a`
 * ([a_rows, K], the un-expanded activations); the expanded copy is never materialised. Runs on the 256x256 kernel only:
 * XM_ERR_UNSUPPORTED (caller expands and calls group_gemm) when that kernel cannot take the shape or no MoE scratch is
 * registered. Results are bit-identical to index_select + group_gemm. */
XM_API int xllm_mi355_group_gemm_gather(const void* a, int64_t a_rows, const int32_t* row_index, int64_t index_div,
                                        const void* w, const int32_t* token_count, void* out, int64_t max_rows,
                                        int64_t n_experts, int64_t N, int64_t K, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * HOST side of the path: per-step indexing data, built on the CPU as the reference's BatchInputBuilder does.
 * No GPU is touched by these two entry points.
 *
 * xllm_mi355_host_cache_slots = KVCacheState::cache_slots (framework/request/sequence_kv_state.cpp:86-104):
 *   slots[i - pos_start] = block_ids[i / block_size] * block_size + i % block_size for i in [pos_start, pos_end).
 * xllm_mi355_host_build_batch = BatchInputBuilder::setup_kv_cache_info + finalisation on the CUDA / DCU branch
 *   (framework/batch/batch_input_builder.cpp:525-537, 739-830, 900-938) and the length bookkeeping of
 *   build_attention_metadata (layers/common/attention_metadata_builder.cpp:45-244): for sequence b with
 *   n_kv_cache_tokens[b] tokens already cached and seq_lens[b] tokens after this step (q = the difference) and the
 *   blocks block_ids[block_indptr[b] .. block_indptr[b+1]):
 *     new_cache_slots / positions [sum q], paged_kv_indptr [B+1], paged_kv_indices [sum blocks],
 *     paged_kv_last_page_len [B] (= len % block_size, or block_size), block_tables [B, max_blocks] padded with 0,
 *     q_cu_seq_lens / kv_cu_seq_lens [B+1] (cumulative, leading 0), q_seq_lens / kv_seq_lens [B] (their differences),
 *     q_max_seq_len, kv_max_seq_len, total_kv_len.
 *   Output arrays are caller-owned host buffers with capacities cap_*; a null array is skipped. The counts
 *   (n_tokens, n_indices, max_blocks) are always written; XM_ERR_WORKSPACE = a capacity is too small (size the buffers from
 *   the counts and call again). XM_ERR_INVALID where the reference CHECK-fails (no blocks, sequence longer than its pages). */
typedef struct {
  int64_t cap_tokens, cap_indices, cap_sequences, cap_block_table; /* in: capacities (elements) of the arrays below */
  int32_t* new_cache_slots;
  int32_t* positions;
  int32_t* paged_kv_indptr;
  int32_t* paged_kv_indices;
  int32_t* paged_kv_last_page_len;
  int32_t* block_tables;
  int32_t* q_cu_seq_lens;
  int32_t* kv_cu_seq_lens;
  int32_t* q_seq_lens;
  int32_t* kv_seq_lens;
  int32_t num_sequences, q_max_seq_len, kv_max_seq_len; /* out */
  int64_t n_tokens, n_indices, max_blocks, total_kv_len; /* out */
} xllm_mi355_host_batch_t;
XM_API int xllm_mi355_host_cache_slots(const int32_t* block_ids, int64_t n_blocks, int64_t block_size, int64_t pos_start,
                                       int64_t pos_end, int32_t* slots);
XM_API int xllm_mi355_host_build_batch(const int32_t* n_kv_cache_tokens, const int32_t* seq_lens,
                                       const int32_t* block_indptr, const int32_t* block_ids, int64_t num_sequences,
                                       int64_t block_size, xllm_mi355_host_batch_t* out);

/* HOST side: the contiguous input buffer of one step (ForwardInputBufferPlan, runtime/forward_params.h:87-175): every host
 * tensor of a ForwardInput in ONE byte buffer -- entries in insertion order, each at the next multiple of `alignment` bytes
 * (the reference uses 16), tails zero-filled -- so that the step reaches the device in one H2D copy and the device tensors are
 * views of one device buffer. plan: fills offset / aligned_bytes of every entry and *total_bytes. pack: copies the entries
 * into `buffer` (XM_ERR_WORKSPACE when it is too small). No GPU is touched. */
typedef struct {
  const void* data;        /* in: the host tensor's bytes (may be NULL when bytes == 0) */
  uint64_t bytes;          /* in */
  uint64_t offset;         /* out (plan) / in (pack) */
  uint64_t aligned_bytes;  /* out (plan) / in (pack) */
} xllm_mi355_host_buffer_entry_t;
XM_API int xllm_mi355_host_plan_input_buffer(xllm_mi355_host_buffer_entry_t* entries, int64_t n_entries,
                                             uint64_t alignment, uint64_t* total_bytes);
XM_API int xllm_mi355_host_pack_input_buffer(const xllm_mi355_host_buffer_entry_t* entries, int64_t n_entries,
                                             void* buffer, uint64_t buffer_bytes);

/* ---------------------------------------------------------------------------------------------------------------
 * One-shot SUM all-reduce over peer-mapped buffers (xGMI), for the small tensor-parallel messages of a decode step.
 * Replaces, for messages <= max_message_bytes, parallel_state::reduce -> ProcessGroup::allreduce -> ProcessGroupNCCL
 * (framework/parallel_state/parallel_state.cpp:183-192, process_group.cpp:98-108). Opt-in; RCCL stays the default.
 *
 * Set-up, once per rank (one process per GPU): allocate the shared buffer with xllm_mi355_ipc_alloc
 * (xllm_mi355_oneshot_allreduce_buffer_bytes(max_message_bytes) bytes, zeroed; *kind in: the first memory kind to try,
 * out: the kind obtained -- 0 fine-grained, 1 uncached, 2 plain hipMalloc), export it (xllm_mi355_ipc_get_handle: XLLM_MI355_IPC_HANDLE_BYTES opaque bytes that
 * travel to the peers over any host channel), open every peer's handle (xllm_mi355_ipc_open_handle), keep
 * peer_buffers[world] with peer_buffers[rank] = the rank's own buffer. epoch_state: 2 zero-initialised uint32 in the
 * rank's own device memory; status: 1 zero-initialised int (set to 1 if a wait exceeded timeout_s -- the result of that
 * launch is undefined; a peer died or did not issue the same sequence of all-reduces).
 *
 * xllm_mi355_oneshot_allreduce: inout[count] (XM_F32 / XM_BF16 / XM_F16, 16-byte aligned, count * size % 16 == 0) becomes
 * the sum over ranks, accumulated in fp32 in rank order -- bit-identical on every rank. Every rank must call it with the
 * same count, in the same order. A plain kernel on `stream`: no host synchronisation, HIP-graph capturable.
 * XM_ERR_WORKSPACE: message larger than max_message_bytes (fall back to RCCL). */
#define XLLM_MI355_IPC_HANDLE_BYTES 64
XM_API size_t xllm_mi355_oneshot_allreduce_buffer_bytes(size_t max_message_bytes);
XM_API int xllm_mi355_ipc_alloc(size_t bytes, void** ptr, int* kind);
XM_API int xllm_mi355_ipc_free(void* ptr);
XM_API int xllm_mi355_ipc_get_handle(void* dev_ptr, void* handle64);
XM_API int xllm_mi355_ipc_open_handle(const void* handle64, void** ptr);
XM_API int xllm_mi355_ipc_close_handle(void* ptr);
XM_API int xllm_mi355_oneshot_allreduce(void* inout, int64_t count, int dtype, void* const* peer_buffers, int rank,
                                        int world, size_t max_message_bytes, uint32_t* epoch_state, int* status,
                                        double timeout_s, void* stream);

/* The one-shot all-reduce FUSED with what follows it in a tensor-parallel half-layer: row-parallel linear -> SUM all-reduce ->
 * fused_add_rms_norm (-> scaled_quantize of the next W8A8 linear) (linear.cpp:1518-1520, qwen2_decoder_layer.cpp:66-110,
 * kernels/cuda/norm.cu:229-425). partial [M, H] = this rank's 16-bit output of the row-parallel linear (read only);
 * y = rT(sum over ranks, fp32, rank order); residual [M, H] <- rT(y + residual) in place; then EITHER out_norm [M, H] = the
 * 16-bit RMSNorm OR out_q [M, H] int8 + out_q_scale [M] = its per-token quantisation. out_sum (optional, may be NULL) receives y.
 * Bit-identical to xllm_mi355_oneshot_allreduce followed by xllm_mi355_fused_add_rms_norm / xllm_mi355_rms_norm_dynamic_int8_quant
 * with a residual. Same buffers, epoch and status as xllm_mi355_oneshot_allreduce (the two may be mixed freely on one stream);
 * every rank calls it with the same M, H. H % 8 == 0, H <= 16384, M * H * 2 <= max_message_bytes (XM_ERR_WORKSPACE otherwise).
 * grid_limit: the most blocks a launch may use (whole rows per block), 0 = 64, at most 256. Every block waits for its peers'
 * blocks, so all ranks' grids must be co-resident: one rank per GPU may pass 256 (a 256-row decode message then has one row per
 * block -- one dependent exchange + norm chain instead of four), ranks sharing a GPU must keep 64. The same value on every rank. */
XM_API int xllm_mi355_oneshot_allreduce_add_rms_norm(const void* partial, void* residual, const void* norm_weight, float eps,
                                                     void* out_norm, int8_t* out_q, float* out_q_scale, void* out_sum,
                                                     int64_t M, int64_t H, int dtype, void* const* peer_buffers, int rank,
                                                     int world, size_t max_message_bytes, uint32_t* epoch_state, int* status,
                                                     double timeout_s, int grid_limit, void* stream);

/* The same with the row-parallel W8A8 linear in front of it (linear.cpp:481-507 + 1518-1520): the packed-weight GEMM of
 * xllm_mi355_scaled_matmul_packed leaves this rank's exact int32 K-slice sums in `workspace` and step 1 of the one-shot kernel
 * dequantises them -- rT(sum * a_scale[m] * w_scale[n] + bias[n]), the GEMM's own epilogue expression -- on their way into the
 * exchange slot, so a TP half-layer is GEMM -> ONE kernel with no dequant pass in between. a [M, K] int8, w_packed =
 * pack_weight_i8 of this rank's [N, K] shard, bias (rank 0 only, or NULL) in the output dtype; the rest as above with H = N.
 * Bit-identical to xllm_mi355_scaled_matmul_packed -> xllm_mi355_oneshot_allreduce_add_rms_norm. M <= 512 and the envelope of
 * the packed GEMM (XM_ERR_UNSUPPORTED otherwise, nothing written); workspace >= 4 M N bytes (XM_ERR_WORKSPACE). */
XM_API int xllm_mi355_scaled_matmul_oneshot_allreduce_add_rms_norm(
    const int8_t* a, const int8_t* w_packed, const float* a_scale, const float* w_scale, const void* bias, void* residual,
    const void* norm_weight, float eps, void* out_norm, int8_t* out_q, float* out_q_scale, void* out_sum, int64_t M, int64_t N,
    int64_t K, int dtype, void* workspace, size_t ws_bytes, void* const* peer_buffers, int rank, int world,
    size_t max_message_bytes, uint32_t* epoch_state, int* status, double timeout_s, int grid_limit, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XLLM_MI355_H_ */
