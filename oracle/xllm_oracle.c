/*
 * xllm_oracle.c -- CPU restatement of the xLLM decode/prefill hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (xllm_amd/, include/,
 * shim/) may include, link or call this file.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() use it, as the checker.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic -- including the 16-bit cast points -- it restates.
 * The reference has no CPU backend and cannot be compiled here,
 * so this is a "port"; it is pinned against
 *   - the golden vectors of tests/core/layers/mlu/qwen2_attention_test.cpp:254-328
 *     (prefill B=2,S=128 and paged decode B=4,S=257, seeded_tensor inputs),
 *     -- the decode vector to the last printed digit once P is rounded to bf16 before PV (p_round), the reference's own
 *     eager spec (layers/cuda/flashinfer_attention.cpp:84-90),
 *   - the statistics / values hard-coded in tests/core/layers/mlu/{moe_gate,dense_mlp,fused_moe}_test.cpp (grouped gate:
 *     min / max / sum of weights and ids for three seeded cases; W8A8 MLP = 1105920.0; W8A8 MoE layer = 992.0): reproduced
 *     exactly, tests/test_reference_fixtures.py -- these pin the per-token int8 quantiser, the int32 GEMM + scale
 *     epilogue, SiLU * mul's cast points, the grouped gate and the weighted combine,
 *   - the in-test CPU references of tests/core/kernels/dcu/ (the _test.cpp files) (torch CPU
 *     ops with the reference's tolerances), see tests/test_oracle_*.py,
 *   - end to end: oracle/model.py against the HuggingFace Qwen2 implementation (tests/test_oracle_model.py).
 *
 * Plain C99 + OpenMP.  dtype codes: 0 = f32, 1 = bf16, 2 = f16.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* scalar conversions                                                        */
/* ------------------------------------------------------------------------- */
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float bf16_to_f32(uint16_t h) { return u2f((uint32_t)h << 16); }
static inline uint16_t f32_to_bf16(float f) { /* round-to-nearest-even, NaN kept quiet */
  uint32_t u = f2u(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

static inline float f16_to_f32(uint16_t h) {
  uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff;
  if (e == 0) {
    if (m == 0) return u2f(s);
    float v = (float)m * (1.0f / 16777216.0f); /* 2^-24 */
    return (s ? -v : v);
  }
  if (e == 31) return u2f(s | 0x7f800000u | (m << 13));
  return u2f(s | ((e + 112) << 23) | (m << 13));
}
static inline uint16_t f32_to_f16(float f) { /* RNE */
  uint32_t u = f2u(f), s = (u >> 16) & 0x8000u;
  u &= 0x7fffffffu;
  if (u > 0x7f800000u) return (uint16_t)(s | 0x7e00);
  if (u >= 0x47800000u) return (uint16_t)(s | 0x7c00); /* >= 65536 -> inf (65520 rounds to inf below) */
  if (u < 0x38800000u) { /* subnormal half or zero */
    if (u < 0x33000000u) return (uint16_t)s;
    uint32_t e = u >> 23, m = (u & 0x7fffffu) | 0x800000u;
    uint32_t shift = 126 - e; /* 14..24 */
    uint32_t r = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1))) r++;
    return (uint16_t)(s | r);
  }
  uint32_t r = u - 0x38000000u; /* rebias */
  uint32_t rem = r & 0x1fffu;
  r >>= 13;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
  return (uint16_t)(s | r); /* may carry into inf correctly */
}

/* fp8 e4m3fn (OCP): 1-4-3, bias 7, max 448, no inf, NaN = S.1111.111 */
static inline float e4m3_to_f32(uint8_t b) {
  uint32_t s = b >> 7, e = (b >> 3) & 0xf, m = b & 7;
  float v;
  if (e == 0xf && m == 7) return NAN;
  if (e == 0) v = (float)m * (1.0f / 512.0f); /* 2^-6 * m/8 */
  else v = ldexpf(1.0f + (float)m / 8.0f, (int)e - 7);
  return s ? -v : v;
}
static inline uint8_t f32_to_e4m3_sat(float f) { /* saturating, RNE; input NaN -> 0x7f */
  if (f != f) return 0x7f;
  uint8_t s = (f2u(f) >> 24) & 0x80;
  float a = fabsf(f);
  if (a >= 448.0f) return s | 0x7e;
  if (a < 0.015625f) { /* below 2^-6: subnormal, step 2^-9 */
    float q = a * 512.0f;
    float r = nearbyintf(q); /* RNE in default rounding mode */
    return s | (uint8_t)r;   /* r == 8 -> 0x08 == 2^-6: correct carry */
  }
  int e; float fr = frexpf(a, &e); /* a = fr * 2^e, fr in [0.5,1) */
  float m = fr * 16.0f;            /* [8,16) */
  float r = nearbyintf(m);
  e -= 1;                          /* a = (m/8) * 2^e */
  if (r == 16.0f) { r = 8.0f; e += 1; }
  int be = e + 7;
  if (be > 15 || (be == 15 && r - 8.0f > 6.0f)) return s | 0x7e;
  return s | (uint8_t)((be << 3) | ((int)r - 8));
}

static inline float ld(const void* p, int dt, int64_t i) {
  switch (dt) {
    case 0: return ((const float*)p)[i];
    case 1: return bf16_to_f32(((const uint16_t*)p)[i]);
    default: return f16_to_f32(((const uint16_t*)p)[i]);
  }
}
static inline void st(void* p, int dt, int64_t i, float v) {
  switch (dt) {
    case 0: ((float*)p)[i] = v; break;
    case 1: ((uint16_t*)p)[i] = f32_to_bf16(v); break;
    default: ((uint16_t*)p)[i] = f32_to_f16(v); break;
  }
}
/* r16: round a float to the tensor dtype and back (the "scalar_t" cast point) */
static inline float r16(float v, int dt) {
  switch (dt) {
    case 0: return v;
    case 1: return bf16_to_f32(f32_to_bf16(v));
    default: return f16_to_f32(f32_to_f16(v));
  }
}
static inline int esz(int dt) { return dt == 0 ? 4 : 2; }

ORC_API void orc_convert(const void* src, int sdt, void* dst, int ddt, int64_t n) {
  for (int64_t i = 0; i < n; ++i) st(dst, ddt, i, ld(src, sdt, i));
}
ORC_API void orc_e4m3_to_f32(const uint8_t* src, float* dst, int64_t n) {
  for (int64_t i = 0; i < n; ++i) dst[i] = e4m3_to_f32(src[i]);
}
ORC_API void orc_f32_to_e4m3(const float* src, uint8_t* dst, int64_t n) {
  for (int64_t i = 0; i < n; ++i) dst[i] = f32_to_e4m3_sat(src[i]);
}

/* ------------------------------------------------------------------------- */
/* seeded_tensor: tests/core/layers/mlu/tests_utils.cpp:159-274              */
/* FNV-1a(key) -> SplitMix64 stream; floats = (u>>11) * 2^-53 in [0,1)       */
/* ------------------------------------------------------------------------- */
static uint64_t fnv1a64(const char* s) {
  uint64_t h = 0xcbf29ce484222325ULL;
  for (; *s; ++s) { h ^= (unsigned char)*s; h *= 0x100000001b3ULL; }
  return h;
}
static inline uint64_t splitmix_at(uint64_t seed, uint64_t i) { /* i-th output, 0-based */
  uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
  z ^= (z >> 30); z *= 0xBF58476D1CE4E5B9ULL;
  z ^= (z >> 27); z *= 0x94D049BB133111EBULL;
  z ^= (z >> 31);
  return z;
}
ORC_API void orc_seeded_u64(const char* key, int64_t n, uint64_t* out) {
  uint64_t seed = fnv1a64(key);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) out[i] = splitmix_at(seed, (uint64_t)i);
}
/* floating seeded_tensor: double in [0,1), then cast to dtype (torch .to(dtype) = RNE) */
ORC_API void orc_seeded_float(const char* key, int64_t n, void* out, int dt) {
  uint64_t seed = fnv1a64(key);
  const double inv = 1.0 / 9007199254740992.0;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    double v = (double)(splitmix_at(seed, (uint64_t)i) >> 11) * inv;
    /* torch double->bf16 goes double->float->bf16 */
    st(out, dt, i, (float)v);
  }
}
/* integer seeded_tensor: min + (u % span) (tests_utils.cpp:222-238) */
ORC_API void orc_seeded_i8(const char* key, int64_t n, int8_t* out) {
  uint64_t seed = fnv1a64(key);
  for (int64_t i = 0; i < n; ++i) out[i] = (int8_t)(-128 + (int)(splitmix_at(seed, (uint64_t)i) % 256u));
}
ORC_API void orc_seeded_i32(const char* key, int64_t n, int32_t* out) {
  uint64_t seed = fnv1a64(key);
  for (int64_t i = 0; i < n; ++i)
    out[i] = (int32_t)(-2147483648LL + (int64_t)(splitmix_at(seed, (uint64_t)i) % 4294967296ULL));
}

/* ------------------------------------------------------------------------- */
/* KV write: kernels/cuda/reshape_paged_cache.cu:24-63                       */
/* cache[slot/bs][slot%bs][h][:] = k[t][h][:], slot < 0 skipped              */
/* ------------------------------------------------------------------------- */
ORC_API int orc_reshape_paged_cache(const int32_t* slot_ids, const void* k, const void* v,
                                    void* k_cache, void* v_cache, int64_t T, int64_t nkv,
                                    int64_t d, int64_t block_size, int64_t n_blocks,
                                    int64_t k_stride, int64_t v_stride, int elt_bytes) {
  const int64_t row = nkv * d;
  for (int64_t t = 0; t < T; ++t) {
    int64_t slot = slot_ids[t];
    if (slot < 0) continue;
    int64_t blk = slot / block_size, off = slot % block_size;
    if (blk >= n_blocks) return -1;
    int64_t dst = (blk * block_size + off) * row;
    memcpy((char*)k_cache + dst * elt_bytes, (const char*)k + t * k_stride * elt_bytes, row * elt_bytes);
    if (v) /* K-only = MLA store_latent_cache, layers/dcu/deepseek_v2_attention.cpp:170-178 */
      memcpy((char*)v_cache + dst * elt_bytes, (const char*)v + t * v_stride * elt_bytes, row * elt_bytes);
  }
  return 0;
}

/* CSR -> dense block table: kernels/dcu/build_block_table_from_paged_kv.hip:44-72
 * (rows -1 padded, width = total_pages) */
ORC_API void orc_build_block_table(const int32_t* indptr, const int32_t* indices, int32_t B,
                                   int32_t total_pages, int32_t* table) {
  for (int64_t i = 0; i < (int64_t)B * total_pages; ++i) table[i] = -1;
  for (int32_t s = 0; s < B; ++s)
    for (int32_t j = indptr[s]; j < indptr[s + 1]; ++j)
      table[(int64_t)s * total_pages + (j - indptr[s])] = indices[j];
}

/* N2 decode metadata refresh: kernels/cuda/llm_decode_metadata_update.cu:27-60 restated as plain loops (the copies,
 * the zeroed padded tail of tokens / slots, kv_seq_lens_delta), plus the dense 0-padded block table and per-sequence
 * lengths the MI355X entry point derives in the same pass (block table rule: batch_input_builder.cpp:904-938). */
ORC_API void orc_decode_metadata_update(
    const int32_t* src_tokens, const int32_t* src_positions, const int32_t* src_slots, const int32_t* src_kv_seq_lens,
    const int32_t* src_indptr, const int32_t* src_indices, const int32_t* src_last_page_len, int32_t* dst_tokens,
    int32_t* dst_positions, int32_t* dst_slots, int32_t* dst_kv_seq_lens, int32_t* dst_delta, int32_t* dst_indptr,
    int32_t* dst_indices, int32_t* dst_last_page_len, int64_t n_tok, int64_t n_tok_padded, int64_t B, int64_t n_idx,
    int32_t* dst_block_table, int32_t* dst_kv_lens, int64_t max_blocks, int64_t B_padded) {
  for (int64_t i = 0; i < n_tok; ++i) {
    if (dst_tokens) dst_tokens[i] = src_tokens[i];
    if (dst_positions) dst_positions[i] = src_positions[i];
    if (dst_slots) dst_slots[i] = src_slots[i];
  }
  for (int64_t i = n_tok; i < n_tok_padded; ++i) {
    if (dst_tokens) dst_tokens[i] = 0;
    if (dst_slots) dst_slots[i] = 0;
  }
  for (int64_t i = 0; i < B + 1; ++i) {
    if (dst_kv_seq_lens) dst_kv_seq_lens[i] = src_kv_seq_lens[i];
    if (dst_indptr) dst_indptr[i] = src_indptr[i];
  }
  for (int64_t b = 0; b < B; ++b) {
    const int32_t len = src_kv_seq_lens[b + 1] - src_kv_seq_lens[b];
    if (dst_delta) dst_delta[b] = len;
    if (dst_kv_lens) dst_kv_lens[b] = len;
    if (dst_last_page_len) dst_last_page_len[b] = src_last_page_len[b];
  }
  if (B_padded < B) B_padded = B;
  for (int64_t b = B; b < B_padded; ++b)
    if (dst_kv_lens) dst_kv_lens[b] = 0;
  for (int64_t i = 0; i < n_idx; ++i)
    if (dst_indices) dst_indices[i] = src_indices[i];
  if (dst_block_table) {
    for (int64_t i = 0; i < B_padded * max_blocks; ++i) dst_block_table[i] = 0;
    for (int64_t b = 0; b < B; ++b)
      for (int32_t j = src_indptr[b]; j < src_indptr[b + 1] && j - src_indptr[b] < max_blocks; ++j)
        dst_block_table[b * max_blocks + (j - src_indptr[b])] = src_indices[j];
  }
}

/* ------------------------------------------------------------------------- */
/* RMSNorm family: kernels/cuda/norm.cu:45-174, 229-270                      */
/* ------------------------------------------------------------------------- */
/* y_i = r16( r16(x_i * inv) * w_i ), inv = rsqrt(mean(x^2)+eps) (norm.cu:55-73) */
ORC_API void orc_rms_norm(void* out, const void* in, const void* w, float eps, int64_t T,
                          int64_t H, int64_t in_stride, int dt) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < T; ++t) {
    double acc = 0.0; /* sum order is free in the reference (block reduce); use a wide accumulator */
    for (int64_t i = 0; i < H; ++i) { float x = ld(in, dt, t * in_stride + i); acc += (double)x * x; }
    float inv = 1.0f / sqrtf((float)(acc / (double)H) + eps);
    for (int64_t i = 0; i < H; ++i) {
      float x = ld(in, dt, t * in_stride + i);
      float n = r16(x * inv, dt);
      st(out, dt, t * H + i, n * ld(w, dt, i));
    }
  }
}
/* residual <- r16(input + residual); input <- norm(residual)  (norm.cu:82-174) */
ORC_API void orc_fused_add_rms_norm(void* in, void* res, const void* w, float eps, int64_t T,
                                    int64_t H, int64_t in_stride, int dt) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < T; ++t) {
    double acc = 0.0;
    for (int64_t i = 0; i < H; ++i) {
      float z = r16(ld(in, dt, t * in_stride + i) + ld(res, dt, t * H + i), dt);
      st(res, dt, t * H + i, z);
      acc += (double)z * z;
    }
    float inv = 1.0f / sqrtf((float)(acc / (double)H) + eps);
    for (int64_t i = 0; i < H; ++i) {
      float z = ld(res, dt, t * H + i);
      float n = r16(z * inv, dt);
      st(in, dt, t * in_stride + i, n * ld(w, dt, i));
    }
  }
}
/* v = float(r16(x*inv)) * float(w) (fp32 product, no 2nd 16-bit round);
 * q = e4m3fn_sat(clamp(v * (1/scale), +-448))  (norm.cu:229-270, fp8_quant_utils.cuh:112-129)
 * If res != NULL: fused-add variant (norm.cu:282-425): residual updated in place first. */
ORC_API void orc_rms_norm_static_fp8_quant(uint8_t* out, const void* in, void* res, const void* w,
                                           const float* scale, float eps, int64_t T, int64_t H,
                                           int64_t in_stride, int dt) {
  const float sinv = 1.0f / scale[0];
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < T; ++t) {
    double acc = 0.0;
    for (int64_t i = 0; i < H; ++i) {
      float x = ld(in, dt, t * in_stride + i);
      if (res) { x = r16(x + ld(res, dt, t * H + i), dt); st(res, dt, t * H + i, x); }
      acc += (double)x * x;
    }
    float inv = 1.0f / sqrtf((float)(acc / (double)H) + eps);
    for (int64_t i = 0; i < H; ++i) {
      float x = res ? ld(res, dt, t * H + i) : ld(in, dt, t * in_stride + i);
      float v = r16(x * inv, dt) * ld(w, dt, i);
      float q = v * sinv;
      q = fmaxf(-448.0f, fminf(q, 448.0f));
      out[t * H + i] = f32_to_e4m3_sat(q);
    }
  }
}

/* ------------------------------------------------------------------------- */
/* RoPE: kernels/cuda/rope.cu:27-154; cache build                            */
/* layers/common/rotary_embedding_util.cpp:157-192, rotary_embedding.cpp:46-52 */
/* ------------------------------------------------------------------------- */
/* cache[pos] = [cos(pos*inv_freq_k) (rot/2) || sin(...) (rot/2)], cast to dtype */
ORC_API void orc_build_cos_sin_cache(void* cache, int64_t max_pos, int64_t rot_dim, float theta,
                                     int dt) {
  int64_t half = rot_dim / 2;
  for (int64_t p = 0; p < max_pos; ++p)
    for (int64_t k = 0; k < half; ++k) {
      /* inv_freq = 1 / theta^(2k/rot) computed in fp32 like torch::pow on a float tensor */
      float inv_freq = 1.0f / powf(theta, (float)(2 * k) / (float)rot_dim);
      float fr = (float)p * inv_freq;
      st(cache, dt, p * rot_dim + k, cosf(fr));
      st(cache, dt, p * rot_dim + half + k, sinf(fr));
    }
}
/* arithmetic entirely in scalar_t: q[x] = r16(r16(x*c) - r16(y*s)), q[y] = r16(r16(y*c) + r16(x*s))
 * (rope.cu:50-53 uses scalar_t operators: each * and +- rounds to 16 bit) */
static void rope_one(void* arr, int64_t base, const void* cache, int64_t cbase, int64_t j,
                     int64_t half, int is_neox, int dt) {
  int64_t xi = is_neox ? j : 2 * j, yi = is_neox ? half + j : 2 * j + 1;
  float c = ld(cache, dt, cbase + j), s = ld(cache, dt, cbase + half + j);
  float x = ld(arr, dt, base + xi), y = ld(arr, dt, base + yi);
  float nx = r16(r16(x * c, dt) - r16(y * s, dt), dt);
  float ny = r16(r16(y * c, dt) + r16(x * s, dt), dt);
  st(arr, dt, base + xi, nx);
  st(arr, dt, base + yi, ny);
}
ORC_API void orc_rotary_embedding(const int64_t* positions, void* q, void* k, const void* cache,
                                  int64_t T, int64_t nq, int64_t nk, int64_t head_size,
                                  int64_t rot_dim, int64_t q_stride, int64_t k_stride,
                                  int64_t head_stride, int is_neox, int dt) {
  int64_t half = rot_dim / 2;
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < T; ++t) {
    int64_t cb = positions[t] * rot_dim;
    for (int64_t h = 0; h < nq; ++h)
      for (int64_t j = 0; j < half; ++j)
        rope_one(q, t * q_stride + h * head_stride, cache, cb, j, half, is_neox, dt);
    if (k)
      for (int64_t h = 0; h < nk; ++h)
        for (int64_t j = 0; j < half; ++j)
          rope_one(k, t * k_stride + h * head_stride, cache, cb, j, half, is_neox, dt);
  }
  (void)head_size;
}

/* ------------------------------------------------------------------------- */
/* act_and_mul: kernels/cuda/activation.cu:30-120                            */
/* out = r16( r16(act(float(x))) * y ); mode 0 silu, 1 gelu(erf), 2 gelu_tanh */
/* ------------------------------------------------------------------------- */
static inline float act_fn(float f, int mode) {
  if (mode == 0) return f / (1.0f + expf(-f));
  if (mode == 1) return f * 0.5f * (1.0f + erff(f * 0.70710678118654752440f));
  float kBeta = 1.41421356237309504880f * 1.12837916709551257390f * 0.5f;
  float inner = kBeta * (f + 0.044715f * f * f * f);
  return 0.5f * f * (1.0f + tanhf(inner));
}
ORC_API void orc_act_and_mul(void* out, const void* in, int64_t T, int64_t d, int mode, int dt) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < T; ++t)
    for (int64_t i = 0; i < d; ++i) {
      float a = r16(act_fn(ld(in, dt, t * 2 * d + i), mode), dt);
      st(out, dt, t * d + i, a * ld(in, dt, t * 2 * d + d + i));
    }
}

/* ------------------------------------------------------------------------- */
/* int8 per-token quant: kernels/dcu/scaled_quantize.hip:29-33,66-108        */
/* ------------------------------------------------------------------------- */
ORC_API void orc_scaled_quantize_i8(const void* x, int8_t* out, float* scales, int64_t M, int64_t K,
                                    int dt) {
#pragma omp parallel for schedule(static)
  for (int64_t m = 0; m < M; ++m) {
    float mx = 0.0f;
    for (int64_t k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(ld(x, dt, m * K + k)));
    float inv = (mx > 1e-10f) ? 127.0f / mx : 0.0f;
    for (int64_t k = 0; k < K; ++k) {
      float q = nearbyintf(ld(x, dt, m * K + k) * inv);
      q = fmaxf(-127.0f, fminf(127.0f, q));
      out[m * K + k] = (int8_t)q;
    }
    scales[m] = mx / 127.0f;
  }
}

/* ------------------------------------------------------------------------- */
/* int8 GEMM + dequant epilogue: kernels/dcu/scaled_matmul.cpp:103-300        */
/* acc exact int32; out = r16(float(acc) * a_s[m] * w_s[n] + bias[n])         */
/* acc_out (optional) receives the raw int32 accumulators (bit-exact check). */
/* ------------------------------------------------------------------------- */
ORC_API void orc_scaled_matmul_i8(const int8_t* a, const int8_t* w, const float* a_scale,
                                  const float* w_scale, const void* bias, void* out,
                                  int32_t* acc_out, int64_t M, int64_t N, int64_t K, int out_dt) {
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < N; ++n) {
    const int8_t* wr = w + n * K;
    for (int64_t m = 0; m < M; ++m) {
      const int8_t* ar = a + m * K;
      int32_t acc = 0;
      for (int64_t k = 0; k < K; ++k) acc += (int32_t)ar[k] * (int32_t)wr[k];
      if (acc_out) acc_out[m * N + n] = acc;
      if (out) {
        float v = (float)acc * a_scale[m] * w_scale[n];
        if (bias) v += ld(bias, out_dt, n);
        st(out, out_dt, m * N + n, v);
      }
    }
  }
}

/* bf16/fp16/f32 linear: kernels/dcu/matmul.cpp:20-25 (F::linear): out = r16(sum_fp32 + bias) */
ORC_API void orc_matmul(const void* a, const void* w, const void* bias, void* out, int64_t M,
                        int64_t N, int64_t K, int dt) {
  float* af = (float*)malloc(sizeof(float) * (size_t)(M * K));
  for (int64_t i = 0; i < M * K; ++i) af[i] = ld(a, dt, i);
#pragma omp parallel
  {
    float* wf = (float*)malloc(sizeof(float) * (size_t)K);
#pragma omp for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
      for (int64_t k = 0; k < K; ++k) wf[k] = ld(w, dt, n * K + k);
      for (int64_t m = 0; m < M; ++m) {
        const float* ar = af + m * K;
        float acc = 0.0f;
        for (int64_t k = 0; k < K; ++k) acc += ar[k] * wf[k];
        if (bias) acc += ld(bias, dt, n);
        st(out, dt, m * N + n, acc);
      }
    }
    free(wf);
  }
  free(af);
}

/* ------------------------------------------------------------------------- */
/* fp8: kernels/cuda/fp8_quant.cu:79-106, fp8_scaled_quantize.cpp:36-47,      */
/* cutlass_extensions/epilogue/scaled_mm_epilogues_c3x.hpp:173-270           */
/* ------------------------------------------------------------------------- */
/* q = e4m3fn_sat(clamp(x * (1/scale), +-448)), per-tensor scale[1] */
ORC_API void orc_static_scaled_fp8_quant(uint8_t* out, const void* in, const float* scale,
                                         int64_t n, int dt) {
  const float sinv = 1.0f / scale[0];
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    float q = ld(in, dt, i) * sinv;
    q = fmaxf(-448.0f, fminf(q, 448.0f));
    out[i] = f32_to_e4m3_sat(q);
  }
}
/* dynamic per-tensor scale = max(amax/448, 1e-12); amax taken in the tensor dtype then /448 in that
 * dtype (torch: (amax / 448.0f).clamp_min(1e-12f).to(kFloat32) on a 16-bit 0-dim tensor) */
ORC_API void orc_fp8_dynamic_scale(const void* in, int64_t n, int dt, float* scale) {
  float mx = 0.0f;
  for (int64_t i = 0; i < n; ++i) mx = fmaxf(mx, fabsf(ld(in, dt, i)));
  float s = r16(mx / 448.0f, dt);
  s = r16(fmaxf(s, r16(1e-12f, dt)), dt);
  scale[0] = s;
}
/* out = r16( a_s * (w_s * sum_fp32(a*w)) + bias ); scales: numel 1 (scalar) or M / N (vector) */
ORC_API void orc_fp8_scaled_matmul(const uint8_t* a, const uint8_t* w, const float* a_scale,
                                   int64_t a_scale_n, const float* w_scale, int64_t w_scale_n,
                                   const void* bias, void* out, int64_t M, int64_t N, int64_t K,
                                   int out_dt) {
  float* af = (float*)malloc(sizeof(float) * (size_t)(M * K));
  for (int64_t i = 0; i < M * K; ++i) af[i] = e4m3_to_f32(a[i]);
#pragma omp parallel
  {
    float* wf = (float*)malloc(sizeof(float) * (size_t)K);
#pragma omp for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
      for (int64_t k = 0; k < K; ++k) wf[k] = e4m3_to_f32(w[n * K + k]);
      float ws = w_scale[w_scale_n > 1 ? n : 0];
      for (int64_t m = 0; m < M; ++m) {
        const float* ar = af + m * K;
        float acc = 0.0f;
        for (int64_t k = 0; k < K; ++k) acc += ar[k] * wf[k];
        float v = a_scale[a_scale_n > 1 ? m : 0] * (ws * acc);
        if (bias) v += ld(bias, out_dt, n);
        st(out, out_dt, m * N + n, v);
      }
    }
    free(wf);
  }
  free(af);
}

/* thread count of the OpenMP regions (bench.py: small per-call work drowns in fork / join with hundreds of threads) */
#include <omp.h>
ORC_API void orc_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
ORC_API int orc_get_max_threads(void) { return omp_get_max_threads(); }

/* ------------------------------------------------------------------------- */
/* Attention (all modes): layers/dcu/torch_attention.cpp:40-345 (GQA expand,  */
/* per-sequence SDPA, page gather with last_page_len) and the eager variant   */
/* layers/cuda/flashinfer_attention.cpp:34-95 (fp32 QK^T, fp32 softmax, cast  */
/* P to the tensor dtype when p_round != 0, PV).                              */
/* Causal masks are BOTTOM-RIGHT aligned (query i of q_len sees keys           */
/* j <= kv_len - q_len + i): SURVEY.md 8c caveat (2) -- torch_attention's      */
/* is_causal with S_q < S_k is top-left and is NOT followed.                   */
/* window_left < 0: unbounded; else key j visible iff j >= i_abs - window_left */
/* ------------------------------------------------------------------------- */
/* p_round == 2: the FLASH cast point (round 3). The reference's eager spec rounds the NORMALISED P to the tensor dtype before
 * PV (flashinfer_attention.cpp:84-90: softmax in fp32, .to(dtype), @ V); a flash kernel -- the reference's production kernels
 * (FlashInfer / flash-attention behind layers/cuda and layers/dcu) and xllm_amd/csrc/attention_prefill.hip -- never holds the
 * normalised P: it rounds the UN-normalised tile P = exp(s - running max) and divides the fp32 accumulator by the fp32 row sum
 * at the end. THAT P is rounded to 16 bit is the reference's; the tile order, tile size (64 keys, aligned at key 0) and the lazy
 * maximum (the running maximum only moves when a tile exceeds it by more than 2^8, so P <= 256) are this repository's kernel's
 * (attention_prefill.hip::pf2_softmax), restated here so that the kernel can be held to an ABSOLUTE bar against an oracle with
 * its own cast point. exp is taken as exp2 of the log2-scaled argument with one fma, as the kernel does. */
static void attn_one_query_flash(const float* qv, const void* kbase, const void* vbase, const int64_t* row_off,
                                 int64_t n_keys, int64_t d, int64_t dv, float scale, int dt, float* sc, float* outv) {
  const float sl2 = scale * 1.4426950408889634f;
  const int64_t TILE = 64;
  for (int64_t e = 0; e < dv; ++e) outv[e] = 0.0f;
  int64_t first = 0;
  while (first < n_keys && row_off[first] < 0) ++first;
  if (first == n_keys) return;
  float m_run = -INFINITY, l = 0.0f;
  for (int64_t t0 = first / TILE * TILE; t0 < n_keys; t0 += TILE) {
    const int64_t t1 = t0 + TILE < n_keys ? t0 + TILE : n_keys;
    float mx = -INFINITY;
    for (int64_t j = t0; j < t1; ++j) {
      if (row_off[j] < 0) { sc[j] = -INFINITY; continue; }
      float acc = 0.0f;
      for (int64_t e = 0; e < d; ++e) acc += qv[e] * ld(kbase, dt, row_off[j] + e);
      sc[j] = acc;                       /* RAW score: the maximum is taken before the scale (scale > 0) */
      mx = fmaxf(mx, acc);
    }
    if (mx == -INFINITY) continue;       /* nothing visible in this tile */
    const float mxs = mx * sl2;
    const float m_new = mxs > m_run + 8.0f ? mxs : m_run;
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    l *= alpha;
    if (alpha != 1.0f) for (int64_t e = 0; e < dv; ++e) outv[e] *= alpha;
    for (int64_t j = t0; j < t1; ++j) {
      if (sc[j] == -INFINITY) continue;
      const float p = exp2f(fmaf(sc[j], sl2, -m_new));
      l += p;                            /* the row sum stays fp32 */
      const float pr = r16(p, dt);       /* what PV sees */
      if (pr == 0.0f) continue;
      for (int64_t e = 0; e < dv; ++e) outv[e] += pr * ld(vbase, dt, row_off[j] + e);
    }
  }
  if (l > 0.0f) for (int64_t e = 0; e < dv; ++e) outv[e] /= l;
}

static void attn_one_query(const float* qv, /* d */
                           const void* kbase, const void* vbase, /* rows fetched via row_off */
                           const int64_t* row_off, int64_t n_keys, int64_t d, int64_t dv,
                           float scale, int dt, int p_round, float* sc, float* outv) {
  if (p_round == 2) { attn_one_query_flash(qv, kbase, vbase, row_off, n_keys, d, dv, scale, dt, sc, outv); return; }
  float mx = -INFINITY;
  for (int64_t j = 0; j < n_keys; ++j) {
    if (row_off[j] < 0) { sc[j] = -INFINITY; continue; }
    float acc = 0.0f;
    for (int64_t e = 0; e < d; ++e) acc += qv[e] * ld(kbase, dt, row_off[j] + e);
    sc[j] = acc * scale;
    mx = fmaxf(mx, sc[j]);
  }
  for (int64_t e = 0; e < dv; ++e) outv[e] = 0.0f;
  if (mx == -INFINITY) return;
  float sum = 0.0f;
  for (int64_t j = 0; j < n_keys; ++j) { sc[j] = (sc[j] == -INFINITY) ? 0.0f : expf(sc[j] - mx); sum += sc[j]; }
  float isum = 1.0f / sum;
  for (int64_t j = 0; j < n_keys; ++j) {
    if (sc[j] == 0.0f) continue;
    float p = sc[j] * isum;
    if (p_round) p = r16(p, dt);
    for (int64_t e = 0; e < dv; ++e) outv[e] += p * ld(vbase, dt, row_off[j] + e);
  }
}

/* prefill over packed q,k,v (torch_attention.cpp:152-212 == flash ragged_run):
 * q [Tq, nq, d] (token stride q_stride), k/v [Tk, nkv, d] (strides k_stride/v_stride),
 * out [Tq, nq*d] contiguous. */
ORC_API int orc_attention_varlen(const void* q, const void* k, const void* v, void* out,
                                 const int32_t* cu_q, const int32_t* cu_k, int64_t B, int64_t nq,
                                 int64_t nkv, int64_t d, int64_t q_stride, int64_t k_stride,
                                 int64_t v_stride, float scale, int causal, int64_t window_left,
                                 int dt, int p_round) {
  if (nkv <= 0 || nq % nkv) return -1;
  if (p_round == 2 && k_stride != v_stride) return -3;
  const int64_t grp = nq / nkv;
  int64_t max_k = 0;
  for (int64_t b = 0; b < B; ++b) if (cu_k[b + 1] - cu_k[b] > max_k) max_k = cu_k[b + 1] - cu_k[b];
#pragma omp parallel
  {
    float* sc = (float*)malloc(sizeof(float) * (size_t)(max_k + 1));
    int64_t* ro_k = (int64_t*)malloc(sizeof(int64_t) * (size_t)(max_k + 1));
    int64_t* ro_v = (int64_t*)malloc(sizeof(int64_t) * (size_t)(max_k + 1));
    float* qv = (float*)malloc(sizeof(float) * (size_t)d);
    float* ov = (float*)malloc(sizeof(float) * (size_t)d);
    for (int64_t b = 0; b < B; ++b) {
      int64_t q0 = cu_q[b], ql = cu_q[b + 1] - q0, k0 = cu_k[b], kl = cu_k[b + 1] - k0;
#pragma omp for schedule(dynamic, 4) collapse(2)
      for (int64_t i = 0; i < ql; ++i)
        for (int64_t h = 0; h < nq; ++h) {
          int64_t kvh = h / grp;
          int64_t iabs = kl - ql + i; /* bottom-right */
          for (int64_t j = 0; j < kl; ++j) {
            int vis = 1;
            if (causal && j > iabs) vis = 0;
            if (window_left >= 0 && j < iabs - window_left) vis = 0;
            ro_k[j] = vis ? (k0 + j) * k_stride + kvh * d : -1;
          }
          for (int64_t e = 0; e < d; ++e) qv[e] = ld(q, dt, (q0 + i) * q_stride + h * d + e);
          /* k and v may have different token strides: run twice sharing scores via a 2-pass trick:
           * attn_one_query uses one row_off for both; so materialise v offsets equal when strides match */
          if (k_stride == v_stride) {
            attn_one_query(qv, k, v, ro_k, kl, d, d, scale, dt, p_round, sc, ov);
          } else {   /* (p_round == 2 needs equal strides: checked before the parallel region) */
            /* general: compute with k offsets for scores, then redo PV with v offsets */
            for (int64_t j = 0; j < kl; ++j) ro_v[j] = ro_k[j] < 0 ? -1 : (k0 + j) * v_stride + kvh * d;
            float mx = -INFINITY;
            for (int64_t j = 0; j < kl; ++j) {
              if (ro_k[j] < 0) { sc[j] = -INFINITY; continue; }
              float acc = 0.0f;
              for (int64_t e = 0; e < d; ++e) acc += qv[e] * ld(k, dt, ro_k[j] + e);
              sc[j] = acc * scale; mx = fmaxf(mx, sc[j]);
            }
            for (int64_t e = 0; e < d; ++e) ov[e] = 0.0f;
            if (mx != -INFINITY) {
              float sum = 0.0f;
              for (int64_t j = 0; j < kl; ++j) { sc[j] = (sc[j] == -INFINITY) ? 0.0f : expf(sc[j] - mx); sum += sc[j]; }
              float isum = 1.0f / sum;
              for (int64_t j = 0; j < kl; ++j) {
                if (sc[j] == 0.0f) continue;
                float p = sc[j] * isum; if (p_round) p = r16(p, dt);
                for (int64_t e = 0; e < d; ++e) ov[e] += p * ld(v, dt, ro_v[j] + e);
              }
            }
          }
          for (int64_t e = 0; e < d; ++e) st(out, dt, (q0 + i) * nq * d + h * d + e, ov[e]);
        }
    }
    free(sc); free(ro_k); free(ro_v); free(qv); free(ov);
  }
  return 0;
}

/* chunked prefill + decode over the paged cache (torch_attention.cpp:213-337; arg set of
 * prefix_decode_varlen_fwd, layers/dcu/flash_attention.cpp:74-94, 220-288):
 * q [Tq, nq, d] packed by cu_q; caches [n_blocks, bs, nkv, d]; kv_lens[B]; block_table [B, max_blocks]
 * (padding entries are never read: kv_lens decides, SURVEY 8c caveat 5). A page id >= n_blocks or < 0
 * inside the live range is an error (-2), caveat 3.  MLA: nkv = 1, d = 576, dv = 512, v_cache == k_cache. */
ORC_API int orc_paged_attention(const void* q, const void* k_cache, const void* v_cache, void* out,
                                const int32_t* cu_q, const int32_t* kv_lens,
                                const int32_t* block_table, int64_t max_blocks, int64_t B,
                                int64_t nq, int64_t nkv, int64_t d, int64_t dv, int64_t block_size,
                                int64_t n_blocks, int64_t q_stride, float scale, int causal,
                                int64_t window_left, int dt, int p_round) {
  if (nkv <= 0 || nq % nkv) return -1;
  const int64_t grp = nq / nkv;
  int64_t max_k = 0;
  for (int64_t b = 0; b < B; ++b) {
    if (kv_lens[b] > max_k) max_k = kv_lens[b];
    int64_t np = (kv_lens[b] + block_size - 1) / block_size;
    if (np > max_blocks) return -2;
    for (int64_t p = 0; p < np; ++p) {
      int32_t id = block_table[b * max_blocks + p];
      if (id < 0 || id >= n_blocks) return -2;
    }
  }
  const int64_t row = nkv * d; /* elements per token row in the cache */
#pragma omp parallel
  {
    float* sc = (float*)malloc(sizeof(float) * (size_t)(max_k + 1));
    int64_t* ro = (int64_t*)malloc(sizeof(int64_t) * (size_t)(max_k + 1));
    float* qv = (float*)malloc(sizeof(float) * (size_t)d);
    float* ov = (float*)malloc(sizeof(float) * (size_t)dv);
#pragma omp for schedule(dynamic, 1)
    for (int64_t wi = 0; wi < B * nq; ++wi) { /* one work item per (sequence, q head) */
      const int64_t b = wi / nq, h = wi % nq;
      int64_t q0 = cu_q[b], ql = cu_q[b + 1] - q0, kl = kv_lens[b];
      for (int64_t i = 0; i < ql; ++i)
        {
          int64_t kvh = h / grp;
          int64_t iabs = kl - ql + i;
          for (int64_t j = 0; j < kl; ++j) {
            int vis = 1;
            if (causal && j > iabs) vis = 0;
            if (window_left >= 0 && j < iabs - window_left) vis = 0;
            int64_t pg = block_table[b * max_blocks + j / block_size];
            ro[j] = vis ? (pg * block_size + j % block_size) * row + kvh * d : -1;
          }
          for (int64_t e = 0; e < d; ++e) qv[e] = ld(q, dt, (q0 + i) * q_stride + h * d + e);
          attn_one_query(qv, k_cache, v_cache, ro, kl, d, dv, scale, dt, p_round, sc, ov);
          for (int64_t e = 0; e < dv; ++e) st(out, dt, (q0 + i) * nq * dv + h * dv + e, ov[e]);
        }
    }
    free(sc); free(ro); free(qv); free(ov);
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* fused per-head RMSNorm(q,k) + RoPE inside packed qkv:                      */
/* kernels/cuda/fused_qknorm_rope.cu:88-300 (fp32 math, one 16-bit store)     */
/* qkv [T, (nq+nk+nv)*d]; cos_sin cache fp32-or-dtype [max_pos, d] = [cos||sin]*/
/* ------------------------------------------------------------------------- */
ORC_API void orc_fused_qk_norm_rope(void* qkv, int64_t T, int64_t nq, int64_t nk, int64_t nv,
                                    int64_t d, float eps, const void* qw, const void* kw,
                                    const void* cache, int cache_dt, int interleaved,
                                    const int64_t* positions, int dt) {
  const int64_t rowlen = (nq + nk + nv) * d, half = d / 2;
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < T; ++t) {
    float tmp[1024];
    for (int64_t h = 0; h < nq + nk; ++h) {
      const void* w = h < nq ? qw : kw;
      int64_t base = t * rowlen + h * d;
      float ss = 0.0f;
      for (int64_t e = 0; e < d; ++e) { float x = ld(qkv, dt, base + e); ss += x * x; }
      float inv = 1.0f / sqrtf(ss / (float)d + eps);
      for (int64_t e = 0; e < d; ++e) tmp[e] = ld(qkv, dt, base + e) * inv * ld(w, dt, e);
      int64_t cb = positions[t] * d;
      for (int64_t j = 0; j < half; ++j) {
        int64_t xi = interleaved ? 2 * j : j, yi = interleaved ? 2 * j + 1 : half + j;
        float c = ld(cache, cache_dt, cb + j), s = ld(cache, cache_dt, cb + half + j);
        float x = tmp[xi], y = tmp[yi];
        st(qkv, dt, base + xi, x * c - y * s);
        st(qkv, dt, base + yi, y * c + x * s);
      }
    }
  }
}

/* ------------------------------------------------------------------------- */
/* MoE helpers: kernels/cuda/moe/moe_compute_index.cu:41-160 (a STABLE order  */
/* is produced here; the reference's intra-expert order is atomics-defined so */
/* only permutation-invariant results may be compared), moe_combine.cu:38-62, */
/* kernels/dcu/group_gemm.cpp:25-74                                           */
/* ------------------------------------------------------------------------- */
ORC_API void orc_moe_compute_index(const int32_t* expert_id, int64_t T, int64_t topk, int64_t E,
                                   int32_t* src_dst, int32_t* dst_src, int32_t* expert_sizes) {
  int64_t n = T * topk;
  int32_t* off = (int32_t*)calloc((size_t)E + 1, sizeof(int32_t));
  for (int64_t e = 0; e < E; ++e) expert_sizes[e] = 0;
  for (int64_t i = 0; i < n; ++i) expert_sizes[expert_id[i]]++;
  for (int64_t e = 0; e < E; ++e) off[e + 1] = off[e] + expert_sizes[e];
  for (int64_t i = 0; i < n; ++i) {
    int32_t pos = off[expert_id[i]]++;
    src_dst[i] = pos;   /* expanded row i (= t*topk+k) goes to sorted row pos */
    dst_src[pos] = (int32_t)i;
  }
  free(off);
}
/* out[t] = r16( sum_k w[t,k] * float(gemm2[t*topk+k]) ), fp32 accumulate */
ORC_API void orc_moe_combine(void* out, const void* gemm2, const float* w, int64_t T, int64_t topk,
                             int64_t H, int dt) {
#pragma omp parallel for schedule(static)
  for (int64_t t = 0; t < T; ++t)
    for (int64_t i = 0; i < H; ++i) {
      float acc = 0.0f;
      for (int64_t k = 0; k < topk; ++k) acc += w[t * topk + k] * ld(gemm2, dt, (t * topk + k) * H + i);
      st(out, dt, t * H + i, acc);
    }
}
/* C_e = A[off_e : off_e + M_e] * W_e^T ; W [E, N, K] */
ORC_API void orc_group_gemm(const void* a, const void* w, const int32_t* token_count, void* out,
                            int64_t E, int64_t N, int64_t K, int dt) {
  int64_t off = 0;
  for (int64_t e = 0; e < E; ++e) {
    int64_t Me = token_count[e];
    if (Me > 0)
      orc_matmul((const char*)a + off * K * esz(dt), (const char*)w + e * N * K * esz(dt), NULL,
                 (char*)out + off * N * esz(dt), Me, N, K, dt);
    off += Me;
  }
}


/* ------------------------------------------------------------------------- */
/* N3 sampler: kernels/dcu/random_sample.hip:88-270, rejection_sample.hip:33-139 */
/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Salmon et al. 2011) as hiprand / rocRAND drive it for hiprand_init(seed, subsequence, offset):
 * counter = (offset/4 lo, offset/4 hi, subsequence lo, subsequence hi), key = seed, output word offset%4;
 * hiprand_uniform = 2^-32 + x*2^-32 in fp32 (rocrand_uniform.h). The GPU test pins this against hiprand itself. */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t m0 = (uint64_t)0xD2511F53u * c[0], m1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(m1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)m1, n2 = (uint32_t)(m0 >> 32) ^ c[3] ^ k1,
             n3 = (uint32_t)m0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
ORC_API void orc_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { /* known-answer tests */
  uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
  philox4x32_10(c, key[0], key[1]);
  for (int i = 0; i < 4; ++i) out[i] = c[i];
}
ORC_API void orc_philox_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset) {
  for (int64_t i = 0; i < n; ++i) {
    uint64_t blk = offset >> 2;
    uint32_t c[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)i, (uint32_t)((uint64_t)i >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    volatile float x = (float)c[offset & 3];
    volatile float y = x * 2.3283064e-10f;
    out[i] = 2.3283064e-10f + y;
  }
}
/* CDF inversion in exact-enough arithmetic (fp64 prefix sums): first index with p > 0 and cdf > u, else the last index
 * with p > 0, else 0. Also returns cdf just below / at the chosen index so that callers can bound fp32 kernels. */
ORC_API void orc_random_sample(const float* probs, const float* u, int64_t B, int64_t V, int32_t* out) {
  for (int64_t b = 0; b < B; ++b) {
    const float* p = probs + b * V;
    double cdf = 0.0;
    int32_t pick = -1, last = -1;
    for (int64_t i = 0; i < V; ++i) {
      if (!(p[i] > 0.0f)) continue;
      last = (int32_t)i;
      cdf += (double)p[i];
      if (pick < 0 && cdf > (double)u[b]) pick = (int32_t)i;
    }
    out[b] = pick >= 0 ? pick : (last >= 0 ? last : 0);
  }
}
ORC_API void orc_rejection_sample(const int32_t* draft_token_ids, const int32_t* num_draft_tokens,
                                  const int32_t* cu_num_draft_tokens, const float* draft_probs,
                                  const float* target_probs, const int32_t* bonus_token_ids,
                                  const float* uniform_rand, const float* uniform_probs, int64_t B, int64_t V,
                                  int32_t* output) {
  for (int64_t s = 0; s < B; ++s) {
    const int32_t n = num_draft_tokens[s], end = cu_num_draft_tokens[s], start = end - n;
    const int64_t o = start + s;
    for (int32_t i = 0; i < n + 1; ++i) output[o + i] = -1;
    int32_t di = 0;
    int stopped = 0;
    for (; di < n; ++di) {
      const int64_t row = start + di;
      const int32_t tok = draft_token_ids[row];
      if (tok < 0 || tok >= V) { stopped = 1; break; }
      const float* dp = draft_probs + row * V;
      const float* tp = target_probs + row * V;
      const float d = dp[tok] > 0.0f ? dp[tok] : 0.0f, t = tp[tok] > 0.0f ? tp[tok] : 0.0f;
      const float accept = d > 0.0f ? t / d : (t > 0.0f ? 1.0f : 0.0f);
      if (uniform_rand[row] < accept) { output[o + di] = tok; continue; }
      float best = -1.0f;
      int32_t best_tok = 0;
      for (int64_t v = 0; v < V; ++v) {
        const float rec = fmaxf(tp[v] - dp[v], 0.0f);
        const float uu = fmaxf(uniform_probs[row * V + v], 1.17549435e-38f);
        const float sc = rec / uu;
        if (sc > best) { best = sc; best_tok = (int32_t)v; }   /* ascending v: ties keep the lowest index */
      }
      output[o + di] = best_tok;
      stopped = 1;
      break;
    }
    if (!stopped) output[o + n] = bonus_token_ids[s];
  }
}


/* N4 gating top-k: kernels/cuda/moe/moe_fused_topk.cu:31-61, moe_topk_softmax_kernels.cuh:324-625,
 * moe_topk_sigmoid_kernels.cuh:156-392. gating is fp32 here (callers convert). */
ORC_API void orc_moe_fused_topk(const float* gating, int64_t T, int64_t E, int64_t topk, int renormalize,
                                const float* bias, int sigmoid, float* out_w, int32_t* out_id) {
  float* v = (float*)malloc(sizeof(float) * (size_t)E);
  for (int64_t t = 0; t < T; ++t) {
    const float* x = gating + t * E;
    if (sigmoid) {
      for (int64_t e = 0; e < E; ++e) {
        float s = 1.0f / (1.0f + expf(-x[e]));
        v[e] = bias ? s + bias[e] : s;
      }
    } else {
      float mx = -INFINITY, sum = 0.0f;
      for (int64_t e = 0; e < E; ++e) mx = fmaxf(mx, x[e]);
      for (int64_t e = 0; e < E; ++e) { v[e] = expf(x[e] - mx); sum += v[e]; }
      const float inv = 1.0f / sum;
      for (int64_t e = 0; e < E; ++e) v[e] = v[e] * inv;
    }
    float wsum = 0.0f;
    for (int64_t k = 0; k < topk; ++k) {
      int64_t best = 0;
      for (int64_t e = 1; e < E; ++e)
        if (v[e] > v[best]) best = e;            /* ascending scan: ties keep the lower index */
      float w = v[best];
      if (sigmoid && bias) w = w - bias[best];
      out_w[t * topk + k] = w;
      out_id[t * topk + k] = (int32_t)best;
      wsum += w;
      v[best] = -INFINITY;
    }
    if (renormalize) {
      const float inv = 1.0f / wsum;
      for (int64_t k = 0; k < topk; ++k) out_w[t * topk + k] = out_w[t * topk + k] * inv;
    }
  }
  free(v);
}


/* Grouped gating top-k (DeepSeek-V2 / V3 device-limited routing). Reference call site: dcu::moe_grouped_topk
 * (kernels/dcu/topk_gate.cpp:59-125) -> aiter::native::grouped_topk / biased_grouped_topk. aiter (ROCm/aiter) is an
 * external library that is NOT in the reference tree (declared by hand at topk_gate.cpp:24-46, no pinned version), so this
 * restates the PUBLISHED algorithm (DeepSeek-V2 / V3 gate, as in the open implementations of grouped_topk):
 *   score   s[e] = softmax(x)[e] or sigmoid(x[e]);   choice score c[e] = s[e] + bias[e] (bias only with sigmoid)
 *   group   value = max of c over the group (no bias)  |  sum of the two largest c of the group (with bias)
 *   keep    the topk_group best groups (ties: lower group index); experts of the other groups are out of the race
 *   pick    topk experts by c among the kept groups (ties: lower expert index); weight = s[e] (the UNBIASED score)
 *   renormalize: weights /= their sum;  then weights *= routed_scaling_factor.
 * PINNED on the reference's own fixtures (tests/core/layers/mlu/moe_gate_test.cpp:143-272: min / max / sum of weights and
 * expert ids for sigmoid + correction bias, softmax and the all-ties topk_group = 1 case, seeded inputs): reproduced exactly
 * by tests/test_reference_fixtures.py, which also settles the tie rules (lower index wins). */
ORC_API void orc_moe_grouped_topk(const float* gating, int64_t T, int64_t E, int64_t topk, int64_t G, int64_t topk_group,
                                  int renormalize, const float* bias, int sigmoid, float route_scale, float* out_w,
                                  int32_t* out_id) {
  const int64_t EG = E / G;
  float* s = (float*)malloc(sizeof(float) * (size_t)E);
  float* c = (float*)malloc(sizeof(float) * (size_t)E);
  float* gs = (float*)malloc(sizeof(float) * (size_t)G);
  for (int64_t t = 0; t < T; ++t) {
    const float* x = gating + t * E;
    if (sigmoid) {
      for (int64_t e = 0; e < E; ++e) s[e] = 1.0f / (1.0f + expf(-x[e]));
    } else {
      float mx = -INFINITY, sum = 0.0f;
      for (int64_t e = 0; e < E; ++e) mx = fmaxf(mx, x[e]);
      for (int64_t e = 0; e < E; ++e) { s[e] = expf(x[e] - mx); sum += s[e]; }
      const float inv = 1.0f / sum;
      for (int64_t e = 0; e < E; ++e) s[e] = s[e] * inv;
    }
    for (int64_t e = 0; e < E; ++e) c[e] = bias ? s[e] + bias[e] : s[e];
    for (int64_t g = 0; g < G; ++g) {
      float t1 = -INFINITY, t2 = -INFINITY;
      for (int64_t i = 0; i < EG; ++i) {
        const float v = c[g * EG + i];
        if (v > t1) { t2 = t1; t1 = v; } else if (v > t2) t2 = v;
      }
      gs[g] = bias ? t1 + t2 : t1;
    }
    for (int64_t g = 0; g < G; ++g) {   /* rank of group g; groups outside the best topk_group lose their experts */
      int64_t rank = 0;
      for (int64_t h = 0; h < G; ++h) rank += (gs[h] > gs[g]) || (gs[h] == gs[g] && h < g);
      if (rank >= topk_group)
        for (int64_t i = 0; i < EG; ++i) c[g * EG + i] = -INFINITY;
    }
    float wsum = 0.0f;
    for (int64_t k = 0; k < topk; ++k) {
      int64_t best = 0;
      for (int64_t e = 1; e < E; ++e)
        if (c[e] > c[best]) best = e;
      out_w[t * topk + k] = s[best];
      out_id[t * topk + k] = (int32_t)best;
      wsum += s[best];
      c[best] = -INFINITY;
    }
    const float f = renormalize ? route_scale / wsum : route_scale;
    for (int64_t k = 0; k < topk; ++k) out_w[t * topk + k] = out_w[t * topk + k] * f;
  }
  free(s); free(c); free(gs);
}
