// attention_mla.hip -- MLA (DeepSeek latent attention) paged decode for gfx950.
//
// Reference semantics: flash_mla::dense_decode (xllm/core/kernels/dcu/flash_mla_adapter.cpp:104-160, closed
// flash_mla.so) as used by DeepseekV2AttentionImpl::decode_flash_mla (layers/dcu/deepseek_v2_attention.cpp:189-210)
// and specified by its torch twin prefill_sdpa (:212-262): the cache holds ONE latent row per token
// [c_kv_normed (512) || rope(k_pe) (64)] (kv_cache_shape.cpp:354-364, block 64); the absorbed query is
// [q_nope*W_kc (512) || rope(q_pe) (64)]; scores run over all 576 dims, values are the first 512 dims of the
// same row:  out[b,h,:] = softmax(scale * q[b,h,:] . K[:, :]) @ K[:, :512].
//
// HBM-bound like GQA decode (every head of the rank shares the one latent row: ~30 flop/B at 16 heads), but the
// row is 1152 B and the accumulator 16 x 512 fp32, so the tile is shared by the workgroup: K tiles of 32 tokens
// are staged global -> registers -> LDS once (two buffers, next tile in flight during the MFMAs); each of the
// 4 waves computes the full 16-head score block (S^T = K Q^T, 36 MFMAs) and softmax redundantly, then owns a
// 128-wide slice of the 512 output dims (O^T = V^T P^T through ds_read_b64_tr_b16 on the SAME LDS tile).
#include <stdlib.h>

#include "common.h"

namespace xm {

typedef __bf16 mbf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 mbf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 mf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 mf16x4_t __attribute__((ext_vector_type(4)));
typedef float mf32x4_t __attribute__((ext_vector_type(4)));

template <typename T>
struct MlaTraits;
template <>
struct MlaTraits<bf16_t> {
  using x8 = mbf16x8_t;
  using x4 = mbf16x4_t;
  using elem = __bf16;
  static __device__ __forceinline__ mf32x4_t mfma(x8 a, x8 b, mf32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ x4 tr_read(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) x4*)p);
  }
};
template <>
struct MlaTraits<f16_t> {
  using x8 = mf16x8_t;
  using x4 = mf16x4_t;
  using elem = _Float16;
  static __device__ __forceinline__ mf32x4_t mfma(x8 a, x8 b, mf32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ x4 tr_read(const void* p) {
    typedef __fp16 hfp16x4 __attribute__((__vector_size__(8)));
    hfp16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hfp16x4*)p);
    x4 o;
    __builtin_memcpy(&o, &r, 8);
    return o;
  }
};

constexpr int kMlaD = 576, kMlaDV = 512, kMlaTile = 64;
constexpr float kMlaNegBig = -1e30f;

// One workgroup (4 waves) per (entry, block of 16 heads, split-KV slice); K/V = the latent rows themselves (V = the
// first 512 dims). Tile = 64 tokens staged global -> registers -> ONE LDS buffer (rows padded by 16 B); the next tile
// is in flight in registers (73 KiB per workgroup, two workgroups per CU) while this one is computed.
//   * wave w computes S^T = K Q^T only for tokens 16w..16w+15 of the tile (18 MFMAs); the first version had every wave
//     recompute all of S (72 % of its MFMAs and fragment reads were redundant: profiles/r01_mla_decode.txt);
//   * the per-head maxima of the four token quarters meet in LDS (256 B), every wave turns its 16 tokens into P = hi + lo
//     16-bit parts and publishes them in LDS (4.5 KiB), then accumulates its own 128-wide slice of the 512 value dims
//     over all 64 tokens: O^T += V^T P^T with V^T from transposed LDS reads (ds_read_b64_tr_b16);
//   * l is kept per lane over the lane's own tokens and reduced across lanes / waves once at the end (all waves apply
//     the same running maximum, so the partial sums stay consistent).
// UNIFORM: block_size % 64 == 0, a tile lives in one page -> one scalar page id per tile, fetched a tile ahead
// (otherwise every lane loads its page id: a dependent vector load in front of every row load).
template <typename T, bool UNIFORM>
__global__ __launch_bounds__(256, 2) void mla_decode_kernel(
    const T* __restrict__ q, const T* __restrict__ kc, T* __restrict__ out, float* __restrict__ part_o,
    float* __restrict__ part_ml, const int32_t* __restrict__ seqlens, const int32_t* __restrict__ block_table,
    int max_blocks, int n_heads, int block_size, float scale_log2, int nsplit,
    const int32_t* __restrict__ q_seq, const int32_t* __restrict__ q_kvlen) {
  using TR = MlaTraits<T>;
  using x8 = typename TR::x8;
  using x4 = typename TR::x4;
  using elem = typename TR::elem;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int KK = kMlaD / 32;                    // 18
  constexpr int CH = kMlaD * 2 / 16;                // 72 chunks of 16 B per row
  constexpr int RS = kMlaD * 2 + 16;                // padded LDS row stride
  constexpr int NLD = kMlaTile * CH / 256;          // 18 chunks per thread per tile
  static_assert(NLD == 18 && CH == 72, "the staging map below is written for 64 x 72 chunks");
  constexpr int DBW = (kMlaDV / 4) / 16;            // 8 output blocks of 16 dims per wave
  constexpr int PS = kMlaTile * 2 + 16;             // P row stride: 64 tokens of 16 bits + pad
  __shared__ __attribute__((aligned(16))) char lds[kMlaTile * RS];
  __shared__ __attribute__((aligned(16))) char p_lds[2][16 * PS];  // [hi / lo][head][token]
  __shared__ float xmax[4][16], xl[4][16];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p16 = lane & 15, g = lane >> 4;
  const int split = blockIdx.x % nsplit;
  const int hb = (blockIdx.x / nsplit) % ((n_heads + 15) / 16);  // block of 16 heads
  const int b = blockIdx.x / nsplit / ((n_heads + 15) / 16);
  // prefill / chunked prefill drive the same kernel with one "batch entry" per QUERY TOKEN: q_seq maps the token to
  // its sequence (block-table row) and q_kvlen is its causal key count; decode passes null (entry = sequence)
  const int seq = q_seq ? q_seq[b] : b;
  const int kv_len = q_kvlen ? q_kvlen[b] : seqlens[seq];
  const int ntiles = (kv_len + kMlaTile - 1) / kMlaTile;
  const int per = (ntiles + nsplit - 1) / nsplit;
  const int tile_lo = split * per;
  const int tile_hi = tile_lo + per < ntiles ? tile_lo + per : ntiles;
  const int32_t* bt_row = block_table + (int64_t)seq * max_blocks;
  const int head = hb * 16 + p16;

  // Q as the MFMA B operand: lane (n = head p16, k group g)
  x8 qf[KK];
  {
    const T* qp = q + ((int64_t)b * n_heads + (head < n_heads ? head : 0)) * kMlaD;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (head < n_heads) qf[kk] = *reinterpret_cast<const x8*>(qp + (kk * 4 + g) * 8);
      else qf[kk] = x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  mf32x4_t acc_o[DBW];
#pragma unroll
  for (int i = 0; i < DBW; ++i) acc_o[i] = mf32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = kMlaNegBig, l_run = 0.0f;  // l_run: this lane's tokens only

  u32x4 rk[NLD];  // native vectors: plain 16-byte loads / stores, never spilled through memcpy
  auto page_of = [&](int tile) -> int {  // UNIFORM: scalar page id of a tile (clamped into the live range)
    int t = tile < tile_hi ? tile : tile_hi - 1;
    t = t < 0 ? 0 : t;
    return bt_row[(t * kMlaTile) / block_size];
  };
  // thread (r = tid >> 3, c8 = tid & 7) moves chunks c8 + 8 j (j < 9) of rows r and r + 32: 128 contiguous bytes per
  // 8 lanes, two row addresses per thread and immediates for the rest. Every load is unconditional (rows past kv_len
  // re-load the last row): no branch around a load
  const int ld_row = tid >> 3, ld_col = (tid & 7) * 16;
  auto load_global = [&](int tile, int page) {
    const int t0 = tile * kMlaTile;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int tok = t0 + ld_row + h * 32;
      tok = tok < kv_len ? tok : kv_len - 1;
      int64_t rowi;
      if constexpr (UNIFORM) rowi = (int64_t)page * block_size + tok % block_size;
      else rowi = (int64_t)bt_row[tok / block_size] * block_size + tok % block_size;
      const char* src = reinterpret_cast<const char*>(kc + rowi * kMlaD) + ld_col;
#pragma unroll
      for (int j = 0; j < 9; ++j) rk[h * 9 + j] = *reinterpret_cast<const u32x4*>(src + j * 128);
    }
  };
  auto write_lds = [&](int tile) {
    const int t0 = tile * kMlaTile;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool dead = t0 + ld_row + h * 32 >= kv_len;  // rows past kv_len: zero (they double as V)
      char* dst = lds + (ld_row + h * 32) * RS + ld_col;
#pragma unroll
      for (int j = 0; j < 9; ++j)
        *reinterpret_cast<u32x4*>(dst + j * 128) = dead ? u32x4{0u, 0u, 0u, 0u} : rk[h * 9 + j];
    }
  };

  if (tile_lo < tile_hi) {
    load_global(tile_lo, UNIFORM ? page_of(tile_lo) : 0);
    int page_next = UNIFORM ? page_of(tile_lo + 1) : 0;
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
      __syncthreads();                      // A: every wave has finished reading the previous tile
      write_lds(tile);
      __syncthreads();                      // B: the tile is visible
      if (tile + 1 < tile_hi) load_global(tile + 1, page_next);   // in flight underneath this tile's work
      if constexpr (UNIFORM) page_next = page_of(tile + 2);
      const int t0 = tile * kMlaTile;
      // ---- S^T for this wave's token quarter: lane holds S[head p16][token 16*wave + 4g + r]
      mf32x4_t s4 = mf32x4_t{0.f, 0.f, 0.f, 0.f};
      const char* krow = lds + (wave * 16 + p16) * RS + g * 16;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) s4 = TR::mfma(*reinterpret_cast<const x8*>(krow + kk * 64), qf[kk], s4);
      const bool partial = t0 + kMlaTile > kv_len;
      float mx = kMlaNegBig;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = s4[r] * scale_log2;
        if (partial && t0 + wave * 16 + g * 4 + r >= kv_len) v = -INFINITY;
        s4[r] = v;
        mx = fmaxf(mx, v);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (g == 0) xmax[wave][p16] = mx;
      __syncthreads();                      // C: the four quarter maxima of every head are published
      const float m_new = fmaxf(fmaxf(m_run, fmaxf(xmax[0][p16], xmax[1][p16])), fmaxf(xmax[2][p16], xmax[3][p16]));
      const float alpha = exp2f(m_run - m_new);
      m_run = m_new;
      float psum = 0.0f;
      elem ph[4], pl4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = exp2f(s4[r] - m_new);
        psum += p;
        ph[r] = (elem)p;                     // P = hi + lo 16-bit parts (see attention_decode.hip)
        pl4[r] = (elem)(p - (float)ph[r]);
      }
      l_run = l_run * alpha + psum;
      {
        x4 hv = {ph[0], ph[1], ph[2], ph[3]}, lv = {pl4[0], pl4[1], pl4[2], pl4[3]};
        *reinterpret_cast<x4*>(&p_lds[0][p16 * PS + (wave * 16 + g * 4) * 2]) = hv;
        *reinterpret_cast<x4*>(&p_lds[1][p16 * PS + (wave * 16 + g * 4) * 2]) = lv;
      }
#pragma unroll
      for (int i = 0; i < DBW; ++i) acc_o[i] *= alpha;
      __syncthreads();                      // D: P of all 64 tokens is published
      // ---- O^T += V^T P^T over both 32-token halves; k slot (g, j): j < 4 -> token 4g + j, j >= 4 -> token 16 + 4g + j - 4
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        x8 pf, pl;
        {
          const x4 h0 = *reinterpret_cast<const x4*>(&p_lds[0][p16 * PS + (ks * 32 + g * 4) * 2]);
          const x4 h1 = *reinterpret_cast<const x4*>(&p_lds[0][p16 * PS + (ks * 32 + 16 + g * 4) * 2]);
          const x4 l0 = *reinterpret_cast<const x4*>(&p_lds[1][p16 * PS + (ks * 32 + g * 4) * 2]);
          const x4 l1 = *reinterpret_cast<const x4*>(&p_lds[1][p16 * PS + (ks * 32 + 16 + g * 4) * 2]);
          pf = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
          pl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        const char* trb = lds + (ks * 32 + 4 * g + (p16 >> 2)) * RS + (p16 & 3) * 8 + wave * 256;
#pragma unroll
        for (int db = 0; db < DBW; ++db) {
          const x4 lo = TR::tr_read(trb + db * 32);
          const x4 hi = TR::tr_read(trb + 16 * RS + db * 32);
          const x8 vt = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          acc_o[db] = TR::mfma(vt, pf, acc_o[db]);
          acc_o[db] = TR::mfma(vt, pl, acc_o[db]);
        }
      }
    }
  }

  // l over all tokens of the slice: lanes of a head (g), then the four waves
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  __syncthreads();
  if (g == 0) xl[wave][p16] = l_run;
  __syncthreads();
  l_run = (xl[0][p16] + xl[1][p16]) + (xl[2][p16] + xl[3][p16]);
  if (head >= n_heads) return;
  if (nsplit == 1) {
    const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
    T* op = out + ((int64_t)b * n_heads + head) * kMlaDV + wave * 128;
#pragma unroll
    for (int db = 0; db < DBW; ++db) {
      uint16_t hv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T t = from_f32<T>(acc_o[db][r] * inv);
        __builtin_memcpy(&hv[r], &t, 2);
      }
      *reinterpret_cast<uint2*>(op + db * 16 + g * 4) =
          make_uint2((uint32_t)hv[0] | ((uint32_t)hv[1] << 16), (uint32_t)hv[2] | ((uint32_t)hv[3] << 16));
    }
  } else {
    const int64_t pi = ((int64_t)b * n_heads + head) * nsplit + split;
    float* po = part_o + pi * kMlaDV + wave * 128;
#pragma unroll
    for (int db = 0; db < DBW; ++db) *reinterpret_cast<mf32x4_t*>(po + db * 16 + g * 4) = acc_o[db];
    if (wave == 0 && g == 0) { part_ml[pi * 2] = m_run; part_ml[pi * 2 + 1] = l_run; }
  }
}

// LDS reads of the DMA-filled buffers are inline asm: the compiler fences every LDS load it emits itself against ALL
// outstanding LDS-DMA (s_waitcnt vmcnt(0)), which would serialise the prefetch of the next tiles with this tile's reads.
// Waits are counted (LDS returns in order; ops the compiler adds in between only make a counted wait more conservative).
typedef unsigned mu32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned mu32x2_t __attribute__((ext_vector_type(2)));
#define MLA_DSR128(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define MLA_DSR64(DST, ADDR, OFF) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define MLA_DSR64TR(DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define MLA_LGKM1(N, A) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(A) : "n"(N))
#define MLA_LGKM2(N, A, B) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(A), "+v"(B) : "n"(N))
#define MLA_LGKM4(N, A, B, C, D) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "n"(N))

// The same algorithm for block_size % 64 == 0 with the tile staged by LDS-DMA (buffer_load ... lds) into TWO LDS buffers:
// no staging registers, no register -> LDS copy, and two tiles (147 KB per CU) in flight while a tile is computed; one
// workgroup per CU. profiles/r01_mla_decode.txt has the measurements that led here.
//   * a tile's rows are consecutive rows of ONE page, so the DMA source is (scalar page base) + (per-lane constant
//     offset): 18 offsets per lane computed once; the buffer's num_records = live rows * 1152 zero-fills the rows past
//     kv_len in hardware (they double as V) and turns the prefetch of a tile past the slice into a no-op;
//   * every DMA instruction fills 1 KB of contiguous LDS, so rows are unpadded (1152 B) and bank conflicts are avoided by
//     an XOR swizzle of the 16-byte chunks inside aligned groups of 8: physical = logical ^ f(row),
//     f = row bit1 -> bit1, bit2 -> bit2, bit3 -> bit0 -- distinct over the 16 consecutive rows a K fragment read touches
//     AND over the 8 rows x 2 chunks a transposed V read touches;
//   * wave w DMAs rows 16w..16w+15 -- exactly the rows it scores, so QK^T needs only the wave's own vmcnt, no barrier.
//   * NT (round 4): a pure decode launch with one head block per sequence reads every latent byte ONCE -> non-temporal DMA
//     (cfg4: 211 -> 193-197 us = 0.716 -> 0.77-0.78 of 8 TB/s). With several head blocks per sequence (128 heads: 8 workgroups
//     share a sequence's tiles through the L2) or several query tokens per sequence (the per-token prefill form) the L2 re-use
//     is the point: nt costs 16 % there (1248 -> 1445 us), so those launches keep ordinary loads.
template <typename T, bool NT>
__global__ __launch_bounds__(256, 1) void mla_decode_dma_kernel(
    const T* __restrict__ q, const T* __restrict__ kc, T* __restrict__ out, float* __restrict__ part_o,
    float* __restrict__ part_ml, const int32_t* __restrict__ seqlens, const int32_t* __restrict__ block_table,
    int max_blocks, int n_heads, int block_size, float scale_log2, int nsplit,
    const int32_t* __restrict__ q_seq, const int32_t* __restrict__ q_kvlen) {
  using TR = MlaTraits<T>;
  using x8 = typename TR::x8;
  using x4 = typename TR::x4;
  using elem = typename TR::elem;
  constexpr int KK = kMlaD / 32;                    // 18
  constexpr int ROWB = kMlaD * 2;                   // 1152 bytes per row, unpadded
  constexpr int BUFB = kMlaTile * ROWB;             // 73,728 bytes per tile buffer
  constexpr int NDMA = BUFB / 1024 / 4;             // 18 DMA instructions (1 KB each) per wave per tile
  constexpr int DBW = (kMlaDV / 4) / 16;            // 8 output blocks of 16 dims per wave
  constexpr int PS = kMlaTile * 2 + 16;             // P row stride: 64 tokens of 16 bits + pad
  static_assert(NDMA == 18 && ROWB % 128 == 0, "tile = 64 rows of 9 x 128 bytes");
  __shared__ __attribute__((aligned(1024))) char lds[2 * BUFB];
  __shared__ __attribute__((aligned(16))) char p_lds[2][16 * PS];  // [hi / lo][head][token]
  __shared__ float xmax[4][16], xl[4][16];
  typedef __attribute__((address_space(3))) char* lds_ptr_t;
  const lds_ptr_t lds3 = (lds_ptr_t)lds;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p16 = lane & 15, g = lane >> 4;
  const int split = blockIdx.x % nsplit;
  const int hb = (blockIdx.x / nsplit) % ((n_heads + 15) / 16);
  const int b = blockIdx.x / nsplit / ((n_heads + 15) / 16);
  const int seq = q_seq ? q_seq[b] : b;
  const int kv_len = q_kvlen ? q_kvlen[b] : seqlens[seq];
  const int ntiles = (kv_len + kMlaTile - 1) / kMlaTile;
  const int per = (ntiles + nsplit - 1) / nsplit;
  const int tile_lo = split * per;
  const int tile_hi = tile_lo + per < ntiles ? tile_lo + per : ntiles;
  const int32_t* bt_row = block_table + (int64_t)seq * max_blocks;
  const int head = hb * 16 + p16;

  x8 qf[KK];
  {
    const T* qp = q + ((int64_t)b * n_heads + (head < n_heads ? head : 0)) * kMlaD;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (head < n_heads) qf[kk] = *reinterpret_cast<const x8*>(qp + (kk * 4 + g) * 8);
      else qf[kk] = x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  mf32x4_t acc_o[DBW];
#pragma unroll
  for (int i = 0; i < DBW; ++i) acc_o[i] = mf32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = kMlaNegBig, l_run = 0.0f;

  // DMA source offsets: instruction i of wave w fills LDS bytes [(18 w + i) KB, +1 KB) of the tile buffer; lane's 16
  // bytes are chunk C = (18 w + i) * 64 + lane = (row, physical chunk) and come from logical chunk (physical ^ f(row))
  int voff[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int c = (wave * NDMA + i) * 64 + lane, row = c / 72, pc = c % 72;
    const int f = (row & 6) | ((row >> 3) & 1);
    voff[i] = row * ROWB + ((pc ^ f) << 4);
  }
  auto stage = [&](int tile, int buf) {  // unconditional: a tile past the slice gets num_records = 0 (no traffic)
    int rows = 0;
    int64_t row0 = 0;
    if (tile < tile_hi) {
      const int t0 = tile * kMlaTile;
      rows = kv_len - t0 < kMlaTile ? kv_len - t0 : kMlaTile;
      row0 = (int64_t)bt_row[t0 / block_size] * block_size + t0 % block_size;
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<T*>(kc + row0 * kMlaD), 0, rows * ROWB, 0x00020000);
    const lds_ptr_t dst = lds3 + buf * BUFB + wave * (NDMA * 1024);
#pragma unroll
    for (int i = 0; i < NDMA; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst + i * 1024, 16, voff[i], 0, 0, NT ? 2 : 0);   // aux 2 = nt
  };

  // fragment read offsets inside a tile buffer (per lane, constant over tiles)
  const int fq = (p16 & 6) | ((p16 >> 3) & 1);                   // f(row) of K row 16 w + p16
  const int k_off = (wave * 16 + p16) * ROWB + ((g ^ fq) << 4);  // + (kk >> 1) * 128, ^ 64 for odd kk
  // V^T (transposed) reads: row 32 ks + 16 hi + 4 g + (p16 >> 2); f of that row depends on the lane only
  const int ft = (((p16 >> 3) & 1) << 1) | ((g & 1) << 2) | ((g >> 1) & 1);
  const int v_x = ((((p16 >> 1) & 1) ^ ft) << 4) | ((p16 & 1) << 3);  // chunk (b ^ ft) and the 8-byte half
  const int v_row = (4 * g + (p16 >> 2)) * ROWB + wave * 256;
  const unsigned lds_base = (unsigned)(__UINTPTR_TYPE__)lds3;
  const unsigned p_rd = (unsigned)(__UINTPTR_TYPE__)(lds_ptr_t)p_lds + p16 * PS + g * 8;  // + ks * 64 (+ 32): tokens 32 ks (+ 16) + 4 g

  if (tile_lo < tile_hi) {
    stage(tile_lo, 0);
    stage(tile_lo + 1, 1);
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
      const int buf = (tile - tile_lo) & 1;
      asm volatile("s_waitcnt vmcnt(18)" ::: "memory");  // this wave's rows of `tile` have landed (tile + 1 still flies)
      const int t0 = tile * kMlaTile;
      mf32x4_t sa = mf32x4_t{0.f, 0.f, 0.f, 0.f}, sb = sa;
      {
        const unsigned ke = lds_base + buf * BUFB + k_off, ko = lds_base + buf * BUFB + (k_off ^ 64);
        mu32x4_t kf[KK];
#define MLA_QK_RD(K_) MLA_DSR128(kf[K_], ((K_) & 1) ? ko : ke, ((K_) >> 1) * 128);
#define MLA_QK_MM(K_)                                                             \
  MLA_LGKM1((17 - (K_)) > 15 ? 15 : 17 - (K_), kf[K_]);                           \
  if ((K_) & 1) sb = TR::mfma(__builtin_bit_cast(x8, kf[K_]), qf[K_], sb);        \
  else sa = TR::mfma(__builtin_bit_cast(x8, kf[K_]), qf[K_], sa);
        MLA_QK_RD(0) MLA_QK_RD(1) MLA_QK_RD(2) MLA_QK_RD(3) MLA_QK_RD(4) MLA_QK_RD(5) MLA_QK_RD(6) MLA_QK_RD(7) MLA_QK_RD(8)
        MLA_QK_RD(9) MLA_QK_RD(10) MLA_QK_RD(11) MLA_QK_RD(12) MLA_QK_RD(13) MLA_QK_RD(14) MLA_QK_RD(15) MLA_QK_RD(16)
        MLA_QK_RD(17)
        MLA_QK_MM(0) MLA_QK_MM(1) MLA_QK_MM(2) MLA_QK_MM(3) MLA_QK_MM(4) MLA_QK_MM(5) MLA_QK_MM(6) MLA_QK_MM(7) MLA_QK_MM(8)
        MLA_QK_MM(9) MLA_QK_MM(10) MLA_QK_MM(11) MLA_QK_MM(12) MLA_QK_MM(13) MLA_QK_MM(14) MLA_QK_MM(15) MLA_QK_MM(16)
        MLA_QK_MM(17)
#undef MLA_QK_RD
#undef MLA_QK_MM
      }
      const bool partial = t0 + kMlaTile > kv_len;
      float sv[4], mx = kMlaNegBig;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = (sa[r] + sb[r]) * scale_log2;
        if (partial && t0 + wave * 16 + g * 4 + r >= kv_len) v = -INFINITY;
        sv[r] = v;
        mx = fmaxf(mx, v);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      if (g == 0) xmax[wave][p16] = mx;
      __syncthreads();                      // C: quarter maxima published; every wave's rows of the tile have landed
      const float m_new = fmaxf(fmaxf(m_run, fmaxf(xmax[0][p16], xmax[1][p16])), fmaxf(xmax[2][p16], xmax[3][p16]));
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      float psum = 0.0f;
      elem ph[4], pl4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(sv[r] - m_new);
        psum += p;
        ph[r] = (elem)p;
        pl4[r] = (elem)(p - (float)ph[r]);
      }
      l_run = l_run * alpha + psum;
      {
        x4 hv = {ph[0], ph[1], ph[2], ph[3]}, lv = {pl4[0], pl4[1], pl4[2], pl4[3]};
        *reinterpret_cast<x4*>(&p_lds[0][p16 * PS + (wave * 16 + g * 4) * 2]) = hv;
        *reinterpret_cast<x4*>(&p_lds[1][p16 * PS + (wave * 16 + g * 4) * 2]) = lv;
      }
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int i = 0; i < DBW; ++i) acc_o[i] *= alpha;
      }
      __syncthreads();                      // D: P of all 64 tokens is published
      const unsigned vb = lds_base + buf * BUFB + v_row;
      unsigned va[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) va[j] = vb + (v_x ^ (j << 5));
#define MLA_PV_RD(KS_, DB_)                                                                       \
  MLA_DSR64TR(vt[DB_][0], va[(DB_) & 3], (KS_) * 32 * ROWB + ((DB_) >> 2) * 128);                  \
  MLA_DSR64TR(vt[DB_][1], va[(DB_) & 3], (KS_) * 32 * ROWB + ((DB_) >> 2) * 128 + 16 * ROWB);
#define MLA_PV_MM(DB_)                                                                             \
  {                                                                                                \
    MLA_LGKM2(2 * (DBW - 1 - (DB_)), vt[DB_][0], vt[DB_][1]);                                      \
    const x8 v8 = __builtin_shufflevector(__builtin_bit_cast(x4, vt[DB_][0]), __builtin_bit_cast(x4, vt[DB_][1]), \
                                          0, 1, 2, 3, 4, 5, 6, 7);                                 \
    acc_o[DB_] = TR::mfma(v8, pf, acc_o[DB_]);                                                     \
    acc_o[DB_] = TR::mfma(v8, pl, acc_o[DB_]);                                                     \
  }
#define MLA_PV_KS(KS_)                                                                             \
  {                                                                                                \
    mu32x2_t ph2[2], pl2[2], vt[DBW][2];                                                           \
    MLA_DSR64(ph2[0], p_rd, (KS_) * 64);                                                           \
    MLA_DSR64(ph2[1], p_rd, (KS_) * 64 + 32);                                                      \
    MLA_DSR64(pl2[0], p_rd, 16 * PS + (KS_) * 64);                                                 \
    MLA_DSR64(pl2[1], p_rd, 16 * PS + (KS_) * 64 + 32);                                            \
    MLA_PV_RD(KS_, 0) MLA_PV_RD(KS_, 1) MLA_PV_RD(KS_, 2) MLA_PV_RD(KS_, 3)                        \
    MLA_PV_RD(KS_, 4) MLA_PV_RD(KS_, 5) MLA_PV_RD(KS_, 6) MLA_PV_RD(KS_, 7)                        \
    MLA_LGKM4(15, ph2[0], ph2[1], pl2[0], pl2[1]);                                                 \
    const x8 pf = __builtin_shufflevector(__builtin_bit_cast(x4, ph2[0]), __builtin_bit_cast(x4, ph2[1]), 0, 1, 2, 3, 4, 5, 6, 7); \
    const x8 pl = __builtin_shufflevector(__builtin_bit_cast(x4, pl2[0]), __builtin_bit_cast(x4, pl2[1]), 0, 1, 2, 3, 4, 5, 6, 7); \
    MLA_PV_MM(0) MLA_PV_MM(1) MLA_PV_MM(2) MLA_PV_MM(3) MLA_PV_MM(4) MLA_PV_MM(5) MLA_PV_MM(6) MLA_PV_MM(7)                        \
  }
      MLA_PV_KS(0)
      MLA_PV_KS(1)
#undef MLA_PV_RD
#undef MLA_PV_MM
#undef MLA_PV_KS
      __syncthreads();                      // E: every wave is done with this buffer (and with P)
      stage(tile + 2, buf);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two trailing (empty) prefetches
  }

  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  if (g == 0) xl[wave][p16] = l_run;
  __syncthreads();
  l_run = (xl[0][p16] + xl[1][p16]) + (xl[2][p16] + xl[3][p16]);
  if (head >= n_heads) return;
  if (nsplit == 1) {
    const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
    T* op = out + ((int64_t)b * n_heads + head) * kMlaDV + wave * 128;
#pragma unroll
    for (int db = 0; db < DBW; ++db) {
      uint16_t hv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T t = from_f32<T>(acc_o[db][r] * inv);
        __builtin_memcpy(&hv[r], &t, 2);
      }
      *reinterpret_cast<uint2*>(op + db * 16 + g * 4) =
          make_uint2((uint32_t)hv[0] | ((uint32_t)hv[1] << 16), (uint32_t)hv[2] | ((uint32_t)hv[3] << 16));
    }
  } else {
    const int64_t pi = ((int64_t)b * n_heads + head) * nsplit + split;
    float* po = part_o + pi * kMlaDV + wave * 128;
#pragma unroll
    for (int db = 0; db < DBW; ++db) *reinterpret_cast<mf32x4_t*>(po + db * 16 + g * 4) = acc_o[db];
    if (wave == 0 && g == 0) { part_ml[pi * 2] = m_run; part_ml[pi * 2 + 1] = l_run; }
  }
}

__device__ __forceinline__ float mla_as_f32(unsigned v) { return __builtin_bit_cast(float, v); }

// acc *= alpha on an accumulator that lives in AGPRs, inside the (rare) rescale branch. Written as asm because the
// compiler's own AGPR -> VGPR copies for a plain `acc *= alpha` are hoisted OUT of the branch (128 v_accvgpr_read per tile
// on the hot path). The MFMAs that last wrote / next read the accumulator are a whole QK^T phase / a P conversion away.
__device__ __forceinline__ void mla_scale_acc(mf32x4_t& a, float alpha) {
#ifdef XM_MLA_PLAIN_RESCALE
  a *= alpha;
#else
  float x0 = a[0], x1 = a[1], x2 = a[2], x3 = a[3], t0, t1, t2, t3;
  asm volatile(
      "v_accvgpr_read_b32 %4, %0\n\tv_accvgpr_read_b32 %5, %1\n\tv_accvgpr_read_b32 %6, %2\n\tv_accvgpr_read_b32 %7, %3\n\t"
      "v_mul_f32 %4, %8, %4\n\tv_mul_f32 %5, %8, %5\n\tv_mul_f32 %6, %8, %6\n\tv_mul_f32 %7, %8, %7\n\t"
      "v_accvgpr_write_b32 %0, %4\n\tv_accvgpr_write_b32 %1, %5\n\tv_accvgpr_write_b32 %2, %6\n\tv_accvgpr_write_b32 %3, %7"
      : "+a"(x0), "+a"(x1), "+a"(x2), "+a"(x3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(alpha));
  a = mf32x4_t{x0, x1, x2, x3};
#endif
}
// MLA prefill / chunked prefill with the KV tile SHARED by four query tokens (block_size % 64 == 0).
// The per-token decode kernels above re-read a sequence's whole latent cache once per query token; here a workgroup takes
// four consecutive query tokens (flat index) x 16 heads, stages each 64-token tile ONCE by LDS-DMA (the same two 72 KB
// buffers, swizzle and zero-fill as mla_decode_dma_kernel) and wave w works on query token 4 G + w alone:
//   * S^T = K Q_w^T over all 64 tokens of the tile (72 MFMAs, K fragments rolling through 8 register slots);
//   * softmax stays inside the wave: a lane holds 16 scores of one head, the maximum needs two permlane swaps, and the
//     hi/lo 16-bit P operands of the PV MFMAs are exactly the lane's own registers (the K order of ds_read_b64_tr_b16
//     matches the score layout), so there is no P exchange through LDS and no cross-wave reduction;
//   * O_w^T = V^T P_w^T over all 512 value dims (128 MFMAs, accumulator 16 heads x 512 in 128 registers);
//   * causal masks differ per wave (q_kvlen[token]); the tile loop runs to the longest of the four, rows past the
//     sequence are zero-filled by the DMA bounds check, rows between a wave's own limit and the longest get p = 0.
// Tokens of a group that belong to different sequences are handled in passes (one per distinct sequence, the other
// waves' scores fully masked), so no host- or device-built group table is needed: q_seq / q_kvlen are the per-token
// arrays mla_expand_queries_kernel already writes. Two barriers per tile (tile landed / buffer free); MFMA-bound:
// 200 MFMAs per wave and tile against 544 KB of LDS reads per workgroup and tile (2176 of ~3200 cycles).
// P1: ONE RNE-rounded 16-bit P per score in front of PV, as the reference's prefill computes it (prefill_sdpa, layers/dcu/
// deepseek_v2_attention.cpp:212-262: torch SDPA rounds the probabilities to the tensor dtype) -- half the PV MFMAs of the
// hi + lo form, which remains selectable (XLLM_MI355_MLA_PREFILL_P=2)
template <typename T, bool P1>
__global__ __launch_bounds__(256, 1) void mla_prefill_dma_kernel(
    const T* __restrict__ q, const T* __restrict__ kc, T* __restrict__ out, const int32_t* __restrict__ block_table,
    int max_blocks, int n_heads, int block_size, float scale_log2, const int32_t* __restrict__ q_seq,
    const int32_t* __restrict__ q_kvlen, int n_tokens, int n_groups) {
  using TR = MlaTraits<T>;
  using x8 = typename TR::x8;
  using x4 = typename TR::x4;
  using elem = typename TR::elem;
  constexpr int KK = kMlaD / 32;                    // 18
  constexpr int ROWB = kMlaD * 2;                   // 1152 bytes per row, unpadded
  constexpr int BUFB = kMlaTile * ROWB;             // 73,728 bytes per tile buffer
  constexpr int NDMA = BUFB / 1024 / 4;             // 18 DMA instructions (1 KB each) per wave per tile
  constexpr int DB = kMlaDV / 16;                   // 32 output blocks of 16 dims
  __shared__ __attribute__((aligned(1024))) char lds[2 * BUFB];
  typedef __attribute__((address_space(3))) char* lds_ptr_t;
  const lds_ptr_t lds3 = (lds_ptr_t)lds;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p16 = lane & 15, g = lane >> 4;
  // workgroups i, i + 8, i + 16, ... run on the same XCD: give them the head blocks of one token group (shared L2 lines),
  // groups in descending order (under a causal mask the late tokens are the long ones)
  const int n_hb = (n_heads + 15) / 16;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int hb = slot % n_hb;
  const int grp = n_groups - 1 - ((slot / n_hb) * 8 + xcd);
  if (grp < 0) return;
  const int tok0 = grp * 4;
  const int my_tok = tok0 + wave;
  const bool live = my_tok < n_tokens;
  const int my_seq = live ? q_seq[my_tok] : -1;
  const int my_kv = live ? q_kvlen[my_tok] : 0;
  const int head = hb * 16 + p16;

  x8 qf[KK];
  {
    const T* qp = q + ((int64_t)(live ? my_tok : tok0) * n_heads + (head < n_heads ? head : 0)) * kMlaD;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (head < n_heads && live) qf[kk] = *reinterpret_cast<const x8*>(qp + (kk * 4 + g) * 8);
      else qf[kk] = x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  mf32x4_t acc_o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) acc_o[i] = mf32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = kMlaNegBig, l_run = 0.0f;

  int voff[NDMA];
#ifdef XM_MLA_INTERLEAVE
  // 8-row-interleaved tile [row >> 3][16-B chunk (72)][row & 7][16 B] (9216 B per 8-row group): the K fragment reads
  // (lane = (row, chunk)) then have 8 consecutive lanes inside one aligned 128-B block = full ds_read_b128 rate
  // (tools/lds_pattern_bench.hip). DMA instruction n = 18 w + i fills LDS KB n = group n / 9, chunks 8 (n % 9) .. + 7:
  // lane l fetches row 8 (n / 9) + (l & 7), chunk 8 (n % 9) + (l >> 3) -- a 16-byte gather over 8 rows.
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int n = wave * NDMA + i;
    voff[i] = ((n / 9) * 8 + (lane & 7)) * ROWB + (((n % 9) * 8 + (lane >> 3)) << 4);
  }
  const int k_off = (p16 >> 3) * 9216 + g * 128 + (p16 & 7) * 16;   // + tb * 18432 + kk * 512
  // V^T reads: row 32 ks + 16 hi + 4 g + (p16 >> 2), chunk 2 db + ((p16 >> 1) & 1), 8-byte half p16 & 1
  const int v_row = (g >> 1) * 9216 + (4 * (g & 1) + (p16 >> 2)) * 16 + (p16 & 1) * 8 + ((p16 >> 1) & 1) * 128;
  const int v_x = 0;
#else
  // as in mla_decode_dma_kernel: lane's 16 bytes of DMA instruction i = (row, physical chunk)
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int c = (wave * NDMA + i) * 64 + lane, row = c / 72, pc = c % 72;
    const int f = (row & 6) | ((row >> 3) & 1);
    voff[i] = row * ROWB + ((pc ^ f) << 4);
  }
  const int fq = (p16 & 6) | ((p16 >> 3) & 1);          // f(row) of K row 16 tb + p16 (the same for every tb)
  const int k_off = p16 * ROWB + ((g ^ fq) << 4);       // + tb * 16 rows + (kk >> 1) * 128, ^ 64 for odd kk
  const int ft = (((p16 >> 3) & 1) << 1) | ((g & 1) << 2) | ((g >> 1) & 1);
  const int v_x = ((((p16 >> 1) & 1) ^ ft) << 4) | ((p16 & 1) << 3);
  const int v_row = (4 * g + (p16 >> 2)) * ROWB;
#endif
  const unsigned lds_base = (unsigned)(__UINTPTR_TYPE__)lds3;

  for (int pass = 0; pass < 4; ++pass) {
    const int pt = tok0 + pass;
    if (pt >= n_tokens) break;
    const int seq = q_seq[pt];
    if (pass > 0 && seq == q_seq[pt - 1]) continue;     // tokens are sorted by sequence: one pass per distinct sequence
    int kv_hi = 0;
    for (int j = pass; j < 4 && tok0 + j < n_tokens; ++j)
      if (q_seq[tok0 + j] == seq) kv_hi = q_kvlen[tok0 + j] > kv_hi ? q_kvlen[tok0 + j] : kv_hi;
    const int kv_w = my_seq == seq ? my_kv : 0;          // this wave's own causal limit inside the pass (0: masked out)
    const int tile_hi = (kv_hi + kMlaTile - 1) / kMlaTile;
    const int32_t* bt_row = block_table + (int64_t)seq * max_blocks;
    auto stage = [&](int tile, int buf) {
      int rows = 0;
      int64_t row0 = 0;
      if (tile < tile_hi) {
        const int t0 = tile * kMlaTile;
        rows = kv_hi - t0 < kMlaTile ? kv_hi - t0 : kMlaTile;
        row0 = (int64_t)bt_row[t0 / block_size] * block_size + t0 % block_size;
      }
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<T*>(kc + row0 * kMlaD), 0, rows * ROWB, 0x00020000);
      const lds_ptr_t dst = lds3 + buf * BUFB + wave * (NDMA * 1024);
#pragma unroll
      for (int i = 0; i < NDMA; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst + i * 1024, 16, voff[i], 0, 0, 0);
    };
    if (tile_hi == 0) continue;
    stage(0, 0);
    stage(1, 1);
    for (int tile = 0; tile < tile_hi; ++tile) {
      const int buf = tile & 1;
      asm volatile("s_waitcnt vmcnt(18)" ::: "memory");  // this wave's rows of `tile` have landed (tile + 1 still flies)
      __builtin_amdgcn_s_barrier();                      // ... and so have the other waves' rows
      const int t0 = tile * kMlaTile;
      mf32x4_t s[4];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb) s[tb] = mf32x4_t{0.f, 0.f, 0.f, 0.f};
      {
        const unsigned ke = lds_base + buf * BUFB + k_off, ko = lds_base + buf * BUFB + (k_off ^ 64);
        // item I: kk = I >> 2, token block tb = I & 3 (four independent accumulators back to back); KW fragment reads in flight
        // (one wave per SIMD: the window has to cover the loaded LDS latency at 1 KB per 16-cycle MFMA)
#ifndef XM_MLA_KW
#define XM_MLA_KW 8
#endif
        constexpr int KW = XM_MLA_KW;
        static_assert(KW >= 4 && KW <= 15, "lgkmcnt counts at most 15 outstanding reads");
        mu32x4_t kf[KW];
#ifdef XM_MLA_INTERLEAVE
#define MLP_K_RD(I_) MLA_DSR128(kf[(I_) % KW], ke, ((I_) & 3) * 18432 + ((I_) >> 2) * 512);
#else
#define MLP_K_RD(I_) MLA_DSR128(kf[(I_) % KW], (((I_) >> 2) & 1) ? ko : ke, ((I_) & 3) * 16 * ROWB + ((I_) >> 3) * 128);
#endif
#define MLP_K_IT(I_)                                                                             \
  MLA_LGKM1(((I_) + KW < 72 ? KW - 1 : 71 - (I_)), kf[(I_) % KW]);                               \
  s[(I_) & 3] = TR::mfma(__builtin_bit_cast(x8, kf[(I_) % KW]), qf[(I_) >> 2], s[(I_) & 3]);     \
  if ((I_) + KW < 72) { MLP_K_RD(((I_) + KW < 72 ? (I_) + KW : 71)) }
#define MLP_K_IT8(B_) MLP_K_IT(B_) MLP_K_IT(B_ + 1) MLP_K_IT(B_ + 2) MLP_K_IT(B_ + 3) MLP_K_IT(B_ + 4) MLP_K_IT(B_ + 5) MLP_K_IT(B_ + 6) MLP_K_IT(B_ + 7)
#ifndef XM_ABL_MLP_NOQK  /* ablation builds (timing only, WRONG results): XM_ABL_MLP_NOQK / _NOPV / _NOSM drop a phase */
        MLP_K_RD(0) MLP_K_RD(1) MLP_K_RD(2) MLP_K_RD(3)
        if (KW > 4) { MLP_K_RD(4 < KW ? 4 : 0) }
        if (KW > 5) { MLP_K_RD(5 < KW ? 5 : 0) }
        if (KW > 6) { MLP_K_RD(6 < KW ? 6 : 0) }
        if (KW > 7) { MLP_K_RD(7 < KW ? 7 : 0) }
        if (KW > 8) { MLP_K_RD(8 < KW ? 8 : 0) }
        if (KW > 9) { MLP_K_RD(9 < KW ? 9 : 0) }
        if (KW > 10) { MLP_K_RD(10 < KW ? 10 : 0) }
        if (KW > 11) { MLP_K_RD(11 < KW ? 11 : 0) }
        if (KW > 12) { MLP_K_RD(12 < KW ? 12 : 0) }
        if (KW > 13) { MLP_K_RD(13 < KW ? 13 : 0) }
        if (KW > 14) { MLP_K_RD(14 < KW ? 14 : 0) }
        MLP_K_IT8(0) MLP_K_IT8(8) MLP_K_IT8(16) MLP_K_IT8(24) MLP_K_IT8(32) MLP_K_IT8(40) MLP_K_IT8(48) MLP_K_IT8(56) MLP_K_IT8(64)
#endif
#undef MLP_K_RD
#undef MLP_K_IT
#undef MLP_K_IT8
      }
#ifdef XM_ABL_MLP_NOSM
      x8 pf[2], pl[2];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pf[tb >> 1][(tb & 1) * 4 + r] = (elem)s[tb][r];
          pl[tb >> 1][(tb & 1) * 4 + r] = (elem)s[tb][r];
        }
      l_run += 1.0f;
      (void)kv_w; (void)t0;
#else
      // softmax of the wave's own 16 heads x 64 tokens: lane (p16, g) holds tokens t0 + 16 tb + 4 g + r of head p16
      float mx = kMlaNegBig;
      if (t0 + kMlaTile > kv_w) {
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = s[tb][r] * scale_log2;
            if (t0 + tb * 16 + g * 4 + r >= kv_w) v = -INFINITY;
            s[tb][r] = v;
            mx = fmaxf(mx, v);
          }
      } else {
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
          s[tb] *= scale_log2;
          mx = fmaxf(fmaxf(mx, fmaxf(s[tb][0], s[tb][1])), fmaxf(s[tb][2], s[tb][3]));
        }
      }
      {  // xor-16 / xor-32 maximum by permlane swaps (VALU; see attention_prefill.hip)
        unsigned u = __builtin_bit_cast(unsigned, mx), c;
        asm volatile("v_mov_b32 %0, %1" : "=v"(c) : "v"(u));
        const auto r16 = __builtin_amdgcn_permlane16_swap(u, c, false, false);
        mx = fmaxf(mla_as_f32(r16[0]), mla_as_f32(r16[1]));
        u = __builtin_bit_cast(unsigned, mx);
        asm volatile("v_mov_b32 %0, %1" : "=v"(c) : "v"(u));
        const auto r32 = __builtin_amdgcn_permlane32_swap(u, c, false, false);
        mx = fmaxf(mla_as_f32(r32[0]), mla_as_f32(r32[1]));
      }
      // lazy rescale: the reference maximum only moves when the tile's maximum exceeds it by more than 2^8 (scores are in
      // the log2 domain), so P stays <= 256 -- exact in the fp32 sums, the same relative precision in the hi / lo parts --
      // and the 128-register accumulator rescale (a third of all tiles under a causal mask otherwise) all but disappears
      const float m_new = mx > m_run + 8.0f ? mx : m_run;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      float psum = 0.0f;
      x8 pf[2], pl[2];
#pragma unroll
      for (int tb = 0; tb < 4; ++tb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[tb][r] - m_new);
          psum += p;
          const elem h = (elem)p;
          pf[tb >> 1][(tb & 1) * 4 + r] = h;
          if constexpr (!P1) pl[tb >> 1][(tb & 1) * 4 + r] = (elem)(p - (float)h);
        }
      l_run = l_run * alpha + psum;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int i = 0; i < DB; ++i) mla_scale_acc(acc_o[i], alpha);
      }
#endif
      {
        const unsigned vb = lds_base + buf * BUFB + v_row;
        unsigned va[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) va[j] = vb + (v_x ^ (j << 5));
        mu32x2_t vt[8][2];
        // item I: ks = I >> 5 (32-token half of the tile), db = I & 31; handled in pairs so that the hi / lo MFMAs on one
        // accumulator are not back to back
#ifdef XM_MLA_INTERLEAVE
#define MLP_V_RD(I_)                                                                                           \
  MLA_DSR64TR(vt[(I_) & 7][0], va[0], ((I_) >> 5) * 36864 + ((I_) & 31) * 256);                                 \
  MLA_DSR64TR(vt[(I_) & 7][1], va[0], ((I_) >> 5) * 36864 + ((I_) & 31) * 256 + 18432);
#else
#define MLP_V_RD(I_)                                                                                           \
  MLA_DSR64TR(vt[(I_) & 7][0], va[(I_) & 3], ((I_) >> 5) * 32 * ROWB + (((I_) & 31) >> 2) * 128);               \
  MLA_DSR64TR(vt[(I_) & 7][1], va[(I_) & 3], ((I_) >> 5) * 32 * ROWB + (((I_) & 31) >> 2) * 128 + 16 * ROWB);
#endif
#define MLP_V_MM2(I_, WAIT_)                                                                                   \
  {                                                                                                            \
    MLA_LGKM4(WAIT_, vt[(I_) & 7][0], vt[(I_) & 7][1], vt[((I_) + 1) & 7][0], vt[((I_) + 1) & 7][1]);          \
    const x8 va8 = __builtin_shufflevector(__builtin_bit_cast(x4, vt[(I_) & 7][0]), __builtin_bit_cast(x4, vt[(I_) & 7][1]), 0, 1, 2, 3, 4, 5, 6, 7); \
    const x8 vb8 = __builtin_shufflevector(__builtin_bit_cast(x4, vt[((I_) + 1) & 7][0]), __builtin_bit_cast(x4, vt[((I_) + 1) & 7][1]), 0, 1, 2, 3, 4, 5, 6, 7); \
    acc_o[(I_) & 31] = TR::mfma(va8, pf[(I_) >> 5], acc_o[(I_) & 31]);                                         \
    acc_o[((I_) + 1) & 31] = TR::mfma(vb8, pf[(I_) >> 5], acc_o[((I_) + 1) & 31]);                             \
    if constexpr (!P1) {                                                                                       \
      acc_o[(I_) & 31] = TR::mfma(va8, pl[(I_) >> 5], acc_o[(I_) & 31]);                                       \
      acc_o[((I_) + 1) & 31] = TR::mfma(vb8, pl[(I_) >> 5], acc_o[((I_) + 1) & 31]);                           \
    }                                                                                                          \
  }
#define MLP_V_ST(I_) MLP_V_MM2(I_, 12) MLP_V_RD((I_) + 8) MLP_V_RD((I_) + 9)
#define MLP_V_ST8(B_) MLP_V_ST(B_) MLP_V_ST(B_ + 2) MLP_V_ST(B_ + 4) MLP_V_ST(B_ + 6)
#ifndef XM_ABL_MLP_NOPV
        MLP_V_RD(0) MLP_V_RD(1) MLP_V_RD(2) MLP_V_RD(3) MLP_V_RD(4) MLP_V_RD(5) MLP_V_RD(6) MLP_V_RD(7)
        MLP_V_ST8(0) MLP_V_ST8(8) MLP_V_ST8(16) MLP_V_ST8(24) MLP_V_ST8(32) MLP_V_ST8(40) MLP_V_ST8(48)
        MLP_V_MM2(56, 12) MLP_V_MM2(58, 8) MLP_V_MM2(60, 4) MLP_V_MM2(62, 0)
#else
        asm volatile("" :: "v"(pf[0]), "v"(pf[1]), "v"(pl[0]), "v"(pl[1]), "v"(va[0]));
#endif
#undef MLP_V_RD
#undef MLP_V_MM2
#undef MLP_V_ST
#undef MLP_V_ST8
      }
      __builtin_amdgcn_s_barrier();                      // every wave is done with this buffer
      stage(tile + 2, buf);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the two trailing (empty) prefetches
  }

  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  if (!live || head >= n_heads) return;
  const float inv = l_run > 0.0f ? 1.0f / l_run : 0.0f;
  T* op = out + ((int64_t)my_tok * n_heads + head) * kMlaDV;
#pragma unroll
  for (int db = 0; db < DB; ++db) {
    uint16_t hv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T t = from_f32<T>(acc_o[db][r] * inv);
      __builtin_memcpy(&hv[r], &t, 2);
    }
    *reinterpret_cast<uint2*>(op + db * 16 + g * 4) =
        make_uint2((uint32_t)hv[0] | ((uint32_t)hv[1] << 16), (uint32_t)hv[2] | ((uint32_t)hv[3] << 16));
  }
}

template <typename T>
__global__ void mla_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                 T* __restrict__ out, int nsplit) {
  const int64_t bh = blockIdx.x;
  const int d = threadIdx.x;
  const int64_t base = bh * nsplit;
  float m_star = kMlaNegBig;
  for (int s = 0; s < nsplit; ++s) m_star = fmaxf(m_star, part_ml[(base + s) * 2]);
  float o = 0.0f, l = 0.0f;
  for (int s = 0; s < nsplit; ++s) {
    const float f = exp2f(part_ml[(base + s) * 2] - m_star);
    o += f * part_o[(base + s) * kMlaDV + d];
    l += f * part_ml[(base + s) * 2 + 1];
  }
  out[bh * kMlaDV + d] = from_f32<T>(l > 0.0f ? o / l : 0.0f);
}

// one entry per query token of a ragged batch: its sequence and its (bottom-right aligned) causal key count
__global__ __launch_bounds__(256) void mla_expand_queries_kernel(const int32_t* __restrict__ cu_q,
                                                                 const int32_t* __restrict__ kv_lens, int causal,
                                                                 int32_t* __restrict__ q_seq,
                                                                 int32_t* __restrict__ q_kvlen) {
  const int b = blockIdx.x;
  const int q0 = cu_q[b], ql = cu_q[b + 1] - q0, L = kv_lens[b];
  for (int i = threadIdx.x; i < ql; i += blockDim.x) {
    const int len = causal ? L - (ql - 1 - i) : L;
    q_seq[q0 + i] = b;
    q_kvlen[q0 + i] = len > 0 ? len : 0;
  }
}

static int launch_mla(const void* q, const void* k_cache, void* out, const int32_t* seqlens_k, const int32_t* q_seq,
                      const int32_t* q_kvlen, const int32_t* block_table, int64_t max_blocks, int64_t entries,
                      int64_t n_heads, int64_t block_size, int64_t max_kv_len, float scale, int dtype, void* workspace,
                      size_t workspace_bytes, hipStream_t s) {
  const int64_t hblocks = (n_heads + 15) / 16;
  const int64_t tiles = (max_kv_len + kMlaTile - 1) / kMlaTile;
  // XLLM_MI355_MLA_SPLITS (product switch, read once): force the split-KV count (the parity tests cover the counts the planner
  // picks at other batch sizes). The register-staged kernel on 64-multiple pages (XLLM_MI355_MLA_DMA=0) lost its A/B: tuning only.
  static int split_override = -2;
  if (split_override == -2) split_override = xm_switch("XLLM_MI355_MLA_SPLITS", -1);
  XM_TUNE_VAR(dma_mode, "XLLM_MI355_MLA_DMA", 1);
  // LDS-DMA kernel: one workgroup per CU (256 resident); register-staged kernel: two per CU (512 resident)
  const bool dma = dma_mode != 0 && block_size % kMlaTile == 0;
  const int64_t resident = dma ? 256 : 512;
  int64_t nsplit = (resident + entries * hblocks - 1) / (entries * hblocks);
  if (nsplit > tiles / 8) nsplit = tiles / 8;
  if (split_override > 0) nsplit = split_override;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > 32) nsplit = 32;
  const size_t per_split = (size_t)entries * n_heads * (kMlaDV + 2) * sizeof(float);
  if (!workspace) workspace_bytes = 0;
  if ((size_t)nsplit * per_split > workspace_bytes) nsplit = (int64_t)(workspace_bytes / per_split);
  if (nsplit < 1) nsplit = 1;
  float* part_o = reinterpret_cast<float*>(workspace);
  float* part_ml = part_o ? part_o + (size_t)entries * n_heads * nsplit * kMlaDV : nullptr;
  const float scale_log2 = scale * 1.4426950408889634f;
  const dim3 grid((unsigned)(entries * hblocks * nsplit));
  XM_DISPATCH_HALF(dtype, T, {
    if (dma && hblocks == 1 && !q_seq)   // every latent byte is read once: non-temporal DMA
      hipLaunchKernelGGL((mla_decode_dma_kernel<T, true>), grid, dim3(256), 0, s, (const T*)q, (const T*)k_cache, (T*)out,
                         part_o, part_ml, seqlens_k, block_table, (int)max_blocks, (int)n_heads, (int)block_size,
                         scale_log2, (int)nsplit, q_seq, q_kvlen);
    else if (dma)
      hipLaunchKernelGGL((mla_decode_dma_kernel<T, false>), grid, dim3(256), 0, s, (const T*)q, (const T*)k_cache, (T*)out,
                         part_o, part_ml, seqlens_k, block_table, (int)max_blocks, (int)n_heads, (int)block_size,
                         scale_log2, (int)nsplit, q_seq, q_kvlen);
#ifdef XM_TUNING
    else if (block_size % kMlaTile == 0)
      hipLaunchKernelGGL((mla_decode_kernel<T, true>), grid, dim3(256), 0, s, (const T*)q, (const T*)k_cache, (T*)out,
                         part_o, part_ml, seqlens_k, block_table, (int)max_blocks, (int)n_heads, (int)block_size,
                         scale_log2, (int)nsplit, q_seq, q_kvlen);
#endif
    else
      hipLaunchKernelGGL((mla_decode_kernel<T, false>), grid, dim3(256), 0, s, (const T*)q, (const T*)k_cache, (T*)out,
                         part_o, part_ml, seqlens_k, block_table, (int)max_blocks, (int)n_heads, (int)block_size,
                         scale_log2, (int)nsplit, q_seq, q_kvlen);
    if (nsplit > 1)
      hipLaunchKernelGGL((mla_merge_kernel<T>), dim3((unsigned)(entries * n_heads)), dim3(kMlaDV), 0, s, part_o, part_ml,
                         (T*)out, (int)nsplit);
  });
  return hip_check_launch();
}

}  // namespace xm

using namespace xm;

extern "C" int xllm_mi355_mla_decode(const void* q, const void* k_cache, void* out, const int32_t* seqlens_k,
                                     const int32_t* block_table, int64_t max_blocks, int64_t batch, int64_t n_heads,
                                     int64_t head_dim, int64_t head_dim_v, int64_t block_size, int64_t n_blocks,
                                     int64_t max_kv_len, float scale, int dtype, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  (void)n_blocks;
  if (!q || !k_cache || !out || !seqlens_k || !block_table || batch < 0 || n_heads <= 0 || block_size <= 0)
    return XM_ERR_INVALID;
  if (head_dim != kMlaD || head_dim_v != kMlaDV) return XM_ERR_UNSUPPORTED;
  if ((uintptr_t)q % 16 || (uintptr_t)k_cache % 16) return XM_ERR_UNSUPPORTED;
  if (batch == 0) return XM_OK;
  return launch_mla(q, k_cache, out, seqlens_k, nullptr, nullptr, block_table, max_blocks, batch, n_heads, block_size,
                    max_kv_len, scale, dtype, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int xllm_mi355_mla_prefill(const void* q, const void* k_cache, void* out, const int32_t* cu_q,
                                      const int32_t* kv_lens, const int32_t* block_table, int64_t max_blocks,
                                      int64_t batch, int64_t total_q_tokens, int64_t n_heads, int64_t head_dim,
                                      int64_t head_dim_v, int64_t block_size, int64_t n_blocks, int64_t max_kv_len,
                                      float scale, int causal, int dtype, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  (void)n_blocks;
  if (!q || !k_cache || !out || !cu_q || !kv_lens || !block_table || batch < 0 || total_q_tokens < 0 || n_heads <= 0 ||
      block_size <= 0)
    return XM_ERR_INVALID;
  if (head_dim != kMlaD || head_dim_v != kMlaDV) return XM_ERR_UNSUPPORTED;
  if ((uintptr_t)q % 16 || (uintptr_t)k_cache % 16) return XM_ERR_UNSUPPORTED;
  if (batch == 0 || total_q_tokens == 0) return XM_OK;
  const size_t idx_bytes = (((size_t)total_q_tokens * 2 * sizeof(int32_t)) + 255) & ~(size_t)255;
  if (!workspace || workspace_bytes < idx_bytes) return XM_ERR_WORKSPACE;
  int32_t* q_seq = reinterpret_cast<int32_t*>(workspace);
  int32_t* q_kvlen = q_seq + total_q_tokens;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(mla_expand_queries_kernel, dim3((unsigned)batch), dim3(256), 0, s, cu_q, kv_lens, causal, q_seq,
                     q_kvlen);
  // enough query tokens to fill the chip without a split-KV: the kernel that shares every KV tile between four tokens
  // (XLLM_MI355_MLA_PREFILL = 0: never, 1: whenever the page size allows it; default: by workgroup count)
  static int share_mode = -2;
  if (share_mode == -2) share_mode = xm_switch("XLLM_MI355_MLA_PREFILL", -1);
  const int64_t n_groups = (total_q_tokens + 3) / 4, hblocks = (n_heads + 15) / 16;
  if (block_size % kMlaTile == 0 && share_mode != 0 && (share_mode == 1 || n_groups * hblocks >= 128)) {
    const dim3 grid((unsigned)(((n_groups + 7) / 8) * 8 * hblocks));
    static int p_mode = -2;   // XLLM_MI355_MLA_PREFILL_P=2: P = hi + lo (fp32-P accuracy), product switch
    if (p_mode == -2) p_mode = xm_switch("XLLM_MI355_MLA_PREFILL_P", 1);
    if (p_mode == 2) {
      XM_DISPATCH_HALF(dtype, T, {
        hipLaunchKernelGGL((mla_prefill_dma_kernel<T, false>), grid, dim3(256), 0, s, (const T*)q, (const T*)k_cache,
                           (T*)out, block_table, (int)max_blocks, (int)n_heads, (int)block_size,
                           scale * 1.4426950408889634f, q_seq, q_kvlen, (int)total_q_tokens, (int)n_groups);
      });
    } else {
      XM_DISPATCH_HALF(dtype, T, {
        hipLaunchKernelGGL((mla_prefill_dma_kernel<T, true>), grid, dim3(256), 0, s, (const T*)q, (const T*)k_cache,
                           (T*)out, block_table, (int)max_blocks, (int)n_heads, (int)block_size,
                           scale * 1.4426950408889634f, q_seq, q_kvlen, (int)total_q_tokens, (int)n_groups);
      });
    }
    return hip_check_launch();
  }
  return launch_mla(q, k_cache, out, kv_lens, q_seq, q_kvlen, block_table, max_blocks, total_q_tokens, n_heads,
                    block_size, max_kv_len, scale, dtype, reinterpret_cast<uint8_t*>(workspace) + idx_bytes,
                    workspace_bytes - idx_bytes, s);
}
