// attention_decode.hip -- paged decode attention for gfx950 (the HBM-bound 70%-of-roofline target).
//
// Reference semantics: AttentionImpl::paged_forward(is_chunked_prefill=false)
// (xllm/core/layers/dcu/flash_attention.cpp:220-288) == TorchAttentionImpl decode branch
// (layers/dcu/torch_attention.cpp:278-337): one query token per sequence, non-causal softmax over the
// kv_len cached tokens gathered page by page through the block table; GQA by kv_head = h / (nq/nkv).
//
// MI355X design (DESIGN.md "paged decode"):
//  * one WAVE owns (sequence, kv head, token range); the nq/nkv query heads of the GQA group are stacked
//    as the N dimension of the MFMA so every K/V byte is read from HBM exactly once.
//  * K is loaded straight from HBM into the MFMA A-fragment layout (16 tokens x 8 contiguous d per lane,
//    16-byte loads) -- no LDS round trip; S^T = K * Q^T ("swapped QK^T") leaves each lane with the scores
//    of ONE query head, so the online softmax is lane-local plus two cross-lane maxima per 32 tokens.
//  * V is loaded with fully coalesced 16-byte row loads, written once to a wave-private LDS tile
//    (row padded by 32 B -> conflict-free) and read back transposed with ds_read_b64_tr_b16 as the
//    A operand of O^T = V^T * P^T; the accumulator O^T keeps one query head per lane, so the rescale by
//    exp2(m_old - m_new) is lane-local too. No workgroup barrier inside the token loop.
//  * a workgroup = 4 waves = the kv heads of one token range (contiguous 1 KiB rows of the page are
//    consumed by one CU at about the same time), or 4 token sub-ranges when the rank holds < 4 kv heads
//    (TP >= 2); grid-level split-KV partials (m, l, O) are merged by a second tiny kernel.
//  * tile t+1 (16 KiB of K+V per wave) is in flight in registers while tile t is computed:
//    8 waves/CU x 16-32 KiB = 128-256 KiB outstanding per CU.
#include "common.h"

namespace xm {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <typename T>
struct AttnTraits;
template <>
struct AttnTraits<bf16_t> {
  using x8 = bf16x8_t;
  using x4 = bf16x4_t;
  using elem = __bf16;
  static __device__ __forceinline__ f32x4_t mfma(x8 a, x8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ x4 tr_read(const void* lds_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) x4*)lds_ptr);
  }
};
template <>
struct AttnTraits<f16_t> {
  using x8 = f16x8_t;
  using x4 = f16x4_t;
  using elem = _Float16;
  static __device__ __forceinline__ f32x4_t mfma(x8 a, x8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ x4 tr_read(const void* lds_ptr) {
    typedef __fp16 hfp16x4 __attribute__((__vector_size__(8)));
    hfp16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) hfp16x4*)lds_ptr);
    x4 o;
    __builtin_memcpy(&o, &r, 8);
    return o;
  }
};

constexpr int kTile = 32;         // tokens per tile (= K dim of the PV MFMA)
constexpr float kNegBig = -1e30f;  // finite "-inf" for the running max (log2 domain)

// D = head dim (K and V), UNIFORM: block_size % 32 == 0 so a tile lives in one page
// KROWS: how the K tile is fetched (see issue_loads) -- chosen by the launch plan, bit-identical results either way
template <typename T, int D, bool UNIFORM, bool DEEP, bool KROWS>
__global__ __launch_bounds__(256, DEEP ? 1 : 2) void paged_decode_kernel(
    const T* __restrict__ q, const T* __restrict__ kc, const T* __restrict__ vc, T* __restrict__ out,
    float* __restrict__ part_o, float* __restrict__ part_ml, const int32_t* __restrict__ cu_q,
    const int32_t* __restrict__ kv_lens, const int32_t* __restrict__ block_table, int max_blocks, int nq,
    int nkv, int block_size, int64_t q_stride, float scale_log2, int nsplit, int hpw, int window_left,
    int8_t* __restrict__ out_q, float* __restrict__ out_scale, int part_mode) {
  using TR = AttnTraits<T>;
  using x8 = typename TR::x8;
  using x4 = typename TR::x4;
  using elem = typename TR::elem;
  constexpr int KK = D / 32;            // MFMA k-steps over the head dim
  constexpr int CH = D * 2 / 16;        // 16-byte chunks per K/V row
  constexpr int TPI = 64 / CH;          // V rows fetched per wave-wide load instruction
  constexpr int NV = kTile / TPI;       // V load instructions per tile
  constexpr int DB = D / 16;            // 16-wide output d blocks
  constexpr int RSB = D * 2 + 32;       // padded LDS row stride in bytes
  constexpr int WAVE_LDS = kTile * RSB; // per-wave LDS bytes (>= 16*D*4 for the epilogue)
  static_assert(WAVE_LDS >= 16 * D * 4, "epilogue staging must fit");

  __shared__ __attribute__((aligned(16))) char lds[4 * WAVE_LDS];
  __shared__ float ml_sh[4][16][2];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p16 = lane & 15, g = lane >> 4;
  const int nsub = 4 / hpw;
  const int ngroups = nkv / hpw;
  int wg = blockIdx.x;
  const int split = wg % nsplit; wg /= nsplit;
  const int hg = wg % ngroups;
  const int b = wg / ngroups;
  const int hs = wave % hpw, sub = wave / hpw;
  const int kvh = hg * hpw + hs;
  const int G = nq / nkv;

  const int kv_len = kv_lens[b];
  const int t_lo = (window_left >= 0 && kv_len - 1 - window_left > 0) ? kv_len - 1 - window_left : 0;
  const int tile_lo = t_lo / kTile, tile_hi = (kv_len + kTile - 1) / kTile;
  const int nslots = nsplit * nsub, slot = split * nsub + sub;
  const int ntiles = tile_hi - tile_lo > 0 ? tile_hi - tile_lo : 0;
  const int per = (ntiles + nslots - 1) / nslots;
  int my_lo = tile_lo + slot * per;
  int my_hi = my_lo + per < tile_hi ? my_lo + per : tile_hi;
  my_lo = __builtin_amdgcn_readfirstlane(my_lo);
  my_hi = __builtin_amdgcn_readfirstlane(my_hi);

  const int32_t* bt_row = block_table + (int64_t)b * max_blocks;
  const int64_t row_elems = (int64_t)nkv * D;  // elements per token row of the cache
  char* my_lds = lds + wave * WAVE_LDS;

  // ---- Q as the MFMA B operand: lane (n = q head p16, k group g) holds Q[p16][(kk*4+g)*8 .. +7]
  x8 qf[KK];
  {
    const int64_t qtok = cu_q ? cu_q[b] : b;
    const T* qp = q + qtok * q_stride + (int64_t)(kvh * G + p16) * D;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (p16 < G) qf[kk] = *reinterpret_cast<const x8*>(qp + (kk * 4 + g) * 8);
      else qf[kk] = x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }

  f32x4_t acc_o[DB];
#pragma unroll
  for (int i = 0; i < DB; ++i) acc_o[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = kNegBig, l_run = 0.0f;

  // K/V register stages: named (never runtime-indexed) so they stay in VGPRs; native vector types so hipcc
  // emits plain 16-byte loads/stores (no memcpy allocas). DEEP: three stages = two tiles (32 KiB) in flight per
  // wave while the third is computed -- with ONE workgroup per CU (the best split count, see attention_api.hip)
  // a wave may use up to 512 registers.
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  x8 k0[2][KK], k1[2][KK], k2[2][KK];
  u32x4 v0[NV], v1[NV], v2[NV];

  // The KV stream is read exactly ONCE per launch: NON-TEMPORAL loads (global_load_dwordx4 ... nt). Round 4, tools/attn_ab.py on
  // one box, alternating libraries: 355 -> 329 us per launch at cfg3 = 6.05 -> 6.53 TB/s (0.756 -> 0.817 of 8 TB/s) -- above the
  // 6.25-6.47 TB/s "read ceiling" earlier rounds measured with ordinary loads (tools/hbm_read_bench.hip): a streaming read that
  // allocates in the L2 pays for evicting 2 GB of lines nobody will hit. -DXM_ATTN_TEMPORAL_LOADS restores the old loads (A/B).
#ifndef XM_ATTN_TEMPORAL_LOADS
#define XM_KV_LOAD(P) __builtin_nontemporal_load(P)
#else
#define XM_KV_LOAD(P) (*(P))
#endif
  auto page_of_tile = [&](int tile) -> int {  // UNIFORM only: scalar page id of a tile (clamped)
    int idx = (tile * kTile) / block_size;
    const int last = (kv_len - 1) / block_size;
    idx = idx > last ? last : idx;
    idx = idx < 0 ? 0 : idx;
    return bt_row[idx];
  };
  // every load is unconditional (tiles past the range re-load the last tile, tokens past kv_len the last token):
  // a branch around a load would make hipcc fall back to s_waitcnt vmcnt(0) and collapse the prefetch depth
  // The prefetch of the iteration that computes the wave's LAST tile asks for tile my_hi: nobody consumes it. It used to re-load
  // the whole last tile (16 KiB per wave: 0.8 % of the launch's bytes at cfg3, 6 % on a 16-tile range -- and with non-temporal
  // loads those lines are often gone from the L2 again, so they came from HBM: PMC traffic 1.007 x / 1.047 x algorithmic); now
  // every lane of such a request reads the FIRST row of that tile (one 2D-byte row, two cache lines). Still unconditional.
  auto issue_loads = [&](int tile, int page, x8 (&kr)[2][KK], u32x4 (&vr)[NV]) {
    const bool past = tile >= my_hi;            // wave-uniform
    tile = past ? my_hi - 1 : tile;
    const int t0 = tile * kTile;
    if constexpr (KROWS) {
      // K like V: an instruction fetches WHOLE head rows (64 / CH token rows x 2 D bytes) and compute_tile() re-lays the tile
      // through the wave's LDS. For a wave whose head is a strided 2D-byte slice of the token rows and whose workgroup does not
      // hold the row's other heads (hpw < nkv: small batches, DP replicas) this is the faster fetch -- round 4, alternating
      // libraries on one box (profiles/r04_attn_krows.txt): cfg2 54.5 -> 50.2 us, B = 128 / 64 / 32 at ctx 4096 177 -> 167,
      // 99.5 -> 92.5, 57.8 -> 52.4 us; with all kv heads of a row in the workgroup (hpw == nkv: the B = 256 headline, TP shards)
      // the direct operand-layout fetch below ties or wins by 0.6 %, so the plan picks per launch.
#pragma unroll
      for (int j = 0; j < 2 * KK; ++j) {
        int tok = past ? t0 : t0 + j * TPI + lane / CH;
        tok = tok < kv_len ? tok : kv_len - 1;
        int64_t rowi;
        if constexpr (UNIFORM) rowi = (int64_t)page * block_size + (tok % block_size);
        else rowi = (int64_t)bt_row[tok / block_size] * block_size + (tok % block_size);
        kr[j / KK][j % KK] = XM_KV_LOAD(reinterpret_cast<const x8*>(kc + rowi * row_elems + (int64_t)kvh * D + (lane % CH) * 8));
      }
    } else {
      // K straight into the MFMA A-operand layout: lane (p16, g) holds K[token blk*16 + p16][(kk*4+g)*8 .. +7] (64-byte pieces of
      // 16 token rows per instruction)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        int tok = past ? t0 : t0 + blk * 16 + p16;
        tok = tok < kv_len ? tok : kv_len - 1;
        int64_t rowi;
        if constexpr (UNIFORM) rowi = (int64_t)page * block_size + (tok % block_size);
        else rowi = (int64_t)bt_row[tok / block_size] * block_size + (tok % block_size);
        const T* kp = kc + rowi * row_elems + (int64_t)kvh * D + g * 8;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) kr[blk][kk] = XM_KV_LOAD(reinterpret_cast<const x8*>(kp + kk * 32));
      }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int tok = past ? t0 : t0 + i * TPI + lane / CH;
      tok = tok < kv_len ? tok : kv_len - 1;
      int64_t rowi;
      if constexpr (UNIFORM) rowi = (int64_t)page * block_size + (tok % block_size);
      else rowi = (int64_t)bt_row[tok / block_size] * block_size + (tok % block_size);
      vr[i] = XM_KV_LOAD(reinterpret_cast<const u32x4*>(vc + rowi * row_elems + (int64_t)kvh * D + (lane % CH) * 8));
    }
  };
  auto page_clamped = [&](int tile) -> int {
    if constexpr (UNIFORM) return page_of_tile(tile < my_hi ? tile : my_hi - 1);
    else return 0;
  };

  auto compute_tile = [&](int tile, const x8 (&kr)[2][KK], const u32x4 (&vr)[NV]) {
#ifdef XM_ABL_ATTN_NOCOMPUTE  /* ablation build: consume the loaded registers, no LDS / MFMA / softmax work */
    {
      unsigned fold = 0;
#pragma unroll
      for (int i = 0; i < NV; ++i) fold ^= vr[i].x ^ vr[i].y ^ vr[i].z ^ vr[i].w;
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
          u32x4 t;
          __builtin_memcpy(&t, &kr[blk][kk], 16);
          fold ^= t.x ^ t.y ^ t.z ^ t.w;
        }
      l_run += (fold == 0x12345678u) ? 1.0f : 0.0f;
      return;
    }
#endif
    const int t0 = tile * kTile;
    const bool partial = (t0 + kTile > kv_len) || (t0 < t_lo);

    // ---- KROWS: K tile -> wave-private LDS (row major, padded rows) -> MFMA A-operand fragments. The same buffer takes the V
    //      tile next: the LDS serves a wave's requests in order, so the V writes below cannot overtake these reads.
    x8 kf[2][KK];
    if constexpr (KROWS) {
#pragma unroll
      for (int j = 0; j < 2 * KK; ++j)
        *reinterpret_cast<x8*>(my_lds + (j * TPI + lane / CH) * RSB + (lane % CH) * 16) = kr[j / KK][j % KK];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
          kf[blk][kk] = *reinterpret_cast<const x8*>(my_lds + (blk * 16 + p16) * RSB + (kk * 4 + g) * 16);
    } else {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) kf[blk][kk] = kr[blk][kk];
    }

    // ---- V tile -> wave-private LDS (row major, padded rows); invalid rows zeroed (0 * NaN guard)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int r = i * TPI + lane / CH;
      u32x4 v = vr[i];
      if (partial && (t0 + r >= kv_len)) v = u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(my_lds + r * RSB + (lane % CH) * 16) = v;
    }

    // ---- S^T = K * Q^T : lane holds S[q = p16][token = blk*16 + 4g + r]
    f32x4_t s[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      s[blk] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) s[blk] = TR::mfma(kf[blk][kk], qf[kk], s[blk]);
    }
    // ---- online softmax (log2 domain), lane-local except the 2 cross-group maxima
    float mx = kNegBig;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = s[blk][r] * scale_log2;
        if (partial) {
          const int tok = t0 + blk * 16 + g * 4 + r;
          if (tok >= kv_len || tok < t_lo) v = -INFINITY;
        }
        s[blk][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.0f;
    // P is fed to the matrix core as hi + lo 16-bit parts (p = hi + lo to ~2^-17 relative): the PV MFMAs
    // are idle-cheap in this HBM-bound kernel and the output then matches an fp32-P reference to rounding.
    x8 pf, pl;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = exp2f(s[blk][r] - m_new);
        psum += p;
        const elem hi = (elem)p;
        pf[blk * 4 + r] = hi;
        pl[blk * 4 + r] = (elem)(p - (float)hi);
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int i = 0; i < DB; ++i) acc_o[i] *= alpha;

    // ---- O^T += V^T * P^T : A = V^T via transposed LDS reads, k slot (g, j): j<4 -> token 4g+j,
    //      j>=4 -> token 16+4g+(j-4), matching the P fragment above
    const char* trb = my_lds + (4 * g + (p16 >> 2)) * RSB + (p16 & 3) * 8;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      x4 lo = TR::tr_read(trb + db * 32);
      x4 hi = TR::tr_read(trb + 16 * RSB + db * 32);
      x8 vt;
      vt[0] = lo[0]; vt[1] = lo[1]; vt[2] = lo[2]; vt[3] = lo[3];
      vt[4] = hi[0]; vt[5] = hi[1]; vt[6] = hi[2]; vt[7] = hi[3];
      acc_o[db] = TR::mfma(vt, pf, acc_o[db]);
      acc_o[db] = TR::mfma(vt, pl, acc_o[db]);
    }
  };

  if (my_lo < my_hi) {
    int tile = my_lo;
    if constexpr (DEEP) {
      int pg2 = page_clamped(my_lo + 2), pg3;
      issue_loads(my_lo, page_clamped(my_lo), k0, v0);
      issue_loads(my_lo + 1, page_clamped(my_lo + 1), k1, v1);
      while (true) {
        pg3 = page_clamped(tile + 3);
        issue_loads(tile + 2, pg2, k2, v2);
        compute_tile(tile, k0, v0);
        pg2 = pg3;
        if (++tile >= my_hi) break;
        pg3 = page_clamped(tile + 3);
        issue_loads(tile + 2, pg2, k0, v0);
        compute_tile(tile, k1, v1);
        pg2 = pg3;
        if (++tile >= my_hi) break;
        pg3 = page_clamped(tile + 3);
        issue_loads(tile + 2, pg2, k1, v1);
        compute_tile(tile, k2, v2);
        pg2 = pg3;
        if (++tile >= my_hi) break;
      }
    } else {
      int pg1 = page_clamped(my_lo + 1), pg2;
      issue_loads(my_lo, page_clamped(my_lo), k0, v0);
      while (true) {
        pg2 = page_clamped(tile + 2);
        issue_loads(tile + 1, pg1, k1, v1);
        compute_tile(tile, k0, v0);
        pg1 = pg2;
        if (++tile >= my_hi) break;
        pg2 = page_clamped(tile + 2);
        issue_loads(tile + 1, pg1, k0, v0);
        compute_tile(tile, k1, v1);
        pg1 = pg2;
        if (++tile >= my_hi) break;
      }
    }
  }

  // ---- epilogue: stage (m, l, O) of every wave in LDS, merge the sub-ranges of each head
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  {
    float* o_st = reinterpret_cast<float*>(my_lds);
#pragma unroll
    for (int db = 0; db < DB; ++db)
      *reinterpret_cast<f32x4_t*>(o_st + p16 * D + db * 16 + g * 4) = acc_o[db];
    if (g == 0) { ml_sh[wave][p16][0] = m_run; ml_sh[wave][p16][1] = l_run; }
  }
  __syncthreads();
  const int64_t qtok = cu_q ? cu_q[b] : b;
  if (out_q) {
    // N1 fusion: the workgroup holds ALL heads of this token (host guarantees nsplit == 1 and one head group),
    // so the per-token int8 quantisation that feeds o_proj (scaled_quantize of the 16-bit output) is done here:
    // pass 1 = 16-bit output + |max|, pass 2 = quantise. Bit-identical to paged_attention -> scaled_quantize.
    __shared__ float red[32];
    auto value = [&](int e, int& head, int& d) -> float {
      const int h_s = e / (16 * D), qq = (e / D) & 15;
      d = e % D;
      head = h_s * G + qq;
      if (qq >= G) return 0.0f;
      float m_star = kNegBig;
      for (int sb = 0; sb < nsub; ++sb) m_star = fmaxf(m_star, ml_sh[sb * hpw + h_s][qq][0]);
      float o = 0.0f, l = 0.0f;
      for (int sb = 0; sb < nsub; ++sb) {
        const int w = sb * hpw + h_s;
        const float f = exp2f(ml_sh[w][qq][0] - m_star);
        o += f * reinterpret_cast<const float*>(lds + w * WAVE_LDS)[qq * D + d];
        l += f * ml_sh[w][qq][1];
      }
      return r16<T>(l > 0.0f ? o / l : 0.0f);
    };
    float amax = 0.0f;
    for (int e = threadIdx.x; e < hpw * 16 * D; e += 256) {
      int head, d;
      const float v = value(e, head, d);
      if (((e / D) & 15) >= G) continue;
      amax = fmaxf(amax, fabsf(v));
      if (out) out[qtok * (int64_t)nq * D + (int64_t)head * D + d] = from_f32<T>(v);
    }
    amax = block_max(amax, red);
    const float qinv = (amax > 1e-10f) ? 127.0f / amax : 0.0f;
    for (int e = threadIdx.x; e < hpw * 16 * D; e += 256) {
      int head, d;
      const float v = value(e, head, d);
      if (((e / D) & 15) >= G) continue;
      out_q[qtok * (int64_t)nq * D + (int64_t)head * D + d] = (int8_t)fmaxf(-127.0f, fminf(127.0f, rintf(v * qinv)));
    }
    if (threadIdx.x == 0) out_scale[qtok] = amax / 127.0f;
    return;
  }
  for (int e = threadIdx.x; e < hpw * 16 * D; e += 256) {
    const int h_s = e / (16 * D), qq = (e / D) & 15, d = e % D;
    if (qq >= G) continue;
    float m_star = kNegBig;
    for (int sb = 0; sb < nsub; ++sb) m_star = fmaxf(m_star, ml_sh[sb * hpw + h_s][qq][0]);
    float o = 0.0f, l = 0.0f;
    for (int sb = 0; sb < nsub; ++sb) {
      const int w = sb * hpw + h_s;
      const float f = exp2f(ml_sh[w][qq][0] - m_star);
      o += f * reinterpret_cast<const float*>(lds + w * WAVE_LDS)[qq * D + d];
      l += f * ml_sh[w][qq][1];
    }
    const int head = (hg * hpw + h_s) * G + qq;
    if (!part_mode) {
      out[qtok * (int64_t)nq * D + (int64_t)head * D + d] = from_f32<T>(l > 0.0f ? o / l : 0.0f);
    } else {  // grid-level split-KV, or a finishing kernel that wants (o, m, l) of the whole range (nsplit == 1)
      const int64_t pi = ((int64_t)b * nq + head) * nsplit + split;
      part_o[pi * D + d] = o;
      if (d == 0) { part_ml[pi * 2] = m_star; part_ml[pi * 2 + 1] = l; }
    }
  }
}

// merge of the grid-level split-KV partials: one workgroup (D threads) per (sequence, head)
template <typename T, int D>
__global__ void paged_decode_merge_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml,
                                          T* __restrict__ out, const int32_t* __restrict__ cu_q, int nq, int nsplit) {
  const int b = blockIdx.x / nq, head = blockIdx.x % nq;
  const int d = threadIdx.x;
  const int64_t base = ((int64_t)b * nq + head) * nsplit;
  float m_star = kNegBig;
  for (int s = 0; s < nsplit; ++s) m_star = fmaxf(m_star, part_ml[(base + s) * 2]);
  float o = 0.0f, l = 0.0f;
  for (int s = 0; s < nsplit; ++s) {
    const float f = exp2f(part_ml[(base + s) * 2] - m_star);
    o += f * part_o[(base + s) * D + d];
    l += f * part_ml[(base + s) * 2 + 1];
  }
  const int64_t qtok = cu_q ? cu_q[b] : b;
  out[qtok * (int64_t)nq * D + (int64_t)head * D + d] = from_f32<T>(l > 0.0f ? o / l : 0.0f);
}

// merge + per-token int8 quantisation in one launch (N1 fusion for the plans whose workgroups do not hold a whole token:
// fewer than all kv heads per workgroup, or grid-level split-KV): one workgroup per token merges the nsplit partials of all
// nq heads exactly like paged_decode_merge_kernel, rounds to T, and quantises the row like scaled_quantize -- the bits of
// paged_attention followed by scaled_quantize, one launch and one 16-bit round trip less.
// NSP > 0 (round 6): at most NSP splits, and EVERY load of the launch -- the (m, l) pairs and all partial rows -- is requested
// before anything is consumed: the launch is one memory latency deep instead of 1 + nsplit (the round-2 form, NSP = 0, waited
// for the (m, l) pairs, then for each split's rows in turn: 12.7 us against 4.9 + 4.8 for the two launches it replaces)
template <typename T, int D, int VPT, int NSP, int NT>
__global__ __launch_bounds__(NT) void paged_decode_finish_int8_kernel(const float* __restrict__ part_o,
                                                                       const float* __restrict__ part_ml, T* __restrict__ out,
                                                                       int8_t* __restrict__ out_q, float* __restrict__ out_scale,
                                                                       int nq, int nsplit) {
  constexpr int kMaxHeads = 32, kMaxSplit = 32;     // nq <= 32 (VPT * NT / D), nsplit <= 32 (decode_num_splits)
  __shared__ float red[32];
  __shared__ float fs[kMaxHeads][kMaxSplit];        // exp2(m_s - m*) per (head, split)
  __shared__ float ls[kMaxHeads];                   // merged denominators
  const int b = blockIdx.x;
  const int n = nq * D;
  float v[VPT];
  if constexpr (NSP > 0) {
    float po[NSP][VPT];
#pragma unroll
    for (int s = 0; s < NSP; ++s)
#pragma unroll
      for (int i = 0; i < VPT; ++i) {
        const int e = threadIdx.x + i * NT;
        po[s][i] = (s < nsplit && e < n) ? part_o[(((int64_t)b * nq + e / D) * nsplit + s) * D + e % D] : 0.0f;
      }
    if ((int)threadIdx.x < nq) {
      const int head = threadIdx.x;
      const int64_t base = ((int64_t)b * nq + head) * nsplit;
      float2 ml[NSP];
#pragma unroll
      for (int s = 0; s < NSP; ++s)
        ml[s] = s < nsplit ? *reinterpret_cast<const float2*>(part_ml + (base + s) * 2) : make_float2(kNegBig, 0.0f);
      float m_star = kNegBig;
#pragma unroll
      for (int s = 0; s < NSP; ++s)
        if (s < nsplit) m_star = fmaxf(m_star, ml[s].x);
      float l = 0.0f;
#pragma unroll
      for (int s = 0; s < NSP; ++s)
        if (s < nsplit) {
          const float f = exp2f(ml[s].x - m_star);
          fs[head][s] = f;
          l += f * ml[s].y;
        }
      ls[head] = l;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VPT; ++i) v[i] = 0.0f;
#pragma unroll
    for (int s = 0; s < NSP; ++s)
      if (s < nsplit) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
          const int e = threadIdx.x + i * NT;
          if (e < n) v[i] += fs[e / D][s] * po[s][i];
        }
      }
  } else {
    // phase A: one thread per head turns the (m, l) pairs into the split weights
    if ((int)threadIdx.x < nq) {
      const int head = threadIdx.x;
      const int64_t base = ((int64_t)b * nq + head) * nsplit;
      float m_star = kNegBig;
      for (int s = 0; s < nsplit; ++s) m_star = fmaxf(m_star, part_ml[(base + s) * 2]);
      float l = 0.0f;
      for (int s = 0; s < nsplit; ++s) {
        const float f = exp2f(part_ml[(base + s) * 2] - m_star);
        fs[head][s] = f;
        l += f * part_ml[(base + s) * 2 + 1];
      }
      ls[head] = l;
    }
    __syncthreads();
    // phase B: o = sum_s f_s * o_s in split order (the merge kernel's order), all of a thread's elements per split at once
#pragma unroll
    for (int i = 0; i < VPT; ++i) v[i] = 0.0f;
    for (int s = 0; s < nsplit; ++s) {
#pragma unroll
      for (int i = 0; i < VPT; ++i) {
        const int e = threadIdx.x + i * NT;
        if (e < n) {
          const int head = e / D, d = e % D;
          v[i] += fs[head][s] * part_o[(((int64_t)b * nq + head) * nsplit + s) * D + d];
        }
      }
    }
  }
  float amax = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int e = threadIdx.x + i * NT;
    if (e < n) {
      const float l = ls[e / D];
      v[i] = r16<T>(l > 0.0f ? v[i] / l : 0.0f);
      amax = fmaxf(amax, fabsf(v[i]));
      if (out) out[(int64_t)b * n + e] = from_f32<T>(v[i]);
    }
  }
  amax = block_max(amax, red);
  const float qinv = (amax > 1e-10f) ? 127.0f / amax : 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int e = threadIdx.x + i * NT;
    if (e < n) out_q[(int64_t)b * n + e] = (int8_t)fmaxf(-127.0f, fminf(127.0f, rintf(v[i] * qinv)));
  }
  if (threadIdx.x == 0) out_scale[b] = amax / 127.0f;
}

int decode_num_splits(int64_t batch, int64_t nkv, int hpw, int64_t max_kv_len);
bool decode_deep_prefetch();
int decode_heads_per_wg(int64_t batch, int64_t nkv);
int decode_exclusive_cu();

template <typename T, int D>
int launch_paged_decode(const void* q, const void* kc, const void* vc, void* out, const int32_t* cu_q,
                        const int32_t* kv_lens, const int32_t* block_table, int64_t max_blocks, int64_t batch,
                        int64_t nq, int64_t nkv, int64_t block_size, int64_t q_stride, int64_t max_kv_len,
                        float scale, int64_t window_left, void* workspace, size_t ws_bytes, hipStream_t s,
                        int8_t* out_q, float* out_scale) {
  const int hpw = decode_heads_per_wg(batch, nkv);
  int nsplit = decode_num_splits(batch, nkv, hpw, max_kv_len);
  // the fused int8 epilogue needs the whole token in one workgroup; the other plans (split-KV, fewer than all kv heads per
  // workgroup) leave (o, m, l) partials in the workspace and a finishing launch merges + quantises them. Without a
  // workspace those plans are declined (the caller then runs paged_attention + scaled_quantize)
  const size_t per_split = (size_t)batch * nq * (D + 2) * sizeof(float);
  if (!workspace) ws_bytes = 0;
  // (plans with one split but fewer than all kv heads per workgroup write their 16-bit output directly: for them the
  // finishing launch would only replace scaled_quantize by a launch of the same cost -- measured -- so they are declined)
  if (out_q && nsplit == 1 && nkv / hpw != 1) return XM_ERR_UNSUPPORTED;
  const bool finish = out_q && nsplit != 1;
  if (finish && (ws_bytes < (size_t)nsplit * per_split || nq * D > 1024 * 4 || nq > 32 || cu_q)) return XM_ERR_UNSUPPORTED;
  // degrade the split count to what the caller's workspace holds (1 split needs none)
  if ((size_t)nsplit * per_split > ws_bytes) nsplit = (int)(ws_bytes / per_split);
  if (nsplit < 1) nsplit = 1;
  const int part_mode = (nsplit > 1 || finish) ? 1 : 0;
  int8_t* const kq = finish ? nullptr : out_q;          // the attention kernel's own int8 epilogue
  float* const kqs = finish ? nullptr : out_scale;
  float* part_o = reinterpret_cast<float*>(workspace);
  float* part_ml = part_o ? part_o + (size_t)batch * nq * nsplit * D : nullptr;
  const float scale_log2 = scale * 1.4426950408889634f;
  const dim3 grid((unsigned)(batch * (nkv / hpw) * nsplit));
  const int wl = window_left < 0 ? -1 : (window_left > 0x3fffffff ? 0x3fffffff : (int)window_left);
  // tuning arm (XLLM_MI355_DECODE_EXCL=1, -DXM_TUNING flavour only): grids of at most one workgroup per CU reserve enough extra LDS that two
  // workgroups cannot share a CU, so the dispatcher has to spread them over all 256
  const size_t dyn = (decode_exclusive_cu() && grid.x <= 256) ? 48 * 1024 : 0;
  const bool krows = hpw < nkv;   // the wave's head is a strided slice of rows whose other heads are elsewhere (see issue_loads)
#define XM_DECODE_LAUNCH(UNI_, DEEP_, KROWS_, DYN_)                                                                        \
  hipLaunchKernelGGL((paged_decode_kernel<T, D, UNI_, DEEP_, KROWS_>), grid, dim3(256), DYN_, s, (const T*)q, (const T*)kc, \
                     (const T*)vc, (T*)out, part_o, part_ml, cu_q, kv_lens, block_table, (int)max_blocks, (int)nq,        \
                     (int)nkv, (int)block_size, q_stride, scale_log2, nsplit, hpw, wl, kq, kqs, part_mode)
#ifdef XM_TUNING  /* the three-stage arm (round-1 A/B loser) exists only in the tuning flavour */
  if (block_size % kTile == 0 && decode_deep_prefetch()) XM_DECODE_LAUNCH(true, true, false, dyn);
  else
#endif
  if (block_size % kTile == 0) {
    if (krows) XM_DECODE_LAUNCH(true, false, true, dyn);
    else XM_DECODE_LAUNCH(true, false, false, dyn);
  } else {
    if (krows) XM_DECODE_LAUNCH(false, false, true, 0);
    else XM_DECODE_LAUNCH(false, false, false, 0);
  }
#undef XM_DECODE_LAUNCH
  if (finish) {
    const int vpt = (int)((nq * D + 255) / 256);
#define XM_FINISH_(V_, N_, T_)                                                                                           \
  hipLaunchKernelGGL((paged_decode_finish_int8_kernel<T, D, V_, N_, T_>), dim3((unsigned)batch), dim3(T_), 0, s, part_o,  \
                     part_ml, (T*)out, out_q, out_scale, (int)nq, nsplit)
#define XM_FINISH(V_, T_)                                                                                               \
  {                                                                                                                     \
    if (nsplit <= 2) XM_FINISH_(V_, 2, T_);                                                                             \
    else if (nsplit <= 4) XM_FINISH_(V_, 4, T_);                                                                        \
    else XM_FINISH_(V_, 0, T_);                                                                                         \
  }
    // rows beyond 1024 elements: 1024 threads (few tokens per launch there -- the launch is latency, not throughput)
    if (vpt <= 4) XM_FINISH(4, 256)
    else XM_FINISH(4, 1024)
#undef XM_FINISH
#undef XM_FINISH_
  } else if (nsplit > 1) {
    hipLaunchKernelGGL((paged_decode_merge_kernel<T, D>), dim3((unsigned)(batch * nq)), dim3(D), 0, s, part_o,
                       part_ml, (T*)out, cu_q, (int)nq, nsplit);
  }
  return hip_check_launch();
}

template int launch_paged_decode<bf16_t, 128>(const void*, const void*, const void*, void*, const int32_t*,
                                              const int32_t*, const int32_t*, int64_t, int64_t, int64_t, int64_t,
                                              int64_t, int64_t, int64_t, float, int64_t, void*, size_t, hipStream_t, int8_t*, float*);
template int launch_paged_decode<bf16_t, 64>(const void*, const void*, const void*, void*, const int32_t*,
                                             const int32_t*, const int32_t*, int64_t, int64_t, int64_t, int64_t,
                                             int64_t, int64_t, int64_t, float, int64_t, void*, size_t, hipStream_t, int8_t*, float*);
template int launch_paged_decode<f16_t, 128>(const void*, const void*, const void*, void*, const int32_t*,
                                             const int32_t*, const int32_t*, int64_t, int64_t, int64_t, int64_t,
                                             int64_t, int64_t, int64_t, float, int64_t, void*, size_t, hipStream_t, int8_t*, float*);
template int launch_paged_decode<f16_t, 64>(const void*, const void*, const void*, void*, const int32_t*,
                                            const int32_t*, const int32_t*, int64_t, int64_t, int64_t, int64_t,
                                            int64_t, int64_t, int64_t, float, int64_t, void*, size_t, hipStream_t, int8_t*, float*);

}  // namespace xm
