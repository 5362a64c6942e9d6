// attention_prefill.hip -- causal varlen flash attention for prefill (packed K/V) and chunked prefill
// (K/V gathered from the paged cache through the block table) on gfx950.
//
// Reference semantics: FlashAttentionImpl::prefill_forward / paged_forward(is_chunked_prefill=true)
// (xllm/core/layers/dcu/flash_attention.cpp:167-288) == TorchAttentionImpl prefill / chunked branches
// (layers/dcu/torch_attention.cpp:152-277) with BOTTOM-RIGHT aligned causal masks: query i of a chunk
// of q_len sees keys j <= kv_len - q_len + i (SURVEY.md 8c caveat 2); window_left >= 0 additionally
// hides keys j < pos - window_left.
//
// Structure: one workgroup (4 waves) per (sequence, q head, block of 128 queries); each wave owns 32
// queries as two N=16 MFMA column blocks. K/V tiles of 32 tokens are staged global -> registers -> LDS
// (next tile in flight during the MFMAs, two LDS buffers, one barrier per tile). As in the decode kernel
// S^T = K Q^T and O^T = V^T P^T so softmax state is lane-local: each lane owns one query of each block.
// K fragments: ds_read_b128 from rows padded by 16 B; V^T fragments: ds_read_b64_tr_b16 from rows
// padded by 32 B.
#include "common.h"

namespace xm {

typedef __bf16 pbf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 pbf16x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 pf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 pf16x4_t __attribute__((ext_vector_type(4)));
typedef float pf32x4_t __attribute__((ext_vector_type(4)));

template <typename T>
struct PfTraits;
template <>
struct PfTraits<bf16_t> {
  using x8 = pbf16x8_t;
  using x4 = pbf16x4_t;
  using elem = __bf16;
  static __device__ __forceinline__ pf32x4_t mfma(x8 a, x8 b, pf32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ x4 tr_read(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) x4*)p);
  }
};
template <>
struct PfTraits<f16_t> {
  using x8 = pf16x8_t;
  using x4 = pf16x4_t;
  using elem = _Float16;
  static __device__ __forceinline__ pf32x4_t mfma(x8 a, x8 b, pf32x4_t c) {
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

constexpr int kPfTile = 32;
constexpr int kPfQBlock = 128;
constexpr float kPfNegBig = -1e30f;

template <typename T, int D, bool PAGED>
__global__ __launch_bounds__(256, 2) void flash_prefill_kernel(
    const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v, T* __restrict__ out,
    const int32_t* __restrict__ cu_q, const int32_t* __restrict__ cu_k, const int32_t* __restrict__ kv_lens,
    const int32_t* __restrict__ block_table, int max_blocks, int nq, int nkv, int block_size, int64_t q_stride,
    int64_t k_stride, int64_t v_stride, float scale_log2, int causal, int window_left) {
  using TR = PfTraits<T>;
  using x8 = typename TR::x8;
  using x4 = typename TR::x4;
  using elem = typename TR::elem;
  constexpr int KK = D / 32, CH = D * 2 / 16, DB = D / 16;
  constexpr int NLD = kPfTile * CH / 256;  // 16-B chunks per thread per tile and operand
  constexpr int RSK = D * 2 + 16, RSV = D * 2 + 32;
  static_assert(NLD >= 1, "head dim too small");
  __shared__ __attribute__((aligned(16))) char lds[2][kPfTile * RSK + kPfTile * RSV];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p16 = lane & 15, g = lane >> 4;
  const int h = blockIdx.x, b = blockIdx.z;
  const int qb = gridDim.y - 1 - blockIdx.y;  // heaviest (latest) causal blocks first
  const int q_start = cu_q[b], q_len = cu_q[b + 1] - q_start;
  const int q0 = qb * kPfQBlock;
  if (q0 >= q_len) return;
  const int kv_len = PAGED ? kv_lens[b] : (cu_k[b + 1] - cu_k[b]);
  const int k_start = PAGED ? 0 : cu_k[b];
  const int G = nq / nkv, kvh = h / G;
  const int kvoff = kv_len - q_len;  // position of query 0 (bottom-right alignment)

  // tile range of the workgroup
  int q_hi = q0 + kPfQBlock < q_len ? q0 + kPfQBlock : q_len;  // exclusive
  int hi_tok = kv_len;
  if (causal) { int c = kvoff + q_hi; hi_tok = c < kv_len ? c : kv_len; }
  if (hi_tok < 0) hi_tok = 0;
  int lo_tok = 0;
  if (window_left >= 0) { lo_tok = kvoff + q0 - window_left; lo_tok = lo_tok > 0 ? lo_tok : 0; }
  const int tile_lo = lo_tok / kPfTile, tile_hi = (hi_tok + kPfTile - 1) / kPfTile;

  const int32_t* bt_row = PAGED ? block_table + (int64_t)b * max_blocks : nullptr;
  const int64_t row_elems = (int64_t)nkv * D;
  auto k_row_ptr = [&](int tok) -> const T* {
    tok = tok < kv_len ? tok : kv_len - 1;
    tok = tok < 0 ? 0 : tok;
    if constexpr (PAGED) return k + ((int64_t)bt_row[tok / block_size] * block_size + tok % block_size) * row_elems + (int64_t)kvh * D;
    else return k + (int64_t)(k_start + tok) * k_stride + (int64_t)kvh * D;
  };
  auto v_row_ptr = [&](int tok) -> const T* {
    tok = tok < kv_len ? tok : kv_len - 1;
    tok = tok < 0 ? 0 : tok;
    if constexpr (PAGED) return v + ((int64_t)bt_row[tok / block_size] * block_size + tok % block_size) * row_elems + (int64_t)kvh * D;
    else return v + (int64_t)(k_start + tok) * v_stride + (int64_t)kvh * D;
  };

  // Q fragments (B operand): lane (n = query p16 of block nb, k group g)
  x8 qf[2][KK];
  int qidx[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    qidx[nb] = q0 + wave * 32 + nb * 16 + p16;
    const bool ok = qidx[nb] < q_len;
    const T* qp = q + (int64_t)(q_start + (ok ? qidx[nb] : 0)) * q_stride + (int64_t)h * D;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      if (ok) qf[nb][kk] = *reinterpret_cast<const x8*>(qp + (kk * 4 + g) * 8);
      else qf[nb][kk] = x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  // wave-level query position bounds (for tile skipping / mask detection)
  const int wq_lo = kvoff + q0 + wave * 32;
  int wq_hi = kvoff + (q0 + wave * 32 + 31 < q_len - 1 ? q0 + wave * 32 + 31 : q_len - 1);
  const bool wave_active = (q0 + wave * 32) < q_len;

  pf32x4_t acc_o[2][DB];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int i = 0; i < DB; ++i) acc_o[nb][i] = pf32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {kPfNegBig, kPfNegBig}, l_run[2] = {0.f, 0.f};

  uint4 rk[NLD], rv[NLD];
  auto load_global = [&](int tile) {
    const int t0 = tile * kPfTile;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * 256, row = c / CH, col = c % CH;
      rk[i] = *reinterpret_cast<const uint4*>(k_row_ptr(t0 + row) + col * 8);
      rv[i] = *reinterpret_cast<const uint4*>(v_row_ptr(t0 + row) + col * 8);
    }
  };
  auto write_lds = [&](int buf, int tile) {
    const int t0 = tile * kPfTile;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = tid + i * 256, row = c / CH, col = c % CH;
      *reinterpret_cast<uint4*>(&lds[buf][row * RSK + col * 16]) = rk[i];
      uint4 vv = rv[i];
      if (t0 + row >= kv_len) vv = make_uint4(0, 0, 0, 0);  // 0 * NaN guard for masked keys
      *reinterpret_cast<uint4*>(&lds[buf][kPfTile * RSK + row * RSV + col * 16]) = vv;
    }
  };

  if (tile_lo < tile_hi) {
    load_global(tile_lo);
    write_lds(0, tile_lo);
    __syncthreads();
    int cur = 0;
    for (int tile = tile_lo; tile < tile_hi; ++tile) {
      const bool more = tile + 1 < tile_hi;
      if (more) load_global(tile + 1);
      const int t0 = tile * kPfTile;
      const bool compute = wave_active && (!causal || t0 <= wq_hi) && (window_left < 0 || t0 + kPfTile > wq_lo - window_left);
      if (compute) {
        const char* lk = lds[cur];
        const char* lv = lds[cur] + kPfTile * RSK;
        pf32x4_t s[2][2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int blk = 0; blk < 2; ++blk) s[nb][blk] = pf32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
          for (int kk = 0; kk < KK; ++kk) {
            const x8 ka = *reinterpret_cast<const x8*>(lk + (blk * 16 + p16) * RSK + (kk * 4 + g) * 16);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) s[nb][blk] = TR::mfma(ka, qf[nb][kk], s[nb][blk]);
          }
        const bool need_mask = (t0 + kPfTile > kv_len) || (causal && t0 + kPfTile - 1 > wq_lo) ||
                               (window_left >= 0 && t0 < wq_hi - window_left);
        x8 pf[2], pl[2];  // P = hi + lo 16-bit parts (see attention_decode.hip)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int qpos = kvoff + qidx[nb];
          float mx = kPfNegBig;
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float val = s[nb][blk][r] * scale_log2;
              if (need_mask) {
                const int tok = t0 + blk * 16 + g * 4 + r;
                const bool vis = tok < kv_len && (!causal || tok <= qpos) && (window_left < 0 || tok >= qpos - window_left);
                if (!vis) val = -INFINITY;
              }
              s[nb][blk][r] = val;
              mx = fmaxf(mx, val);
            }
          mx = fmaxf(mx, __shfl_xor(mx, 16));
          mx = fmaxf(mx, __shfl_xor(mx, 32));
          const float m_new = fmaxf(m_run[nb], mx);
          const float alpha = __builtin_amdgcn_exp2f(m_run[nb] - m_new);  // v_exp_f32: results below 2^-126 flush to 0
          m_run[nb] = m_new;
          float psum = 0.0f;
#pragma unroll
          for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float p = __builtin_amdgcn_exp2f(s[nb][blk][r] - m_new);
              psum += p;
              const elem hi = (elem)p;
              pf[nb][blk * 4 + r] = hi;
              pl[nb][blk * 4 + r] = (elem)(p - (float)hi);
            }
          l_run[nb] = l_run[nb] * alpha + psum;
          // the running maximum settles after a few tiles; once no lane of the wave raised it, alpha == 1 everywhere
          // and the DB*4 multiplies are skipped (wave-uniform branch; multiplying by 1.0f is exact, so same bits)
          if (__any(alpha != 1.0f)) {
#pragma unroll
            for (int i = 0; i < DB; ++i) acc_o[nb][i] *= alpha;
          }
        }
        const char* trb = lv + (4 * g + (p16 >> 2)) * RSV + (p16 & 3) * 8;
#pragma unroll
        for (int db = 0; db < DB; ++db) {
          const x4 lo = TR::tr_read(trb + db * 32);
          const x4 hi = TR::tr_read(trb + 16 * RSV + db * 32);
          const x8 vt = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            acc_o[nb][db] = TR::mfma(vt, pf[nb], acc_o[nb][db]);
            acc_o[nb][db] = TR::mfma(vt, pl[nb], acc_o[nb][db]);
          }
        }
      }
      if (more) write_lds(cur ^ 1, tile + 1);
      __syncthreads();
      cur ^= 1;
    }
  }

  // epilogue: O^T lane (query p16, g) holds d = db*16 + 4g + r
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    float l = l_run[nb];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (qidx[nb] >= q_len) continue;
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
    T* op = out + (int64_t)(q_start + qidx[nb]) * nq * D + (int64_t)h * D;
#pragma unroll
    for (int db = 0; db < DB; ++db) {
      uint16_t hv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        T t = from_f32<T>(acc_o[nb][db][r] * inv);
        __builtin_memcpy(&hv[r], &t, 2);
      }
      *reinterpret_cast<uint2*>(op + db * 16 + g * 4) =
          make_uint2((uint32_t)hv[0] | ((uint32_t)hv[1] << 16), (uint32_t)hv[2] | ((uint32_t)hv[3] << 16));
    }
  }
}

template <typename T, int D, bool PAGED>
int launch_flash_prefill(const void* q, const void* k, const void* v, void* out, const int32_t* cu_q,
                         const int32_t* cu_k, const int32_t* kv_lens, const int32_t* block_table, int64_t max_blocks,
                         int64_t batch, int64_t nq, int64_t nkv, int64_t block_size, int64_t q_stride,
                         int64_t k_stride, int64_t v_stride, int64_t max_q_len, float scale, int causal,
                         int64_t window_left, hipStream_t s) {
  const int qblocks = (int)((max_q_len + kPfQBlock - 1) / kPfQBlock);
  if (qblocks <= 0) return XM_OK;
  const dim3 grid((unsigned)nq, (unsigned)qblocks, (unsigned)batch);
  const int wl = window_left < 0 ? -1 : (window_left > 0x3fffffff ? 0x3fffffff : (int)window_left);
  hipLaunchKernelGGL((flash_prefill_kernel<T, D, PAGED>), grid, dim3(256), 0, s, (const T*)q, (const T*)k,
                     (const T*)v, (T*)out, cu_q, cu_k, kv_lens, block_table, (int)max_blocks, (int)nq, (int)nkv,
                     (int)block_size, q_stride, k_stride, v_stride, scale * 1.4426950408889634f, causal, wl);
  return hip_check_launch();
}

#define XM_INST_PREFILL(T, D, P)                                                                                   \
  template int launch_flash_prefill<T, D, P>(const void*, const void*, const void*, void*, const int32_t*,         \
                                             const int32_t*, const int32_t*, const int32_t*, int64_t, int64_t,     \
                                             int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, float, \
                                             int, int64_t, hipStream_t);
XM_INST_PREFILL(bf16_t, 128, true)
XM_INST_PREFILL(bf16_t, 128, false)
XM_INST_PREFILL(bf16_t, 64, true)
XM_INST_PREFILL(bf16_t, 64, false)
XM_INST_PREFILL(f16_t, 128, true)
XM_INST_PREFILL(f16_t, 128, false)
XM_INST_PREFILL(f16_t, 64, true)
XM_INST_PREFILL(f16_t, 64, false)

}  // namespace xm
