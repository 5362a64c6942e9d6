// rowwise.hip -- HBM-bound row-wise operators of the hot path for gfx950:
// KV write, block-table build, RMSNorm (+residual, +fp8 / +int8 quant), RoPE, fused QK-norm+RoPE,
// act_and_mul (+int8 quant), per-token int8 quant, fp8 quant.
// One workgroup (256 threads = 4 waves) per token row, 16-byte vector IO, the row kept in registers
// between the reduction pass and the write pass (each HBM byte is read once).
#include "common.h"

namespace xm {

constexpr int kRowThreads = 256;
constexpr int kMaxVec = 4;  // 16-B chunks cached per thread -> rows up to 16 KiB

// ------------------------------------------------------------------------------------------------
// KV write (reference: kernels/cuda/reshape_paged_cache.cu:24-63). grid = tokens; 16-B copies when
// rows are 16-B aligned, else element copies.
// ------------------------------------------------------------------------------------------------
template <typename V>
__global__ __launch_bounds__(256) void reshape_paged_cache_kernel(
    const int32_t* __restrict__ slot_ids, const V* __restrict__ k, const V* __restrict__ v,
    V* __restrict__ kc, V* __restrict__ vc, int64_t row_v /* row length in V units */,
    int64_t k_stride_v, int64_t v_stride_v, int64_t block_size, int64_t n_blocks) {
  const int64_t t = blockIdx.x;
  const int64_t slot = slot_ids[t];
  if (slot < 0) return;
  if (slot / block_size >= n_blocks) return;
  // cache row index == slot (block*block_size + offset)
  const V* ks = k + t * k_stride_v;
  V* kd = kc + slot * row_v;
  if (v) {
    const V* vs = v + t * v_stride_v;
    V* vd = vc + slot * row_v;
    for (int64_t i = threadIdx.x; i < row_v; i += blockDim.x) {
      kd[i] = ks[i];
      vd[i] = vs[i];
    }
  } else {  // K-only caches: the MLA latent cache (store_latent_cache, deepseek_v2_attention.cpp:170-178)
    for (int64_t i = threadIdx.x; i < row_v; i += blockDim.x) kd[i] = ks[i];
  }
}

// ------------------------------------------------------------------------------------------------
// KV block copy (reference: kernels/cuda/block_copy.cu:56-118; WorkerImpl::execute_cuda_block_copy_kernel,
// runtime/worker_impl.cpp:1071-1082): beam-search / prefix forks copy whole cache blocks. Destination j belongs to
// source group g = the first g with j < cum_sum[g] (cum_sum = inclusive running count of destinations per source);
// for every layer, key[dst[j]] <- key[src[g]] and value likewise. A pure byte copy: one workgroup moves a 16-KiB
// piece of K and of V (four 16-byte chunks per thread per array, all loads issued before the stores), the layer
// base addresses come from two device arrays of int64 like the reference's. grid = (pieces, destinations, layers).
// ------------------------------------------------------------------------------------------------
constexpr int kBlockCopyChunks = 4;
template <typename V>
__global__ __launch_bounds__(256) void block_copy_kernel(const int64_t* __restrict__ k_ptrs,
                                                         const int64_t* __restrict__ v_ptrs,
                                                         const int32_t* __restrict__ src_blocks,
                                                         const int32_t* __restrict__ dst_blocks,
                                                         const int32_t* __restrict__ cum_sum, int num_groups,
                                                         int64_t units_per_block /* in V units */) {
  const int j = blockIdx.y;
  int lo = 0, hi = num_groups - 1;                 // block_copy.cu:38-50: first group whose running count exceeds j
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if (j < cum_sum[mid]) hi = mid; else lo = mid + 1;
  }
  const int64_t so = (int64_t)src_blocks[lo] * units_per_block, dof = (int64_t)dst_blocks[j] * units_per_block;
  V* kc = reinterpret_cast<V*>(static_cast<uintptr_t>(k_ptrs[blockIdx.z]));
  V* vc = v_ptrs ? reinterpret_cast<V*>(static_cast<uintptr_t>(v_ptrs[blockIdx.z])) : nullptr;
  const int64_t base = (int64_t)blockIdx.x * (256 * kBlockCopyChunks) + threadIdx.x;
  // The unit width is chosen on the host from the block size alone; the cache BASE addresses live in device arrays. A base that
  // is not a multiple of the unit (a cache view at an odd offset; torch allocations never are) takes a byte-wise walk of the same
  // units instead of a misaligned vector access (round-5 advisor). Workgroup-uniform branch.
  if (sizeof(V) > 1 && (((uintptr_t)kc | (uintptr_t)vc) & (sizeof(V) - 1)) != 0) {
    for (int c = 0; c < kBlockCopyChunks; ++c) {
      const int64_t i = base + c * 256;
      if (i >= units_per_block) continue;
      const uint8_t* ks = reinterpret_cast<const uint8_t*>(kc + so + i);
      uint8_t* kd = reinterpret_cast<uint8_t*>(kc + dof + i);
      for (int bb = 0; bb < (int)sizeof(V); ++bb) kd[bb] = ks[bb];
      if (vc) {
        const uint8_t* vs = reinterpret_cast<const uint8_t*>(vc + so + i);
        uint8_t* vd = reinterpret_cast<uint8_t*>(vc + dof + i);
        for (int bb = 0; bb < (int)sizeof(V); ++bb) vd[bb] = vs[bb];
      }
    }
    return;
  }
  V kr[kBlockCopyChunks], vr[kBlockCopyChunks];
#pragma unroll
  for (int c = 0; c < kBlockCopyChunks; ++c) {
    const int64_t i = base + c * 256;
    if (i < units_per_block) {
      kr[c] = kc[so + i];
      if (vc) vr[c] = vc[so + i];
    }
  }
#pragma unroll
  for (int c = 0; c < kBlockCopyChunks; ++c) {
    const int64_t i = base + c * 256;
    if (i < units_per_block) {
      kc[dof + i] = kr[c];
      if (vc) vc[dof + i] = vr[c];
    }
  }
}

__global__ __launch_bounds__(256) void build_block_table_kernel(const int32_t* __restrict__ indptr,
                                                                const int32_t* __restrict__ indices,
                                                                int32_t total_pages,
                                                                int32_t* __restrict__ table) {
  const int32_t s = blockIdx.x;
  const int32_t start = indptr[s], n = indptr[s + 1] - start;
  int32_t* row = table + (int64_t)s * total_pages;
  for (int32_t j = threadIdx.x; j < total_pages; j += blockDim.x) row[j] = (j < n) ? indices[start + j] : -1;
}

// ------------------------------------------------------------------------------------------------
// RMSNorm family (reference: kernels/cuda/norm.cu:45-174, 229-425)
//   ADD  : residual <- r16(input + residual) first (16-bit add), norm reads the sum
//   QUANT: 0 -> out dtype T, y = r16(r16(x*inv)*w)
//          1 -> fp8 e4m3: v = float(r16(x*inv))*float(w); q = sat(clamp(v * (1/scale)))
//          2 -> int8 per token of y (the 16-bit norm output), scale[t] = amax/127
// ------------------------------------------------------------------------------------------------
// 16-byte row IO with an optional non-temporal hint (round 4): prefill-sized tensors (beyond the 32 MB of L2, every byte touched
// once per operator) are streamed; decode-sized ones are consumed from the L2 by the next kernel and keep ordinary accesses
typedef unsigned rw_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 rw_ld16(const void* p, bool nt) {
  const rw_u32x4* q = reinterpret_cast<const rw_u32x4*>(p);
  const rw_u32x4 v = nt ? __builtin_nontemporal_load(q) : *q;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void rw_st16(void* p, const uint4& v, bool nt) {
  const rw_u32x4 w = {v.x, v.y, v.z, v.w};
  if (nt) __builtin_nontemporal_store(w, reinterpret_cast<rw_u32x4*>(p));
  else *reinterpret_cast<rw_u32x4*>(p) = w;
}

template <typename T, bool ADD, int QUANT>
__global__ __launch_bounds__(kRowThreads) void rms_norm_kernel(
    void* __restrict__ out, T* __restrict__ input, T* __restrict__ residual,
    const T* __restrict__ weight, const float* __restrict__ fp8_scale, float* __restrict__ q_scale,
    float eps, int hidden, int64_t in_stride, bool write_input, bool nt = false) {
  constexpr int N = RowVec<T>::N;
  __shared__ float smem[32];
  const int64_t t = blockIdx.x;
  const int nvec = hidden / N;
  T* in_row = input + t * in_stride;
  T* res_row = ADD ? residual + t * (int64_t)hidden : nullptr;
  RowVec<T> xv[kMaxVec];
  float ss = 0.0f;
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i) {
    const int c = threadIdx.x + i * kRowThreads;
    if (c < nvec) {
      xv[i].raw = rw_ld16(reinterpret_cast<const uint4*>(in_row) + c, nt);
      if constexpr (ADD) {
        RowVec<T> rv;
        rv.raw = rw_ld16(reinterpret_cast<const uint4*>(res_row) + c, nt);
#pragma unroll
        for (int j = 0; j < N; ++j) xv[i].set(j, xv[i].get(j) + rv.get(j));  // r16(x + r)
        rw_st16(reinterpret_cast<uint4*>(res_row) + c, xv[i].raw, nt);
      }
#pragma unroll
      for (int j = 0; j < N; ++j) { float x = xv[i].get(j); ss += x * x; }
    }
  }
  ss = block_sum(ss, smem);
  const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
  float amax = 0.0f;
  [[maybe_unused]] float sinv = 0.0f;
  if constexpr (QUANT == 1) sinv = 1.0f / fp8_scale[0];
#pragma unroll
  for (int i = 0; i < kMaxVec; ++i) {
    const int c = threadIdx.x + i * kRowThreads;
    if (c < nvec) {
      RowVec<T> wv;
      wv.raw = reinterpret_cast<const uint4*>(weight)[c];
      if constexpr (QUANT == 1) {
        uint32_t pk[N / 4];
#pragma unroll
        for (int j = 0; j < N; j += 4) {
          uint32_t w = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float vv = r16<T>(xv[i].get(j + e) * inv) * wv.get(j + e);
            w |= (uint32_t)f32_to_e4m3_sat(vv * sinv) << (8 * e);
          }
          pk[j / 4] = w;
        }
        uint8_t* o = reinterpret_cast<uint8_t*>(out) + t * (int64_t)hidden + (int64_t)c * N;
        if constexpr (N == 8) *reinterpret_cast<uint2*>(o) = make_uint2(pk[0], pk[1]);
        else *reinterpret_cast<uint32_t*>(o) = pk[0];
      } else {
#pragma unroll
        for (int j = 0; j < N; ++j) {
          float y = r16<T>(r16<T>(xv[i].get(j) * inv) * wv.get(j));
          xv[i].set(j, y);
          amax = fmaxf(amax, fabsf(y));
        }
        if constexpr (QUANT == 0) {
          reinterpret_cast<uint4*>(reinterpret_cast<T*>(out) + t * (int64_t)hidden)[c] = xv[i].raw;
        } else if (write_input) {  // fused_add semantics also return the 16-bit norm in `input`
          reinterpret_cast<uint4*>(in_row)[c] = xv[i].raw;
        }
      }
    }
  }
  if constexpr (QUANT == 2) {
    amax = block_max(amax, smem);
    const float qinv = (amax > 1e-10f) ? 127.0f / amax : 0.0f;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
      const int c = threadIdx.x + i * kRowThreads;
      if (c < nvec) {
        uint32_t pk[N / 4];
#pragma unroll
        for (int j = 0; j < N; j += 4) {
          uint32_t w = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float qv = fmaxf(-127.0f, fminf(127.0f, rintf(xv[i].get(j + e) * qinv)));
            w |= ((uint32_t)(int)qv & 0xffu) << (8 * e);
          }
          pk[j / 4] = w;
        }
        int8_t* o = reinterpret_cast<int8_t*>(out) + t * (int64_t)hidden + (int64_t)c * N;
        if constexpr (N == 8) *reinterpret_cast<uint2*>(o) = make_uint2(pk[0], pk[1]);
        else *reinterpret_cast<uint32_t*>(o) = pk[0];
      }
    }
    if (threadIdx.x == 0) q_scale[t] = amax / 127.0f;
  }
}

// N1 fusion across the GEMM boundary: the row-parallel int8 GEMMs of a decode step (o_proj, down_proj) run split-K and
// leave exact int32 sums in the GEMM workspace; instead of a dequant kernel that writes the 16-bit output and a
// fused_add_rms_norm(+quant) kernel that reads it back, this kernel dequantises the sums in registers (same
// expression and rounding as the split-K epilogue: r16(float(acc)*a_s[m]*w_s[n] + bias[n])), re-zeroes the workspace
// (its invariant), adds the residual, normalises and quantises exactly like rms_norm_kernel<T, true, QUANT>.
// Bit-identical to scaled_matmul -> fused_add_rms_norm(-> scaled_quantize). QUANT: 0 = 16-bit norm out, 2 = int8.
// Round 6: NT threads x NV 16-byte chunks per thread (512 x 1 up to 4096 columns, 512 x 2 up to 8192, 256 x 4 beyond), and EVERY load
// of the row -- the slabs, both scale vectors, bias, residual, norm weight -- is requested before anything is consumed: the
// kernel is one memory latency deep instead of five (slabs / scales + residual per chunk iteration, then the norm weight), 8.5 ->
// 6.x us at M = 256 inside the decode step (profiles/r06_slab_consumers.txt). Same expressions in the same order: bit-identical.
template <typename T, int QUANT, int NT, int NV>
__global__ __launch_bounds__(NT) void acc_add_rms_norm_kernel(
    void* __restrict__ out, float* __restrict__ q_scale, int32_t* __restrict__ acc, const float* __restrict__ a_scale,
    const float* __restrict__ w_scale, const T* __restrict__ bias, T* __restrict__ residual,
    const T* __restrict__ weight, float eps, int hidden, int n_slabs, int64_t slab_stride) {
  // n_slabs == 0: the zero-at-rest split-K workspace of gemm.hip (read, then re-zeroed); n_slabs >= 1: that many K-slice
  // slabs of exact int32 partial sums written by gemm_ws.hip (summed here, left as they are)
  static_assert(sizeof(T) == 2, "16-bit activations");
  constexpr int N = 8;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  __shared__ float smem[32];
  const int64_t t = blockIdx.x;
  const int nvec = hidden / N;
  int32_t* acc_row = acc + t * (int64_t)hidden;
  T* res_row = residual + t * (int64_t)hidden;
  const float as = a_scale[t];
  RowVec<T> xv[NV];
  i32x4 a0[NV], a1[NV], b0[NV][7], b1[NV][7];
  float4 w0[NV], w1[NV];
  RowVec<T> bv[NV], rv[NV], nw[NV];
  // ---- every load of the row in flight (chunks past the row read chunk 0 and are ignored)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c0 = threadIdx.x + i * NT;
    const int c = c0 < nvec ? c0 : 0;
    a0[i] = reinterpret_cast<const i32x4*>(acc_row)[2 * c];
    a1[i] = reinterpret_cast<const i32x4*>(acc_row)[2 * c + 1];
#pragma unroll
    for (int u = 0; u < 7; ++u) {   // slabs 1..7 (the planner makes at most 8 slices; more: the loop below)
      b0[i][u] = i32x4{0, 0, 0, 0};
      b1[i][u] = i32x4{0, 0, 0, 0};
      if (1 + u < n_slabs) {
        const int32_t* p = acc_row + (1 + u) * slab_stride;
        b0[i][u] = reinterpret_cast<const i32x4*>(p)[2 * c];
        b1[i][u] = reinterpret_cast<const i32x4*>(p)[2 * c + 1];
      }
    }
    w0[i] = reinterpret_cast<const float4*>(w_scale)[2 * c];
    w1[i] = reinterpret_cast<const float4*>(w_scale)[2 * c + 1];
    bv[i].raw = bias ? reinterpret_cast<const uint4*>(bias)[c] : make_uint4(0, 0, 0, 0);
    rv[i].raw = reinterpret_cast<const uint4*>(res_row)[c];
    nw[i].raw = reinterpret_cast<const uint4*>(weight)[c];
  }
  float ss = 0.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = threadIdx.x + i * NT;
    if (c < nvec) {
      if (n_slabs == 0) {
        reinterpret_cast<i32x4*>(acc_row)[2 * c] = i32x4{0, 0, 0, 0};
        reinterpret_cast<i32x4*>(acc_row)[2 * c + 1] = i32x4{0, 0, 0, 0};
      }
      // integer sums: any order is exact (the grouping of round 3's loop is kept)
      i32x4 s0 = a0[i] + (((b0[i][0] + b0[i][1]) + (b0[i][2] + b0[i][3])) + ((b0[i][4] + b0[i][5]) + b0[i][6]));
      i32x4 s1 = a1[i] + (((b1[i][0] + b1[i][1]) + (b1[i][2] + b1[i][3])) + ((b1[i][4] + b1[i][5]) + b1[i][6]));
      for (int sl = 8; sl < n_slabs; ++sl) {   // (not reached by the planner's plans)
        const int32_t* p = acc_row + sl * slab_stride;
        s0 += reinterpret_cast<const i32x4*>(p)[2 * c];
        s1 += reinterpret_cast<const i32x4*>(p)[2 * c + 1];
      }
      const int av[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
      const float wv[8] = {w0[i].x, w0[i].y, w0[i].z, w0[i].w, w1[i].x, w1[i].y, w1[i].z, w1[i].w};
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float y = r16<T>((float)av[j] * as * wv[j] + (bias ? bv[i].get(j) : 0.0f));  // the GEMM's 16-bit output
        xv[i].set(j, y + rv[i].get(j));                                                      // r16(y + residual)
      }
      reinterpret_cast<uint4*>(res_row)[c] = xv[i].raw;
#pragma unroll
      for (int j = 0; j < N; ++j) { float x = xv[i].get(j); ss += x * x; }
    }
  }
  ss = block_sum(ss, smem);
  const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
  float amax = 0.0f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = threadIdx.x + i * NT;
    if (c < nvec) {
#pragma unroll
      for (int j = 0; j < N; ++j) {
        float y = r16<T>(r16<T>(xv[i].get(j) * inv) * nw[i].get(j));
        xv[i].set(j, y);
        amax = fmaxf(amax, fabsf(y));
      }
      if constexpr (QUANT == 0) reinterpret_cast<uint4*>(reinterpret_cast<T*>(out) + t * (int64_t)hidden)[c] = xv[i].raw;
    }
  }
  if constexpr (QUANT == 2) {
    amax = block_max(amax, smem);
    const float qinv = (amax > 1e-10f) ? 127.0f / amax : 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = threadIdx.x + i * NT;
      if (c < nvec) {
        uint32_t pk[2];
#pragma unroll
        for (int j = 0; j < N; j += 4) {
          uint32_t w = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float qv = fmaxf(-127.0f, fminf(127.0f, rintf(xv[i].get(j + e) * qinv)));
            w |= ((uint32_t)(int)qv & 0xffu) << (8 * e);
          }
          pk[j / 4] = w;
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<int8_t*>(out) + t * (int64_t)hidden + (int64_t)c * N) =
            make_uint2(pk[0], pk[1]);
      }
    }
    if (threadIdx.x == 0) q_scale[t] = amax / 127.0f;
  }
}

// launcher used by gemm.hip (declared in common.h)
int launch_acc_add_rms_norm(void* out, float* q_scale, int32_t* acc, const float* a_scale, const float* w_scale,
                            const void* bias, void* residual, const void* weight, float eps, int64_t M, int64_t N,
                            int dtype, int quant, hipStream_t s, int n_slabs) {
  if (N % 8 != 0 || N > (int64_t)kRowThreads * kMaxVec * 8) return XM_ERR_UNSUPPORTED;
  if (((uintptr_t)out | (uintptr_t)residual | (uintptr_t)weight | (uintptr_t)bias | (uintptr_t)w_scale | (uintptr_t)acc) % 16)
    return XM_ERR_UNSUPPORTED;
#define XM_ACC_NORM(Q_, NT_, NV_)                                                                                       \
  hipLaunchKernelGGL((acc_add_rms_norm_kernel<T, Q_, NT_, NV_>), dim3(M), dim3(NT_), 0, s, out, q_scale, acc, a_scale,  \
                     w_scale, (const T*)bias, (T*)residual, (const T*)weight, eps, (int)N, n_slabs, M * N)
  XM_DISPATCH_HALF(dtype, T, {
    if (quant) {
      if (N <= 4096) XM_ACC_NORM(2, 512, 1);
      else if (N <= 8192) XM_ACC_NORM(2, 512, 2);
      else XM_ACC_NORM(2, 256, 4);
    } else {
      if (N <= 4096) XM_ACC_NORM(0, 512, 1);
      else if (N <= 8192) XM_ACC_NORM(0, 512, 2);
      else XM_ACC_NORM(0, 256, 4);
    }
  });
#undef XM_ACC_NORM
  return hip_check_launch();
}

// generic fallback (any hidden / alignment): scalar, two passes over global memory
template <typename T, bool ADD, int QUANT>
__global__ __launch_bounds__(kRowThreads) void rms_norm_generic_kernel(
    void* __restrict__ out, T* __restrict__ input, T* __restrict__ residual,
    const T* __restrict__ weight, const float* __restrict__ fp8_scale, float* __restrict__ q_scale,
    float eps, int hidden, int64_t in_stride, bool write_input) {
  __shared__ float smem[32];
  const int64_t t = blockIdx.x;
  T* in_row = input + t * in_stride;
  T* res_row = ADD ? residual + t * (int64_t)hidden : nullptr;
  float ss = 0.0f;
  for (int i = threadIdx.x; i < hidden; i += blockDim.x) {
    float x = to_f32(in_row[i]);
    if constexpr (ADD) { x = r16<T>(x + to_f32(res_row[i])); res_row[i] = from_f32<T>(x); }
    ss += x * x;
  }
  ss = block_sum(ss, smem);
  const float inv = 1.0f / sqrtf(ss / (float)hidden + eps);
  const float sinv = (QUANT == 1) ? 1.0f / fp8_scale[0] : 0.0f;
  const T* src = ADD ? res_row : in_row;
  float amax = 0.0f;
  if constexpr (ADD) __syncthreads();
  for (int i = threadIdx.x; i < hidden; i += blockDim.x) {
    float n = r16<T>(to_f32(src[i]) * inv);
    if constexpr (QUANT == 1) {
      reinterpret_cast<uint8_t*>(out)[t * (int64_t)hidden + i] = f32_to_e4m3_sat(n * to_f32(weight[i]) * sinv);
    } else {
      float y = r16<T>(n * to_f32(weight[i]));
      amax = fmaxf(amax, fabsf(y));
      if constexpr (QUANT == 0) reinterpret_cast<T*>(out)[t * (int64_t)hidden + i] = from_f32<T>(y);
    }
  }
  if constexpr (QUANT == 2) {
    amax = block_max(amax, smem);
    const float qinv = (amax > 1e-10f) ? 127.0f / amax : 0.0f;
    for (int i = threadIdx.x; i < hidden; i += blockDim.x) {
      float y = r16<T>(r16<T>(to_f32(src[i]) * inv) * to_f32(weight[i]));
      float qv = fmaxf(-127.0f, fminf(127.0f, rintf(y * qinv)));
      reinterpret_cast<int8_t*>(out)[t * (int64_t)hidden + i] = (int8_t)qv;
      if (write_input) in_row[i] = from_f32<T>(y);
    }
    if (threadIdx.x == 0) q_scale[t] = amax / 127.0f;
  }
}

template <typename T, bool ADD, int QUANT>
int launch_rms_norm(void* out, void* input, void* residual, const void* weight, const float* fp8_scale,
                    float* q_scale, float eps, int64_t T_, int64_t H, int64_t in_stride,
                    bool write_input, hipStream_t s) {
  if (T_ == 0) return XM_OK;
  constexpr int N = Vec16B<T>::N;
  const bool vec_ok = (H % N == 0) && (H / N <= kRowThreads * kMaxVec) && (in_stride % N == 0) &&
                      ((uintptr_t)input % 16 == 0) && ((uintptr_t)weight % 16 == 0) &&
                      (!ADD || (uintptr_t)residual % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (vec_ok)
    hipLaunchKernelGGL((rms_norm_kernel<T, ADD, QUANT>), dim3(T_), dim3(kRowThreads), 0, s, out, (T*)input,
                       (T*)residual, (const T*)weight, fp8_scale, q_scale, eps, (int)H, in_stride, write_input,
                       T_ * H * (int64_t)sizeof(T) > (48ll << 20));
  else
    hipLaunchKernelGGL((rms_norm_generic_kernel<T, ADD, QUANT>), dim3(T_), dim3(kRowThreads), 0, s, out,
                       (T*)input, (T*)residual, (const T*)weight, fp8_scale, q_scale, eps, (int)H, in_stride,
                       write_input);
  return hip_check_launch();
}

// ------------------------------------------------------------------------------------------------
// RoPE (reference: kernels/cuda/rope.cu:27-154): all arithmetic in T (each op rounds to 16 bit)
// one workgroup per token; thread i handles (head, rot_offset) pairs
// ------------------------------------------------------------------------------------------------
template <typename T, bool NEOX>
__global__ __launch_bounds__(256) void rope_kernel(const int64_t* __restrict__ positions, T* __restrict__ q,
                                                   T* __restrict__ k, const T* __restrict__ cache, int rot_dim,
                                                   int64_t q_stride, int64_t k_stride, int64_t head_stride,
                                                   int nq, int nk) {
  const int64_t t = blockIdx.x;
  const int half = rot_dim >> 1;
  const T* cp = cache + positions[t] * rot_dim;
  const int total = (nq + nk) * half;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int h = i / half, j = i - h * half;
    T* arr = (h < nq) ? q + t * q_stride + (int64_t)h * head_stride
                      : k + t * k_stride + (int64_t)(h - nq) * head_stride;
    const int xi = NEOX ? j : 2 * j, yi = NEOX ? half + j : 2 * j + 1;
    const float c = to_f32(cp[j]), s = to_f32(cp[half + j]);
    const float x = to_f32(arr[xi]), y = to_f32(arr[yi]);
    arr[xi] = from_f32<T>(r16<T>(x * c) - r16<T>(y * s));
    arr[yi] = from_f32<T>(r16<T>(y * c) + r16<T>(x * s));
  }
}

// N1 fusion: RoPE on q,k (in place) + KV write of the rotated k and of v in one pass (one workgroup per token).
// Same arithmetic as rope_kernel followed by reshape_paged_cache_kernel: bit-identical results.
template <typename T, bool NEOX>
__global__ __launch_bounds__(256) void rope_and_cache_kernel(
    const int64_t* __restrict__ positions, T* __restrict__ q, T* __restrict__ k, const T* __restrict__ v,
    const T* __restrict__ cache, const int32_t* __restrict__ slot_ids, T* __restrict__ kc, T* __restrict__ vc,
    int rot_dim, int64_t q_stride, int64_t k_stride, int64_t v_stride, int head_size, int nq, int nk,
    int64_t block_size, int64_t n_blocks) {
  const int64_t t = blockIdx.x;
  const int half = rot_dim >> 1;
  const T* cp = cache + positions[t] * rot_dim;
  const int64_t slot = slot_ids[t];
  const bool store = slot >= 0 && slot / block_size < n_blocks;
  T* kc_row = kc + slot * (int64_t)nk * head_size;
  T* vc_row = vc + slot * (int64_t)nk * head_size;
  const int total = (nq + nk) * half;
  for (int i = threadIdx.x; i < total; i += blockDim.x) {
    const int h = i / half, j = i - h * half;
    const bool is_k = h >= nq;
    T* arr = is_k ? k + t * k_stride + (int64_t)(h - nq) * head_size : q + t * q_stride + (int64_t)h * head_size;
    const int xi = NEOX ? j : 2 * j, yi = NEOX ? half + j : 2 * j + 1;
    const float c = to_f32(cp[j]), s = to_f32(cp[half + j]);
    const float x = to_f32(arr[xi]), y = to_f32(arr[yi]);
    const T nx = from_f32<T>(r16<T>(x * c) - r16<T>(y * s));
    const T ny = from_f32<T>(r16<T>(y * c) + r16<T>(x * s));
    arr[xi] = nx;
    arr[yi] = ny;
    if (is_k && store) {
      kc_row[(h - nq) * head_size + xi] = nx;
      kc_row[(h - nq) * head_size + yi] = ny;
    }
  }
  if (!store) return;
  // un-rotated tail of k (rot_dim < head_size) and the whole v row
  const int tail = head_size - rot_dim;
  for (int i = threadIdx.x; i < nk * tail; i += blockDim.x) {
    const int h = i / tail, e = rot_dim + (i - h * tail);
    kc_row[h * head_size + e] = k[t * k_stride + (int64_t)h * head_size + e];
  }
  const int row = nk * head_size;
  const T* vs = v + t * v_stride;
  if ((row * sizeof(T)) % 16 == 0 && ((uintptr_t)vs % 16 == 0) && ((uintptr_t)vc_row % 16 == 0)) {
    const int nv = row * (int)sizeof(T) / 16;
    for (int i = threadIdx.x; i < nv; i += blockDim.x)
      reinterpret_cast<uint4*>(vc_row)[i] = reinterpret_cast<const uint4*>(vs)[i];
  } else {
    for (int i = threadIdx.x; i < row; i += blockDim.x) vc_row[i] = vs[i];
  }
}

// The same operator with 16-byte accesses (round 4): NeoX layout, rot_dim == head_size, 16-bit T, head_size % 16 == 0, 16-byte
// aligned rows. A work item is (head, 8-element block j of the first half): x = row[j], y = row[half + j], both rotated with the
// expressions of rope_and_cache_kernel (bit-identical), written back and -- for k -- into the cache row. The scalar kernel moved
// 2 bytes per access: 39 us for the 159 MB of a Qwen2-7B prefill chunk (4 TB/s); nt = prefill-sized tensors are streamed.
template <typename T>
__global__ __launch_bounds__(256) void rope_and_cache_vec_kernel(
    const int64_t* __restrict__ positions, T* __restrict__ q, T* __restrict__ k, const T* __restrict__ v,
    const T* __restrict__ cache, const int32_t* __restrict__ slot_ids, T* __restrict__ kc, T* __restrict__ vc,
    int64_t q_stride, int64_t k_stride, int64_t v_stride, int head_size, int nq, int nk, int64_t block_size,
    int64_t n_blocks, bool nt) {
  static_assert(sizeof(T) == 2, "16-bit rows");
  const int64_t t = blockIdx.x;
  const int half = head_size >> 1, hb = half >> 3;
  const T* cp = cache + positions[t] * head_size;
  const int64_t slot = slot_ids[t];
  const bool store = slot >= 0 && slot / block_size < n_blocks;
  T* kc_row = kc + slot * (int64_t)nk * head_size;
  T* vc_row = vc + slot * (int64_t)nk * head_size;
  const int total = (nq + nk) * hb;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int h = i / hb, j0 = (i - h * hb) * 8;
    const bool is_k = h >= nq;
    T* arr = is_k ? k + t * k_stride + (int64_t)(h - nq) * head_size : q + t * q_stride + (int64_t)h * head_size;
    RowVec<T> x, y, c, sn, nx, ny;
    x.raw = rw_ld16(arr + j0, nt);
    y.raw = rw_ld16(arr + half + j0, nt);
    c.raw = *reinterpret_cast<const uint4*>(cp + j0);
    sn.raw = *reinterpret_cast<const uint4*>(cp + half + j0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xf = x.get(e), yf = y.get(e), cf = c.get(e), sf = sn.get(e);
      nx.set(e, r16<T>(xf * cf) - r16<T>(yf * sf));
      ny.set(e, r16<T>(yf * cf) + r16<T>(xf * sf));
    }
    rw_st16(arr + j0, nx.raw, nt);
    rw_st16(arr + half + j0, ny.raw, nt);
    if (is_k && store) {
      rw_st16(kc_row + (h - nq) * head_size + j0, nx.raw, nt);
      rw_st16(kc_row + (h - nq) * head_size + half + j0, ny.raw, nt);
    }
  }
  if (!store) return;
  const int nv = nk * head_size / 8;
  const T* vs = v + t * v_stride;
  for (int i = threadIdx.x; i < nv; i += 256) rw_st16(vc_row + i * 8, rw_ld16(vs + i * 8, nt), nt);
}

// N1 fusion across the GEMM boundary: the qkv projection's dequant epilogue + RoPE + KV write in ONE pass over the token's row.
// The packed-weight GEMM (gemm_ws.hip) leaves exact int32 K-slice slabs; this kernel adds them, applies the scaled_matmul
// epilogue (r16(acc * a_s[t] * w_s[n] + bias[n]): the 16-bit qkv row the reference's linear would have written,
// linear.cpp:481-507), rotates q and k with the arithmetic of rope_and_cache_kernel, writes the packed qkv
// row (q is the attention's input) and scatters the rotated k and v to the caches. Bit-identical to scaled_matmul ->
// rotary_embedding -> reshape_paged_cache; one launch and one round trip of the qkv row instead of three.
template <typename T, bool NEOX>
__global__ __launch_bounds__(256) void slab_rope_and_cache_kernel(
    const int32_t* __restrict__ slabs, int n_slabs, int64_t slab_stride, const float* __restrict__ a_scale,
    const float* __restrict__ w_scale, const T* __restrict__ bias, T* __restrict__ qkv, int n_cols,
    const int64_t* __restrict__ positions, const T* __restrict__ cache, const int32_t* __restrict__ slot_ids,
    T* __restrict__ kc, T* __restrict__ vc, int rot_dim, int head_size, int nq, int nk, int64_t block_size, int64_t n_blocks) {
  // no row-wide quantity is needed, so the row is cut into independent work items -- one RoPE pair of a q / k head, or one
  // un-rotated element (tails of q / k, all of v) -- and spread over blockIdx.y: a decode batch of 32 tokens still fills the chip
  // (one workgroup per token made this launch as slow as the two it replaces: measured, profiles/r02_fusions.txt)
  const int64_t t = blockIdx.x;
  const int half = rot_dim >> 1, tail = head_size - rot_dim;
  const int n_pairs = (nq + nk) * half, n_tail = (nq + nk) * tail, n_v = nk * head_size;
  const int item = blockIdx.y * blockDim.x + threadIdx.x;
  if (item >= n_pairs + n_tail + n_v) return;
  const float as = a_scale[t];
  const int32_t* acc_row = slabs + t * (int64_t)n_cols;
  auto value = [&](int c) -> float {   // the 16-bit qkv element the GEMM epilogue would have written
    int a = acc_row[c];
    for (int sl = 1; sl < n_slabs; ++sl) a += acc_row[sl * slab_stride + c];
    return r16<T>((float)a * as * w_scale[c] + (bias ? to_f32(bias[c]) : 0.0f));
  };
  const int64_t slot = slot_ids[t];
  const bool store = slot >= 0 && slot / block_size < n_blocks;
  T* kc_row = kc + slot * (int64_t)nk * head_size;
  T* vc_row = vc + slot * (int64_t)nk * head_size;
  T* out_row = qkv + t * (int64_t)n_cols;
  if (item < n_pairs) {
    const int h = item / half, j = item - h * half;
    const int base = h * head_size;                 // q heads then k heads are contiguous in the packed row
    const int xi = NEOX ? j : 2 * j, yi = NEOX ? half + j : 2 * j + 1;
    const T* cp = cache + positions[t] * rot_dim;
    const float c = to_f32(cp[j]), sn = to_f32(cp[half + j]);
    const float x = value(base + xi), y = value(base + yi);
    const T nx = from_f32<T>(r16<T>(x * c) - r16<T>(y * sn));
    const T ny = from_f32<T>(r16<T>(y * c) + r16<T>(x * sn));
    out_row[base + xi] = nx;
    out_row[base + yi] = ny;
    if (h >= nq && store) {
      kc_row[(h - nq) * head_size + xi] = nx;
      kc_row[(h - nq) * head_size + yi] = ny;
    }
  } else if (item < n_pairs + n_tail) {             // un-rotated tail of a q / k head (rot_dim < head_size)
    const int i2 = item - n_pairs;
    const int h = i2 / tail, e = rot_dim + (i2 - h * tail);
    const T v = from_f32<T>(value(h * head_size + e));
    out_row[h * head_size + e] = v;
    if (h >= nq && store) kc_row[(h - nq) * head_size + e] = v;
  } else {                                          // v
    const int i2 = item - n_pairs - n_tail;
    const int c = (nq + nk) * head_size + i2;
    const T v = from_f32<T>(value(c));
    out_row[c] = v;
    if (store) vc_row[i2] = v;
  }
}

// The same operator four elements per thread (round 3): 16-byte slab loads with all slabs of a group in flight, 8-byte stores.
// Items: four consecutive RoPE pairs of a q / k head, four tail elements, or four elements of v. Needs rot_dim / 2, the tail and
// head_size to be multiples of 4 (the launcher falls back to the scalar kernel otherwise). Same arithmetic, bit for bit.
template <typename T, bool NEOX>
__global__ __launch_bounds__(256) void slab_rope_and_cache_vec_kernel(
    const int32_t* __restrict__ slabs, int n_slabs, int64_t slab_stride, const float* __restrict__ a_scale,
    const float* __restrict__ w_scale, const T* __restrict__ bias, T* __restrict__ qkv, int n_cols,
    const int64_t* __restrict__ positions, const T* __restrict__ cache, const int32_t* __restrict__ slot_ids,
    T* __restrict__ kc, T* __restrict__ vc, int rot_dim, int head_size, int nq, int nk, int64_t block_size, int64_t n_blocks) {
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const int64_t t = blockIdx.x;
  const int half = rot_dim >> 1, tail = head_size - rot_dim;
  const int n_pq = (nq + nk) * half / 4, n_tq = (nq + nk) * tail / 4, n_vq = nk * head_size / 4;
  const int item = blockIdx.y * blockDim.x + threadIdx.x;
  if (item >= n_pq + n_tq + n_vq) return;
  const float as = a_scale[t];
  const int32_t* acc_row = slabs + t * (int64_t)n_cols;
  // the four 16-bit qkv elements c .. c + 3 the GEMM epilogue would have written, in two steps (round 6): load4 REQUESTS everything
  // (slab 0, slabs 1..7, scales, bias), finish4 consumes -- a thread issues the loads of both of its column groups and the cos / sin
  // pair before it waits for any of them (before: slabs -> scales + bias, once per group: four dependent memory latencies)
  struct Pending4 {
    i32x4 a, b[7];
    float4 w;
    uint2 braw;
    int c;
  };
  auto load4 = [&](int c) {
    Pending4 p;
    p.c = c;
    p.a = *reinterpret_cast<const i32x4*>(acc_row + c);
#pragma unroll
    for (int u = 0; u < 7; ++u) {
      p.b[u] = i32x4{0, 0, 0, 0};
      if (1 + u < n_slabs) p.b[u] = *reinterpret_cast<const i32x4*>(acc_row + (1 + u) * slab_stride + c);
    }
    p.w = *reinterpret_cast<const float4*>(w_scale + c);
    p.braw = bias ? *reinterpret_cast<const uint2*>(bias + c) : make_uint2(0u, 0u);
    return p;
  };
  auto finish4 = [&](const Pending4& p, float (&v)[4]) {
    i32x4 a = p.a + (((p.b[0] + p.b[1]) + (p.b[2] + p.b[3])) + ((p.b[4] + p.b[5]) + p.b[6]));
    for (int sl = 8; sl < n_slabs; ++sl) a += *reinterpret_cast<const i32x4*>(acc_row + sl * slab_stride + p.c);   // (no plan makes > 8)
    const float wv[4] = {p.w.x, p.w.y, p.w.z, p.w.w};
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
      T b4[4];
      __builtin_memcpy(b4, &p.braw, 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = to_f32(b4[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = r16<T>((float)a[e] * as * wv[e] + bv[e]);
  };
  auto store4 = [](T* dst, const T (&v)[4]) {
    uint2 raw;
    __builtin_memcpy(&raw, v, 8);
    *reinterpret_cast<uint2*>(dst) = raw;
  };
  const int64_t slot = slot_ids[t];
  const bool store = slot >= 0 && slot / block_size < n_blocks;
  T* kc_row = kc + slot * (int64_t)nk * head_size;
  T* vc_row = vc + slot * (int64_t)nk * head_size;
  T* out_row = qkv + t * (int64_t)n_cols;
  if (item < n_pq) {
    const int hq = half / 4;
    const int h = item / hq, j = (item - h * hq) * 4;   // pairs j .. j + 3 of head h
    const int base = h * head_size;
    const Pending4 p0 = load4(NEOX ? base + j : base + 2 * j), p1 = load4(NEOX ? base + half + j : base + 2 * j + 4);
    const T* cp = cache + positions[t] * rot_dim;
    uint2 craw = *reinterpret_cast<const uint2*>(cp + j), sraw = *reinterpret_cast<const uint2*>(cp + half + j);
    T c4[4], s4[4];
    __builtin_memcpy(c4, &craw, 8);
    __builtin_memcpy(s4, &sraw, 8);
    float x[4], y[4];
    T nx[4], ny[4];
    if constexpr (NEOX) {
      finish4(p0, x);
      finish4(p1, y);
    } else {                       // interleaved: elements 2 j .. 2 j + 7 = (x0 y0 x1 y1 | x2 y2 x3 y3)
      float lo[4], hi[4];
      finish4(p0, lo);
      finish4(p1, hi);
      x[0] = lo[0]; y[0] = lo[1]; x[1] = lo[2]; y[1] = lo[3];
      x[2] = hi[0]; y[2] = hi[1]; x[3] = hi[2]; y[3] = hi[3];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float c = to_f32(c4[e]), sn = to_f32(s4[e]);
      nx[e] = from_f32<T>(r16<T>(x[e] * c) - r16<T>(y[e] * sn));
      ny[e] = from_f32<T>(r16<T>(y[e] * c) + r16<T>(x[e] * sn));
    }
    if constexpr (NEOX) {
      store4(out_row + base + j, nx);
      store4(out_row + base + half + j, ny);
      if (h >= nq && store) {
        store4(kc_row + (h - nq) * head_size + j, nx);
        store4(kc_row + (h - nq) * head_size + half + j, ny);
      }
    } else {
      const T lo[4] = {nx[0], ny[0], nx[1], ny[1]}, hi[4] = {nx[2], ny[2], nx[3], ny[3]};
      store4(out_row + base + 2 * j, lo);
      store4(out_row + base + 2 * j + 4, hi);
      if (h >= nq && store) {
        store4(kc_row + (h - nq) * head_size + 2 * j, lo);
        store4(kc_row + (h - nq) * head_size + 2 * j + 4, hi);
      }
    }
  } else if (item < n_pq + n_tq) {                  // un-rotated tail of a q / k head (rot_dim < head_size)
    const int i2 = item - n_pq, tq = tail / 4;
    const int h = i2 / tq, e = rot_dim + (i2 - h * tq) * 4;
    float v[4];
    finish4(load4(h * head_size + e), v);
    const T o[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
    store4(out_row + h * head_size + e, o);
    if (h >= nq && store) store4(kc_row + (h - nq) * head_size + e, o);
  } else {                                          // v
    const int i2 = (item - n_pq - n_tq) * 4;
    const int c = (nq + nk) * head_size + i2;
    float v[4];
    finish4(load4(c), v);
    const T o[4] = {from_f32<T>(v[0]), from_f32<T>(v[1]), from_f32<T>(v[2]), from_f32<T>(v[3])};
    store4(out_row + c, o);
    if (store) store4(vc_row + i2, o);
  }
}

int launch_slab_rope_and_cache(const int32_t* slabs, int n_slabs, const float* a_scale, const float* w_scale,
                               const void* bias, void* qkv, int64_t M, int64_t N, const int64_t* positions,
                               const void* cos_sin_cache, const int32_t* slot_ids, void* k_cache, void* v_cache,
                               int64_t n_q_heads, int64_t n_kv_heads, int64_t head_size, int64_t rot_dim, int64_t block_size,
                               int64_t n_blocks, int is_neox, int dtype, hipStream_t s) {
  if (N != (n_q_heads + 2 * n_kv_heads) * head_size || N % 4 || N * 2 > 64 * 1024 || rot_dim <= 0 || (rot_dim & 1) ||
      rot_dim > head_size || ((uintptr_t)slabs % 16) || ((uintptr_t)w_scale % 16))
    return XM_ERR_UNSUPPORTED;
  const int64_t items = (n_q_heads + n_kv_heads) * (rot_dim / 2) + (n_q_heads + n_kv_heads) * (head_size - rot_dim) +
                        n_kv_heads * head_size;
  const int64_t half = rot_dim / 2, tail = head_size - rot_dim;
  // (the scalar kernel below is the path of odd head geometries / unaligned tensors; forcing it, XLLM_MI355_SLAB_ROPE_VEC=0, is a
  // tuning arm of the -DXM_TUNING flavour)
  XM_TUNE_VAR(use_vec, "XLLM_MI355_SLAB_ROPE_VEC", 1);
  if (use_vec && half % 4 == 0 && tail % 4 == 0 && head_size % 4 == 0 && (!bias || (uintptr_t)bias % 8 == 0) &&
      (uintptr_t)qkv % 8 == 0 && (uintptr_t)k_cache % 8 == 0 && (uintptr_t)v_cache % 8 == 0 && (uintptr_t)cos_sin_cache % 8 == 0) {
    const dim3 gridv((unsigned)M, (unsigned)((items / 4 + 255) / 256));
    XM_DISPATCH_HALF(dtype, T, {
      if (is_neox)
        hipLaunchKernelGGL((slab_rope_and_cache_vec_kernel<T, true>), gridv, dim3(256), 0, s, slabs, n_slabs, M * N,
                           a_scale, w_scale, (const T*)bias, (T*)qkv, (int)N, positions, (const T*)cos_sin_cache, slot_ids,
                           (T*)k_cache, (T*)v_cache, (int)rot_dim, (int)head_size, (int)n_q_heads, (int)n_kv_heads, block_size,
                           n_blocks);
      else
        hipLaunchKernelGGL((slab_rope_and_cache_vec_kernel<T, false>), gridv, dim3(256), 0, s, slabs, n_slabs, M * N,
                           a_scale, w_scale, (const T*)bias, (T*)qkv, (int)N, positions, (const T*)cos_sin_cache, slot_ids,
                           (T*)k_cache, (T*)v_cache, (int)rot_dim, (int)head_size, (int)n_q_heads, (int)n_kv_heads, block_size,
                           n_blocks);
    });
    return hip_check_launch();
  }
  const dim3 grid((unsigned)M, (unsigned)((items + 255) / 256));
  XM_DISPATCH_HALF(dtype, T, {
    if (is_neox)
      hipLaunchKernelGGL((slab_rope_and_cache_kernel<T, true>), grid, dim3(256), 0, s, slabs, n_slabs, M * N,
                         a_scale, w_scale, (const T*)bias, (T*)qkv, (int)N, positions, (const T*)cos_sin_cache, slot_ids,
                         (T*)k_cache, (T*)v_cache, (int)rot_dim, (int)head_size, (int)n_q_heads, (int)n_kv_heads, block_size,
                         n_blocks);
    else
      hipLaunchKernelGGL((slab_rope_and_cache_kernel<T, false>), grid, dim3(256), 0, s, slabs, n_slabs, M * N,
                         a_scale, w_scale, (const T*)bias, (T*)qkv, (int)N, positions, (const T*)cos_sin_cache, slot_ids,
                         (T*)k_cache, (T*)v_cache, (int)rot_dim, (int)head_size, (int)n_q_heads, (int)n_kv_heads, block_size,
                         n_blocks);
  });
  return hip_check_launch();
}

// fused per-head RMSNorm + RoPE inside packed qkv (reference: kernels/cuda/fused_qknorm_rope.cu:88-300)
// one wave per (token, head); fp32 math, one 16-bit store. head_dim <= 256, multiple of 2.
template <typename T, typename CT>
__global__ __launch_bounds__(256) void fused_qk_norm_rope_kernel(T* __restrict__ qkv, int64_t n_tokens, int nq,
                                                                 int nk, int nv, int d, float eps,
                                                                 const T* __restrict__ qw, const T* __restrict__ kw,
                                                                 const CT* __restrict__ cache, int interleaved,
                                                                 const int64_t* __restrict__ positions) {
  const int lane = threadIdx.x & 63;
  const int64_t wid = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int heads = nq + nk;
  if (wid >= n_tokens * heads) return;
  const int64_t t = wid / heads;
  const int h = (int)(wid - t * heads);
  T* base = qkv + t * (int64_t)(nq + nk + nv) * d + (int64_t)h * d;
  const T* w = (h < nq) ? qw : kw;
  const int half = d >> 1;
  // each lane owns pairs j = lane, lane+64, ... (x_j, y_j)
  float xs[2], ys[2];
  float ss = 0.0f;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int j = lane + r * 64;
    if (j < half) {
      const int xi = interleaved ? 2 * j : j, yi = interleaved ? 2 * j + 1 : half + j;
      xs[r] = to_f32(base[xi]);
      ys[r] = to_f32(base[yi]);
      ss += xs[r] * xs[r] + ys[r] * ys[r];
    }
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / sqrtf(ss / (float)d + eps);
  const CT* cp = cache + positions[t] * d;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const int j = lane + r * 64;
    if (j < half) {
      const int xi = interleaved ? 2 * j : j, yi = interleaved ? 2 * j + 1 : half + j;
      const float x = xs[r] * inv * to_f32(w[xi]), y = ys[r] * inv * to_f32(w[yi]);
      const float c = to_f32(cp[j]), s = to_f32(cp[half + j]);
      base[xi] = from_f32<T>(x * c - y * s);
      base[yi] = from_f32<T>(y * c + x * s);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// act_and_mul (reference: kernels/cuda/activation.cu:49-120): out = r16(r16(act(float(x))) * y)
// QUANT: also per-token int8 quantise the 16-bit result
// ------------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(256) void act_and_mul_kernel(T* __restrict__ out, const T* __restrict__ in, int d,
                                                          bool vec) {
  constexpr int N = RowVec<T>::N;
  const int64_t t = blockIdx.x;
  const T* x = in + t * 2 * (int64_t)d;
  const T* y = x + d;
  T* o = out + t * (int64_t)d;
  if (vec) {
    const int nvec = d / N;
    // five chunk pairs of a thread requested before the first is consumed (round 6: one pair per loop iteration was ten dependent
    // round trips per row at d = 18944: 8.3 us for [256, 2 x 18944]); native vector registers (HIP uint4 arrays spill, see below)
    typedef unsigned avec_t __attribute__((ext_vector_type(4)));
    constexpr int U = 5;
    // gridDim.y workgroups share a row (elementwise: no dependency inside a row); few rows (decode) then still put enough bytes in
    // flight per CU -- one 256-thread workgroup per 113-KB row pair kept 40 KB in flight and the launch at 3.5 TB/s
    for (int c0 = blockIdx.y * U * blockDim.x + threadIdx.x; c0 < nvec; c0 += gridDim.y * U * blockDim.x) {
      avec_t xr[U], yr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * blockDim.x;
        const int cc = c < nvec ? c : c0;
        xr[u] = reinterpret_cast<const avec_t*>(x)[cc];
        yr[u] = reinterpret_cast<const avec_t*>(y)[cc];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * blockDim.x;
        if (c >= nvec) break;
        RowVec<T> xv, yv;
        xv.raw = make_uint4(xr[u][0], xr[u][1], xr[u][2], xr[u][3]);
        yv.raw = make_uint4(yr[u][0], yr[u][1], yr[u][2], yr[u][3]);
#pragma unroll
        for (int j = 0; j < N; ++j) xv.set(j, r16<T>(act_f<MODE>(xv.get(j))) * yv.get(j));
        reinterpret_cast<uint4*>(o)[c] = xv.raw;
      }
    }
  } else {
    for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < d; i += gridDim.y * blockDim.x)
      o[i] = from_f32<T>(r16<T>(act_f<MODE>(to_f32(x[i]))) * to_f32(y[i]));
  }
}

// fused act_and_mul + per-token int8 quant, register-resident: a row is one workgroup, every thread issues ALL of
// its loads (VPT x 2 x 16 B, unconditional, clamped index) before the first use, so a row has its whole 4*d bytes in
// flight at once (76 KiB at d = 18944: one HBM round trip per row instead of one per loop iteration); the 16-bit
// products stay packed in registers across the block-wide amax reduction. Native ext_vector registers only: the
// first attempt at this kept HIP uint4 structs in arrays, which hipcc spilled to scratch (2x slower).
template <typename T>
__device__ __forceinline__ float half_bits_to_f32(uint32_t h) {
  if constexpr (__is_same(T, bf16_t)) return __uint_as_float(h << 16);
  else { const uint16_t hh = (uint16_t)h; f16_t x; __builtin_memcpy(&x, &hh, 2); return (float)x; }
}
template <typename T>
__device__ __forceinline__ uint32_t f32_to_half_bits(float f) {
  if constexpr (__is_same(T, bf16_t)) return f32_to_bf16_bits(f);
  else { const f16_t x = (f16_t)f; uint16_t h; __builtin_memcpy(&h, &x, 2); return h; }
}
// expert-parallel rank: only the first sum(live_sizes) sorted rows exist (device-side count, no host read). Every wave sums the
// sizes itself (n_sizes is a few hundred at most). A row past the count is not read; its scale is written as 0 so that nothing
// downstream can pick up a NaN from uninitialised memory, its quantised bytes are left as they are -- the same for every kernel.
__device__ __forceinline__ bool row_is_dead(const int32_t* __restrict__ live_sizes, int n_sizes, int64_t t, int lane) {
  if (!live_sizes) return false;
  int part = 0;
  for (int e = lane; e < n_sizes; e += 64) part += live_sizes[e];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
  return t >= part;
}

template <typename T, int MODE, int VPT>
__global__ __launch_bounds__(512) void act_and_mul_i8_reg_kernel(int8_t* __restrict__ out_q, float* __restrict__ out_s,
                                                                 const T* __restrict__ in, int d,
                                                                 const int32_t* __restrict__ live_sizes = nullptr, int n_sizes = 0) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __shared__ float red[32];
  const int64_t t = blockIdx.x;
  if (row_is_dead(live_sizes, n_sizes, t, threadIdx.x & 63)) {   // (uniform over the workgroup: one row per workgroup)
    if (threadIdx.x == 0) out_s[t] = 0.0f;
    return;
  }
  const int nvec = d / 8;
  const u32x4* x = reinterpret_cast<const u32x4*>(in + t * 2 * (int64_t)d);
  const u32x4* y = x + nvec;
  u32x4 xv[VPT], yv[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    int c = threadIdx.x + i * 512;
    c = c < nvec ? c : nvec - 1;
    xv[i] = x[c];
    yv[i] = y[c];
  }
  float amax = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const bool live = (int)threadIdx.x + i * 512 < nvec;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t xw = xv[i][w], yw = yv[i][w];
      const float r0 = r16<T>(r16<T>(act_f<MODE>(half_bits_to_f32<T>(xw & 0xffffu))) * half_bits_to_f32<T>(yw & 0xffffu));
      const float r1 = r16<T>(r16<T>(act_f<MODE>(half_bits_to_f32<T>(xw >> 16))) * half_bits_to_f32<T>(yw >> 16));
      xv[i][w] = f32_to_half_bits<T>(r0) | (f32_to_half_bits<T>(r1) << 16);
      amax = fmaxf(amax, live ? fmaxf(fabsf(r0), fabsf(r1)) : 0.0f);
    }
  }
  amax = block_max(amax, red);
  const float qinv = (amax > 1e-10f) ? 127.0f / amax : 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = threadIdx.x + i * 512;
    uint32_t pk[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint32_t wq = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t word = xv[i][h * 2 + (e >> 1)];
        const float r = half_bits_to_f32<T>((e & 1) ? (word >> 16) : (word & 0xffffu));
        const float qv = fmaxf(-127.0f, fminf(127.0f, rintf(r * qinv)));
        wq |= ((uint32_t)(int)qv & 0xffu) << (8 * e);
      }
      pk[h] = wq;
    }
    if (c < nvec) *reinterpret_cast<uint2*>(out_q + t * (int64_t)d + (int64_t)c * 8) = make_uint2(pk[0], pk[1]);
  }
  if (threadIdx.x == 0) out_s[t] = amax / 127.0f;
}

// short rows (d <= 1024: MoE expert widths): ONE WAVE per row, 8 rows per workgroup -- a 512-thread workgroup per row
// leaves 7 of 8 waves idle at d = 768 and the launch becomes row-count bound (65536 rows: 405 us; this kernel: ~50 us)
template <typename T, int MODE, int VPT>
__global__ __launch_bounds__(512) void act_and_mul_i8_wave_kernel(int8_t* __restrict__ out_q, float* __restrict__ out_s,
                                                                  const T* __restrict__ in, int d, int64_t n_rows,
                                                                  const int32_t* __restrict__ live_sizes, int n_sizes) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 6);
  if (t >= n_rows) return;
  if (row_is_dead(live_sizes, n_sizes, t, lane)) {
    if (lane == 0) out_s[t] = 0.0f;
    return;
  }
  const int nvec = d / 8;
  const u32x4* x = reinterpret_cast<const u32x4*>(in + t * 2 * (int64_t)d);
  const u32x4* y = x + nvec;
  u32x4 xv[VPT], yv[VPT];
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    int c = lane + i * 64;
    c = c < nvec ? c : nvec - 1;
    xv[i] = x[c];
    yv[i] = y[c];
  }
  float amax = 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const bool live = lane + i * 64 < nvec;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const uint32_t xw = xv[i][w], yw = yv[i][w];
      const float r0 = r16<T>(r16<T>(act_f<MODE>(half_bits_to_f32<T>(xw & 0xffffu))) * half_bits_to_f32<T>(yw & 0xffffu));
      const float r1 = r16<T>(r16<T>(act_f<MODE>(half_bits_to_f32<T>(xw >> 16))) * half_bits_to_f32<T>(yw >> 16));
      xv[i][w] = f32_to_half_bits<T>(r0) | (f32_to_half_bits<T>(r1) << 16);
      amax = fmaxf(amax, live ? fmaxf(fabsf(r0), fabsf(r1)) : 0.0f);
    }
  }
  amax = wave_max(amax);
  const float qinv = (amax > 1e-10f) ? 127.0f / amax : 0.0f;
#pragma unroll
  for (int i = 0; i < VPT; ++i) {
    const int c = lane + i * 64;
    uint32_t pk[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint32_t wq = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t word = xv[i][h * 2 + (e >> 1)];
        const float r = half_bits_to_f32<T>((e & 1) ? (word >> 16) : (word & 0xffffu));
        const float qv = fmaxf(-127.0f, fminf(127.0f, rintf(r * qinv)));
        wq |= ((uint32_t)(int)qv & 0xffu) << (8 * e);
      }
      pk[h] = wq;
    }
    if (c < nvec) *reinterpret_cast<uint2*>(out_q + t * (int64_t)d + (int64_t)c * 8) = make_uint2(pk[0], pk[1]);
  }
  if (lane == 0) out_s[t] = amax / 127.0f;
}

// LDS-staged variant for rows too long for the register kernel (d up to 32768 elements of 2 bytes = 64 KiB)
template <typename T, int MODE>
__global__ __launch_bounds__(512) void act_and_mul_i8_kernel(int8_t* __restrict__ out_q, float* __restrict__ out_s,
                                                             const T* __restrict__ in, int d,
                                                             const int32_t* __restrict__ live_sizes = nullptr, int n_sizes = 0) {
  extern __shared__ __attribute__((aligned(16))) char dyn_smem[];
  __shared__ float red[32];
  constexpr int N = RowVec<T>::N;
  T* stage = reinterpret_cast<T*>(dyn_smem);
  const int64_t t = blockIdx.x;
  if (row_is_dead(live_sizes, n_sizes, t, threadIdx.x & 63)) {
    if (threadIdx.x == 0) out_s[t] = 0.0f;
    return;
  }
  const T* x = in + t * 2 * (int64_t)d;
  const T* y = x + d;
  const int nvec = d / N;
  float amax = 0.0f;
  for (int c = threadIdx.x; c < nvec; c += blockDim.x) {
    RowVec<T> xv, yv;
    xv.raw = reinterpret_cast<const uint4*>(x)[c];
    yv.raw = reinterpret_cast<const uint4*>(y)[c];
#pragma unroll
    for (int j = 0; j < N; ++j) {
      float r = r16<T>(r16<T>(act_f<MODE>(xv.get(j))) * yv.get(j));
      xv.set(j, r);
      amax = fmaxf(amax, fabsf(r));
    }
    reinterpret_cast<uint4*>(stage)[c] = xv.raw;
  }
  amax = block_max(amax, red);
  const float qinv = (amax > 1e-10f) ? 127.0f / amax : 0.0f;
  __syncthreads();
  for (int c = threadIdx.x; c < nvec; c += blockDim.x) {
    RowVec<T> xv;
    xv.raw = reinterpret_cast<const uint4*>(stage)[c];
    uint32_t pk[N / 4];
#pragma unroll
    for (int j = 0; j < N; j += 4) {
      uint32_t w = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float qv = fmaxf(-127.0f, fminf(127.0f, rintf(xv.get(j + e) * qinv)));
        w |= ((uint32_t)(int)qv & 0xffu) << (8 * e);
      }
      pk[j / 4] = w;
    }
    int8_t* o = out_q + t * (int64_t)d + (int64_t)c * N;
    if constexpr (N == 8) *reinterpret_cast<uint2*>(o) = make_uint2(pk[0], pk[1]);
    else *reinterpret_cast<uint32_t*>(o) = pk[0];
  }
  if (threadIdx.x == 0) out_s[t] = amax / 127.0f;
}

// ------------------------------------------------------------------------------------------------
// per-token int8 quant (reference: kernels/dcu/scaled_quantize.hip:29-109)
// the row is read once (register cache up to 16 KiB/row of 16-bit data at 512 threads, else re-read)
// ------------------------------------------------------------------------------------------------
// TH threads per row: 512 for a few long rows (decode: latency), 128 when there are thousands of short ones (prefill: a 7-KiB row per
// 512-thread workgroup left ~7 MB in flight chip-wide and the kernel at 4 TB/s -- round 4); nt = prefill-sized input, streamed
template <typename T, int TH>
__global__ __launch_bounds__(TH) void scaled_quantize_i8_kernel(const T* __restrict__ x, int8_t* __restrict__ out,
                                                                float* __restrict__ scales, int K, bool vec, bool nt) {
  __shared__ float red[32];
  constexpr int N = RowVec<T>::N;
  // chunks cached per thread: 512 x 5 x 8 covers the longest decode row (Qwen2-7B's 18944) in ONE round trip -- with 4 the last 2560
  // elements were a second (and, for the quantising pass, a third) dependent latency of a 5-us launch (round 6)
  constexpr int kCache = TH >= 512 ? 5 : 4;
  const int64_t m = blockIdx.x;
  const T* xr = x + m * (int64_t)K;
  int8_t* orow = out + m * (int64_t)K;
  float amax = 0.0f;
  if (vec) {
    const int nvec = K / N;
    RowVec<T> xv[kCache];
#pragma unroll
    for (int i = 0; i < kCache; ++i) {          // all requests first
      const int c = threadIdx.x + i * TH;
      xv[i].raw = make_uint4(0u, 0u, 0u, 0u);
      if (c < nvec) xv[i].raw = rw_ld16(reinterpret_cast<const uint4*>(xr) + c, nt);
    }
#pragma unroll
    for (int i = 0; i < kCache; ++i) {
      const int c = threadIdx.x + i * TH;
      if (c < nvec) {
#pragma unroll
        for (int j = 0; j < N; ++j) amax = fmaxf(amax, fabsf(xv[i].get(j)));
      }
    }
    for (int c = threadIdx.x + kCache * TH; c < nvec; c += TH) {
      RowVec<T> v;
      v.raw = reinterpret_cast<const uint4*>(xr)[c];
#pragma unroll
      for (int j = 0; j < N; ++j) amax = fmaxf(amax, fabsf(v.get(j)));
    }
    amax = block_max(amax, red);
    const float qinv = (amax > 1e-10f) ? 127.0f / amax : 0.0f;
    auto emit = [&](const RowVec<T>& v, int c) {
      uint32_t pk[N / 4];
#pragma unroll
      for (int j = 0; j < N; j += 4) {
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float qv = fmaxf(-127.0f, fminf(127.0f, rintf(v.get(j + e) * qinv)));
          w |= ((uint32_t)(int)qv & 0xffu) << (8 * e);
        }
        pk[j / 4] = w;
      }
      int8_t* o = orow + (int64_t)c * N;
      if constexpr (N == 8) *reinterpret_cast<uint2*>(o) = make_uint2(pk[0], pk[1]);
      else *reinterpret_cast<uint32_t*>(o) = pk[0];
    };
#pragma unroll
    for (int i = 0; i < kCache; ++i) {
      const int c = threadIdx.x + i * TH;
      if (c < nvec) emit(xv[i], c);
    }
    for (int c = threadIdx.x + kCache * TH; c < nvec; c += TH) {
      RowVec<T> v;
      v.raw = reinterpret_cast<const uint4*>(xr)[c];
      emit(v, c);
    }
  } else {
    for (int i = threadIdx.x; i < K; i += blockDim.x) amax = fmaxf(amax, fabsf(to_f32(xr[i])));
    amax = block_max(amax, red);
    const float qinv = (amax > 1e-10f) ? 127.0f / amax : 0.0f;
    for (int i = threadIdx.x; i < K; i += blockDim.x)
      orow[i] = (int8_t)fmaxf(-127.0f, fminf(127.0f, rintf(to_f32(xr[i]) * qinv)));
  }
  if (threadIdx.x == 0) scales[m] = amax / 127.0f;
}

// ------------------------------------------------------------------------------------------------
// fp8 static quant + per-tensor amax (reference: kernels/cuda/fp8_quant.cu:79-155,
// fp8_scaled_quantize.cpp:36-47). grid-stride, 16-B loads.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void fp8_quant_kernel(uint8_t* __restrict__ out, const T* __restrict__ in,
                                                        const float* __restrict__ scale, int64_t n, bool vec) {
  constexpr int N = RowVec<T>::N;
  const float sinv = 1.0f / scale[0];
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
  if (vec) {
    const int64_t nvec = n / N;
    for (int64_t c = tid; c < nvec; c += nthr) {
      RowVec<T> v;
      v.raw = reinterpret_cast<const uint4*>(in)[c];
      uint32_t pk[N / 4];
#pragma unroll
      for (int j = 0; j < N; j += 4) {
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) w |= (uint32_t)f32_to_e4m3_sat(v.get(j + e) * sinv) << (8 * e);
        pk[j / 4] = w;
      }
      if constexpr (N == 8) reinterpret_cast<uint2*>(out)[c] = make_uint2(pk[0], pk[1]);
      else reinterpret_cast<uint32_t*>(out)[c] = pk[0];
    }
    for (int64_t i = nvec * N + tid; i < n; i += nthr) out[i] = f32_to_e4m3_sat(to_f32(in[i]) * sinv);
  } else {
    for (int64_t i = tid; i < n; i += nthr) out[i] = f32_to_e4m3_sat(to_f32(in[i]) * sinv);
  }
}

// amax over the tensor -> scale = r16-chain of fp8_scaled_quantize.cpp:39-44 evaluated on a 16-bit
// 0-dim tensor: s = r16(amax/448); s = r16(max(s, r16(1e-12))). Non-negative floats order like uints,
// so the cross-block max is an atomicMax on the bit pattern (scale buffer zeroed by the launcher).
template <typename T>
__global__ __launch_bounds__(256) void amax_kernel(const T* __restrict__ in, int64_t n, uint32_t* __restrict__ amax_bits) {
  __shared__ float red[32];
  float m = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(to_f32(in[i])));
  m = block_max(m, red);
  if (threadIdx.x == 0) atomicMax(amax_bits, __float_as_uint(m));
}
// Round 6: the dynamic per-tensor quant in TWO launches instead of four graph nodes (memset, amax with atomics, finalize, quant: ~19 us
// for a [128, 1536] decode operand): amax_partial_kernel leaves one maximum per block in a scratch buffer (plain stores, nothing to
// zero), fp8_quant_dyn_kernel folds them (<= 1024 floats from the L2) in every block, evaluates fp8_scale_finalize_kernel's
// expression itself and quantises; block 0 writes the scale. Same bits as the four-node form.
template <typename T>
__global__ __launch_bounds__(256) void amax_partial_kernel(const T* __restrict__ in, int64_t n, float* __restrict__ part, bool vec) {
  __shared__ float red[32];
  constexpr int N = RowVec<T>::N;
  float m = 0.0f;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
  if (vec) {
    const int64_t nvec = n / N;
    for (int64_t c = tid; c < nvec; c += nthr) {
      RowVec<T> v;
      v.raw = reinterpret_cast<const uint4*>(in)[c];
#pragma unroll
      for (int j = 0; j < N; ++j) m = fmaxf(m, fabsf(v.get(j)));
    }
    for (int64_t i = nvec * N + tid; i < n; i += nthr) m = fmaxf(m, fabsf(to_f32(in[i])));
  } else {
    for (int64_t i = tid; i < n; i += nthr) m = fmaxf(m, fabsf(to_f32(in[i])));
  }
  m = block_max(m, red);
  if (threadIdx.x == 0) part[blockIdx.x] = m;
}
template <typename T>
__global__ __launch_bounds__(256) void fp8_quant_dyn_kernel(uint8_t* __restrict__ out, const T* __restrict__ in,
                                                            const float* __restrict__ part, int n_part,
                                                            float* __restrict__ scale_out, int64_t n, bool vec) {
  __shared__ float red[32];
  constexpr int N = RowVec<T>::N;
  float m = 0.0f;
  for (int i = threadIdx.x; i < n_part; i += blockDim.x) m = fmaxf(m, part[i]);
  m = block_max(m, red);
  float sc = r16<T>(m / 448.0f);                       // fp8_scale_finalize_kernel's expression
  sc = r16<T>(fmaxf(sc, r16<T>(1e-12f)));
  if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = sc;
  const float sinv = 1.0f / sc;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
  if (vec) {
    const int64_t nvec = n / N;
    for (int64_t c = tid; c < nvec; c += nthr) {
      RowVec<T> v;
      v.raw = reinterpret_cast<const uint4*>(in)[c];
      uint32_t pk[N / 4];
#pragma unroll
      for (int j = 0; j < N; j += 4) {
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) w |= (uint32_t)f32_to_e4m3_sat(v.get(j + e) * sinv) << (8 * e);
        pk[j / 4] = w;
      }
      if constexpr (N == 8) reinterpret_cast<uint2*>(out)[c] = make_uint2(pk[0], pk[1]);
      else reinterpret_cast<uint32_t*>(out)[c] = pk[0];
    }
    for (int64_t i = nvec * N + tid; i < n; i += nthr) out[i] = f32_to_e4m3_sat(to_f32(in[i]) * sinv);
  } else {
    for (int64_t i = tid; i < n; i += nthr) out[i] = f32_to_e4m3_sat(to_f32(in[i]) * sinv);
  }
}

template <typename T>
__global__ void fp8_scale_finalize_kernel(float* scale) {
  float s = r16<T>(scale[0] / 448.0f);
  s = r16<T>(fmaxf(s, r16<T>(1e-12f)));
  scale[0] = s;
}

}  // namespace xm

using namespace xm;

// ================================================================================================
// C ABI
// ================================================================================================
// ------------------------------------------------------------------------------------------------
// N2: decode metadata refresh for graph replay (reference: llm_decode_metadata_update.cu:27-60) + dense block table
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_metadata_update_kernel(xllm_mi355_decode_metadata_t p, int64_t work) {
  const int64_t step = (int64_t)blockDim.x * gridDim.x;
  const int64_t nt = p.actual_num_tokens, B = p.actual_batch_size;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < work; idx += step) {
    if (idx < nt) {
      if (p.dst_tokens) p.dst_tokens[idx] = p.src_tokens[idx];
      if (p.dst_positions) p.dst_positions[idx] = p.src_positions[idx];
      if (p.dst_new_cache_slots) p.dst_new_cache_slots[idx] = p.src_new_cache_slots[idx];
    } else if (idx < p.padded_num_tokens) {
      if (p.dst_tokens) p.dst_tokens[idx] = 0;
      if (p.dst_new_cache_slots) p.dst_new_cache_slots[idx] = 0;
    }
    if (idx < B + 1) {
      if (p.dst_kv_seq_lens) p.dst_kv_seq_lens[idx] = p.src_kv_seq_lens[idx];
      if (p.dst_paged_kv_indptr) p.dst_paged_kv_indptr[idx] = p.src_paged_kv_indptr[idx];
    }
    if (idx < B) {
      const int32_t len = p.src_kv_seq_lens ? p.src_kv_seq_lens[idx + 1] - p.src_kv_seq_lens[idx] : 0;
      if (p.dst_kv_seq_lens_delta) p.dst_kv_seq_lens_delta[idx] = len;
      if (p.dst_kv_lens) p.dst_kv_lens[idx] = len;
      if (p.dst_paged_kv_last_page_len) p.dst_paged_kv_last_page_len[idx] = p.src_paged_kv_last_page_len[idx];
    } else if (idx < p.padded_batch_size) {
      if (p.dst_kv_lens) p.dst_kv_lens[idx] = 0;
    }
    if (idx < p.actual_indices_size && p.dst_paged_kv_indices) p.dst_paged_kv_indices[idx] = p.src_paged_kv_indices[idx];
    if (p.dst_block_table && idx < p.padded_batch_size * p.max_blocks_per_seq) {
      const int64_t b = idx / p.max_blocks_per_seq, j = idx - b * p.max_blocks_per_seq;
      int32_t v = 0;
      if (b < B) {
        const int32_t beg = p.src_paged_kv_indptr[b], end = p.src_paged_kv_indptr[b + 1];
        if (j < end - beg) v = p.src_paged_kv_indices[beg + j];
      }
      p.dst_block_table[idx] = v;
    }
  }
}

extern "C" {

int xllm_mi355_decode_metadata_update(const xllm_mi355_decode_metadata_t* params, void* stream) {
  if (!params) return XM_ERR_INVALID;
  xllm_mi355_decode_metadata_t p = *params;
  if (p.actual_num_tokens < 0 || p.actual_batch_size < 0 || p.actual_indices_size < 0) return XM_ERR_INVALID;
  if ((p.dst_tokens && !p.src_tokens) || (p.dst_positions && !p.src_positions) ||
      (p.dst_new_cache_slots && !p.src_new_cache_slots) || (p.dst_paged_kv_indices && !p.src_paged_kv_indices) ||
      (p.dst_paged_kv_indptr && !p.src_paged_kv_indptr) ||
      (p.dst_paged_kv_last_page_len && !p.src_paged_kv_last_page_len) ||
      ((p.dst_kv_seq_lens || p.dst_kv_seq_lens_delta || p.dst_kv_lens) && !p.src_kv_seq_lens))
    return XM_ERR_INVALID;
  if (p.dst_block_table && (!p.src_paged_kv_indptr || !p.src_paged_kv_indices || p.max_blocks_per_seq <= 0))
    return XM_ERR_INVALID;
  if (p.padded_batch_size < p.actual_batch_size) p.padded_batch_size = p.actual_batch_size;
  if (p.padded_num_tokens < p.actual_num_tokens) p.padded_num_tokens = p.actual_num_tokens;
  int64_t work = p.padded_num_tokens;
  if (p.actual_batch_size + 1 > work) work = p.actual_batch_size + 1;
  if (p.padded_batch_size > work) work = p.padded_batch_size;
  if (p.actual_indices_size > work) work = p.actual_indices_size;
  if (p.dst_block_table && p.padded_batch_size * p.max_blocks_per_seq > work) work = p.padded_batch_size * p.max_blocks_per_seq;
  if (work <= 0) return XM_OK;
  int64_t blocks = (work + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(decode_metadata_update_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, work);
  return hip_check_launch();
}


const char* xllm_mi355_strerror(int code) {
  switch (code) {
    case XM_OK: return "ok";
    case XM_ERR_INVALID: return "xllm_mi355: invalid argument";
    case XM_ERR_UNSUPPORTED: return "xllm_mi355: unsupported shape or dtype";
    case XM_ERR_HIP: return "xllm_mi355: HIP launch failed";
    case XM_ERR_WORKSPACE: return "xllm_mi355: workspace too small";
    default: return "xllm_mi355: unknown error";
  }
}
int xllm_mi355_abi_version(void) { return XLLM_MI355_ABI_VERSION; }

int xllm_mi355_reshape_paged_cache(const int32_t* slot_ids, const void* k, const void* v, void* k_cache,
                                   void* v_cache, int64_t n_tokens, int64_t n_kv_heads, int64_t head_dim,
                                   int64_t block_size, int64_t n_blocks, int64_t k_stride, int64_t v_stride,
                                   int elt_bytes, void* stream) {
  if (!slot_ids || !k || !k_cache || n_tokens < 0 || block_size <= 0) return XM_ERR_INVALID;
  if ((v == nullptr) != (v_cache == nullptr)) return XM_ERR_INVALID;  // K-only (MLA latent cache) needs both null
  if (elt_bytes != 2 && elt_bytes != 4 && elt_bytes != 1) return XM_ERR_UNSUPPORTED;
  if (n_tokens == 0) return XM_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t row_b = n_kv_heads * head_dim * elt_bytes;
  const int64_t ks_b = k_stride * elt_bytes, vs_b = v_stride * elt_bytes;
  const bool al16 = (row_b % 16 == 0) && (ks_b % 16 == 0) && (vs_b % 16 == 0) && ((uintptr_t)k % 16 == 0) &&
                    ((uintptr_t)v % 16 == 0) && ((uintptr_t)k_cache % 16 == 0) && ((uintptr_t)v_cache % 16 == 0);
  if (al16) {
    const int64_t rv = row_b / 16;
    const int thr = (int)(rv >= 256 ? 256 : (rv <= 64 ? 64 : ((rv + 63) / 64) * 64));
    hipLaunchKernelGGL((reshape_paged_cache_kernel<uint4>), dim3(n_tokens), dim3(thr), 0, s, slot_ids,
                       (const uint4*)k, (const uint4*)v, (uint4*)k_cache, (uint4*)v_cache, rv, ks_b / 16,
                       vs_b / 16, block_size, n_blocks);
  } else if (elt_bytes == 2) {
    hipLaunchKernelGGL((reshape_paged_cache_kernel<uint16_t>), dim3(n_tokens), dim3(256), 0, s, slot_ids,
                       (const uint16_t*)k, (const uint16_t*)v, (uint16_t*)k_cache, (uint16_t*)v_cache,
                       n_kv_heads * head_dim, k_stride, v_stride, block_size, n_blocks);
  } else if (elt_bytes == 4) {
    hipLaunchKernelGGL((reshape_paged_cache_kernel<uint32_t>), dim3(n_tokens), dim3(256), 0, s, slot_ids,
                       (const uint32_t*)k, (const uint32_t*)v, (uint32_t*)k_cache, (uint32_t*)v_cache,
                       n_kv_heads * head_dim, k_stride, v_stride, block_size, n_blocks);
  } else {
    hipLaunchKernelGGL((reshape_paged_cache_kernel<uint8_t>), dim3(n_tokens), dim3(256), 0, s, slot_ids,
                       (const uint8_t*)k, (const uint8_t*)v, (uint8_t*)k_cache, (uint8_t*)v_cache,
                       n_kv_heads * head_dim, k_stride, v_stride, block_size, n_blocks);
  }
  return hip_check_launch();
}

int xllm_mi355_block_copy(const int64_t* k_cache_ptrs, const int64_t* v_cache_ptrs, const int32_t* src_block_indices,
                          const int32_t* dst_block_indices, const int32_t* cum_sum, int64_t num_layers,
                          int64_t num_groups, int64_t num_dst_blocks, int64_t bytes_per_block, void* stream) {
  if (num_groups == 0 || num_dst_blocks == 0 || num_layers == 0) return XM_OK;     // block_copy.cu:128-130
  if (!k_cache_ptrs || !src_block_indices || !dst_block_indices || !cum_sum || num_layers < 0 || num_groups < 0 ||
      num_dst_blocks < 0 || bytes_per_block <= 0)
    return XM_ERR_INVALID;
  if (num_dst_blocks > 65535 || num_layers > 65535) return XM_ERR_UNSUPPORTED;       // grid y / z limits
  hipStream_t s = (hipStream_t)stream;
  const int ng = (int)num_groups;
  // the widest unit that divides a block; cache blocks are whole [block_size, heads, dim] slabs, so 16 in practice
  const int unit = bytes_per_block % 16 == 0 ? 16 : bytes_per_block % 4 == 0 ? 4 : bytes_per_block % 2 == 0 ? 2 : 1;
  const int64_t units = bytes_per_block / unit;
  const dim3 grid((unsigned)((units + 256 * kBlockCopyChunks - 1) / (256 * kBlockCopyChunks)), (unsigned)num_dst_blocks,
                  (unsigned)num_layers);
  if (unit == 16)
    hipLaunchKernelGGL((block_copy_kernel<uint4>), grid, dim3(256), 0, s, k_cache_ptrs, v_cache_ptrs, src_block_indices,
                       dst_block_indices, cum_sum, ng, units);
  else if (unit == 4)
    hipLaunchKernelGGL((block_copy_kernel<uint32_t>), grid, dim3(256), 0, s, k_cache_ptrs, v_cache_ptrs,
                       src_block_indices, dst_block_indices, cum_sum, ng, units);
  else if (unit == 2)
    hipLaunchKernelGGL((block_copy_kernel<uint16_t>), grid, dim3(256), 0, s, k_cache_ptrs, v_cache_ptrs,
                       src_block_indices, dst_block_indices, cum_sum, ng, units);
  else
    hipLaunchKernelGGL((block_copy_kernel<uint8_t>), grid, dim3(256), 0, s, k_cache_ptrs, v_cache_ptrs, src_block_indices,
                       dst_block_indices, cum_sum, ng, units);
  return hip_check_launch();
}

int xllm_mi355_build_block_table_from_paged_kv(const int32_t* indptr, const int32_t* indices, int32_t batch,
                                               int32_t total_pages, int32_t* block_table, void* stream) {
  if (!indptr || !block_table || batch < 0 || total_pages < 0) return XM_ERR_INVALID;
  if (batch == 0 || total_pages == 0) return XM_OK;
  hipLaunchKernelGGL(build_block_table_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, indptr, indices,
                     total_pages, block_table);
  return hip_check_launch();
}

int xllm_mi355_rms_norm(void* out, const void* input, const void* weight, float eps, int64_t n_tokens,
                        int64_t hidden, int64_t in_stride, int dtype, void* stream) {
  if (!out || !input || !weight || n_tokens < 0 || hidden <= 0) return XM_ERR_INVALID;
  XM_DISPATCH_FLOAT(dtype, T, return (launch_rms_norm<T, false, 0>(out, (void*)input, nullptr, weight, nullptr,
                                                                  nullptr, eps, n_tokens, hidden, in_stride,
                                                                  false, (hipStream_t)stream)));
  return XM_OK;
}

int xllm_mi355_fused_add_rms_norm(void* input, void* residual, const void* weight, float eps, int64_t n_tokens,
                                  int64_t hidden, int64_t in_stride, int dtype, void* stream) {
  if (!input || !residual || !weight || n_tokens < 0 || hidden <= 0) return XM_ERR_INVALID;
  if (in_stride != hidden) {
    // out == input with a token stride: only the generic kernel handles out-stride != hidden
    return XM_ERR_UNSUPPORTED;
  }
  XM_DISPATCH_FLOAT(dtype, T, return (launch_rms_norm<T, true, 0>(input, input, residual, weight, nullptr, nullptr,
                                                                 eps, n_tokens, hidden, in_stride, false,
                                                                 (hipStream_t)stream)));
  return XM_OK;
}

int xllm_mi355_rms_norm_static_fp8_quant(uint8_t* out, const void* input, void* residual, const void* weight,
                                         const float* scale, float eps, int64_t n_tokens, int64_t hidden,
                                         int64_t in_stride, int dtype, void* stream) {
  if (!out || !input || !weight || !scale || n_tokens < 0 || hidden <= 0) return XM_ERR_INVALID;
  if (residual) {
    XM_DISPATCH_HALF(dtype, T, return (launch_rms_norm<T, true, 1>(out, (void*)input, residual, weight, scale,
                                                                  nullptr, eps, n_tokens, hidden, in_stride, false,
                                                                  (hipStream_t)stream)));
  } else {
    XM_DISPATCH_HALF(dtype, T, return (launch_rms_norm<T, false, 1>(out, (void*)input, nullptr, weight, scale,
                                                                   nullptr, eps, n_tokens, hidden, in_stride,
                                                                   false, (hipStream_t)stream)));
  }
  return XM_OK;
}

int xllm_mi355_rms_norm_dynamic_int8_quant(int8_t* out_q, float* out_scale, const void* input, void* residual,
                                           const void* weight, float eps, int64_t n_tokens, int64_t hidden,
                                           int64_t in_stride, int dtype, void* stream) {
  if (!out_q || !out_scale || !input || !weight || n_tokens < 0 || hidden <= 0) return XM_ERR_INVALID;
  if (residual) {
    XM_DISPATCH_HALF(dtype, T, return (launch_rms_norm<T, true, 2>(out_q, (void*)input, residual, weight, nullptr,
                                                                  out_scale, eps, n_tokens, hidden, in_stride,
                                                                  false, (hipStream_t)stream)));
  } else {
    XM_DISPATCH_HALF(dtype, T, return (launch_rms_norm<T, false, 2>(out_q, (void*)input, nullptr, weight, nullptr,
                                                                   out_scale, eps, n_tokens, hidden, in_stride,
                                                                   false, (hipStream_t)stream)));
  }
  return XM_OK;
}

int xllm_mi355_rotary_embedding(const int64_t* positions, void* q, void* k, const void* cos_sin_cache,
                                int64_t n_tokens, int64_t n_q_heads, int64_t n_k_heads, int64_t head_size,
                                int64_t rot_dim, int64_t q_stride, int64_t k_stride, int64_t head_stride,
                                int is_neox, int dtype, void* stream) {
  if (!positions || !q || !cos_sin_cache || n_tokens < 0 || rot_dim <= 0 || (rot_dim & 1) || rot_dim > head_size)
    return XM_ERR_INVALID;
  if (n_tokens == 0) return XM_OK;
  const int nk = k ? (int)n_k_heads : 0;
  const int work = (int)((n_q_heads + nk) * (rot_dim / 2));
  const int thr = work >= 256 ? 256 : ((work + 63) / 64) * 64;
  hipStream_t s = (hipStream_t)stream;
  XM_DISPATCH_FLOAT(dtype, T, {
    if (is_neox)
      hipLaunchKernelGGL((rope_kernel<T, true>), dim3(n_tokens), dim3(thr), 0, s, positions, (T*)q, (T*)k,
                         (const T*)cos_sin_cache, (int)rot_dim, q_stride, k_stride, head_stride, (int)n_q_heads, nk);
    else
      hipLaunchKernelGGL((rope_kernel<T, false>), dim3(n_tokens), dim3(thr), 0, s, positions, (T*)q, (T*)k,
                         (const T*)cos_sin_cache, (int)rot_dim, q_stride, k_stride, head_stride, (int)n_q_heads, nk);
  });
  return hip_check_launch();
}

int xllm_mi355_rotary_embedding_and_cache(const int64_t* positions, void* q, void* k, const void* v,
                                          const void* cos_sin_cache, const int32_t* slot_ids, void* k_cache,
                                          void* v_cache, int64_t n_tokens, int64_t n_q_heads, int64_t n_kv_heads,
                                          int64_t head_size, int64_t rot_dim, int64_t q_stride, int64_t k_stride,
                                          int64_t v_stride, int64_t block_size, int64_t n_blocks, int is_neox,
                                          int dtype, void* stream) {
  if (!positions || !q || !k || !v || !cos_sin_cache || !slot_ids || !k_cache || !v_cache || n_tokens < 0 ||
      rot_dim <= 0 || (rot_dim & 1) || rot_dim > head_size || block_size <= 0)
    return XM_ERR_INVALID;
  if (n_tokens == 0) return XM_OK;
  hipStream_t s = (hipStream_t)stream;
  if (is_neox && rot_dim == head_size && head_size % 16 == 0 && (dtype == XM_BF16 || dtype == XM_F16) && q_stride % 8 == 0 &&
      k_stride % 8 == 0 && v_stride % 8 == 0 &&
      ((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)cos_sin_cache | (uintptr_t)k_cache | (uintptr_t)v_cache) % 16 == 0) {
    // prefill-sized rows (beyond the 32 MB of L2, every byte touched once): streamed
    const bool nt = n_tokens * (n_q_heads + 2 * n_kv_heads) * head_size * 2 > (48ll << 20);
    XM_DISPATCH_HALF(dtype, T, {
      hipLaunchKernelGGL((rope_and_cache_vec_kernel<T>), dim3(n_tokens), dim3(256), 0, s, positions, (T*)q, (T*)k, (const T*)v,
                         (const T*)cos_sin_cache, slot_ids, (T*)k_cache, (T*)v_cache, q_stride, k_stride, v_stride,
                         (int)head_size, (int)n_q_heads, (int)n_kv_heads, block_size, n_blocks, nt);
    });
    return hip_check_launch();
  }
  XM_DISPATCH_FLOAT(dtype, T, {
    if (is_neox)
      hipLaunchKernelGGL((rope_and_cache_kernel<T, true>), dim3(n_tokens), dim3(256), 0, s, positions, (T*)q, (T*)k,
                         (const T*)v, (const T*)cos_sin_cache, slot_ids, (T*)k_cache, (T*)v_cache, (int)rot_dim,
                         q_stride, k_stride, v_stride, (int)head_size, (int)n_q_heads, (int)n_kv_heads, block_size,
                         n_blocks);
    else
      hipLaunchKernelGGL((rope_and_cache_kernel<T, false>), dim3(n_tokens), dim3(256), 0, s, positions, (T*)q, (T*)k,
                         (const T*)v, (const T*)cos_sin_cache, slot_ids, (T*)k_cache, (T*)v_cache, (int)rot_dim,
                         q_stride, k_stride, v_stride, (int)head_size, (int)n_q_heads, (int)n_kv_heads, block_size,
                         n_blocks);
  });
  return hip_check_launch();
}

int xllm_mi355_fused_qk_norm_rope(void* qkv, int64_t n_tokens, int64_t n_q, int64_t n_k, int64_t n_v,
                                  int64_t head_dim, float eps, const void* q_weight, const void* k_weight,
                                  const void* cos_sin_cache, int cache_dtype, int interleaved,
                                  const int64_t* positions, int dtype, void* stream) {
  if (!qkv || !q_weight || !k_weight || !cos_sin_cache || !positions || n_tokens < 0) return XM_ERR_INVALID;
  if (head_dim > 256 || (head_dim & 1)) return XM_ERR_UNSUPPORTED;
  if (n_tokens == 0) return XM_OK;
  const int64_t waves = n_tokens * (n_q + n_k);
  const int64_t blocks = (waves + 3) / 4;
  hipStream_t s = (hipStream_t)stream;
  XM_DISPATCH_HALF(dtype, T, {
    if (cache_dtype == XM_F32)
      hipLaunchKernelGGL((fused_qk_norm_rope_kernel<T, float>), dim3(blocks), dim3(256), 0, s, (T*)qkv, n_tokens,
                         (int)n_q, (int)n_k, (int)n_v, (int)head_dim, eps, (const T*)q_weight, (const T*)k_weight,
                         (const float*)cos_sin_cache, interleaved, positions);
    else if (cache_dtype == dtype)
      hipLaunchKernelGGL((fused_qk_norm_rope_kernel<T, T>), dim3(blocks), dim3(256), 0, s, (T*)qkv, n_tokens,
                         (int)n_q, (int)n_k, (int)n_v, (int)head_dim, eps, (const T*)q_weight, (const T*)k_weight,
                         (const T*)cos_sin_cache, interleaved, positions);
    else
      return XM_ERR_UNSUPPORTED;
  });
  return hip_check_launch();
}

}  // extern "C" (templates need C++ linkage)

template <typename T>
static int launch_act(void* out, const void* input, int64_t n_tokens, int64_t d, int act_mode, hipStream_t s) {
  constexpr int N = Vec16B<T>::N;
  const bool vec = (d % N == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)input % 16 == 0);
  // workgroups per row: enough for ~4 workgroups per CU when the rows alone are few, each at least one 5-chunk batch per thread
  int64_t per_row = (1024 + n_tokens - 1) / n_tokens;
  const int64_t batches = (d / N + 5 * 256 - 1) / (5 * 256);
  per_row = per_row > batches ? batches : per_row;
  per_row = per_row < 1 ? 1 : per_row;
  const dim3 grid_am((unsigned)n_tokens, (unsigned)per_row);
  switch (act_mode) {
    case XM_ACT_SILU:
      hipLaunchKernelGGL((act_and_mul_kernel<T, XM_ACT_SILU>), grid_am, dim3(256), 0, s, (T*)out,
                         (const T*)input, (int)d, vec);
      break;
    case XM_ACT_GELU:
      hipLaunchKernelGGL((act_and_mul_kernel<T, XM_ACT_GELU>), grid_am, dim3(256), 0, s, (T*)out,
                         (const T*)input, (int)d, vec);
      break;
    case XM_ACT_GELU_TANH:
      hipLaunchKernelGGL((act_and_mul_kernel<T, XM_ACT_GELU_TANH>), grid_am, dim3(256), 0, s, (T*)out,
                         (const T*)input, (int)d, vec);
      break;
    default: return XM_ERR_UNSUPPORTED;
  }
  return hip_check_launch();
}

template <typename T>
static int launch_actq(int8_t* out_q, float* out_scale, const void* input, int64_t n_tokens, int64_t d,
                       int act_mode, hipStream_t s, const int32_t* live_sizes = nullptr, int n_sizes = 0) {
  if (sizeof(T) == 2 && act_mode == XM_ACT_SILU && d % 8 == 0 && d <= 1024 && n_tokens >= 512) {  // MoE expert widths
    const unsigned blocks = (unsigned)((n_tokens + 7) / 8);
    if (d <= 512)
      hipLaunchKernelGGL((act_and_mul_i8_wave_kernel<T, XM_ACT_SILU, 1>), dim3(blocks), dim3(512), 0, s, out_q, out_scale,
                         (const T*)input, (int)d, n_tokens, live_sizes, n_sizes);
    else
      hipLaunchKernelGGL((act_and_mul_i8_wave_kernel<T, XM_ACT_SILU, 2>), dim3(blocks), dim3(512), 0, s, out_q, out_scale,
                         (const T*)input, (int)d, n_tokens, live_sizes, n_sizes);
    return hip_check_launch();
  }
  if (sizeof(T) == 2 && act_mode == XM_ACT_SILU && d % 8 == 0 && d <= 20480) {  // the hot-path configuration
    const int nvec = (int)(d / 8);
#define XM_ACTQ_REG(VPT)                                                                                         \
  hipLaunchKernelGGL((act_and_mul_i8_reg_kernel<T, XM_ACT_SILU, VPT>), dim3(n_tokens), dim3(512), 0, s, out_q,   \
                     out_scale, (const T*)input, (int)d, live_sizes, n_sizes)
    if (nvec <= 512 * 2) XM_ACTQ_REG(2);
    else if (nvec <= 512 * 3) XM_ACTQ_REG(3);
    else XM_ACTQ_REG(5);
#undef XM_ACTQ_REG
    return hip_check_launch();
  }
  const size_t lds = (size_t)d * sizeof(T);
  switch (act_mode) {
    case XM_ACT_SILU:
      hipLaunchKernelGGL((act_and_mul_i8_kernel<T, XM_ACT_SILU>), dim3(n_tokens), dim3(512), lds, s, out_q,
                         out_scale, (const T*)input, (int)d, live_sizes, n_sizes);
      break;
    case XM_ACT_GELU:
      hipLaunchKernelGGL((act_and_mul_i8_kernel<T, XM_ACT_GELU>), dim3(n_tokens), dim3(512), lds, s, out_q,
                         out_scale, (const T*)input, (int)d, live_sizes, n_sizes);
      break;
    case XM_ACT_GELU_TANH:
      hipLaunchKernelGGL((act_and_mul_i8_kernel<T, XM_ACT_GELU_TANH>), dim3(n_tokens), dim3(512), lds, s, out_q,
                         out_scale, (const T*)input, (int)d, live_sizes, n_sizes);
      break;
    default: return XM_ERR_UNSUPPORTED;
  }
  return hip_check_launch();
}

// second half of the gate_up -> SiLU.mul -> per-token int8 fusion (round 3): the GEMM's epilogue wrote act = silu(gate) * up in
// 16 bit and folded each row's |max| into row_amax (atomic max on the bits of a non-negative float, zero at rest); this kernel
// quantises the row with that maximum in ONE pass (scaled_quantize's expression: q = rint(v * 127 / amax), scale = amax / 127)
// and puts the row's entry of row_amax back to zero. Bit-identical to act_and_mul -> scaled_quantize.
template <typename T>
__global__ __launch_bounds__(512) void quantize_with_row_amax_kernel(const T* __restrict__ act, float* __restrict__ row_amax,
                                                                    int8_t* __restrict__ out_q, float* __restrict__ out_s, int d,
                                                                    int nt) {
  // nt (round 4): prefill-sized tensors (beyond the L2, each byte touched once) are streamed with non-temporal loads / stores
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const int64_t t = blockIdx.x;
  const float amax = row_amax[t];
  const float qinv = (amax > 1e-10f) ? 127.0f / amax : 0.0f;
  const int nvec = d / 8;
  const u32x4* x = reinterpret_cast<const u32x4*>(act + t * (int64_t)d);
  // six 16-byte loads of a thread in flight before the first is consumed: a row of up to 24576 elements (Qwen2-7B: 18944) is ONE
  // round trip to memory per workgroup. (Round 4: the prefill launch -- 310 MB in, 155 MB out, nothing of it cache-resident --
  // takes 80-83 us = 5.6-5.8 TB/s with one, four or six loads in flight: that is what a 2:1 read/write stream gets from this
  // HBM, not a latency problem; the 7 TB/s of rms_norm in the chunk owes part of its bytes to the Infinity Cache.)
  constexpr int U = 6;
  for (int c0 = threadIdx.x; c0 < nvec; c0 += U * 512) {
    u32x4 vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * 512;
      if (c < nvec) vv[u] = nt ? __builtin_nontemporal_load(&x[c]) : x[c];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int c = c0 + u * 512;
      if (c >= nvec) break;
      const u32x4 v = vv[u];
      uint32_t pk[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        uint32_t wq = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const uint32_t word = v[h * 2 + (e >> 1)];
          const float r = half_bits_to_f32<T>((e & 1) ? (word >> 16) : (word & 0xffffu));
          const float qv = fmaxf(-127.0f, fminf(127.0f, rintf(r * qinv)));
          wq |= ((uint32_t)(int)qv & 0xffu) << (8 * e);
        }
        pk[h] = wq;
      }
      u32x2* const dst = reinterpret_cast<u32x2*>(out_q + t * (int64_t)d + (int64_t)c * 8);
      const u32x2 qv2 = {pk[0], pk[1]};
      if (nt) __builtin_nontemporal_store(qv2, dst);
      else *dst = qv2;
    }
  }
  __syncthreads();                       // every thread has read row_amax[t]
  if (threadIdx.x == 0) { out_s[t] = amax / 127.0f; row_amax[t] = 0.0f; }
}

// out = rT(a + b), 16-bit tensors of n elements (the second pass of scaled_matmul with an addend where no GEMM epilogue takes it)
template <typename T>
__global__ __launch_bounds__(256) void add16_kernel(T* __restrict__ out, const T* a, const T* b, int64_t n) {
  const int64_t nvec = n / 8;
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < nvec; c += (int64_t)gridDim.x * 256) {
    RowVec<T> x, y, r;
    x.raw = reinterpret_cast<const uint4*>(a)[c];
    y.raw = reinterpret_cast<const uint4*>(b)[c];
#pragma unroll
    for (int j = 0; j < 8; ++j) r.set(j, x.get(j) + y.get(j));
    reinterpret_cast<uint4*>(out)[c] = r.raw;
  }
  if (blockIdx.x == 0)
    for (int64_t i = nvec * 8 + threadIdx.x; i < n; i += 256) out[i] = from_f32<T>(to_f32(a[i]) + to_f32(b[i]));
}

extern "C" {

int xllm_mi355_add16(void* out, const void* a, const void* b, int64_t n, int dtype, void* stream) {
  if (!out || !a || !b || n < 0) return XM_ERR_INVALID;
  if (dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (((uintptr_t)out | (uintptr_t)a | (uintptr_t)b) % 16) return XM_ERR_UNSUPPORTED;
  if (n == 0) return XM_OK;
  int64_t blocks = (n / 8 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  XM_DISPATCH_HALF(dtype, T, hipLaunchKernelGGL((add16_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                                                (T*)out, (const T*)a, (const T*)b, n));
  return hip_check_launch();
}

int xllm_mi355_quantize_with_row_amax(const void* act, float* row_amax, int8_t* out_q, float* out_scale, int64_t n_tokens,
                                      int64_t d, int dtype, void* stream) {
  if (!act || !row_amax || !out_q || !out_scale || n_tokens < 0 || d <= 0) return XM_ERR_INVALID;
  if (d % 8 || ((uintptr_t)act % 16) || ((uintptr_t)out_q % 8)) return XM_ERR_UNSUPPORTED;
  if (n_tokens == 0) return XM_OK;
  XM_DISPATCH_HALF(dtype, T,
                   hipLaunchKernelGGL((quantize_with_row_amax_kernel<T>), dim3((unsigned)n_tokens), dim3(512), 0,
                                      (hipStream_t)stream, (const T*)act, row_amax, out_q, out_scale, (int)d,
                                      (n_tokens * d * 2 > (48ll << 20)) ? 1 : 0));
  return hip_check_launch();
}

int xllm_mi355_act_and_mul(void* out, const void* input, int64_t n_tokens, int64_t d, int act_mode, int dtype,
                           void* stream) {
  if (!out || !input || n_tokens < 0 || d <= 0) return XM_ERR_INVALID;
  if (n_tokens == 0) return XM_OK;
  XM_DISPATCH_FLOAT(dtype, T, return launch_act<T>(out, input, n_tokens, d, act_mode, (hipStream_t)stream));
  return XM_OK;
}

int xllm_mi355_act_and_mul_dynamic_int8_quant(int8_t* out_q, float* out_scale, const void* input,
                                              int64_t n_tokens, int64_t d, int act_mode, int dtype, void* stream) {
  if (!out_q || !out_scale || !input || n_tokens < 0 || d <= 0) return XM_ERR_INVALID;
  if (n_tokens == 0) return XM_OK;
  if (d % 8 != 0 || d * 2 > 65536 || (uintptr_t)input % 16 || (uintptr_t)out_q % 8) return XM_ERR_UNSUPPORTED;
  XM_DISPATCH_HALF(dtype, T,
                   return launch_actq<T>(out_q, out_scale, input, n_tokens, d, act_mode, (hipStream_t)stream));
  return XM_OK;
}

int xllm_mi355_act_and_mul_dynamic_int8_quant_live(int8_t* out_q, float* out_scale, const void* input,
                                                   int64_t n_tokens, int64_t d, int act_mode, int dtype,
                                                   const int32_t* live_sizes, int64_t n_sizes, void* stream) {
  if (!out_q || !out_scale || !input || n_tokens < 0 || d <= 0 || (live_sizes && n_sizes <= 0)) return XM_ERR_INVALID;
  if (n_tokens == 0) return XM_OK;
  if (d % 8 != 0 || d * 2 > 65536 || (uintptr_t)input % 16 || (uintptr_t)out_q % 8) return XM_ERR_UNSUPPORTED;
  XM_DISPATCH_HALF(dtype, T,
                   return launch_actq<T>(out_q, out_scale, input, n_tokens, d, act_mode, (hipStream_t)stream, live_sizes,
                                         (int)n_sizes));
  return XM_OK;
}

int xllm_mi355_scaled_quantize(const void* x, int8_t* out, float* out_scale, int64_t M, int64_t K, int dtype,
                               void* stream) {
  if (!x || !out || !out_scale || M < 0 || K <= 0) return XM_ERR_INVALID;
  if (M == 0) return XM_OK;
  hipStream_t s = (hipStream_t)stream;
  XM_DISPATCH_FLOAT(dtype, T, {
    constexpr int N = Vec16B<T>::N;
    const bool vec = (K % N == 0) && ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 8 == 0);
    const bool nt = M * K * (int64_t)sizeof(T) > (48ll << 20);
    if (vec && M >= 1024 && K / N <= 4 * 128)
      hipLaunchKernelGGL((scaled_quantize_i8_kernel<T, 128>), dim3(M), dim3(128), 0, s, (const T*)x, out, out_scale,
                         (int)K, vec, nt);
    else
      hipLaunchKernelGGL((scaled_quantize_i8_kernel<T, 512>), dim3(M), dim3(512), 0, s, (const T*)x, out, out_scale,
                         (int)K, vec, nt);
  });
  return hip_check_launch();
}

int xllm_mi355_static_scaled_fp8_quant(uint8_t* out, const void* input, const float* scale, int64_t numel,
                                       int dtype, void* stream) {
  if (!out || !input || !scale || numel < 0) return XM_ERR_INVALID;
  if (numel == 0) return XM_OK;
  hipStream_t s = (hipStream_t)stream;
  XM_DISPATCH_FLOAT(dtype, T, {
    constexpr int N = Vec16B<T>::N;
    const bool vec = ((uintptr_t)input % 16 == 0) && ((uintptr_t)out % 8 == 0);
    int64_t blocks = (numel / N + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((fp8_quant_kernel<T>), dim3(blocks), dim3(256), 0, s, out, (const T*)input, scale, numel, vec);
  });
  return hip_check_launch();
}

int xllm_mi355_fp8_scaled_quantize(uint8_t* out, const void* input, const float* scale_in, float* scale_out,
                                   int64_t numel, int dtype, void* stream) {
  if (!out || !input || numel < 0 || (!scale_in && !scale_out)) return XM_ERR_INVALID;
  if (scale_in) return xllm_mi355_static_scaled_fp8_quant(out, input, scale_in, numel, dtype, stream);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(scale_out, 0, sizeof(float), s) != hipSuccess) return XM_ERR_HIP;
  if (numel > 0) {
    XM_DISPATCH_FLOAT(dtype, T, {
      int64_t blocks = (numel + 256 * 8 - 1) / (256 * 8);
      if (blocks > 1024) blocks = 1024;
      hipLaunchKernelGGL((amax_kernel<T>), dim3(blocks), dim3(256), 0, s, (const T*)input, numel,
                         (uint32_t*)scale_out);
      hipLaunchKernelGGL((fp8_scale_finalize_kernel<T>), dim3(1), dim3(1), 0, s, scale_out);
    });
  }
  int rc = hip_check_launch();
  if (rc) return rc;
  return xllm_mi355_static_scaled_fp8_quant(out, input, scale_out, numel, dtype, stream);
}

size_t xllm_mi355_fp8_scaled_quantize_workspace_bytes(void) { return 1024 * sizeof(float); }

int xllm_mi355_fp8_scaled_quantize_ws(uint8_t* out, const void* input, float* scale_out, int64_t numel, int dtype,
                                      void* workspace, size_t ws_bytes, void* stream) {
  if (!out || !input || !scale_out || numel < 0) return XM_ERR_INVALID;
  if (!workspace || ws_bytes < 1024 * sizeof(float) || ((uintptr_t)workspace % 4)) return XM_ERR_WORKSPACE;
  if (numel == 0) return xllm_mi355_fp8_scaled_quantize(out, input, nullptr, scale_out, numel, dtype, stream);
  hipStream_t s = (hipStream_t)stream;
  XM_DISPATCH_FLOAT(dtype, T, {
    constexpr int N = Vec16B<T>::N;
    const bool vec = ((uintptr_t)input % 16 == 0) && ((uintptr_t)out % 8 == 0);
    int64_t blocks = (numel / N + 255) / 256;
    blocks = blocks > 1024 ? 1024 : (blocks < 1 ? 1 : blocks);
    hipLaunchKernelGGL((amax_partial_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, (const T*)input, numel,
                       reinterpret_cast<float*>(workspace), vec);
    hipLaunchKernelGGL((fp8_quant_dyn_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, out, (const T*)input,
                       reinterpret_cast<const float*>(workspace), (int)blocks, scale_out, numel, vec);
  });
  return hip_check_launch();
}

}  // extern "C"
