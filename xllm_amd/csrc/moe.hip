// moe.hip -- MoE index build and combine for gfx950.
//
// Reference: cuda::moe_compute_index (xllm/core/kernels/cuda/moe/moe_compute_index.cu:41-160: histogram,
// 1-block prefix sum, atomic placement => NON-deterministic intra-expert order) and cuda::moe_combine_result
// (moe/moe_combine.cu:38-62: out[t] = sum_k w[t,k] * gemm2[t*topk+k], fp32 accumulate).
// Here the placement is a STABLE counting sort (order inside an expert = expanded row index), so results are
// reproducible run to run and graph-replay safe; permutation-invariant results equal the reference's.
#include "common.h"

namespace xm {

constexpr int kMoeChunk = 1024;  // expanded rows per chunk
constexpr int kMaxExperts = 1024;

// pass 1: per-chunk histogram  cnt[chunk][e]
__global__ __launch_bounds__(256) void moe_hist_kernel(const int32_t* __restrict__ expert_id, int64_t n, int E,
                                                       int32_t* __restrict__ chunk_cnt) {
  extern __shared__ int32_t h[];
  for (int e = threadIdx.x; e < E; e += blockDim.x) h[e] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kMoeChunk;
  for (int i = threadIdx.x; i < kMoeChunk; i += blockDim.x)
    if (base + i < n) atomicAdd(&h[expert_id[base + i]], 1);
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += blockDim.x) chunk_cnt[(int64_t)blockIdx.x * E + e] = h[e];
}

// pass 2 (one workgroup): expert sizes, expert offsets (exclusive scan over experts), and per-chunk bases
// chunk_cnt[c][e] <- offset[e] + sum_{c' < c} cnt[c'][e]
__global__ __launch_bounds__(1024) void moe_scan_kernel(int32_t* __restrict__ chunk_cnt, int nchunks, int E,
                                                        int32_t* __restrict__ expert_sizes) {
  __shared__ int32_t sizes[kMaxExperts];
  __shared__ int32_t offs[kMaxExperts];
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    int32_t s = 0;
    for (int c = 0; c < nchunks; ++c) s += chunk_cnt[(int64_t)c * E + e];
    sizes[e] = s;
    expert_sizes[e] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t run = 0;
    for (int e = 0; e < E; ++e) { offs[e] = run; run += sizes[e]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    int32_t run = offs[e];
    for (int c = 0; c < nchunks; ++c) {
      const int32_t v = chunk_cnt[(int64_t)c * E + e];
      chunk_cnt[(int64_t)c * E + e] = run;
      run += v;
    }
  }
}

// pass 3: one wave per chunk walks its rows in order, 64 at a time; rank inside the wave by ballot
__global__ __launch_bounds__(64) void moe_place_kernel(const int32_t* __restrict__ expert_id, int64_t n, int E,
                                                       const int32_t* __restrict__ chunk_base,
                                                       int32_t* __restrict__ src_dst, int32_t* __restrict__ dst_src) {
  extern __shared__ int32_t run[];  // running position per expert inside this chunk
  const int lane = threadIdx.x;
  for (int e = lane; e < E; e += 64) run[e] = chunk_base[(int64_t)blockIdx.x * E + e];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kMoeChunk;
  for (int it = 0; it < kMoeChunk / 64; ++it) {
    const int64_t i = base + it * 64 + lane;
    const bool valid = i < n;
    const int e = valid ? expert_id[i] : -1;
    unsigned long long todo = __ballot(valid);
    int pos = -1;
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int e0 = __shfl(e, leader);
      const unsigned long long m = __ballot(valid && e == e0);
      if (valid && e == e0) {
        const unsigned long long lt = (lane == 0) ? 0ull : (m & ((1ull << lane) - 1ull));
        pos = run[e0] + __popcll(lt);
      }
      __syncthreads();  // single wave: orders the LDS read above before the update below
      if (lane == leader) run[e0] += __popcll(m);
      __syncthreads();
      todo &= ~m;
    }
    if (valid) {
      src_dst[i] = pos;
      dst_src[pos] = (int32_t)i;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void moe_combine_kernel(T* __restrict__ out, const T* __restrict__ gemm2,
                                                          const float* __restrict__ w, int topk, int H) {
  const int64_t t = blockIdx.x;
  for (int i = threadIdx.x; i < H; i += blockDim.x) {
    float acc = 0.0f;
    for (int k = 0; k < topk; ++k) acc += w[t * topk + k] * to_f32(gemm2[(t * topk + k) * (int64_t)H + i]);
    out[t * (int64_t)H + i] = from_f32<T>(acc);
  }
}

}  // namespace xm

using namespace xm;

// scratch for the chunk histograms (nchunks * E int32); sized for 1M expanded rows x 1024 experts at most
static int32_t* g_moe_scratch = nullptr;
static size_t g_moe_scratch_elems = 0;

extern "C" {

XM_API int xllm_mi355_set_moe_workspace(void* ws, size_t bytes) {
  g_moe_scratch = reinterpret_cast<int32_t*>(ws);
  g_moe_scratch_elems = bytes / 4;
  return XM_OK;
}

int xllm_mi355_moe_compute_index(const int32_t* expert_id, int64_t n_tokens, int64_t topk, int64_t n_experts,
                                 int32_t* src_dst, int32_t* dst_src, int32_t* expert_sizes, void* stream) {
  if (!expert_id || !src_dst || !dst_src || !expert_sizes || n_tokens < 0 || topk <= 0 || n_experts <= 0)
    return XM_ERR_INVALID;
  if (n_experts > kMaxExperts) return XM_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = n_tokens * topk;
  const int E = (int)n_experts;
  if (n == 0) return hipMemsetAsync(expert_sizes, 0, E * sizeof(int32_t), s) == hipSuccess ? XM_OK : XM_ERR_HIP;
  const int nchunks = (int)((n + kMoeChunk - 1) / kMoeChunk);
  if (!g_moe_scratch || g_moe_scratch_elems < (size_t)nchunks * E) return XM_ERR_WORKSPACE;
  hipLaunchKernelGGL(moe_hist_kernel, dim3(nchunks), dim3(256), E * sizeof(int32_t), s, expert_id, n, E, g_moe_scratch);
  hipLaunchKernelGGL(moe_scan_kernel, dim3(1), dim3(1024), 0, s, g_moe_scratch, nchunks, E, expert_sizes);
  hipLaunchKernelGGL(moe_place_kernel, dim3(nchunks), dim3(64), E * sizeof(int32_t), s, expert_id, n, E,
                     g_moe_scratch, src_dst, dst_src);
  return hip_check_launch();
}

int xllm_mi355_moe_combine(void* out, const void* gemm2, const float* weights, int64_t n_tokens, int64_t topk,
                           int64_t hidden, int dtype, void* stream) {
  if (!out || !gemm2 || !weights || n_tokens < 0 || topk <= 0 || hidden <= 0) return XM_ERR_INVALID;
  if (n_tokens == 0) return XM_OK;
  XM_DISPATCH_FLOAT(dtype, T,
                    hipLaunchKernelGGL((moe_combine_kernel<T>), dim3(n_tokens), dim3(256), 0, (hipStream_t)stream,
                                       (T*)out, (const T*)gemm2, weights, (int)topk, (int)hidden));
  return hip_check_launch();
}

}  // extern "C"
