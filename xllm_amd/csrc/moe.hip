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
    if (base + i < n) {
      // ids outside [0, E) (padding rows, -1) are not counted, as in the reference's histogram
      // (kernels/cuda/moe/moe_compute_index.cu: `eid >= 0 && eid < num_experts`); the place pass marks them below
      const int32_t e = expert_id[base + i];
      if (e >= 0 && e < E) atomicAdd(&h[e], 1);
    }
  __syncthreads();
  for (int e = threadIdx.x; e < E; e += blockDim.x) chunk_cnt[(int64_t)blockIdx.x * E + e] = h[e];
}

// pass 2 (one workgroup): expert sizes, expert offsets (exclusive scan over experts), and per-chunk bases
// chunk_cnt[c][e] <- offset[e] + sum_{c' < c} cnt[c'][e]. The 1024 threads are (expert, part): 1024 / Epad parts share the
// chunks of an expert, so the chain of dependent loads per thread is nchunks / parts long instead of nchunks (64 chunks at
// T * topk = 65536, E = 128: 8 instead of 64 -- the kernel is pure latency).
__global__ __launch_bounds__(1024) void moe_scan_kernel(int32_t* __restrict__ chunk_cnt, int nchunks, int E,
                                                        int32_t* __restrict__ expert_sizes) {
  __shared__ int32_t incl[kMaxExperts];
  __shared__ int32_t part_sum[kMaxExperts];   // [part][e], parts * Epad == 1024
  int epad = 1;
  while (epad < E) epad <<= 1;
  const int parts = kMaxExperts / epad;
  const int e = threadIdx.x & (epad - 1), part = threadIdx.x / epad;
  const int per = (nchunks + parts - 1) / parts;
  const int c0 = part * per, c1 = c0 + per < nchunks ? c0 + per : nchunks;
  int32_t s = 0;
  if (e < E)
    for (int c = c0; c < c1; ++c) s += chunk_cnt[(int64_t)c * E + e];
  part_sum[part * epad + e] = s;
  __syncthreads();
  int32_t tot = 0, before = 0;   // the expert's size and the rows of the parts in front of this one
  for (int p = 0; p < parts; ++p) {
    const int32_t v = part_sum[p * epad + e];
    tot += v;
    before += p < part ? v : 0;
  }
  const int t = threadIdx.x;     // the scan over experts runs on threads 0 .. 1023 = expert index (part 0 holds the sizes)
  if (part == 0 && e < E) expert_sizes[e] = tot;
  __syncthreads();
  incl[t] = 0;
  if (part == 0) incl[e] = e < E ? tot : 0;
  __syncthreads();
  for (int d = 1; d < kMaxExperts; d <<= 1) {  // Hillis-Steele inclusive scan over the experts
    const int32_t v = t >= d ? incl[t - d] : 0;
    __syncthreads();
    incl[t] += v;
    __syncthreads();
  }
  if (e < E) {
    int32_t run = incl[e] - tot + before;
    for (int c = c0; c < c1; ++c) {
      const int32_t v = chunk_cnt[(int64_t)c * E + e];
      chunk_cnt[(int64_t)c * E + e] = run;
      run += v;
    }
  }
}

// pass 3: one workgroup (16 waves) per chunk of 1024 rows; wave w owns rows 64 w .. 64 w + 63 of the chunk.
//   A  every wave walks the DISTINCT experts among its 64 rows (leader's id broadcast, ballot of the lanes that share it): a row's
//      rank among the wave's rows of its expert = popcount of the lower lanes in that ballot, the ballot's popcount = the
//      wave's count for the expert (LDS, [wave][expert]);
//   B  thread e turns the 16 per-wave counts of expert e into running offsets on top of the chunk's base (pass 2);
//   C  position = offset[wave][expert] + rank: the rows of an expert keep their order inside a wave, across the waves and (pass 2)
//      across the chunks -> stable, no atomics-defined order.
// (Round 6. Before: every wave held the whole chunk and walked E / 16 experts x 16 row groups -- 256 ballot rounds per wave at
//  256 experts, 19.7 us for the 1024 rows of a decode step; now <= 64 rounds, typically the number of distinct experts in 64 rows.)
// SINGLE (n <= 1024 rows: every decode step): the one workgroup also does passes 1 and 2 -- expert sizes = the sums of its per-wave
// counts, their exclusive scan in LDS -- so the index build of a decode step is ONE launch instead of three.
template <bool SINGLE>
__global__ __launch_bounds__(1024) void moe_place_kernel(const int32_t* __restrict__ expert_id, int64_t n, int E,
                                                         const int32_t* __restrict__ chunk_base,
                                                         int32_t* __restrict__ src_dst, int32_t* __restrict__ dst_src,
                                                         int32_t* __restrict__ expert_sizes) {
  extern __shared__ int32_t wcnt[];   // [16 waves][E]
  __shared__ int32_t incl[SINGLE ? kMaxExperts : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * kMoeChunk + threadIdx.x;
  for (int k = threadIdx.x; k < 16 * E; k += 1024) wcnt[k] = 0;
  const int eid = i < n ? expert_id[i] : -1;
  const bool live = eid >= 0 && eid < E;
  __syncthreads();
  int rank = 0;
  {
    const unsigned long long lt = (1ull << lane) - 1ull;
    unsigned long long todo = __ballot(live);
    while (todo) {                                   // wave-uniform
      const int leader = __ffsll((long long)todo) - 1;
      const int e = __shfl(eid, leader);
      const unsigned long long m = __ballot(live && eid == e);
      if (live && eid == e) rank = __popcll(m & lt);
      if (lane == leader) wcnt[wave * E + e] = __popcll(m);
      todo &= ~m;
    }
  }
  __syncthreads();
  int single_base = 0;
  if constexpr (SINGLE) {
    const int t = threadIdx.x;
    int tot = 0;
    if (t < E) {
#pragma unroll
      for (int w = 0; w < 16; ++w) tot += wcnt[w * E + t];
      expert_sizes[t] = tot;
    }
    incl[t] = t < E ? tot : 0;
    __syncthreads();
    for (int d = 1; d < kMaxExperts; d <<= 1) {      // Hillis-Steele inclusive scan over the experts (as moe_scan_kernel)
      const int32_t v = t >= d ? incl[t - d] : 0;
      __syncthreads();
      incl[t] += v;
      __syncthreads();
    }
    single_base = incl[t] - tot;
  }
  for (int e = threadIdx.x; e < E; e += 1024) {      // (E <= 1024: one expert per thread)
    int run = SINGLE ? single_base : chunk_base[(int64_t)blockIdx.x * E + e];
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int c = wcnt[w * E + e];
      wcnt[w * E + e] = run;
      run += c;
    }
  }
  __syncthreads();
  if (i < n) {
    if (live) {
      const int pos = wcnt[wave * E + eid] + rank;
      src_dst[i] = pos;
      dst_src[pos] = (int32_t)i;
    } else {
      src_dst[i] = -1;  // a row whose id is outside [0, E) has no sorted position: the sorted combine skips it
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

// N1-style fusion of the reference's index_copy_ + moe_combine_result (layers/dcu/fused_moe.cpp:296-303): the second
// grouped GEMM's rows stay in expert order and are gathered through src_dst while they are combined:
// out[t] = sum_k w[t, k] * gemm2_sorted[src_dst[t * topk + k]] (same fp32 sum, same order as the two operators).
// EP form (local_sizes != nullptr): only the first sum(local_sizes[0 .. n_local)) sorted rows exist (the rank's own experts,
// sorted to the front); the other rows are the ZERO rows of the reference's gemm2_full (fused_moe.cpp:291-297) and are
// skipped instead of being materialised.
// HOIST8 (round 6): top-8 with every row present (no expert-parallel slice) -- a separate instantiation, so that the plain loop of
// the other form keeps its register footprint (its launch is 8192 workgroups whose occupancy hides the loop's latencies)
template <typename T, bool HOIST8>
__global__ __launch_bounds__(256) void moe_combine_sorted_kernel(T* __restrict__ out, const T* __restrict__ gemm2,
                                                                 const int32_t* __restrict__ src_dst,
                                                                 const float* __restrict__ w, int topk, int H,
                                                                 const int32_t* __restrict__ local_sizes, int n_local) {
  const int64_t t = blockIdx.x;
  constexpr int kMaxTopk = 16;
  __shared__ int rows[kMaxTopk];
  __shared__ float ws[kMaxTopk];
  __shared__ int n_valid;
  if (threadIdx.x == 0) n_valid = local_sizes ? 0 : 0x7fffffff;
  __syncthreads();
  if (local_sizes) {
    int part = 0;
    for (int e = threadIdx.x; e < n_local; e += blockDim.x) part += local_sizes[e];
    if (part) atomicAdd(&n_valid, part);
  }
  if (threadIdx.x < topk) {
    rows[threadIdx.x] = src_dst[t * topk + threadIdx.x];
    ws[threadIdx.x] = w[t * topk + threadIdx.x];
  }
  __syncthreads();
  const int nv = n_valid;
  for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) {  // H % 8 == 0: one 16-byte load per row and thread
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (HOIST8) {
      // the common top-8 with every row present (no expert-parallel slice: there most rows are skipped and the plain loop is the
      // cheaper form -- 16.5 vs 21 us at cfg5): all eight rows requested before the first is consumed (the loads sat behind the skip test, one memory
      // latency per row: 19.5 us for 128 tokens at H = 7168, round 6); skipped rows stay out of the sum (same terms, same order)
      RowVec<T> v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k].raw = make_uint4(0u, 0u, 0u, 0u);
        if (rows[k] >= 0 && rows[k] < nv) v[k].raw = *reinterpret_cast<const uint4*>(gemm2 + (int64_t)rows[k] * H + i);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (rows[k] >= 0 && rows[k] < nv) {
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += ws[k] * v[k].get(j);
        }
      }
    } else {
      for (int k = 0; k < topk; ++k) {
        if (rows[k] < 0 || rows[k] >= nv) continue;
        const uint4 v = *reinterpret_cast<const uint4*>(gemm2 + (int64_t)rows[k] * H + i);
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += ws[k] * to_f32(e[j]);
      }
    }
    uint4 o;
    T* oe = reinterpret_cast<T*>(&o);
#pragma unroll
    for (int j = 0; j < 8; ++j) oe[j] = from_f32<T>(acc[j]);
    *reinterpret_cast<uint4*>(out + t * (int64_t)H + i) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// N4: fused gating top-k. Reference: cuda::moe_fused_topk (kernels/cuda/moe/moe_fused_topk.cu:31-61) ->
// topk_softmax / topk_sigmoid (moe_topk_softmax_kernels.cuh:324-625, moe_topk_sigmoid_kernels.cuh:156-392):
//   softmax: p = exp(x - max) * (1 / sum), k rounds of argmax (lower index wins ties), weight = p
//   sigmoid: s = 1 / (1 + exp(-x)); selection score v = s + bias[e] when a correction bias is given, weight = v - bias[e]
//   renormalize: weights *= 1 / (sum of the selected weights)
// One wave per token; lane l holds experts l, l + 64, ... (E <= 512), the k argmax rounds are 6-step butterflies.
// ------------------------------------------------------------------------------------------------
constexpr int kTopkMaxPerLane = 8;

// PL = experts per lane (ceil(E / 64) rounded up to 1, 2, 4, 8: round 6 -- every per-expert loop ran all 8 slots at E = 128)
template <typename T, int PL>
__global__ __launch_bounds__(256) void moe_fused_topk_kernel(const T* __restrict__ gating, int n_tokens, int E, int topk,
                                                             int renormalize, const float* __restrict__ bias,
                                                             int sigmoid, float* __restrict__ out_w,
                                                             int32_t* __restrict__ out_id) {
  const int lane = threadIdx.x & 63;
  const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= n_tokens) return;
  const T* row = gating + (int64_t)tok * E;
  float v[PL];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    const int e = lane + 64 * j;
    v[j] = e < E ? to_f32(row[e]) : -INFINITY;
    mx = fmaxf(mx, v[j]);
  }
  if (sigmoid) {
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      const int e = lane + 64 * j;
      if (e < E) {
        float sg = 1.0f / (1.0f + expf(-v[j]));
        if (bias) sg = sg + bias[e];
        v[j] = sg;
      }
    }
  } else {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      const int e = lane + 64 * j;
      v[j] = e < E ? expf(v[j] - mx) : 0.0f;
      sum += v[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      const int e = lane + 64 * j;
      v[j] = e < E ? v[j] * inv : -INFINITY;
    }
  }
  float wsum = 0.0f;
  float my_w = 0.0f;   // lane i keeps the i-th selected weight for the renormalisation pass
  for (int kk = 0; kk < topk; ++kk) {
    float best = -INFINITY;
    int best_e = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      const int e = lane + 64 * j;
      if (e < E && (v[j] > best || (v[j] == best && e < best_e))) { best = v[j]; best_e = e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int oe = __shfl_xor(best_e, o);
      if (ob > best || (ob == best && oe < best_e)) { best = ob; best_e = oe; }
    }
    if ((best_e & 63) == lane) {  // the owner retires the expert
#pragma unroll
      for (int j = 0; j < PL; ++j)
        if (lane + 64 * j == best_e) v[j] = -INFINITY;
    }
    float w = best;
    if (sigmoid && bias) w = best - bias[best_e];
    wsum += w;
    if (lane == kk) my_w = w;
    if (lane == 0) out_id[(int64_t)tok * topk + kk] = best_e;
  }
  if (lane < topk) {
    if (renormalize) my_w = my_w * (1.0f / wsum);
    out_w[(int64_t)tok * topk + lane] = my_w;
  }
}

// ------------------------------------------------------------------------------------------------
// a16 / N4: grouped gating top-k (DeepSeek-V2 / V3 device-limited routing). Reference call site: dcu::moe_grouped_topk
// (kernels/dcu/topk_gate.cpp:59-125) -> aiter::native::grouped_topk / biased_grouped_topk -- an external library that is
// not in the reference tree; the published DeepSeek-V2 / V3 gate algorithm:
//   s = softmax(x) | sigmoid(x); choice score c = s (+ bias); group value = max c (no bias) | sum of the two largest c
//   (bias); keep the topk_group best groups; topk experts by c among them (ties: lower index); weight = s (unbiased);
//   renormalize by the selected sum; times routed_scaling_factor.
// One wave per token as in moe_fused_topk_kernel; the choice scores go through 2 KB of LDS per wave once so that lane g
// can walk group g's experts (a wave's own LDS writes are visible to its later reads: no barrier).
// ------------------------------------------------------------------------------------------------
template <typename T, int PL>
__global__ __launch_bounds__(256) void moe_grouped_topk_kernel(const T* __restrict__ gating, int n_tokens, int E, int topk,
                                                               int G, int topk_group, int renormalize,
                                                               const float* __restrict__ bias, int sigmoid,
                                                               float route_scale, float* __restrict__ out_w,
                                                               int32_t* __restrict__ out_id) {
  __shared__ float choice[4][64 * PL];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tok = blockIdx.x * 4 + wv;
  if (tok >= n_tokens) return;
  const T* row = gating + (int64_t)tok * E;
  float sc[PL], v[PL];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    const int e = lane + 64 * j;
    sc[j] = e < E ? to_f32(row[e]) : -INFINITY;
    mx = fmaxf(mx, sc[j]);
  }
  if (sigmoid) {
#pragma unroll
    for (int j = 0; j < PL; ++j) sc[j] = lane + 64 * j < E ? 1.0f / (1.0f + expf(-sc[j])) : 0.0f;
  } else {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      sc[j] = lane + 64 * j < E ? expf(sc[j] - mx) : 0.0f;
      sum += sc[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int j = 0; j < PL; ++j) sc[j] = sc[j] * inv;
  }
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    const int e = lane + 64 * j;
    v[j] = e < E ? (bias ? sc[j] + bias[e] : sc[j]) : -INFINITY;
    if (e < E) choice[wv][e] = v[j];
  }
  // group values (lane g walks group g), their ranks, the set of kept groups
  const int EG = E / G;
  float gs = -INFINITY;
  if (lane < G) {
    float t1 = -INFINITY, t2 = -INFINITY;
    for (int i = 0; i < EG; ++i) {
      const float x = choice[wv][lane * EG + i];
      if (x > t1) { t2 = t1; t1 = x; } else if (x > t2) t2 = x;
    }
    gs = bias ? t1 + t2 : t1;
  }
  int rank = 0;
  for (int h = 0; h < G; ++h) {
    const float oh = __shfl(gs, h);
    rank += (oh > gs) || (oh == gs && h < lane);
  }
  const unsigned long long kept = __ballot(lane < G && rank < topk_group);
#pragma unroll
  for (int j = 0; j < PL; ++j) {
    const int e = lane + 64 * j;
    if (e < E && !((kept >> (e / EG)) & 1ull)) v[j] = -INFINITY;
  }
  float wsum = 0.0f;
  float my_w = 0.0f;
  for (int kk = 0; kk < topk; ++kk) {
    float best = -INFINITY;
    int best_e = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < PL; ++j) {
      const int e = lane + 64 * j;
      if (e < E && (v[j] > best || (v[j] == best && e < best_e))) { best = v[j]; best_e = e; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int oe = __shfl_xor(best_e, o);
      if (ob > best || (ob == best && oe < best_e)) { best = ob; best_e = oe; }
    }
    float ws = 0.0f;              // the owner retires the expert and supplies its unbiased score
#pragma unroll
    for (int j = 0; j < PL; ++j)
      if (lane + 64 * j == best_e) { ws = sc[j]; v[j] = -INFINITY; }
    const float w = __shfl(ws, best_e & 63);
    wsum += w;
    if (lane == kk) my_w = w;
    if (lane == 0) out_id[(int64_t)tok * topk + kk] = best_e;
  }
  if (lane < topk) out_w[(int64_t)tok * topk + lane] = my_w * (renormalize ? route_scale / wsum : route_scale);
}

}  // namespace xm

using namespace xm;

// scratch for the chunk histograms (nchunks * E int32); sized for 1M expanded rows x 1024 experts at most. Registered per
// device / per stream (workspace.hip, kind 1); the grouped GEMM keeps its tile table in the tail of the same buffer.
namespace xm {
void xm_moe_scratch(void* stream, void** ws, size_t* bytes) { ws_get(1, stream, ws, bytes); }
}  // namespace xm

extern "C" {

XM_API int xllm_mi355_set_moe_workspace(void* ws, size_t bytes) { return ws_set_device(1, ws, bytes); }
XM_API int xllm_mi355_set_moe_workspace_for_stream(void* stream, void* ws, size_t bytes) {
  return ws_set_stream(1, stream, ws, bytes);
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
  void* sc_v = nullptr;
  size_t sc_bytes = 0;
  xm_moe_scratch(stream, &sc_v, &sc_bytes);
  int32_t* const sc = reinterpret_cast<int32_t*>(sc_v);
  if (!sc || sc_bytes / 4 < (size_t)nchunks * E) return XM_ERR_WORKSPACE;
  if (nchunks == 1) {   // a decode step: histogram, scan and placement in one workgroup, one launch
    hipLaunchKernelGGL(moe_place_kernel<true>, dim3(1), dim3(1024), (size_t)16 * E * sizeof(int32_t), s, expert_id, n, E,
                       (const int32_t*)nullptr, src_dst, dst_src, expert_sizes);
    return hip_check_launch();
  }
  hipLaunchKernelGGL(moe_hist_kernel, dim3(nchunks), dim3(256), E * sizeof(int32_t), s, expert_id, n, E, sc);
  hipLaunchKernelGGL(moe_scan_kernel, dim3(1), dim3(1024), 0, s, sc, nchunks, E, expert_sizes);
  hipLaunchKernelGGL(moe_place_kernel<false>, dim3(nchunks), dim3(1024), (size_t)16 * E * sizeof(int32_t), s, expert_id, n, E, sc,
                     src_dst, dst_src, (int32_t*)nullptr);
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

int xllm_mi355_moe_combine_sorted(void* out, const void* gemm2_sorted, const int32_t* src_dst, const float* weights,
                                  int64_t n_tokens, int64_t topk, int64_t hidden, int dtype, void* stream) {
  if (!out || !gemm2_sorted || !src_dst || !weights || n_tokens < 0 || topk <= 0 || hidden <= 0) return XM_ERR_INVALID;
  if (dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (topk > 16 || hidden % 8 != 0 || ((uintptr_t)gemm2_sorted % 16) || ((uintptr_t)out % 16)) return XM_ERR_UNSUPPORTED;
  if (n_tokens == 0) return XM_OK;
  XM_DISPATCH_HALF(dtype, T, {
    if (topk == 8)
      hipLaunchKernelGGL((moe_combine_sorted_kernel<T, true>), dim3(n_tokens), dim3(256), 0, (hipStream_t)stream,
                         (T*)out, (const T*)gemm2_sorted, src_dst, weights, (int)topk, (int)hidden, (const int32_t*)nullptr, 0);
    else
      hipLaunchKernelGGL((moe_combine_sorted_kernel<T, false>), dim3(n_tokens), dim3(256), 0, (hipStream_t)stream,
                         (T*)out, (const T*)gemm2_sorted, src_dst, weights, (int)topk, (int)hidden, (const int32_t*)nullptr, 0);
  });
  return hip_check_launch();
}

int xllm_mi355_moe_combine_sorted_local(void* out, const void* gemm2_sorted, const int32_t* src_dst, const float* weights,
                                        const int32_t* local_expert_sizes, int64_t n_local_experts, int64_t n_tokens,
                                        int64_t topk, int64_t hidden, int dtype, void* stream) {
  if (!out || !gemm2_sorted || !src_dst || !weights || !local_expert_sizes || n_local_experts <= 0 || n_tokens < 0 ||
      topk <= 0 || hidden <= 0)
    return XM_ERR_INVALID;
  if (dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (topk > 16 || hidden % 8 != 0 || ((uintptr_t)gemm2_sorted % 16) || ((uintptr_t)out % 16)) return XM_ERR_UNSUPPORTED;
  if (n_tokens == 0) return XM_OK;
  XM_DISPATCH_HALF(dtype, T,
                   hipLaunchKernelGGL((moe_combine_sorted_kernel<T, false>), dim3(n_tokens), dim3(256), 0, (hipStream_t)stream,
                                      (T*)out, (const T*)gemm2_sorted, src_dst, weights, (int)topk, (int)hidden,
                                      local_expert_sizes, (int)n_local_experts));
  return hip_check_launch();
}

}  // extern "C"

extern "C" int xllm_mi355_moe_fused_topk(const void* gating, int dtype, int64_t n_tokens, int64_t n_experts, int64_t topk,
                                         int renormalize, const float* correction_bias, int scoring,
                                         float* topk_weights, int32_t* topk_ids, void* stream) {
  if (!gating || !topk_weights || !topk_ids || n_tokens < 0 || n_experts <= 0 || topk <= 0) return XM_ERR_INVALID;
  if (scoring != 0 && scoring != 1) return XM_ERR_INVALID;
  if (scoring == 0 && correction_bias) return XM_ERR_INVALID;  // the reference's softmax path drops the bias (:47-53)
  if (n_experts > 64 * kTopkMaxPerLane || topk > 64 || topk > n_experts) return XM_ERR_UNSUPPORTED;
  if (n_tokens == 0) return XM_OK;
  const dim3 grid((unsigned)((n_tokens + 3) / 4));
#define XM_FUSED_TOPK(PL_)                                                                                          \
  hipLaunchKernelGGL((moe_fused_topk_kernel<T, PL_>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)gating,     \
                     (int)n_tokens, (int)n_experts, (int)topk, renormalize, correction_bias, scoring, topk_weights,  \
                     topk_ids)
  XM_DISPATCH_FLOAT(dtype, T, {
    if (n_experts <= 64) XM_FUSED_TOPK(1);
    else if (n_experts <= 128) XM_FUSED_TOPK(2);
    else if (n_experts <= 256) XM_FUSED_TOPK(4);
    else XM_FUSED_TOPK(8);
  });
#undef XM_FUSED_TOPK
  return hip_check_launch();
}

extern "C" int xllm_mi355_moe_grouped_topk(const void* gating, int dtype, int64_t n_tokens, int64_t n_experts, int64_t topk,
                                           int64_t num_expert_group, int64_t topk_group, int renormalize,
                                           const float* correction_bias, int scoring, float routed_scaling_factor,
                                           float* topk_weights, int32_t* topk_ids, void* stream) {
  if (!gating || !topk_weights || !topk_ids || n_tokens < 0 || n_experts <= 0 || topk <= 0) return XM_ERR_INVALID;
  if (scoring != 0 && scoring != 1) return XM_ERR_INVALID;
  if (scoring == 0 && correction_bias) return XM_ERR_INVALID;  // topk_gate.cpp:96-98: the bias needs sigmoid scoring
  if (num_expert_group <= 1 || topk_group <= 0 || topk_group > num_expert_group) return XM_ERR_INVALID;  // :73-80
  if (n_experts % num_expert_group != 0) return XM_ERR_INVALID;
  const int64_t per_group = n_experts / num_expert_group;
  if (topk > topk_group * per_group || (correction_bias && per_group < 2)) return XM_ERR_INVALID;
  if (n_experts > 64 * kTopkMaxPerLane || topk > 64 || num_expert_group > 64) return XM_ERR_UNSUPPORTED;
  if (n_tokens == 0) return XM_OK;
  const dim3 grid((unsigned)((n_tokens + 3) / 4));
#define XM_GROUPED_TOPK(PL_)                                                                                        \
  hipLaunchKernelGGL((moe_grouped_topk_kernel<T, PL_>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)gating,   \
                     (int)n_tokens, (int)n_experts, (int)topk, (int)num_expert_group, (int)topk_group, renormalize,  \
                     correction_bias, scoring, routed_scaling_factor, topk_weights, topk_ids)
  XM_DISPATCH_FLOAT(dtype, T, {
    if (n_experts <= 64) XM_GROUPED_TOPK(1);
    else if (n_experts <= 128) XM_GROUPED_TOPK(2);
    else if (n_experts <= 256) XM_GROUPED_TOPK(4);
    else XM_GROUPED_TOPK(8);
  });
#undef XM_GROUPED_TOPK
  return hip_check_launch();
}
