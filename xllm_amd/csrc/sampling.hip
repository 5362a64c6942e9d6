// sampling.hip -- N3: the sampler kernels of the decode step.
//   random_sample     reference: dcu::random_sample (kernels/dcu/random_sample.hip:88-270): one token per row of
//                     probs [B, V] by CDF inversion: u ~ U(0,1] from Philox4x32-10 (seed, subsequence = row, offset),
//                     token = first index with p > 0 whose inclusive prefix sum exceeds u; rows whose total stays
//                     below u fall back to their last index with p > 0 (0 if there is none).
//   rejection_sample  reference: dcu::rejection_sample (kernels/dcu/rejection_sample.hip:33-139): speculative decoding
//                     accept / recover / bonus, fully determined by its inputs (bit-exact integer output).
// MI355X design of random_sample: a 608-KiB row (V = 152064) is read ONCE with every load independent of the others
// (no block-wide reduce + barrier per 2048-element chunk as in the reference): pass 1 leaves one partial sum per
// (4096-element segment, wave) in LDS, one thread walks the <= 64 segment totals in index order to find the crossing
// segment, and only that 16-KiB segment is re-read (L2 hit) for the in-segment scan. The prefix sums are fp32 like
// the reference's, in a different association order: the selected index can differ from the reference's only when u
// lies within fp32 summation error of a CDF step (tests bound it with an fp64 CDF).
#include "gemm_types.h"

namespace xm {

// ---- Philox4x32-10 (Salmon et al., SC'11; the generator behind hiprandStatePhilox4_32_10_t) --------------------
struct u32x4s { uint32_t x, y, z, w; };
__host__ __device__ __forceinline__ u32x4s philox4x32_10(u32x4s c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t m0 = (uint64_t)0xD2511F53u * c.x, m1 = (uint64_t)0xCD9E8D57u * c.z;
    const u32x4s n = {(uint32_t)(m1 >> 32) ^ c.y ^ k0, (uint32_t)m1, (uint32_t)(m0 >> 32) ^ c.w ^ k1, (uint32_t)m0};
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
// the first uniform of hiprand_init(seed, subsequence, offset) + hiprand_uniform: counter = (offset / 4, subsequence),
// output word offset % 4, u = 2^-32 + x * 2^-32 in fp32 (rocrand_uniform.h: never 0, may round to 1)
__host__ __device__ __forceinline__ float philox_first_uniform(uint64_t seed, uint64_t subsequence, uint64_t offset) {
  const uint64_t blk = offset >> 2;
  const u32x4s ctr = {(uint32_t)blk, (uint32_t)(blk >> 32), (uint32_t)subsequence, (uint32_t)(subsequence >> 32)};
  const u32x4s r = philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  const uint32_t sel = (uint32_t)(offset & 3);
  const uint32_t x = sel == 0 ? r.x : (sel == 1 ? r.y : (sel == 2 ? r.z : r.w));
  return 2.3283064e-10f + ((float)x * 2.3283064e-10f);
}

__global__ __launch_bounds__(256) void philox_uniform_kernel(float* __restrict__ out, int64_t n, uint64_t seed,
                                                             uint64_t offset) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = philox_first_uniform(seed, (uint64_t)i, offset);
}

// ---- random_sample ------------------------------------------------------------------------------------------------
constexpr int kRsThreads = 1024, kRsSeg = kRsThreads * 4, kRsMaxSeg = 64, kRsWaves = kRsThreads / 64;

__device__ __forceinline__ float rs_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <bool VEC>
__global__ __launch_bounds__(kRsThreads) void random_sample_kernel(const float* __restrict__ probs,
                                                                  int32_t* __restrict__ out, int d,
                                                                  const float* __restrict__ uniform, uint64_t seed,
                                                                  uint64_t offset) {
  __shared__ float part[kRsMaxSeg][kRsWaves];
  __shared__ float wave_tot[kRsWaves];
  __shared__ int wave_last[kRsWaves];
  __shared__ int s_seg, s_sampled, s_last, s_seg_last;
  __shared__ float s_prefix;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row = blockIdx.x;
  const float* p = probs + row * (int64_t)d;
  const float u = uniform ? uniform[row] : philox_first_uniform(seed, (uint64_t)row, offset);
  const int nseg = (d + kRsSeg - 1) / kRsSeg;

  auto load4 = [&](int base, float (&v)[4]) {
    if (VEC && base + 3 < d) {
      const float4 t = *reinterpret_cast<const float4*>(p + base);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = base + j < d ? p[base + j] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.0f;  // p <= 0 and NaN do not count (reference :148)
  };

  // pass 1: per-(segment, wave) partial sums + the last index with p > 0
  int last_valid = -1;
  for (int s = 0; s < nseg; ++s) {
    const int base = s * kRsSeg + tid * 4;
    float v[4];
    load4(base, v);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (v[j] > 0.0f) last_valid = base + j;
    const float ws = rs_wave_sum((v[0] + v[1]) + (v[2] + v[3]));
    if (lane == 0) part[s][wave] = ws;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(last_valid, o); last_valid = t > last_valid ? t : last_valid; }
  if (lane == 0) wave_last[wave] = last_valid;
  __syncthreads();
  // walk the segment totals in index order (one thread: <= 64 x 16 adds)
  if (tid == 0) {
    float agg = 0.0f;
    int seg = -1;
    for (int s = 0; s < nseg; ++s) {
      float tot = 0.0f;
#pragma unroll
      for (int w = 0; w < kRsWaves; ++w) tot += part[s][w];
      if (agg + tot > u) { seg = s; break; }
      agg += tot;
    }
    int lv = -1;
    for (int w = 0; w < kRsWaves; ++w) lv = wave_last[w] > lv ? wave_last[w] : lv;
    s_seg = seg;
    s_prefix = agg;
    s_last = lv;
    s_sampled = d;
    s_seg_last = -1;
  }
  __syncthreads();
  const int seg = s_seg;
  if (seg >= 0) {
    // in-segment inclusive scan: thread-sequential over its 4 values, then wave scan, then the wave totals in order
    const int base = seg * kRsSeg + tid * 4;
    float v[4];
    load4(base, v);
    float c[4];
    c[0] = v[0];
    c[1] = c[0] + v[1];
    c[2] = c[1] + v[2];
    c[3] = c[2] + v[3];
    float incl = c[3];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    float pre = s_prefix;
    for (int w = 0; w < wave; ++w) pre += wave_tot[w];
    pre += incl - c[3];  // exclusive prefix of this thread
    int cand = d;
#pragma unroll
    for (int j = 3; j >= 0; --j)
      if (v[j] > 0.0f && pre + c[j] > u) cand = base + j;
    if (cand < d) atomicMin(&s_sampled, cand);
    // the segment was chosen from the wave-tree sums, the crossing is looked for with the sequential scan: when u falls in
    // the rounding gap between the two orders no index crosses, and the answer is the segment's own last p > 0 index
    int seg_last = -1;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (v[j] > 0.0f) seg_last = base + j;
    if (seg_last >= 0) atomicMax(&s_seg_last, seg_last);
    __syncthreads();
  }
  if (tid == 0) {
    int r = s_sampled;
    if (r >= d) r = s_seg_last >= 0 ? s_seg_last : (s_last >= 0 ? s_last : 0);  // sum(probs) <= u (u ~ 1): last valid index of
                                                                                // the row (reference :231-235)
    out[row] = r;
  }
}

// ---- softmax + random_sample in one launch (round 6) -----------------------------------------------------------------
// Sampler::forward's tail (framework/sampling/sampler.cpp:118-137): probs = softmax(sample_logits, -1, fp32); samples =
// random_sample(probs) | greedy_sample(probs) | where(do_sample, random, greedy). The unfused form materialises [B, V] fp32
// probabilities (and, in this repository's engine until round 5, an fp32 copy of the logits in front of it): 156 MB written and
// read back per decode step at B = 256. Here one workgroup per row reads the (already temperature-scaled, top-k / top-p masked)
// logits of ANY dtype three times from L2 / Infinity Cache -- row maximum; e = exp(x - max) with one partial sum per (4096-element
// segment, wave) in LDS and the row total Z; the crossing segment again for the in-segment scan -- and writes one token id.
// Arithmetic: p_i = e_i / Z is never rounded on its own; the prefix sums run over e_i (fp32, the association of
// random_sample_kernel) and are compared against u * Z. Against softmax -> random_sample the selected index can differ only
// when u lies within fp32 rounding of a CDF step (the caveat random_sample already carries; tests bound it with an fp64 CDF).
// -inf logits (masked columns) and NaN carry no mass; a row without a finite maximum yields index 0 (the reference's probs are
// NaN there and no index has p > 0). Greedy rows (do_sample[b] == 0): the FIRST column holding the row maximum = argmax(probs).
template <typename T>
__global__ __launch_bounds__(kRsThreads) void softmax_random_sample_kernel(const T* __restrict__ logits, int64_t row_stride,
                                                                          int32_t* __restrict__ out, int d,
                                                                          const float* __restrict__ uniform, uint64_t seed,
                                                                          uint64_t offset, const uint8_t* __restrict__ do_sample) {
  constexpr int VEC = 16 / sizeof(T);          // elements per 16-byte load
  constexpr int SEG = kRsThreads * VEC;        // elements per segment
  constexpr int MAXSEG = kRsSeg * kRsMaxSeg / SEG < 1 ? 1 : kRsSeg * kRsMaxSeg / SEG;
  __shared__ float part[kRsMaxSeg][kRsWaves];
  __shared__ float wave_red[kRsWaves];
  __shared__ int wave_idx[kRsWaves];
  __shared__ float wave_tot[kRsWaves];
  __shared__ int s_seg, s_sampled, s_seg_last, s_last;
  __shared__ float s_prefix, s_target;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row = blockIdx.x;
  const T* p = logits + row * row_stride;
  const bool vec = (d % VEC == 0) && (((uintptr_t)p) % 16 == 0);
  const int nseg = (d + SEG - 1) / SEG;
  (void)MAXSEG;

  auto loadv = [&](int base, float (&v)[VEC]) {   // NaN and out-of-range columns read as -inf: no mass, never the maximum
    if (vec && base + VEC - 1 < d) {
      const uint4 raw = *reinterpret_cast<const uint4*>(p + base);
      T e[VEC];
      __builtin_memcpy(e, &raw, 16);
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = to_f32<T>(e[j]);
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[j] = base + j < d ? to_f32<T>(p[base + j]) : -__builtin_inff();
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = v[j] == v[j] ? v[j] : -__builtin_inff();
  };

  // ---- pass 1: row maximum and its first column
  float mx = -__builtin_inff();
  int mi = 0x7fffffff;
  for (int s = 0; s < nseg; ++s) {
    const int base = s * SEG + tid * VEC;
    float v[VEC];
    loadv(base, v);
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      if (v[j] > mx) { mx = v[j]; mi = base + j; }    // strictly greater: the first column of a thread's maximum
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(mx, o);
    const int oi = __shfl_xor(mi, o);
    if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
  }
  if (lane == 0) { wave_red[wave] = mx; wave_idx[wave] = mi; }
  __syncthreads();
  mx = wave_red[0];
  mi = wave_idx[0];
  for (int w = 1; w < kRsWaves; ++w)
    if (wave_red[w] > mx || (wave_red[w] == mx && wave_idx[w] < mi)) { mx = wave_red[w]; mi = wave_idx[w]; }
  __syncthreads();
  const bool finite_max = mx > -__builtin_inff() && mx < __builtin_inff();
  if (!finite_max || (do_sample && !do_sample[row])) {
    // greedy row: argmax(probs) = the first column of the maximum; a row with no finite maximum has no probability mass at all
    // (softmax gives NaN): index 0 like random_sample's "no p > 0" answer -- except a +inf maximum, whose first column wins
    if (tid == 0) out[row] = (mx == __builtin_inff() || (finite_max && mi != 0x7fffffff)) ? mi : 0;
    return;
  }
  const float u = uniform ? uniform[row] : philox_first_uniform(seed, (uint64_t)row, offset);

  // ---- pass 2: e = exp(x - max); one partial sum per (segment, wave); last column with mass
  int last_valid = -1;
  for (int s = 0; s < nseg; ++s) {
    const int base = s * SEG + tid * VEC;
    float v[VEC];
    loadv(base, v);
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float e = expf(v[j] - mx);     // exp(-inf) = 0: masked columns carry no mass
      if (e > 0.0f) last_valid = base + j;
      sum += e;
    }
    const float ws = rs_wave_sum(sum);
    if (lane == 0) part[s][wave] = ws;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(last_valid, o); last_valid = t > last_valid ? t : last_valid; }
  if (lane == 0) wave_idx[wave] = last_valid;
  __syncthreads();
  if (tid == 0) {
    float z = 0.0f;
    for (int s = 0; s < nseg; ++s) {
      float tot = 0.0f;
#pragma unroll
      for (int w = 0; w < kRsWaves; ++w) tot += part[s][w];
      z += tot;
    }
    const float target = u * z;            // prefix(e) > u * Z  <=>  prefix(e / Z) > u up to rounding
    float agg = 0.0f;
    int seg = -1;
    for (int s = 0; s < nseg; ++s) {
      float tot = 0.0f;
#pragma unroll
      for (int w = 0; w < kRsWaves; ++w) tot += part[s][w];
      if (agg + tot > target) { seg = s; break; }
      agg += tot;
    }
    int lv = -1;
    for (int w = 0; w < kRsWaves; ++w) lv = wave_idx[w] > lv ? wave_idx[w] : lv;
    s_seg = seg;
    s_prefix = agg;
    s_target = target;
    s_last = lv;
    s_sampled = d;
    s_seg_last = -1;
  }
  __syncthreads();
  const int seg = s_seg;
  if (seg >= 0) {
    const float target = s_target;
    const int base = seg * SEG + tid * VEC;
    float v[VEC], c[VEC];
    loadv(base, v);
    float run = 0.0f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { v[j] = expf(v[j] - mx); run += v[j]; c[j] = run; }
    float incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    float pre = s_prefix;
    for (int w = 0; w < wave; ++w) pre += wave_tot[w];
    pre += incl - run;
    int cand = d, seg_last = -1;
#pragma unroll
    for (int j = VEC - 1; j >= 0; --j)
      if (v[j] > 0.0f && pre + c[j] > target) cand = base + j;
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      if (v[j] > 0.0f) seg_last = base + j;
    if (cand < d) atomicMin(&s_sampled, cand);
    if (seg_last >= 0) atomicMax(&s_seg_last, seg_last);
    __syncthreads();
  }
  if (tid == 0) {
    int r = s_sampled;
    if (r >= d) r = s_seg_last >= 0 ? s_seg_last : (s_last >= 0 ? s_last : 0);
    out[row] = r;
  }
}

// ---- rejection_sample ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rejection_sample_kernel(
    const int32_t* __restrict__ draft_token_ids, const int32_t* __restrict__ num_draft_tokens,
    const int32_t* __restrict__ cu_num_draft_tokens, const float* __restrict__ draft_probs,
    const float* __restrict__ target_probs, const int32_t* __restrict__ bonus_token_ids,
    const float* __restrict__ uniform_rand, const float* __restrict__ uniform_probs, int vocab,
    int32_t* __restrict__ output) {
  __shared__ float w_score[4];
  __shared__ int w_token[4];
  const int seq = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n_draft = num_draft_tokens[seq];
  const int draft_end = cu_num_draft_tokens[seq];
  const int draft_start = draft_end - n_draft;
  const int out_start = draft_start + seq;
  for (int i = tid; i < n_draft + 1; i += blockDim.x) output[out_start + i] = -1;
  __syncthreads();
  for (int di = 0; di < n_draft; ++di) {
    const int row = draft_start + di;
    const int tok = draft_token_ids[row];
    if (tok < 0 || tok >= vocab) return;  // uniform over the block
    const int64_t ro = (int64_t)row * vocab;
    const float dp = draft_probs[ro + tok] > 0.0f ? draft_probs[ro + tok] : 0.0f;
    const float tp = target_probs[ro + tok] > 0.0f ? target_probs[ro + tok] : 0.0f;
    const float accept = dp > 0.0f ? tp / dp : (tp > 0.0f ? 1.0f : 0.0f);
    if (uniform_rand[row] < accept) {
      if (tid == 0) output[out_start + di] = tok;
      continue;
    }
    // rejected: recovered token = argmax_t max(target - draft, 0) / max(u_t, FLT_MIN), lowest index on ties
    float best = -1.0f;
    int best_tok = 0;
    for (int t = tid; t < vocab; t += blockDim.x) {
      const float rec = fmaxf(target_probs[ro + t] - draft_probs[ro + t], 0.0f);
      const float uu = fmaxf(uniform_probs[ro + t], 1.17549435e-38f);
      const float sc = rec / uu;
      if (sc > best || (sc == best && t < best_tok)) { best = sc; best_tok = t; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float os = __shfl_xor(best, o);
      const int ot = __shfl_xor(best_tok, o);
      if (os > best || (os == best && ot < best_tok)) { best = os; best_tok = ot; }
    }
    if (lane == 0) { w_score[wave] = best; w_token[wave] = best_tok; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        if (w_score[w] > best || (w_score[w] == best && w_token[w] < best_tok)) { best = w_score[w]; best_tok = w_token[w]; }
      output[out_start + di] = best_tok;
    }
    return;
  }
  if (tid == 0) output[out_start + n_draft] = bonus_token_ids[seq];
}


// ---- greedy_argmax ------------------------------------------------------------------------------------------------
// The greedy branch of the sampler (reference: Sampler::greedy_sample = logits.argmax(-1),
// framework/sampling/sampler.cpp): one workgroup per row streams the row once with 16-byte loads (torch's generic reduce
// took 55 us for [256, 152064] bf16 in the decode step, 1.4 TB/s). torch.argmax semantics: the FIRST index of the maximum; a
// NaN is larger than every number (the first NaN wins).
template <typename T>
__device__ __forceinline__ float argmax_key(T v) { return to_f32(v); }
// (argmax_better: gemm_types.h -- shared with the lm_head GEMM's fused argmax epilogue)
template <typename T>
__global__ __launch_bounds__(1024) void greedy_argmax_kernel(const T* __restrict__ logits, int64_t* __restrict__ out, int d) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float sv[16];
  __shared__ int si[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* p = logits + (int64_t)blockIdx.x * d;
  float bv = -__builtin_inff();
  int bi = 0x7fffffff;
  const bool vec = (d % VEC == 0) && ((uintptr_t)p % 16 == 0);
  if (vec) {
    for (int base = tid * VEC; base < d; base += 1024 * VEC) {
      const uint4 raw = *reinterpret_cast<const uint4*>(p + base);
      T e[VEC];
      __builtin_memcpy(e, &raw, 16);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float v = argmax_key(e[j]);
        if (argmax_better(v, base + j, bv, bi)) { bv = v; bi = base + j; }
      }
    }
  } else {
    for (int i = tid; i < d; i += 1024) {
      const float v = argmax_key(p[i]);
      if (argmax_better(v, i, bv, bi)) { bv = v; bi = i; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o);
    const int oi = __shfl_xor(bi, o);
    if (argmax_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (argmax_better(sv[w], si[w], bv, bi)) { bv = sv[w]; bi = si[w]; }
    out[blockIdx.x] = bi == 0x7fffffff ? 0 : bi;
  }
}

}  // namespace xm

using namespace xm;

extern "C" {

int xllm_mi355_philox_uniform(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
  if (!out || n < 0) return XM_ERR_INVALID;
  if (n == 0) return XM_OK;
  hipLaunchKernelGGL(philox_uniform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, n,
                     seed, offset);
  return hip_check_launch();
}

int xllm_mi355_random_sample(const float* probs, int32_t* out, int64_t batch, int64_t vocab, const float* uniform,
                             uint64_t philox_seed, uint64_t philox_offset, void* stream) {
  if (!probs || !out || batch < 0 || vocab <= 0) return XM_ERR_INVALID;
  if (vocab > (int64_t)kRsSeg * kRsMaxSeg) return XM_ERR_UNSUPPORTED;
  if (batch == 0) return XM_OK;
  const bool vec = (vocab % 4 == 0) && ((uintptr_t)probs % 16 == 0);
  if (vec)
    hipLaunchKernelGGL((random_sample_kernel<true>), dim3((unsigned)batch), dim3(kRsThreads), 0, (hipStream_t)stream,
                       probs, out, (int)vocab, uniform, philox_seed, philox_offset);
  else
    hipLaunchKernelGGL((random_sample_kernel<false>), dim3((unsigned)batch), dim3(kRsThreads), 0, (hipStream_t)stream,
                       probs, out, (int)vocab, uniform, philox_seed, philox_offset);
  return hip_check_launch();
}

int xllm_mi355_softmax_random_sample(const void* logits, int32_t* out, int64_t batch, int64_t vocab, int64_t row_stride, int dtype,
                                     const float* uniform, uint64_t philox_seed, uint64_t philox_offset, const uint8_t* do_sample,
                                     void* stream) {
  if (!logits || !out || batch < 0 || vocab <= 0 || row_stride < vocab) return XM_ERR_INVALID;
  if (batch == 0) return XM_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t seg = (int64_t)kRsThreads * (dtype == XM_F32 ? 4 : 8);
  if ((vocab + seg - 1) / seg > kRsMaxSeg) return XM_ERR_UNSUPPORTED;
#define XM_SRS(T)                                                                                                         \
  hipLaunchKernelGGL((softmax_random_sample_kernel<T>), dim3((unsigned)batch), dim3(kRsThreads), 0, s, (const T*)logits,  \
                     row_stride, out, (int)vocab, uniform, philox_seed, philox_offset, do_sample)
  if (dtype == XM_F32) XM_SRS(float);
  else if (dtype == XM_BF16) XM_SRS(bf16_t);
  else if (dtype == XM_F16) XM_SRS(f16_t);
  else return XM_ERR_UNSUPPORTED;
#undef XM_SRS
  return hip_check_launch();
}

int xllm_mi355_greedy_argmax(const void* logits, int64_t* out, int64_t batch, int64_t vocab, int dtype, void* stream) {
  if (!logits || !out || batch < 0 || vocab <= 0 || vocab >= (1ll << 31)) return XM_ERR_INVALID;
  if (batch == 0) return XM_OK;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == XM_BF16)
    hipLaunchKernelGGL((greedy_argmax_kernel<bf16_t>), dim3((unsigned)batch), dim3(1024), 0, s, (const bf16_t*)logits, out, (int)vocab);
  else if (dtype == XM_F16)
    hipLaunchKernelGGL((greedy_argmax_kernel<f16_t>), dim3((unsigned)batch), dim3(1024), 0, s, (const f16_t*)logits, out, (int)vocab);
  else if (dtype == XM_F32)
    hipLaunchKernelGGL((greedy_argmax_kernel<float>), dim3((unsigned)batch), dim3(1024), 0, s, (const float*)logits, out, (int)vocab);
  else
    return XM_ERR_UNSUPPORTED;
  return hip_check_launch();
}

int xllm_mi355_rejection_sample(const int32_t* draft_token_ids, const int32_t* num_draft_tokens,
                                const int32_t* cu_num_draft_tokens, const float* draft_probs,
                                const float* target_probs, const int32_t* bonus_token_ids, const float* uniform_rand,
                                const float* uniform_probs, int64_t batch, int64_t vocab, int32_t* output,
                                void* stream) {
  if (!draft_token_ids || !num_draft_tokens || !cu_num_draft_tokens || !draft_probs || !target_probs ||
      !bonus_token_ids || !uniform_rand || !uniform_probs || !output || batch < 0 || vocab <= 0)
    return XM_ERR_INVALID;
  if (batch == 0) return XM_OK;
  hipLaunchKernelGGL(rejection_sample_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, draft_token_ids,
                     num_draft_tokens, cu_num_draft_tokens, draft_probs, target_probs, bonus_token_ids, uniform_rand,
                     uniform_probs, (int)vocab, output);
  return hip_check_launch();
}

}  // extern "C"
