// logits_processors.hip -- N3: the logits processors in front of the sampler.
// Reference: framework/sampling/logits_utils.cpp:24-155, called by Sampler::forward (framework/sampling/sampler.cpp:33-116) in
// this order: frequency + presence penalties -> repetition penalties -> temperatures -> top-k -> top-p -> softmax -> sample.
//   apply_penalties      logits_utils.cpp:24-52: score = logits.gather(1, ids); score -= counts * freq; score -= (counts > 0) * presence;
//                        scatter; then score = gather again; score < 0 ? score * rep : score / rep; scatter. ids are the padded
//                        [B, U] unique-token table of SamplingParameters (padding repeats id 0 with count 0, sampling_params.cpp:
//                        127-134), so a column may appear several times in a row: every copy is computed from the row as it was
//                        BEFORE the call (gather, then scatter), never from a sibling's result.
//   apply_temperatures   :54-64: logits /= (t == 0 ? 1 : t), per row.
//   apply_top_k_top_p    :66-155. The reference sorts every row (torch.sort, 152064 logits), masks ranks >= k with -inf, softmaxes
//                        the sorted row, masks by the cumulative probability and scatters back. Two rules exist in the file:
//                        (a) "one of them" (:121-153, what a CUDA / DCU build runs): top_k <= 0 disables; top-p masks rank i when
//                            cumsum(probs)[i] - probs[i] > p  (the EXCLUSIVE prefix: rank 0 always survives);
//                        (b) "both" (apply_top_k_top_p_torch_impl, :66-90): k = clamp(top_k, 1, V); top-p masks rank i > 0 when
//                            cumsum(probs)[i] > p (the INCLUSIVE prefix). On a CUDA / DCU build the both-defined case falls
//                            through the NPU / MLU #if (:107-120) and nothing is applied; this backend applies rule (b) -- the
//                            function the reference ships for it -- and says so.
// MI355X design of top-k / top-p: NO SORT. A row is reduced by radix selection on an order-preserving 32-bit key of the logit
// (4 passes of 8 bits each, 256-bin LDS histograms): the k-th largest key and how many of its ties survive; then the same
// descent over PROBABILITY MASS (exp(x - max) in 2^-40 fixed point, 64-bit integer LDS atomics: exact and order-independent, so
// the result is deterministic) finds the key at which the cumulative probability crosses p. Ties at a boundary key are ranked
// by column index (a stable descending sort). One workgroup per row, 12 coalesced passes over the row from L2.
#include "common.h"

namespace xm {

constexpr int kLpThreads = 1024, kLpWaves = kLpThreads / 64;

template <typename T>
__device__ __forceinline__ float lp_load(const T* p, int64_t i) { return to_f32<T>(p[i]); }

// ---- penalties: phase 1 gathers + computes into a [B, U] scratch, phase 2 scatters (duplicates then write identical values)
template <typename T>
__global__ __launch_bounds__(256) void penalties_gather_kernel(const T* __restrict__ logits, int64_t row_stride, int64_t V,
                                                              const int64_t* __restrict__ ids, const int32_t* __restrict__ counts,
                                                              int64_t U, const float* __restrict__ freq,
                                                              const float* __restrict__ presence, const float* __restrict__ rep,
                                                              float* __restrict__ scratch) {
  const int64_t b = blockIdx.y;
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= U) return;
  const int64_t id = ids[b * U + u];
  float s = 0.0f;
  if (id >= 0 && id < V) {
    s = lp_load(logits, b * row_stride + id);
    if (freq) {   // logits_utils.cpp:30-32 (two in-place subtractions on the logits dtype)
      const int c = counts[b * U + u];
      s = r16<T>(s - (float)c * freq[b]);
      s = r16<T>(s - (c > 0 ? presence[b] : 0.0f));
    }
    if (rep) s = r16<T>(s < 0.0f ? s * rep[b] : s / rep[b]);   // :44-51
  }
  scratch[b * U + u] = s;
}
template <typename T>
__global__ __launch_bounds__(256) void penalties_scatter_kernel(T* __restrict__ logits, int64_t row_stride, int64_t V,
                                                               const int64_t* __restrict__ ids, int64_t U,
                                                               const float* __restrict__ scratch) {
  const int64_t b = blockIdx.y;
  const int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= U) return;
  const int64_t id = ids[b * U + u];
  if (id >= 0 && id < V) logits[b * row_stride + id] = from_f32<T>(scratch[b * U + u]);
}

template <typename T>
__global__ __launch_bounds__(256) void temperatures_kernel(T* __restrict__ logits, int64_t row_stride, int64_t V,
                                                          const float* __restrict__ temperatures) {
  const int64_t b = blockIdx.y;
  float t = temperatures[b];
  t = t == 0.0f ? 1.0f : t;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x)
    logits[b * row_stride + i] = from_f32<T>(lp_load(logits, b * row_stride + i) / t);
}

// order-preserving key: a larger float is a larger unsigned integer (-0 < +0; NaNs sort above +inf / below -inf by sign)
__device__ __forceinline__ uint32_t lp_key(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float lp_unkey(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
// probability mass of a logit in 2^-40 fixed point (x <= row max: the mass is in (0, 1]). The mass is ALWAYS fp32 exp(x - max):
// for fp32 logits (what the engine passes, engine.py: logits(hidden).float()) this is the reference's softmax -> fp32 cumsum; for
// 16-bit logits the reference's one-of-them branch runs softmax in the logits dtype first (logits_utils.cpp:145), i.e. rounds each
// probability to 16 bit before the fp32 cumsum, so its cut-off rank can differ from this kernel's by the ranks whose cumulative
// mass lies within that rounding of p. Parity with the sorting reference is therefore asserted for fp32 logits (exact up to
// boundary cases) and for 16-bit logits only up to such boundary ranks (tests/test_gpu_parity.py::
// test_apply_top_k_top_p_matches_the_sorting_reference).
// (a NaN logit has no mass: the float -> integer conversion of a NaN is undefined behaviour)
__device__ __forceinline__ unsigned long long lp_mass(float x, float mx) {
  const float e = __expf(x - mx) * 1099511627776.0f;
  return e == e ? (unsigned long long)e : 0ull;
}

// The histograms are filled through kLpRep lane-salted replicas (round 6): most logits of a row share a handful of exponent bins, so
// the 64 lanes of an atomic instruction hit ONE counter and serialise (measured: the first digit's sweep dominated the kernel on
// model-like logits); with bin * 16 + (lane & 15) at most four lanes share an address. The replicas are summed before the scan.
constexpr int kLpRep = 16;
struct LpShared {
  unsigned int cnt[256];
  unsigned long long mass[256];
  unsigned int cnt_r[256 * kLpRep];
  unsigned long long mass_r[256 * kLpRep];
  unsigned int wave_ties[2][kLpWaves];
  float red[kLpWaves];
  // broadcast slots
  uint32_t prefix;
  unsigned long long carried;
  unsigned int remaining;
  unsigned int ties;             // columns equal to the k-th key (counted by the first top-p sweep)
  unsigned int kth_live;         // ... of which top-k keeps this many
  unsigned long long target;     // p * Z in 2^-40 fixed point
};

// top-k / top-p for one row. rule: 0 = "one of them" (exclusive prefix), 1 = "both" (inclusive prefix, at least one); k <= 0 disables top-k under both
template <typename T>
__global__ __launch_bounds__(kLpThreads) void top_k_top_p_kernel(T* __restrict__ logits, int64_t row_stride, int V,
                                                                const float* __restrict__ temperatures,
                                                                const int64_t* __restrict__ top_k, const float* __restrict__ top_p,
                                                                int rule) {
  __shared__ LpShared sh;
  // 16-bit logits (also after the temperature division, which rounds to the dtype) are fp32 values whose low 16 (bf16) / 13 (f16)
  // bits are zero, so the low 16 / 8 bits of every key are a function of its sign alone (zeros, or ones for a negative value):
  // two (three) radix digits select, the remaining sweeps over the row would only re-count the survivors into one bin
  // (round 6: 13 -> 9 sweeps per row for bf16). lp_low() completes a selected prefix with those constant bits.
  constexpr int kLowShift = __is_same(T, bf16_t) ? 16 : (__is_same(T, f16_t) ? 8 : 0);
  auto lp_low = [](uint32_t prefix) -> uint32_t {
    if constexpr (kLowShift == 0) return prefix;
    else return (prefix & 0x80000000u) ? prefix : (prefix | ((1u << kLowShift) - 1u));   // key bit 31 clear = negative value
  };
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T* const row = logits + (int64_t)b * row_stride;
  // temperature: the division the reference does in place first (apply_temperatures); here folded into every load and written
  // back by the final pass. (x / t in fp32, rounded to the logits dtype: the reference's div_ on a 16-bit tensor does the same)
  float temp = 1.0f;
  bool scale = false;
  if (temperatures) {
    const float t = temperatures[b];
    if (t != 0.0f && t != 1.0f) { temp = t; scale = true; }
  }
  // (a division, not a reciprocal multiply: bit-equal to logits.div_(t))
  auto value = [&](int i) { const float x = lp_load(row, i); return scale ? r16<T>(x / temp) : x; };
  // One sweep over the row, order-free: f(value) for every column. 16-byte loads (8 / 4 columns per thread and instruction; round
  // 6: the sweeps were 2-byte loads, 150 per thread and sweep) when the row is 16-byte aligned, scalar head / tail otherwise.
  constexpr int VEC = 16 / (int)sizeof(T);
  const int v_lo = (int)((16 - ((uintptr_t)row & 15)) & 15) / (int)sizeof(T);      // columns before the first aligned one
  const int v_head = v_lo < V ? v_lo : V;
  const int v_n = (V - v_head) / VEC;                                               // whole vectors
  auto sweep = [&](auto&& f) {
    for (int i = tid; i < v_head; i += kLpThreads) f(value(i));
    for (int k = tid; k < v_n; k += kLpThreads) {
      const uint4 raw = *reinterpret_cast<const uint4*>(row + v_head + k * VEC);
      T e[VEC];
      __builtin_memcpy(e, &raw, 16);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float x = to_f32<T>(e[j]);
        f(scale ? r16<T>(x / temp) : x);
      }
    }
    for (int i = v_head + v_n * VEC + tid; i < V; i += kLpThreads) f(value(i));
  };

  const unsigned salt = (unsigned)lane & (kLpRep - 1);
  auto fold_cnt = [&]() {    // replicas -> sh.cnt (every thread of the block calls it, between two barriers)
    for (int bb = tid; bb < 256; bb += kLpThreads) {
      unsigned c = 0;
#pragma unroll
      for (int r = 0; r < kLpRep; ++r) c += sh.cnt_r[bb * kLpRep + r];
      sh.cnt[bb] = c;
    }
  };
  auto fold_mass = [&]() {
    for (int bb = tid; bb < 256; bb += kLpThreads) {
      unsigned long long m = 0ull;
#pragma unroll
      for (int r = 0; r < kLpRep; ++r) m += sh.mass_r[bb * kLpRep + r];
      sh.mass[bb] = m;
    }
  };
  // ---- pass 0: row maximum -- found by the sweep of the first top-k digit when there is one (round 6: one sweep less)
  long long k = top_k ? top_k[b] : 0;
  if (rule == 1 && k > V) k = V;
  const bool want_k = top_k && k > 0 && k < V;
  float mx = -__builtin_inff();
  if (want_k) {
    for (int i = tid; i < 256 * kLpRep; i += kLpThreads) sh.cnt_r[i] = 0;
    __syncthreads();
    sweep([&](float x) {
      mx = fmaxf(mx, x);
      atomicAdd(&sh.cnt_r[((lp_key(x) >> 24) & 255u) * kLpRep + salt], 1u);
    });
  } else {
    sweep([&](float x) { mx = fmaxf(mx, x); });
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) sh.red[wave] = mx;
  __syncthreads();
  mx = sh.red[0];
  for (int w = 1; w < kLpWaves; ++w) mx = fmaxf(mx, sh.red[w]);
  __syncthreads();

  // ---- top-k: the k-th largest key and how many of its ties stay
  // k <= 0 means "no limit" under BOTH rules: the reference's default is top_k = -1, the batch tensor exists as soon as ANY
  // request sets top_k > 0, and every reference branch that runs maps k <= 0 to "unlimited" (the one-of-them branch,
  // logits_utils.cpp:126-133, and the NPU both-given branch, :109-116). Only the otherwise unused torch_impl clamps to 1 (:70),
  // which would turn every request that left top_k at its default into greedy sampling in a mixed batch (round-4 advisor).
  // a row without a finite maximum (all -inf, all NaN) has no ranking and no probability mass (x - mx is NaN): it is left
  // unfiltered -- only the temperature is applied -- instead of running the selections on garbage (round-4 advisor)
  const bool degenerate = !(mx > -__builtin_inff());
  const bool use_k = want_k && !degenerate;
  uint32_t kth_key = 0u;             // keep every key >= kth_key ...
  unsigned k_rem = 0xffffffffu;      // ... but of the ties at kth_key only the first k_rem by index
  if (use_k) {
    uint32_t prefix = 0u;
    unsigned remaining = (unsigned)k;
    for (int shift = 24; shift >= kLowShift; shift -= 8) {
      const uint32_t hi_mask = shift == 24 ? 0u : (0xffffffffu << (shift + 8));
      if (shift != 24) {   // (the first digit's histogram was filled by the sweep that found the maximum)
        for (int i = tid; i < 256 * kLpRep; i += kLpThreads) sh.cnt_r[i] = 0;
        __syncthreads();
        sweep([&](float x) {
          const uint32_t kk = lp_key(x);
          if ((kk & hi_mask) == (prefix & hi_mask)) atomicAdd(&sh.cnt_r[((kk >> shift) & 255u) * kLpRep + salt], 1u);
        });
        __syncthreads();
      }
      fold_cnt();
      __syncthreads();
      if (tid == 0) {
        unsigned acc = 0;
        int bsel = 0;
        for (int bb = 255; bb >= 0; --bb) {
          if (acc + sh.cnt[bb] >= remaining) { bsel = bb; break; }
          acc += sh.cnt[bb];
        }
        sh.prefix = prefix | ((uint32_t)bsel << shift);
        sh.remaining = remaining - acc;
      }
      __syncthreads();
      prefix = sh.prefix;
      remaining = sh.remaining;
      __syncthreads();
    }
    kth_key = lp_low(prefix);
    k_rem = remaining;
  }

  // ---- top-p: the key K* at which the cumulative probability of the sorted, top-k-masked row crosses p
  uint32_t p_key = 0u;               // keep every key > p_key, and of the ties at p_key the first p_keep (by index)
  unsigned p_keep = 0xffffffffu;
  if (top_p && !degenerate) {
    const float p = top_p[b];
    const unsigned long long q_kth = lp_mass(lp_unkey(kth_key), mx);
    unsigned long long target = 0ull;
    uint32_t prefix = 0u;
    unsigned long long carried = 0ull;   // mass of every kept key above the current prefix range
    for (int shift = 24; shift >= kLowShift; shift -= 8) {
      const bool first = shift == 24;
      for (int i = tid; i < 256 * kLpRep; i += kLpThreads) { sh.cnt_r[i] = 0; sh.mass_r[i] = 0ull; }
      if (first && tid == 0) sh.ties = 0u;
      __syncthreads();
      const uint32_t hi_mask = first ? 0u : (0xffffffffu << (shift + 8));
      unsigned my_ties = 0;
      sweep([&](float x) {
        const uint32_t kk = lp_key(x);
        if ((kk & hi_mask) != (prefix & hi_mask) || kk < kth_key) return;
        const unsigned bb = (kk >> shift) & 255u;
        if (kk == kth_key && use_k) { ++my_ties; return; }   // the boundary ties are added once below, kth_live times
        atomicAdd(&sh.cnt_r[bb * kLpRep + salt], 1u);
        atomicAdd(&sh.mass_r[bb * kLpRep + salt], lp_mass(x, mx));
      });
      if (first && use_k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) my_ties += __shfl_xor(my_ties, o);
        if (lane == 0 && my_ties) atomicAdd(&sh.ties, my_ties);
      }
      __syncthreads();
      fold_cnt();
      fold_mass();
      __syncthreads();
      if (tid == 0) {
        if (first) {
          // Z = sum of the kept masses = this sweep's bins + the boundary ties top-k keeps (round 6: the separate Z sweep is gone);
          // target = p * Z in fixed point (fp64: exact enough for 2^-40 granules); p >= 1 keeps everything, p < 0 rank 0 only
          const unsigned kth_total = sh.ties;
          const unsigned live = use_k ? (k_rem < kth_total ? k_rem : kth_total) : 0u;
          unsigned long long Z = (unsigned long long)live * q_kth;
          for (int bb = 0; bb < 256; ++bb) Z += sh.mass[bb];
          const double tgt_d = (double)p * (double)Z;
          sh.target = tgt_d <= 0.0 ? 0ull : (tgt_d >= 1.8e19 ? ~0ull : (unsigned long long)tgt_d);
          sh.kth_live = live;
        }
        const unsigned long long tgt = sh.target;
        const unsigned kth_live = sh.kth_live;
        if (use_k && (kth_key & hi_mask) == (prefix & hi_mask)) {
          const unsigned bb = (kth_key >> shift) & 255u;
          sh.cnt[bb] += kth_live;
          sh.mass[bb] += (unsigned long long)kth_live * q_kth;
        }
        // the LOWEST non-empty bin whose mass-above is still <= target holds K* (the first non-empty bin always qualifies:
        // carried <= target is an invariant of the descent, 0 <= target at the top)
        unsigned long long acc = carried;
        int bsel = -1;
        unsigned long long acc_sel = carried;
        for (int bb = 255; bb >= 0; --bb) {
          if (sh.cnt[bb] == 0) continue;
          if (acc > tgt && bsel >= 0) break;
          bsel = bb;
          acc_sel = acc;
          acc += sh.mass[bb];
        }
        sh.prefix = prefix | ((uint32_t)(bsel < 0 ? 0 : bsel) << shift);
        sh.carried = acc_sel;
        sh.remaining = bsel < 0 ? 0u : sh.cnt[bsel];
      }
      __syncthreads();
      prefix = sh.prefix;
      carried = sh.carried;
      target = sh.target;
      __syncthreads();
    }
    p_key = lp_low(prefix);
    const unsigned grp = sh.remaining;                               // kept-by-top-k members of the K* group
    const unsigned long long q = lp_mass(lp_unkey(p_key), mx);
    // exclusive prefix of tie r in the group = carried + r * q; it survives while that is <= target
    unsigned long long n = 1;
    if (carried <= target) n = q > 0ull ? (target - carried) / q + 1ull : (unsigned long long)grp;
    unsigned keep = (unsigned)(n > grp ? grp : n);
    if (rule == 1) {
      // inclusive rule (:80-86): rank i > 0 survives iff its INCLUSIVE prefix = the exclusive prefix of rank i + 1 is <= target,
      // i.e. exactly one rank fewer than the exclusive rule keeps (p >= 1 keeps the row: the last inclusive prefix is Z itself)
      if (!(p >= 1.0f)) keep = keep > 0 ? keep - 1 : 0;
    }
    p_keep = keep;
    // at least one element survives (rank 0: the first column holding the row maximum)
    if (p_keep == 0 && p_key == lp_key(mx)) p_keep = 1;
  }

  // ---- ordered tie ranks + the final pass. Each wave owns a contiguous segment of the row. Rows that are 16-byte aligned with a
  // whole number of 16-byte vectors (the model's vocabularies) take the vector form (round 6): a lane holds VEC consecutive columns
  // per step, the index order inside a step is lane-major, so a tie's rank = ties of earlier waves + of earlier steps + of lower
  // lanes (a wave scan, only in steps that hold a tie) + of its own earlier columns. Otherwise: one column per lane and step.
  const bool need_k_rank = use_k, need_p_rank = top_p != nullptr;
  const float ninf = -__builtin_inff();
  if (v_head == 0 && v_n * VEC == V) {
    constexpr int STEP = 64 * VEC;
    const int seg = ((V + kLpWaves - 1) / kLpWaves + STEP - 1) / STEP * STEP;
    const int s0 = wave * seg, s1 = s0 + seg < V ? s0 + seg : V;
    auto load_keys = [&](int i0, float (&x)[VEC], uint32_t (&kk)[VEC]) {
      const uint4 raw = *reinterpret_cast<const uint4*>(row + i0);
      T e[VEC];
      __builtin_memcpy(e, &raw, 16);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float v = to_f32<T>(e[j]);
        x[j] = scale ? r16<T>(v / temp) : v;
        kk[j] = lp_key(x[j]);
      }
    };
    unsigned c_k = 0, c_p = 0;
    if (need_k_rank || need_p_rank) {
      for (int base = s0; base < s1; base += STEP) {
        const int i0 = base + lane * VEC;
        if (i0 >= s1) continue;
        float x[VEC];
        uint32_t kk[VEC];
        load_keys(i0, x, kk);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          c_k += (need_k_rank && kk[j] == kth_key) ? 1u : 0u;
          c_p += (need_p_rank && kk[j] == p_key) ? 1u : 0u;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { c_k += __shfl_xor(c_k, o); c_p += __shfl_xor(c_p, o); }
    }
    if (lane == 0) { sh.wave_ties[0][wave] = c_k; sh.wave_ties[1][wave] = c_p; }
    __syncthreads();
    unsigned r_k = 0, r_p = 0;
    for (int w = 0; w < wave; ++w) { r_k += sh.wave_ties[0][w]; r_p += sh.wave_ties[1][w]; }
    for (int base = s0; base < s1; base += STEP) {
      const int i0 = base + lane * VEC;
      const bool in = i0 < s1;
      float x[VEC];
      uint32_t kk[VEC];
      if (in) load_keys(i0, x, kk);
      unsigned mk = 0, mp = 0;          // this lane's tie columns, one bit each
      if (in) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          mk |= (need_k_rank && kk[j] == kth_key) ? (1u << j) : 0u;
          mp |= (need_p_rank && kk[j] == p_key) ? (1u << j) : 0u;
        }
      }
      unsigned ex_k = 0, ex_p = 0;      // ties in lower lanes of this step
      if (__ballot(mk != 0 || mp != 0)) {   // (wave-uniform: most steps hold no boundary tie at all)
        unsigned ik = __popc(mk), ip = __popc(mp);
        const unsigned ck0 = ik, cp0 = ip;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned tk = __shfl_up(ik, o), tp = __shfl_up(ip, o);
          if (lane >= o) { ik += tk; ip += tp; }
        }
        ex_k = ik - ck0;
        ex_p = ip - cp0;
        const unsigned tot_k = __shfl(ik, 63), tot_p = __shfl(ip, 63);
        ex_k += r_k;
        ex_p += r_p;
        r_k += tot_k;
        r_p += tot_p;
      } else {
        ex_k = r_k;
        ex_p = r_p;
      }
      if (!in) continue;
      T o[VEC];
      bool dirty = scale;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        bool keep = true;
        if (use_k) keep = kk[j] > kth_key || (kk[j] == kth_key && ex_k + __popc(mk & ((1u << j) - 1u)) < k_rem);
        if (keep && top_p) keep = kk[j] > p_key || (kk[j] == p_key && ex_p + __popc(mp & ((1u << j) - 1u)) < p_keep);
        o[j] = from_f32<T>(keep ? x[j] : ninf);
        dirty = dirty || !keep;
      }
      if (dirty) {   // (an untouched vector is not rewritten)
        uint4 raw;
        __builtin_memcpy(&raw, o, 16);
        *reinterpret_cast<uint4*>(row + i0) = raw;
      }
    }
    return;
  }
  const int seg = ((V + kLpWaves - 1) / kLpWaves + 63) / 64 * 64;
  const int s0 = wave * seg, s1 = s0 + seg < V ? s0 + seg : V;
  unsigned c_k = 0, c_p = 0;
  for (int base = s0; base < s1; base += 64) {
    const int i = base + lane;
    const uint32_t kk = i < s1 ? lp_key(value(i)) : 0u;
    c_k += __popcll(__ballot(i < s1 && need_k_rank && kk == kth_key));
    c_p += __popcll(__ballot(i < s1 && need_p_rank && kk == p_key));
  }
  if (lane == 0) { sh.wave_ties[0][wave] = c_k; sh.wave_ties[1][wave] = c_p; }
  __syncthreads();
  unsigned r_k = 0, r_p = 0;
  for (int w = 0; w < wave; ++w) { r_k += sh.wave_ties[0][w]; r_p += sh.wave_ties[1][w]; }
  // ---- final pass: temperature-scaled value or -inf
  for (int base = s0; base < s1; base += 64) {
    const int i = base + lane;
    const bool in = i < s1;
    const float x = in ? value(i) : 0.0f;
    const uint32_t kk = in ? lp_key(x) : 0u;
    const bool tie_k = in && need_k_rank && kk == kth_key, tie_p = in && need_p_rank && kk == p_key;
    const unsigned long long mk = __ballot(tie_k), mp = __ballot(tie_p);
    const unsigned long long below = (1ull << lane) - 1ull;
    const unsigned my_k = r_k + __popcll(mk & below), my_p = r_p + __popcll(mp & below);
    r_k += __popcll(mk);
    r_p += __popcll(mp);
    if (!in) continue;
    bool keep = true;
    if (use_k) keep = kk > kth_key || (kk == kth_key && my_k < k_rem);
    if (keep && top_p) {
      // a tie at p_key that is ALSO a top-k boundary tie ranks among the top-k survivors, which are the first k_rem by index:
      // the same index order, so its rank is unchanged
      keep = kk > p_key || (kk == p_key && my_p < p_keep);
    }
    if (!keep) row[i] = from_f32<T>(ninf);
    else if (scale) row[i] = from_f32<T>(x);
  }
}

}  // namespace xm

using namespace xm;

#define XM_DISPATCH_LOGITS(DT, T, ...)                        \
  switch (DT) {                                              \
    case XM_F32: { using T = float; __VA_ARGS__; break; }    \
    case XM_BF16: { using T = bf16_t; __VA_ARGS__; break; }  \
    case XM_F16: { using T = f16_t; __VA_ARGS__; break; }    \
    default: return XM_ERR_UNSUPPORTED;                      \
  }

static int lp_check_launch() { return hipGetLastError() == hipSuccess ? XM_OK : XM_ERR_HIP; }

extern "C" {

size_t xllm_mi355_apply_penalties_workspace_bytes(int64_t batch, int64_t n_unique) {
  return batch > 0 && n_unique > 0 ? (size_t)batch * (size_t)n_unique * sizeof(float) : 0;
}

int xllm_mi355_apply_penalties(void* logits, int64_t batch, int64_t vocab, int64_t row_stride, int dtype,
                               const int64_t* unique_token_ids, const int32_t* unique_token_counts, int64_t n_unique,
                               const float* frequency_penalties, const float* presence_penalties,
                               const float* repetition_penalties, void* workspace, size_t workspace_bytes, void* stream) {
  if (!logits || batch < 0 || vocab <= 0 || row_stride < vocab || n_unique < 0) return XM_ERR_INVALID;
  if ((frequency_penalties != nullptr) != (presence_penalties != nullptr)) return XM_ERR_INVALID;   // sampler.cpp:35-41: one pair
  if (!frequency_penalties && !repetition_penalties) return XM_OK;
  if (batch == 0 || n_unique == 0) return XM_OK;
  if (!unique_token_ids || (frequency_penalties && !unique_token_counts)) return XM_ERR_INVALID;
  if (!workspace || workspace_bytes < (size_t)batch * n_unique * sizeof(float)) return XM_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)((n_unique + 255) / 256), (unsigned)batch);
  float* scratch = reinterpret_cast<float*>(workspace);
  XM_DISPATCH_LOGITS(dtype, T, {
    hipLaunchKernelGGL((penalties_gather_kernel<T>), grid, dim3(256), 0, s, (const T*)logits, row_stride, vocab, unique_token_ids,
                       unique_token_counts, n_unique, frequency_penalties, presence_penalties, repetition_penalties, scratch);
    hipLaunchKernelGGL((penalties_scatter_kernel<T>), grid, dim3(256), 0, s, (T*)logits, row_stride, vocab, unique_token_ids,
                       n_unique, scratch);
  });
  return lp_check_launch();
}

int xllm_mi355_apply_temperatures(void* logits, int64_t batch, int64_t vocab, int64_t row_stride, int dtype,
                                  const float* temperatures, void* stream) {
  if (!logits || !temperatures || batch < 0 || vocab <= 0 || row_stride < vocab) return XM_ERR_INVALID;
  if (batch == 0) return XM_OK;
  int64_t bx = (vocab + 255) / 256;
  bx = bx > 64 ? 64 : bx;
  XM_DISPATCH_LOGITS(dtype, T, {
    hipLaunchKernelGGL((temperatures_kernel<T>), dim3((unsigned)bx, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, (T*)logits,
                       row_stride, vocab, temperatures);
  });
  return lp_check_launch();
}

int xllm_mi355_apply_top_k_top_p(void* logits, int64_t batch, int64_t vocab, int64_t row_stride, int dtype,
                                 const float* temperatures, const int64_t* top_k, const float* top_p, void* stream) {
  if (!logits || batch < 0 || vocab <= 0 || vocab >= (1ll << 31) || row_stride < vocab) return XM_ERR_INVALID;
  if (batch == 0) return XM_OK;
  if (!top_k && !top_p) {   // logits_utils.cpp:96-101: temperatures only
    if (!temperatures) return XM_OK;
    return xllm_mi355_apply_temperatures(logits, batch, vocab, row_stride, dtype, temperatures, stream);
  }
  const int rule = (top_k && top_p) ? 1 : 0;
  XM_DISPATCH_LOGITS(dtype, T, {
    hipLaunchKernelGGL((top_k_top_p_kernel<T>), dim3((unsigned)batch), dim3(kLpThreads), 0, (hipStream_t)stream, (T*)logits,
                       row_stride, (int)vocab, temperatures, top_k, top_p, rule);
  });
  return lp_check_launch();
}

}  // extern "C"
