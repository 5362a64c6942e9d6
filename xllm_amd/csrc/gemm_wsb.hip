// gemm_wsb.hip -- weight-stream GEMM for 16-bit weights at decode shapes (M <= 64 rows per problem), dense and grouped.
//
// Operators served: kernel::matmul (ops_api.h:48 -> kernels/dcu/matmul.cpp:20-25, F::linear) for the bf16 / f16 linears of a
// decode step (cfg2: Qwen2-7B bf16, B = 64), and kernel::group_gemm (ops_api.h:77 -> kernels/dcu/group_gemm.cpp:25-74) when
// the experts hold a handful of rows each (MoE decode: 128 tokens x top-8 over 256 experts = 4 rows per expert).
//
// At these shapes the problem is ONE pass over the weights: out[m][n] = sum_k x[m][k] w[n][k] with every byte of w used
// once. So there is no LDS stage and no barrier in the K loop -- the structure of the paged-decode attention kernel:
//   * workgroup = 64 output columns x all rows x one K slice; its 4 waves split the K steps of the slice round-robin, each
//     wave owns ALL 64 columns and ALL rows for its steps (no operand is loaded twice inside a workgroup);
//   * w goes HBM -> VGPR directly in MFMA-operand order: w is the MFMA *row* operand (D[n][m]), lane (n = l & 15,
//     kq = l >> 4) loads the 32 contiguous bytes w[n][64 s + 16 kq .. +16) of K step s -- four lanes cover one 128-byte line
//     of a row -- and feeds two v_mfma_f32_16x16x32 (k permutation identical on both operands); x (<= 64 rows, L2-resident)
//     is loaded the same way as the column operand;
//   * one K step (16 KiB of operands per wave at 64 rows) is in flight in registers underneath the previous step's 32 MFMAs;
//   * the four waves' fp32 partial tiles meet in LDS once, at the end, and are summed in wave order; K slices (few-column
//     problems: qkv / o / down) leave fp32 slabs that a second kernel adds in slice order -- deterministic, no atomics.
// A lane of the accumulator holds 4 consecutive n of one row, so partial tiles move as 16-byte LDS stores and the 16-bit
// result leaves as 8-byte stores of whole 128-byte row segments.
//
// Grouped form: blockIdx.y = expert; the workgroup finds its expert's rows from the device-side sizes (prefix sum over
// <= 1024 experts in registers), leaves at once when the expert has no row (its weights are never read), and walks the
// expert's rows 16 at a time; the expand of the reference (index_select(hidden, dst_src / topk)) is an index on the x rows.
#include <stdlib.h>

#include "common.h"

namespace xm {

typedef __bf16 wsb_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wsb_f16x8 __attribute__((ext_vector_type(8)));
typedef float wsb_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned wsb_u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct WsbTraits;
template <>
struct WsbTraits<bf16_t> {
  static __device__ __forceinline__ wsb_f32x4 mfma(wsb_u32x4 a, wsb_u32x4 b, wsb_f32x4 c) {
    wsb_bf16x8 av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
  }
};
template <>
struct WsbTraits<f16_t> {
  static __device__ __forceinline__ wsb_f32x4 mfma(wsb_u32x4 a, wsb_u32x4 b, wsb_f32x4 c) {
    wsb_f16x8 av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
  }
};

constexpr int kWsbCols = 64;        // output columns per workgroup = 4 MFMA column blocks
constexpr int kWsbK = 64;           // k elements per step = two MFMAs
constexpr int kWsbPad = 64;         // floats per LDS row of a partial tile; the 16 float4 of a row are XOR-swizzled by the row
                                    // (lanes of one store hold 16 different rows: without it a 16-way bank conflict)

struct WsbGroup {
  const int32_t* counts;      // rows per expert (device), null = dense
  const int32_t* row_index;   // sorted row r reads x row row_index[r] / index_div (null: x row r)
  int n_experts, index_div;
};

// hipcc moves plain loads of a software pipeline next to their use (below a mid-loop exit, or behind the compute of a
// counted loop: both seen in the ISA of the first two versions of this kernel), so the operand loads are inline asm, retired by
// counted s_waitcnt that carry the stage's registers as operands (the MFMAs cannot be scheduled above them)
// (scalar base + 32-bit lane offset: the K walk lives in SGPRs, one VGPR per row pointer)
#define WSB_LD(DST, VOFF, SBASE, OFF) \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFF : "=v"(DST) : "v"(VOFF), "s"(SBASE))
// Non-temporal weight loads were TRIED here in round 4 (-DWSB_W_NT=1) and LOSE by 15-20 % (bf16 layer at M = 16: 12.0 / 10.6 / 59.0 /
// 30.5 -> 14.0 / 12.3 / 71.5 / 36.4 us, cfg4-slice 1.02 -> 1.19 ms; profiles/r04_wsb_nt.txt): a lane reads 16 bytes of a row-major
// weight row per instruction, so a 128-byte line is consumed by EIGHT consecutive K steps -- it has to stay cached in between.
// (nt pays where one instruction consumes whole lines: the KV stream of attention_decode.hip, the packed fragments of gemm_ws.hip.)
#ifndef WSB_W_NT
#define WSB_W_NT 0
#endif
#if WSB_W_NT
#define WSB_LDW(DST, VOFF, SBASE, OFF) \
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFF " nt" : "=v"(DST) : "v"(VOFF), "s"(SBASE))
#else
#define WSB_LDW(DST, VOFF, SBASE, OFF) WSB_LD(DST, VOFF, SBASE, OFF)
#endif

__device__ __forceinline__ void wsb_issue_w(wsb_u32x4 (&wr)[4][2], const uint32_t (&woff)[4], const char* base) {
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    WSB_LDW(wr[nb][0], woff[nb], base, 0);
    WSB_LDW(wr[nb][1], woff[nb], base, 16);
  }
}
template <int MB>
__device__ __forceinline__ void wsb_issue_x(wsb_u32x4 (&xr)[MB][2], const uint32_t (&xoff)[MB], const char* base) {
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    WSB_LD(xr[mb][0], xoff[mb], base, 0);
    WSB_LD(xr[mb][1], xoff[mb], base, 16);
  }
}

// wait until at most CNT vector loads are outstanding; the registers about to be consumed are in/out operands
template <int MB, int CNT>
__device__ __forceinline__ void wsb_wait(wsb_u32x4 (&wr)[4][2], wsb_u32x4 (&xr)[MB][2]) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(wr[0][0]), "+v"(wr[0][1]), "+v"(wr[1][0]), "+v"(wr[1][1]), "+v"(wr[2][0]), "+v"(wr[2][1]),
                 "+v"(wr[3][0]), "+v"(wr[3][1])
               : "n"(CNT));
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) asm volatile("" : "+v"(xr[mb][0]), "+v"(xr[mb][1]));
}

// WAVE_COLS (round 6, short K: the w2 of a MoE decode step, K = moe_intermediate / tp = 256): the four waves do not split the K
// steps (one step each, no pipeline, a 32-KiB weight block per workgroup behind a per-workgroup prologue) -- every wave owns its OWN
// 64 columns over all K steps, a workgroup covers 256 columns, no cross-wave sum.
template <typename T, int MB, bool GROUPED, bool WAVE_COLS = false>
__global__ __launch_bounds__(256, 2) void gemm_wsb_kernel(const T* __restrict__ x, const T* __restrict__ w,
                                                       T* __restrict__ out, float* __restrict__ slabs,
                                                       const T* __restrict__ bias, int M, int N, int K, int steps_per_slice,
                                                       WsbGroup grp) {
  using TR = WsbTraits<T>;
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][16 MB rows][kWsbPad]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p16 = lane & 15, kq = lane >> 4;
  const int n0 = WAVE_COLS ? (blockIdx.x * 4 + wave) * kWsbCols : blockIdx.x * kWsbCols;
  const int n_steps = K / kWsbK;

  int row0 = 0, cnt = M, slice = 0;
  const T* wb = w;
  if constexpr (GROUPED) {
    const int e = blockIdx.y;
    int before = 0;
    for (int i = lane; i < e; i += 64) before += grp.counts[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    row0 = __builtin_amdgcn_readfirstlane(before);
    cnt = grp.counts[e];
    if (cnt <= 0) return;                       // the expert's weights are never touched
    wb = w + (int64_t)e * N * K;
  } else {
    slice = blockIdx.y;
  }
  const int s_lo = slice * steps_per_slice;
  int s_hi = s_lo + steps_per_slice;
  s_hi = s_hi < n_steps ? s_hi : n_steps;
  // this wave's steps: s_lo + wave, + 4, ... (a wave without a step adds a zero tile); WAVE_COLS: all of them
  constexpr int kStride = WAVE_COLS ? 1 : 4;
  const int my_first = WAVE_COLS ? s_lo : s_lo + wave;
  const int my_n = my_first < s_hi ? (s_hi - my_first + kStride - 1) / kStride : 0;

  // per-lane byte offsets of the 4 column blocks' rows inside the workgroup's 64 rows of w (< 2^31: checked on the host)
  const char* const wbase = reinterpret_cast<const char*>(wb + (int64_t)n0 * K);
  uint32_t woff[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) woff[nb] = (uint32_t)(((nb * 16 + p16) * K + kq * 16) * 2);

  for (int rb0 = 0; rb0 < cnt; rb0 += 16 * MB) {
    uint32_t xoff[MB];                         // byte offsets of this lane's x rows (all of x < 2^31 bytes: host check)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      int r = rb0 + mb * 16 + p16;
      r = r < cnt ? r : cnt - 1;               // rows past the end re-read the last row; their results are not stored
      int64_t src = row0 + r;
      if constexpr (GROUPED) {
        if (grp.row_index) src = grp.row_index[src] / grp.index_div;
      }
      xoff[mb] = (uint32_t)((src * (int64_t)K + kq * 16) * 2);
    }
    const char* const xbase = reinterpret_cast<const char*>(x);
    wsb_f32x4 acc[4][MB];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = wsb_f32x4{0.f, 0.f, 0.f, 0.f};

    // register pipeline: the weights (HBM) run TWO steps ahead in a ring of three stages, the activations (L2) one step ahead
    // in a ring of two. The vector-memory counter retires in order, so the issue order inside a step is x(i+1) THEN w(i+2):
    // waiting for x(i) then leaves w(i+1), x(i+1), w(i+2) in flight (16 + 2 MB loads) -- two weight stages per wave, twice
    // what a plain double buffer keeps outstanding (measured at M = 64: 3.0 TB/s with the double buffer). Loads are
    // unconditional: a step past the wave's range re-loads its last step.
    wsb_u32x4 wr0[4][2], wr1[4][2], wr2[4][2], xr0[MB][2], xr1[MB][2];
    auto koff_of = [&](int i) -> int64_t {     // byte offset of the wave's step i along K (wave-uniform: SGPRs)
      i = i < my_n ? i : my_n - 1;
      return (int64_t)(my_first + kStride * i) * (kWsbK * 2);
    };
#define WSB_STEP(I_, WC_, WN_, XC_, XN_)                                                        \
  {                                                                                              \
    wsb_issue_x<MB>(XN_, xoff, xbase + koff_of((I_) + 1));                                       \
    wsb_issue_w(WN_, woff, wbase + koff_of((I_) + 2));                                           \
    wsb_wait<MB, 16 + 2 * MB>(WC_, XC_);                                                         \
    _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                \
    _Pragma("unroll") for (int nb = 0; nb < 4; ++nb)                                             \
    _Pragma("unroll") for (int mb = 0; mb < MB; ++mb)                                            \
        acc[nb][mb] = TR::mfma(WC_[nb][h], XC_[mb][h], acc[nb][mb]);                             \
  }
    if constexpr (MB == 1) {
      // 16 rows: a step is 8 KiB of weights against 8 MFMAs and the wave needs few registers -- four workgroups per CU keep
      // enough in flight with a plain double buffer (the three-stage form makes hipcc spill here)
      if (my_n > 0) {
        wsb_issue_w(wr0, woff, wbase + koff_of(0));
        wsb_issue_x<MB>(xr0, xoff, xbase + koff_of(0));
#pragma nounroll
        for (int i = 0; i < my_n; i += 2) {
          wsb_issue_w(wr1, woff, wbase + koff_of(i + 1));
          wsb_issue_x<MB>(xr1, xoff, xbase + koff_of(i + 1));
          wsb_wait<MB, 8 + 2 * MB>(wr0, xr0);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb][0] = TR::mfma(wr0[nb][h], xr0[0][h], acc[nb][0]);
          if (i + 1 >= my_n) break;
          wsb_issue_w(wr0, woff, wbase + koff_of(i + 2));
          wsb_issue_x<MB>(xr0, xoff, xbase + koff_of(i + 2));
          wsb_wait<MB, 8 + 2 * MB>(wr1, xr1);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) acc[nb][0] = TR::mfma(wr1[nb][h], xr1[0][h], acc[nb][0]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else if (my_n > 0) {
      wsb_issue_w(wr0, woff, wbase + koff_of(0));
      wsb_issue_x<MB>(xr0, xoff, xbase + koff_of(0));
      wsb_issue_w(wr1, woff, wbase + koff_of(1));
#pragma nounroll
      for (int i = 0; i < my_n; i += 6) {
        WSB_STEP(i, wr0, wr2, xr0, xr1)
        if (i + 1 >= my_n) break;
        WSB_STEP(i + 1, wr1, wr0, xr1, xr0)
        if (i + 2 >= my_n) break;
        WSB_STEP(i + 2, wr2, wr1, xr0, xr1)
        if (i + 3 >= my_n) break;
        WSB_STEP(i + 3, wr0, wr2, xr1, xr0)
        if (i + 4 >= my_n) break;
        WSB_STEP(i + 4, wr1, wr0, xr0, xr1)
        if (i + 5 >= my_n) break;
        WSB_STEP(i + 5, wr2, wr1, xr1, xr0)
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing (redundant) prefetches
    }
#undef WSB_STEP

    // partial tiles -> LDS: lane holds D[n = nb*16 + 4 kq + r][m = mb*16 + p16]: 4 consecutive n of one row
    float* mine = red + wave * (16 * MB * kWsbPad);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
        *reinterpret_cast<wsb_f32x4*>(mine + (mb * 16 + p16) * kWsbPad + (((nb * 4 + kq) ^ p16) << 2)) = acc[nb][mb];
    if constexpr (!WAVE_COLS) __syncthreads();   // (WAVE_COLS: a wave re-reads only its own tile)
    int rows = cnt - rb0;
    rows = rows < 16 * MB ? rows : 16 * MB;
    for (int idx = WAVE_COLS ? lane : (int)threadIdx.x; idx < rows * 16; idx += WAVE_COLS ? 64 : 256) {
      const int m = idx >> 4, c4 = (idx & 15) * 4;
      const int sw = ((idx & 15) ^ (m & 15)) << 2;
      wsb_f32x4 v = *reinterpret_cast<const wsb_f32x4*>((WAVE_COLS ? mine : red) + m * kWsbPad + sw);
      if constexpr (!WAVE_COLS) {
#pragma unroll
        for (int wv = 1; wv < 4; ++wv) {
          const wsb_f32x4 t = *reinterpret_cast<const wsb_f32x4*>(red + wv * (16 * MB * kWsbPad) + m * kWsbPad + sw);
          v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
        }
      }
      const int64_t o = (int64_t)(row0 + rb0 + m) * N + n0 + c4;
      if (slabs) {
        *reinterpret_cast<wsb_f32x4*>(slabs + (int64_t)slice * M * N + o) = v;
      } else {
        uint16_t hv[4];
        T bv[4];
        if (bias) {
          const uint2 braw = *reinterpret_cast<const uint2*>(bias + n0 + c4);   // 8-byte aligned (checked on the host)
          __builtin_memcpy(bv, &braw, 8);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float b = bias ? to_f32<T>(bv[j]) : 0.0f;
          const T t = from_f32<T>(v[j] + b);
          __builtin_memcpy(&hv[j], &t, 2);
        }
        *reinterpret_cast<uint2*>(out + o) =
            make_uint2((uint32_t)hv[0] | ((uint32_t)hv[1] << 16), (uint32_t)hv[2] | ((uint32_t)hv[3] << 16));
      }
    }
    if constexpr (!WAVE_COLS) __syncthreads();   // the next row pass reuses the LDS tiles
  }
}

// K slices: out = sum of the fp32 slabs IN SLICE ORDER (+ bias); the slabs are re-zeroed (the registered GEMM scratch is
// zero at rest: the int8 split-K path relies on it)
template <typename T>
__global__ __launch_bounds__(256) void wsb_reduce_kernel(float* __restrict__ slabs, T* __restrict__ out,
                                                         const T* __restrict__ bias, int64_t M, int64_t N, int slices) {
  const int64_t total = M * N;
  for (int64_t idx = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; idx < total; idx += (int64_t)gridDim.x * 1024) {
    wsb_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int sl = 0; sl < slices; ++sl) {
      wsb_f32x4* p = reinterpret_cast<wsb_f32x4*>(slabs + (int64_t)sl * total + idx);
      const wsb_f32x4 v = *p;
      *p = wsb_f32x4{0.f, 0.f, 0.f, 0.f};
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    const int64_t n = idx % N;
    uint16_t hv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float b = bias ? to_f32<T>(bias[n + j]) : 0.0f;
      const T t = from_f32<T>(acc[j] + b);
      __builtin_memcpy(&hv[j], &t, 2);
    }
    *reinterpret_cast<uint2*>(out + idx) =
        make_uint2((uint32_t)hv[0] | ((uint32_t)hv[1] << 16), (uint32_t)hv[2] | ((uint32_t)hv[3] << 16));
  }
}

// XLLM_MI355_WSB (product switch, read once): 0 = never (the tiled kernels take the decode shapes: parity test of that fallback),
// 1 = default policy, 2 = up to 64 rows. XLLM_MI355_WSB_SLICES (forced K-slice count of the dense form) is a tuning override.
static int g_wsb_mode = -2;
XM_TUNE_VAR(g_wsb_slices, "XLLM_MI355_WSB_SLICES", -1);
static void wsb_env() {
  if (g_wsb_mode == -2) g_wsb_mode = xm_switch("XLLM_MI355_WSB", 1);
}

template <typename T, int MB, bool GROUPED, bool WAVE_COLS = false>
static void wsb_launch(dim3 grid, hipStream_t s, const T* x, const T* w, T* out, float* slabs, const T* bias, int M, int N,
                       int K, int per, WsbGroup g) {
  const size_t lds = (size_t)4 * 16 * MB * kWsbPad * sizeof(float);   // 16 / 32 / 64 KiB
  hipLaunchKernelGGL((gemm_wsb_kernel<T, MB, GROUPED, WAVE_COLS>), grid, dim3(256), lds, s, x, w, out, slabs, bias, M, N, K, per, g);
}

// dense: returns XM_ERR_UNSUPPORTED when the shape is not this kernel's (the caller keeps its tiled kernels)
template <typename T>
int launch_gemm_wsb_dense(const void* x, const void* w, const void* bias, void* out, int64_t M, int64_t N, int64_t K,
                          void* workspace, size_t ws_bytes, hipStream_t s) {
  wsb_env();
  // measured (profiles/r02_gemm_wsb.txt, Qwen2-7B bf16 layer, us): M = 16: 118.7 against 166.3 for the tiled kernels, M = 32:
  // 139.8 against 174.4, M = 64: 180.9 against 177.3 (gate_up 95 against 90: both stop at ~3 TB/s there) -- so 33 .. 64 rows stay
  // on the tiled kernels unless XLLM_MI355_WSB=2 asks for this one (A/B, tests)
  if (g_wsb_mode != 2 && M > 32) return XM_ERR_UNSUPPORTED;
  if (!g_wsb_mode || M <= 0 || M > 64 || N % kWsbCols || K % kWsbK || K / kWsbK < 4 || N * K >= (1ll << 40) ||
      64 * K * 2 >= (1ll << 31) ||
      ((uintptr_t)x % 16) || ((uintptr_t)w % 16) || ((uintptr_t)out % 8) || ((uintptr_t)bias % 8))
    return XM_ERR_UNSUPPORTED;
  const int n_tiles = (int)(N / kWsbCols), n_steps = (int)(K / kWsbK);
  // K slices: as many as keep the grid AT OR BELOW one workgroup per CU (measured at M = 64, profiles/r02_gemm_wsb.txt:
  // down_proj 56 tiles x 4 slices = 224 workgroups 48.8 us, x 5 = 280 workgroups 68.4 us -- a second workgroup on some CUs
  // costs more than the idle CUs do), every wave of a slice keeps >= 2 steps, slabs fit
  int slices = 1;
  while (n_tiles * (slices + 1) <= 256 && slices < 8 && n_steps / (slices + 1) >= 8 &&
         (size_t)(slices + 1) * M * N * 4 <= ws_bytes)
    ++slices;
  if (g_wsb_slices > 0 && (size_t)g_wsb_slices * M * N * 4 <= ws_bytes && n_steps / g_wsb_slices >= 1) slices = g_wsb_slices;
  if (!workspace) slices = 1;
  const int per = (n_steps + slices - 1) / slices;
  slices = (n_steps + per - 1) / per;
  float* slabs = slices > 1 ? reinterpret_cast<float*>(workspace) : nullptr;
  const dim3 grid((unsigned)n_tiles, (unsigned)slices);
  const WsbGroup g{nullptr, nullptr, 0, 1};
  if (M <= 16) wsb_launch<T, 1, false>(grid, s, (const T*)x, (const T*)w, (T*)out, slabs, (const T*)bias, (int)M, (int)N, (int)K, per, g);
  else if (M <= 32) wsb_launch<T, 2, false>(grid, s, (const T*)x, (const T*)w, (T*)out, slabs, (const T*)bias, (int)M, (int)N, (int)K, per, g);
  else wsb_launch<T, 4, false>(grid, s, (const T*)x, (const T*)w, (T*)out, slabs, (const T*)bias, (int)M, (int)N, (int)K, per, g);
  if (slices > 1) {
    int64_t blocks = (M * N / 4 + 255) / 256;
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL((wsb_reduce_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, slabs, (T*)out, (const T*)bias, M, N,
                       slices);
  }
  return hip_check_launch();
}

// grouped: experts with a handful of rows each (max_rows <= 16 * n_experts on average); any count is handled (16 rows per pass)
template <typename T>
int launch_gemm_wsb_grouped(const void* x, const void* w, const int32_t* counts, void* out, int64_t max_rows,
                            int64_t n_experts, int64_t N, int64_t K, const int32_t* row_index, int64_t index_div,
                            hipStream_t s) {
  wsb_env();
  if (!g_wsb_mode || n_experts > 65535 || N % kWsbCols || K % kWsbK || max_rows > 16 * n_experts ||
      64 * K * 2 >= (1ll << 31) || max_rows * K * 2 >= (1ll << 31) ||
      ((uintptr_t)x % 16) || ((uintptr_t)w % 16) || ((uintptr_t)out % 8))
    return XM_ERR_UNSUPPORTED;
  const WsbGroup g{counts, row_index, (int)n_experts, (int)(index_div > 0 ? index_div : 1)};
  // Plan (round 6, tools/group_gemm_decode_ab.sh, profiles/r06_group_gemm_decode.txt; 256 experts, 1024 rows, uniform / skewed):
  //   * rows per pass stay 16: 32 (experts with 17-32 rows read their weights once instead of twice) loses on every shape but the
  //     longest K under skewed routing (250 vs 261 us) -- 327 -> 344, 287 -> 432 us with 4 rows per expert (tuning arm only);
  //   * K <= 512 (w2 of a decode step): every wave owns its own 64 columns over all K steps (WAVE_COLS): 287 -> 195 us at
  //     [256, 7168, 256], 123 -> 103 us at K = 512; longer K keeps the K split over the waves (K = 2048: 111 vs 130 us).
  XM_TUNE_VAR(grp_mb, "XLLM_MI355_WSB_GROUP_MB", 0);
  const bool mb2 = grp_mb == 2;
  XM_TUNE_VAR(wave_cols, "XLLM_MI355_WSB_WAVE_COLS", -1);
  const bool wc = (wave_cols < 0 ? K / kWsbK <= 8 : wave_cols != 0) && N % (4 * kWsbCols) == 0;
  const dim3 grid((unsigned)(N / (wc ? 4 * kWsbCols : kWsbCols)), (unsigned)n_experts);
#define XM_WSB_G(MB_, WC_)                                                                                             \
  wsb_launch<T, MB_, true, WC_>(grid, s, (const T*)x, (const T*)w, (T*)out, nullptr, nullptr, (int)max_rows, (int)N,  \
                                (int)K, (int)(K / kWsbK), g)
  if (wc) { if (mb2) XM_WSB_G(2, true); else XM_WSB_G(1, true); }
  else { if (mb2) XM_WSB_G(2, false); else XM_WSB_G(1, false); }
#undef XM_WSB_G
  return hip_check_launch();
}

template int launch_gemm_wsb_dense<bf16_t>(const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, void*,
                                           size_t, hipStream_t);
template int launch_gemm_wsb_dense<f16_t>(const void*, const void*, const void*, void*, int64_t, int64_t, int64_t, void*,
                                          size_t, hipStream_t);
template int launch_gemm_wsb_grouped<bf16_t>(const void*, const void*, const int32_t*, void*, int64_t, int64_t, int64_t,
                                             int64_t, const int32_t*, int64_t, hipStream_t);
template int launch_gemm_wsb_grouped<f16_t>(const void*, const void*, const int32_t*, void*, int64_t, int64_t, int64_t,
                                            int64_t, const int32_t*, int64_t, hipStream_t);

// ------------------------------------------------------------------------------------------------ per-head batched GEMM
// out[t, h, n] = r16( sum_k x[t, h, k] * w[h, n, k] ): the two weight-absorption products of MLA attention,
//   q_nope . W_kc  (DeepseekV2AttentionImpl::forward, layers/dcu/deepseek_v2_attention.cpp:310-311: K = qk_nope 128, N = kv_lora 512)
//   attn   . W_vc  (project_output, :180-187:                                              K = kv_lora 512, N = v_head 128)
// which the reference runs as torch::bmm over transposed views (rocBLAS) plus two transposes. Here the token-major tensors are
// read and written in place through their (token, head) strides -- no transpose, no copy -- and the weights are taken K-contiguous
// per output column ([h, N, K]: for W_vc that IS kv_b_proj's weight slice before the reference transposes it for bmm, :336-338;
// W_kc is transposed once at load time). A few hundred MFLOP and a few MB per call: the structure of the weight-stream kernel above
// without K slices or an LDS stage -- wave = 16 tokens x 64 columns of one head, both operands straight from L2 into MFMA operand
// order (lane (r = l & 15, kq = l >> 4) loads the 16 bytes [32 s + 8 kq, +8) of row r), W as the MFMA row operand so that a
// lane of the accumulator holds 4 consecutive n of one token and the result leaves as 8-byte stores. fp32 accumulation, k ascending.
template <typename T>
__global__ __launch_bounds__(256) void bmm_heads_kernel(const T* __restrict__ x, int64_t x_st, int64_t x_sh,
                                                        const T* __restrict__ w, int64_t w_sh, int64_t w_sn,
                                                        T* __restrict__ out, int64_t o_st, int64_t o_sh, int Tn, int N, int K) {
  const int h = blockIdx.z, n0 = blockIdx.y * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, kq = lane >> 4;
  const int t = blockIdx.x * 64 + wave * 16 + r;
  const bool t_ok = t < Tn;
  const T* const xrow = x + (int64_t)(t_ok ? t : 0) * x_st + (int64_t)h * x_sh + kq * 8;
  const T* wrow[4];
  bool n_ok[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int n = n0 + nb * 16 + r;
    n_ok[nb] = n < N;
    wrow[nb] = w + (int64_t)h * w_sh + (int64_t)(n_ok[nb] ? n : 0) * w_sn + kq * 8;
  }
  wsb_f32x4 acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) acc[nb] = wsb_f32x4{0.f, 0.f, 0.f, 0.f};
  const wsb_u32x4 zero = {0u, 0u, 0u, 0u};
  for (int k = 0; k < K; k += 32) {
    const wsb_u32x4 xv = t_ok ? *reinterpret_cast<const wsb_u32x4*>(xrow + k) : zero;
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const wsb_u32x4 wv = n_ok[nb] ? *reinterpret_cast<const wsb_u32x4*>(wrow[nb] + k) : zero;
      acc[nb] = WsbTraits<T>::mfma(wv, xv, acc[nb]);     // D[n][m]: lane & 15 = token, registers = 4 consecutive n
    }
  }
  if (!t_ok) return;
  T* const orow = out + (int64_t)t * o_st + (int64_t)h * o_sh;
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) {
    const int n = n0 + nb * 16 + 4 * kq;
    if (n + 3 < N) {
      uint16_t hb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const T v16 = from_f32<T>(acc[nb][e]);     // RNE, like every 16-bit store of this file
        __builtin_memcpy(&hb[e], &v16, 2);
      }
      *reinterpret_cast<uint2*>(orow + n) = make_uint2((unsigned)hb[0] | ((unsigned)hb[1] << 16), (unsigned)hb[2] | ((unsigned)hb[3] << 16));
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < N) orow[n + e] = from_f32<T>(acc[nb][e]);
    }
  }
}

template <typename T>
int launch_bmm_heads(const void* x, int64_t x_st, int64_t x_sh, const void* w, int64_t w_sh, int64_t w_sn, void* out, int64_t o_st,
                     int64_t o_sh, int64_t Tn, int64_t H, int64_t N, int64_t K, hipStream_t s) {
  const dim3 grid((unsigned)((Tn + 63) / 64), (unsigned)((N + 63) / 64), (unsigned)H);
  hipLaunchKernelGGL((bmm_heads_kernel<T>), grid, dim3(256), 0, s, (const T*)x, x_st, x_sh, (const T*)w, w_sh, w_sn, (T*)out, o_st,
                     o_sh, (int)Tn, (int)N, (int)K);
  return hip_check_launch();
}

}  // namespace xm

extern "C" {
int xllm_mi355_bmm_heads(const void* x, int64_t x_stride_t, int64_t x_stride_h, const void* w, int64_t w_stride_h,
                         int64_t w_stride_n, void* out, int64_t out_stride_t, int64_t out_stride_h, int64_t n_tokens,
                         int64_t n_heads, int64_t N, int64_t K, int dtype, void* stream) {
  using namespace xm;
  if (!x || !w || !out || n_tokens < 0 || n_heads <= 0 || N <= 0 || K <= 0) return XM_ERR_INVALID;
  if (n_tokens == 0) return XM_OK;
  // 16-byte operand loads, 8-byte stores: every row of every operand starts on such a boundary
  if (K % 32 || x_stride_t % 8 || x_stride_h % 8 || w_stride_h % 8 || w_stride_n % 8 || out_stride_t % 4 || out_stride_h % 4 ||
      ((uintptr_t)x % 16) || ((uintptr_t)w % 16) || ((uintptr_t)out % 8) || n_heads > 65535 || (N + 63) / 64 > 65535 ||
      n_tokens >= (1ll << 31) || N >= (1ll << 31) || K >= (1ll << 31))
    return XM_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == XM_BF16)
    return launch_bmm_heads<bf16_t>(x, x_stride_t, x_stride_h, w, w_stride_h, w_stride_n, out, out_stride_t, out_stride_h, n_tokens,
                                    n_heads, N, K, s);
  if (dtype == XM_F16)
    return launch_bmm_heads<f16_t>(x, x_stride_t, x_stride_h, w, w_stride_h, w_stride_n, out, out_stride_t, out_stride_h, n_tokens,
                                   n_heads, N, K, s);
  return XM_ERR_UNSUPPORTED;
}
}
