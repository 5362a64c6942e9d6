// gemm.hip -- C[M,N] = A[M,K] * W[N,K]^T on the gfx950 matrix cores, with the dequant epilogues of the
// hot path fused in:
//   int8 W8A8  (reference: dcu::scaled_matmul, kernels/dcu/scaled_matmul.cpp:103-300)
//              out = r16( float(int32 acc) * a_scale[m] * w_scale[n] + bias[n] )   v_mfma_i32_32x32x32_i8
//   fp8 e4m3   (reference: cutlass_scaled_mm, kernels/cuda/cutlass_w8a8/scaled_mm_entry.cu:55-116)
//              out = r16( a_scale * (w_scale * acc_f32) + bias )                    v_mfma_f32_32x32x16_fp8_fp8
//   bf16/f16   (reference: dcu::matmul == F::linear, kernels/dcu/matmul.cpp:20-25)
//              out = r16( acc_f32 + bias )                                          v_mfma_f32_32x32x16_{bf16,f16}
// Both operands are K-contiguous, so every MFMA fragment is one 16-byte LDS read.
//
// Structure (DESIGN.md "quant GEMM"): 128x128 block tile, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA
// 32x32 tiles; K step = 128 BYTES of each operand row; global -> registers -> LDS with the next tile's
// global loads in flight during the MFMAs (issue-early / write-late), LDS rows XOR-swizzled on the
// 16-byte chunk index by (row>>1)&7 so ds_read_b128 fragment reads are bank-conflict free; one barrier
// per K step, two LDS buffers. Block ids are remapped so the M-tiles that share a W tile run on the
// same XCD (same L2) back to back: W is fetched from HBM once.
// int8 only: optional split-K with exact int32 atomics into a workspace + a tiny dequant epilogue kernel
// (integer adds commute, so the result stays bit-exact and deterministic).
#include <stdlib.h>

#include "gemm_types.h"

namespace xm {

constexpr int BM = 128, BN = 128, BKB = 128;  // block tile, K step in bytes

// A [M, Kb bytes per row], W [N, Kb]; grid.x = tiles (XCD-remapped), grid.z = split-K slices
template <int KIND, bool SPLITK, int KB, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_kernel(const uint8_t* __restrict__ A_, const uint8_t* __restrict__ W_,
                                                      int M_, int N, int64_t Kb, int m_tiles, int n_tiles,
                                                      int ksteps_per_split, GemmEpi epi) {
  // grouped mode rebases these per workgroup, so they are locals, not the (read-only) kernel arguments
  const uint8_t* A = A_;
  const uint8_t* W = W_;
  int M = M_;
  void* out_base = epi.out;
  using MT = MmaTraits<KIND>;
  using acc_t = typename MT::acc_t;
  // KB = K step in bytes (64 or 128); rows of KB bytes = CPR 16-byte chunks, swizzled by SW(row)
  constexpr int CPR = KB / 16;                      // chunks per row: 4 or 8
  constexpr int NST = BM * CPR / 256;               // DMA instructions per thread and operand: 2 or 4
  __shared__ __attribute__((aligned(16))) uint8_t lds[2][2][BM * KB];  // [buf][A/W][tile]
  auto SW = [](int row) { return CPR == 8 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // XCD-aware 2-D rasterisation. Block b runs on XCD b%8 and the j = b/8 -th slot of that XCD. Each XCD walks its
  // own sequence of 8x8-tile super-blocks (64 concurrently resident workgroups = 32 CUs x 2): the 8 m-tiles and
  // 8 n-tiles of a super-block are each re-used 8 times out of that XCD's L2 while the K loops advance roughly
  // in step, so both operands leave HBM / Infinity Cache once per super-block instead of once per tile
  // (round-1 finding: with n-major order the A operand was re-streamed once per n-tile: 8.7 GB per gate_up
  // launch at M = 8192, i.e. the kernel was HBM-bound at 1.1 POP/s).
  int mt, nt;
  {
    const int b = blockIdx.x;
    if (!epi.group_counts && !SPLITK) {
      const int xcd = b & 7, j = b >> 3;
      const int sb = j >> 6, within = j & 63;
      const int S = sb * 8 + xcd;                       // global super-block index
      // super-block = 2^lm x 2^(6-lm) tiles (8x8 when there are >= 8 m-tiles, else all m-tiles x more n-tiles)
      const int lm = m_tiles >= 8 ? 3 : (m_tiles >= 4 ? 2 : (m_tiles >= 2 ? 1 : 0));
      const int n_sb_m = (m_tiles + (1 << lm) - 1) >> lm;
      const int SM = S % n_sb_m, SN = S / n_sb_m;
      mt = (SM << lm) + (within & ((1 << lm) - 1));
      nt = (SN << (6 - lm)) + (within >> lm);
      if (mt >= m_tiles || nt >= n_tiles) return;       // padding of the rasterised grid
    } else {
      mt = b % m_tiles;
      nt = b / m_tiles;
    }
  }
  int m0 = mt * BM;
  const int n0 = nt * BN;
  if (epi.group_counts) {
    // grouped (MoE) mode, reference dcu::group_gemm (kernels/dcu/group_gemm.cpp:25-74): rows of A are sorted by
    // expert, expert e owns rows [off_e, off_e + count_e) and weight W[e]. The m-tile index walks the experts'
    // tiles in order; counts are read on the DEVICE (no host sync, graph-capturable, unlike group_gemm.cpp:45).
    // NOTE: keep the walk on the scalar unit (readfirstlane). With `tile` in a VGPR, hipcc (ROCm 7.2) emitted
    // v_cmp (-> VCC) followed by s_cselect (reads SCC) for the selects below: a stale-SCC miscompile.
    int tile = __builtin_amdgcn_readfirstlane(mt), e = 0, off = 0;
    for (; e < epi.n_groups; ++e) {
      const int c = epi.group_counts[e];
      const int t = (c + BM - 1) / BM;
      if (tile < t) { M = c; break; }
      tile -= t;
      off += c;
    }
    if (e >= epi.n_groups) return;  // surplus workgroup (grid is sized for the worst case)
    m0 = tile * BM;
    A += (int64_t)off * Kb;
    W += (int64_t)e * N * Kb;
    out_base = reinterpret_cast<uint8_t*>(epi.out) + (int64_t)off * N * 2;
  }
  const int total_ksteps = (int)((Kb + KB - 1) / KB);
  const int ks_begin = blockIdx.z * ksteps_per_split;
  int ks_end = ks_begin + ksteps_per_split;
  ks_end = ks_end > total_ksteps ? total_ksteps : ks_end;

  // staging: global -> LDS by LDS-DMA (global_load_lds_dwordx4): no VGPR round trip, no ds_write pass.
  // One wave-instruction moves 64 x 16 B = 8 rows x 128 B into a lane-linear 1-KiB LDS span, so the XOR swizzle
  // that keeps the ds_read_b128 fragment reads conflict-free is applied to the per-lane GLOBAL source address:
  // LDS slot (row, s) holds global chunk s ^ ((row >> 1) & 7) of that row (an involution, also used by the reads).
  // Each thread issues 4 pieces of A and 4 of W per K step (chunk c = tid + i*256: row c/8, slot c%8).
  const uint8_t* ga[NST];
  const uint8_t* gw[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int c = tid + i * 256, row = c / CPR, slot = c % CPR;
    const int col = slot ^ SW(row);
    int ar = m0 + row; ar = ar < M ? ar : M - 1;
    int wr = n0 + row; wr = wr < N ? wr : N - 1;
    ga[i] = A + (int64_t)ar * Kb + col * 16;
    gw[i] = W + (int64_t)wr * Kb + col * 16;
  }
  auto stage = [&](int ks, int buf) {
    const int64_t kb0 = (int64_t)ks * KB;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      // wave-uniform LDS base of this instruction's 1-KiB span: (wave + 4*i) * 1024
      uint8_t* la = &lds[buf][0][(wave + 4 * i) * 1024];
      uint8_t* lw = &lds[buf][1][(wave + 4 * i) * 1024];
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga[i] + kb0),
                                       (__attribute__((address_space(3))) void*)la, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gw[i] + kb0),
                                       (__attribute__((address_space(3))) void*)lw, 16, 0, 0);
    }
  };

  acc_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = MT::zero();

  if (ks_begin < ks_end) {
    stage(ks_begin, 0);
    __syncthreads();  // hipcc drains the LDS-DMA (vmcnt(0)) before the barrier
    int cur = 0;
    for (int ks = ks_begin; ks < ks_end; ++ks) {
      const bool more = ks + 1 < ks_end;
      if (more) stage(ks + 1, cur ^ 1);
      const uint8_t* la = lds[cur][0];
      const uint8_t* lw = lds[cur][1];
      // all 16 fragment reads of the K step are issued before the first MFMA (the sched_barrier keeps hipcc from
      // sinking them next to their consumers, which exposes one LDS latency per MFMA pair); the MFMAs then start
      // on counted lgkmcnt as the fragments land, and the second wave of the SIMD fills the gaps
      uint4 fa[KB / 32][2], fw[KB / 32][2];
#pragma unroll
      for (int kk = 0; kk < KB / 32; ++kk) {
        const int chunk = kk * 2 + (lane >> 5);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int rowa = wm * 64 + t * 32 + (lane & 31);
          fa[kk][t] = *reinterpret_cast<const uint4*>(la + rowa * KB + ((chunk ^ SW(rowa)) << 4));
          const int roww = wn * 64 + t * 32 + (lane & 31);
          fw[kk][t] = *reinterpret_cast<const uint4*>(lw + roww * KB + ((chunk ^ SW(roww)) << 4));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < KB / 32; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = MT::mma(fa[kk][i], fw[kk][j], acc[i][j]);
      __syncthreads();
      cur ^= 1;
    }
  }

  // epilogue. C layout of a 32x32 tile: col n = lane&31, row m = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + (lane & 31);
      if (n >= N) continue;
      float ws = 1.0f, bs = 0.0f;
      if constexpr (!SPLITK) {
        if constexpr (KIND == kI8) ws = epi.w_scale[n];
        if constexpr (KIND == kFP8) ws = epi.w_scale[epi.w_scale_n > 1 ? n : 0];
        if (epi.bias) bs = load16(epi.bias, n, epi.out_bf16);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= M) continue;
        const int64_t idx = (int64_t)m * N + n;
        if constexpr (KIND == kI8) {
          const int a = acc[i][j][r];
          if constexpr (SPLITK) {
            atomicAdd(epi.acc_out + idx, a);
          } else {
            if (epi.acc_out) epi.acc_out[idx] = a;
            if (out_base) store16(out_base, idx, (float)a * epi.a_scale[m] * ws + bs, epi.out_bf16);
          }
        } else if constexpr (KIND == kFP8) {
          const float as = epi.a_scale[epi.a_scale_n > 1 ? m : 0];
          store16(out_base, idx, as * (ws * acc[i][j][r]) + bs, epi.out_bf16);
        } else {
          if constexpr (SPLITK)   // 16-bit kinds: this K slice's fp32 partial into its slab (plain store; f32_splitk_reduce_zero_kernel sums in order)
            reinterpret_cast<float*>(epi.acc_out)[(int64_t)blockIdx.z * M * N + idx] = acc[i][j][r];
          else
            store16(out_base, idx, acc[i][j][r] + bs, epi.out_bf16);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// "skinny" kernel for decode-shaped GEMMs (M <= 256 rows per m-tile): the weight matrix is streamed from
// HBM exactly once while the small activation matrix is re-read from L2 by every workgroup.
//   * block tile 256 rows x (NT*32) columns, NT = 1..3 chosen on the host so that the grid is ONE wave of
//     workgroups (<= 256 CUs, no tail round); 4 waves (one per SIMD, up to 512 VGPRs each), each wave
//     64 rows x all NT column tiles.
//   * the dependency chain per workgroup is what bounds a weight-streaming GEMM (one HBM round trip per
//     K step), so operands are prefetched DEPTH=3 K-steps ahead in registers (issue order == stage order,
//     so hipcc's counted s_waitcnt vmcnt(N) keeps 2 stages in flight across the LDS write + barrier):
//     3 x (32 KiB of A + NT*4 KiB of W) outstanding per CU.
//   * int8: optional split-K (grid.z) with exact int32 atomics when the N tiles alone cannot fill the chip.
// ------------------------------------------------------------------------------------------------
constexpr int SK_BM = 256;

template <int KIND, int NT, bool SPLITK, int WAVES, int DEPTH, int BM = SK_BM>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void gemm_skinny_kernel(const uint8_t* __restrict__ A,
                                                                   const uint8_t* __restrict__ W, int M, int N,
                                                                   int64_t Kb, int ksteps_per_split, GemmEpi epi) {
  using MT = MmaTraits<KIND>;
  using acc_t = typename MT::acc_t;
  constexpr int BNW = NT * 32;                            // columns per workgroup
  constexpr int A_BYTES = BM * BKB, W_BYTES = BNW * BKB;           // BM = 256 rows, or 128 for M <= 128 (4 waves)
  constexpr int SK_THREADS = WAVES * 64, MT_PER_WAVE = (BM / 32) / WAVES;  // m-tiles (32 rows) per wave
  static_assert(MT_PER_WAVE >= 1 && MT_PER_WAVE * WAVES * 32 == BM, "rows per workgroup = waves x m-tiles x 32");
  constexpr int NLA = A_BYTES / 16 / SK_THREADS;
  constexpr int NLW = (W_BYTES / 16 + SK_THREADS - 1) / SK_THREADS;
  __shared__ __attribute__((aligned(16))) uint8_t lds[2][A_BYTES + W_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = blockIdx.x * BNW, m0 = blockIdx.y * BM;
  const int total_ksteps = (int)((Kb + BKB - 1) / BKB);
  const int ks_begin = blockIdx.z * ksteps_per_split;
  int ks_end = ks_begin + ksteps_per_split;
  ks_end = ks_end > total_ksteps ? total_ksteps : ks_end;
  const int nsteps = ks_end - ks_begin;
  // K-phase stagger: workgroup b walks its K steps starting at a different offset (wrapping around). All
  // workgroups read the SAME activation slab per K step; with a row stride of K bytes (3584 = 28 lines,
  // 18944 = 148 lines) the 256 rows of one slab fall on only 4 of the 16 L2 channels, so lock-step readers
  // serialise on them. Staggering spreads concurrent readers over all slabs (= all channels). Integer
  // accumulation is order independent; for floating kinds the order is still a fixed function of blockIdx.
  const int phase = nsteps > 0 ? (int)((blockIdx.x * 5u + blockIdx.z * 3u) % (unsigned)nsteps) : 0;

  // three register stages with compile-time names (never runtime-indexed, never address-taken: they must
  // stay in VGPRs). NOTE: every load is unconditional (clamped addresses; Kb % BKB == 0 is checked on the
  // host; tail stages re-load the last K step) -- a branch around a global load makes hipcc fall back to
  // s_waitcnt vmcnt(0), which collapses the 3-stage pipeline to depth 1.
  static_assert(DEPTH == 2 || DEPTH == 3, "the main loop is hand-unrolled for 2 or 3 register stages");
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // native vector: plain 16-B load/store, no memcpy
  u32x4 a0[NLA], a1[NLA], a2[NLA], w0[NLW], w1[NLW], w2[NLW];
  // buffer (SRSRC) addressing: one 32-bit per-lane offset per load, the K offset rides in the scalar soffset
  // operand -> no 64-bit address arithmetic in VGPRs (the operands are < 4 GiB each, checked on the host)
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A), 0, (int)((int64_t)M * Kb), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(W), 0, (int)((int64_t)N * Kb), 0x00020000);
  int a_off[NLA], w_off[NLW];
  int lds_off_a[NLA], lds_off_w[NLW];
#pragma unroll
  for (int i = 0; i < NLA; ++i) {
    const int c = tid + i * SK_THREADS, row = c >> 3, col = c & 7;
    int ar = m0 + row; ar = ar < M ? ar : M - 1;
    a_off[i] = (int)((int64_t)ar * Kb + col * 16);
    lds_off_a[i] = row * BKB + ((col ^ ((row >> 1) & 7)) << 4);
  }
#pragma unroll
  for (int i = 0; i < NLW; ++i) {
    int c = tid + i * SK_THREADS;
    const bool live = c < W_BYTES / 16;  // NT*256 chunks over SK_THREADS threads may not divide evenly
    c = live ? c : W_BYTES / 16 - 1;
    const int row = c >> 3, col = c & 7;
    int wr = n0 + row; wr = wr < N ? wr : N - 1;
    w_off[i] = (int)((int64_t)wr * Kb + col * 16);
    lds_off_w[i] = live ? A_BYTES + row * BKB + ((col ^ ((row >> 1) & 7)) << 4) : -1;
  }
#ifdef XM_ABL_NO_LOAD  /* ablation build: no global loads at all */
#define XM_SK_LOAD(KS, AR, WR)                                                                     \
  {                                                                                                \
    static_for<NLA>([&](auto I_) { AR[I_] = u32x4{1u, 2u, 3u, (unsigned)(KS)}; });                 \
    static_for<NLW>([&](auto I_) { WR[I_] = u32x4{1u, 2u, 3u, (unsigned)(KS)}; });                 \
  }
#else
#define XM_SK_LOAD(KS, AR, WR)                                                                     \
  {                                                                                                \
    int ks_ = (KS) + phase;                                                                        \
    ks_ = ks_begin + (ks_ % nsteps);                                                               \
    const int kb0_ = ks_ * BKB;                                                                    \
    static_for<NLA>([&](auto I_) { AR[I_] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, a_off[I_], kb0_, 0); }); \
    static_for<NLW>([&](auto I_) { WR[I_] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_off[I_], kb0_, 0); }); \
  }
#endif
#define XM_SK_WRITE(BUF, AR, WR)                                                                   \
  {                                                                                                \
    static_for<NLA>([&](auto I_) { *reinterpret_cast<u32x4*>(&lds[BUF][lds_off_a[I_]]) = AR[I_]; });               \
    static_for<NLW>([&](auto I_) { if (lds_off_w[I_] >= 0) *reinterpret_cast<u32x4*>(&lds[BUF][lds_off_w[I_]]) = WR[I_]; }); \
  }

  acc_t acc[MT_PER_WAVE][NT];
#pragma unroll
  for (int t = 0; t < MT_PER_WAVE; ++t)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[t][j] = MT::zero();

  auto compute = [&](int buf) {
    const uint8_t* la = lds[buf];
    const uint8_t* lw = lds[buf] + A_BYTES;
    if constexpr (KIND == kFP8) {
      // fp8 on the gfx950 rate: v_mfma_f32_32x32x64_f8f6f4 takes 32 bytes of K per lane and operand = the fragment pair
      // (chunks 4 kp + h, 4 kp + 2 + h) of the 128-byte K step (see mma_fp8x2 in gemm_types.h)
#pragma unroll
      for (int kp = 0; kp < BKB / 64; ++kp) {
        const int c0 = kp * 4 + (lane >> 5), c1 = c0 + 2;
        u32x4 fa0[MT_PER_WAVE], fa1[MT_PER_WAVE];
#pragma unroll
        for (int t = 0; t < MT_PER_WAVE; ++t) {
          const int rowa = wave * (32 * MT_PER_WAVE) + t * 32 + (lane & 31);
          fa0[t] = *reinterpret_cast<const u32x4*>(la + rowa * BKB + ((c0 ^ ((rowa >> 1) & 7)) << 4));
          fa1[t] = *reinterpret_cast<const u32x4*>(la + rowa * BKB + ((c1 ^ ((rowa >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int roww = j * 32 + (lane & 31);
          const u32x4 fw0 = *reinterpret_cast<const u32x4*>(lw + roww * BKB + ((c0 ^ ((roww >> 1) & 7)) << 4));
          const u32x4 fw1 = *reinterpret_cast<const u32x4*>(lw + roww * BKB + ((c1 ^ ((roww >> 1) & 7)) << 4));
#pragma unroll
          for (int t = 0; t < MT_PER_WAVE; ++t) acc[t][j] = mma_fp8x2(fa0[t], fa1[t], fw0, fw1, acc[t][j]);
        }
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < BKB / 32; ++kk) {
      const int chunk = kk * 2 + (lane >> 5);
      uint4 fa[MT_PER_WAVE];
#pragma unroll
      for (int t = 0; t < MT_PER_WAVE; ++t) {
        const int rowa = wave * (32 * MT_PER_WAVE) + t * 32 + (lane & 31);
        fa[t] = *reinterpret_cast<const uint4*>(la + rowa * BKB + ((chunk ^ ((rowa >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int roww = j * 32 + (lane & 31);
        const uint4 fw = *reinterpret_cast<const uint4*>(lw + roww * BKB + ((chunk ^ ((roww >> 1) & 7)) << 4));
#pragma unroll
        for (int t = 0; t < MT_PER_WAVE; ++t) {
#ifdef XM_ABL_NO_MFMA
          asm volatile("" ::"v"(fa[t].x), "v"(fw.x));  // ablation build: keep the LDS reads, drop the MFMA
#else
          acc[t][j] = MT::mma(fa[t], fw, acc[t][j]);
#endif
        }
      }
    }
  };

  if (nsteps > 0) {
    // prologue: DEPTH stages in flight; stage 0 -> LDS
    XM_SK_LOAD(0, a0, w0)
    XM_SK_LOAD(1, a1, w1)
    if constexpr (DEPTH == 3) XM_SK_LOAD(2, a2, w2)
    XM_SK_WRITE(0, a0, w0)
    __syncthreads();
    // one K step: refill the register stage that was consumed (stage i+DEPTH), compute stage i from LDS, then
    // publish stage i+1. Loads are issued in stage order, so waiting for stage i+1 leaves the later ones in flight.
#define XM_SK_STEP(RA, RW, NA, NW)                          \
    XM_SK_LOAD(i + DEPTH, RA, RW)                           \
    compute(i & 1);                                         \
    if (i + 1 < nsteps) XM_SK_WRITE((i + 1) & 1, NA, NW)    \
    __syncthreads();                                        \
    if (++i >= nsteps) break;
    int i = 0;
    if constexpr (DEPTH == 3) {
      while (true) {
        XM_SK_STEP(a0, w0, a1, w1)
        XM_SK_STEP(a1, w1, a2, w2)
        XM_SK_STEP(a2, w2, a0, w0)
      }
    } else {
      while (true) {
        XM_SK_STEP(a0, w0, a1, w1)
        XM_SK_STEP(a1, w1, a0, w0)
      }
    }
#undef XM_SK_STEP
  }
#undef XM_SK_LOAD
#undef XM_SK_WRITE

  // epilogue: C tile layout col n = lane&31, row m = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + j * 32 + (lane & 31);
    if (n >= N) continue;
    float ws = 1.0f, bs = 0.0f;
    if constexpr (!SPLITK) {
      if constexpr (KIND == kI8) ws = epi.w_scale[n];
      if constexpr (KIND == kFP8) ws = epi.w_scale[epi.w_scale_n > 1 ? n : 0];
      if (epi.bias) bs = load16(epi.bias, n, epi.out_bf16);
    }
#pragma unroll
    for (int t = 0; t < MT_PER_WAVE; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wave * (32 * MT_PER_WAVE) + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= M) continue;
        const int64_t idx = (int64_t)m * N + n;
        if constexpr (KIND == kI8) {
          const int a = acc[t][j][r];
          if constexpr (SPLITK) {
#ifdef XM_ABL_SK_STORE  /* ablation build (timing only, WRONG results): what the split-K atomics cost */
            epi.acc_out[idx] = a;
#else
            atomicAdd(epi.acc_out + idx, a);
#endif
          } else {
            if (epi.acc_out) epi.acc_out[idx] = a;
            if (epi.out) store16(epi.out, idx, (float)a * epi.a_scale[m] * ws + bs, epi.out_bf16);
          }
        } else if constexpr (SPLITK) {
          // fp8 / 16-bit split-K: this K slice's fp32 partial sums go to slab blockIdx.z of the workspace (plain stores);
          // the reduce kernel adds the slabs in slice order (deterministic), applies scales / bias and re-zeroes them
          reinterpret_cast<float*>(epi.acc_out)[(int64_t)blockIdx.z * M * N + idx] = acc[t][j][r];
        } else if constexpr (KIND == kFP8) {
          const float as = epi.a_scale[epi.a_scale_n > 1 ? m : 0];
          store16(epi.out, idx, as * (ws * acc[t][j][r]) + bs, epi.out_bf16);
        } else {
          store16(epi.out, idx, acc[t][j][r] + bs, epi.out_bf16);
        }
      }
  }
}

// dequant epilogue after int8 split-K: reads the int32 sums, applies scales + bias, and re-zeroes the
// workspace for the next call (the workspace is zero-filled once when it is registered)
__global__ __launch_bounds__(256) void i8_splitk_epilogue_zero_kernel(int32_t* __restrict__ acc, int64_t M, int64_t N,
                                                                      GemmEpi epi) {
  const int64_t total = M * N;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = idx / N, n = idx - m * N;
    float bs = epi.bias ? load16(epi.bias, n, epi.out_bf16) : 0.0f;
    const int32_t a = acc[idx];
    acc[idx] = 0;
    store16(epi.out, idx, (float)a * epi.a_scale[m] * epi.w_scale[n] + bs, epi.out_bf16);
  }
}

// reduce after 16-bit split-K: out = sum over the K slices IN ORDER of the fp32 partial slabs (+ bias), and the slabs
// are re-zeroed (the workspace is zero at rest: the int8 split-K path relies on it)
__global__ __launch_bounds__(256) void f32_splitk_reduce_zero_kernel(float* __restrict__ part, int64_t M, int64_t N,
                                                                     int splits, GemmEpi epi) {
  const int64_t total = M * N;
  for (int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x * 4) {  // N % 4 == 0 (checked on the host): four columns of one row
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sl = 0; sl < splits; ++sl) {
      float4* p = reinterpret_cast<float4*>(part + (int64_t)sl * total + idx);
      const float4 v = *p;
      *p = make_float4(0.f, 0.f, 0.f, 0.f);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const int64_t m = idx / N, n = idx - m * N;
    const float a4[4] = {acc.x, acc.y, acc.z, acc.w};
    const float as = epi.a_scale ? epi.a_scale[epi.a_scale_n > 1 ? m : 0] : 1.0f;  // fp8: the unsplit epilogue's formula
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float bs = epi.bias ? load16(epi.bias, n + j, epi.out_bf16) : 0.0f;
      const float v = epi.a_scale ? as * (epi.w_scale[epi.w_scale_n > 1 ? n + j : 0] * a4[j]) + bs : a4[j] + bs;
      store16(epi.out, idx + j, v, epi.out_bf16);
    }
  }
}

struct SkinnyPlan { int nt; int splits; };
static int g_sk_disable = -2;
XM_TUNE_VAR(g_sk_nt, "XLLM_MI355_SKINNY_NT", -1);
XM_TUNE_VAR(g_sk_splits, "XLLM_MI355_SKINNY_SPLITS", -1);

// 16-bit kinds: pick (NT, K slices) by a small cost model instead of the int8 rule below (which is tuned on the W8A8
// decode shapes and left alone): rounds of 256 workgroups x K steps per slice x relative step cost (the 32 KiB activation
// stage dominates, each 32-column block adds 4 KiB of weights), plus the reduce pass when K is split
inline SkinnyPlan plan_skinny_half(int64_t M, int64_t N, int ksteps, bool can_split, int max_slices) {
  const int64_t bm = M <= 128 ? 128 : SK_BM;
  const int64_t m_tiles = (M + bm - 1) / bm;
  const int64_t nt32 = (N + 31) / 32;
  SkinnyPlan best{1, 1};
  double best_cost = 1e30;
  for (int nt = 1; nt <= 5; ++nt) {
    const int64_t wgs = ((nt32 + nt - 1) / nt) * m_tiles;
    const int smax = can_split ? (max_slices < 16 ? max_slices : 16) : 1;
    for (int sp = 1; sp <= smax; ++sp) {
      const int per = (ksteps + sp - 1) / sp;
      if (sp > 1 && per < 6) break;
      const int64_t rounds = (wgs * sp + 255) / 256;
      // units ~ 0.06 us; the reduce pass is one more launch plus sp fp32 slabs of M x N written, read and re-zeroed
      const double cost = (double)rounds * (per + 10) * ((bm == 128 ? 4.0 : 8.0) + nt) +
                          (sp > 1 ? 70.0 + 4.0 * sp + (double)M * N * sp / 30000.0 : 0.0);
      if (cost < best_cost) { best_cost = cost; best = SkinnyPlan{nt, sp}; }
    }
  }
  return best;
}

inline SkinnyPlan plan_skinny(int64_t M, int64_t N, int ksteps, bool can_split) {  // (m_tiles = 1 for every M <= 256)
  const int64_t m_tiles = (M + SK_BM - 1) / SK_BM;
  const int64_t nt32 = (N + 31) / 32;
  int nt = (int)((nt32 * m_tiles + 255) / 256);  // smallest NT whose grid fits one round of 256 CUs
  if (nt == 1 && ksteps >= 64 && can_split) nt = 2;  // long K, few columns: wider tiles + more K slices
  nt = nt < 1 ? 1 : (nt > 5 ? 5 : nt);
  if (g_sk_nt > 0) nt = g_sk_nt > 5 ? 5 : g_sk_nt;
  const int64_t wgs = ((nt32 + nt - 1) / nt) * m_tiles;
  int splits = 1;
  if (can_split) {
    splits = (int)(256 / wgs);
    const int by_k = ksteps / 6 > 0 ? ksteps / 6 : 1;
    splits = splits > by_k ? by_k : splits;
    splits = splits < 1 ? 1 : (splits > 32 ? 32 : splits);
    if (g_sk_splits > 0) splits = g_sk_splits;
  }
  return SkinnyPlan{nt, splits};
}


template <int KIND, int NT, int WV, int DP, int BM = SK_BM>
int launch_skinny_cfg(const void* A, const void* W, int64_t M, int64_t N, int64_t Kb, GemmEpi epi, int splits,
                      void* workspace, hipStream_t s) {
  const int ksteps = (int)((Kb + BKB - 1) / BKB);
  int per = (ksteps + splits - 1) / splits;
  splits = (ksteps + per - 1) / per;
  const dim3 grid((unsigned)((N + NT * 32 - 1) / (NT * 32)), (unsigned)((M + BM - 1) / BM), (unsigned)splits);
  if (splits > 1) {
    if constexpr (KIND == kI8) {
      GemmEpi e2 = epi;
      e2.acc_out = reinterpret_cast<int32_t*>(workspace);
      hipLaunchKernelGGL((gemm_skinny_kernel<KIND, NT, true, WV, DP, BM>), grid, dim3(WV * 64), 0, s, (const uint8_t*)A,
                         (const uint8_t*)W, (int)M, (int)N, Kb, per, e2);
      if (!epi.defer) {
        int64_t blocks = (M * N + 255) / 256;
        blocks = blocks > 1024 ? 1024 : blocks;
        hipLaunchKernelGGL(i8_splitk_epilogue_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                           reinterpret_cast<int32_t*>(workspace), M, N, epi);
      }
    } else if constexpr (KIND == kBF16 || KIND == kF16 || KIND == kFP8) {
      GemmEpi e2 = epi;
      e2.acc_out = reinterpret_cast<int32_t*>(workspace);  // fp32 partial slabs [splits][M][N]
      hipLaunchKernelGGL((gemm_skinny_kernel<KIND, NT, true, WV, DP, BM>), grid, dim3(WV * 64), 0, s, (const uint8_t*)A,
                         (const uint8_t*)W, (int)M, (int)N, Kb, per, e2);
      int64_t blocks = (M * N / 4 + 255) / 256;
      blocks = blocks > 1024 ? 1024 : blocks;
      hipLaunchKernelGGL(f32_splitk_reduce_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                         reinterpret_cast<float*>(workspace), M, N, splits, epi);
    } else {
      return XM_ERR_UNSUPPORTED;
    }
  } else if (epi.defer) {
    GemmEpi e2 = epi;  // single pass: plain int32 stores into the workspace, nothing else
    e2.acc_out = reinterpret_cast<int32_t*>(workspace);
    e2.out = nullptr;
    hipLaunchKernelGGL((gemm_skinny_kernel<KIND, NT, false, WV, DP, BM>), grid, dim3(WV * 64), 0, s, (const uint8_t*)A,
                       (const uint8_t*)W, (int)M, (int)N, Kb, per, e2);
  } else {
    hipLaunchKernelGGL((gemm_skinny_kernel<KIND, NT, false, WV, DP, BM>), grid, dim3(WV * 64), 0, s, (const uint8_t*)A,
                       (const uint8_t*)W, (int)M, (int)N, Kb, per, epi);
  }
  return hip_check_launch();
}

template <int KIND, int NT>
int launch_skinny_nt(const void* A, const void* W, int64_t M, int64_t N, int64_t Kb, GemmEpi epi, int splits,
                     void* workspace, hipStream_t s) {
  // round-1 sweep (tools/gemm_sweep.sh, profiles/r01_gemm_sweep.txt): 8 waves x 32 rows with 2 register stages wins on every
  // decode shape (two waves per SIMD overlap one wave's MFMAs with the other's LDS traffic; a third register stage only costs
  // occupancy) -- the 4-wave and 3-stage forms left the library in round 4.
  // M <= 128: 128-row tile, 4 waves x 32 rows (half the activation stage, half the padded MFMA rows; the 256-row tile for
  // M <= 128, XLLM_MI355_SKINNY_BM128=0, lost its A/B and left with them)
  if (M <= 128) return launch_skinny_cfg<KIND, NT, 4, 2, 128>(A, W, M, N, Kb, epi, splits, workspace, s);
  return launch_skinny_cfg<KIND, NT, 8, 2>(A, W, M, N, Kb, epi, splits, workspace, s);
}

template <int KIND>
int launch_skinny(const void* A, const void* W, int64_t M, int64_t N, int64_t Kb, GemmEpi epi, void* workspace,
                  size_t ws_bytes, hipStream_t s) {
  const int ksteps = (int)((Kb + BKB - 1) / BKB);
  const bool half_kind = KIND == kBF16 || KIND == kF16 || KIND == kFP8;  // kinds that split K through fp32 slabs
  const bool can_split = (KIND == kI8 || (half_kind && N % 4 == 0 && ((uintptr_t)epi.out % 8) == 0)) && workspace &&
                         ws_bytes >= (size_t)M * N * 4 && !epi.acc_out && (epi.out || epi.defer);
  SkinnyPlan p;
  if (half_kind) {  // one fp32 slab per K slice must fit the workspace
    const int64_t fit = can_split ? (int64_t)(ws_bytes / ((size_t)M * N * 4)) : 1;
    p = plan_skinny_half(M, N, ksteps, can_split, (int)(fit > 16 ? 16 : fit));
  } else {
    p = plan_skinny(M, N, ksteps, can_split);
  }
  switch (p.nt) {
    case 1: return launch_skinny_nt<KIND, 1>(A, W, M, N, Kb, epi, p.splits, workspace, s);
    case 2: return launch_skinny_nt<KIND, 2>(A, W, M, N, Kb, epi, p.splits, workspace, s);
    case 3: return launch_skinny_nt<KIND, 3>(A, W, M, N, Kb, epi, p.splits, workspace, s);
    case 4: return launch_skinny_nt<KIND, 4>(A, W, M, N, Kb, epi, p.splits, workspace, s);
    default: return launch_skinny_nt<KIND, 5>(A, W, M, N, Kb, epi, p.splits, workspace, s);
  }
}


template <int KIND>
int launch_gemm(const void* A, const void* W, int64_t M, int64_t N, int64_t Kb, GemmEpi epi, void* workspace,
                size_t ws_bytes, hipStream_t s) {
  if (M == 0 || N == 0) return XM_OK;
  // 128x128 / skinny / 8-phase family behind this entry: no gate_up epilogue, no device tile table (those have their own entries)
  if (!epi_fits(epi, kCapDefer | kCapAccOut | kCapGroupCounts | kCapAddend)) return XM_ERR_UNSUPPORTED;
  if (epi.addend && (KIND != kI8 || epi.defer)) return XM_ERR_UNSUPPORTED;   // (only the un-split 8-phase int8 kernel below honours it)
  if (epi.defer) {  // deferred dequant (see xllm_mi355_scaled_matmul_add_rms_norm): decode-shaped int8 problems only
    if (KIND != kI8 || M > 512 || Kb % BKB != 0 || M * Kb >= (1ll << 31) || N * Kb >= (1ll << 31) || !workspace ||
        ws_bytes < (size_t)M * N * 4)
      return XM_ERR_UNSUPPORTED;
    return launch_skinny<KIND>(A, W, M, N, Kb, epi, workspace, ws_bytes, s);
  }
  if (g_sk_disable == -2) g_sk_disable = xm_switch("XLLM_MI355_SKINNY_DISABLE", 0);   // product switch (general kernel only)
  // decode-shaped problems take the skinny kernel (32-bit buffer offsets: operands < 2 GiB); for the 16-bit / fp8
  // kinds only while the general kernel's 128x128 grid would under-fill the chip (measured at M=256: lm_head
  // 436 vs 553 us, bf16 gate_up 119 vs 146 us in favour of the general kernel)
  // 256x256 8-phase kernel (gemm_p8.hip): XLLM_MI355_P8 = 0 off, 1 forced wherever it is legal, unset = the
  // planner below (round-1 sweep, profiles/r01_gemm_p8.txt): it wins whenever its grid has enough tiles to fill the
  // chip without split-K; small grids stay on the skinny / 128x128 split-K kernels.
  static int p8 = -2;   // product switch, read once
  if (p8 == -2) p8 = xm_switch("XLLM_MI355_P8", -1);
  XM_TUNE_VAR(p8_splits, "XLLM_MI355_P8_SPLITS", -1);
  XM_TUNE_VAR(p8_min_tiles, "XLLM_MI355_P8_MIN_TILES", 50);
  if (p8 && Kb % BKB == 0 && (N & 7) == 0 && ((uintptr_t)epi.out & 15) == 0 && M * Kb < (1ll << 31) &&
      N * Kb < (1ll << 31) && !epi.group_counts) {
    int splits = 1;
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    const bool can_split = KIND == kI8 && workspace && ws_bytes >= (size_t)M * N * 4 && !epi.acc_out && epi.out;
    if (p8 == 1 && can_split && tiles < 256) {
      splits = (int)(256 / tiles);
      const int by_k = (int)(Kb / BKB) / 4 > 0 ? (int)(Kb / BKB) / 4 : 1;
      splits = splits > by_k ? by_k : splits;
    }
    if (p8_splits > 0 && can_split) splits = p8_splits;
    // a long K loop over few tiles (down_proj at M ~ 1024) is better served by the split-K 128x128 kernel
    const bool long_k_few_tiles = KIND == kI8 && tiles < 100 && Kb >= 8192 && can_split;
    if (p8 == 1 || (tiles >= p8_min_tiles && !long_k_few_tiles)) {
      if (splits > 1) {
        if (epi.addend) return XM_ERR_UNSUPPORTED;
        GemmEpi e2 = epi;
        e2.acc_out = reinterpret_cast<int32_t*>(workspace);
        const int rc = launch_gemm_p8<KIND>(A, W, M, N, Kb, e2, workspace, ws_bytes, splits, s);
        if (rc != XM_OK) return rc;
        if constexpr (KIND == kI8) {
          int64_t blocks = (M * N + 255) / 256;
          blocks = blocks > 2048 ? 2048 : blocks;
          hipLaunchKernelGGL(i8_splitk_epilogue_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                             reinterpret_cast<int32_t*>(workspace), M, N, epi);
        }
        return hip_check_launch();
      }
      return launch_gemm_p8<KIND>(A, W, M, N, Kb, epi, workspace, ws_bytes, 1, s);
    }
  }
  if (epi.addend) return XM_ERR_UNSUPPORTED;   // the kernels below have no addend epilogue: the caller adds in a second pass
  const bool skinny_pays = KIND == kI8 || ((M + BM - 1) / BM) * ((N + BN - 1) / BN) < 256;
  if (M <= 512 && Kb % BKB == 0 && M * Kb < (1ll << 31) && N * Kb < (1ll << 31) && skinny_pays && !g_sk_disable)
    return launch_skinny<KIND>(A, W, M, N, Kb, epi, workspace, ws_bytes, s);
  const int m_tiles = (int)((M + BM - 1) / BM), n_tiles = (int)((N + BN - 1) / BN);
  const int ksteps = (int)((Kb + BKB - 1) / BKB);
  int splits = 1;
  if constexpr (KIND == kI8) {
    // split K until ~2 workgroups per CU are available (small-N decode GEMMs), each slice >= 4 K steps
    const int64_t tiles = (int64_t)m_tiles * n_tiles;
    if (workspace && ws_bytes >= (size_t)M * N * 4 && !epi.acc_out && tiles < 384) {
      splits = (int)((512 + tiles - 1) / tiles);
      const int max_by_k = ksteps / 4 > 0 ? ksteps / 4 : 1;
      splits = splits > max_by_k ? max_by_k : splits;
      splits = splits > 16 ? 16 : splits;
    }
  }
  if constexpr (KIND == kBF16 || KIND == kF16) {
    // Tall problems with few columns (MoE router gates: [T, n_experts] at T = 8192 is 64 tiles of 128 x 128 on 256 CUs, each
    // workgroup alone on its CU with every stage -> barrier -> MFMA latency exposed: 50 us for 33 MB, round 6): split K through
    // fp32 slabs in the registered workspace, summed in slice order by f32_splitk_reduce_zero_kernel (deterministic), as the
    // skinny kernel does for decode shapes.
    const int64_t tiles = (int64_t)m_tiles * n_tiles;
    if (workspace && tiles < 128 && ksteps >= 8 && N % 4 == 0 && ((uintptr_t)epi.out % 8) == 0 && !epi.acc_out && epi.out &&
        !epi.group_counts) {
      int64_t sp = 256 / tiles;
      const int64_t fit = (int64_t)(ws_bytes / ((size_t)M * N * 4));
      sp = sp > ksteps / 4 ? ksteps / 4 : sp;
      sp = sp > fit ? fit : sp;
      sp = sp > 16 ? 16 : sp;
      if (sp >= 2) splits = (int)sp;
    }
  }
  const int per = (ksteps + splits - 1) / splits;
  splits = (ksteps + per - 1) / per;
  // rasterised grid (see the kernel): super-blocks of 8x8 tiles, padded to a multiple of 8 super-blocks
  const int lm = m_tiles >= 8 ? 3 : (m_tiles >= 4 ? 2 : (m_tiles >= 2 ? 1 : 0));
  const int n_sb = ((m_tiles + (1 << lm) - 1) >> lm) * ((n_tiles + (1 << (6 - lm)) - 1) >> (6 - lm));
  const unsigned raster_blocks = (unsigned)(((n_sb + 7) / 8) * 8 * 64);
  const dim3 grid(splits > 1 ? (unsigned)(m_tiles * n_tiles) : raster_blocks, 1, (unsigned)splits);
  if (splits > 1) {
    if constexpr (KIND == kI8) {
      // invariant: the registered workspace is all-zero between calls (zero-filled at registration, re-zeroed
      // by the dequant epilogue below), so no memset launch is needed here
      GemmEpi e2 = epi;
      e2.acc_out = reinterpret_cast<int32_t*>(workspace);
      hipLaunchKernelGGL((gemm_kernel<KIND, true, 128, 2>), grid, dim3(256), 0, s, (const uint8_t*)A, (const uint8_t*)W,
                         (int)M, (int)N, Kb, m_tiles, n_tiles, per, e2);
      int64_t blocks = (M * N + 255) / 256;
      blocks = blocks > 2048 ? 2048 : blocks;
      hipLaunchKernelGGL(i8_splitk_epilogue_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                         reinterpret_cast<int32_t*>(workspace), M, N, epi);
    } else if constexpr (KIND == kBF16 || KIND == kF16) {
      GemmEpi e2 = epi;
      e2.acc_out = reinterpret_cast<int32_t*>(workspace);   // (fp32 slabs: the kernel re-interprets)
      hipLaunchKernelGGL((gemm_kernel<KIND, true, 128, 2>), grid, dim3(256), 0, s, (const uint8_t*)A, (const uint8_t*)W,
                         (int)M, (int)N, Kb, m_tiles, n_tiles, per, e2);
      int64_t blocks = (M * N / 4 + 255) / 256;
      blocks = blocks > 1024 ? 1024 : blocks;
      hipLaunchKernelGGL(f32_splitk_reduce_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s,
                         reinterpret_cast<float*>(workspace), M, N, splits, epi);
    }
  } else {
    // K step 64 B + 4 workgroups per CU (4 waves/SIMD) measured +17 % over 128 B + 2 workgroups at M = 8192
    // (profiles/r01_gemm_notes.txt): more independent waves hide the stage -> barrier -> read -> MFMA chain
    XM_TUNE_VAR(kb64, "XLLM_MI355_GEMM_KB64", 1);
    if (kb64 && Kb % 64 == 0)
      hipLaunchKernelGGL((gemm_kernel<KIND, false, 64, 4>), grid, dim3(256), 0, s, (const uint8_t*)A, (const uint8_t*)W,
                         (int)M, (int)N, Kb, m_tiles, n_tiles, per * 2, epi);
    else
      hipLaunchKernelGGL((gemm_kernel<KIND, false, 128, 2>), grid, dim3(256), 0, s, (const uint8_t*)A, (const uint8_t*)W,
                         (int)M, (int)N, Kb, m_tiles, n_tiles, per, epi);
  }
  return hip_check_launch();
}

// grouped GEMM tile table (one workgroup): slot t of the m-tile axis -> (expert, first row, rows, tile inside the expert),
// experts in order, ceil(rows / bm) slots each; the remaining slots get expert = -1
__global__ __launch_bounds__(1024) void group_plan_kernel(const int32_t* __restrict__ counts, int E, int bm,
                                                          int32_t* __restrict__ table, int slots) {
  __shared__ int32_t rows_incl[1024], tiles_incl[1024];
  const int e = threadIdx.x;
  const int c = e < E ? counts[e] : 0;
  rows_incl[e] = c;
  tiles_incl[e] = (c + bm - 1) / bm;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scans
    const int r = e >= d ? rows_incl[e - d] : 0, t = e >= d ? tiles_incl[e - d] : 0;
    __syncthreads();
    rows_incl[e] += r;
    tiles_incl[e] += t;
    __syncthreads();
  }
  const int used = tiles_incl[1023];
  if (e == 0) table[4 * slots] = used < slots ? used : slots;   // live slots (round 6): the 8-phase kernels balance them over the XCDs
  for (int t = used + e; t < slots; t += 1024) reinterpret_cast<int4*>(table)[t] = make_int4(-1, 0, 0, 0);
  if (e < E) {
    const int nt = (c + bm - 1) / bm, t0 = tiles_incl[e] - nt, off = rows_incl[e] - c;
    for (int j = 0; j < nt && t0 + j < slots; ++j) reinterpret_cast<int4*>(table)[t0 + j] = make_int4(e, off, c, j);
  }
}

}  // namespace xm

using namespace xm;

extern "C" {

// workspace for int8 split-K (optional): M*N*4 bytes, registered per device / per stream (workspace.hip); the ops_api-shaped
// entry points below have no workspace argument because the reference operators have none.
static void gemm_ws_for(void* stream, void** ws, size_t* bytes) { ws_get(0, stream, ws, bytes); }
XM_API int xllm_mi355_set_gemm_workspace(void* ws, size_t bytes) {
  if (ws && bytes && hipMemset(ws, 0, bytes) != hipSuccess) return XM_ERR_HIP;
  return ws_set_device(0, ws, bytes);
}
XM_API int xllm_mi355_set_gemm_workspace_for_stream(void* stream, void* ws, size_t bytes) {
  if (ws && bytes && hipMemset(ws, 0, bytes) != hipSuccess) return XM_ERR_HIP;
  return ws_set_stream(0, stream, ws, bytes);
}

int xllm_mi355_scaled_matmul(const int8_t* a, const int8_t* w, const float* a_scale, const float* w_scale,
                             const void* bias, void* out, int32_t* acc_out, int64_t M, int64_t N, int64_t K,
                             int out_dtype, void* stream) {
  if (!a || !w || M < 0 || N < 0 || K <= 0) return XM_ERR_INVALID;
  if (out && (!a_scale || !w_scale)) return XM_ERR_INVALID;
  if (!out && !acc_out) return XM_ERR_INVALID;
  if (out_dtype != XM_BF16 && out_dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (K % 128 != 0 || ((uintptr_t)a % 16) || ((uintptr_t)w % 16)) return XM_ERR_UNSUPPORTED;  // K step = 128 B
  GemmEpi epi{a_scale, M, w_scale, N, bias, out, acc_out, out_dtype == XM_BF16, nullptr, 0};
  void* ws;
  size_t ws_bytes;
  gemm_ws_for(stream, &ws, &ws_bytes);
  return launch_gemm<kI8>(a, w, M, N, K, epi, ws, ws_bytes, (hipStream_t)stream);
}

int xllm_mi355_scaled_matmul_add(const int8_t* a, const int8_t* w, const float* a_scale, const float* w_scale,
                                 const void* bias, const void* c, void* out, int64_t M, int64_t N, int64_t K, int out_dtype,
                                 void* stream) {
  if (!a || !w || !a_scale || !w_scale || !c || !out || M < 0 || N < 0 || K <= 0) return XM_ERR_INVALID;
  if (out_dtype != XM_BF16 && out_dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (K % 128 != 0 || ((uintptr_t)a % 16) || ((uintptr_t)w % 16) || ((uintptr_t)c % 16) || ((uintptr_t)out % 16) || N % 8)
    return XM_ERR_UNSUPPORTED;
  GemmEpi epi{a_scale, M, w_scale, N, bias, out, nullptr, out_dtype == XM_BF16, nullptr, 0};
  epi.addend = c;
  void* ws;
  size_t ws_bytes;
  gemm_ws_for(stream, &ws, &ws_bytes);
  return launch_gemm<kI8>(a, w, M, N, K, epi, ws, ws_bytes, (hipStream_t)stream);
}

int xllm_mi355_scaled_matmul_add_rms_norm(const int8_t* a, const int8_t* w, const float* a_scale,
                                          const float* w_scale, const void* bias, void* residual,
                                          const void* norm_weight, float eps, void* out_norm, int8_t* out_q,
                                          float* out_q_scale, int64_t M, int64_t N, int64_t K, int dtype, void* stream) {
  if (!a || !w || !a_scale || !w_scale || !residual || !norm_weight || M < 0 || N <= 0 || K <= 0) return XM_ERR_INVALID;
  if ((out_q != nullptr) == (out_norm != nullptr)) return XM_ERR_INVALID;  // exactly one output form
  if (out_q && !out_q_scale) return XM_ERR_INVALID;
  if (dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (K % 128 != 0 || ((uintptr_t)a % 16) || ((uintptr_t)w % 16)) return XM_ERR_UNSUPPORTED;
  if (M == 0) return XM_OK;
  void* ws;
  size_t ws_bytes;
  gemm_ws_for(stream, &ws, &ws_bytes);
  if (!ws || ws_bytes < (size_t)M * N * 4) return XM_ERR_WORKSPACE;
  GemmEpi epi{a_scale, M, w_scale, N, bias, nullptr, nullptr, dtype == XM_BF16, nullptr, 0, 1};
  const int rc = launch_gemm<kI8>(a, w, M, N, K, epi, ws, ws_bytes, (hipStream_t)stream);
  if (rc != XM_OK) return rc;
  return launch_acc_add_rms_norm(out_q ? (void*)out_q : out_norm, out_q_scale, reinterpret_cast<int32_t*>(ws),
                                 a_scale, w_scale, bias, residual, norm_weight, eps, M, N, dtype, out_q != nullptr,
                                 (hipStream_t)stream);
}

int xllm_mi355_pack_weight_i8(const int8_t* w, int8_t* packed, int64_t N, int64_t K, void* stream) {
  if (!w || !packed || N <= 0 || K <= 0) return XM_ERR_INVALID;
  return launch_pack_weight_i8(w, packed, N, K, (hipStream_t)stream);
}

int xllm_mi355_scaled_matmul_packed(const int8_t* a, const int8_t* w_packed, const float* a_scale, const float* w_scale,
                                    const void* bias, void* out, int32_t* acc_out, int64_t M, int64_t N, int64_t K,
                                    int out_dtype, void* workspace, size_t ws_bytes, void* stream) {
  if (!a || !w_packed || M < 0 || N < 0 || K <= 0) return XM_ERR_INVALID;
  if (out && (!a_scale || !w_scale)) return XM_ERR_INVALID;
  if (!out && !acc_out) return XM_ERR_INVALID;
  if (out_dtype != XM_BF16 && out_dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (M == 0 || N == 0) return XM_OK;
  GemmEpi epi{a_scale, M, w_scale, N, bias, out, acc_out, out_dtype == XM_BF16, nullptr, 0};
  return launch_gemm_ws_i8(a, w_packed, M, N, K, epi, workspace, ws_bytes, nullptr, (hipStream_t)stream);
}

int xllm_mi355_scaled_matmul_add_rms_norm_packed(const int8_t* a, const int8_t* w_packed, const float* a_scale,
                                                 const float* w_scale, const void* bias, void* residual,
                                                 const void* norm_weight, float eps, void* out_norm, int8_t* out_q,
                                                 float* out_q_scale, int64_t M, int64_t N, int64_t K, int dtype,
                                                 void* workspace, size_t ws_bytes, void* stream) {
  if (!a || !w_packed || !a_scale || !w_scale || !residual || !norm_weight || M < 0 || N <= 0 || K <= 0) return XM_ERR_INVALID;
  if ((out_q != nullptr) == (out_norm != nullptr)) return XM_ERR_INVALID;  // exactly one output form
  if (out_q && !out_q_scale) return XM_ERR_INVALID;
  if (dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (M == 0) return XM_OK;
  if (!workspace || ws_bytes < (size_t)M * N * 4) return XM_ERR_WORKSPACE;
  GemmEpi epi{a_scale, M, w_scale, N, bias, nullptr, nullptr, dtype == XM_BF16, nullptr, 0, 1};
  int n_slabs = 1;
  const int rc = launch_gemm_ws_i8(a, w_packed, M, N, K, epi, workspace, ws_bytes, &n_slabs, (hipStream_t)stream);
  if (rc != XM_OK) return rc;
  return launch_acc_add_rms_norm(out_q ? (void*)out_q : out_norm, out_q_scale, reinterpret_cast<int32_t*>(workspace),
                                 a_scale, w_scale, bias, residual, norm_weight, eps, M, N, dtype, out_q != nullptr,
                                 (hipStream_t)stream, n_slabs);
}

int xllm_mi355_scaled_matmul_rope_cache_packed(const int8_t* a, const int8_t* w_packed, const float* a_scale,
                                               const float* w_scale, const void* bias, void* qkv, int64_t M, int64_t N,
                                               int64_t K, int dtype, const int64_t* positions, const void* cos_sin_cache,
                                               const int32_t* slot_ids, void* k_cache, void* v_cache, int64_t n_q_heads,
                                               int64_t n_kv_heads, int64_t head_size, int64_t rot_dim, int64_t block_size,
                                               int64_t n_blocks, int is_neox, void* workspace, size_t ws_bytes,
                                               void* stream) {
  if (!a || !w_packed || !a_scale || !w_scale || !qkv || !positions || !cos_sin_cache || !slot_ids || !k_cache || !v_cache ||
      M < 0 || N <= 0 || K <= 0 || block_size <= 0)
    return XM_ERR_INVALID;
  if (dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (M == 0) return XM_OK;
  if (!workspace || ws_bytes < (size_t)M * N * 4) return XM_ERR_WORKSPACE;
  if (N != (n_q_heads + 2 * n_kv_heads) * head_size || N % 4 || N * 2 > 64 * 1024 || rot_dim <= 0 || (rot_dim & 1) ||
      rot_dim > head_size || ((uintptr_t)w_scale % 16))
    return XM_ERR_UNSUPPORTED;               // checked BEFORE the GEMM is launched: a declined call has no side effect
  GemmEpi epi{a_scale, M, w_scale, N, bias, nullptr, nullptr, dtype == XM_BF16, nullptr, 0, 1};
  int n_slabs = 1;
  const int rc = launch_gemm_ws_i8(a, w_packed, M, N, K, epi, workspace, ws_bytes, &n_slabs, (hipStream_t)stream);
  if (rc != XM_OK) return rc;
  return launch_slab_rope_and_cache(reinterpret_cast<const int32_t*>(workspace), n_slabs, a_scale, w_scale, bias, qkv, M, N,
                                    positions, cos_sin_cache, slot_ids, k_cache, v_cache, n_q_heads, n_kv_heads, head_size,
                                    rot_dim, block_size, n_blocks, is_neox, dtype, (hipStream_t)stream);
}

int xllm_mi355_scaled_matmul_gate_up_act(const int8_t* a, const int8_t* w, const int8_t* w_packed, const float* a_scale,
                                         const float* w_scale, const void* bias, void* act_out, float* row_amax, int64_t M,
                                         int64_t N, int64_t K, int dtype, void* workspace, size_t ws_bytes, void* stream) {
  if (!a || (!w && !w_packed) || !a_scale || !w_scale || !act_out || !row_amax || M < 0 || N <= 0 || K <= 0)
    return XM_ERR_INVALID;
  if (dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (N % 256 != 0 || K % 128 != 0 || ((uintptr_t)a % 16) || ((uintptr_t)act_out % 16) || ((uintptr_t)w_scale % 16))
    return XM_ERR_UNSUPPORTED;                                   // I = N / 2 a multiple of the 128-column act tile
  if (M == 0) return XM_OK;
  GemmEpi epi{a_scale, M, w_scale, N, bias, nullptr, nullptr, dtype == XM_BF16, nullptr, 0};
  epi.gate_up = 1;
  epi.act_out = act_out;
  epi.row_amax = row_amax;
  if (w_packed && M <= 512) {
    const int rc = launch_gemm_ws_i8(a, w_packed, M, N, K, epi, workspace, ws_bytes, nullptr, (hipStream_t)stream);
    if (rc != XM_ERR_UNSUPPORTED) return rc;
  }
  if (!w || ((uintptr_t)w % 16)) return XM_ERR_UNSUPPORTED;
  return launch_gemm_p8<kI8>(a, w, M, N, K, epi, nullptr, 0, 1, (hipStream_t)stream);   // 256 x 256 tiles, any M
}

int xllm_mi355_fp8_scaled_matmul(const uint8_t* a, const uint8_t* w, const float* a_scale, int64_t a_scale_numel,
                                 const float* w_scale, int64_t w_scale_numel, const void* bias, void* out, int64_t M,
                                 int64_t N, int64_t K, int out_dtype, void* stream) {
  if (!a || !w || !a_scale || !w_scale || !out || M < 0 || N < 0 || K <= 0) return XM_ERR_INVALID;
  if (out_dtype != XM_BF16 && out_dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if ((a_scale_numel != 1 && a_scale_numel != M) || (w_scale_numel != 1 && w_scale_numel != N)) return XM_ERR_INVALID;
  if (K % 128 != 0 || ((uintptr_t)a % 16) || ((uintptr_t)w % 16)) return XM_ERR_UNSUPPORTED;
  GemmEpi epi{a_scale, a_scale_numel, w_scale, w_scale_numel, bias, out, nullptr, out_dtype == XM_BF16, nullptr, 0};
  void* ws = nullptr;  // decode shapes split K through the registered workspace (fp32 slabs, deterministic reduce)
  size_t ws_bytes = 0;
  gemm_ws_for(stream, &ws, &ws_bytes);
  return launch_gemm<kFP8>(a, w, M, N, K, epi, ws, ws_bytes, (hipStream_t)stream);
}

int xllm_mi355_pack_weight_fp8(const uint8_t* w, uint8_t* packed, int64_t N, int64_t K, void* stream) {
  if (!w || !packed || N <= 0 || K <= 0) return XM_ERR_INVALID;
  return launch_pack_weight_i8(w, packed, N, K, (hipStream_t)stream);  // a byte permutation: the same for both 8-bit kinds
}

int xllm_mi355_fp8_scaled_matmul_packed(const uint8_t* a, const uint8_t* w_packed, const float* a_scale,
                                        int64_t a_scale_numel, const float* w_scale, int64_t w_scale_numel,
                                        const void* bias, void* out, int64_t M, int64_t N, int64_t K, int out_dtype,
                                        void* workspace, size_t ws_bytes, void* stream) {
  if (!a || !w_packed || !a_scale || !w_scale || !out || M < 0 || N < 0 || K <= 0) return XM_ERR_INVALID;
  if (out_dtype != XM_BF16 && out_dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if ((a_scale_numel != 1 && a_scale_numel != M) || (w_scale_numel != 1 && w_scale_numel != N)) return XM_ERR_INVALID;
  if (M == 0 || N == 0) return XM_OK;
  GemmEpi epi{a_scale, a_scale_numel, w_scale, w_scale_numel, bias, out, nullptr, out_dtype == XM_BF16, nullptr, 0};
  return launch_gemm_ws_fp8(a, w_packed, M, N, K, epi, workspace, ws_bytes, (hipStream_t)stream);
}

int xllm_mi355_pack_weight_16(const void* w, void* packed, int64_t N, int64_t K, void* stream) {
  if (!w || !packed || N <= 0 || K <= 0) return XM_ERR_INVALID;
  return launch_pack_weight_i8(w, packed, N, K * 2, (hipStream_t)stream);   // the byte permutation of the 8-bit kinds, rows of 2 K bytes
}

int xllm_mi355_matmul_packed(const void* a, const void* w_packed, const void* bias, void* out, int64_t M, int64_t N, int64_t K,
                             int dtype, void* workspace, size_t ws_bytes, void* stream) {
  if (!a || !w_packed || !out || M < 0 || N < 0 || K <= 0) return XM_ERR_INVALID;
  if (dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (M == 0 || N == 0) return XM_OK;
  GemmEpi epi{nullptr, 0, nullptr, 0, bias, out, nullptr, dtype == XM_BF16, nullptr, 0};
  return launch_gemm_ws_h16(a, w_packed, M, N, K * 2, epi, workspace, ws_bytes, (hipStream_t)stream);
}

int xllm_mi355_matmul_gate_up_act(const void* a, const void* w_packed, const void* bias, void* act_out, int64_t M, int64_t N,
                                  int64_t K, int dtype, void* workspace, size_t ws_bytes, void* stream) {
  if (!a || !w_packed || !act_out || M < 0 || N <= 0 || K <= 0) return XM_ERR_INVALID;
  if (dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (N % 32 != 0 || ((uintptr_t)act_out % 16)) return XM_ERR_UNSUPPORTED;
  if (M == 0) return XM_OK;
  GemmEpi epi{nullptr, 0, nullptr, 0, bias, nullptr, nullptr, dtype == XM_BF16, nullptr, 0};
  epi.gate_up = 1;
  epi.act_out = act_out;
  return launch_gemm_ws_h16(a, w_packed, M, N, K * 2, epi, workspace, ws_bytes, (hipStream_t)stream);
}

size_t xllm_mi355_matmul_argmax_workspace_bytes(int64_t M, int64_t N) {
  return M > 0 && N > 0 ? (size_t)M * (size_t)(N / 16 > 1024 ? N / 16 : 1024) * 8 : 0;
}

int xllm_mi355_matmul_argmax_packed(const void* a, const void* w_packed, const void* bias, int64_t* out_idx, float* out_val, int64_t M,
                                    int64_t N, int64_t K, int dtype, void* workspace, size_t ws_bytes, void* stream) {
  if (!a || !w_packed || !out_idx || M < 0 || N <= 0 || K <= 0) return XM_ERR_INVALID;
  if (dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (M == 0) return XM_OK;
  GemmEpi epi{nullptr, 0, nullptr, 0, bias, nullptr, nullptr, dtype == XM_BF16, nullptr, 0};
  return launch_gemm_ws_h16_argmax(a, w_packed, M, N, K * 2, epi, out_idx, out_val, workspace, ws_bytes, (hipStream_t)stream);
}

int xllm_mi355_matmul(const void* a, const void* w, const void* bias, void* out, int64_t M, int64_t N, int64_t K,
                      int dtype, void* stream) {
  if (!a || !w || !out || M < 0 || N < 0 || K <= 0) return XM_ERR_INVALID;
  if (dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (K % 64 != 0 || ((uintptr_t)a % 16) || ((uintptr_t)w % 16)) return XM_ERR_UNSUPPORTED;  // K step = 128 B
  GemmEpi epi{nullptr, 0, nullptr, 0, bias, out, nullptr, dtype == XM_BF16, nullptr, 0};
  // decode-shaped problems with a long K and few columns split K through the registered workspace (fp32 partial slabs,
  // reduced in slice order: deterministic); without a workspace the launch is unsplit as before
  void* ws = nullptr;
  size_t ws_bytes = 0;
  gemm_ws_for(stream, &ws, &ws_bytes);
  if (M <= 64) {  // one pass over the weights: the weight-stream kernel (gemm_wsb.hip); declines shapes that are not its own
    const int rc = dtype == XM_BF16
                       ? launch_gemm_wsb_dense<bf16_t>(a, w, bias, out, M, N, K, ws, ws_bytes, (hipStream_t)stream)
                       : launch_gemm_wsb_dense<f16_t>(a, w, bias, out, M, N, K, ws, ws_bytes, (hipStream_t)stream);
    if (rc != XM_ERR_UNSUPPORTED) return rc;
  }
  if (dtype == XM_BF16) return launch_gemm<kBF16>(a, w, M, N, K * 2, epi, ws, ws_bytes, (hipStream_t)stream);
  return launch_gemm<kF16>(a, w, M, N, K * 2, epi, ws, ws_bytes, (hipStream_t)stream);
}

static int group_gemm_impl(const void* a, const void* w, const int32_t* token_count, void* out, int64_t max_rows,
                           int64_t n_experts, int64_t N, int64_t K, int dtype, const int32_t* gather_rows,
                           int64_t gather_div, int64_t src_rows, void* stream) {
  if (!a || !w || !token_count || !out || max_rows < 0 || n_experts <= 0 || N <= 0 || K <= 0) return XM_ERR_INVALID;
  if (dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (K % 64 != 0 || ((uintptr_t)a % 16) || ((uintptr_t)w % 16)) return XM_ERR_UNSUPPORTED;
  if (gather_rows && (gather_div <= 0 || src_rows <= 0 || src_rows * K * 2 >= (1ll << 31))) return XM_ERR_INVALID;
  if (max_rows == 0) return XM_OK;
  GemmEpi epi{nullptr, 0, nullptr, 0, nullptr, out, nullptr, dtype == XM_BF16, token_count, (int)n_experts};
  hipStream_t s = (hipStream_t)stream;
  if (max_rows <= 16 * n_experts) {  // a handful of rows per expert (MoE decode): one pass over the live experts' weights
    const int rc = dtype == XM_BF16 ? launch_gemm_wsb_grouped<bf16_t>(a, w, token_count, out, max_rows, n_experts, N, K,
                                                                      gather_rows, gather_div, s)
                                    : launch_gemm_wsb_grouped<f16_t>(a, w, token_count, out, max_rows, n_experts, N, K,
                                                                     gather_rows, gather_div, s);
    if (rc != XM_ERR_UNSUPPORTED) return rc;
  }
  // 256x256 8-phase kernel behind a device-built tile table (kept in the tail of the MoE scratch); the 128x128 kernel
  // with its per-workgroup expert walk is the fallback (no scratch registered, shape outside the 8-phase envelope)
  static int p8_mode = -2;  // XLLM_MI355_GROUP_P8=0: 128x128 kernel only (product switch: the fallback's parity test), read once
  if (p8_mode == -2) p8_mode = xm_switch("XLLM_MI355_GROUP_P8", 1);
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  xm_moe_scratch(stream, &scratch, &scratch_bytes);
  const int64_t slots = (max_rows + 255) / 256 + n_experts;
  const size_t table_bytes = (size_t)(slots + 1) * 16;   // + the live-slot count behind the last slot
  // 256-row tiles pay once the experts hold rows of their own: below ~64 rows per expert the launch streams weights only
  if (p8_mode && scratch && scratch_bytes >= table_bytes + 64 && n_experts <= 1024 && max_rows >= 256 * 4 &&
      max_rows >= 64 * n_experts) {
    int32_t* table = reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(scratch) + ((scratch_bytes - table_bytes) & ~(size_t)15));
    GemmEpi e2 = epi;
    e2.group_inline = n_experts <= 256 ? 1 : 0;   // the workgroups plan for themselves (gemm_types.h: group_locate): no plan launch
    if (!e2.group_inline)
      hipLaunchKernelGGL(group_plan_kernel, dim3(1), dim3(1024), 0, s, token_count, (int)n_experts, 256, table, (int)slots);
    e2.group_tiles = table;
    e2.gather_rows = gather_rows;
    e2.gather_div = (int)gather_div;
    e2.gather_src_rows = (int)src_rows;
    const int rc = dtype == XM_BF16 ? launch_gemm_p8<kBF16>(a, w, max_rows, N, K * 2, e2, nullptr, 0, 1, s)
                                    : launch_gemm_p8<kF16>(a, w, max_rows, N, K * 2, e2, nullptr, 0, 1, s);
    if (rc != XM_ERR_UNSUPPORTED) return rc;
  }
  if (gather_rows) return XM_ERR_UNSUPPORTED;  // the caller expands with index_select and calls group_gemm
  // worst case number of 128-row tiles over all experts: every expert may waste < 1 tile
  const int m_tiles = (int)((max_rows + BM - 1) / BM + n_experts);
  const int n_tiles = (int)((N + BN - 1) / BN);
  const int ksteps = (int)((K * 2 + BKB - 1) / BKB);
  const dim3 grid((unsigned)(m_tiles * n_tiles), 1, 1);
  if (dtype == XM_BF16)
    hipLaunchKernelGGL((gemm_kernel<kBF16, false, 128, 2>), grid, dim3(256), 0, s, (const uint8_t*)a, (const uint8_t*)w,
                       (int)max_rows, (int)N, K * 2, m_tiles, n_tiles, ksteps, epi);
  else
    hipLaunchKernelGGL((gemm_kernel<kF16, false, 128, 2>), grid, dim3(256), 0, s, (const uint8_t*)a, (const uint8_t*)w,
                       (int)max_rows, (int)N, K * 2, m_tiles, n_tiles, ksteps, epi);
  return hip_check_launch();
}

int xllm_mi355_group_gemm_w8a8(const int8_t* a, int64_t a_rows, const float* a_scale, const int32_t* row_index,
                               int64_t index_div, const int8_t* w, const float* w_scale, const int32_t* token_count,
                               void* out, int64_t max_rows, int64_t n_experts, int64_t N, int64_t K, int out_dtype,
                               void* stream) {
  if (!a || !a_scale || !w || !w_scale || !token_count || !out || max_rows < 0 || n_experts <= 0 || N <= 0 || K <= 0)
    return XM_ERR_INVALID;
  if (out_dtype != XM_BF16 && out_dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (row_index && (index_div <= 0 || a_rows <= 0 || a_rows * K >= (1ll << 31))) return XM_ERR_INVALID;
  if (K % 128 != 0 || N % 8 != 0 || ((uintptr_t)a % 16) || ((uintptr_t)w % 16) || ((uintptr_t)out % 16) ||
      ((uintptr_t)w_scale % 16) || n_experts > 1024)
    return XM_ERR_UNSUPPORTED;
  if (max_rows == 0) return XM_OK;
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  xm_moe_scratch(stream, &scratch, &scratch_bytes);
  const int64_t slots = (max_rows + 255) / 256 + n_experts;
  const size_t table_bytes = (size_t)(slots + 1) * 16;   // + the live-slot count behind the last slot
  if (!scratch || scratch_bytes < table_bytes + 64) return XM_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  int32_t* table = reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(scratch) + ((scratch_bytes - table_bytes) & ~(size_t)15));
  GemmEpi epi{a_scale, max_rows, w_scale, N, nullptr, out, nullptr, out_dtype == XM_BF16, token_count, (int)n_experts};
  epi.group_inline = n_experts <= 256 ? 1 : 0;     // the workgroups plan for themselves (gemm_types.h: group_locate): no plan launch
  if (!epi.group_inline)
    hipLaunchKernelGGL(group_plan_kernel, dim3(1), dim3(1024), 0, s, token_count, (int)n_experts, 256, table, (int)slots);
  epi.group_tiles = table;
  epi.gather_rows = row_index;
  epi.gather_div = (int)index_div;
  epi.gather_src_rows = (int)a_rows;
  return launch_gemm_p8<kI8>(a, w, max_rows, N, K, epi, nullptr, 0, 1, s);
}

int xllm_mi355_group_gemm(const void* a, const void* w, const int32_t* token_count, void* out, int64_t max_rows,
                          int64_t n_experts, int64_t N, int64_t K, int dtype, void* stream) {
  return group_gemm_impl(a, w, token_count, out, max_rows, n_experts, N, K, dtype, nullptr, 0, 0, stream);
}

int xllm_mi355_group_gemm_gather(const void* a, int64_t a_rows, const int32_t* row_index, int64_t index_div, const void* w,
                                 const int32_t* token_count, void* out, int64_t max_rows, int64_t n_experts, int64_t N,
                                 int64_t K, int dtype, void* stream) {
  if (!row_index) return XM_ERR_INVALID;
  return group_gemm_impl(a, w, token_count, out, max_rows, n_experts, N, K, dtype, row_index, index_div, a_rows, stream);
}

}  // extern "C"
