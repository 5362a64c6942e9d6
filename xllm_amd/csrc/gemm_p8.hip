// gemm_p8.hip -- C[M,N] = A[M,K] * W[N,K]^T, 256x256 block tile, "8-phase" software pipeline for gfx950.
//
// Why a second GEMM structure: the 128x128 kernel of gemm.hip is bound by its stage -> barrier -> read -> MFMA
// dependency chain (profiles/r01_gemm_notes.txt: matrix pipe 27 % busy, no bank conflicts, not HBM bound; hipcc
// drains every in-flight LDS-DMA with s_waitcnt vmcnt(0) before the first ds_read of a K step). This kernel removes
// the drain by hand:
//   * both operands are staged HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds) in 16-KiB HALF-TILES
//     (128 rows x 128 B), four per K tile, two K tiles of LDS (128 KiB); three half-tiles stay in flight across the
//     barriers at all times, retired once per K tile by a COUNTED s_waitcnt vmcnt(6) -- never vmcnt(0);
//   * the fragment reads are inline-asm ds_read_b128 (the compiler's wait-count pass never sees an LDS load, so it
//     never inserts the drain) retired by explicit lgkmcnt waits that carry the fragment registers as operands;
//   * 8 waves = 2 groups of 4 (one wave of each group on every SIMD). The groups run one barrier apart: while one
//     group issues its 8 MFMAs of a phase (256 matrix-pipe cycles) the other issues its LDS reads and DMA requests,
//     so each SIMD's matrix pipe is fed alternately by its two waves.
// A K tile (128 bytes of K) is 4 phases, one per 64x32 quadrant of the wave's 128 (m) x 64 (n) output:
//   P1: read W(nh=0) + A(mh=0), stage, [lgkmcnt(8)] bar, MFMA (0,0), bar
//   P2: read W(nh=1),           stage,              bar, MFMA (0,1), bar
//   P3: read A(mh=1),           stage,              bar, MFMA (1,1), bar
//   P4:                         stage, vmcnt(6),    bar, MFMA (1,0), bar
// Hazards (two groups one barrier apart): a slot staged in phase q may have been read last in phase q-2, or in
// phase q-1 if those reads were retired before that phase's first barrier (P1's lgkmcnt(8) retires the W(nh=0)
// reads, whose slot P2 restages); a slot is read no earlier than the phase after the vmcnt wait that retires it.
//
// MFMA operand roles are swapped with respect to gemm.hip: the W fragment is the "row" operand, so a lane's 16
// accumulator registers are 4 groups of 4 CONSECUTIVE n for one m -> 8-byte stores in the epilogue.
// int8 results are bit-identical to gemm.hip (exact integer accumulation, same epilogue expression).
#include <stdlib.h>

#include "gemm_types.h"

#ifndef P8_FP8_K64  /* -DP8_FP8_K64=0: A/B build on the two 32x32x16 fp8 MFMAs per fragment (half the matrix rate) */
#define P8_FP8_K64 1
#endif

namespace xm {

constexpr int P8_BM = 256, P8_BN = 256, P8_BK = 128, P8_THREADS = 512;
constexpr int P8_SLOT = 128 * P8_BK;  // one half-tile: 16 KiB

// Shared epilogue of the 256x256 kernels. On entry every wave has drained its DMAs (vmcnt(0)); the function
// synchronises the workgroup before it reuses the LDS.
#ifdef P8_ABL_TIMING
__device__ long long p8_dbg[4];
#endif

template <int KIND, bool SPLITK, bool OUT_BF16>
__device__ __forceinline__ void p8_epilogue_impl(typename MmaTraits<KIND>::acc_t (&acc)[4][2], uint8_t* lds, int M,
                                                 int N, int m0, int n0, int wr, int wc, int wave, int lane, int tid,
                                                 const GemmEpi& epi) {
  // ---- epilogue (N % 8 == 0 is checked on the host). Tile acc[mb][nb]: lane & 31 = m within the block, register
  // r = 4*g + e <-> n within the block = 8*g + 4*(lane >> 5) + e: four consecutive n per g -> one 8-byte store.
#ifdef P8_ABL_NOEPI  /* ablation build: one store per lane so the accumulators stay live */
  if (epi.out) {
    int sum = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += (int)acc[i][j][r];
    if (sum == 0x12345) reinterpret_cast<int*>(epi.out)[tid] = sum;
  }
  return;
#endif
  const int half = lane >> 5, ml = lane & 31;
  const bool has_bias = epi.bias != nullptr, out_bf16 = epi.out_bf16 != 0;
  const uint16_t* bias16 = reinterpret_cast<const uint16_t*>(epi.bias);
  if constexpr (SPLITK) {
    // split-K partial sums: transpose each 32-row block through a wave-private 8-KiB LDS block ([32][64] int32,
    // 16-B units XOR-swizzled by row & 15) so that every atomic instruction adds one whole 256-B row segment
    // (lane = column) instead of 64 scattered dwords.
    __builtin_amdgcn_s_barrier();  // every wave has drained its DMAs and finished its fragment reads
    uint8_t* const tbuf = lds + wave * 8192;
    const int n_at = n0 + wc * 64 + lane;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          i32x4_t raw = {(int)acc[mb][nb][4 * g], (int)acc[mb][nb][4 * g + 1], (int)acc[mb][nb][4 * g + 2],
                         (int)acc[mb][nb][4 * g + 3]};
          *reinterpret_cast<i32x4_t*>(tbuf + ml * 256 + (((nb * 8 + 2 * g + half) ^ (ml & 15)) << 4)) = raw;
        }
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const int v = *reinterpret_cast<const int*>(tbuf + r * 256 + ((((lane >> 2) ^ (r & 15)) << 4) | ((lane & 3) << 2)));
        const int mr = m0 + wr * 128 + mb * 32 + r;
        if (mr < M && n_at < N) atomicAdd(epi.acc_out + (int64_t)mr * N + n_at, v);
      }
    }
  } else {
    // dequantise in the accumulator layout (m = lane, n = register), transpose the 16-bit results through a
    // wave-private 4-KiB LDS block per 32 rows (16-B units XOR-swizzled by row & 7: conflict-free reads, 2-way
    // writes) and store whole 128-B row segments: 8 lanes x 16 B per row, 8 rows per instruction.
    __builtin_amdgcn_s_barrier();  // every wave has drained its DMAs and finished its fragment reads
    float wsv[2][4][4], bsv[2][4][4];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int n = n0 + wc * 64 + nb * 32 + 8 * g + 4 * half;
        n = n + 3 < N ? n : N - 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          wsv[nb][g][e] = 1.0f;
          bsv[nb][g][e] = 0.0f;
        }
        if constexpr (KIND == kI8) {
          if (epi.out) {
            const float4 w4 = *reinterpret_cast<const float4*>(epi.w_scale + n);
            wsv[nb][g][0] = w4.x; wsv[nb][g][1] = w4.y; wsv[nb][g][2] = w4.z; wsv[nb][g][3] = w4.w;
          }
        }
        if constexpr (KIND == kFP8) {
#pragma unroll
          for (int e = 0; e < 4; ++e) wsv[nb][g][e] = epi.w_scale[epi.w_scale_n > 1 ? n + e : 0];
        }
        if (has_bias) {
          const uint2 bw = *reinterpret_cast<const uint2*>(bias16 + n);
          const uint16_t b16[4] = {(uint16_t)(bw.x & 0xffff), (uint16_t)(bw.x >> 16), (uint16_t)(bw.y & 0xffff),
                                   (uint16_t)(bw.y >> 16)};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f16_t hv;
            __builtin_memcpy(&hv, &b16[e], 2);
            const float as_f16 = (float)hv, as_bf16 = bf16_bits_to_f32(b16[e]);
            bsv[nb][g][e] = out_bf16 ? as_bf16 : as_f16;
          }
        }
      }
    uint8_t* const tbuf = lds + wave * 16384;
    const int rrow = lane >> 3;                                  // row within an 8-row read group
    const int rcol = ((lane & 7) ^ (rrow & 7)) << 4;             // swizzled 16-B unit of this lane's 8 columns
    const int n_st = n0 + wc * 64 + (lane & 7) * 8;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb) {
      const int m = m0 + wr * 128 + mb * 32 + ml;
      const int mc = m < M ? m : M - 1;
      float as = 1.0f;
      if constexpr (KIND == kI8) as = epi.a_scale ? epi.a_scale[mc] : 1.0f;
      if constexpr (KIND == kFP8) as = epi.a_scale[epi.a_scale_n > 1 ? mc : 0];
      if constexpr (KIND == kI8) {
        if (epi.acc_out) {  // raw accumulators requested (tests): direct 16-byte stores
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int n = n0 + wc * 64 + nb * 32 + 8 * g + 4 * half;
              if (m < M && n < N) {
                i32x4_t raw = {acc[mb][nb][4 * g], acc[mb][nb][4 * g + 1], acc[mb][nb][4 * g + 2],
                               acc[mb][nb][4 * g + 3]};
                *reinterpret_cast<i32x4_t*>(epi.acc_out + (int64_t)m * N + n) = raw;
              }
            }
        }
      }
      if (!epi.out) continue;
      uint8_t* const blk = tbuf + mb * 4096;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (KIND == kI8) v[e] = (float)acc[mb][nb][4 * g + e] * as * wsv[nb][g][e] + bsv[nb][g][e];
            else if constexpr (KIND == kFP8) v[e] = as * (wsv[nb][g][e] * acc[mb][nb][4 * g + e]) + bsv[nb][g][e];
            else v[e] = acc[mb][nb][4 * g + e] + bsv[nb][g][e];
          }
          uint2 pk;
          pk.x = pack2x16<OUT_BF16>(v[0], v[1]);
          pk.y = pack2x16<OUT_BF16>(v[2], v[3]);
          *reinterpret_cast<uint2*>(blk + ml * 128 + (((nb * 4 + g) ^ (ml & 7)) << 4) + 8 * half) = pk;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32x4 row16 = *reinterpret_cast<const u32x4*>(blk + (i * 8 + rrow) * 128 + rcol);
        const int mr = m0 + wr * 128 + mb * 32 + i * 8 + rrow;
        if (mr < M && n_st < N) {
          u32x4* const dst = reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(epi.out) + (int64_t)mr * N + n_st);
          if (epi.out_nt) __builtin_nontemporal_store(row16, dst);   // outputs beyond the L2 are streamed out (gemm_p8i.hip, round 4)
          else *dst = row16;
        }
      }
    }
  }
}

template <int KIND, bool SPLITK>
__device__ __forceinline__ void p8_epilogue(typename MmaTraits<KIND>::acc_t (&acc)[4][2], uint8_t* lds, int M, int N,
                                            int m0, int n0, int wr, int wc, int wave, int lane, int tid,
                                            const GemmEpi& epi) {
  // the output dtype is a launch constant: one uniform branch instead of a select per converted element
  if (epi.out_bf16) p8_epilogue_impl<KIND, SPLITK, true>(acc, lds, M, N, m0, n0, wr, wc, wave, lane, tid, epi);
  else p8_epilogue_impl<KIND, SPLITK, false>(acc, lds, M, N, m0, n0, wr, wc, wave, lane, tid, epi);
}

template <int KIND, bool SPLITK>
__global__ __launch_bounds__(P8_THREADS, 1) void gemm_p8_kernel(const uint8_t* __restrict__ A_in,
                                                               const uint8_t* __restrict__ W_in, int M_in, int N,
                                                               int64_t Kb, int m_tiles, int n_tiles,
                                                               int ktiles_per_split, GemmEpi epi_in) {
  using acc_t = typename MmaTraits<KIND>::acc_t;
  const uint8_t* A = A_in;
  const uint8_t* W = W_in;
  int M = M_in;
  GemmEpi epi = epi_in;
  // [K-tile buffer 2][slot 4][128 rows x 128 B]; slot 0 = W rows nh=0, 1 = A rows mh=0, 2 = W nh=1, 3 = A mh=1
  __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * 4 * P8_SLOT];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;  // waves w and w+4 share a SIMD: wr is the phase group

  // XCD-aware rasterisation: block b runs on XCD b%8; every XCD walks its own super-blocks of 32 tiles
  // (2^lm m-tiles x 2^(5-lm) n-tiles = the 32 workgroups resident on its 32 CUs), so the operands of a super-block
  // are fetched into that XCD's L2 once and re-used 4-8 times while the K loops advance together.
  int mt, nt;
  GroupSlot gslot{0, 0, 0, 0};
  if (epi.group_tiles) {   // grouped (MoE) mode: balanced over the XCDs, slot from the plan table or inline (gemm_types.h)
    if (!group_locate(epi, m_tiles, n_tiles, P8_BM, lane, mt, nt, gslot)) return;
  } else {
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int sb = j >> 5, within = j & 31;
    const int S = sb * 8 + xcd;
    const int lm = m_tiles >= 8 ? 3 : (m_tiles >= 4 ? 2 : (m_tiles >= 2 ? 1 : 0));
    const int n_sb_m = (m_tiles + (1 << lm) - 1) >> lm;
    const int SM = S % n_sb_m, SN = S / n_sb_m;
    mt = (SM << lm) + (within & ((1 << lm) - 1));
    nt = (SN << (5 - lm)) + (within >> lm);
    if (mt >= m_tiles || nt >= n_tiles) return;  // padding of the rasterised grid (whole workgroup)
  }
  int grow0 = 0;  // first sorted row of the expert (gather mode)
  if (epi.group_tiles) {
    // grouped (MoE) mode, reference dcu::group_gemm (kernels/dcu/group_gemm.cpp:25-74): rows of A are sorted by expert,
    // expert e owns rows [off, off + cnt) and weight W[e]. m-tile slot mt -> (e, off, cnt, tile inside e) from the
    // table group_plan_kernel built on the DEVICE from the expert sizes (no host sync: graph-capturable)
    const int ge = gslot.e, goff = gslot.off;
    M = gslot.cnt;
    mt = gslot.tile;
    if (!epi.gather_rows) A += (int64_t)goff * Kb;
    W += (int64_t)ge * N * Kb;
    epi.out = reinterpret_cast<uint8_t*>(epi.out) + (int64_t)goff * N * 2;
    grow0 = goff;
  }
  const int m0 = mt * P8_BM, n0 = nt * P8_BN;
  const int total_kt = (int)(Kb / P8_BK);
  const int kt_begin = blockIdx.z * ktiles_per_split;
  int kt_end = kt_begin + ktiles_per_split;
  kt_end = kt_end > total_kt ? total_kt : kt_end;
  const int nk = kt_end - kt_begin;
  if (nk <= 0) return;

  // K walk: every workgroup walks K in the same order (steps past the end re-load the last tile: every DMA is
  // unconditional, so the vmcnt arithmetic is static). Starting each workgroup at a different K tile (to spread the
  // readers of a shared operand panel over more L2 channels) was measured and is WORSE: 1033 -> 1272 us on
  // gate_up at M = 8192 -- the workgroups of a super-block re-use each other's L2 lines only while they move in step.
  auto kwalk = [&](int kt) { return kt < kt_end ? kt : kt_end - 1; };
  // ---- staging: each DMA instruction of a wave fills a lane-linear 1-KiB span = 8 rows x 128 B of a slot; the
  // XOR swizzle of the 16-B chunk index (conflict-free ds_read_b128) is applied to the per-lane SOURCE address.
  // A half-tile = 2 instructions per thread (i = 0, 1: LDS rows i*64 + wave*8 + lane/8).
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(A), 0, (int)((int64_t)(epi.gather_rows ? epi.gather_src_rows : M) * Kb), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(W), 0, (int)((int64_t)N * Kb), 0x00020000);
  int voff_a[2][2], voff_w[2][2];  // [i][half]
  {
    const int srow = wave * 8 + (lane >> 3);
    const int scol = ((lane & 7) ^ ((srow >> 1) & 7)) << 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // A slot (mh = h): LDS row i*64 + r  <->  activation row m0 + i*128 + h*64 + r   (i = reading group wr)
        int ar = m0 + i * 128 + h * 64 + srow;
        ar = ar < M ? ar : M - 1;
        if (epi.gather_rows) ar = epi.gather_rows[grow0 + ar] / epi.gather_div;  // expand fused into the staging
        voff_a[i][h] = (int)((int64_t)ar * Kb) + scol;
        // W slot (nh = h): LDS row wc*32 + c  <->  weight row n0 + wc*64 + h*32 + c, wc = (i*64 + srow) / 32
        int wrow = n0 + (i * 2 + (srow >> 5)) * 64 + h * 32 + (srow & 31);
        wrow = wrow < N ? wrow : N - 1;
        voff_w[i][h] = (int)((int64_t)wrow * Kb) + scol;
      }
  }
  typedef __attribute__((address_space(3))) uint8_t* lds_ptr_t;
  const lds_ptr_t lds3 = (lds_ptr_t)lds;
  // stage one half-tile: (buffer, slot) <- K tile kt (clamped: tail prefetches re-load the last tile, every load
  // is unconditional so the vmcnt arithmetic is static)
  auto stage = [&](int buf, int slot, int kt, bool in_loop = true) {
#ifdef P8_ABL_NOSTAGE  /* ablation build: no DMA inside the K loop (stale LDS is computed on) */
    if (in_loop) return;
#endif
#ifdef P8_ABL_NOSTAGE_W  /* ablation builds: drop only the weight / only the activation stream inside the K loop */
    if (in_loop && !(slot & 1)) return;
#endif
#ifdef P8_ABL_NOSTAGE_A
    if (in_loop && (slot & 1)) return;
#endif
    const int soff = kwalk(kt) * P8_BK;
    const lds_ptr_t dst = lds3 + (buf * 4 + slot) * P8_SLOT + wave * 1024;
    const int h = slot >> 1;
    if (slot & 1) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, dst, 16, voff_a[0][h], soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, dst + 8192, 16, voff_a[1][h], soff, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, dst, 16, voff_w[0][h], soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, dst + 8192, 16, voff_w[1][h], soff, 0, 0);
    }
  };

  // ---- fragment read addresses. MFMA 32x32 fragment: lane l holds row (l & 31), 16 K-bytes at chunk
  // 2*kk + (l >> 5) of the 128-B row; physical chunk = logical ^ ((row >> 1) & 7). One VGPR per kk and buffer.
  unsigned rd_w[2][4], rd_a[2][4];
  {
    const unsigned base = (unsigned)(__UINTPTR_TYPE__)lds3;
    const int f = ((lane & 31) >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const unsigned o = base + (lane & 31) * P8_BK + (((2 * kk + (lane >> 5)) ^ f) << 4);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        rd_w[b][kk] = o + b * 4 * P8_SLOT + wc * 32 * P8_BK;
        rd_a[b][kk] = o + b * 4 * P8_SLOT + wr * 64 * P8_BK;
      }
    }
  }

  acc_t acc[4][2];  // [m block of 32][n block of 32]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = MmaTraits<KIND>::zero();

  // ---- prologue: K tile 0 complete + three half-tiles of K tile 1 in flight
  stage(0, 0, kt_begin, false);
  stage(0, 1, kt_begin, false);
  stage(0, 2, kt_begin, false);
  stage(0, 3, kt_begin, false);
  stage(1, 0, kt_begin + 1, false);
  stage(1, 1, kt_begin + 1, false);
  stage(1, 2, kt_begin + 1, false);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0

  u32x4 fw0[4], fw1[4], fa[8];  // W fragments nh=0 / nh=1 [kk]; A fragments [mbl*4 + kk] of the current m half

#ifdef P8_ABL_NOMFMA  /* ablation build: keep the fragment reads alive, no matrix work */
#define P8_MMA(MB, NB, FW)                                                                    \
  _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) asm volatile("" ::"v"(FW[kk]), "v"(fa[kk]), "v"(fa[4 + kk])); \
  __builtin_amdgcn_sched_barrier(0);
#else
#define P8_MMA(MB, NB, FW)                                                                    \
  __builtin_amdgcn_s_setprio(1);                                                              \
  if constexpr (KIND == kFP8 && P8_FP8_K64) {                                                 \
    _Pragma("unroll") for (int kp = 0; kp < 2; ++kp) {                                        \
      acc[MB][NB] = mma_fp8x2(FW[2 * kp], FW[2 * kp + 1], fa[2 * kp], fa[2 * kp + 1], acc[MB][NB]);                 \
      acc[MB + 1][NB] = mma_fp8x2(FW[2 * kp], FW[2 * kp + 1], fa[4 + 2 * kp], fa[5 + 2 * kp], acc[MB + 1][NB]);     \
    }                                                                                         \
  } else {                                                                                    \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                        \
      acc[MB][NB] = mma4<KIND>(FW[kk], fa[kk], acc[MB][NB]);                                  \
      acc[MB + 1][NB] = mma4<KIND>(FW[kk], fa[4 + kk], acc[MB + 1][NB]);                      \
    }                                                                                         \
  }                                                                                           \
  __builtin_amdgcn_s_setprio(0);                                                              \
  __builtin_amdgcn_sched_barrier(0);
#endif

  auto ktile = [&](auto BUF_, int kt) {
    constexpr int BUF = decltype(BUF_)::value;
    // ---- P1
    P8_DSR(fw0[0], rd_w[BUF][0], 0); P8_DSR(fw0[1], rd_w[BUF][1], 0);
    P8_DSR(fw0[2], rd_w[BUF][2], 0); P8_DSR(fw0[3], rd_w[BUF][3], 0);
    P8_DSR(fa[0], rd_a[BUF][0], 1 * P8_SLOT); P8_DSR(fa[1], rd_a[BUF][1], 1 * P8_SLOT);
    P8_DSR(fa[2], rd_a[BUF][2], 1 * P8_SLOT); P8_DSR(fa[3], rd_a[BUF][3], 1 * P8_SLOT);
    P8_DSR(fa[4], rd_a[BUF][0], 1 * P8_SLOT + 32 * P8_BK); P8_DSR(fa[5], rd_a[BUF][1], 1 * P8_SLOT + 32 * P8_BK);
    P8_DSR(fa[6], rd_a[BUF][2], 1 * P8_SLOT + 32 * P8_BK); P8_DSR(fa[7], rd_a[BUF][3], 1 * P8_SLOT + 32 * P8_BK);
    stage(BUF ^ 1, 3, kt + 1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");  // the W(nh=0) reads are done: P2 may restage slot 0
    __builtin_amdgcn_s_barrier();
    P8_WAIT4(fw0);
    P8_WAIT8(fa);
    P8_MMA(0, 0, fw0)
    __builtin_amdgcn_s_barrier();
    // ---- P2
    P8_DSR(fw1[0], rd_w[BUF][0], 2 * P8_SLOT); P8_DSR(fw1[1], rd_w[BUF][1], 2 * P8_SLOT);
    P8_DSR(fw1[2], rd_w[BUF][2], 2 * P8_SLOT); P8_DSR(fw1[3], rd_w[BUF][3], 2 * P8_SLOT);
    stage(BUF, 0, kt + 2);
    __builtin_amdgcn_s_barrier();
    P8_WAIT4(fw1);
    P8_MMA(0, 1, fw1)
    __builtin_amdgcn_s_barrier();
    // ---- P3
    P8_DSR(fa[0], rd_a[BUF][0], 3 * P8_SLOT); P8_DSR(fa[1], rd_a[BUF][1], 3 * P8_SLOT);
    P8_DSR(fa[2], rd_a[BUF][2], 3 * P8_SLOT); P8_DSR(fa[3], rd_a[BUF][3], 3 * P8_SLOT);
    P8_DSR(fa[4], rd_a[BUF][0], 3 * P8_SLOT + 32 * P8_BK); P8_DSR(fa[5], rd_a[BUF][1], 3 * P8_SLOT + 32 * P8_BK);
    P8_DSR(fa[6], rd_a[BUF][2], 3 * P8_SLOT + 32 * P8_BK); P8_DSR(fa[7], rd_a[BUF][3], 3 * P8_SLOT + 32 * P8_BK);
    stage(BUF, 1, kt + 2);
    __builtin_amdgcn_s_barrier();
    P8_WAIT8(fa);
    P8_MMA(2, 1, fw1)
    __builtin_amdgcn_s_barrier();
    // ---- P4
    stage(BUF, 2, kt + 2);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // K tile kt+1 has landed; 3 half-tiles of kt+2 stay in flight
    __builtin_amdgcn_s_barrier();
    P8_MMA(2, 0, fw0)
    __builtin_amdgcn_s_barrier();
  };

#ifdef P8_ABL_TIMING  /* ablation build: shader-clock / wall-clock span of the K loop of workgroup 0 */
  const long long dbg_c0 = clock64(), dbg_w0 = wall_clock64();
#endif
  for (int t = 0; t < nk; t += 2) {
    ktile(std::integral_constant<int, 0>{}, kt_begin + t);
    if (t + 1 >= nk) break;
    ktile(std::integral_constant<int, 1>{}, kt_begin + t + 1);
  }
#ifdef P8_ABL_TIMING
  if (blockIdx.x == 0 && tid == 0) {
    p8_dbg[0] = clock64() - dbg_c0;
    p8_dbg[1] = wall_clock64() - dbg_w0;
    p8_dbg[2] = nk;
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail prefetches must land before the LDS is released
  if (wr == 0) __builtin_amdgcn_s_barrier();        // balance group 1's extra barrier
#undef P8_MMA

  p8_epilogue<KIND, SPLITK>(acc, lds, M, N, m0, n0, wr, wc, wave, lane, tid, epi);
}

int launch_gemm_p8i(const void* A, const void* W, int64_t M, int64_t N, int64_t Kb, GemmEpi epi, int m_tiles, int n_tiles,
                    int per, int splits, dim3 grid, hipStream_t s);  // gemm_p8i.hip

template <int KIND>
int launch_gemm_p8(const void* A, const void* W, int64_t M, int64_t N, int64_t Kb, GemmEpi epi, void* workspace,
                   size_t ws_bytes, int splits, hipStream_t s) {
  (void)workspace;
  (void)ws_bytes;
  // an epilogue mode the selected kernel cannot honour is declined, never silently dropped
  if (!epi_fits(epi, KIND == kI8 ? (kCapGateUp | kCapGroupTiles | kCapGather | kCapAccOut | kCapAddend)
                                 : (KIND == kFP8 ? 0u : (kCapGroupTiles | kCapGather))))
    return XM_ERR_UNSUPPORTED;
  if (Kb % P8_BK != 0 || (N & 7) != 0 || ((uintptr_t)epi.out & 15) || M * Kb >= (1ll << 31) || N * Kb >= (1ll << 31) ||
      (epi.group_counts && !epi.group_tiles))
    return XM_ERR_UNSUPPORTED;
  if (epi.group_tiles && (splits > 1 || KIND == kFP8)) return XM_ERR_UNSUPPORTED;
  if (epi.addend && (splits > 1 || epi.gate_up || epi.group_tiles || !epi.out)) return XM_ERR_UNSUPPORTED;   // plain dequant epilogue only
  // grouped: M = total rows; every expert may add one partial tile (the table has that many slots)
  const int m_tiles = (int)((M + P8_BM - 1) / P8_BM) + (epi.group_tiles ? epi.n_groups : 0);
  const int n_tiles = (int)((N + P8_BN - 1) / P8_BN);
  const int ktiles = (int)(Kb / P8_BK);
  splits = splits < 1 ? 1 : splits;
  const int per = (ktiles + splits - 1) / splits;
  splits = (ktiles + per - 1) / per;
  const int lm = m_tiles >= 8 ? 3 : (m_tiles >= 4 ? 2 : (m_tiles >= 2 ? 1 : 0));
  const int n_sb = ((m_tiles + (1 << lm) - 1) >> lm) * ((n_tiles + (1 << (5 - lm)) - 1) >> (5 - lm));
  dim3 grid((unsigned)(((n_sb + 7) / 8) * 8 * 32), 1, (unsigned)splits);
  if (epi.group_tiles)   // eight equal ranges of the live (m slot, n tile) units, sized for the worst case (see the kernels)
    grid.x = (unsigned)(((m_tiles * n_tiles + 7) / 8) * 8);
  if constexpr (KIND == kI8) {
    // int8 lives in the 16x16x64 specialisation (gemm_p8i.hip): the only 8-phase kernel with the gate_up epilogue, the grouped
    // mode and split-K. (The 32x32x32 int8 arm of this file lost its round-1 A/B and left the library in round 4.)
    if (splits > 1 && !epi.acc_out) return XM_ERR_INVALID;
    return launch_gemm_p8i(A, W, M, N, Kb, epi, m_tiles, n_tiles, per, splits, grid, s);
  } else {
    if (splits > 1) return XM_ERR_UNSUPPORTED;
    epi.out_nt = (M * N * 2 > (48ll << 20)) ? 1 : 0;
    hipLaunchKernelGGL((gemm_p8_kernel<KIND, false>), grid, dim3(P8_THREADS), 0, s, (const uint8_t*)A,
                       (const uint8_t*)W, (int)M, (int)N, Kb, m_tiles, n_tiles, per, epi);
  }
  return hip_check_launch();
}

template int launch_gemm_p8<kI8>(const void*, const void*, int64_t, int64_t, int64_t, GemmEpi, void*, size_t, int,
                                 hipStream_t);
template int launch_gemm_p8<kFP8>(const void*, const void*, int64_t, int64_t, int64_t, GemmEpi, void*, size_t, int,
                                  hipStream_t);
template int launch_gemm_p8<kBF16>(const void*, const void*, int64_t, int64_t, int64_t, GemmEpi, void*, size_t, int,
                                   hipStream_t);
template int launch_gemm_p8<kF16>(const void*, const void*, int64_t, int64_t, int64_t, GemmEpi, void*, size_t, int,
                                  hipStream_t);

}  // namespace xm

#ifdef P8_ABL_TIMING
extern "C" __attribute__((visibility("default"))) int xllm_mi355_debug_p8(long long* out4) {
  return hipMemcpyFromSymbol(out4, HIP_SYMBOL(xm::p8_dbg), 4 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif
