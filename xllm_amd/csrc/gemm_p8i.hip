// gemm_p8i.hip -- the int8 specialisation of the 256x256 8-phase kernel (gemm_p8.hip) on v_mfma_i32_16x16x64_i8.
//
// Why a second MFMA shape: the 8-phase int8 kernel issues an MFMA in 94.7 % of the ideal slots, but the shader clock
// it is granted under that load is only 1.05-1.16 GHz (profiles/r01_gemm_p8_timing.txt): it is power-bound. Per MAC
// the 16x16x64 shape moves 20 % fewer accumulator registers through the register file than 32x32x32 (C/D = 4 + 4
// registers per 16384 MACs instead of 16 + 16 per 32768) at the same operand traffic; with the same MAC count the
// clock rises to 1.25 GHz at M = 8192 / 1.65 GHz at M = 256 and the wall time per K step drops by 10-12 % even though
// the shape needs 17 instead of 16 cycles per instruction (measured with a mock before this file was written).
//
// Everything but the fragment geometry is gemm_p8_kernel's: half-tile LDS-DMA staging, counted vmcnt(6), inline-asm
// ds_read_b128, two wave groups one barrier apart, four phases per K tile (hazard rules: header of gemm_p8.hip).
//   fragment: lane l <-> row (l & 15), 16 K-bytes at chunk 4*kb + (l >> 4) of the 128-B row (kb = 64-byte K half);
//   a 64 (m) x 32 (n) quadrant = 4 x 2 tiles of 16 x 16, 2 K halves: 16 MFMAs per phase, 8 + 4 fragment reads;
//   D layout with the W fragment as the row operand: lane & 15 = m, register r <-> n = 4*(lane >> 4) + r.
#include <stdlib.h>

#include "gemm_types.h"

// Output stores of the 8-phase int8 kernel: when the [M, N] output is larger than the L2 (prefill: [8192, 37888] bf16 = 620 MB,
// read once by the next operator) they are NON-TEMPORAL (GemmEpi::out_nt, set by the launcher) so that they do not evict the
// operand panels the super-block re-uses. Round 4, GEMM_DIST=gauss tools/gemm_bench.py 8192 int8, alternating libraries on one box:
// gate_up 911 -> 884 us (2.44 -> 2.52 POP/s), qkv 127.3 -> 123.8, o 89.5 -> 88.4, down 414.4 -> 413.5 (profiles/r04_p8i_nt.txt).
// Decode-sized outputs (a few MB, consumed from the L2 by the next kernel) keep ordinary stores.
#define P8I_STORE(PTR, VAL)                                          \
  do {                                                               \
    if (epi.out_nt) __builtin_nontemporal_store((VAL), (PTR));       \
    else *(PTR) = (VAL);                                             \
  } while (0)

namespace xm {

constexpr int P8_BM = 256, P8_BN = 256, P8_BK = 128, P8_THREADS = 512;
constexpr int P8_SLOT = 128 * P8_BK;  // one half-tile: 16 KiB

// epilogue of the 16x16 accumulator layout: acc[mb][nb] (mb = 16-row block 0..7, nb = 16-column block 0..3)
template <bool SPLITK, bool OUT_BF16, bool ADDEND>
__device__ __forceinline__ void p8i_epilogue(i32x4_t (&acc)[8][4], uint8_t* lds, int M, int N, int m0, int n0, int wr,
                                             int wc, int wave, int lane, const GemmEpi& epi) {
  const int g4 = lane >> 4, ml = lane & 15;
  __builtin_amdgcn_s_barrier();  // every wave has drained its DMAs and finished its fragment reads
  if constexpr (SPLITK) {
    // [32 rows][64 cols] int32 per pair of m blocks, 16-B units swizzled by row & 15; one atomic instruction adds a
    // whole 256-B row segment (lane = column)
    uint8_t* const tbuf = lds + wave * 8192;
    const int n_at = n0 + wc * 64 + lane;
#pragma unroll
    for (int mp = 0; mp < 4; ++mp) {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int row = h * 16 + ml;
          *reinterpret_cast<i32x4_t*>(tbuf + row * 256 + (((nb * 4 + g4) ^ (row & 15)) << 4)) = acc[mp * 2 + h][nb];
        }
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        const int v = *reinterpret_cast<const int*>(tbuf + r * 256 + ((((lane >> 2) ^ (r & 15)) << 4) | ((lane & 3) << 2)));
        const int mr = m0 + wr * 128 + mp * 32 + r;
        if (mr < M && n_at < N) atomicAdd(epi.acc_out + (int64_t)mr * N + n_at, v);
      }
    }
  } else {
    const bool has_bias = epi.bias != nullptr;
    const uint16_t* bias16 = reinterpret_cast<const uint16_t*>(epi.bias);
    // GemmEpi::addend: the lane's 16 row segments are requested up front (a load issued next to its store put one memory
    // latency per segment on the epilogue's critical path: prefill chunk 55.97 -> 56.75 ms, tools/prefill_ab.py)
    u32x4 cadd[ADDEND ? 4 : 1][4];
    if constexpr (ADDEND) {
      const int n_ld = n0 + wc * 64 + (lane & 7) * 8;
#pragma unroll
      for (int mp = 0; mp < 4; ++mp)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int mr = m0 + wr * 128 + mp * 32 + i * 8 + (lane >> 3);
          cadd[mp][i] = u32x4{0u, 0u, 0u, 0u};
          if (mr < M && n_ld < N)
            cadd[mp][i] = __builtin_nontemporal_load(
                reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(epi.addend) + (int64_t)mr * N + n_ld));
        }
    }
    float wsv[4][4], bsv[4][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      int n = n0 + wc * 64 + nb * 16 + 4 * g4;
      n = n + 3 < N ? n : N - 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { wsv[nb][e] = 1.0f; bsv[nb][e] = 0.0f; }
      if (epi.out) {
        const float4 w4 = *reinterpret_cast<const float4*>(epi.w_scale + n);
        wsv[nb][0] = w4.x; wsv[nb][1] = w4.y; wsv[nb][2] = w4.z; wsv[nb][3] = w4.w;
      }
      if (has_bias) {
        const uint2 bw = *reinterpret_cast<const uint2*>(bias16 + n);
        const uint16_t b16[4] = {(uint16_t)(bw.x & 0xffff), (uint16_t)(bw.x >> 16), (uint16_t)(bw.y & 0xffff),
                                 (uint16_t)(bw.y >> 16)};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f16_t hv;
          __builtin_memcpy(&hv, &b16[e], 2);
          bsv[nb][e] = OUT_BF16 ? bf16_bits_to_f32(b16[e]) : (float)hv;
        }
      }
    }
    uint8_t* const tbuf = lds + wave * 16384;
    const int rrow = lane >> 3;
    const int rcol = ((lane & 7) ^ (rrow & 7)) << 4;
    const int n_st = n0 + wc * 64 + (lane & 7) * 8;
#pragma unroll
    for (int mp = 0; mp < 4; ++mp) {  // pairs of 16-row blocks = 32 rows = one 4-KiB transposition block
      uint8_t* const blk = tbuf + mp * 4096;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int mb = mp * 2 + h, row = h * 16 + ml;
        const int m = m0 + wr * 128 + mb * 16 + ml;
        const int mc = m < M ? m : M - 1;
        const float as = epi.a_scale ? epi.a_scale[epi.gather_rows ? epi.gather_rows[mc] / epi.gather_div : mc] : 1.0f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          if (epi.acc_out) {  // raw accumulators requested (tests)
            const int n = n0 + wc * 64 + nb * 16 + 4 * g4;
            if (m < M && n < N) *reinterpret_cast<i32x4_t*>(epi.acc_out + (int64_t)m * N + n) = acc[mb][nb];
          }
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (float)acc[mb][nb][e] * as * wsv[nb][e] + bsv[nb][e];
          uint2 pk;
          pk.x = pack2x16<OUT_BF16>(v[0], v[1]);
          pk.y = pack2x16<OUT_BF16>(v[2], v[3]);
          // n within the wave's 64 columns = nb*16 + 4*g4 (+e): 16-B unit nb*2 + (g4 >> 1), 8-B half g4 & 1
          *reinterpret_cast<uint2*>(blk + row * 128 + (((nb * 2 + (g4 >> 1)) ^ (row & 7)) << 4) + 8 * (g4 & 1)) = pk;
        }
      }
      if (!epi.out) continue;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        u32x4 row16 = *reinterpret_cast<const u32x4*>(blk + (i * 8 + rrow) * 128 + rcol);
        const int mr = m0 + wr * 128 + mp * 32 + i * 8 + rrow;
        if (mr < M && n_st < N) {
          if constexpr (ADDEND) {   // GemmEpi::addend: rT(y + c) on the 16-bit result y (8 columns of one row per lane)
            const u32x4 c16 = cadd[mp][i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const uint32_t yw = row16[e], cw = c16[e];
              auto cvt = [](uint32_t h) -> float {
                if constexpr (OUT_BF16) return bf16_bits_to_f32((uint16_t)h);
                else { const uint16_t b = (uint16_t)h; f16_t hv; __builtin_memcpy(&hv, &b, 2); return (float)hv; }
              };
              const float y0 = cvt(yw & 0xffffu), y1 = cvt(yw >> 16), c0 = cvt(cw & 0xffffu), c1 = cvt(cw >> 16);
              row16[e] = pack2x16<OUT_BF16>(y0 + c0, y1 + c1);
            }
          }
          P8I_STORE(reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(epi.out) + (int64_t)mr * N + n_st), row16);
        }
      }
    }
  }
}

// gate_up projection with the SiLU.mul of DenseMLP fused (GemmEpi::gate_up, round 3; prefill shapes). Tile nt covers ACT columns
// [nt * 128, + 128): its W slots nh = 0 hold the gate rows of those columns, nh = 1 the up rows (rows I + ...), so a wave's
// accumulator blocks nb = 0, 1 are the gate values and nb = 2, 3 the up values of the SAME (row, column) positions: the
// activation is lane-local. rT(acc * a_scale * w_scale + bias) for both, act = rT(rT(silu(g)) * u) (act_and_mul_i8_reg_kernel's
// expression), transposed through wave-private LDS for 16-byte row-segment stores; the row |max| of the per-token int8
// quantisation that follows goes through a per-row LDS maximum into GemmEpi::row_amax (atomic max on float bits).
template <bool OUT_BF16>
__device__ __forceinline__ void p8i_epilogue_gate_up(i32x4_t (&acc)[8][4], uint8_t* lds, int M, int N, int m0, int nt, int wr,
                                                     int wc, int wave, int lane, int tid, const GemmEpi& epi) {
  using T = typename std::conditional<OUT_BF16, bf16_t, f16_t>::type;
  constexpr int PITCH = 80;                      // 32 act columns x 2 B + 16
  const int g4 = lane >> 4, ml = lane & 15;
  const int64_t I = N / 2;
  __builtin_amdgcn_s_barrier();                  // every wave has drained its DMAs and finished its fragment reads
  unsigned* const rowmax = reinterpret_cast<unsigned*>(lds + 8 * 128 * PITCH);
  if (tid < 256) rowmax[tid] = 0u;
  __builtin_amdgcn_s_barrier();
  uint8_t* const tb = lds + wave * (128 * PITCH);
  const int ncol0 = nt * 128 + wc * 32;          // first act column of the wave
  float wg[2][4], wu[2][4], bg[2][4], bu[2][4];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int n = ncol0 + nb * 16 + 4 * g4;
    const float4 a4 = *reinterpret_cast<const float4*>(epi.w_scale + n);
    const float4 b4 = *reinterpret_cast<const float4*>(epi.w_scale + I + n);
    wg[nb][0] = a4.x; wg[nb][1] = a4.y; wg[nb][2] = a4.z; wg[nb][3] = a4.w;
    wu[nb][0] = b4.x; wu[nb][1] = b4.y; wu[nb][2] = b4.z; wu[nb][3] = b4.w;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bg[nb][e] = epi.bias ? load16(epi.bias, n + e, OUT_BF16) : 0.0f;
      bu[nb][e] = epi.bias ? load16(epi.bias, I + n + e, OUT_BF16) : 0.0f;
    }
  }
#pragma unroll
  for (int mb = 0; mb < 8; ++mb) {
    const int m = m0 + wr * 128 + mb * 16 + ml;
    const float as = epi.a_scale[m < M ? m : M - 1];
    float amax = 0.0f;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      float r[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float g = r16<T>((float)acc[mb][nb][e] * as * wg[nb][e] + bg[nb][e]);
        const float u = r16<T>((float)acc[mb][nb + 2][e] * as * wu[nb][e] + bu[nb][e]);
        r[e] = r16<T>(r16<T>(act_f<XM_ACT_SILU>(g)) * u);
        amax = fmaxf(amax, fabsf(r[e]));
      }
      uint2 pk;
      pk.x = pack2x16<OUT_BF16>(r[0], r[1]);
      pk.y = pack2x16<OUT_BF16>(r[2], r[3]);
      *reinterpret_cast<uint2*>(tb + (mb * 16 + ml) * PITCH + nb * 32 + g4 * 8) = pk;
    }
    if (m < M) atomicMax(&rowmax[wr * 128 + mb * 16 + ml], __float_as_uint(amax));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own writes (in-order LDS queue): visible to its reads below
#pragma unroll
  for (int i = 0; i < 8; ++i) {                  // 128 rows x 4 chunks of 8 columns over 64 lanes
    const int idx = i * 64 + lane, row = idx >> 2, c = idx & 3;
    const uint4 v = *reinterpret_cast<const uint4*>(tb + row * PITCH + c * 16);
    const int m = m0 + wr * 128 + row;
    if (m < M) {
      const u32x4 nv = {v.x, v.y, v.z, v.w};
      P8I_STORE(reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(epi.act_out) + (int64_t)m * I + ncol0 + c * 8), nv);
    }
  }
  __builtin_amdgcn_s_barrier();
  if (tid < 256) {
    const int m = m0 + tid;
    if (m < M && rowmax[tid]) atomicMax(reinterpret_cast<unsigned*>(epi.row_amax) + m, rowmax[tid]);
  }
}

// ADDEND: GemmEpi::addend in the dequant epilogue -- its own instantiation, so the plain kernel's code is untouched (with a run-time
// branch the plain gate_up GEMM at M = 8192 measured +0.45 %: profiles/r04_gemm_addend.txt)
#ifdef P8I_TIMING  /* timing build (tools/p8i_timing.py): wall-clock (100 MHz) and shader-clock spans of one workgroup's phases */
__device__ long long p8i_dbg[16];
#endif
template <bool SPLITK, bool ADDEND = false>
__global__ __launch_bounds__(P8_THREADS, 1) void gemm_p8i_kernel(const uint8_t* __restrict__ A_in,
                                                                const uint8_t* __restrict__ W_in, int M_in, int N,
                                                                int64_t Kb, int m_tiles, int n_tiles,
                                                                int ktiles_per_split, GemmEpi epi_in, int stagger) {
  const uint8_t* A = A_in;
  const uint8_t* W = W_in;
  int M = M_in;
  GemmEpi epi = epi_in;
  // [K-tile buffer 2][slot 4][128 rows x 128 B]; slot 0 = W rows nh=0, 1 = A rows mh=0, 2 = W nh=1, 3 = A mh=1
  __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * 4 * P8_SLOT];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;  // waves w and w+4 share a SIMD: wr is the phase group
#ifdef P8I_TIMING
  const long long tm_w0 = wall_clock64(), tm_c0 = clock64();
#endif

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
  if (epi.group_tiles) {
    // grouped (MoE) W8A8: see gemm_p8_kernel. Expert e owns sorted rows [off, off + cnt), weight W[e] and the weight
    // scales w_scale[e * N ..]; the per-token activation scales follow the rows (a_scale[off + m], or -- with the
    // expand fused in -- a_scale[gather_rows[off + m] / gather_div] of the un-expanded activations)
    const int ge = gslot.e, goff = gslot.off;
    M = gslot.cnt;
    mt = gslot.tile;
    if (epi.gather_rows) epi.gather_rows += goff;  // row r of the expert -> source row gather_rows[r] / gather_div
    else { A += (int64_t)goff * Kb; epi.a_scale += goff; }
    W += (int64_t)ge * N * Kb;
    epi.w_scale += (int64_t)ge * N;
    epi.out = reinterpret_cast<uint8_t*>(epi.out) + (int64_t)goff * N * 2;
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
  // Exception (round 3): ONE m tile (decode, M <= 256) -- the weight panels are private to their workgroups, nothing is
  // re-used between them, and the only shared operand is the small activation slab of the K tile, which the address hash
  // puts on a few L2 channels: there each workgroup starts at another tile and wraps around (exact int32 sums: any order).
  const int phase = (stagger && m_tiles == 1 && !epi.group_tiles) ? (int)((unsigned)nt % (unsigned)nk) : 0;
  auto kwalk = [&](int kt) {
    int r = (kt < kt_end ? kt : kt_end - 1) - kt_begin + phase;
    r = r >= nk ? r - nk : r;
    return kt_begin + r;
  };
  // ---- staging: each DMA instruction of a wave fills a lane-linear 1-KiB span = 8 rows x 128 B of a slot; the
  // XOR swizzle of the 16-B chunk index (conflict-free ds_read_b128) is applied to the per-lane SOURCE address.
  // A half-tile = 2 instructions per thread (i = 0, 1: LDS rows i*64 + wave*8 + lane/8).
  const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(A), 0, (int)((int64_t)(epi.gather_rows ? epi.gather_src_rows : M) * Kb), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(W), 0, (int)((int64_t)N * Kb), 0x00020000);
  int voff_a[2][2], voff_w[2][2];  // [i][half]
  {
#ifdef P8I_INTERLEAVE
    // EXPERIMENT (-DP8I_INTERLEAVE, bit-exact, SLOWER: gate_up M = 8192 1157 vs 910 us, M = 256 65 vs 50 us): 8-row-interleaved
    // slot layout [row >> 3][16-B chunk][row & 7]. ds_read_b128 runs at full rate only when 8 consecutive lanes read inside
    // one aligned 128-B block (tools/lds_pattern_bench.hip: 5.3 vs 8 cycles per fragment read), so the 8 rows of a chunk
    // are stored next to each other; but the DMA writes LDS lane-linearly, hence lane l must FETCH row (l & 7), chunk
    // (l >> 3) of its 8-row group -- a 16-byte gather over 8 rows per 8 lanes that the load path serves far slower than
    // the row-contiguous fetch, which costs more than the reads gain.
    const int srow = wave * 8 + (lane & 7);
    const int scol = (lane >> 3) << 4;
#else
    const int srow = wave * 8 + (lane >> 3);
    const int scol = ((lane & 7) ^ ((srow >> 1) & 7)) << 4;
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // A slot (mh = h): LDS row i*64 + r  <->  activation row m0 + i*128 + h*64 + r   (i = reading group wr)
        int ar = m0 + i * 128 + h * 64 + srow;
        ar = ar < M ? ar : M - 1;
        if (epi.gather_rows) ar = epi.gather_rows[ar] / epi.gather_div;  // expand fused into the staging
        voff_a[i][h] = (int)((int64_t)ar * Kb) + scol;
        // W slot (nh = h): LDS row wc*32 + c  <->  weight row n0 + wc*64 + h*32 + c, wc = (i*64 + srow) / 32
        int wrow = n0 + (i * 2 + (srow >> 5)) * 64 + h * 32 + (srow & 31);
        if (epi.gate_up)   // slot nh = 0: gate rows of act columns nt*128 + wc*32 + c; nh = 1: their up rows (I = N / 2 further)
          wrow = h * (N / 2) + nt * 128 + (i * 2 + (srow >> 5)) * 32 + (srow & 31);
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

  // ---- fragment read addresses: lane l reads row (l & 15) of a 16-row block, logical chunk 4*kb + (l >> 4),
  // physical chunk = logical ^ ((row >> 1) & 7) (16-row block offsets do not change the swizzle term)
  unsigned rd_w[2][2], rd_a[2][2];  // [buffer][kb]
  {
    const unsigned base = (unsigned)(__UINTPTR_TYPE__)lds3;
    const int f = (lane & 15) >> 1;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#ifdef P8I_INTERLEAVE
      const unsigned o = base + ((lane & 15) >> 3) * 1024 + ((4 * kb + (lane >> 4)) << 7) + ((lane & 7) << 4) + 0 * f;
#else
      const unsigned o = base + (lane & 15) * P8_BK + (((4 * kb + (lane >> 4)) ^ f) << 4);
#endif
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        rd_w[b][kb] = o + b * 4 * P8_SLOT + wc * 32 * P8_BK;
        rd_a[b][kb] = o + b * 4 * P8_SLOT + wr * 64 * P8_BK;
      }
    }
  }

  i32x4_t acc[8][4];  // [16-row m block][16-column n block]
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = i32x4_t{0, 0, 0, 0};

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

  u32x4 fw0[4], fw1[4], fa[8];  // W fragments [nb*2 + kb] of nh = 0 / 1; A fragments [mb*2 + kb] of the current m half

  // quadrant (MH, NH): m blocks MH*4 .. +3, n blocks NH*2 .. +1; 16 MFMAs, every accumulator touched twice 8 apart
#define P8I_MMA(MH, NH, FW)                                                                          \
  __builtin_amdgcn_s_setprio(1);                                                                     \
  _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                   \
  _Pragma("unroll") for (int mb = 0; mb < 4; ++mb)                                                   \
  _Pragma("unroll") for (int nb = 0; nb < 2; ++nb) {                                                 \
    const i32x4_t av = __builtin_bit_cast(i32x4_t, FW[nb * 2 + kb]);                                 \
    const i32x4_t bv = __builtin_bit_cast(i32x4_t, fa[mb * 2 + kb]);                                 \
    acc[(MH) * 4 + mb][(NH) * 2 + nb] =                                                              \
        __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, acc[(MH) * 4 + mb][(NH) * 2 + nb], 0, 0, 0);   \
  }                                                                                                  \
  __builtin_amdgcn_s_setprio(0);                                                                     \
  __builtin_amdgcn_sched_barrier(0);
#define P8I_RD_W(FW, BUF, SLOT)                                                                      \
  P8_DSR(FW[0], rd_w[BUF][0], (SLOT)); P8_DSR(FW[1], rd_w[BUF][1], (SLOT));                          \
  P8_DSR(FW[2], rd_w[BUF][0], (SLOT) + 16 * P8_BK); P8_DSR(FW[3], rd_w[BUF][1], (SLOT) + 16 * P8_BK);
#define P8I_RD_A(BUF, SLOT)                                                                          \
  P8_DSR(fa[0], rd_a[BUF][0], (SLOT)); P8_DSR(fa[1], rd_a[BUF][1], (SLOT));                          \
  P8_DSR(fa[2], rd_a[BUF][0], (SLOT) + 16 * P8_BK); P8_DSR(fa[3], rd_a[BUF][1], (SLOT) + 16 * P8_BK); \
  P8_DSR(fa[4], rd_a[BUF][0], (SLOT) + 32 * P8_BK); P8_DSR(fa[5], rd_a[BUF][1], (SLOT) + 32 * P8_BK); \
  P8_DSR(fa[6], rd_a[BUF][0], (SLOT) + 48 * P8_BK); P8_DSR(fa[7], rd_a[BUF][1], (SLOT) + 48 * P8_BK);

  auto ktile = [&](auto BUF_, int kt) {
    constexpr int BUF = decltype(BUF_)::value;
    // ---- P1: W(nh=0) first (retired by lgkmcnt(8): P2 may restage that slot), then A(mh=0)
    P8I_RD_W(fw0, BUF, 0)
    P8I_RD_A(BUF, 1 * P8_SLOT)
    stage(BUF ^ 1, 3, kt + 1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    P8_WAIT4(fw0);
    P8_WAIT8(fa);
    P8I_MMA(0, 0, fw0)
    __builtin_amdgcn_s_barrier();
    // ---- P2: W(nh=1)
    P8I_RD_W(fw1, BUF, 2 * P8_SLOT)
    stage(BUF, 0, kt + 2);
    __builtin_amdgcn_s_barrier();
    P8_WAIT4(fw1);
    P8I_MMA(0, 1, fw1)
    __builtin_amdgcn_s_barrier();
    // ---- P3: A(mh=1)
    P8I_RD_A(BUF, 3 * P8_SLOT)
    stage(BUF, 1, kt + 2);
    __builtin_amdgcn_s_barrier();
    P8_WAIT8(fa);
    P8I_MMA(1, 1, fw1)
    __builtin_amdgcn_s_barrier();
    // ---- P4
    stage(BUF, 2, kt + 2);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // K tile kt+1 has landed; 3 half-tiles of kt+2 stay in flight
    __builtin_amdgcn_s_barrier();
    P8I_MMA(1, 0, fw0)
    __builtin_amdgcn_s_barrier();
  };

#ifdef P8I_TIMING
  const long long tm_w1 = wall_clock64(), tm_c1 = clock64();
#endif
  for (int t = 0; t < nk; t += 2) {
    ktile(std::integral_constant<int, 0>{}, kt_begin + t);
    if (t + 1 >= nk) break;
    ktile(std::integral_constant<int, 1>{}, kt_begin + t + 1);
  }
#ifdef P8I_TIMING
  const long long tm_w2 = wall_clock64(), tm_c2 = clock64();
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tail prefetches must land before the LDS is released
  if (wr == 0) __builtin_amdgcn_s_barrier();        // balance group 1's extra barrier
#undef P8I_MMA
#undef P8I_RD_W
#undef P8I_RD_A
  if constexpr (!SPLITK) {
    if (epi.gate_up) {
      if (epi.out_bf16) p8i_epilogue_gate_up<true>(acc, lds, M, N, m0, nt, wr, wc, wave, lane, tid, epi);
      else p8i_epilogue_gate_up<false>(acc, lds, M, N, m0, nt, wr, wc, wave, lane, tid, epi);
      return;
    }
  }
  if (epi.out_bf16) p8i_epilogue<SPLITK, true, ADDEND>(acc, lds, M, N, m0, n0, wr, wc, wave, lane, epi);
  else p8i_epilogue<SPLITK, false, ADDEND>(acc, lds, M, N, m0, n0, wr, wc, wave, lane, epi);
#ifdef P8I_TIMING
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the epilogue's stores have been accepted
    const long long tm_w3 = wall_clock64(), tm_c3 = clock64();
    if (blockIdx.x == (gridDim.x / 2 & ~7u) + 3 && tid == 0) {
      p8i_dbg[0] = tm_w1 - tm_w0; p8i_dbg[1] = tm_w2 - tm_w1; p8i_dbg[2] = tm_w3 - tm_w2;
      p8i_dbg[3] = tm_c1 - tm_c0; p8i_dbg[4] = tm_c2 - tm_c1; p8i_dbg[5] = tm_c3 - tm_c2;
      p8i_dbg[6] = nk;
    }
  }
#endif
}

// same grid / envelope as launch_gemm_p8 (gemm_p8.hip decides which of the two int8 kernels runs)
XM_TUNE_VAR(p8i_kstagger, "XLLM_MI355_KSTAGGER", 1);   // K-walk stagger (round-3 A/B winner; 0 = tuning arm)
int launch_gemm_p8i(const void* A, const void* W, int64_t M, int64_t N, int64_t Kb, GemmEpi epi, int m_tiles, int n_tiles,
                    int per, int splits, dim3 grid, hipStream_t s) {
  // 16-bit output (or the fused act [M, N / 2]) beyond the 32 MB of L2: streamed out (see P8I_STORE)
  epi.out_nt = (M * (epi.gate_up ? N / 2 : N) * 2 > (48ll << 20)) ? 1 : 0;
  if (splits > 1)
    hipLaunchKernelGGL((gemm_p8i_kernel<true>), grid, dim3(P8_THREADS), 0, s, (const uint8_t*)A, (const uint8_t*)W, (int)M,
                       (int)N, Kb, m_tiles, n_tiles, per, epi, p8i_kstagger);
  else if (epi.addend)
    hipLaunchKernelGGL((gemm_p8i_kernel<false, true>), grid, dim3(P8_THREADS), 0, s, (const uint8_t*)A, (const uint8_t*)W,
                       (int)M, (int)N, Kb, m_tiles, n_tiles, per, epi, p8i_kstagger);
  else
    hipLaunchKernelGGL((gemm_p8i_kernel<false>), grid, dim3(P8_THREADS), 0, s, (const uint8_t*)A, (const uint8_t*)W,
                       (int)M, (int)N, Kb, m_tiles, n_tiles, per, epi, p8i_kstagger);
  return hip_check_launch();
}

}  // namespace xm

#ifdef P8I_TIMING
extern "C" __attribute__((visibility("default"))) int xllm_mi355_debug_p8i(long long* out16) {
  return hipMemcpyFromSymbol(out16, HIP_SYMBOL(xm::p8i_dbg), 16 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif
