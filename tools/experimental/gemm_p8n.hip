// gemm_p8n.hip -- C[M,N] = A[M,K] * W[N,K]^T for DECODE-shaped problems (M <= 256 rows per tile, small or medium N):
// block tile 256 (m) x 32*NB (n), the hand-pipelined structure of gemm_p8.hip re-cut for weight streaming.
//
// What bounds a decode GEMM on MI355X (profiles/r01_gemm_p8.txt): the weight bytes must stream from
// HBM with >= 48 KiB in flight per CU (HBM latency ~2 us), every workgroup re-reads its K range of the (L2-resident)
// activation matrix, and split-K partial sums cost 4*M*N bytes of L2 atomics per slice. A 256-wide n tile needs ~14
// K slices to fill the chip at N ~ 4k (reduction traffic dominates); a 32-wide tile re-reads the activations 144
// times. This kernel sits between: 64..128 columns, 3..7 K slices, and
//   * two independent LDS rings filled by LDS-DMA: activations 3 x 32 KiB (two tiles in flight, L2 latency) and
//     weights DW x NB*4 KiB (DW-2 tiles in flight, HBM latency) -- 160 KiB of LDS in total;
//   * the two DMA streams are issued by DIFFERENT waves (waves 0-3: activations, waves 4-7: weights). vmcnt retires
//     in order per wave, so a wave that issued both streams would have to wait for the deep weight prefetch whenever
//     it needs the next activation tile; split by role, each wave counts only its own stream
//     (vmcnt(8) / vmcnt((DW-2)*NB), never 0);
//   * wave w owns rows 32w..32w+31 x all columns (4 + 4*NB fragment reads for 4*NB MFMAs per 128-byte K step); the
//     wave groups {0-3} / {4-7} (one wave of each per SIMD) run one barrier apart, so one group's MFMAs overlap the
//     other group's LDS reads and DMA issue;
//   * fragment reads are inline-asm ds_read_b128 retired BEFORE the phase's first barrier, which makes a slot
//     reusable by the very next phase (ring depth D keeps D-1 tiles in flight);
//   * workgroup b starts its K walk at a different K step (integer sums are order independent): all workgroups read
//     the same activation slab per K step, and one slab (256 rows, stride K) touches only a few L2 channels.
// Epilogue: results are transposed through wave-private LDS so that stores / split-K atomics cover whole row segments.
#include <stdlib.h>

#include "gemm_types.h"

namespace xm {

constexpr int PN_BM = 256, PN_BK = 128, PN_THREADS = 512, PN_DA = 3;
constexpr int PN_A_TILE = PN_BM * PN_BK;  // 32 KiB

template <int KIND, int NB, int DW, bool SPLITK>
__global__ __launch_bounds__(PN_THREADS, 1) void gemm_p8n_kernel(const uint8_t* __restrict__ A,
                                                                const uint8_t* __restrict__ W, int M, int N,
                                                                int64_t Kb, int ktiles_per_split, GemmEpi epi) {
  using acc_t = typename MmaTraits<KIND>::acc_t;
  constexpr int W_TILE = NB * 32 * PN_BK;
  constexpr int W_BASE = PN_DA * PN_A_TILE;
  static_assert(W_BASE + DW * W_TILE <= 160 * 1024, "LDS budget");
  static_assert(NB == 2 || NB == 4, "fragment reads are written out for 2 or 4 column blocks");
  __shared__ __attribute__((aligned(1024))) uint8_t lds[W_BASE + DW * W_TILE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3;  // waves w and w+4 share a SIMD; grp 0 streams A, grp 1 streams W
  const int n0 = blockIdx.x * (NB * 32), m0 = blockIdx.y * PN_BM;
  const int total_kt = (int)(Kb / PN_BK);
  const int kt_begin = blockIdx.z * ktiles_per_split;
  int kt_end = kt_begin + ktiles_per_split;
  kt_end = kt_end > total_kt ? total_kt : kt_end;
  const int nk = kt_end - kt_begin;
  if (nk <= 0) return;
  const int phase = (int)((blockIdx.x * 5u + blockIdx.z * 3u) % (unsigned)nk);  // K-walk start of this workgroup

  // ---- DMA source offsets. One DMA instruction of a wave fills 8 rows x 128 B (lane-linear 1 KiB); piece i of this
  // wave covers tile rows i*32 + wq*8 + lane/8. The 16-B chunk swizzle (row>>1)&7 is applied to the source address.
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(grp == 0 ? A : W), 0, (int)((int64_t)(grp == 0 ? M : N) * Kb), 0x00020000);
  constexpr int NPIECE = 8;  // activations: 8 pieces per wave and tile; weights use the first NB
  int voff[NPIECE];
  {
    const int srow = wq * 8 + (lane >> 3);
    const int scol = ((lane & 7) ^ ((srow >> 1) & 7)) << 4;
    const int lim = (grp == 0 ? M : N) - 1, r0 = grp == 0 ? m0 : n0;
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
      int r = r0 + i * 32 + srow;
      r = r < lim ? r : lim;
      voff[i] = (int)((int64_t)r * Kb) + scol;
    }
  }
  typedef __attribute__((address_space(3))) uint8_t* lds_ptr_t;
  const lds_ptr_t lds3 = (lds_ptr_t)lds;
  const unsigned lds_base = (unsigned)(__UINTPTR_TYPE__)lds3;

  // K-walk positions (tile index within [0, nk), wrapping) of the next tile to stage for each stream, and ring slots
  int k_stage_a = phase, k_stage_w = phase;      // advanced once per staged tile
  int slot_stage_a = 0, slot_stage_w = 0;        // ring slot the next staged tile goes to
  auto advance = [&](int& k) { k = (k + 1 == nk) ? 0 : k + 1; };
  auto stage_a = [&]() {  // group 0: one 32-KiB activation tile = 8 pieces per wave
    const int soff = (kt_begin + k_stage_a) * PN_BK;
    const lds_ptr_t dst = lds3 + slot_stage_a * PN_A_TILE + wq * 1024;
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst + i * 4096, 16, voff[i], soff, 0, 0);
    advance(k_stage_a);
    slot_stage_a = (slot_stage_a + 1 == PN_DA) ? 0 : slot_stage_a + 1;
  };
  auto stage_w = [&]() {  // group 1: one NB*4-KiB weight tile = NB pieces per wave
    const int soff = (kt_begin + k_stage_w) * PN_BK;
    const lds_ptr_t dst = lds3 + W_BASE + slot_stage_w * W_TILE + wq * 1024;
#pragma unroll
    for (int i = 0; i < NB; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst + i * 4096, 16, voff[i], soff, 0, 0);
    advance(k_stage_w);
    slot_stage_w = (slot_stage_w + 1 == DW) ? 0 : slot_stage_w + 1;
  };

  // ---- fragment read offsets: lane l reads row (l & 31), logical chunk 2*kk + (l >> 5), physical = ^ ((row>>1)&7)
  unsigned rdoff[4];
  {
    const int f = ((lane & 31) >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
      rdoff[kk] = lds_base + (lane & 31) * PN_BK + (((2 * kk + (lane >> 5)) ^ f) << 4);
  }

  acc_t acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) acc[j] = MmaTraits<KIND>::zero();

  // ---- prologue. Steady state at the top of tile t: A(t) landed, A(t+1) in flight; W(t) landed, W(t+1..t+DW-2)
  // in flight.
  if (grp == 0) {
    stage_a();
    stage_a();
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  } else {
#pragma unroll
    for (int i = 0; i < DW - 1; ++i) stage_w();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DW - 2) * NB) : "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0

  int slot_a = 0, slot_w = 0;  // ring slots of the tile being computed
  u32x4 fa[4], fw[NB][4];
  for (int t = 0; t < nk; ++t) {
    // ---- L: fragment reads of tile t, DMA issue, counted waits
    const unsigned base_a = slot_a * PN_A_TILE + wave * (32 * PN_BK);
    const unsigned base_w = W_BASE + slot_w * W_TILE;
    unsigned va[4], vw[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      va[kk] = rdoff[kk] + base_a;
      vw[kk] = rdoff[kk] + base_w;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) P8_DSR(fa[kk], va[kk], 0);
#define PN_RDW(J)                                                                                          \
  P8_DSR(fw[J][0], vw[0], (J) * 4096); P8_DSR(fw[J][1], vw[1], (J) * 4096);                               \
  P8_DSR(fw[J][2], vw[2], (J) * 4096); P8_DSR(fw[J][3], vw[3], (J) * 4096);
    PN_RDW(0)
    PN_RDW(1)
    if constexpr (NB > 2) {
      PN_RDW(2)
      PN_RDW(NB - 1)
    }
#undef PN_RDW
    if (grp == 0) {
      stage_a();                                          // A(t+2) -> the slot read in the previous phase
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");    // A(t+1) has landed
    } else {
      stage_w();                                          // W(t+DW-1)
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DW - 2) * NB) : "memory");  // W(t+1) has landed
    }
    // the reads retire before the barrier: the slots of tile t may be restaged from the next phase on
    if constexpr (NB == 2) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fw[0][0]), "+v"(fw[0][1]), "+v"(fw[0][2]),
                     "+v"(fw[0][3]), "+v"(fw[1][0]), "+v"(fw[1][1]), "+v"(fw[1][2]), "+v"(fw[1][3]));
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fw[0][0]), "+v"(fw[0][1]), "+v"(fw[0][2]),
                     "+v"(fw[0][3]), "+v"(fw[1][0]), "+v"(fw[1][1]), "+v"(fw[1][2]), "+v"(fw[1][3]));
      P8_WAIT4(fw[2]);
      P8_WAIT4(fw[NB - 1]);
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- M
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = mma4<KIND>(fw[j][kk], fa[kk], acc[j]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    slot_a = (slot_a + 1 == PN_DA) ? 0 : slot_a + 1;
    slot_w = (slot_w + 1 == DW) ? 0 : slot_w + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // wrapped-around tail prefetches must land before LDS reuse
  if (grp == 0) __builtin_amdgcn_s_barrier();       // balance group 1's extra barrier
  __builtin_amdgcn_s_barrier();                     // every wave has drained its DMAs and finished its reads

  // ---- epilogue. acc[nb][r]: lane & 31 = m within the wave's 32 rows, r = 4*g + e <-> n = nb*32 + 8*g + 4*half + e
  const int half = lane >> 5, ml = lane & 31;
  const int mrow0 = m0 + wave * 32;
  uint8_t* const tbuf = lds + wave * 16384;  // wave-private transposition block (<= 16 KiB)
  if constexpr (SPLITK) {
    // [32 rows][NB*32 cols] int32, 16-B units swizzled by row & 15; one atomic instruction per 64 columns of a row
    constexpr int PITCH = NB * 128;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        i32x4_t raw = {(int)acc[nb][4 * g], (int)acc[nb][4 * g + 1], (int)acc[nb][4 * g + 2], (int)acc[nb][4 * g + 3]};
        const int u = nb * 8 + 2 * g + half;
        *reinterpret_cast<i32x4_t*>(tbuf + ml * PITCH + ((u ^ (ml & 15)) << 4)) = raw;
      }
#pragma unroll 4
    for (int r = 0; r < 32; ++r) {
      const int mr = mrow0 + r;
#pragma unroll
      for (int c0 = 0; c0 < NB * 32; c0 += 64) {
        const int col = c0 + lane;
        if (NB * 32 - c0 < 64 && col >= NB * 32) continue;
        const int v = *reinterpret_cast<const int*>(tbuf + r * PITCH + ((((col >> 2) ^ (r & 15)) << 4) | ((col & 3) << 2)));
        if (mr < M && n0 + col < N) atomicAdd(epi.acc_out + (int64_t)mr * N + n0 + col, v);
      }
    }
  } else {
    const bool has_bias = epi.bias != nullptr, out_bf16 = epi.out_bf16 != 0;
    const uint16_t* bias16 = reinterpret_cast<const uint16_t*>(epi.bias);
    const int m = mrow0 + ml, mc = m < M ? m : M - 1;
    float as = 1.0f;
    if constexpr (KIND == kI8) as = epi.a_scale ? epi.a_scale[mc] : 1.0f;
    if constexpr (KIND == kFP8) as = epi.a_scale[epi.a_scale_n > 1 ? mc : 0];
    constexpr int PITCH = NB * 64;  // bytes per 16-bit row
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        int n = n0 + nb * 32 + 8 * g + 4 * half;
        const bool n_ok = n < N;
        n = n + 3 < N ? n : N - 4;
        if constexpr (KIND == kI8) {
          if (epi.acc_out && m < M && n_ok) {
            i32x4_t raw = {acc[nb][4 * g], acc[nb][4 * g + 1], acc[nb][4 * g + 2], acc[nb][4 * g + 3]};
            *reinterpret_cast<i32x4_t*>(epi.acc_out + (int64_t)m * N + n) = raw;
          }
        }
        float wsv[4] = {1.0f, 1.0f, 1.0f, 1.0f}, bsv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (KIND == kI8) {
          if (epi.out) {
            const float4 w4 = *reinterpret_cast<const float4*>(epi.w_scale + n);
            wsv[0] = w4.x; wsv[1] = w4.y; wsv[2] = w4.z; wsv[3] = w4.w;
          }
        }
        if constexpr (KIND == kFP8) {
#pragma unroll
          for (int e = 0; e < 4; ++e) wsv[e] = epi.w_scale[epi.w_scale_n > 1 ? n + e : 0];
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
            bsv[e] = out_bf16 ? as_bf16 : as_f16;
          }
        }
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (KIND == kI8) v[e] = (float)acc[nb][4 * g + e] * as * wsv[e] + bsv[e];
          else if constexpr (KIND == kFP8) v[e] = as * (wsv[e] * acc[nb][4 * g + e]) + bsv[e];
          else v[e] = acc[nb][4 * g + e] + bsv[e];
        }
        uint2 pk;
        pk.x = pack16(v[0], out_bf16) | (pack16(v[1], out_bf16) << 16);
        pk.y = pack16(v[2], out_bf16) | (pack16(v[3], out_bf16) << 16);
        *reinterpret_cast<uint2*>(tbuf + ml * PITCH + (((nb * 4 + g) ^ (ml & 7)) << 4) + 8 * half) = pk;
      }
    if (epi.out) {
      constexpr int LPR = NB * 4;       // lanes (16 B each) per row
      constexpr int RPI = 64 / LPR;     // rows per store instruction
      const int rr = lane / LPR, uu = lane % LPR;
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int row = i * RPI + rr;
        const u32x4 row16 = *reinterpret_cast<const u32x4*>(tbuf + row * PITCH + ((uu ^ (row & 7)) << 4));
        const int mr = mrow0 + row, nn = n0 + uu * 8;
        if (mr < M && nn < N)
          *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(epi.out) + (int64_t)mr * N + nn) = row16;
      }
    }
  }
}

template <int KIND, int NB, int DW>
static int launch_p8n_cfg(const void* A, const void* W, int64_t M, int64_t N, int64_t Kb, GemmEpi epi, int splits,
                          hipStream_t s) {
  const int ktiles = (int)(Kb / PN_BK);
  splits = splits < 1 ? 1 : splits;
  const int per = (ktiles + splits - 1) / splits;
  splits = (ktiles + per - 1) / per;
  const dim3 grid((unsigned)((N + NB * 32 - 1) / (NB * 32)), (unsigned)((M + PN_BM - 1) / PN_BM), (unsigned)splits);
  if (splits > 1) {
    if constexpr (KIND == kI8) {
      if (!epi.acc_out) return XM_ERR_INVALID;
      hipLaunchKernelGGL((gemm_p8n_kernel<KIND, NB, DW, true>), grid, dim3(PN_THREADS), 0, s, (const uint8_t*)A,
                         (const uint8_t*)W, (int)M, (int)N, Kb, per, epi);
    } else {
      return XM_ERR_UNSUPPORTED;
    }
  } else {
    hipLaunchKernelGGL((gemm_p8n_kernel<KIND, NB, DW, false>), grid, dim3(PN_THREADS), 0, s, (const uint8_t*)A,
                       (const uint8_t*)W, (int)M, (int)N, Kb, per, epi);
  }
  return hip_check_launch();
}

template <int KIND>
int launch_gemm_p8n(const void* A, const void* W, int64_t M, int64_t N, int64_t Kb, GemmEpi epi, int nb, int splits,
                    hipStream_t s) {
  if (Kb % PN_BK != 0 || (N & 7) != 0 || ((uintptr_t)epi.out & 15) || M * Kb >= (1ll << 31) || N * Kb >= (1ll << 31) ||
      epi.group_counts)
    return XM_ERR_UNSUPPORTED;
  if (nb == 2) return launch_p8n_cfg<KIND, 2, 8>(A, W, M, N, Kb, epi, splits, s);
  if (nb == 4) return launch_p8n_cfg<KIND, 4, 4>(A, W, M, N, Kb, epi, splits, s);
  return XM_ERR_UNSUPPORTED;
}

template int launch_gemm_p8n<kI8>(const void*, const void*, int64_t, int64_t, int64_t, GemmEpi, int, int, hipStream_t);

}  // namespace xm
