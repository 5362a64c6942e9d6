// gemm_ws.hip -- weight-stream int8 GEMM for decode-shaped problems (M <= 512 rows): out[M,N] = A[M,K] . W[N,K]^T
// (kernel::scaled_matmul, kernels/dcu/scaled_matmul.cpp:103-300; exact int32 sums, the dequant epilogue of gemm_p8i.hip).
//
// Why another kernel (round-2 measurements, profiles/r02_gemm_ws.txt): a decode GEMM moves every weight byte exactly once,
// a CU sustains only ~20-25 GB/s of an HBM stream, so the time of such a GEMM is set by HOW MANY CUs pull weights and
// how deep their request queues are -- the 256 x 256 tiles of the 8-phase kernel put gate_up (N = 37888) on 148 of the 256
// CUs (47 us at M = 32 and at M = 256 alike), the skinny kernel re-reads the activations from every workgroup.
//
// Design:
//  * the weights are PRE-PACKED once (xllm_mi355_pack_weight_i8, at weight-load time) in MFMA-fragment order:
//      [n16 group g][K tile kt (128 B)][k step ks (64 B)][lane l (64)][16 B],  lane l = row g*16 + (l & 15),
//      bytes kt*128 + ks*64 + (l >> 4)*16 .. +16  -- one fragment = 1 KiB contiguous, a group = K*16 B contiguous;
//  * a workgroup (4 waves, one per SIMD) owns (WM*MB*16 rows) x (WN*NG*16 columns) x one K slice. The W fragments of a K
//    tile go HBM -> LDS by LDS-DMA, 1 KiB per instruction, source AND destination contiguous (a ring of AD + 1 slots, AD
//    tiles in flight), and are read back lane-linearly with ds_read_b128 (full LDS rate, no bank conflicts, no swizzle);
//  * the activations never touch the LDS: wave (wm, wn) loads the fragments of ITS OWN MB*16 rows straight from global /
//    L2 into a register ring of the same depth (buffer_load_dwordx4, rows past M read as zeros);
//  * v_mfma_i32_16x16x64_i8 with W as the row operand: D[n][m], lane & 15 = m, register r = n = 4*(lane >> 4) + r;
//  * the column width of a tile is any even number of 16-column groups, so N = 37888 becomes 237 tiles of 160 columns
//    (all CUs busy) instead of 148 of 256; few-column problems take K slices, whose exact int32 partial sums go to
//    separate slabs with plain stores (no atomics, nothing to zero) and are summed by the consumer;
//  * one barrier per K tile; LDS-DMA and activation loads retire in order on vmcnt, so both run AD tiles ahead.
//
// Round 3:
//  * EIGHT-wave workgroups (4 x 2 waves, two per SIMD) for 128 < M <= 512. The four-wave 256-row tile keeps ONE wave on each
//    SIMD, so the 13 LDS-DMA pieces a wave issues per K tile (60-180 issue cycles each, MI355X_MICROARCH.md), its 28 fragment
//    reads and its 80 MFMAs all serialise in one instruction stream: 2.86 us per K tile against 1280 matrix-pipe cycles
//    (profiles/r02_gemm_ws.txt). With two waves per SIMD each wave carries half of every kind of work and one wave's DMA
//    issue / lgkmcnt waits sit beside its partner's MFMAs. Fragments that do not divide over 8 waves are padded with
//    out-of-range pieces (no memory request, zeros into a 1-KiB dump area behind the ring).
//  * KIND = kFP8 (e4m3, fp8_scaled_matmul semantics, linear.cpp:137-182): the SAME packed byte layout -- the two 16-byte
//    k-step fragments of a lane are the two halves of the 32-byte operand of v_mfma_f32_16x16x128_f8f6f4 (any K permutation
//    shared by both operands is a valid contraction order) -- fp32 accumulators, K slices through fp32 slabs summed in
//    slice order (deterministic).
#include <stdlib.h>

#include "gemm_types.h"

namespace xm {

constexpr int WS_BK = 128;     // K tile in bytes
constexpr int WS_FRAG = 1024;  // one 16 x 64-byte fragment
#ifndef WS_W_AUX
#define WS_W_AUX 2  // cache policy of the weight stream: nt (every byte is read once)
#endif

// ------------------------------------------------------------------------------------------------ weight packing
__global__ __launch_bounds__(256) void pack_weight_i8_kernel(const uint8_t* __restrict__ W, uint8_t* __restrict__ Wp,
                                                             int64_t N, int64_t K) {
  const int64_t chunks_per_row = K / 16, total = N * chunks_per_row;
  const int64_t KT = K / WS_BK;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / chunks_per_row, c = i - n * chunks_per_row;
    const int64_t g = n >> 4, r = n & 15, kt = c >> 3, ks = (c >> 2) & 1, kq = c & 3;
    const uint4 v = *reinterpret_cast<const uint4*>(W + n * K + c * 16);
    *reinterpret_cast<uint4*>(Wp + ((g * KT + kt) * 2 + ks) * WS_FRAG + (kq * 16 + r) * 16) = v;
  }
}

#define WS_BUFLOAD(DST, VOFF, RSRC, SOFF, IMM)                                             \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4"                        \
               : "=v"(DST) : "v"(VOFF), "s"(RSRC), "s"(SOFF), "n"(IMM) : "memory")

// ---- pieces of the K loop as plain device functions (hipcc drops the host stub of a kernel whose lambdas nest or capture
// arrays of template-dependent size, so the kernel body below uses no lambda at all)
typedef __attribute__((address_space(3))) uint8_t* ws_lds_ptr_t;

// The accumulators are pinned to AGPRs and updated in place by an asm MFMA: with the builtin, hipcc (512-register budget,
// AGPR form) gives vdst and srcC different registers across unrolled tile bodies and copies ~2 accumulator registers
// per MFMA back and forth inside the loop. Hazards the compiler can no longer see: srcC == vdst back-to-back is forwarded by
// the hardware (and every accumulator is touched once per MB * NG MFMAs here); the MFMA -> VALU read of the epilogue gets
// explicit s_nops after the loop; operands come from waited ds_reads.
#ifdef WS_ABL_NOMFMA
#define WS_MFMA(ACC, W_, A_) asm volatile("" : "+a"(ACC) : "v"(W_), "v"(A_))
#else
#define WS_MFMA(ACC, W_, A_) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(ACC) : "v"(W_), "v"(A_))
#endif
#define WS_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))

// the LDS-DMA instructions a wave issues for ONE K tile (NDA activation + NDW weight pieces of 1 KiB), handed to the compute
// routine so that it can spread them between its MFMA groups: issuing a piece costs the wave 60-180 cycles (MI355X_MICROARCH.md),
// 13 pieces in a row in front of the MFMAs of a 256-row tile were as long as the MFMAs themselves (measured: the tile time was
// their SUM); between MFMA groups the matrix pipe keeps running underneath
struct WsDma {
  __amdgpu_buffer_rsrc_t rsrc_a, rsrc_w;
  ws_lds_ptr_t dst_a, dst_w;  // this wave's first piece in the destination slot
  int so_a, so_w;             // scalar offsets of the K tile
  ws_lds_ptr_t dump;          // 1-KiB area behind the ring: destination of a wave's padding pieces (eight-wave tiles whose
  int nvalid_w;               // weight fragments do not divide over the waves); pieces >= nvalid_w are out of range
  int w_nt;                   // cache policy of the weight stream: nt when ONE workgroup reads a column range's weights (one m
                              // tile), the default policy when the m tiles of a column range share them through their XCD's L2
                              // (wave-uniform: a scalar branch; round 3, M = 256 on 128-row tiles: qkv 19.4 -> 17.2, o 19.7 -> 17.9 us)
};
// WNT: the weight stream's cache policy as a compile-time constant (1 = nt, 0 = default) or -1 = the run-time flag d.w_nt. The
// staggered eight-wave kernel passes a constant: with the run-time flag every DMA issue of its matrix phase sat behind a scalar
// branch (14 s_cbranch per K tile, each taken branch an instruction-fetch bubble between MFMAs; probe at the end of round 4:
// gate_up + SiLU.mul at M = 256 52.0 -> 49.7-50.6 us, profiles/r04_next_round_probes.txt)
template <int NDA, int NDW, int WNT = -1>
__device__ __forceinline__ void ws_dma_piece(const WsDma& d, const int (&voff_a)[8], const int (&voff_w)[8], int j) {
  if (j < NDA) __builtin_amdgcn_raw_ptr_buffer_load_lds(d.rsrc_a, d.dst_a + j * WS_FRAG, 16, voff_a[j], d.so_a, 0, 0);
  else if (WNT == 1 || (WNT < 0 && d.w_nt))
    __builtin_amdgcn_raw_ptr_buffer_load_lds(d.rsrc_w, (j - NDA) < d.nvalid_w ? d.dst_w + (j - NDA) * WS_FRAG : d.dump, 16,
                                             voff_w[j - NDA], d.so_w, 0, WS_W_AUX);
  else
    __builtin_amdgcn_raw_ptr_buffer_load_lds(d.rsrc_w, (j - NDA) < d.nvalid_w ? d.dst_w + (j - NDA) * WS_FRAG : d.dump, 16,
                                             voff_w[j - NDA], d.so_w, 0, 0);
}
// pieces due after MFMA group `slot` of 2 * NG (an even spread of ND pieces over the groups)
template <int NDA, int NDW, int NG>
__device__ __forceinline__ void ws_dma_slot(const WsDma& d, const int (&voff_a)[8], const int (&voff_w)[8], int slot) {
  constexpr int ND = NDA + NDW;
  const int lo = slot * ND / (2 * NG), hi = (slot + 1) * ND / (2 * NG);
#pragma unroll
  for (int j = lo; j < hi; ++j) ws_dma_piece<NDA, NDW>(d, voff_a, voff_w, j);
}

// pieces [LO, HI) of the tile, the part due after MFMA group `slot` of `nslots` (staggered kernel: a tile's pieces are split
// between a read phase and a matrix phase)
template <int NDA, int NDW, int LO, int HI, int WNT = -1>
__device__ __forceinline__ void ws_dma_range(const WsDma& d, const int (&voff_a)[8], const int (&voff_w)[8], int slot, int nslots) {
  const int lo = LO + slot * (HI - LO) / nslots, hi = LO + (slot + 1) * (HI - LO) / nslots;
#pragma unroll
  for (int j = lo; j < hi; ++j) ws_dma_piece<NDA, NDW, WNT>(d, voff_a, voff_w, j);
}

// k step 1 of a tile, group by group (compile-time recursion: the wait counts are immediates)
template <int MB, int NG, int NDA, int NDW, int NG_LEFT>
__device__ __forceinline__ void ws_kstep1(i32x4_t (&acc)[MB][NG], u32x4 (&fw)[NG], const u32x4 (&a1)[MB], const WsDma& d,
                                          const int (&voff_a)[8], const int (&voff_w)[8]) {
  if constexpr (NG_LEFT > 0) {
    constexpr int ng = NG - NG_LEFT;
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fw[ng]) : "n"(NG_LEFT - 1));
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) WS_MFMA(acc[mb][ng], fw[ng], a1[mb]);
    ws_dma_slot<NDA, NDW, NG>(d, voff_a, voff_w, NG + ng);
    ws_kstep1<MB, NG, NDA, NDW, NG_LEFT - 1>(acc, fw, a1, d, voff_a, voff_w);
  }
}
// The MFMAs of one K tile. `rw`: this lane's address of the wave's first W fragment in the LDS slot (lane-linear 1-KiB
// blocks, [group][k step]); `ra0` / `ra1`: its address in the wave's first row block of the row-major, swizzled activation
// tile for k step 0 / 1. LDS operations retire in order; they are issued as
//   A0_0..A0_{MB-1}, A1_0..A1_{MB-1}, W0_0..W0_{NG-1}, then W1_ng right behind the k-step-0 MFMAs of group ng (into the SAME
//   register: its data returns tens of cycles after those MFMAs have read their operands).
//   group (0, ng) needs W0_ng: NG-1-ng later W0s + ng W1s = NG - 1 outstanding; group (1, ng) needs W1_ng: NG - 1 - ng.
// (LDS-DMA pieces count on vmcnt, not lgkmcnt: interleaving them does not disturb these counts.)
template <int MB, int NG, int NDA, int NDW>
__device__ __forceinline__ void ws_compute(i32x4_t (&acc)[MB][NG], unsigned rw, unsigned ra0, unsigned ra1, const WsDma& d,
                                           const int (&voff_a)[8], const int (&voff_w)[8]) {
  u32x4 a0[MB], a1[MB], fw[NG];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) WS_DSR(a0[mb], ra0, mb * 16 * WS_BK);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) WS_DSR(a1[mb], ra1, mb * 16 * WS_BK);
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) WS_DSR(fw[ng], rw, ng * 2 * WS_FRAG);
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    if (ng == 0) {
      if constexpr (MB == 4)
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(fw[0]) : "n"(NG - 1));
      else
        asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a0[0]), "+v"(a0[1]), "+v"(fw[0]) : "n"(NG - 1));
    } else {
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(fw[ng]) : "n"(NG - 1));
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) WS_MFMA(acc[mb][ng], fw[ng], a0[mb]);
    WS_DSR(fw[ng], rw, ng * 2 * WS_FRAG + WS_FRAG);
    ws_dma_slot<NDA, NDW, NG>(d, voff_a, voff_w, ng);
  }
  if constexpr (MB == 4) asm volatile("" : "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]));  // (older than every W read)
  else asm volatile("" : "+v"(a1[0]), "+v"(a1[1]));
  ws_kstep1<MB, NG, NDA, NDW, NG>(acc, fw, a1, d, voff_a, voff_w);
  __builtin_amdgcn_sched_barrier(0);
}


// ---- fp8 (e4m3): one v_mfma_f32_16x16x128_f8f6f4 per (row block, column group) and K tile; its 32-byte operands are the
// (k step 0, k step 1) fragment pairs. All fragment reads of the tile are issued up front (LDS operations retire in order):
//   A0_0..A0_{MB-1}, A1_0..A1_{MB-1}, then (W0_ng, W1_ng) for ng = 0..NG-1; group ng needs its pair: 2 * (NG - 1 - ng) younger
//   reads may still be outstanding. The DMA pieces of the next tile are spread over the NG groups.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4_t ws_mma_fp8(const u32x4 w0, const u32x4 w1, const u32x4 a0, const u32x4 a1, f32x4_t c) {
  const gi32x8_t wv = {(int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w, (int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w};
  const gi32x8_t av = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
  return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wv, av, c, 0, 0, 0, 0, 0, 0);
}
template <int NDA, int NDW, int NSLOTS>
__device__ __forceinline__ void ws_dma_slot_n(const WsDma& d, const int (&voff_a)[8], const int (&voff_w)[8], int slot) {
  constexpr int ND = NDA + NDW;
  const int lo = slot * ND / NSLOTS, hi = (slot + 1) * ND / NSLOTS;
#pragma unroll
  for (int j = lo; j < hi; ++j) ws_dma_piece<NDA, NDW>(d, voff_a, voff_w, j);
}
template <int MB, int NG, int NDA, int NDW, int NG_LEFT>
__device__ __forceinline__ void ws_fp8_groups(f32x4_t (&acc)[MB][NG], u32x4 (&fw0)[NG], u32x4 (&fw1)[NG], const u32x4 (&a0)[MB],
                                              const u32x4 (&a1)[MB], const WsDma& d, const int (&voff_a)[8],
                                              const int (&voff_w)[8]) {
  if constexpr (NG_LEFT > 0) {
    constexpr int ng = NG - NG_LEFT;
    // (lgkmcnt is a 4-bit counter: a clamped count waits for a few more reads than needed, never fewer)
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fw0[ng]), "+v"(fw1[ng]) : "n"(2 * (NG_LEFT - 1) > 15 ? 15 : 2 * (NG_LEFT - 1)));
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb][ng] = ws_mma_fp8(fw0[ng], fw1[ng], a0[mb], a1[mb], acc[mb][ng]);
    ws_dma_slot_n<NDA, NDW, NG>(d, voff_a, voff_w, ng);
    ws_fp8_groups<MB, NG, NDA, NDW, NG_LEFT - 1>(acc, fw0, fw1, a0, a1, d, voff_a, voff_w);
  }
}
template <int MB, int NG, int NDA, int NDW>
__device__ __forceinline__ void ws_compute_fp8(f32x4_t (&acc)[MB][NG], unsigned rw, unsigned ra0, unsigned ra1, const WsDma& d,
                                               const int (&voff_a)[8], const int (&voff_w)[8]) {
  u32x4 a0[MB], a1[MB], fw0[NG], fw1[NG];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) WS_DSR(a0[mb], ra0, mb * 16 * WS_BK);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) WS_DSR(a1[mb], ra1, mb * 16 * WS_BK);
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    WS_DSR(fw0[ng], rw, ng * 2 * WS_FRAG);
    WS_DSR(fw1[ng], rw, ng * 2 * WS_FRAG + WS_FRAG);
  }
  // the activation fragments are older than every weight read: the first group's wait covers them
  if constexpr (MB == 4)
    asm volatile("" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]));
  else
    asm volatile("" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a1[0]), "+v"(a1[1]));
  ws_fp8_groups<MB, NG, NDA, NDW, NG>(acc, fw0, fw1, a0, a1, d, voff_a, voff_w);
  __builtin_amdgcn_sched_barrier(0);
}

// ---- 16-bit weights (bf16 / f16, kernel::matmul = F::linear, kernels/dcu/matmul.cpp:20-25) on the same packed BYTE layout: a
// K tile of 128 bytes is 64 elements, a k-step fragment (16 B per lane = 8 consecutive elements of one row) is the operand of
// v_mfma_f32_16x16x32_{bf16,f16}; two MFMAs per (row block, column group) and K tile, like the int8 form. fp32 accumulation,
// K tiles in order, K slices summed in slice order (deterministic); out = r16(acc + bias).
template <int KIND>
__device__ __forceinline__ f32x4_t ws_mma_h16(const u32x4 w, const u32x4 a, f32x4_t c) {
  if constexpr (KIND == kBF16)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gbf16x8_t, w), __builtin_bit_cast(gbf16x8_t, a), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(gf16x8_t, w), __builtin_bit_cast(gf16x8_t, a), c, 0, 0, 0);
}
template <int KIND, int MB, int NG, int NDA, int NDW, int NG_LEFT>
__device__ __forceinline__ void ws_h16_groups(f32x4_t (&acc)[MB][NG], u32x4 (&fw0)[NG], u32x4 (&fw1)[NG], const u32x4 (&a0)[MB],
                                              const u32x4 (&a1)[MB], const WsDma& d, const int (&voff_a)[8],
                                              const int (&voff_w)[8]) {
  if constexpr (NG_LEFT > 0) {
    constexpr int ng = NG - NG_LEFT;
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fw0[ng]), "+v"(fw1[ng]) : "n"(2 * (NG_LEFT - 1) > 15 ? 15 : 2 * (NG_LEFT - 1)));
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb][ng] = ws_mma_h16<KIND>(fw0[ng], a0[mb], acc[mb][ng]);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb][ng] = ws_mma_h16<KIND>(fw1[ng], a1[mb], acc[mb][ng]);
    ws_dma_slot_n<NDA, NDW, NG>(d, voff_a, voff_w, ng);
    ws_h16_groups<KIND, MB, NG, NDA, NDW, NG_LEFT - 1>(acc, fw0, fw1, a0, a1, d, voff_a, voff_w);
  }
}
template <int KIND, int MB, int NG, int NDA, int NDW>
__device__ __forceinline__ void ws_compute_h16(f32x4_t (&acc)[MB][NG], unsigned rw, unsigned ra0, unsigned ra1, const WsDma& d,
                                               const int (&voff_a)[8], const int (&voff_w)[8]) {
  u32x4 a0[MB], a1[MB], fw0[NG], fw1[NG];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) WS_DSR(a0[mb], ra0, mb * 16 * WS_BK);
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) WS_DSR(a1[mb], ra1, mb * 16 * WS_BK);
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    WS_DSR(fw0[ng], rw, ng * 2 * WS_FRAG);
    WS_DSR(fw1[ng], rw, ng * 2 * WS_FRAG + WS_FRAG);
  }
  if constexpr (MB == 4)
    asm volatile("" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]));
  else
    asm volatile("" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a1[0]), "+v"(a1[1]));
  ws_h16_groups<KIND, MB, NG, NDA, NDW, NG>(acc, fw0, fw1, a0, a1, d, voff_a, voff_w);
  __builtin_amdgcn_sched_barrier(0);
}

template <int KIND>
struct WsAcc { using type = i32x4_t; };
template <>
struct WsAcc<kFP8> { using type = f32x4_t; };
template <>
struct WsAcc<kBF16> { using type = f32x4_t; };
template <>
struct WsAcc<kF16> { using type = f32x4_t; };

// ---- epilogue shared by the weight-stream kernels. Accumulator layout: lane & 15 = m inside the row block, registers = four
// consecutive n. Round 3 (in-kernel timing, profiles/r03_gemm_ws.txt): storing that layout directly -- 8-byte pieces, 16 rows x
// 32 B per instruction -- cost gate_up's workgroups 20 k of their 81 k cycles at M = 256 (store-issue bound). The wave now
// transposes its tile through a private LDS block (the ring is free: every DMA has landed and a barrier has passed) and stores
// whole row segments, 16 B per lane, NG * 32 B (16-bit results) or NG * 64 B (slabs) contiguous per row.
// K slices > 1: exact int32 (fp8: fp32) partial sums of this slice -> its slab; otherwise the dequant epilogue of
// scaled_matmul / fp8_scaled_matmul (and the raw sums when epi.acc_out is set: tests).
template <int KIND, int MB, int NG>
__device__ __forceinline__ void ws_epilogue(typename WsAcc<KIND>::type (&acc)[MB][NG], const GemmEpi& epi,
                                            int32_t* __restrict__ slabs, int slice, int n_slices, int M, int N, int m_base,
                                            int g0, int g_live, int wn, int lane, uint8_t* lds, int wave, int part_slot = 0) {
  using acc_t = typename WsAcc<KIND>::type;
  const int g4 = lane >> 4, ml = lane & 15;
  const int gw = wn * NG;  // first column group of the wave inside the workgroup tile
  if (n_slices > 1) {
    // one 16-row block at a time: [16 rows][NG * 64 B + 16] in LDS, read back as NG 16-byte chunks per lane
    constexpr int ROWB = NG * 64, PITCH = ROWB + 16, CH = ROWB / 16;
    uint8_t* const tb = lds + wave * (16 * PITCH);
    int32_t* const slab = slabs + (int64_t)slice * M * N;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int ng = 0; ng < NG; ++ng) *reinterpret_cast<acc_t*>(tb + ml * PITCH + ng * 64 + g4 * 16) = acc[mb][ng];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const int idx = i * 64 + lane, row = idx / CH, c = idx % CH;
        const uint4 v = *reinterpret_cast<const uint4*>(tb + row * PITCH + c * 16);
        const int m = m_base + mb * 16 + row;
        if (m < M && gw + (c >> 2) < g_live) {
          uint4* const dst = reinterpret_cast<uint4*>(slab + (int64_t)m * N + (g0 + gw) * 16 + c * 4);
          // non-temporal: the slabs are read once by the fused consumer and dead afterwards; as ordinary stores they sit DIRTY in
          // the L2 (up to 19 MB after the qkv GEMM) and are written back while the decode-attention kernel streams the KV cache
          // (tools/attn_dirty_l2.py: +10 us on that launch after 17 MB of dirty lines)
          if (epi.slab_nt) {
            typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
            const nt_u32x4 nv = {v.x, v.y, v.z, v.w};
            __builtin_nontemporal_store(nv, reinterpret_cast<nt_u32x4*>(dst));
          } else {
            *dst = v;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the block is rewritten by the next row block
    }
    return;
  }
  const bool has_bias = epi.bias != nullptr, out_bf16 = epi.out_bf16 != 0;
  const uint16_t* bias16 = reinterpret_cast<const uint16_t*>(epi.bias);
  const bool wide = epi.out && (((uintptr_t)epi.out & 15) == 0);  // (N % 16 == 0: every row segment is then 16-byte aligned)
  const bool amax = (KIND == kBF16 || KIND == kF16) && epi.argmax_val != nullptr;   // fused greedy sampling (GemmEpi::argmax_*)
  float best_v[MB];
  int best_i[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) { best_v[mb] = -__builtin_inff(); best_i[mb] = 0x7fffffff; }
  constexpr int ROWB = NG * 32, PITCH = ROWB + 16, CH = ROWB / 16;  // 16-bit tile of the wave: [MB * 16 rows][NG * 32 B + 16]
  uint8_t* const tb = lds + wave * (MB * 16 * PITCH);
  float asv[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = m_base + mb * 16 + ml, mc = m < M ? m : M - 1;
    if constexpr (KIND == kI8) asv[mb] = epi.out ? epi.a_scale[mc] : 1.0f;
    else if constexpr (KIND == kFP8) asv[mb] = epi.out ? epi.a_scale[epi.a_scale_n > 1 ? mc : 0] : 1.0f;
    else asv[mb] = 1.0f;
  }
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    const int g = g0 + gw + ng;
    const bool live = gw + ng < g_live;
    const int n = (live ? g : g0) * 16 + 4 * g4;  // (dead groups compute on a valid column and are never stored)
    float wsv[4] = {1.f, 1.f, 1.f, 1.f}, bsv[4] = {0.f, 0.f, 0.f, 0.f};
    if (epi.out) {
      if constexpr (KIND == kI8) {
        const float4 w4 = *reinterpret_cast<const float4*>(epi.w_scale + n);
        wsv[0] = w4.x; wsv[1] = w4.y; wsv[2] = w4.z; wsv[3] = w4.w;
      } else if constexpr (KIND == kFP8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) wsv[e] = epi.w_scale[epi.w_scale_n > 1 ? n + e : 0];
      }
    }
    if (has_bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bsv[e] = load16(bias16, n + e, out_bf16);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int m = m_base + mb * 16 + ml;
      if (epi.acc_out && live && m < M)
        *reinterpret_cast<acc_t*>(epi.acc_out + (int64_t)m * N + n) = acc[mb][ng];  // raw sums (tests, one slab)
      if (!epi.out && !amax) continue;
      float v[4];
      if constexpr (KIND == kI8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (float)acc[mb][ng][e] * asv[mb] * wsv[e] + bsv[e];
      } else if constexpr (KIND == kFP8) {  // the fp8 epilogue of gemm_p8.hip: as * (ws * acc) + bias
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = asv[mb] * (wsv[e] * acc[mb][ng][e]) + bsv[e];
      } else {                              // 16-bit linear: acc + bias (gemm_p8.hip)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mb][ng][e] + bsv[e];
      }
      uint2 pk;
      if (out_bf16) { pk.x = pack2x16<true>(v[0], v[1]); pk.y = pack2x16<true>(v[2], v[3]); }
      else { pk.x = pack2x16<false>(v[0], v[1]); pk.y = pack2x16<false>(v[2], v[3]); }
      if constexpr (KIND == kBF16 || KIND == kF16) {
        if (amax && live) {   // the values argmax sees are the 16-bit logits the plain epilogue would have stored
          const unsigned h[4] = {pk.x & 0xffffu, pk.x >> 16, pk.y & 0xffffu, pk.y >> 16};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float lv;
            if (out_bf16) lv = bf16_bits_to_f32(h[e]);
            else { const uint16_t hb = (uint16_t)h[e]; f16_t hv; __builtin_memcpy(&hv, &hb, 2); lv = (float)hv; }
            if (n + e < N && argmax_better(lv, n + e, best_v[mb], best_i[mb])) { best_v[mb] = lv; best_i[mb] = n + e; }
          }
        }
      }
      if (!epi.out) continue;
      if (wide) *reinterpret_cast<uint2*>(tb + (mb * 16 + ml) * PITCH + ng * 32 + g4 * 8) = pk;
      else if (live && m < M) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(epi.out) + (int64_t)m * N + n) = pk;
    }
  }
  if constexpr (KIND == kBF16 || KIND == kF16) {
    if (amax) {   // the four lanes (g4) of a row hold disjoint columns: two xor steps, then one pair per (row, wave)
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
          const float ov = __shfl_xor(best_v[mb], o);
          const int oi = __shfl_xor(best_i[mb], o);
          if (argmax_better(ov, oi, best_v[mb], best_i[mb])) { best_v[mb] = ov; best_i[mb] = oi; }
        }
        const int m = m_base + mb * 16 + ml;
        // [slot][M] pairs: the 16 rows of a block are one contiguous 128-byte store (a [M][slot] layout made every pair its
        // own partial cache line: 974 k scattered 4-byte stores at lm_head's shape, as much write traffic as the logits)
        if (g4 == 0 && m < M)
          reinterpret_cast<uint2*>(epi.argmax_val)[(int64_t)part_slot * M + m] = make_uint2(__float_as_uint(best_v[mb]), (unsigned)best_i[mb]);
      }
    }
  }
  if (!wide) return;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (a wave's LDS operations execute in order; this retires the writes)
#pragma unroll
  for (int i = 0; i < MB * NG / 2; ++i) {  // MB * 16 rows x CH chunks over 64 lanes
    const int idx = i * 64 + lane, row = idx / CH, c = idx % CH;
    const uint4 v = *reinterpret_cast<const uint4*>(tb + row * PITCH + c * 16);
    const int m = m_base + row;
    if (m < M && gw + (c >> 1) < g_live)
      *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(epi.out) + (int64_t)m * N + (g0 + gw) * 16 + c * 8) = v;
  }
}

// ---- gate_up epilogue with SiLU.mul fused (GemmEpi::gate_up; int8 and the 16-bit kinds). Slot groups [0, G / 2) of the workgroup tile hold the
// GATE columns of act groups ga0 .. ga0 + ga_live, slots [G / 2, G) the UP columns of the same act groups. Phase 1: every wave
// dequantises its accumulators exactly like the plain epilogue (rT(acc * a_scale * w_scale + bias)) into ONE workgroup-shared
// 16-bit tile in the LDS; phase 2: every thread takes (row, 8 act columns) items, reads the gate and the up chunk, computes
// rT(rT(silu(g)) * u) -- act_and_mul_i8_reg_kernel's expression, bit for bit -- stores 16 bytes of act and folds the |max|
// into a per-row LDS maximum, which leaves the workgroup as one atomic max per row.
template <int KIND, int MB, int NG, int WM, int WN, int NWV>
__device__ __forceinline__ void ws_epilogue_gate_up(typename WsAcc<KIND>::type (&acc)[MB][NG], const GemmEpi& epi, int M, int N,
                                                    int m_tile0, int wm, int wn, int lane, int tid, int ga0, int ga_live,
                                                    int n_groups_act, uint8_t* lds) {
  constexpr int G = WN * NG, GH = G / 2, ROWS = WM * MB * 16, PITCH = G * 32 + 16;
  static_assert(G % 2 == 0 && ROWS * PITCH + ROWS * 4 <= 160 * 1024, "gate_up tile");
  const int g4 = lane >> 4, ml = lane & 15;
  const bool out_bf16 = epi.out_bf16 != 0, has_bias = epi.bias != nullptr;
  unsigned* const rowmax = reinterpret_cast<unsigned*>(lds + ROWS * PITCH);
  for (int r = tid; r < ROWS; r += NWV * 64) rowmax[r] = 0u;
  float asv[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m = m_tile0 + wm * MB * 16 + mb * 16 + ml;
    asv[mb] = KIND == kI8 ? epi.a_scale[m < M ? m : M - 1] : 1.0f;
  }
#pragma unroll
  for (int ng = 0; ng < NG; ++ng) {
    const int q = wn * NG + ng, half = q / GH, j = q % GH;
    const int jl = j < ga_live ? j : 0;                                     // (dead slots compute on a valid column, never read)
    const int n = (half * n_groups_act + ga0 + jl) * 16 + 4 * g4;
    float wsv[4] = {1.f, 1.f, 1.f, 1.f};
    if constexpr (KIND == kI8) {
      const float4 w4 = *reinterpret_cast<const float4*>(epi.w_scale + n);
      wsv[0] = w4.x; wsv[1] = w4.y; wsv[2] = w4.z; wsv[3] = w4.w;
    }
    float bsv[4] = {0.f, 0.f, 0.f, 0.f};
    if (has_bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bsv[e] = load16(epi.bias, n + e, out_bf16);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      float v[4];
      if constexpr (KIND == kI8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (float)acc[mb][ng][e] * asv[mb] * wsv[e] + bsv[e];
      } else {                                // 16-bit linear: acc + bias (ws_epilogue)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mb][ng][e] + bsv[e];
      }
      uint2 pk;
      if (out_bf16) { pk.x = pack2x16<true>(v[0], v[1]); pk.y = pack2x16<true>(v[2], v[3]); }
      else { pk.x = pack2x16<false>(v[0], v[1]); pk.y = pack2x16<false>(v[2], v[3]); }
      *reinterpret_cast<uint2*>(lds + (wm * MB * 16 + mb * 16 + ml) * PITCH + q * 32 + g4 * 8) = pk;
    }
  }
  __syncthreads();
  const int64_t I = (int64_t)n_groups_act * 16;
  for (int item = tid; item < ROWS * G; item += NWV * 64) {
    const int row = item / G, c = item % G, j = c >> 1, hc = c & 1;
    const int m = m_tile0 + row;
    if (j >= ga_live || m >= M) continue;
    const uint4 gv = *reinterpret_cast<const uint4*>(lds + row * PITCH + j * 32 + hc * 16);
    const uint4 uv = *reinterpret_cast<const uint4*>(lds + row * PITCH + (GH + j) * 32 + hc * 16);
    const uint32_t gw[4] = {gv.x, gv.y, gv.z, gv.w}, uw[4] = {uv.x, uv.y, uv.z, uv.w};
    uint32_t ow[4];
    float amax = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      float r0, r1;
      if (out_bf16) {
        r0 = r16<bf16_t>(r16<bf16_t>(act_f<XM_ACT_SILU>(bf16_bits_to_f32(gw[w] & 0xffffu))) * bf16_bits_to_f32(uw[w] & 0xffffu));
        r1 = r16<bf16_t>(r16<bf16_t>(act_f<XM_ACT_SILU>(bf16_bits_to_f32(gw[w] >> 16))) * bf16_bits_to_f32(uw[w] >> 16));
        ow[w] = f32_to_bf16_bits(r0) | ((uint32_t)f32_to_bf16_bits(r1) << 16);
      } else {
        uint16_t g0h = (uint16_t)(gw[w] & 0xffffu), g1h = (uint16_t)(gw[w] >> 16), u0h = (uint16_t)(uw[w] & 0xffffu), u1h = (uint16_t)(uw[w] >> 16);
        f16_t g0f, g1f, u0f, u1f;
        __builtin_memcpy(&g0f, &g0h, 2); __builtin_memcpy(&g1f, &g1h, 2); __builtin_memcpy(&u0f, &u0h, 2); __builtin_memcpy(&u1f, &u1h, 2);
        r0 = r16<f16_t>(r16<f16_t>(act_f<XM_ACT_SILU>((float)g0f)) * (float)u0f);
        r1 = r16<f16_t>(r16<f16_t>(act_f<XM_ACT_SILU>((float)g1f)) * (float)u1f);
        const f16_t o0 = (f16_t)r0, o1 = (f16_t)r1;
        uint16_t o0h, o1h;
        __builtin_memcpy(&o0h, &o0, 2); __builtin_memcpy(&o1h, &o1, 2);
        ow[w] = (uint32_t)o0h | ((uint32_t)o1h << 16);
      }
      amax = fmaxf(amax, fmaxf(fabsf(r0), fabsf(r1)));
    }
    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(epi.act_out) + (int64_t)m * I + (int64_t)(ga0 + j) * 16 + hc * 8) =
        make_uint4(ow[0], ow[1], ow[2], ow[3]);
#ifndef WS_ABL_GU_NOLDSATOMIC
    if (epi.row_amax) atomicMax(&rowmax[row], __float_as_uint(amax));   // non-negative floats order like their bits (NaN: above everything)
#endif
  }
  if (!epi.row_amax) return;                   // (16-bit linears: the activation is the result, nothing to quantise)
  __syncthreads();
  for (int r = tid; r < ROWS; r += NWV * 64) {
    const int m = m_tile0 + r;
#ifndef WS_ABL_GU_NOATOMIC
    if (m < M && rowmax[r]) atomicMax(reinterpret_cast<unsigned*>(epi.row_amax) + m, rowmax[r]);
#endif
  }
}

// wave tile: MB 16-row blocks x NG 16-column groups; workgroup: WM x WN waves (four or eight); DW K tiles in flight. BOTH
// operands go HBM / L2 -> LDS by LDS-DMA in fragment order, so that every VMEM operation of a wave is an LDS-DMA with the same
// prefetch distance: vmcnt retires in order, and with activation loads into registers next to the weight DMAs the weights could
// never run further ahead than the (register-bound) activations -- measured: 2.1-3.0 TB/s. A wave issues its share of the
// tile's NDW weight + NDA activation fragments per iteration and waits for all but the (DW - 1) newest tiles.

template <int KIND, int NWV, int WM, int WN, int MB, int NG, int DW>
__global__ __launch_bounds__(NWV * 64, 1) void gemm_ws_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ Wp,
                                                             int M, int N, int64_t K, int m_tiles, int n_tiles,
                                                             int kt_per_slice, int n_slices, GemmEpi epi,
                                                             int32_t* __restrict__ slabs, int stagger) {
  static_assert(NWV == 4 || NWV == 8, "one or two waves per SIMD");
  static_assert(WM * WN == NWV, "waves per workgroup");
  static_assert(MB == 2 || MB == 4, "row blocks per wave");
  static_assert(KIND == kI8 || KIND == kFP8 || KIND == kBF16 || KIND == kF16, "operand kinds");
  using acc_t = typename WsAcc<KIND>::type;
  constexpr int G = WN * NG;              // 16-column groups of a workgroup tile
  constexpr int NS = DW + 1;              // LDS slots
  constexpr int SLOT_W = G * 2 * WS_FRAG, SLOT_A = WM * MB * 2 * WS_FRAG, SLOT = SLOT_W + SLOT_A;
  constexpr int NDW = (2 * G + NWV - 1) / NWV;  // weight LDS-DMA instructions per wave and K tile (the last ones may be padding)
  constexpr int NDA = (2 * WM * MB) / NWV;      // activation LDS-DMA instructions per wave and K tile
  constexpr bool PAD = (2 * G) % NWV != 0;      // some waves issue out-of-range padding pieces into the dump area
  constexpr int VMCNT = (DW - 1) * (NDW + NDA);
  static_assert((2 * WM * MB) % NWV == 0, "activation fragments per tile divide over the waves");
  static_assert(NS * SLOT + (PAD ? WS_FRAG : 0) <= 160 * 1024, "LDS ring");
  static_assert(VMCNT < 64, "vmcnt is a 6-bit counter");
  __shared__ __attribute__((aligned(1024))) uint8_t lds[NS * SLOT + (PAD ? WS_FRAG : 0)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // block -> (m tile, n tile, K slice): the m tiles of one column range sit on the same XCD (block b runs on XCD b % 8)
  // next to each other in dispatch order, so the second one finds the weights in that XCD's L2
  int mt, nt, slice;
  {
    const int b = blockIdx.x, x = b & 7, y = b >> 3;
    mt = y % m_tiles;
    const int rest = (y / m_tiles) * 8 + x;
    if (rest >= n_tiles * n_slices) return;
    nt = rest % n_tiles;
    slice = rest / n_tiles;
  }
  const int KT = (int)(K / WS_BK);
  const int w_nt = epi.w_policy ? epi.w_policy == 1 : m_tiles == 1;   // (WsDma::w_nt)
  const int kt0 = slice * kt_per_slice;
  int kt1 = kt0 + kt_per_slice;
  kt1 = kt1 > KT ? KT : kt1;
  const int nk = kt1 - kt0;  // >= 1 (the planner never makes an empty slice)
  // K-walk stagger (round 3): every workgroup of a launch reads the SAME activation slab per K tile -- 256 rows x 128 B at a
  // row pitch of K bytes, which the address hash puts on a few of the 16 L2 channels of an XCD -- so lock-step walkers queue
  // on those channels (the three structurally different kernels all stopped at ~7 TB/s of LDS fill at M = 256). The weights
  // of a decode GEMM are private to their workgroup (no L2 re-use to lose), so each workgroup starts its walk at another
  // tile and wraps around: concurrent readers are spread over all slabs = all channels. Integer sums do not depend on the
  // order; for fp8 the order is a fixed function of the block index (deterministic).
  const int phase = stagger ? (int)(((unsigned)blockIdx.x >> 3) % (unsigned)nk) : 0;
#define WS_ROT(X_) ((X_) + phase >= nk ? (X_) + phase - nk : (X_) + phase)
  const int n_groups = N >> 4;
  // balanced column split: tile nt owns groups [nt * n_groups / n_tiles, (nt + 1) * n_groups / n_tiles) -- at most G of them
  // (the launcher guarantees it), so that 2368 groups over 256 workgroups become 9 or 10 groups each instead of 197 x 12 + 4
  // gate_up mode (GemmEpi::gate_up): the tile owns ACT groups [ga0, ga0 + ga_live) -- its first G / 2 slot groups are their gate
  // columns, the other G / 2 their up columns (groups n_groups / 2 + ...); the descriptor then spans the whole matrix
  const bool gu = KIND != kFP8 && epi.gate_up != 0;
  const int n_groups_act = n_groups >> 1;
  const int ga0 = gu ? (int)((int64_t)nt * n_groups_act / n_tiles) : 0;
  const int ga_live = gu ? (int)((int64_t)(nt + 1) * n_groups_act / n_tiles) - ga0 : 0;
  const int g0 = gu ? 0 : (int)((int64_t)nt * n_groups / n_tiles);
  const int g_live = gu ? n_groups : (int)((int64_t)(nt + 1) * n_groups / n_tiles) - g0;
  const int m_tile0 = mt * (WM * MB * 16), m_base = m_tile0 + wm * (MB * 16);

  // ---- sources (rows past M and column groups past N read as zeros: buffer range check)
  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A), 0, (int)((int64_t)M * K), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(Wp) + (int64_t)g0 * KT * (2 * WS_FRAG), 0, (int)((int64_t)g_live * KT * (2 * WS_FRAG)), 0x00020000);
  int voff_a[8], voff_w[8];  // [NDA] / [NDW] used: an array of template-dependent size next to the LDS-DMA builtin makes hipcc drop the kernel's host stub
  static_assert(NDW <= 8 && NDA <= 8, "voff arrays");
  // activations: the tile sits in the LDS ROW-MAJOR (128 B per row), one DMA instruction = 8 rows x 128 B with lane j <->
  // (row j / 8, physical 16-B chunk j % 8): 8 lanes fetch one 128-B row segment (coalesced; a fragment-order gather of 16-B
  // pieces from 16 rows per instruction ran at 16 GB/s per CU, measured). The chunk index is XOR-swizzled by (row / 2) % 8 on
  // the SOURCE side so that the fragment reads (16 rows, one chunk) are free of bank conflicts.
#pragma unroll
  for (int i = 0; i < NDA; ++i) {
    const int row = (wave * NDA + i) * 8 + (lane >> 3);
    voff_a[i] = (m_tile0 + row) * (int)K + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
  }
  // this wave's weight fragments of a tile: f = wave * NDW + i (group f / 2, k step f % 2); f >= 2 G is padding
  const int nvalid_w = PAD ? (2 * G - wave * NDW < 0 ? 0 : (2 * G - wave * NDW > NDW ? NDW : 2 * G - wave * NDW)) : NDW;
#pragma unroll
  for (int i = 0; i < NDW; ++i) {
    const int f = wave * NDW + i;
    int grp_i = f >> 1;
    bool ok = i < nvalid_w;
    if (gu) {   // slot group -> (gate | up) group of the act group it belongs to
      const int half = grp_i / (G / 2), j = grp_i % (G / 2);
      ok = ok && j < ga_live;
      grp_i = half * n_groups_act + ga0 + j;
    }
    voff_w[i] = ok ? grp_i * KT * (2 * WS_FRAG) + (f & 1) * WS_FRAG + lane * 16 : 0x7ffff000;
  }
  const ws_lds_ptr_t lds3 = (ws_lds_ptr_t)lds;
  const ws_lds_ptr_t lds_dump = lds3 + NS * SLOT;  // (only addressed when PAD)
  const unsigned rd_w = (unsigned)(__UINTPTR_TYPE__)lds3 + wn * NG * (2 * WS_FRAG) + lane * 16;
  // fragment (row block mb, k step ks): lane l reads row mb*16 + (l & 15), logical chunk ks*4 + (l >> 4)
  const unsigned rd_a_base = (unsigned)(__UINTPTR_TYPE__)lds3 + SLOT_W + (wm * MB * 16 + (lane & 15)) * WS_BK;
  const unsigned rd_a0 = rd_a_base + ((((lane >> 4)) ^ ((lane & 15) >> 1)) << 4);
  const unsigned rd_a1 = rd_a_base + (((4 + (lane >> 4)) ^ ((lane & 15) >> 1)) << 4);

  acc_t acc[MB][NG];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NG; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

  // tile T_ of the slice -> LDS slot T_ % NS (tiles past the end re-load the last tile: every load is unconditional, so
  // the vmcnt arithmetic is static; nobody reads them)
#define WS_ISSUE(T_)                                                                                                 \
  {                                                                                                                  \
    const int kc_ = (T_) < nk ? (T_) : nk - 1;                                                                       \
    const int kt_ = kt0 + WS_ROT(kc_);                                                                               \
    const ws_lds_ptr_t dw_ = lds3 + ((T_) % NS) * SLOT + wave * NDW * WS_FRAG;                                       \
    const ws_lds_ptr_t da_ = lds3 + ((T_) % NS) * SLOT + SLOT_W + wave * NDA * WS_FRAG;                              \
    _Pragma("unroll") for (int i_ = 0; i_ < NDA; ++i_)                                                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(WS_ABL_RSRC_A, da_ + i_ * WS_FRAG, 16, voff_a[i_], kt_ * WS_BK, 0, 0); \
    _Pragma("unroll") for (int i_ = 0; i_ < NDW; ++i_) {                                                             \
      if (w_nt)                                                                                                      \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(WS_ABL_RSRC_W, i_ < nvalid_w ? dw_ + i_ * WS_FRAG : lds_dump, 16,   \
                                                 voff_w[i_], kt_ * (2 * WS_FRAG), 0, WS_W_AUX);                      \
      else                                                                                                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(WS_ABL_RSRC_W, i_ < nvalid_w ? dw_ + i_ * WS_FRAG : lds_dump, 16,   \
                                                 voff_w[i_], kt_ * (2 * WS_FRAG), 0, 0);                             \
    }                                                                                                                \
  }

  // ablation builds (tools/build_ablations.sh; timing only, WRONG results): an empty buffer descriptor makes every DMA of
  // that operand an out-of-range access (no memory request, zeros to the LDS) at unchanged instruction counts
#ifdef WS_ABL_NOADMA
  const __amdgpu_buffer_rsrc_t rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A), 0, 0, 0x00020000);
#define WS_ABL_RSRC_A rsrc_a0
#else
#define WS_ABL_RSRC_A rsrc_a
#endif
#ifdef WS_ABL_NOWDMA
  const __amdgpu_buffer_rsrc_t rsrc_w0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(Wp), 0, 0, 0x00020000);
#define WS_ABL_RSRC_W rsrc_w0
#else
#define WS_ABL_RSRC_W rsrc_w
#endif
  // eight waves: the second-dispatched half loses every arbitration against the older half (MI355X_MICROARCH.md, "static
  // priority for the younger half"); one static raise, no per-segment flips
  if constexpr (NWV == 8) {
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  }
#pragma unroll
  for (int i = 0; i < DW; ++i) WS_ISSUE(i)
  for (int t = 0; t < nk; ++t) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMCNT) : "memory");  // this wave's pieces of tile t have landed
    __builtin_amdgcn_s_barrier();  // everybody's have; and everybody has finished reading tile t - 1
    __builtin_amdgcn_sched_barrier(0);
    // tile t + DW goes into the slot of tile t - 1, piece by piece between the MFMA groups of tile t
    const int tn = t + DW, tc = tn < nk ? tn : nk - 1, ktn = kt0 + WS_ROT(tc);
    WsDma d;
    d.rsrc_a = WS_ABL_RSRC_A;
    d.rsrc_w = WS_ABL_RSRC_W;
    d.dst_w = lds3 + (tn % NS) * SLOT + wave * NDW * WS_FRAG;
    d.dst_a = lds3 + (tn % NS) * SLOT + SLOT_W + wave * NDA * WS_FRAG;
    d.so_a = ktn * WS_BK;
    d.so_w = ktn * (2 * WS_FRAG);
    d.dump = lds_dump;
    d.nvalid_w = nvalid_w;
    d.w_nt = w_nt;
    const unsigned so = (t % NS) * SLOT;
#ifndef WS_ABL_NOCOMPUTE
    if constexpr (KIND == kI8) ws_compute<MB, NG, NDA, NDW>(acc, rd_w + so, rd_a0 + so, rd_a1 + so, d, voff_a, voff_w);
    else if constexpr (KIND == kFP8) ws_compute_fp8<MB, NG, NDA, NDW>(acc, rd_w + so, rd_a0 + so, rd_a1 + so, d, voff_a, voff_w);
    else ws_compute_h16<KIND, MB, NG, NDA, NDW>(acc, rd_w + so, rd_a0 + so, rd_a1 + so, d, voff_a, voff_w);
#else
    WS_ISSUE(tn)
#endif
  }
#undef WS_ISSUE
#undef WS_ROT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail prefetches have landed before the LDS is released
  if constexpr (KIND == kI8) {
    // MFMA -> VALU read hazard of the asm MFMAs: the accumulators of the LAST column group were written by the last MB MFMAs;
    // the nops carry them as operands so that the epilogue's reads of exactly those registers are ordered behind the nops
    // (every other accumulator was written >= MB * 16 cycles before the loop ended)
    if constexpr (MB == 4)
      asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0][NG - 1]), "+a"(acc[1][NG - 1]), "+a"(acc[2][NG - 1]), "+a"(acc[3][NG - 1]));
    else
      asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0][NG - 1]), "+a"(acc[1][NG - 1]));
  }

  __builtin_amdgcn_s_barrier();  // every wave's DMAs have landed and nobody reads the ring any more: the epilogue re-uses the LDS
  if constexpr (KIND != kFP8 && (WN * NG) % 2 == 0) {
    if (gu) {
      ws_epilogue_gate_up<KIND, MB, NG, WM, WN, NWV>(acc, epi, M, N, m_tile0, wm, wn, lane, tid, ga0, ga_live, n_groups_act, lds);
      return;
    }
  }
  ws_epilogue<KIND, MB, NG>(acc, epi, slabs, slice, n_slices, M, N, m_base, g0, g_live, wn, lane, lds, wave, nt * WN + wn);
}

// ------------------------------------------------------------------------------------------------ staggered eight-wave tile
// Round 3, after the ablation builds of the plain eight-wave tile (profiles/r03_gemm_ws.txt): its K-tile time is the SUM of
// its fragment-read, MFMA and DMA-issue phases (gate_up, M = 256: 53.8 us full, 49.7 without any compute, 40.4 without the
// weight DMA; 4500 shader cycles per K tile against 1280 matrix-pipe cycles) -- the two waves of a SIMD leave the per-tile
// barrier together, both read fragments, both then want the matrix pipe, both then wait for DMAs: nothing overlaps.
// This kernel runs the SAME tile with the two wave groups (waves 0-3 = rows 0-127, waves 4-7 = rows 128-255; wave w and
// w + 4 share a SIMD) ONE BARRIER APART, the way the 8-phase kernel of gemm_p8.hip does:
//     group 0:         READ(t) | MFMA(t)   | READ(t+1) | MFMA(t+1) | ...
//     group 1:  (bar)          | READ(t)   | MFMA(t)   | READ(t+1) | ...          ("|" = workgroup barrier)
// so every SIMD has one wave in its matrix phase while the other reads its fragments: the LDS and the matrix pipe work at the
// same time. READ(t) = all 2 MB + 2 NG fragment reads of the K tile into registers, retired (lgkmcnt(0)) before the barrier;
// MFMA(t) = 2 MB NG MFMAs back to back from registers, the wave's LDS-DMA pieces of a later tile in between, then the counted
// vmcnt wait. Slot / landing rules (NS = DW + 1 ring slots, tile T lives in slot T % NS; global phase 2t = group 0 reads
// tile t, phase 2t + 1 = group 1 reads it):
//   * during MFMA(t) group 0 (phase 2t + 1) requests tile t + DW -> the slot of tile t - 1, last read in phase 2t - 1;
//     group 1 (phase 2t + 2) requests tile t + DW + 1 -> the slot of tile t, last read (by itself, retired) in phase 2t + 1;
//   * the wait at the end of MFMA(t) leaves DW - 1 tiles of the wave's own pieces in flight: group 0 then knows tile t + 1
//     (read from phase 2t + 2 on) has landed, group 1 knows tile t + 2 has; a tile is only read after the barrier behind
//     every wave's wait for it. The prologue requests tiles 0 .. DW - 1 (+ tile DW in group 1) and waits for tiles 0 (and 1).
//   * second version (in-kernel timing: the matrix phase was 900 cycles for 680 of MFMA issue, the read phase 324 -- the DMA
//     issue sat in the longer phase): a tile's ND pieces are split, NRD of them are issued in a READ phase where the wave
//     only waits for the LDS. Group 0: pieces [0, NRD) of tile t + DW in READ(t), the rest in MFMA(t); group 1: pieces
//     [0, NRD) of tile t + DW + 1 in MFMA(t), the rest in READ(t + 1) (its slot, that of tile t, is being read by group 1
//     itself during READ(t), so nothing of that tile may be requested before MFMA(t)). Group 1 therefore waits (vmcnt) at the
//     end of its READ phase, group 0 at the end of its matrix phase, both leaving DW - 1 whole tiles in flight.
#ifdef WS8_TIMING
__device__ long long ws8_dbg[64];
#endif
template <int KIND, int NG, int DW, int NRD, int WNT>
__global__ __launch_bounds__(512, 1) void gemm_ws8s_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ Wp,
                                                          int M, int N, int64_t K, int m_tiles, int n_tiles,
                                                          int kt_per_slice, int n_slices, GemmEpi epi,
                                                          int32_t* __restrict__ slabs) {
  constexpr int NWV = 8, WM = 4, WN = 2, MB = 4;
  using acc_t = typename WsAcc<KIND>::type;
  constexpr int G = WN * NG, NS = DW + 1;
  constexpr int SLOT_W = G * 2 * WS_FRAG, SLOT_A = WM * MB * 2 * WS_FRAG, SLOT = SLOT_W + SLOT_A;
  constexpr int NDW = (2 * G + NWV - 1) / NWV, NDA = (2 * WM * MB) / NWV, ND = NDA + NDW;
  constexpr bool PAD = (2 * G) % NWV != 0;
  constexpr int VMCNT = (DW - 1) * ND;
  static_assert(NS * SLOT + (PAD ? WS_FRAG : 0) <= 160 * 1024, "LDS ring");
  static_assert(DW * ND + ND < 64, "vmcnt is a 6-bit counter");
  static_assert(NRD >= 0 && NRD <= ND && DW >= 2, "pieces of a tile requested from a read phase");
  __shared__ __attribute__((aligned(1024))) uint8_t lds[NS * SLOT + (PAD ? WS_FRAG : 0)];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN, grp = wave >> 2;
#ifdef WS8_TIMING
  const long long t_entry = __builtin_readcyclecounter(), r_entry = __builtin_amdgcn_s_memrealtime();
#endif
  int mt, nt, slice;
  {
    const int b = blockIdx.x, x = b & 7, y = b >> 3;
    mt = y % m_tiles;
    const int rest = (y / m_tiles) * 8 + x;
    if (rest >= n_tiles * n_slices) return;
    nt = rest % n_tiles;
    slice = rest / n_tiles;
  }
  const int KT = (int)(K / WS_BK);
  constexpr int w_nt = WNT;   // (the launcher turns epi.w_policy / the tile count into the template argument)
  const int kt0 = slice * kt_per_slice;
  int kt1 = kt0 + kt_per_slice;
  kt1 = kt1 > KT ? KT : kt1;
  const int nk = kt1 - kt0;
  const int n_groups = N >> 4;
  // gate_up mode (GemmEpi::gate_up): the tile owns ACT groups [ga0, ga0 + ga_live) -- its first G / 2 slot groups are their gate
  // columns, the other G / 2 their up columns (groups n_groups / 2 + ...); the descriptor then spans the whole matrix
  const bool gu = KIND != kFP8 && epi.gate_up != 0;
  const int n_groups_act = n_groups >> 1;
  const int ga0 = gu ? (int)((int64_t)nt * n_groups_act / n_tiles) : 0;
  const int ga_live = gu ? (int)((int64_t)(nt + 1) * n_groups_act / n_tiles) - ga0 : 0;
  const int g0 = gu ? 0 : (int)((int64_t)nt * n_groups / n_tiles);
  const int g_live = gu ? n_groups : (int)((int64_t)(nt + 1) * n_groups / n_tiles) - g0;
  const int m_tile0 = mt * (WM * MB * 16), m_base = m_tile0 + wm * (MB * 16);

  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A), 0, (int)((int64_t)M * K), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(Wp) + (int64_t)g0 * KT * (2 * WS_FRAG), 0, (int)((int64_t)g_live * KT * (2 * WS_FRAG)), 0x00020000);
  int voff_a[8], voff_w[8];
  static_assert(NDW <= 8 && NDA <= 8, "voff arrays");
#pragma unroll
  for (int i = 0; i < NDA; ++i) {
    const int row = (wave * NDA + i) * 8 + (lane >> 3);
    voff_a[i] = (m_tile0 + row) * (int)K + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
  }
  const int nvalid_w = PAD ? (2 * G - wave * NDW < 0 ? 0 : (2 * G - wave * NDW > NDW ? NDW : 2 * G - wave * NDW)) : NDW;
#pragma unroll
  for (int i = 0; i < NDW; ++i) {
    const int f = wave * NDW + i;
    int grp_i = f >> 1;
    bool ok = i < nvalid_w;
    if (gu) {   // slot group -> (gate | up) group of the act group it belongs to
      const int half = grp_i / (G / 2), j = grp_i % (G / 2);
      ok = ok && j < ga_live;
      grp_i = half * n_groups_act + ga0 + j;
    }
    voff_w[i] = ok ? grp_i * KT * (2 * WS_FRAG) + (f & 1) * WS_FRAG + lane * 16 : 0x7ffff000;
  }
  const ws_lds_ptr_t lds3 = (ws_lds_ptr_t)lds;
  const ws_lds_ptr_t lds_dump = lds3 + NS * SLOT;
  const unsigned rd_w = (unsigned)(__UINTPTR_TYPE__)lds3 + wn * NG * (2 * WS_FRAG) + lane * 16;
  const unsigned rd_a_base = (unsigned)(__UINTPTR_TYPE__)lds3 + SLOT_W + (wm * MB * 16 + (lane & 15)) * WS_BK;
  const unsigned rd_a0 = rd_a_base + ((((lane >> 4)) ^ ((lane & 15) >> 1)) << 4);
  const unsigned rd_a1 = rd_a_base + (((4 + (lane >> 4)) ^ ((lane & 15) >> 1)) << 4);

  acc_t acc[MB][NG];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NG; ++j) acc[i][j] = acc_t{0, 0, 0, 0};

  // all pieces of tile T_ (clamped to the slice: requests past the end re-load the last tile into a slot nobody reads any more)
#define WS8_ISSUE(T_)                                                                                                \
  {                                                                                                                  \
    const int kt_ = kt0 + ((T_) < nk ? (T_) : nk - 1);                                                               \
    const ws_lds_ptr_t dw_ = lds3 + ((T_) % NS) * SLOT + wave * NDW * WS_FRAG;                                       \
    const ws_lds_ptr_t da_ = lds3 + ((T_) % NS) * SLOT + SLOT_W + wave * NDA * WS_FRAG;                              \
    _Pragma("unroll") for (int i_ = 0; i_ < NDA; ++i_)                                                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, da_ + i_ * WS_FRAG, 16, voff_a[i_], kt_ * WS_BK, 0, 0);     \
    _Pragma("unroll") for (int i_ = 0; i_ < NDW; ++i_) {                                                             \
      if (w_nt)                                                                                                      \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, i_ < nvalid_w ? dw_ + i_ * WS_FRAG : lds_dump, 16,          \
                                                 voff_w[i_], kt_ * (2 * WS_FRAG), 0, WS_W_AUX);                      \
      else                                                                                                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, i_ < nvalid_w ? dw_ + i_ * WS_FRAG : lds_dump, 16,          \
                                                 voff_w[i_], kt_ * (2 * WS_FRAG), 0, 0);                             \
    }                                                                                                                \
  }
#pragma unroll
  for (int i = 0; i < DW; ++i) WS8_ISSUE(i)
  // the descriptor of tile T_'s requests (clamped to the slice like WS8_ISSUE)
#define WS8_DESC(D_, T_)                                                                                             \
  WsDma D_;                                                                                                          \
  {                                                                                                                  \
    const int tt_ = (T_), kk_ = kt0 + (tt_ < nk ? tt_ : nk - 1);                                                     \
    D_.rsrc_a = rsrc_a;                                                                                              \
    D_.rsrc_w = rsrc_w;                                                                                              \
    D_.dst_w = lds3 + (tt_ % NS) * SLOT + wave * NDW * WS_FRAG;                                                      \
    D_.dst_a = lds3 + (tt_ % NS) * SLOT + SLOT_W + wave * NDA * WS_FRAG;                                             \
    D_.so_a = kk_ * WS_BK;                                                                                           \
    D_.so_w = kk_ * (2 * WS_FRAG);                                                                                   \
    D_.dump = lds_dump;                                                                                              \
    D_.nvalid_w = nvalid_w;                                                                                          \
    D_.w_nt = w_nt;                                                                                                  \
  }
  if (grp) {
    WS8_DESC(dp, DW)
    ws_dma_range<NDA, NDW, 0, NRD, WNT>(dp, voff_a, voff_w, 0, 1);   // group 1 is one half-tile of requests ahead
    // tiles 0 and 1 have landed: DW - 2 whole tiles + the NRD pieces stay in flight
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DW - 2) * ND + NRD) : "memory");
    __builtin_amdgcn_s_setprio(1);  // the younger half loses every arbitration otherwise (MI355X_MICROARCH.md)
  } else {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMCNT) : "memory");  // tile 0 has landed
  }
  __builtin_amdgcn_s_barrier();
  if (grp) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0

  u32x4 a0[MB], a1[MB], w0[NG], w1[NG];
#ifdef WS8_TIMING  /* timing build (tools/r03): shader cycles per phase, summed over the K tiles, block 0, waves 0 and 4 */
  const long long t_loop = __builtin_readcyclecounter();
  long long tacc[5] = {0, 0, 0, 0, 0}, tq0, tq1;
#define WS8_T(i_) { tq1 = __builtin_readcyclecounter(); tacc[i_] += tq1 - tq0; tq0 = tq1; }
#else
#define WS8_T(i_)
#endif
  for (int t = 0; t < nk; ++t) {
    // ---- READ(t): every fragment of the tile into registers
    const unsigned so = (t % NS) * SLOT;
    __builtin_amdgcn_sched_barrier(0);
#ifdef WS8_TIMING
    tq0 = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) WS_DSR(a0[mb], rd_a0 + so, mb * 16 * WS_BK);
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) WS_DSR(w0[ng], rd_w + so, ng * 2 * WS_FRAG);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) WS_DSR(a1[mb], rd_a1 + so, mb * 16 * WS_BK);
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) WS_DSR(w1[ng], rd_w + so, ng * 2 * WS_FRAG + WS_FRAG);
    {  // this phase's share of tile t + DW: group 0 opens the tile, group 1 completes it (and then waits for tile t + 1)
      WS8_DESC(dr, t + DW)
      if (grp) {
        ws_dma_range<NDA, NDW, NRD, ND, WNT>(dr, voff_a, voff_w, 0, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMCNT) : "memory");
      } else {
        ws_dma_range<NDA, NDW, 0, NRD, WNT>(dr, voff_a, voff_w, 0, 1);
      }
    }
    // retired before the barrier: the other group (or this one) re-stages the slot in the next phase
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0[0]), "+v"(a0[1]), "+v"(a0[2]), "+v"(a0[3]), "+v"(a1[0]), "+v"(a1[1]),
                 "+v"(a1[2]), "+v"(a1[3]));
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) asm volatile("" : "+v"(w0[ng]), "+v"(w1[ng]));
    __builtin_amdgcn_sched_barrier(0);
    WS8_T(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    WS8_T(1)
    // ---- MFMA(t) from registers; between the column groups: group 0 the rest of tile t + DW, group 1 the head of t + DW + 1
    WS8_DESC(d, t + DW + grp)
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) {
      if constexpr (KIND == kI8) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) WS_MFMA(acc[mb][ng], w0[ng], a0[mb]);
        if (grp) ws_dma_range<NDA, NDW, 0, NRD, WNT>(d, voff_a, voff_w, ng, 2 * NG);
        else ws_dma_range<NDA, NDW, NRD, ND, WNT>(d, voff_a, voff_w, ng, 2 * NG);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) WS_MFMA(acc[mb][ng], w1[ng], a1[mb]);
        if (grp) ws_dma_range<NDA, NDW, 0, NRD, WNT>(d, voff_a, voff_w, NG + ng, 2 * NG);
        else ws_dma_range<NDA, NDW, NRD, ND, WNT>(d, voff_a, voff_w, NG + ng, 2 * NG);
      } else if constexpr (KIND == kFP8) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb][ng] = ws_mma_fp8(w0[ng], w1[ng], a0[mb], a1[mb], acc[mb][ng]);
        if (grp) ws_dma_range<NDA, NDW, 0, NRD, WNT>(d, voff_a, voff_w, ng, NG);
        else ws_dma_range<NDA, NDW, NRD, ND, WNT>(d, voff_a, voff_w, ng, NG);
      } else {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb][ng] = ws_mma_h16<KIND>(w0[ng], a0[mb], acc[mb][ng]);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb][ng] = ws_mma_h16<KIND>(w1[ng], a1[mb], acc[mb][ng]);
        if (grp) ws_dma_range<NDA, NDW, 0, NRD, WNT>(d, voff_a, voff_w, ng, NG);
        else ws_dma_range<NDA, NDW, NRD, ND, WNT>(d, voff_a, voff_w, ng, NG);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    WS8_T(2)
    if (!grp) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMCNT) : "memory");
    WS8_T(3)
    __builtin_amdgcn_s_barrier();
    WS8_T(4)
  }
#ifdef WS8_TIMING
  const long long t_loop_end = __builtin_readcyclecounter();
#endif
#undef WS8_DESC
#undef WS8_T
#undef WS8_ISSUE
  if (!grp) __builtin_amdgcn_s_barrier();          // balance group 1's extra barrier
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the tail requests have landed before the LDS is released
  if constexpr (KIND == kI8)
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[0][NG - 1]), "+a"(acc[1][NG - 1]), "+a"(acc[2][NG - 1]), "+a"(acc[3][NG - 1]));
  __builtin_amdgcn_s_barrier();  // every wave's DMAs have landed: the epilogue re-uses the LDS
  if constexpr (KIND != kFP8) {
    if (gu) ws_epilogue_gate_up<KIND, MB, NG, WM, WN, NWV>(acc, epi, M, N, m_tile0, wm, wn, lane, tid, ga0, ga_live, n_groups_act, lds);
  }
  if (!gu) ws_epilogue<KIND, MB, NG>(acc, epi, slabs, slice, n_slices, M, N, m_base, g0, g_live, wn, lane, lds, wave, nt * WN + wn);
#ifdef WS8_TIMING
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the epilogue's stores have been accepted
    const long long t_end = __builtin_readcyclecounter(), r_end = __builtin_amdgcn_s_memrealtime();
    const int sel = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x / 2 ? 1 : -1);
    if (sel >= 0 && (wave & 3) == 0 && lane == 0) {
      long long* o = ws8_dbg + sel * 32 + grp * 16;
      for (int i = 0; i < 5; ++i) o[i] = tacc[i];
      o[5] = nk;
      o[6] = t_loop - t_entry;      // prologue: entry -> first K tile readable
      o[7] = t_loop_end - t_loop;   // K loop
      o[8] = t_end - t_loop_end;    // epilogue
      o[9] = r_end - r_entry;       // the same span on the 100 MHz counter
      o[10] = t_end - t_entry;
    }
  }
#endif
}

// sum of the K-slice slabs (exact int32; fp8: fp32 in slice order) + the dequant epilogue; nothing is zeroed
template <int KIND>
__global__ __launch_bounds__(256) void ws_slab_epilogue_kernel(const int32_t* __restrict__ slabs, int n_slices, int64_t M,
                                                               int64_t N, GemmEpi epi) {
  using acc_t = typename WsAcc<KIND>::type;
  const int64_t total = M * N;
  for (int64_t idx = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x * 4) {
    // every load of the element group is requested before the first is consumed (round 6: the slab loop, then the scales, then the
    // bias were up to ten dependent memory latencies of a 5.8-us launch); sums in slice order as before (fp32 kinds: same bits)
    const int64_t m = idx / N, n = idx - m * N;
    acc_t sl[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      sl[s] = acc_t{0, 0, 0, 0};
      if (s < n_slices) sl[s] = *reinterpret_cast<const acc_t*>(slabs + (int64_t)s * total + idx);
    }
    float as = 1.0f, wsv[4] = {1.f, 1.f, 1.f, 1.f}, bsv[4] = {0.f, 0.f, 0.f, 0.f};
    if (epi.out) {
      if constexpr (KIND == kI8) {
        as = epi.a_scale[m];
#pragma unroll
        for (int e = 0; e < 4; ++e) wsv[e] = epi.w_scale[n + e];
      } else if constexpr (KIND == kFP8) {
        as = epi.a_scale[epi.a_scale_n > 1 ? m : 0];
#pragma unroll
        for (int e = 0; e < 4; ++e) wsv[e] = epi.w_scale[epi.w_scale_n > 1 ? n + e : 0];
      }
      if (epi.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) bsv[e] = load16(epi.bias, n + e, epi.out_bf16);
      }
    }
    acc_t a = sl[0];
#pragma unroll
    for (int s = 1; s < 8; ++s)
      if (s < n_slices) a += sl[s];
    for (int s = 8; s < n_slices; ++s) a += *reinterpret_cast<const acc_t*>(slabs + (int64_t)s * total + idx);   // (no plan makes > 8)
    if constexpr (KIND == kI8) {
      if (epi.acc_out) *reinterpret_cast<i32x4_t*>(epi.acc_out + idx) = a;
    }
    if (!epi.out) continue;
    float v[4];
    if constexpr (KIND == kI8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (float)a[e] * as * wsv[e] + bsv[e];
    } else if constexpr (KIND == kFP8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = as * (wsv[e] * a[e]) + bsv[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = a[e] + bsv[e];
    }
    uint2 pk;
    if (epi.out_bf16) { pk.x = pack2x16<true>(v[0], v[1]); pk.y = pack2x16<true>(v[2], v[3]); }
    else { pk.x = pack2x16<false>(v[0], v[1]); pk.y = pack2x16<false>(v[2], v[3]); }
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(epi.out) + idx) = pk;
  }
}

// ------------------------------------------------------------------------------------------------ planner + launch
struct WsPlan { int waves, wm, wn, mb, ng, slices; };

// Tuning arms (compiled only into the -DXM_TUNING flavour, `make tuning`; constants in the product library):
//   XLLM_MI355_KSTAGGER=0     every workgroup walks K from its first tile (round-2 behaviour; lost the round-3 A/B)
//   XLLM_MI355_WS8_STAGGER=0  the eight-wave tile with all waves in phase (lost the round-3 A/B)
//   w_policy                  cache policy of the weight stream: 0 by shape (default) | 1 nt | 2 default
XM_TUNE_VAR(f_kstagger, "XLLM_MI355_KSTAGGER", 1);
XM_TUNE_VAR(f_ws8s, "XLLM_MI355_WS8_STAGGER", 1);
XM_TUNE_VAR(f_wpolicy, "XLLM_MI355_WS_WPOLICY", 0);
XM_TUNE_VAR(f_slab_nt, "XLLM_MI355_SLAB_NT", 1);   // non-temporal slab stores: -0.09 … -0.125 ms per step (profiles/r04_step_ab_nt.txt)
static thread_local int g_argmax_slots_used = 0;   // slots the last argmax-mode launch of this thread filled (read by its caller)
template <int KIND, int NWV, int WM, int WN, int MB, int NG, int DW>
int ws_launch_cfg(const void* A, const void* Wp, int64_t M, int64_t N, int64_t K, int slices, GemmEpi epi,
                         int32_t* slabs, hipStream_t s) {
  constexpr int G = WN * NG;
  const int m_tiles = (int)((M + WM * MB * 16 - 1) / (WM * MB * 16));
  // gate_up mode: a tile holds G / 2 ACT groups (their gate and their up columns); the split is over the N / 32 act groups
  const bool gu = epi.gate_up != 0;
  epi.w_policy = f_wpolicy;
  epi.slab_nt = f_slab_nt;
  if (gu && (G % 2 != 0 || KIND == kFP8)) return -1;
  const int n_groups = gu ? (int)(N / 32) : (int)(N / 16);
  const int KT = (int)(K / WS_BK);
  if (gu || epi.argmax_val) slices = 1;
  int per = (KT + slices - 1) / slices;
  slices = (KT + per - 1) / per;  // no empty slice
  // at least ceil(n_groups / G) column tiles; more (narrower, balanced) ones while the grid still fits one round of 256 CUs
  constexpr int GCAP = G;   // (gate_up: capacity G / 2 act groups, applied below)
  int n_tiles = gu ? (n_groups + G / 2 - 1) / (G / 2) : (n_groups + GCAP - 1) / GCAP;
  const int fit = 256 / (m_tiles * slices);
  if (fit > n_tiles) n_tiles = fit < n_groups ? fit : n_groups;
  const int rest = n_tiles * slices;
  const unsigned grid = (unsigned)(((rest + 7) / 8) * 8 * m_tiles);
  if (epi.argmax_val) {   // one partial slot per (column tile, wave column): the caller's arrays must hold them
    if (n_tiles * WN > epi.argmax_slots) return -2;
    g_argmax_slots_used = n_tiles * WN;
  }
  if constexpr (NWV == 8) {
#ifdef XM_TUNING
    if (!f_ws8s) {
      hipLaunchKernelGGL((gemm_ws_kernel<KIND, NWV, WM, WN, MB, NG, DW>), dim3(grid), dim3(NWV * 64), 0, s, (const uint8_t*)A,
                         (const uint8_t*)Wp, (int)M, (int)N, K, m_tiles, n_tiles, per, slices, epi, slabs, f_kstagger);
      return slices;
    }
#endif
    constexpr int ND8 = (2 * WN * NG + 7) / 8 + 4;  // pieces per wave and tile (weights + 4 activation pieces)
    // (a first version issued every request from the matrix phase, NRD = 0: 46.0 vs 45.4 us for gate_up at M = 256 -- removed)
    const bool w_nt = epi.w_policy ? epi.w_policy == 1 : m_tiles == 1;   // (WsDma::w_nt, decided here since round 5)
    if (w_nt)
      hipLaunchKernelGGL((gemm_ws8s_kernel<KIND, NG, DW, ND8 / 2, 1>), dim3(grid), dim3(512), 0, s, (const uint8_t*)A,
                         (const uint8_t*)Wp, (int)M, (int)N, K, m_tiles, n_tiles, per, slices, epi, slabs);
    else
      hipLaunchKernelGGL((gemm_ws8s_kernel<KIND, NG, DW, ND8 / 2, 0>), dim3(grid), dim3(512), 0, s, (const uint8_t*)A,
                         (const uint8_t*)Wp, (int)M, (int)N, K, m_tiles, n_tiles, per, slices, epi, slabs);
  } else {
    hipLaunchKernelGGL((gemm_ws_kernel<KIND, NWV, WM, WN, MB, NG, DW>), dim3(grid), dim3(NWV * 64), 0, s, (const uint8_t*)A,
                       (const uint8_t*)Wp, (int)M, (int)N, K, m_tiles, n_tiles, per, slices, epi, slabs, f_kstagger);
  }
  return slices;
}

// (M, N, K) -> tile shape and K slices. The tile height follows M; the width and the slice count are chosen so that the
// grid comes as close as possible to one workgroup on each of the 256 CUs (two for the small tiles) without exceeding it,
// with the partial-sum slabs (slices * M * N * 4 bytes written and read back) priced in.
// Planner hint of the CALLING THREAD (xllm_mi355_gemm_plan_hint, include/xllm_mi355.h): tile width (16-column groups per wave),
// K-slice count and tile height of the following packed-GEMM launches from this thread; 0 = planner. thread_local: a worker
// thread's hint never races with another worker's launches.
static thread_local int f_ng = 0, f_sl = 0, f_rows = 0;
#ifdef XM_TUNING
struct WsShapePlan { int64_t N, K; int ng, slices; };
static WsShapePlan shape_plans[8];
static int n_shape_plans = 0;
#endif
// 128-row tiles for 128 < M <= 512 when N <= 20480 and (K <= 8192 or M % 256 == 0); 256-row eight-wave tiles otherwise
// (round-3 in-step A/B, profiles/r03_step_ab.txt; the round-2 four-wave 256-row tile left the library in round 3)
static WsPlan ws_plan(int64_t M, int64_t N, int64_t K, bool can_slice, size_t ws_bytes, bool gu = false) {
  WsPlan p;
  p.waves = 4;
  if (M <= 32) { p.wm = 1; p.wn = 4; p.mb = 2; }
  else if (M <= 64) { p.wm = 1; p.wn = 4; p.mb = 4; }
  else if (M <= 128) { p.wm = 2; p.wn = 2; p.mb = 4; }
  else if (f_rows == 128 || (f_rows != 256 && N <= 20480 && (K <= 8192 || M % 256 == 0))) {
    // 128-row tiles also above 128 rows for the few-column problems (qkv, o, down: N <= 8192). They need K slices to fill the
    // chip; with 2+ m tiles per column range half as many slices do, i.e. half the slab bytes written here and read back by the
    // fused consumer -- at the price of streaming the weights once per m tile (the m tiles of a column range are neighbours in
    // dispatch order on one XCD, but its L2 does not hold the stream between them). Measured (profiles/r03_step_ab.txt):
    // in-process A/B of the step at B = 256 -0.14 ms (qkv / o -0.05, down -0.09), gate_up (never sliced) +0.35 ms; stand-alone
    // qkv / o win or tie at every M (M = 512: 27.5 -> 18.6, 24.0 -> 15.7 us), the long-K down projection only ties at
    // M = 256 / 512 and loses at 160 / 384 (30.7 -> 36.4, 46.9 -> 72.8 us): hence K <= 8192 or whole 256-row multiples.
    // N up to 20480: the gate_up shards of TP = 2 / 4 (N = 18944 / 9472) gain 0.06 / 0.07 ms per step (profiles/r03_tp_shapes.txt).
    p.wm = 2; p.wn = 2; p.mb = 4;
  }
  else { p.waves = 8; p.wm = 4; p.wn = 2; p.mb = 4; }
  const int rows = p.wm * p.mb * 16;
  const int m_tiles = (int)((M + rows - 1) / rows);
  // gate_up mode: the columns are split over the N / 32 act groups and a tile of G slot groups holds G / 2 of them
  const int n_groups = gu ? (int)(N / 32) : (int)(N / 16), KT = (int)(K / WS_BK);
  const int cap_div = gu ? 2 : 1;
  if (gu) can_slice = false;
  // candidates of the family, narrowest first (a narrower ring slot = more tiles in flight in the same LDS)
  static const int ngs_w1[] = {2, 4, 6, 8, 10}, ngs_w2[] = {1, 2, 3, 4, 5}, ngs_w4[] = {1, 2, 3};
  const int* ngs = p.wn == 1 ? ngs_w1 : (p.wn == 2 ? ngs_w2 : ngs_w4);   // (wn == 1: no family left since round 3)
  const int n_ngs = p.wn == 4 ? 3 : 5;
  const int g_max = p.wn * ngs[n_ngs - 1] / cap_div;
  const int per_simd = p.waves / 4;  // waves sharing one matrix pipe
  double best = 1e30;
  p.ng = ngs[n_ngs - 1];
  p.slices = 1;
  for (int sl = 1; sl <= 8; ++sl) {
    if (sl > 1 && (!can_slice || (size_t)sl * M * N * 4 > ws_bytes || KT / sl < 4)) break;
    int fit = 256 / (m_tiles * sl);
    fit = fit < 1 ? 1 : fit;
    int nt = fit < n_groups ? fit : n_groups;              // column tiles per (m tile, slice), balanced split
    int gl = (n_groups + nt - 1) / nt;                     // live groups of the widest tile
    if (gl > g_max) { gl = g_max; nt = (n_groups + g_max - 1) / g_max; }
    int ng = ngs[n_ngs - 1];
    for (int i = 0; i < n_ngs; ++i)
      if (p.wn * ngs[i] / cap_div >= gl && (p.wn * ngs[i]) % cap_div == 0) { ng = ngs[i]; break; }
    const double rounds = (double)(((int64_t)m_tiles * nt * sl + 255) / 256);
    const int nk = (KT + sl - 1) / sl;
    // per K tile and workgroup: matrix-pipe cycles of a SIMD against the cycles its CU needs to pull the tile's weights
    // (~12 B / clk of HBM stream per CU) and activations (L2, ~40 B / clk)
    const double mfma = per_simd * p.mb * ng * 2 * 17.0, mem = gl * 2048.0 / 12.0 + p.wm * p.mb * 2048.0 / 40.0;
    double t = rounds * (nk * (mfma > mem ? mfma : mem) + 4000.0);
    if (sl > 1) t += (double)sl * M * N * 4 / 256.0 / 8.0 + 800.0;   // slab write + read-back, spread over the chip (800: in-step sweep at M = 32, round 3)
    if (t < best) { best = t; p.ng = ng; p.slices = sl; }
  }
  if (f_ng > 0) p.ng = f_ng;
  if (f_sl > 0 && can_slice) p.slices = f_sl;
#ifdef XM_TUNING
  for (int i = 0; i < n_shape_plans; ++i)   // per-shape overrides (tools/step_ab.py sweeps one GEMM of the step at a time)
    if (shape_plans[i].N == N && shape_plans[i].K == K) {
      if (shape_plans[i].ng > 0) p.ng = shape_plans[i].ng;
      if (shape_plans[i].slices > 0 && (can_slice || shape_plans[i].slices == 1)) p.slices = shape_plans[i].slices;
    }
#endif
  return p;
}

#define WS_CASE(NWV_, WM_, WN_, MB_, NG_, DW_)                                                                        \
  if (p.waves == NWV_ && p.wm == WM_ && p.wn == WN_ && p.mb == MB_ && p.ng == NG_)                                    \
    return ws_launch_cfg<KIND, NWV_, WM_, WN_, MB_, NG_, DW_>(A, Wp, M, N, K, p.slices, epi, slabs, s);

template <int KIND>
int ws_dispatch(const WsPlan& p, const void* A, const void* Wp, int64_t M, int64_t N, int64_t K, GemmEpi epi,
                       int32_t* slabs, hipStream_t s) {
  // (waves, WM, WN, MB, NG, DW): the ring of DW + 1 slots fills most of the 160 KiB of LDS (a CU needs of the order of 100 KiB
  // of requests in flight to pull its share of the HBM stream, MI355X_MICROARCH.md "ldsdma-fill"); a slot holds the tile's
  // weight fragments (2 KiB per column group) and activation fragments (2 KiB per 16-row block)
  WS_CASE(8, 4, 2, 4, 5, 2) WS_CASE(8, 4, 2, 4, 4, 2) WS_CASE(8, 4, 2, 4, 3, 2) WS_CASE(8, 4, 2, 4, 2, 3) WS_CASE(8, 4, 2, 4, 1, 3)
  WS_CASE(4, 2, 2, 4, 5, 3) WS_CASE(4, 2, 2, 4, 4, 4) WS_CASE(4, 2, 2, 4, 3, 4) WS_CASE(4, 2, 2, 4, 2, 5) WS_CASE(4, 2, 2, 4, 1, 6)
  WS_CASE(4, 1, 4, 4, 3, 4) WS_CASE(4, 1, 4, 4, 2, 5) WS_CASE(4, 1, 4, 4, 1, 8)
  WS_CASE(4, 1, 4, 2, 3, 4) WS_CASE(4, 1, 4, 2, 2, 6) WS_CASE(4, 1, 4, 2, 1, 8)
  return -1;
}

// Returns XM_ERR_UNSUPPORTED when the shape is outside the envelope. With epi.defer (int8) the exact int32 sums are left in
// `workspace` as *n_slabs slabs of M*N (the fused consumer adds them); otherwise the 16-bit result is written to epi.out.
template <int KIND>
static int launch_gemm_ws(const void* A, const void* Wp, int64_t M, int64_t N, int64_t K, GemmEpi epi, void* workspace,
                          size_t ws_bytes, int* n_slabs, hipStream_t s) {
  if (!epi_fits(epi, kCapGateUp | kCapDefer | kCapAccOut | ((KIND == kBF16 || KIND == kF16) ? kCapArgmax : 0u)))
    return XM_ERR_UNSUPPORTED;
  if (epi.argmax_val && (!epi.argmax_idx || epi.argmax_slots <= 0 || epi.gate_up || epi.defer || epi.acc_out)) return XM_ERR_INVALID;
  if (M <= 0 || M > 512 || N % 16 != 0 || K % WS_BK != 0 || K / WS_BK < 4 || M * K >= (1ll << 31) || epi.group_counts ||
      ((uintptr_t)A % 16) || ((uintptr_t)Wp % 16) || (epi.out && (uintptr_t)epi.out % 8) ||
      (KIND == kI8 && epi.w_scale && (uintptr_t)epi.w_scale % 16))
    return XM_ERR_UNSUPPORTED;
  if (N * K >= (1ll << 31) * 16ll) return XM_ERR_UNSUPPORTED;
  const bool need_slab = epi.defer != 0;
  if (need_slab && (KIND != kI8 || !workspace || ws_bytes < (size_t)M * N * 4)) return XM_ERR_WORKSPACE;
  const bool can_slice = !epi.argmax_val && workspace && ws_bytes >= (size_t)2 * M * N * 4;
  if (epi.gate_up && (KIND == kFP8 || N % 32 != 0 || !epi.act_out || ((uintptr_t)epi.act_out % 16) || N * K >= (1ll << 31) ||
                      (KIND == kI8 && (!epi.row_amax || !epi.a_scale || !epi.w_scale))))
    return XM_ERR_UNSUPPORTED;
  WsPlan p = ws_plan(M, N, K, can_slice, ws_bytes, epi.gate_up != 0);
  int32_t* const slabs = reinterpret_cast<int32_t*>(workspace);
  GemmEpi e2 = epi;
  if (need_slab && p.slices == 1) {  // one slab = the kernel's raw-accumulator output
    e2.acc_out = slabs;
    e2.out = nullptr;
  }
  const int slices = ws_dispatch<KIND>(p, A, Wp, M, N, K, e2, slabs, s);
  if (slices == -2) return XM_ERR_WORKSPACE;
  if (slices < 0) return XM_ERR_UNSUPPORTED;
  if (n_slabs) *n_slabs = slices;
  if (slices > 1 && !need_slab) {
    int64_t blocks = (M * N / 4 + 255) / 256;
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(ws_slab_epilogue_kernel<KIND>, dim3((unsigned)blocks), dim3(256), 0, s, slabs, slices, M, N, epi);
  }
  return hip_check_launch();
}

int launch_gemm_ws_i8(const void* A, const void* Wp, int64_t M, int64_t N, int64_t K, GemmEpi epi, void* workspace,
                      size_t ws_bytes, int* n_slabs, hipStream_t s) {
  return launch_gemm_ws<kI8>(A, Wp, M, N, K, epi, workspace, ws_bytes, n_slabs, s);
}
int launch_gemm_ws_fp8(const void* A, const void* Wp, int64_t M, int64_t N, int64_t K, GemmEpi epi, void* workspace,
                       size_t ws_bytes, hipStream_t s) {
  return launch_gemm_ws<kFP8>(A, Wp, M, N, K, epi, workspace, ws_bytes, nullptr, s);
}
// 16-bit weights: Kb = K * 2 BYTES per row (the kernel and the packing work on bytes)
int launch_gemm_ws_h16(const void* A, const void* Wp, int64_t M, int64_t N, int64_t Kb, GemmEpi epi, void* workspace,
                       size_t ws_bytes, hipStream_t s) {
  if (epi.out_bf16) return launch_gemm_ws<kBF16>(A, Wp, M, N, Kb, epi, workspace, ws_bytes, nullptr, s);
  return launch_gemm_ws<kF16>(A, Wp, M, N, Kb, epi, workspace, ws_bytes, nullptr, s);
}

// finishing launches of the fused greedy sampling: partial (max, first index) pairs [slots][M]. One workgroup per (16 rows, chunk of
// the slots): lane (tid & 15) = row, (tid >> 4) = one of 16 slots per sweep (every load instruction reads 16 x 128 contiguous
// bytes); the 16 threads of a row meet in the LDS. Two levels -- [slots] -> [chunks] -> 1 -- because one workgroup per 16 rows alone
// walks ~120 dependent sweeps (measured: 60 us at lm_head's 1902 slots, more than the fusion saves).
__global__ __launch_bounds__(256) void argmax_finish_kernel(const uint2* __restrict__ part, int slots, int M, int per_chunk,
                                                           uint2* __restrict__ part_out, int64_t* __restrict__ out_idx,
                                                           float* __restrict__ out_val) {
  __shared__ float sv[16][17];
  __shared__ int si[16][17];
  const int tid = threadIdx.x, r = tid & 15, j0 = tid >> 4;
  const int m = blockIdx.x * 16 + r;
  const int lo = blockIdx.y * per_chunk, hi = lo + per_chunk < slots ? lo + per_chunk : slots;
  float bv = -__builtin_inff();
  int bi = 0x7fffffff;
  if (m < M) {
    int j = lo + j0;
    for (; j + 48 < hi; j += 64) {   // four independent loads in flight
      const uint2 p0 = part[(int64_t)j * M + m], p1 = part[(int64_t)(j + 16) * M + m], p2 = part[(int64_t)(j + 32) * M + m],
                  p3 = part[(int64_t)(j + 48) * M + m];
      if (argmax_better(__uint_as_float(p0.x), (int)p0.y, bv, bi)) { bv = __uint_as_float(p0.x); bi = (int)p0.y; }
      if (argmax_better(__uint_as_float(p1.x), (int)p1.y, bv, bi)) { bv = __uint_as_float(p1.x); bi = (int)p1.y; }
      if (argmax_better(__uint_as_float(p2.x), (int)p2.y, bv, bi)) { bv = __uint_as_float(p2.x); bi = (int)p2.y; }
      if (argmax_better(__uint_as_float(p3.x), (int)p3.y, bv, bi)) { bv = __uint_as_float(p3.x); bi = (int)p3.y; }
    }
    for (; j < hi; j += 16) {
      const uint2 p = part[(int64_t)j * M + m];
      if (argmax_better(__uint_as_float(p.x), (int)p.y, bv, bi)) { bv = __uint_as_float(p.x); bi = (int)p.y; }
    }
  }
  sv[j0][r] = bv;
  si[j0][r] = bi;
  __syncthreads();
  if (tid < 16 && m < M) {
    for (int w = 1; w < 16; ++w)
      if (argmax_better(sv[w][r], si[w][r], bv, bi)) { bv = sv[w][r]; bi = si[w][r]; }
    if (part_out) {
      part_out[(int64_t)blockIdx.y * M + m] = make_uint2(__float_as_uint(bv), (unsigned)bi);
    } else {
      out_idx[m] = bi == 0x7fffffff ? 0 : bi;
      if (out_val) out_val[m] = bv;
    }
  }
}

// matmul + argmax(-1) on packed 16-bit weights: workspace = [M][slots] floats, then [M][slots] int32 (slots = N / 16, the bound)
int launch_gemm_ws_h16_argmax(const void* A, const void* Wp, int64_t M, int64_t N, int64_t Kb, GemmEpi epi, int64_t* out_idx,
                              float* out_val, void* workspace, size_t ws_bytes, hipStream_t s) {
  // slots of a launch = column tiles x waves across a tile: <= N / 16 / NG for the wide problems, <= 256 tiles x 4 waves otherwise
  const int64_t slots = N / 16 > 1024 ? N / 16 : 1024;
  if (!workspace || ws_bytes < (size_t)M * slots * 8) return XM_ERR_WORKSPACE;
  epi.out = nullptr;
  epi.argmax_val = reinterpret_cast<float*>(workspace);
  epi.argmax_idx = reinterpret_cast<int32_t*>(epi.argmax_val + M * slots);
  epi.argmax_slots = (int)slots;
  g_argmax_slots_used = 0;
  const int rc = epi.out_bf16 ? launch_gemm_ws<kBF16>(A, Wp, M, N, Kb, epi, nullptr, 0, nullptr, s)
                              : launch_gemm_ws<kF16>(A, Wp, M, N, Kb, epi, nullptr, 0, nullptr, s);
  if (rc != XM_OK) return rc;
  const uint2* part = reinterpret_cast<const uint2*>(epi.argmax_val);
  const int used = g_argmax_slots_used, rb = (int)((M + 15) / 16);
  if (used > 64) {          // level 1: 64-slot chunks -> the tail of the workspace ([chunks][M], behind the used slots)
    const int per = 64, chunks = (used + per - 1) / per;
    uint2* part2 = const_cast<uint2*>(part) + (int64_t)used * M;
    if ((int64_t)(used + chunks) > slots) return XM_ERR_WORKSPACE;
    hipLaunchKernelGGL(argmax_finish_kernel, dim3((unsigned)rb, (unsigned)chunks), dim3(256), 0, s, part, used, (int)M, per, part2,
                       nullptr, nullptr);
    hipLaunchKernelGGL(argmax_finish_kernel, dim3((unsigned)rb, 1), dim3(256), 0, s, part2, chunks, (int)M, chunks, nullptr, out_idx,
                       out_val);
  } else {
    hipLaunchKernelGGL(argmax_finish_kernel, dim3((unsigned)rb, 1), dim3(256), 0, s, part, used, (int)M, used, nullptr, out_idx,
                       out_val);
  }
  return hip_check_launch();
}

int launch_pack_weight_i8(const void* W, void* Wp, int64_t N, int64_t K, hipStream_t s) {
  if (N % 16 != 0 || K % WS_BK != 0 || ((uintptr_t)W % 16) || ((uintptr_t)Wp % 16)) return XM_ERR_UNSUPPORTED;
  int64_t blocks = (N * (K / 16) + 255) / 256;
  blocks = blocks > 65536 ? 65536 : blocks;
  hipLaunchKernelGGL(pack_weight_i8_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t*)W, (uint8_t*)Wp, N, K);
  return hip_check_launch();
}

}  // namespace xm

#ifdef WS8_TIMING
extern "C" __attribute__((visibility("default"))) int xllm_mi355_debug_ws8(long long* out64) {
  return hipMemcpyFromSymbol(out64, HIP_SYMBOL(xm::ws8_dbg), 64 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif
extern "C" XM_API void xllm_mi355_gemm_plan_hint(int ng, int slices, int tile_rows) {
  xm::f_ng = ng > 0 ? ng : 0;
  xm::f_sl = slices > 0 ? slices : 0;
  xm::f_rows = tile_rows == 128 || tile_rows == 256 ? tile_rows : 0;
}
#ifdef XM_TUNING
// tuning: plan override for ONE problem shape (N, K) of the following launches; N <= 0 clears all of them
extern "C" XM_API void xllm_mi355_debug_ws_plan_shape(long long N, long long K, int ng, int slices) {
  if (N <= 0) { xm::n_shape_plans = 0; return; }
  for (int i = 0; i < xm::n_shape_plans; ++i)
    if (xm::shape_plans[i].N == N && xm::shape_plans[i].K == K) { xm::shape_plans[i].ng = ng; xm::shape_plans[i].slices = slices; return; }
  if (xm::n_shape_plans < 8) xm::shape_plans[xm::n_shape_plans++] = {N, K, ng, slices};
}
// tuning: 80 / 81 = eight waves in phase / one barrier apart (default); 140 .. 142 = cache policy of the weight stream; 0 = defaults
extern "C" XM_API void xllm_mi355_debug_ws_waves(int waves) {
  if (waves == 80 || waves == 81) xm::f_ws8s = waves - 80;
  if (waves >= 140 && waves <= 142) xm::f_wpolicy = waves - 140;
  if (waves == 150 || waves == 151) xm::f_slab_nt = waves - 150;
  if (waves == 0) { xm::f_wpolicy = 0; xm::f_ws8s = 1; xm::f_slab_nt = 1; }
}
#endif
