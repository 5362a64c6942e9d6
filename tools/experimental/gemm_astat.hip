// gemm_astat.hip -- "activation-stationary" int8 kernel for decode GEMMs with few columns (128 < M <= 256; qkv / o / down).
//
// Why: the skinny kernel (gemm.hip) is bound by its per-K-step chain (global -> VGPR -> LDS -> barrier -> fragment reads,
// one barrier per 128 bytes of K) and by every workgroup re-reading the whole activation matrix for 32..160 columns
// (profiles/r01_gemm_notes.txt). Here
//   * a workgroup owns 128 columns x a K range; the activations of a 256-byte K slab (256 rows x 256 B = 64 KB) are DMA'd
//     into LDS once (two slabs in flight), 16-byte chunks XOR-swizzled by (row & 15) so that the 16 rows of a B-operand
//     fragment read hit 16 different bank groups;
//   * the eight waves split the COLUMNS (16 each, all 256 rows): v_mfma_i32_16x16x64_i8 with the weight fragment as the A
//     operand, so a weight byte is used by exactly one wave and goes HBM -> VGPR directly in fragment order (lane (r, g)
//     loads bytes [64 j + 16 g, +16) of weight row n0 + r: a row's 256-byte slab is consumed whole by one wave) -- the
//     weight leg never touches the LDS write port and there is one barrier pair per 256 bytes of K;
//   * K is split across workgroups until the chip is full; the exact int32 partial sums are added into the zero-at-rest
//     split-K workspace (the same contract as the skinny kernel's SPLITK path: i8_splitk_epilogue_zero_kernel or the fused
//     add + norm consumer reads and re-zeroes it).
// LDS reads are inline asm with counted lgkmcnt waits (the compiler would drain the in-flight DMA before its own ds_reads).
#include "gemm_types.h"

namespace xm {

#define AS_DSR128(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define AS_LGKM1(N_, A_) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(A_) : "n"(N_))

constexpr int AS_SLAB = 256, AS_ROWS = 256, AS_BUF = AS_ROWS * AS_SLAB, AS_COLS = 128;

__global__ __launch_bounds__(512, 1) void gemm_astat_i8_kernel(const uint8_t* __restrict__ A,
                                                              const uint8_t* __restrict__ W, int M, int N, int Kb,
                                                              int slabs_per_split, int32_t* __restrict__ acc_out) {
  __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * AS_BUF];
  typedef __attribute__((address_space(3))) uint8_t* lds_ptr_t;
  const lds_ptr_t lds3 = (lds_ptr_t)lds;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.x * AS_COLS + wave * 16;
  const int total_slabs = Kb / AS_SLAB;
  const int sb = blockIdx.y * slabs_per_split;
  int ns = total_slabs - sb;
  ns = ns < slabs_per_split ? ns : slabs_per_split;
  if (ns <= 0) return;

  const __amdgpu_buffer_rsrc_t rsrc_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(A), 0, (int)((int64_t)M * Kb), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(W), 0, (int)((int64_t)N * Kb), 0x00020000);
  // DMA instruction i of wave w fills LDS bytes [(8 w + i) KB, +1 KB) = rows 4 (8 w + i) .. +3; lane's 16 bytes are physical
  // chunk (lane & 15) of row 4 (8 w + i) + (lane >> 4) and come from logical chunk (physical ^ (row & 15)); rows >= M are
  // clamped (their sums are never written)
  int voff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (wave * 8 + i) * 4 + (lane >> 4);
    const int c = (lane & 15) ^ (row & 15);
    const int ar = row < M ? row : M - 1;
    voff[i] = ar * Kb + c * 16;
  }
  int wn = n0 + r;
  wn = wn < N ? wn : N - 1;
  const int w_off = wn * Kb + g * 16;
  auto stage = [&](int t, int buf) {  // slabs past the range re-load the last one (keeps the vmcnt arithmetic uniform)
    const int sl = sb + (t < ns ? t : ns - 1);
    const lds_ptr_t dst = lds3 + buf * AS_BUF + wave * 8192;
#pragma unroll
    for (int i = 0; i < 8; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, dst + i * 1024, 16, voff[i], sl * AS_SLAB, 0, 0);
  };
  u32x4 w0[4], w1[4];
#define AS_LOADW(T_, WR)                                                                            \
  {                                                                                                 \
    const int sl_ = sb + ((T_) < ns ? (T_) : ns - 1);                                               \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
        WR[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_off, sl_ * AS_SLAB + j * 64, 0);    \
  }
  i32x4_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = i32x4_t{0, 0, 0, 0};

  const unsigned lds_base = (unsigned)(__UINTPTR_TYPE__)lds3;
  unsigned a_rd[4];  // fragment read address of k64 block j: row r (+ 16 mb rows = mb * 4096 bytes), chunk (4 j + g) ^ r
#pragma unroll
  for (int j = 0; j < 4; ++j) a_rd[j] = lds_base + r * AS_SLAB + (((4 * j + g) ^ r) << 4);

  // item I: k64 block j = I >> 4, row block mb = I & 15; 8 fragment reads in flight
#define AS_RD(I_, AD) AS_DSR128(af[(I_) & 7], AD[(I_) >> 4], ((I_) & 15) * 4096);
#define AS_IT(I_, AD, WR)                                                                            \
  AS_LGKM1(((I_) + 8 < 64 ? 7 : 63 - (I_)), af[(I_) & 7]);                                           \
  acc[(I_) & 15] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4_t, WR[(I_) >> 4]), \
                                                         __builtin_bit_cast(i32x4_t, af[(I_) & 7]), acc[(I_) & 15], 0, 0, 0); \
  if ((I_) + 8 < 64) { AS_RD(((I_) + 8 < 64 ? (I_) + 8 : 63), AD) }
#define AS_IT8(B_, AD, WR) AS_IT(B_, AD, WR) AS_IT(B_ + 1, AD, WR) AS_IT(B_ + 2, AD, WR) AS_IT(B_ + 3, AD, WR) \
  AS_IT(B_ + 4, AD, WR) AS_IT(B_ + 5, AD, WR) AS_IT(B_ + 6, AD, WR) AS_IT(B_ + 7, AD, WR)
#define AS_COMPUTE(BUF_, WR)                                                                         \
  {                                                                                                  \
    unsigned ad[4];                                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) ad[j] = a_rd[j] + (BUF_) * AS_BUF;                 \
    u32x4 af[8];                                                                                     \
    AS_RD(0, ad) AS_RD(1, ad) AS_RD(2, ad) AS_RD(3, ad) AS_RD(4, ad) AS_RD(5, ad) AS_RD(6, ad) AS_RD(7, ad) \
    AS_IT8(0, ad, WR) AS_IT8(8, ad, WR) AS_IT8(16, ad, WR) AS_IT8(24, ad, WR)                        \
    AS_IT8(32, ad, WR) AS_IT8(40, ad, WR) AS_IT8(48, ad, WR) AS_IT8(56, ad, WR)                      \
  }
#define AS_STEP(T_, WR)                                                                              \
  asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); /* slab T_ (DMA + weights) landed, slab T_ + 1 still flies */ \
  __builtin_amdgcn_s_barrier();                                                                      \
  AS_COMPUTE((T_) & 1, WR)                                                                           \
  __builtin_amdgcn_s_barrier();                    /* every wave is done with this buffer */         \
  stage((T_) + 2, (T_) & 1);                                                                         \
  AS_LOADW((T_) + 2, WR)

  stage(0, 0);
  AS_LOADW(0, w0)
  stage(1, 1);
  AS_LOADW(1, w1)
  for (int t = 0; t < ns; t += 2) {
    AS_STEP(t, w0)
    if (t + 1 >= ns) break;
    AS_STEP(t + 1, w1)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef AS_STEP
#undef AS_COMPUTE
#undef AS_IT8
#undef AS_IT
#undef AS_RD
#undef AS_LOADW

  // D[i][j]: i = weight row (column n) = 4 g + e, j = activation row m = lane & 15
  const int nb = n0 + 4 * g;
#pragma unroll
  for (int mb = 0; mb < 16; ++mb) {
    const int m = mb * 16 + r;
    if (m >= M) continue;
    int32_t* dst = acc_out + (int64_t)m * N + nb;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (nb + e < N) atomicAdd(dst + e, acc[mb][e]);
  }
}

int launch_gemm_astat_i8(const void* A, const void* W, int64_t M, int64_t N, int64_t Kb, int32_t* acc_ws, int splits,
                         hipStream_t s) {
  if (M < 1 || M > AS_ROWS || Kb % AS_SLAB != 0 || M * Kb >= (1ll << 31) || N * Kb >= (1ll << 31) || !acc_ws)
    return XM_ERR_UNSUPPORTED;
  const int slabs = (int)(Kb / AS_SLAB);
  splits = splits < 1 ? 1 : (splits > slabs ? slabs : splits);
  const int per = (slabs + splits - 1) / splits;
  splits = (slabs + per - 1) / per;
  const dim3 grid((unsigned)((N + AS_COLS - 1) / AS_COLS), (unsigned)splits);
  hipLaunchKernelGGL(gemm_astat_i8_kernel, grid, dim3(512), 0, s, (const uint8_t*)A, (const uint8_t*)W, (int)M, (int)N,
                     (int)Kb, per, acc_ws);
  return hip_check_launch();
}

}  // namespace xm
