// gemm_types.h -- definitions shared by the GEMM translation units (gemm.hip, gemm_p8.hip): MFMA wrappers per operand
// kind, the fused-epilogue descriptor and small helpers.
#pragma once
#include <type_traits>

#include "common.h"

namespace xm {

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef __bf16 gbf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 gf16x8_t __attribute__((ext_vector_type(8)));

enum GemmKind { kI8 = 0, kFP8 = 1, kBF16 = 2, kF16 = 3 };

template <int KIND>
struct MmaTraits;
template <>
struct MmaTraits<kI8> {
  using acc_t = i32x16_t;
  static __device__ __forceinline__ acc_t zero() { return acc_t{0}; }
  static __device__ __forceinline__ acc_t mma(const uint4& a, const uint4& b, acc_t c) {
    i32x4_t av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w}, bv = {(int)b.x, (int)b.y, (int)b.z, (int)b.w};
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
  }
};
template <>
struct MmaTraits<kFP8> {
  using acc_t = f32x16_t;
  static __device__ __forceinline__ acc_t zero() { return acc_t{0}; }
  static __device__ __forceinline__ acc_t mma(const uint4& a, const uint4& b, acc_t c) {
    // 16 fp8 per lane = two K=16 MFMAs (the k permutation is identical on both operands)
    long a0 = (long)(((unsigned long)a.y << 32) | a.x), a1 = (long)(((unsigned long)a.w << 32) | a.z);
    long b0 = (long)(((unsigned long)b.y << 32) | b.x), b1 = (long)(((unsigned long)b.w << 32) | b.z);
    c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a0, b0, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a1, b1, c, 0, 0, 0);
  }
};
template <>
struct MmaTraits<kBF16> {
  using acc_t = f32x16_t;
  static __device__ __forceinline__ acc_t zero() { return acc_t{0}; }
  static __device__ __forceinline__ acc_t mma(const uint4& a, const uint4& b, acc_t c) {
    gbf16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
  }
};
template <>
struct MmaTraits<kF16> {
  using acc_t = f32x16_t;
  static __device__ __forceinline__ acc_t zero() { return acc_t{0}; }
  static __device__ __forceinline__ acc_t mma(const uint4& a, const uint4& b, acc_t c) {
    gf16x8_t av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
  }
};

struct GemmEpi {
  const float* a_scale;   // int8: [M]; fp8: [1] or [M]
  int64_t a_scale_n;
  const float* w_scale;   // int8: [N]; fp8: [1] or [N]
  int64_t w_scale_n;
  const void* bias;       // out dtype, [N] or null
  void* out;              // 16-bit out [M,N] (may be null when only acc_out is wanted)
  int32_t* acc_out;       // int8: raw accumulators [M,N] (null normally); split-K workspace
  int out_bf16;           // 1 bf16, 0 f16
  const int32_t* group_counts;  // grouped GEMM (MoE): rows per expert, DEVICE array [n_groups]; null otherwise
  int n_groups;
  int defer;  // int8: leave the exact int32 sums in the (zeroed) split-K workspace, no dequant epilogue launch
  // grouped GEMM on the 256x256 kernels: DEVICE table built by group_plan_kernel, one int4 per m-tile slot =
  // (expert or -1, first row of the expert, rows of the expert, tile index inside the expert); null otherwise
  const int32_t* group_tiles;
  // grouped GEMM with the expand fused in (layers/dcu/fused_moe.cpp:195-197: index_select(hidden, dst_src / topk)):
  // sorted row r of the grouped problem is row gather_rows[r] / gather_div of A (A = the un-expanded activations)
  const int32_t* gather_rows;
  int gather_div;
  int gather_src_rows;  // rows of the un-expanded A
  // round 3: gate_up projection with the SiLU.mul of DenseMLP fused into the epilogue (dense_mlp.cpp:97-116: gate_up_proj ->
  // act_fn(gate) * up; linear.cpp:481-507). N = 2 I, weight rows [0, I) = gate, [I, 2 I) = up. The kernel computes the gate and
  // the up columns of the SAME act columns in one workgroup, writes act = rT(rT(silu(g)) * u) [M, I] (16 bit) to act_out and
  // folds each row's |max| into row_amax (atomic max on the bits of a non-negative float; zero at rest) for the per-token int8
  // quantisation that follows (xllm_mi355_quantize_with_row_amax). epi.out is unused in this mode.
  int gate_up;
  void* act_out;
  float* row_amax;
  int w_policy;   // packed kernels, cache policy of the weight stream: 0 = by the launch shape, 1 = nt, 2 = default (tuning arm)
  // round 4: greedy sampling fused into the lm_head GEMM (Sampler::greedy_sample = argmax(-1), framework/sampling/sampler.cpp:
  // 160-168; the logits of ColumnParallelLinear lm_head, linear.cpp:512-520). Every wave reduces the 16-bit-rounded results of
  // its NG column groups to one (max, first index) pair per row and writes it to slot `nt * WN + wn` of the partial array
  // uint2 [argmax_slots][M] at argmax_val (argmax_idx only marks the mode); xllm_mi355_matmul_argmax_packed's finishing launch
  // reduces the slots. The [M, N] logits are never written
  // (epi.out may be null). argmax commutes with the column tiling; NaN > everything and the first index wins, as torch.argmax.
  float* argmax_val;
  int32_t* argmax_idx;
  int argmax_slots;
  int slab_nt;   // packed kernels: K-slice slabs leave with non-temporal stores (round 4, see ws_epilogue)
  int out_nt;    // 8-phase int8 kernel: non-temporal output stores (outputs larger than the L2; set by launch_gemm_p8i)
  // round 4: ScaledMatmulParams::c with alpha = beta = 1 (kernels/param.h:852-866, "alpha * (a @ b) + beta * c"): a 16-bit [M, N]
  // addend folded into the dequant epilogue, out = rT(rT(acc * a_scale * w_scale + bias) + addend) -- the 16-bit GEMM result,
  // then a 16-bit add, i.e. bit-identical to scaled_matmul followed by the residual add of fused_add_rms_norm. May alias `out`
  // (every element is read by the lane that then writes it). 8-phase int8 kernel only.
  const void* addend;
  // round 6: grouped GEMM on the 256x256 kernels WITHOUT the plan launch -- every workgroup derives its slot from group_counts
  // itself (group_locate below; n_groups <= 256). group_tiles still points at valid scratch (unused) so the mode checks stay one test.
  int group_inline;
};

// torch.argmax order: NaN above every number, the first index among equals
__device__ __forceinline__ bool argmax_better(float v, int i, float bv, int bi) {
  const bool vn = v != v, bn = bv != bv;
  if (vn || bn) return vn && (!bn || i < bi);
  return v > bv || (v == bv && i < bi);
}

// ---- grouped (MoE) mode of the 256x256 8-phase kernels: which (m-tile slot, n tile) a block owns, and the slot's expert.
// The live m-tile slots are COMPACT (experts in order, ceil(rows / bm) slots each) and how many there are is known on the device
// only. The live (m slot, n tile) units, n tile fastest, are cut into eight equal contiguous ranges, one per XCD (block b runs on
// XCD b % 8): every XCD gets the same number of workgroups whatever the routing, the n tiles of an m slot -- one gathered
// activation panel -- and an expert's m slots -- one weight panel -- stay neighbours on one XCD, and the surplus blocks of the
// worst-case grid exit at once (round 6; before, the super-block walk put cfg5's 32-40 live m tiles of 48 slots on XCDs 0-3 and
// 6-7 in two rounds). Slot lookup: from the table group_plan_kernel built, or -- group_inline, n_groups <= 256 -- straight from
// the expert sizes: every wave redundantly scans them in registers (4 experts per lane, one wave scan), no plan launch.
struct GroupSlot { int e, off, cnt, tile; };
__device__ __forceinline__ bool group_locate(const GemmEpi& epi, int m_slots, int n_tiles, int bm, int lane, int& mt, int& nt,
                                             GroupSlot& gs) {
  const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
  int used;
  int ct = 0, cr = 0, it = 0, ir = 0;      // inline: this lane's tiles / rows, inclusive scans over the lanes
  int c4[4] = {0, 0, 0, 0};
  if (epi.group_inline) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = lane * 4 + q;
      c4[q] = e < epi.n_groups ? epi.group_counts[e] : 0;
      c4[q] = c4[q] > 0 ? c4[q] : 0;
      ct += (c4[q] + bm - 1) / bm;
      cr += c4[q];
    }
    it = ct;
    ir = cr;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int tt = __shfl_up(it, o), tr = __shfl_up(ir, o);
      if (lane >= o) { it += tt; ir += tr; }
    }
    used = __shfl(it, 63);
    used = used < m_slots ? used : m_slots;
  } else {
    used = epi.group_tiles[4 * m_slots];
  }
  used = __builtin_amdgcn_readfirstlane(used);
  const int live = used * n_tiles;
  const int lo = (int)((int64_t)xcd * live / 8), hi = (int)((int64_t)(xcd + 1) * live / 8);
  if (lo + j >= hi) return false;
  mt = (lo + j) / n_tiles;
  nt = (lo + j) % n_tiles;
  if (epi.group_inline) {
    const int ex_t = it - ct;                 // slots before this lane's experts
    const bool mine = mt >= ex_t && mt < it;
    const int owner = __builtin_ctzll(__ballot(mine));   // (mt < used: exactly one lane owns it)
    int e = 0, off = ir - cr, cnt = 0, tile = 0;
    if (mine) {
      int s = ex_t;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int tq = (c4[q] + bm - 1) / bm;
        if (mt >= s && mt < s + tq) { e = lane * 4 + q; cnt = c4[q]; tile = mt - s; break; }
        s += tq;
        off += c4[q];
      }
    }
    gs.e = __shfl(e, owner);
    gs.off = __shfl(off, owner);
    gs.cnt = __shfl(cnt, owner);
    gs.tile = __shfl(tile, owner);
  } else {
    const int4 gt = reinterpret_cast<const int4*>(epi.group_tiles)[mt];
    gs.e = gt.x; gs.off = gt.y; gs.cnt = gt.z; gs.tile = gt.w;
  }
  gs.e = __builtin_amdgcn_readfirstlane(gs.e);
  gs.off = __builtin_amdgcn_readfirstlane(gs.off);
  gs.cnt = __builtin_amdgcn_readfirstlane(gs.cnt);
  gs.tile = __builtin_amdgcn_readfirstlane(gs.tile);
  return gs.e >= 0;
}

// Epilogue modes a launcher can honour. Every launcher starts with epi_fits(epi, its capabilities): a mode the selected kernel
// lacks is DECLINED (XM_ERR_UNSUPPORTED) -- never dropped with XM_OK (round-3 review: the removed 32x32x32 int8 arm returned OK
// with nothing written in gate_up mode).
enum EpiCap : unsigned { kCapGateUp = 1, kCapDefer = 2, kCapGroupTiles = 4, kCapGather = 8, kCapGroupCounts = 16, kCapAccOut = 32,
                         kCapArgmax = 64, kCapAddend = 128 };
inline bool epi_fits(const GemmEpi& e, unsigned caps) {
  const unsigned need = (e.gate_up || e.act_out || e.row_amax ? kCapGateUp : 0u) | (e.defer ? kCapDefer : 0u) |
                        (e.group_tiles ? kCapGroupTiles : 0u) | (e.gather_rows ? kCapGather : 0u) |
                        (e.group_counts && !e.group_tiles ? kCapGroupCounts : 0u) | (e.acc_out ? kCapAccOut : 0u) |
                        (e.argmax_val || e.argmax_idx ? kCapArgmax : 0u) | (e.addend ? kCapAddend : 0u);
  if (!e.out && !e.acc_out && !e.defer && !(e.gate_up && e.act_out) && !(e.argmax_val && e.argmax_idx)) return false;   // nowhere to write
  return (need & ~caps) == 0;
}

__device__ __forceinline__ void store16(void* out, int64_t idx, float v, int out_bf16) {
  if (out_bf16) reinterpret_cast<uint16_t*>(out)[idx] = f32_to_bf16_bits(v);
  else reinterpret_cast<f16_t*>(out)[idx] = (f16_t)v;
}
__device__ __forceinline__ float load16(const void* p, int64_t idx, int is_bf16) {
  if (is_bf16) return bf16_bits_to_f32(reinterpret_cast<const uint16_t*>(p)[idx]);
  return (float)reinterpret_cast<const f16_t*>(p)[idx];
}

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// ---- helpers of the hand-pipelined kernels (gemm_p8.hip, gemm_p8i.hip)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ typename MmaTraits<KIND>::acc_t mma4(const u32x4 a, const u32x4 b,
                                                                typename MmaTraits<KIND>::acc_t c) {
  if constexpr (KIND == kI8) {
    i32x4_t av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w}, bv = {(int)b.x, (int)b.y, (int)b.z, (int)b.w};
#ifdef P8_ABL_MFMA16  /* ablation build (timing / power only, WRONG results): the same MACs as 2 x 16x16x64 */
    i32x4_t c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
    c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, c1, 0, 0, 0);
    c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; c[3] = c0[3];
    c[4] = c1[0]; c[5] = c1[1]; c[6] = c1[2]; c[7] = c1[3];
    return c;
#else
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
#endif
  } else if constexpr (KIND == kFP8) {
    long a0 = (long)(((unsigned long)a.y << 32) | a.x), a1 = (long)(((unsigned long)a.w << 32) | a.z);
    long b0 = (long)(((unsigned long)b.y << 32) | b.x), b1 = (long)(((unsigned long)b.w << 32) | b.z);
    c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a0, b0, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a1, b1, c, 0, 0, 0);
  } else if constexpr (KIND == kBF16) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gbf16x8_t, a), __builtin_bit_cast(gbf16x8_t, b),
                                                   c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(gf16x8_t, a), __builtin_bit_cast(gf16x8_t, b), c,
                                                  0, 0, 0);
  }
}

// fp8 (OCP e4m3) on the gfx950 rate: one v_mfma_f32_32x32x64_f8f6f4 consumes 32 bytes of K per lane and operand (two 16-byte
// fragments) -- 1024 MAC/clk/SIMD, twice the rate of the two v_mfma_f32_32x32x16_fp8_fp8 it replaces. Zero scale operands
// select the unscaled encoding (scales of 1). Lane (row, h = lane >> 5) of BOTH operands holds the same 32 K-bytes of its
// row, so any fragment pair works as long as A and B use the same pair.
typedef int gi32x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16_t mma_fp8x2(const u32x4 a0, const u32x4 a1, const u32x4 b0, const u32x4 b1, f32x16_t c) {
  const gi32x8_t av = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
  const gi32x8_t bv = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, 0, 0, 0);
}

// 16-bit output conversion without branches (same bits as f32_to_bf16_bits / the f16 cast of common.h)
__device__ __forceinline__ unsigned pack16(float v, bool out_bf16) {
  const unsigned bf = f32_to_bf16_bits(v);
  const f16_t hv = (f16_t)v;
  uint16_t hb;
  __builtin_memcpy(&hb, &hv, 2);
  return out_bf16 ? (bf & 0xffffu) : (unsigned)hb;
}

// two fp32 -> one packed pair of 16-bit outputs (RNE): v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 on gfx950
template <bool OUT_BF16>
__device__ __forceinline__ unsigned pack2x16(float a, float b) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  if constexpr (OUT_BF16) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  } else {
    typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  }
}

#define P8_DSR(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define P8_WAIT4(F)                                                                                             \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]))
#define P8_WAIT8(F)                                                                                             \
  asm volatile("s_waitcnt lgkmcnt(0)"                                                                           \
               : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(F[4]), "+v"(F[5]), "+v"(F[6]), "+v"(F[7]))

// 256x256 8-phase kernel (gemm_p8.hip). Returns XM_ERR_UNSUPPORTED when the shape is outside its envelope.
template <int KIND>
int launch_gemm_p8(const void* A, const void* W, int64_t M, int64_t N, int64_t Kb, GemmEpi epi, void* workspace,
                   size_t ws_bytes, int splits, hipStream_t s);

// weight-stream decode kernel on pre-packed int8 weights (gemm_ws.hip)
int launch_gemm_ws_i8(const void* A, const void* Wp, int64_t M, int64_t N, int64_t K, GemmEpi epi, void* workspace,
                      size_t ws_bytes, int* n_slabs, hipStream_t s);
int launch_gemm_ws_fp8(const void* A, const void* Wp, int64_t M, int64_t N, int64_t K, GemmEpi epi, void* workspace,
                       size_t ws_bytes, hipStream_t s);
int launch_gemm_ws_h16(const void* A, const void* Wp, int64_t M, int64_t N, int64_t Kb, GemmEpi epi, void* workspace,
                       size_t ws_bytes, hipStream_t s);
int launch_gemm_ws_h16_argmax(const void* A, const void* Wp, int64_t M, int64_t N, int64_t Kb, GemmEpi epi, int64_t* out_idx,
                              float* out_val, void* workspace, size_t ws_bytes, hipStream_t s);
int launch_pack_weight_i8(const void* W, void* Wp, int64_t N, int64_t K, hipStream_t s);

// gemm_wsb.hip: weight-stream GEMM for 16-bit weights at decode shapes (dense M <= 64, grouped with few rows per expert)
template <typename T>
int launch_gemm_wsb_dense(const void* x, const void* w, const void* bias, void* out, int64_t M, int64_t N, int64_t K,
                          void* workspace, size_t ws_bytes, hipStream_t s);
template <typename T>
int launch_gemm_wsb_grouped(const void* x, const void* w, const int32_t* counts, void* out, int64_t max_rows,
                            int64_t n_experts, int64_t N, int64_t K, const int32_t* row_index, int64_t index_div,
                            hipStream_t s);

}  // namespace xm
