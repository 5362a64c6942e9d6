// common.h -- device-side helpers shared by the gfx950 kernels (wave = 64 lanes everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/xllm_mi355.h"

namespace xm {

constexpr int kWave = 64;

// ---- run-time switches (host side). The product library reads only the switches of DESIGN section 4.7 through xm_switch()
// (once per process, cached by the caller). Tuning overrides -- A/B arms whose verdict is recorded, planner constants, sweeps --
// are XM_TUNE_VAR: compile-time constants here, environment-backed mutable variables only in the -DXM_TUNING flavour of the
// library that `make tuning` builds for tools/ (lib/libxllm_mi355_tuning.so, never loaded by the product or the tests).
inline int xm_switch(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
#ifdef XM_TUNING
#define XM_TUNE_VAR(var, name, dflt) static int var = ::xm::xm_switch(name, dflt)
#else
#define XM_TUNE_VAR(var, name, dflt) [[maybe_unused]] static constexpr int var = dflt
#endif

struct bf16_t {
  uint16_t v;
};
using f16_t = _Float16;

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {  // RNE, quiet NaN: gfx950 v_cvt_pk_bf16_f32
  const __bf16 h = (__bf16)f;
  uint16_t s;
  __builtin_memcpy(&s, &h, 2);
  return s;
}

template <typename T>
__device__ __forceinline__ float to_f32(T x);
template <>
__device__ __forceinline__ float to_f32<float>(float x) { return x; }
template <>
__device__ __forceinline__ float to_f32<bf16_t>(bf16_t x) { return bf16_bits_to_f32(x.v); }
template <>
__device__ __forceinline__ float to_f32<f16_t>(f16_t x) { return (float)x; }

template <typename T>
__device__ __forceinline__ T from_f32(float x);
template <>
__device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return bf16_t{f32_to_bf16_bits(x)}; }
template <>
__device__ __forceinline__ f16_t from_f32<f16_t>(float x) { return (f16_t)x; }

// round through the tensor dtype (the reference's scalar_t cast point)
template <typename T>
__device__ __forceinline__ float r16(float x) { return to_f32<T>(from_f32<T>(x)); }

// OCP e4m3fn, saturating RNE (matches torch .to(float8_e4m3fn) on clamped input)
__device__ __forceinline__ uint8_t f32_to_e4m3_sat(float f) {
  // gfx950 has v_cvt_pk_fp8_f32 (OCP); it saturates when the clamp bit is set. We clamp explicitly
  // and use the hardware conversion (RNE).
  f = fminf(fmaxf(f, -448.0f), 448.0f);
  uint32_t r = __builtin_amdgcn_cvt_pk_fp8_f32(f, f, 0, false);
  return (uint8_t)(r & 0xff);
}

template <typename T>
struct Vec16B;  // 16-byte vector of T
template <>
struct Vec16B<float> { static constexpr int N = 4; };
template <>
struct Vec16B<bf16_t> { static constexpr int N = 8; };
template <>
struct Vec16B<f16_t> { static constexpr int N = 8; };

// 16-byte register vector of a row-wise kernel: element access with the tensor dtype's conversions
template <typename T>
struct RowVec {
  static constexpr int N = Vec16B<T>::N;
  uint4 raw;
  __device__ __forceinline__ float get(int i) const {
    if constexpr (sizeof(T) == 4) {
      return __uint_as_float((&raw.x)[i]);
    } else {
      uint32_t w = (&raw.x)[i >> 1];
      uint16_t h = (i & 1) ? (uint16_t)(w >> 16) : (uint16_t)(w & 0xffff);
      if constexpr (sizeof(T) == 2 && __is_same(T, bf16_t)) return bf16_bits_to_f32(h);
      else { f16_t x; __builtin_memcpy(&x, &h, 2); return (float)x; }
    }
  }
  __device__ __forceinline__ void set(int i, float f) {
    if constexpr (sizeof(T) == 4) {
      (&raw.x)[i] = __float_as_uint(f);
    } else {
      uint16_t h;
      if constexpr (__is_same(T, bf16_t)) h = f32_to_bf16_bits(f);
      else { f16_t x = (f16_t)f; __builtin_memcpy(&h, &x, 2); }
      uint32_t& w = (&raw.x)[i >> 1];
      w = (i & 1) ? ((w & 0x0000ffffu) | ((uint32_t)h << 16)) : ((w & 0xffff0000u) | h);
    }
  }
};

// gated activations (kernels/cuda/activation.cu:143-185); shared by rowwise.hip and the GEMM epilogues that fuse SiLU.mul
template <int MODE>
__device__ __forceinline__ float act_f(float f) {
  // SiLU on the hardware transcendental units (v_exp_f32, v_rcp_f32) with the two cheap corrections that bring it
  // back to ~2 ulp in f32: the argument of exp2 carries its rounding residual (x*log2e in two pieces), and the
  // reciprocal takes one Newton step. 12 VALU instead of the 23 of libm expf + IEEE division, which made the fused
  // silu+quant kernel VALU-bound (2.6 TB/s at 8192 x 18944; 4.8 TB/s with this form).
  if constexpr (MODE == XM_ACT_SILU) {
    const float kL2E = 1.44269504088896340736f, kL2E_lo = 1.92596299112661746e-8f, kLn2 = 0.69314718055994530942f;
    const float a = -f;
    const float hi = a * kL2E;
    const float lo = fmaf(a, kL2E, -hi) + a * kL2E_lo;
    float e = __builtin_amdgcn_exp2f(hi);
    e = fmaf(e, lo * kLn2, e);
    const float dn = 1.0f + e;
    float r = __builtin_amdgcn_rcpf(dn);
    const float rn = fmaf(r, fmaf(-dn, r, 1.0f), r);
    r = (dn < 3.0e38f) ? rn : r;  // dn = inf: keep r = 0 (the Newton step would be inf * 0); NaN keeps NaN
    return f * r;
  }
  else if constexpr (MODE == XM_ACT_GELU) return f * 0.5f * (1.0f + erff(f * 0.70710678118654752440f));
  else {
    const float kBeta = 1.41421356237309504880f * 1.12837916709551257390f * 0.5f;
    const float inner = kBeta * (f + 0.044715f * f * f * f);
    return 0.5f * f * (1.0f + tanhf(inner));
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
  return v;
}

// block-wide reductions for blockDim.x <= 1024 (multiple of 64); smem >= 17 floats
__device__ __forceinline__ float block_sum(float v, float* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  if (nw == 1) return v;
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  float r = (lane < nw) ? smem[lane] : 0.0f;
  r = wave_sum(r);
  return r;
}
__device__ __forceinline__ float block_max(float v, float* smem) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  if (nw == 1) return v;
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  float r = (lane < nw) ? smem[lane] : -INFINITY;
  r = wave_max(r);
  return r;
}

inline int hip_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? XM_OK : XM_ERR_HIP;
}

// rowwise.hip: consumes (and re-zeroes) int32 GEMM sums: dequant + residual add + RMSNorm (+ int8 quant)
// library scratch registered by xllm_mi355_set_moe_workspace (moe.hip): chunk counts of the index build; the grouped GEMM
// keeps its tile table in the tail of it
void xm_moe_scratch(void* stream, void** ws, size_t* bytes);
// workspace.hip: per-device defaults + per-stream overrides of the registered scratch buffers (kind 0 = int8 split-K GEMM
// scratch, zero at rest; kind 1 = MoE scratch), mutex-guarded
int ws_set_device(int kind, void* ws, size_t bytes);
int ws_set_stream(int kind, void* stream, void* ws, size_t bytes);
void ws_get(int kind, void* stream, void** ws, size_t* bytes);
int launch_acc_add_rms_norm(void* out, float* q_scale, int32_t* acc, const float* a_scale, const float* w_scale,
                            const void* bias, void* residual, const void* weight, float eps, int64_t M, int64_t N,
                            int dtype, int quant, hipStream_t s, int n_slabs = 0);
// rowwise.hip: consumes int32 K-slice slabs of the packed qkv projection: dequant + RoPE + KV write in one pass
int launch_slab_rope_and_cache(const int32_t* slabs, int n_slabs, const float* a_scale, const float* w_scale,
                               const void* bias, void* qkv, int64_t M, int64_t N, const int64_t* positions,
                               const void* cos_sin_cache, const int32_t* slot_ids, void* k_cache, void* v_cache,
                               int64_t n_q_heads, int64_t n_kv_heads, int64_t head_size, int64_t rot_dim, int64_t block_size,
                               int64_t n_blocks, int is_neox, int dtype, hipStream_t s);

}  // namespace xm

#define XM_DISPATCH_FLOAT(DT, T, ...)                   \
  switch (DT) {                                         \
    case XM_F32: { using T = float; __VA_ARGS__; } break;       \
    case XM_BF16: { using T = xm::bf16_t; __VA_ARGS__; } break; \
    case XM_F16: { using T = xm::f16_t; __VA_ARGS__; } break;   \
    default: return XM_ERR_UNSUPPORTED;                 \
  }
#define XM_DISPATCH_HALF(DT, T, ...)                    \
  switch (DT) {                                         \
    case XM_BF16: { using T = xm::bf16_t; __VA_ARGS__; } break; \
    case XM_F16: { using T = xm::f16_t; __VA_ARGS__; } break;   \
    default: return XM_ERR_UNSUPPORTED;                 \
  }
