// mfma_peak.hip -- what the matrix pipe sustains on this chip (calibrates the "peak" the GEMM kernels are priced
// against): back-to-back MFMAs on independent accumulators, 1 or 2 waves per SIMD, shader clock vs wall clock.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak tools/mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, int* out, long long* clk) {
  const int lane = threadIdx.x;
  i32x4 a = {lane, lane * 3, lane * 5, lane * 7}, b = {lane * 11, lane ^ 5, lane + 9, 1};
  long long c0 = clock64(), w0 = wall_clock64();
  if constexpr (MODE == 0) {  // i8 32x32x32
    i32x16 acc[4] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
    }
    int s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 0x7fffffff) out[0] = s;
  } else if constexpr (MODE == 1) {  // i8 16x16x64
    i32x4 acc[8] = {};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[j], 0, 0, 0);
    }
    int s = 0;
    for (int j = 0; j < 8; ++j) for (int r = 0; r < 4; ++r) s += acc[j][r];
    if (s == 0x7fffffff) out[0] = s;
  } else {  // bf16 32x32x16
    f32x16 acc[4] = {};
    bf16x8 av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    if (s == 12345.0f) out[0] = (int)s;
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int MODE>
void run(const char* name, double ops_per_mfma, int mfma_per_iter, int threads) {
  int* out; long long* clk;
  hipMalloc(&out, 4); hipMalloc(&clk, 16);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256, threads>>>(100, out, clk);
  hipEventRecord(e0);
  k<MODE><<<256, threads>>>(iters, out, clk);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double total = ops_per_mfma * mfma_per_iter * iters * (threads / 64.0) * 256;
  printf("%-14s waves/SIMD=%d  %.1f TOP/s  %.2f cyc/MFMA/SIMD  shader clock %.0f MHz\n", name, threads / 256,
         total / ms / 1e9, (double)h[0] / ((double)iters * mfma_per_iter * (threads / 256)),
         (double)h[0] / ((double)h[1] / 100.0));
}
int main() {
  for (int t : {256, 512}) {
    run<0>("i8 32x32x32", 2.0 * 32 * 32 * 32, 4, t);
    run<1>("i8 16x16x64", 2.0 * 16 * 16 * 64, 8, t);
    run<2>("bf16 32x32x16", 2.0 * 32 * 32 * 16, 4, t);
  }
  return 0;
}
