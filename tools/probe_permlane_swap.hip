// probe of v_permlane16_swap / v_permlane32_swap (gfx950) as exposed by __builtin_amdgcn_permlane{16,32}_swap:
// prints both results for a = lane id, b = 100 + lane id
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/pl tools/probe_permlane_swap.hip && /tmp/pl
#include <hip/hip_runtime.h>
#include <stdio.h>
// NOTE: __builtin_bit_cast(float, vec[i]) on a vector ELEMENT reads element 0 for every i (hipcc, ROCm 7.2): go through a
// by-value function argument.
__device__ __forceinline__ float as_f32(unsigned v) { return __builtin_bit_cast(float, v); }
__device__ __forceinline__ float group4_max(float x) {
  unsigned u = __builtin_bit_cast(unsigned, x), c;
  asm volatile("v_mov_b32 %0, %1" : "=v"(c) : "v"(u));
  const auto r16 = __builtin_amdgcn_permlane16_swap(u, c, false, false);
  x = fmaxf(as_f32(r16[0]), as_f32(r16[1]));
  u = __builtin_bit_cast(unsigned, x);
  asm volatile("v_mov_b32 %0, %1" : "=v"(c) : "v"(u));
  const auto r32 = __builtin_amdgcn_permlane32_swap(u, c, false, false);
  return fmaxf(as_f32(r32[0]), as_f32(r32[1]));
}
__global__ void kmax(float* p) { p[threadIdx.x] = group4_max(p[threadIdx.x]); }
__global__ void k(unsigned* o) {
  const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  auto r32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r16[0]; o[64 + threadIdx.x] = r16[1]; o[128 + threadIdx.x] = r32[0]; o[192 + threadIdx.x] = r32[1];
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, 1024);
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane16_swap r[0]", "permlane16_swap r[1]", "permlane32_swap r[0]", "permlane32_swap r[1]"};
  for (int j = 0; j < 4; ++j) {
    printf("%s:", names[j]);
    for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[j * 64 + i]);
    printf("\n");
  }
  float hf[64], of[64], *df;
  for (int i = 0; i < 64; ++i) hf[i] = (float)((i * 37) % 64);
  hipMalloc(&df, 256); hipMemcpy(df, hf, 256, hipMemcpyHostToDevice);
  kmax<<<1, 64>>>(df);
  hipMemcpy(of, df, 256, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int i = 0; i < 64; ++i) ok &= of[i] == fmaxf(fmaxf(hf[i], hf[i ^ 16]), fmaxf(hf[i ^ 32], hf[i ^ 48]));
  printf("4-group maximum through the swaps: %s\n", ok ? "OK" : "WRONG");
  if (!ok) for (int i = 0; i < 64; i += 5) printf("  lane %d: in %g out %g expect %g\n", i, hf[i], of[i], fmaxf(fmaxf(hf[i], hf[i ^ 16]), fmaxf(hf[i ^ 32], hf[i ^ 48])));
  return 0;
}
// (second probe) the 4-group maximum the attention kernels need: lanes l, l^16, l^32, l^48 -> same value. Passing the SAME
// SSA value for both operands miscompiles (hipcc, ROCm 7.2: the swap of a register with itself), so the second operand
// is an opaque copy.
