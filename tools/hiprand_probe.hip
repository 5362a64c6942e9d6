// hiprand_probe.hip -- prints hiprand_uniform() of hiprand_init(seed, subsequence = i, offset) for i = 0..511 and two
// (seed, offset) pairs as C99 hex floats; tests/test_gpu_parity.py compares xllm_mi355_philox_uniform against it.
// build: hipcc --offload-arch=gfx950 -O2 -o probe tools/hiprand_probe.hip
#include <hip/hip_runtime.h>
#include <hiprand/hiprand_kernel.h>
#include <stdio.h>
__global__ void k(float* o, int n, unsigned long long seed, unsigned long long off) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  hiprandStatePhilox4_32_10_t st;
  hiprand_init(seed, i, off, &st);
  o[i] = hiprand_uniform(&st);
}
int main() {
  const int n = 512;
  float* d;
  if (hipMalloc(&d, n * 4) != hipSuccess) return 1;
  float h[n];
  const unsigned long long seeds[2] = {1234567ull, 99ull}, offs[2] = {5ull, 4000000003ull};
  for (int c = 0; c < 2; ++c) {
    k<<<2, 256>>>(d, n, seeds[c], offs[c]);
    if (hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    for (int i = 0; i < n; ++i) printf("%a\n", h[i]);
  }
  return 0;
}
