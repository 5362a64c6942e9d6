// probe_tr16.hip -- dumps the lane/element mapping of ds_read_b64_tr_b16 on gfx950 (used once to pin
// the V^T fragment addressing of the attention kernels).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short i16x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2048];
  int l = threadIdx.x;
  for (int i = l; i < 2048; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  // mode 0: lane-linear addresses (lane*8 bytes). mode 1: the attention kernels' addressing with a
  // 32-element row stride: row = 4*(l>>4) + ((l&15)>>2), col chunk = (l&3)*4
  int idx = mode == 0 ? l * 4 : (4 * (l >> 4) + ((l & 15) >> 2)) * 32 + (l & 3) * 4;
  i16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(lds + idx));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)t[j];
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  }
  // expected (model): mode 0 lane l elem j = (l&15) + j*16 + (l>>4)*64 ; mode 1: row 4g+j, col p16 => (4*(l>>4)+j)*32 + (l&15)
  return 0;
}
