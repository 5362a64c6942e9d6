// dma_bench.hip -- how fast can a CU pull operand tiles into LDS with LDS-DMA (buffer_load_dwordx4 ... lds)?
// Calibrates the L2 -> LDS leg of the GEMM kernels: per-CU and chip-wide rates when (a) every workgroup streams its
// OWN rows (weights pattern) or (b) ALL workgroups stream the SAME 256 x K panel (activation pattern at M = 256).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_bench tools/dma_bench.hip && /tmp/dma_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

// each workgroup (512 threads) copies `ktiles` tiles of 256 rows x 128 B (32 KiB) into a 4-deep LDS ring
template <int INFLIGHT>
__global__ __launch_bounds__(512, 1) void k(const uint8_t* base, long long wg_stride, int row_bytes, int ktiles, int* sink) {
  __shared__ __attribute__((aligned(1024))) uint8_t lds[4 * 32768];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint8_t* src = base + (long long)blockIdx.x * wg_stride;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, 0x7fffffff, 0x00020000);
  int voff[4];
  for (int i = 0; i < 4; ++i) voff[i] = ((i * 8 + wave) * 8 + (lane >> 3)) * row_bytes + (lane & 7) * 16;
  typedef __attribute__((address_space(3))) uint8_t* lp;
  for (int t = 0; t < ktiles; ++t) {
    const lp dst = (lp)lds + (t & 3) * 32768 + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst + i * 8192, 16, voff[i], t * 128, 0, 0);
    if (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (INFLIGHT == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lds[tid * 64] == 0x77 && sink) sink[0] = 1;
}

template <int INFLIGHT>
void run(const char* name, const uint8_t* buf, int nwg, long long wg_stride, int row_bytes, int ktiles) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<INFLIGHT><<<nwg, 512>>>(buf, wg_stride, row_bytes, ktiles, nullptr);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<INFLIGHT><<<nwg, 512>>>(buf, wg_stride, row_bytes, ktiles, nullptr);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double bytes = (double)nwg * ktiles * 32768;
  printf("%-34s wgs=%3d tiles_in_flight=%d  %7.1f us  %6.2f TB/s  %5.1f GB/s per CU\n", name, nwg, INFLIGHT, ms * 1e3,
         bytes / ms / 1e9, bytes / ms / 1e6 / nwg);
}

int main() {
  const int row_bytes = 3584, ktiles = 28;
  uint8_t* buf; const size_t sz = (size_t)256 * 256 * row_bytes + (1 << 20);
  hipMalloc(&buf, sz); hipMemset(buf, 1, sz);
  for (int nwg : {32, 64, 128, 256}) {
    run<1>("own rows (weights, L2/MALL warm)", buf, nwg, (long long)256 * row_bytes, row_bytes, ktiles);
    run<2>("own rows (weights, L2/MALL warm)", buf, nwg, (long long)256 * row_bytes, row_bytes, ktiles);
    run<3>("own rows (weights, L2/MALL warm)", buf, nwg, (long long)256 * row_bytes, row_bytes, ktiles);
    run<1>("same panel (activations)", buf, nwg, 0, row_bytes, ktiles);
    run<2>("same panel (activations)", buf, nwg, 0, row_bytes, ktiles);
    run<3>("same panel (activations)", buf, nwg, 0, row_bytes, ktiles);
  }
  return 0;
}
