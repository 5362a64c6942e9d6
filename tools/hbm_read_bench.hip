// hbm_read_bench.hip -- the read-only HBM ceiling the paged-decode kernel is priced against: a 2.15-GB buffer (the
// KV bytes of one cfg3 layer) streamed once by 16-B loads, for several (waves per CU, loads in flight per lane) points.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm tools/hbm_read_bench.hip && /tmp/hbm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ p, size_t n_vec, unsigned* sink) {
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  unsigned acc = 0;
  for (size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; base + 256 * (UNROLL - 1) < n_vec; base += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) v[i] = p[base + (size_t)i * 256];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// every block streams its own contiguous region (what a paged-attention workgroup does with its sequence), `chunk`
// 16-byte vectors at a time per block, instead of the chip-wide interleaved grid-stride walk of k<>
template <int UNROLL>
__global__ __launch_bounds__(256) void kp(const u32x4* __restrict__ p, size_t n_vec, unsigned* sink) {
  const size_t per = n_vec / gridDim.x;
  const u32x4* q = p + per * blockIdx.x;
  unsigned acc = 0;
  for (size_t base = threadIdx.x; base + 256 * (UNROLL - 1) < per; base += 256 * UNROLL) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) v[i] = q[base + (size_t)i * 256];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int UNROLL>
void run_private(const u32x4* buf, size_t bytes, int blocks_per_cu, unsigned* sink) {
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kp<UNROLL><<<grid, 256>>>(buf, bytes / 16, sink);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) kp<UNROLL><<<grid, 256>>>(buf, bytes / 16, sink);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("private regions: blocks/CU=%2d loads_in_flight/lane=%2d  %7.1f us  %5.2f TB/s\n", blocks_per_cu, UNROLL, ms * 1e3,
         bytes / ms / 1e9);
}

template <int UNROLL>
void run(const u32x4* buf, size_t bytes, int blocks_per_cu, unsigned* sink) {
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<UNROLL><<<grid, 256>>>(buf, bytes / 16, sink);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) k<UNROLL><<<grid, 256>>>(buf, bytes / 16, sink);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("blocks/CU=%2d (waves/CU=%2d) loads_in_flight/lane=%2d  %7.1f us  %5.2f TB/s\n", blocks_per_cu, blocks_per_cu * 4,
         UNROLL, ms * 1e3, bytes / ms / 1e9);
}

int main() {
  const size_t bytes = 2151153664ull / 4096 * 4096;
  u32x4* buf; unsigned* sink;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 4); hipMemset(buf, 1, bytes);
  for (int b : {1, 2, 4, 8}) { run<4>(buf, bytes, b, sink); run<8>(buf, bytes, b, sink); run<16>(buf, bytes, b, sink); }
  for (int b : {1, 2, 4}) { run_private<8>(buf, bytes, b, sink); run_private<16>(buf, bytes, b, sink); }
  return 0;
}
