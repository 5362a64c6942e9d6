// hbm_read_bench.hip -- the read-only HBM ceiling the paged-decode kernel is priced against: a 2.15-GB buffer (the
// KV bytes of one cfg3 layer) streamed once by 16-B loads, for several (waves per CU, loads in flight per lane) points.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 [-DHBM_NT] -o /tmp/hbm tools/hbm_read_bench.hip && /tmp/hbm [small]
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// -DHBM_NT: non-temporal loads (global_load_dwordx4 ... nt) -- round 4: what the decode-attention kernel now uses for its KV stream
#ifdef HBM_NT
#define LD(X) __builtin_nontemporal_load(&(X))
#else
#define LD(X) (X)
#endif

template <int UNROLL>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ p, size_t n_vec, unsigned* sink) {
  const size_t stride = (size_t)gridDim.x * 256 * UNROLL;
  unsigned acc = 0;
  for (size_t base = (size_t)blockIdx.x * 256 * UNROLL + threadIdx.x; base + 256 * (UNROLL - 1) < n_vec; base += stride) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) v[i] = LD(p[base + (size_t)i * 256]);
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
    for (int i = 0; i < UNROLL; ++i) v[i] = LD(q[base + (size_t)i * 256]);
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

// the paged-decode access pattern: the buffer is [pages][128 tokens][NH heads][256 B]; a block = (sequence, head) streams
// its sequence's 32 + 32 pages and reads only ITS head's 256-byte slice of every token row (stride NH * 256 B);
// HEAD_FAST: the NH blocks of a sequence are adjacent in the grid (as in the kernel). Compare with kp<> (one block
// reads the whole 1 KiB rows = what a workgroup covering all kv heads of a sequence would do).
template <int UNROLL, int NH, bool HEAD_FAST>
__global__ __launch_bounds__(256) void ks(const u32x4* __restrict__ p, size_t n_vec, unsigned* sink) {
  const int nseq = gridDim.x / NH;
  const int seq = HEAD_FAST ? blockIdx.x / NH : blockIdx.x % nseq, head = HEAD_FAST ? blockIdx.x % NH : blockIdx.x / nseq;
  const size_t per_seq = n_vec / nseq;               // 16-byte vectors of one sequence (all heads)
  const u32x4* q = p + per_seq * seq;
  const size_t rows = per_seq / (NH * 16);             // token rows of NH * 256 B
  unsigned acc = 0;
  // thread t: row (t / 16) of a group of 16 rows, 16-byte chunk (t % 16) of the head's 256-byte slice
  for (size_t r0 = 0; r0 + 16 * UNROLL <= rows; r0 += 16 * UNROLL) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) v[i] = LD(q[(r0 + i * 16 + threadIdx.x / 16) * (NH * 16) + head * 16 + threadIdx.x % 16]);
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int UNROLL, int NH, bool HEAD_FAST>
void run_strided(const u32x4* buf, size_t bytes, int nseq, unsigned* sink) {
  const int grid = nseq * NH;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  ks<UNROLL, NH, HEAD_FAST><<<grid, 256>>>(buf, bytes / 16, sink);
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) ks<UNROLL, NH, HEAD_FAST><<<grid, 256>>>(buf, bytes / 16, sink);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  printf("head slices (256 B of every %4d-B row): %4d sequences x %d heads, %s, loads_in_flight/lane=%2d  %7.1f us  %5.2f TB/s\n",
         NH * 256, nseq, NH, HEAD_FAST ? "heads adjacent in the grid" : "heads far apart in the grid", UNROLL, ms * 1e3, bytes / ms / 1e9);
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

// short launches (the 268 MB of KV a cfg2 / one-DP-replica-of-8 decode attention launch reads): 8 launches back to back, each
// over its own eighth of the buffer (nothing comes from the 256 MB Infinity Cache), average time per launch
template <int UNROLL, bool PRIVATE>
void run_small(const u32x4* buf, size_t bytes_total, int blocks_per_cu, unsigned* sink) {
  const int grid = 256 * blocks_per_cu;
  const size_t each = bytes_total / 8 / 4096 * 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 8; ++r) k<UNROLL><<<grid, 256>>>(buf + (size_t)r * each / 16, each / 16, sink);
  hipEventRecord(e0);
  for (int r = 0; r < 8; ++r) {
    if (PRIVATE) kp<UNROLL><<<grid, 256>>>(buf + (size_t)r * each / 16, each / 16, sink);
    else k<UNROLL><<<grid, 256>>>(buf + (size_t)r * each / 16, each / 16, sink);
  }
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 8;
  printf("short launch, %zu MB each, %s: blocks/CU=%2d loads_in_flight/lane=%2d  %7.1f us  %5.2f TB/s\n", each >> 20,
         PRIVATE ? "private regions" : "interleaved walk", blocks_per_cu, UNROLL, ms * 1e3, each / ms / 1e9);
}

int main(int argc, char** argv) {
  const size_t bytes = 2151153664ull / 4096 * 4096;
  u32x4* buf; unsigned* sink;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 4); hipMemset(buf, 1, bytes);
  if (argc > 1) {  // "small": the short-launch ceiling only
    for (int b : {1, 2, 4}) {
      run_small<8, false>(buf, bytes, b, sink); run_small<16, false>(buf, bytes, b, sink);
      run_small<8, true>(buf, bytes, b, sink); run_small<16, true>(buf, bytes, b, sink);
    }
    return 0;
  }
  for (int b : {1, 2, 4, 8}) { run<4>(buf, bytes, b, sink); run<8>(buf, bytes, b, sink); run<16>(buf, bytes, b, sink); }
  for (int b : {1, 2, 4}) { run_private<8>(buf, bytes, b, sink); run_private<16>(buf, bytes, b, sink); }
  run_strided<8, 4, true>(buf, bytes, 256, sink);
  run_strided<16, 4, true>(buf, bytes, 256, sink);
  run_strided<8, 4, false>(buf, bytes, 256, sink);
  run_strided<16, 4, false>(buf, bytes, 256, sink);
  run_strided<16, 4, true>(buf, bytes, 512, sink);
  run_strided<16, 1, true>(buf, bytes, 1024, sink);
  return 0;
}
