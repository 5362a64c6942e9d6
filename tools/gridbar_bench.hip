// gridbar_bench.hip -- what a grid-wide barrier inside one persistent launch costs on MI355X against a kernel boundary
// in a HIP graph; the sizing measurement for a persistent "layer tail" launch.
// Every step: each workgroup writes a chunk (value = f(step, wg)), the grid synchronises, each workgroup reads the chunk
// of a workgroup on ANOTHER XCD and checks it (the visibility the fused phases would rely on).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/gridbar tools/gridbar_bench.hip && /tmp/gridbar
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: __syncthreads, thread 0 release-add / acquire-poll (agent scope), __syncthreads
// MODE 1: every thread __threadfence() before and after (what cooperative groups does)
// MODE 2: as 0, polling without s_sleep
template <int MODE>
__device__ inline void grid_barrier(unsigned* ctr, unsigned target) {
  if (MODE == 1) __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (MODE != 2) __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
  if (MODE == 1) __threadfence();
}

__device__ inline unsigned f(unsigned step, unsigned wg, unsigned i) { return step * 2654435761u + wg * 40503u + i; }

template <int MODE>
__global__ __launch_bounds__(256) void persistent(unsigned* buf, int chunk_words, int steps, unsigned* ctr, unsigned* errs) {
  const unsigned G = gridDim.x, wg = blockIdx.x;
  const unsigned base = ctr[1];                       // barrier generation carried across launches
  unsigned bad = 0;
  for (int s = 0; s < steps; ++s) {
    unsigned* b = buf + (size_t)(s & 1) * G * chunk_words;
    for (int i = threadIdx.x; i < chunk_words; i += 256) b[(size_t)wg * chunk_words + i] = f(s, wg, i);
    grid_barrier<MODE>(ctr, base + (unsigned)(s + 1) * G);
    const unsigned src = (wg + 37) % G;
    for (int i = threadIdx.x; i < chunk_words; i += 256) bad += b[(size_t)src * chunk_words + i] != f(s, src, i);
  }
  if (bad) atomicAdd(errs, bad);
  // last one out publishes the new generation for the next launch
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned done = atomicAdd(ctr + 2, 1u);
    if (done == G - 1) { ctr[2] = 0; ctr[1] = base + (unsigned)steps * G; }
  }
}

__global__ __launch_bounds__(256) void step_write(unsigned* buf, int chunk_words, int s) {
  unsigned* b = buf + (size_t)(s & 1) * gridDim.x * chunk_words;
  for (int i = threadIdx.x; i < chunk_words; i += 256) b[(size_t)blockIdx.x * chunk_words + i] = f(s, blockIdx.x, i);
}
__global__ __launch_bounds__(256) void step_check(const unsigned* buf, int chunk_words, int s, unsigned* errs) {
  const unsigned* b = buf + (size_t)(s & 1) * gridDim.x * chunk_words;
  const unsigned src = (blockIdx.x + 37) % gridDim.x;
  unsigned bad = 0;
  for (int i = threadIdx.x; i < chunk_words; i += 256) bad += b[(size_t)src * chunk_words + i] != f(s, src, i);
  if (bad) atomicAdd(errs, bad);
}
// one kernel per step = check(s-1) + write(s): the boundary count of the persistent loop
__global__ __launch_bounds__(256) void step_both(unsigned* buf, int chunk_words, int s, unsigned* errs) {
  if (s > 0) {
    const unsigned* b = buf + (size_t)((s - 1) & 1) * gridDim.x * chunk_words;
    const unsigned src = (blockIdx.x + 37) % gridDim.x;
    unsigned bad = 0;
    for (int i = threadIdx.x; i < chunk_words; i += 256) bad += b[(size_t)src * chunk_words + i] != f(s - 1, src, i);
    if (bad) atomicAdd(errs, bad);
  }
  unsigned* b = buf + (size_t)(s & 1) * gridDim.x * chunk_words;
  for (int i = threadIdx.x; i < chunk_words; i += 256) b[(size_t)blockIdx.x * chunk_words + i] = f(s, blockIdx.x, i);
}

template <int MODE>
static void run_persistent(const char* name, int G, unsigned* buf, int cw, int steps, unsigned* ctr, unsigned* errs, hipStream_t st) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) persistent<MODE><<<G, 256, 0, st>>>(buf, cw, steps, ctr, errs);
  CHECK(hipStreamSynchronize(st));
  const int reps = 10;
  CHECK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) persistent<MODE><<<G, 256, 0, st>>>(buf, cw, steps, ctr, errs);
  CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned h; CHECK(hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost));
  printf("%-34s G=%3d chunk=%6d B  %7.2f us/step  (launch of %d steps %.1f us)  errs=%u\n", name, G, cw * 4, ms * 1e3 / reps / steps, steps,
         ms * 1e3 / reps, h);
}

int main() {
  hipStream_t st; CHECK(hipStreamCreate(&st));
  const int steps = 200;
  unsigned *buf, *ctr, *errs;
  CHECK(hipMalloc(&buf, (size_t)2 * 512 * 65536 * 4)); CHECK(hipMalloc(&ctr, 64)); CHECK(hipMalloc(&errs, 4));
  CHECK(hipMemset(ctr, 0, 64)); CHECK(hipMemset(errs, 0, 4));
  for (int G : {256, 128}) {
    for (int cw : {256, 4096, 16384}) {     // 1 KiB, 16 KiB, 64 KiB per workgroup and step (0.25 / 4 / 16 MiB per step at G = 256)
      run_persistent<0>("persistent, thread-0 rel/acq", G, buf, cw, steps, ctr, errs, st);
      run_persistent<1>("persistent, all-thread fences", G, buf, cw, steps, ctr, errs, st);
      run_persistent<2>("persistent, thread-0, no sleep", G, buf, cw, steps, ctr, errs, st);
      // graph of `steps` kernels
      hipGraph_t g; hipGraphExec_t ge;
      CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int s = 0; s < steps; ++s) step_both<<<G, 256, 0, st>>>(buf, cw, s, errs);
      CHECK(hipStreamEndCapture(st, &g)); CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
      for (int w = 0; w < 3; ++w) CHECK(hipGraphLaunch(ge, st));
      CHECK(hipStreamSynchronize(st));
      CHECK(hipEventRecord(e0, st));
      for (int r = 0; r < 10; ++r) CHECK(hipGraphLaunch(ge, st));
      CHECK(hipEventRecord(e1, st)); CHECK(hipStreamSynchronize(st));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      unsigned h; CHECK(hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost));
      printf("%-34s G=%3d chunk=%6d B  %7.2f us/step                               errs=%u\n", "graph, one kernel per step", G, cw * 4,
             ms * 1e3 / 10 / steps, h);
      CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
    }
  }
  return 0;
}
