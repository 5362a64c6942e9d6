// launch_ramp_bench.hip -- what a kernel boundary costs on MI355X as a function of the workgroup's LDS allocation and size.
// Question behind it: the small decode projections (qkv, o at M = 256) spend ~6 us in "launch + dispatch ramp of 256
// workgroups x 160 KiB of LDS" around 9.5 us of work. Is that cost tied to the LDS allocation / the 512-thread workgroups?
// Each kernel does one global load + store per thread (so the launch is not optimised away) and exits; N launches back to back on
// one stream, timed with events, and the same inside a HIP graph.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ramp tools/launch_ramp_bench.hip && /tmp/ramp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void touch_kernel(float* p, int spin) {
  extern __shared__ char lds[];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = p[i];
  if (spin) {   // a body of about `spin` x 100 ns so that consecutive launches cannot overlap their ramps entirely
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < spin * 10) __builtin_amdgcn_s_sleep(8);
  }
  if (threadIdx.x == 0) lds[0] = (char)v;
  __syncthreads();
  p[i] = v + (float)lds[0] * 0.0f + 1.0f;
}

static double time_launches(int grid, int block, size_t lds, int spin, float* buf, int n, bool graph) {
  hipStream_t s;
  hipStreamCreate(&s);
  hipFuncSetAttribute((const void*)touch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(block), lds, s, buf, spin);
  hipStreamSynchronize(s);
  float ms = 0;
  if (graph) {
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(block), lds, s, buf, spin);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
  } else {
    hipEventRecord(e0, s);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(touch_kernel, dim3(grid), dim3(block), lds, s, buf, spin);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    hipEventElapsedTime(&ms, e0, e1);
  }
  hipStreamDestroy(s);
  return ms * 1e3 / n;
}

int main() {
  float* buf;
  hipMalloc(&buf, 1024 * 512 * sizeof(float));
  hipMemset(buf, 0, 1024 * 512 * sizeof(float));
  const int n = 200;
  printf("# us per launch, %d launches back to back; body = load + store (+ a %s-us wait)\n", n, "0 / 5");
  printf("%-8s %-8s %-10s %-6s %12s %12s\n", "grid", "block", "lds_KiB", "spin", "stream_us", "graph_us");
  const int grids[] = {256, 512, 1024};
  const int blocks[] = {256, 512};
  const int ldss[] = {0, 32, 64, 96, 128, 160};
  for (int spin : {0, 50})
    for (int g : grids)
      for (int b : blocks)
        for (int l : ldss) {
          if (g * b > 1024 * 512) continue;
          if (g > 256 && l > 64) continue;   // more than one workgroup per CU only fits with small allocations
          const double ts = time_launches(g, b, (size_t)l * 1024, spin, buf, n, false);
          const double tg = time_launches(g, b, (size_t)l * 1024, spin, buf, n, true);
          printf("%-8d %-8d %-10d %-6d %12.2f %12.2f\n", g, b, l, spin, ts, tg);
        }
  hipFree(buf);
  return 0;
}
