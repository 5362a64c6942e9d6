// hbm_rw_bench.hip -- what a MIXED read / write stream gets from the MI355X HBM (the row-wise passes of a prefill chunk read two bytes
// for every byte they write; tools/hbm_read_bench.hip has the read-only ceilings). Each thread loads R 16-byte vectors and stores W of
// them (R : W = 2 : 1, 1 : 1, 1 : 0, 0 : 1), all loads of an iteration in flight before the first store; non-temporal accesses;
// buffers far larger than the L2 + Infinity Cache and rotated, so nothing is cache-resident.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbmrw tools/hbm_rw_bench.hip && /tmp/hbmrw
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int R, int W, bool NT>
__global__ __launch_bounds__(256) void rw_kernel(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n_iter_vecs) {
  // iteration i of the grid covers vectors [i * stride, (i + 1) * stride) of every one of the R input / W output planes
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t base = (size_t)blockIdx.x * 256 + threadIdx.x; base < n_iter_vecs; base += stride) {
    u32x4 v[R > 0 ? R : 1];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = NT ? __builtin_nontemporal_load(&in[r * n_iter_vecs + base]) : in[r * n_iter_vecs + base];
    u32x4 acc = {1u, 2u, 3u, 4u};
#pragma unroll
    for (int r = 0; r < R; ++r) acc ^= v[r];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const u32x4 o = acc + (unsigned)w;
      if (NT) __builtin_nontemporal_store(o, &out[w * n_iter_vecs + base]);
      else out[w * n_iter_vecs + base] = o;
    }
    if (W == 0 && acc.x == 0x12345678u) out[0] = acc;   // keep the loads alive
  }
}

template <int R, int W, bool NT>
static void run(const char* name, u32x4* in, u32x4* out, size_t plane_vecs, int blocks_per_cu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * blocks_per_cu;
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((rw_kernel<R, W, NT>), dim3(grid), dim3(256), 0, 0, in, out, plane_vecs);
  hipDeviceSynchronize();
  const int n = 5;
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL((rw_kernel<R, W, NT>), dim3(grid), dim3(256), 0, 0, in, out, plane_vecs);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= n;
  const double bytes = (double)plane_vecs * 16 * (R + W);
  printf("%-22s %s  blocks/CU=%d  %8.1f us  %6.2f TB/s  (%.0f MB read, %.0f MB written)\n", name, NT ? "nt " : "def", blocks_per_cu,
         ms * 1e3, bytes / ms / 1e9, plane_vecs * 16.0 * R / 1e6, plane_vecs * 16.0 * W / 1e6);
}

int main() {
  const size_t plane_bytes = 512ull << 20;           // per plane; 2 read planes + 1 write plane = 1.5 GB per launch
  const size_t plane_vecs = plane_bytes / 16;
  u32x4 *in, *out;
  hipMalloc(&in, plane_bytes * 2);
  hipMalloc(&out, plane_bytes * 2);
  hipMemset(in, 1, plane_bytes * 2);
  hipMemset(out, 0, plane_bytes * 2);
  for (int bpc : {2, 4, 8}) {
    run<2, 1, true>("read 2 : write 1", in, out, plane_vecs, bpc);
    run<1, 1, true>("read 1 : write 1 (copy)", in, out, plane_vecs, bpc);
    run<2, 0, true>("read only", in, out, plane_vecs, bpc);
    run<0, 2, true>("write only", in, out, plane_vecs, bpc);
    run<2, 1, false>("read 2 : write 1", in, out, plane_vecs, bpc);
    run<1, 1, false>("read 1 : write 1 (copy)", in, out, plane_vecs, bpc);
  }
  return 0;
}
