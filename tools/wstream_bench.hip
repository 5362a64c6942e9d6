// wstream_bench.hip -- how fast can the CUs stream DISTINCT weight rows from HBM (cold: the buffer is far larger than the
// 256 MB Infinity Cache and every launch starts at a new offset) into LDS with LDS-DMA, as a function of
//   * the stage shape: R rows x CB contiguous bytes per row (R * CB = 32 KiB per stage; row pitch = K bytes),
//   * the number of stages in flight per workgroup (LDS ring depth, up to 160 KiB),
//   * the number of workgroups (148 = tiles of gate_up at M = 256 on the 256x256 kernel; 256; 512 = two per CU).
// This is the weight leg of a decode GEMM (M <= 256): which design can reach the ~6.2 TB/s HBM read ceiling?
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/wstream tools/wstream_bench.hip && /tmp/wstream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <int R, int CB, int DEPTH, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const uint8_t* base, long long wg_stride, int row_pitch, int steps, int* sink) {
  static_assert(R * CB == 32768, "one stage = 32 KiB");
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
  constexpr int NI = 32768 / 16 / THREADS;  // DMA instructions per thread and stage
  constexpr int CPR = CB / 16;              // 16-byte chunks per row
  const int tid = threadIdx.x;
  const uint8_t* src = base + (long long)blockIdx.x * wg_stride;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, 0x7fffffff, 0x00020000);
  int voff[8];  // [NI] used (a template-dependent array size in a kernel template makes hipcc drop the host stub)
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = i * THREADS + tid;  // lane-linear 16-byte chunk of the stage
    voff[i] = (c / CPR) * row_pitch + (c % CPR) * 16;
  }
  typedef __attribute__((address_space(3))) uint8_t* lp;
  const int wave = tid >> 6;
  // prologue: DEPTH - 1 stages in flight, then one new stage per retired one
  for (int t = 0; t < steps + DEPTH - 1; ++t) {
    if (t < steps) {
      const lp dst = (lp)lds + (t % DEPTH) * 32768 + wave * 1024;
#pragma unroll
      for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst + i * (THREADS * 16), 16, voff[i], t * CB, 0, 0);
    }
    if (t >= DEPTH - 1) {  // retire the oldest stage: at most (DEPTH - 1) * NI younger DMAs may stay outstanding
      if constexpr ((DEPTH - 1) * NI == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if constexpr ((DEPTH - 1) * NI >= 63) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * NI) : "memory");
      __builtin_amdgcn_s_barrier();  // a real kernel publishes the stage here
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lds[tid * 64] == 0x77 && sink) sink[0] = 1;
}

template <int R, int CB, int DEPTH, int THREADS>
void run(const uint8_t* buf, size_t buf_bytes, int nwg, int K, int rows_per_wg) {
  const int steps_per_slab = K / CB;            // K walk of one R-row slab
  const int slabs = rows_per_wg / R;            // a workgroup owns rows_per_wg rows = `slabs` slabs walked one after another
  (void)slabs;
  // simplification: the workgroup walks K for its first slab, then the next slab, ... = steps_per_slab * slabs stages
  // (address pattern per stage is what matters); here: one long K walk over a row pitch of K * slabs keeps the kernel simple
  const int steps = steps_per_slab * (rows_per_wg / R);
  const long long wg_stride = (long long)rows_per_wg * K;
  const double bytes = (double)nwg * steps * 32768;
  hipFuncSetAttribute((const void*)k<R, CB, DEPTH, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 32768);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 6;
  size_t off = 0;
  const size_t span = (size_t)nwg * wg_stride + (1 << 20);
  float best = 1e30f, sum = 0;
  for (int r = 0; r < reps; ++r) {
    if (off + span > buf_bytes) off = 0;
    hipEventRecord(e0);
    k<R, CB, DEPTH, THREADS><<<nwg, THREADS, DEPTH * 32768>>>(buf + off, wg_stride, K * (rows_per_wg / R), steps, nullptr);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r > 0) { best = ms < best ? ms : best; sum += ms; }
    off += span;
  }
  const float ms = sum / (reps - 1);
  printf("stage %3d rows x %4d B  depth %d (%3d KiB in flight)  wgs %3d x %d thr: %7.1f us  %5.2f TB/s (best %5.2f)  %5.1f GB/s per WG\n", R, CB,
         DEPTH, (DEPTH - 1) * 32, nwg, THREADS, ms * 1e3, bytes / ms / 1e9, bytes / best / 1e9, bytes / ms / 1e6 / nwg);
}

int main() {
  const size_t buf_bytes = (size_t)3 << 30;  // 3 GiB: every launch reads a region the Infinity Cache has not seen
  uint8_t* buf;
  if (hipMalloc(&buf, buf_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 1, buf_bytes);
  const int K = 3584;
  // gate_up at M = 256: N = 37888 rows of K bytes = 136 MB per launch
  printf("== 148 workgroups x 256 rows (the 256x256 tiling of gate_up at M = 256) ==\n");
  run<256, 128, 2, 512>(buf, buf_bytes, 148, K, 256);
  run<256, 128, 3, 512>(buf, buf_bytes, 148, K, 256);
  run<256, 128, 4, 512>(buf, buf_bytes, 148, K, 256);
  run<256, 128, 5, 512>(buf, buf_bytes, 148, K, 256);
  run<64, 512, 3, 512>(buf, buf_bytes, 148, K, 256);
  run<64, 512, 5, 512>(buf, buf_bytes, 148, K, 256);
  printf("== 256 workgroups x 148 rows (one N slab per CU) -> 37888 rows; stage rows must divide: use 4-row-multiple slabs ==\n");
  run<64, 512, 3, 512>(buf, buf_bytes, 256, K, 128);
  run<64, 512, 5, 512>(buf, buf_bytes, 256, K, 128);
  run<128, 256, 3, 512>(buf, buf_bytes, 256, K, 128);
  run<128, 256, 5, 512>(buf, buf_bytes, 256, K, 128);
  run<32, 1024, 3, 512>(buf, buf_bytes, 256, K, 128);
  run<32, 1024, 5, 512>(buf, buf_bytes, 256, K, 128);
  printf("== 512 workgroups (two per CU, 256 threads each) x 64 rows ==\n");
  run<64, 512, 2, 256>(buf, buf_bytes, 512, K, 64);
  run<64, 512, 3, 256>(buf, buf_bytes, 512, K, 64);
  run<32, 1024, 2, 256>(buf, buf_bytes, 512, K, 64);
  run<32, 1024, 3, 256>(buf, buf_bytes, 512, K, 64);
  printf("== 1024 workgroups (four per CU) x 32 rows, depth 2 ==\n");
  run<32, 1024, 2, 256>(buf, buf_bytes, 1024, K, 32);
  hipFree(buf);
  return 0;
}
