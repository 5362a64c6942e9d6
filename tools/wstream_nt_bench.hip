// wstream_nt_bench.hip -- the weight leg of a decode GEMM alone, as the packed layout streams it (round 5): every workgroup (512
// threads, one per CU) pulls ONE CONTIGUOUS region of cold HBM into an LDS ring with LDS-DMA (buffer_load_dwordx4 ... lds), 32-KiB
// stages, DEPTH - 1 stages in flight, no matrix work, no activations. tools/wstream_bench.hip (round 1) measured row-pitched
// patterns with the default cache policy: 5.0-5.25 TB/s whatever the depth. Questions here: does the non-temporal policy the product
// kernel uses since round 4/5 (aux = 2) lift that ceiling, and what do launches of gate_up's (136 MB) and lm_head's (1.09 GB) size get?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wsnt tools/wstream_nt_bench.hip && /tmp/wsnt
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int DEPTH, int AUX>
__global__ __launch_bounds__(512, 1) void k(const uint8_t* base, long long wg_bytes, int steps, int* sink) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint8_t* src = base + (long long)blockIdx.x * wg_bytes;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, 0x7fffffff, 0x00020000);
  typedef __attribute__((address_space(3))) uint8_t* lp;
  const int voff = tid * 16;   // instruction i of a stage covers bytes [i * 8 KiB, (i + 1) * 8 KiB) of it: 4 per thread and stage
  for (int t = 0; t < steps + DEPTH - 1; ++t) {
    if (t < steps) {
      const lp dst = (lp)lds + (t % DEPTH) * 32768 + wave * 1024;
#pragma unroll
      for (int i = 0; i < 4; ++i) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dst + i * 8192, 16, voff + i * 8192, t * 32768, 0, AUX);
    }
    if (t >= DEPTH - 1) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DEPTH - 1) * 4) : "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (lds[tid * 64] == 0x77 && sink) sink[0] = 1;
}

template <int DEPTH, int AUX>
static void run(const uint8_t* buf, size_t buf_bytes, int nwg, size_t launch_bytes, size_t* off) {
  const int steps = (int)(launch_bytes / nwg / 32768);
  const long long wg_bytes = (long long)steps * 32768;
  const size_t span = (size_t)nwg * wg_bytes;
  hipFuncSetAttribute((const void*)k<DEPTH, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, DEPTH * 32768);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float sum = 0, best = 1e30f;
  const int reps = 6;
  for (int r = 0; r < reps; ++r) {   // every launch reads a region the Infinity Cache has not seen
    if (*off + span > buf_bytes) *off = 0;
    hipEventRecord(e0);
    k<DEPTH, AUX><<<nwg, 512, DEPTH * 32768>>>(buf + *off, wg_bytes, steps, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r > 0) { sum += ms; best = ms < best ? ms : best; }
    *off += span;
  }
  const float ms = sum / (reps - 1);
  printf("%-7s depth %d (%3d KiB in flight per CU)  wgs %3d  %7.1f MB: %7.1f us  %5.2f TB/s (best %5.2f)  %5.1f GB/s per CU\n",
         AUX ? "nt" : "default", DEPTH, (DEPTH - 1) * 32, nwg, span / 1e6, ms * 1e3, span / ms / 1e9, span / best / 1e9, span / ms / 1e6 / nwg);
}

template <int AUX>
static void sweep(const uint8_t* buf, size_t buf_bytes, int nwg, size_t launch_bytes, size_t* off) {
  run<2, AUX>(buf, buf_bytes, nwg, launch_bytes, off);
  run<3, AUX>(buf, buf_bytes, nwg, launch_bytes, off);
  run<4, AUX>(buf, buf_bytes, nwg, launch_bytes, off);
  run<5, AUX>(buf, buf_bytes, nwg, launch_bytes, off);
}

int main() {
  const size_t buf_bytes = (size_t)6 << 30;
  uint8_t* buf;
  if (hipMalloc(&buf, buf_bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 1, buf_bytes);
  size_t off = 0;
  for (int nwg : {237, 256}) {
    printf("== gate_up-sized launch (136 MB), %d workgroups ==\n", nwg);
    sweep<0>(buf, buf_bytes, nwg, (size_t)136 << 20, &off);
    sweep<2>(buf, buf_bytes, nwg, (size_t)136 << 20, &off);
  }
  printf("== lm_head-sized launch (1.09 GB), 256 workgroups ==\n");
  sweep<0>(buf, buf_bytes, 256, (size_t)1090 << 20, &off);
  sweep<2>(buf, buf_bytes, 256, (size_t)1090 << 20, &off);
  hipFree(buf);
  return 0;
}
