// lds_pattern_bench.hip -- LDS read throughput of the MLA kernels' access patterns vs. linear baselines (no DMA, no MFMA):
// 4 waves per workgroup (one per SIMD), one workgroup per CU, each wave issues REPS x 8 reads; prints LDS cycles per
// wave-instruction as seen by the CU (4 waves share the LDS: 4.0 = 256 B/clk for b128, 2.0 for b64).
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsp tools/lds_pattern_bench.hip && /tmp/ldsp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
constexpr int ROWB = 1152, REPS = 2000;

#define RD128(D, A, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(D) : "v"(A), "n"(OFF))
#define RD64TR(D, A, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(D) : "v"(A), "n"(OFF))
#define RD64(D, A, OFF) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(D) : "v"(A), "n"(OFF))

// candidate: t = (row >> 1) & 7 -> {0,3,4,7,2,1,6,5}[t]: rows of one parity inside an 8-row half take one value from each
// chunk pair {c, c ^ 2} (b128 groups = 8 rows x lane quarters {0,2} / {1,3}) AND distinct f >> 1 (the transposed reads)
__device__ __forceinline__ int f2(int r) { return (((r >> 2) & 1) << 2) | ((((r >> 1) ^ (r >> 3)) & 1) << 1) | ((r >> 1) & 1); }

template <int PAT>
__global__ __launch_bounds__(256, 1) void k(long long* cyc, unsigned* sink) {
  __shared__ __attribute__((aligned(1024))) char lds[64 * ROWB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, p16 = lane & 15, g = lane >> 4;
  for (int i = tid; i < 64 * ROWB / 4; i += 256) ((unsigned*)lds)[i] = i;
  __syncthreads();
  const unsigned base = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char*)lds;
  unsigned a0 = base, a1 = base;
  if (PAT == 0) { a0 = base + lane * 16; a1 = a0 + 1024; }                       // b128 linear
  if (PAT == 1 || PAT == 4) {                                                      // MLA K fragments (prefill: every wave all rows)
    const int fq = (p16 & 6) | ((p16 >> 3) & 1);
    const int row = (PAT == 4 ? wave * 16 : 0) + p16;                              // PAT 4: decode (wave w rows 16 w ..)
    a0 = base + row * ROWB + ((g ^ fq) << 4);
    a1 = base + row * ROWB + (((g ^ fq) << 4) ^ 64);
  }
  if (PAT == 2) {                                                                  // MLA V^T transposed reads
    const int ft = (((p16 >> 3) & 1) << 1) | ((g & 1) << 2) | ((g >> 1) & 1);
    const int v_x = ((((p16 >> 1) & 1) ^ ft) << 4) | ((p16 & 1) << 3);
    a0 = base + (4 * g + (p16 >> 2)) * ROWB + v_x;
    a1 = base + (4 * g + (p16 >> 2)) * ROWB + (v_x ^ 32);
  }
  if (PAT == 3) { a0 = base + lane * 8; a1 = a0 + 512; }                          // b64 linear
  if (PAT == 5) { a0 = base + p16 * ROWB + g * 16; a1 = a0 + 64; }               // K fragments WITHOUT the swizzle
  if (PAT == 6) {                                                                  // K fragments, candidate swizzle f2
    const int fq = f2(p16);
    a0 = base + p16 * ROWB + ((g ^ fq) << 4);
    a1 = base + p16 * ROWB + (((g ^ fq) << 4) ^ 64);
  }
  if (PAT == 8) { a0 = base + p16 * 272 + g * 16; a1 = a0 + 64; }                 // fragment rows padded to 272 B (no swizzle)
  if (PAT == 9) { a0 = base + (lane >> 2) * ROWB + (lane & 3) * 16; a1 = a0 + 64; }  // 4 lanes = 64 contiguous bytes per row
  if (PAT == 10) { a0 = base + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4); a1 = a0 + 1024; }  // 8 lanes = one 128-B row, chunks permuted
  if (PAT == 11) { a0 = base + (lane >> 1) * 272 + (lane & 1) * 16; a1 = a0 + 64; }  // lane pairs = 32 contiguous bytes, rows padded
  // MFMA fragment mapping (lane = (row p16, chunk g)) over an 8-row-interleaved tile: [row >> 3][chunk][row & 7][16 B]
  if (PAT == 12) { a0 = base + (p16 >> 3) * 9216 + g * 128 + (p16 & 7) * 16; a1 = a0 + 512; }
  if (PAT == 7) {                                                                  // V^T transposed reads under f2
    const int row = 4 * g + (p16 >> 2);
    const int v_x = ((((p16 >> 1) & 1) ^ f2(row)) << 4) | ((p16 & 1) << 3);
    a0 = base + row * ROWB + v_x;
    a1 = base + row * ROWB + (v_x ^ 32);
  }
  unsigned acc = 0;
  __syncthreads();
  const long long t0 = clock64();
  for (int r = 0; r < REPS; ++r) {
    if (PAT == 0 || PAT == 1 || PAT == 4 || PAT == 5 || PAT == 6 || PAT >= 8) {
      u4 d0, d1, d2, d3, d4, d5, d6, d7;
      RD128(d0, a0, 0); RD128(d1, a1, 0); RD128(d2, a0, 128); RD128(d3, a1, 128);
      RD128(d4, a0, 256); RD128(d5, a1, 256); RD128(d6, a0, 384); RD128(d7, a1, 384);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
      acc ^= d0.x ^ d1.y ^ d2.z ^ d3.w ^ d4.x ^ d5.y ^ d6.z ^ d7.w;
    } else {
      u2 d0, d1, d2, d3, d4, d5, d6, d7;
      if (PAT == 2 || PAT == 7) {
        RD64TR(d0, a0, 0); RD64TR(d1, a0, 16 * ROWB); RD64TR(d2, a1, 0); RD64TR(d3, a1, 16 * ROWB);
        RD64TR(d4, a0, 128); RD64TR(d5, a0, 16 * ROWB + 128); RD64TR(d6, a1, 128); RD64TR(d7, a1, 16 * ROWB + 128);
      } else {
        RD64(d0, a0, 0); RD64(d1, a1, 0); RD64(d2, a0, 1024); RD64(d3, a1, 1024);
        RD64(d4, a0, 2048); RD64(d5, a1, 2048); RD64(d6, a0, 3072); RD64(d7, a1, 3072);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
      acc ^= d0.x ^ d1.y ^ d2.x ^ d3.y ^ d4.x ^ d5.y ^ d6.x ^ d7.y;
    }
  }
  const long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
  if (acc == 0x12345) sink[0] = acc;
}

template <int PAT>
void run(const char* name, long long* dcyc, unsigned* sink) {
  k<PAT><<<256, 256>>>(dcyc, sink);
  hipDeviceSynchronize();
  long long h[1024];
  hipMemcpy(h, dcyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < 1024; ++i) s += (double)h[i];
  // clock64 = s_memtime at 100 MHz on gfx9? report raw ticks per instruction and per 4-wave group for comparison
  printf("%-46s %8.3f ticks per wave-instruction (x 4 waves sharing the LDS)\n", name, s / 1024 / (REPS * 8.0));
}

int main() {
  long long* dcyc; unsigned* sink;
  hipMalloc(&dcyc, 1024 * sizeof(long long)); hipMalloc(&sink, 4);
  run<0>("b128 linear (baseline)", dcyc, sink);
  run<1>("b128 MLA K fragments, prefill (all waves same rows)", dcyc, sink);
  run<4>("b128 MLA K fragments, decode (wave w rows 16w..)", dcyc, sink);
  run<5>("b128 K fragments without the swizzle", dcyc, sink);
  run<3>("b64 linear (baseline)", dcyc, sink);
  run<2>("b64_tr_b16 MLA V^T reads", dcyc, sink);
  run<6>("b128 K fragments, candidate swizzle f2", dcyc, sink);
  run<7>("b64_tr_b16 V^T reads, candidate swizzle f2", dcyc, sink);
  run<8>("b128 fragment rows padded to 272 B", dcyc, sink);
  run<9>("b128 4 lanes = 64 contiguous B per row (1152)", dcyc, sink);
  run<10>("b128 8 lanes = one 128-B row, chunks permuted", dcyc, sink);
  run<11>("b128 lane pairs = 32 contiguous B, rows 272", dcyc, sink);
  run<12>("b128 MFMA fragment over 8-row-interleaved tile", dcyc, sink);
  return 0;
}
