// allreduce.hip -- one-shot SUM all-reduce of the small TP messages of a decode step over peer-mapped buffers.
//
// What it replaces: parallel_state::reduce -> ProcessGroup::allreduce -> c10d::ProcessGroupNCCL (RCCL ring)
// (framework/parallel_state/parallel_state.cpp:183-192, process_group.cpp:98-108) for messages of at most a few MiB: a
// Qwen2-7B decode step at TP = 4 issues 57 all-reduces of 256 x 3584 bf16 = 1.75 MiB. A ring moves 2 (W-1)/W of the message
// through 2 (W-1) dependent hops; on the fully connected xGMI mesh of one node every rank can instead read every peer's
// message directly: ONE hop, (W-1)/W of the message per incoming link, all links busy at once.
//
// Protocol (per rank one shared buffer, allocated by xllm_mi355_ipc_alloc and mapped into every peer through
// hipIpcGetMemHandle / hipIpcOpenMemHandle): [flags u32 [2][kArBlocks][kArMaxWorld]] [data slot 0] [data slot 1].
// A launch with epoch e uses slot e & 1. Block b of rank r:
//   1. copies slice b of its input into its own slot, fences (system scope), and stores e into flag [e&1][b][r] of EVERY
//      rank's buffer (release, system scope);
//   2. waits until its own flags [e&1][b][0..W) all hold e (acquire, system scope; bounded by a wall-clock timeout that
//      raises a status word instead of hanging the queue);
//   3. reads slice b from every rank's slot IN RANK ORDER, sums in fp32, writes the result over its input slice.
// The same order on every rank makes the result bit-identical across ranks. No second barrier: slot e & 1 is next written
// at epoch e + 2, which a rank can only enter after epoch e + 1, whose step 2 needs every peer's flag of e + 1, which a peer
// raises only after it has finished epoch e (kernels of one rank are ordered on its stream). The epoch lives in device
// memory and is advanced by the last block to leave, so the launch is a plain kernel: capturable into a HIP graph, and a
// TP step needs no eager collective between graph pieces.
#include "common.h"
#include "gemm_types.h"

namespace xm {

constexpr int kArMaxWorld = 8;
// flag rows per slot = the largest grid of a launch. 256 since round 4 (was 64): with `grid_limit` = 256 the fused add + norm
// kernel gives a decode-step message of 256 rows ONE row per block, so step 3 -- peer loads over xGMI, two block-wide reductions,
// the stores -- is one dependent chain per block instead of four in a row (round-3 review, next #8.iii). The CALLER picks the
// limit (0 = 64): every block spin-waits for its peers' blocks, so all ranks' grids must be co-resident -- one rank per GPU may
// use 256, ranks that SHARE a GPU (the two-process protocol test) must stay at 64 (256 blocks of 512 threads of one rank fill the
// chip with waiting blocks: measured time-out, round 3). The plain kernel keeps its 64 blocks of >= 16 KiB.
constexpr int kArBlocks = 256;
constexpr int kArPlainBlocks = 64;
constexpr size_t kArFlagBytes = 2 * kArBlocks * kArMaxWorld * sizeof(uint32_t);   // 16 KiB
constexpr int kArThreads = 512;

struct ArPeers {
  char* base[kArMaxWorld];
};

template <typename T>
__global__ __launch_bounds__(kArThreads) void oneshot_allreduce_kernel(ArPeers peers, T* __restrict__ inout, int64_t n_vec,
                                                                      int rank, int world, int64_t slot_bytes,
                                                                      uint32_t* __restrict__ epoch_state,
                                                                      int* __restrict__ status, long long timeout_ticks) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int N = Vec16B<T>::N;
  const uint32_t epoch = epoch_state[0] + 1;   // every block reads it before the last one to leave advances it
  const int buf = epoch & 1;
  const int b = blockIdx.x;
  const int64_t per = (n_vec + gridDim.x - 1) / gridDim.x;
  const int64_t v0 = (int64_t)b * per;
  const int64_t v1 = v0 + per < n_vec ? v0 + per : n_vec;
  u32x4* io = reinterpret_cast<u32x4*>(inout);

  // 1. my slice -> my slot
  u32x4* mine = reinterpret_cast<u32x4*>(peers.base[rank] + kArFlagBytes + (int64_t)buf * slot_bytes);
  for (int64_t i = v0 + threadIdx.x; i < v1; i += kArThreads) mine[i] = io[i];
  __threadfence_system();
  __syncthreads();
  if ((int)threadIdx.x < world) {
    uint32_t* f = reinterpret_cast<uint32_t*>(peers.base[threadIdx.x]) + ((int64_t)buf * kArBlocks + b) * kArMaxWorld + rank;
    __hip_atomic_store(f, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 2. wait for slice b of every rank
  __shared__ int timed_out;
  if (threadIdx.x == 0) timed_out = 0;
  __syncthreads();
  if ((int)threadIdx.x < world) {
    const uint32_t* f = reinterpret_cast<const uint32_t*>(peers.base[rank]) + ((int64_t)buf * kArBlocks + b) * kArMaxWorld + threadIdx.x;
    const long long t0 = wall_clock64();
    // a launch that already finds the status word raised does not wait at all: after ONE timed-out wait every later collective
    // fails fast (its result is undefined anyway and the host drops the path at its next check), so a dead peer costs one
    // timeout per step sequence, not one per collective
    const long long limit = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 0 : timeout_ticks;
    while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
      if (wall_clock64() - t0 > limit) { timed_out = 1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  if (timed_out) {
    if (threadIdx.x == 0) atomicExch(status, 1);   // the result of this launch is undefined; the caller reads status
  } else {
    // 3. sum the slices in rank order (fp32), one rounding into T
    for (int64_t i = v0 + threadIdx.x; i < v1; i += kArThreads) {
      float acc[N];
#pragma unroll
      for (int j = 0; j < N; ++j) acc[j] = 0.0f;
      for (int p = 0; p < world; ++p) {
        const u32x4* src = reinterpret_cast<const u32x4*>(peers.base[p] + kArFlagBytes + (int64_t)buf * slot_bytes);
        // volatile = system-coherent load (sc0 sc1): the bytes were written by another agent while this kernel runs; the
        // flag acquire above (threads < world) + the barrier order this read after the peer's release
        const u32x4 v = *reinterpret_cast<const volatile u32x4*>(&src[i]);
        const T* e = reinterpret_cast<const T*>(&v);
#pragma unroll
        for (int j = 0; j < N; ++j) acc[j] += to_f32<T>(e[j]);
      }
      u32x4 o;
      T* oe = reinterpret_cast<T*>(&o);
#pragma unroll
      for (int j = 0; j < N; ++j) oe[j] = from_f32<T>(acc[j]);
      io[i] = o;
    }
  }
  // the last block to leave advances the epoch (all blocks have read it by then)
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&epoch_state[1], 1u) == gridDim.x - 1) {
      epoch_state[1] = 0;
      __threadfence();
      epoch_state[0] = epoch;
    }
  }
}


// ---- one-shot all-reduce FUSED with the residual add + RMSNorm (+ per-token int8 quant) that follows it (round 3) ----------
// A tensor-parallel half-layer is: row-parallel linear -> SUM all-reduce -> fused_add_rms_norm -> (scaled_quantize of the next
// W8A8 linear) (linear.cpp:1518-1520, qwen2_decoder_layer.cpp:66-110). In step 3 of the protocol above every block already
// holds the summed values of its slice in registers, so when the slices are whole token rows the block finishes the row:
//   y = rT(sum over ranks, fp32, rank order)                  -- the all-reduce result, as xllm_mi355_oneshot_allreduce
//   residual <- rT(y + residual); n = rT(rT(residual * inv_rms) * w)       -- rms_norm_kernel<T, ADD, QUANT> of rowwise.hip
//   QUANT: out_q = int8(n * 127 / amax), out_scale = amax / 127; else out_norm = n
// bit-identical to xllm_mi355_oneshot_allreduce followed by xllm_mi355_fused_add_rms_norm (/ _rms_norm_dynamic_int8_quant),
// which is how the tests check it. `residual` is rank-local (every rank holds the same residual and ends with the same bits).
// Block b owns rows [b * rpb, (b + 1) * rpb); flags, slots and the epoch are shared with the plain kernel (same buffer).
constexpr int kArNormMaxVec = 4;   // 16-byte chunks per thread: rows up to 512 * 4 * 8 = 16384 elements

// this rank's partial sums as the row-parallel W8A8 GEMM left them: n_slabs K-slice slabs of exact int32 sums (gemm_ws.hip, defer
// mode) + the dequant operands. Step 1 of the fused kernel then computes the rank's 16-bit partial itself -- r16(sum * a_s[m] *
// w_s[n] + bias[n]), the expression of ws_slab_epilogue_kernel -- instead of copying one: the TP half-layer is GEMM -> this kernel.
struct ArSlabs {
  const int32_t* slabs;   // null: the 16-bit `partial` is the input
  int n_slabs;
  int64_t stride;         // elements between slabs (M * H)
  const float* a_scale;
  const float* w_scale;
  const void* bias;
};

template <typename T, bool QUANT>
__global__ __launch_bounds__(kArThreads) void oneshot_allreduce_norm_kernel(
    ArPeers peers, ArSlabs sl, const T* __restrict__ partial, T* __restrict__ residual, const T* __restrict__ weight, float eps,
    T* __restrict__ out_norm, int8_t* __restrict__ out_q, float* __restrict__ out_scale, T* __restrict__ out_sum, int M, int H,
    int rows_per_block, int rank, int world, int64_t slot_bytes, uint32_t* __restrict__ epoch_state, int* __restrict__ status,
    long long timeout_ticks) {
  static_assert(sizeof(T) == 2, "16-bit activations");
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  constexpr int N = 8;
  __shared__ float smem[32];
  __shared__ int timed_out;
  const uint32_t epoch = epoch_state[0] + 1;
  const int buf = epoch & 1;
  const int b = blockIdx.x;
  const int nvec = H / N;
  const int r0 = b * rows_per_block, r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  const u32x4* in = reinterpret_cast<const u32x4*>(partial);
  // 1. my rows -> my slot
  u32x4* mine = reinterpret_cast<u32x4*>(peers.base[rank] + kArFlagBytes + (int64_t)buf * slot_bytes);
  if (sl.slabs) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const T* bias = reinterpret_cast<const T*>(sl.bias);
    for (int64_t i = (int64_t)r0 * nvec + threadIdx.x; i < (int64_t)r1 * nvec; i += kArThreads) {
      const int r = (int)(i / nvec), c = (int)(i - (int64_t)r * nvec);
      const int32_t* row = sl.slabs + (int64_t)r * H;
      i32x4 a0 = reinterpret_cast<const i32x4*>(row)[2 * c], a1 = reinterpret_cast<const i32x4*>(row)[2 * c + 1];
      for (int s0 = 1; s0 < sl.n_slabs; s0 += 7) {   // (at most 8 slices: one pass with every load in flight)
        i32x4 b0[7], b1[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) {
          const bool ok = s0 + u < sl.n_slabs;
          const int32_t* p = row + (ok ? s0 + u : 0) * sl.stride;
          b0[u] = reinterpret_cast<const i32x4*>(p)[2 * c];
          b1[u] = reinterpret_cast<const i32x4*>(p)[2 * c + 1];
          if (!ok) { b0[u] = i32x4{0, 0, 0, 0}; b1[u] = i32x4{0, 0, 0, 0}; }
        }
        a0 += ((b0[0] + b0[1]) + (b0[2] + b0[3])) + ((b0[4] + b0[5]) + b0[6]);
        a1 += ((b1[0] + b1[1]) + (b1[2] + b1[3])) + ((b1[4] + b1[5]) + b1[6]);
      }
      const float as = sl.a_scale[r];
      const float4 w0 = reinterpret_cast<const float4*>(sl.w_scale)[2 * c], w1 = reinterpret_cast<const float4*>(sl.w_scale)[2 * c + 1];
      const int av[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      RowVec<T> bv, yv;
      bv.raw = bias ? reinterpret_cast<const uint4*>(bias)[c] : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < N; ++j) yv.set(j, (float)av[j] * as * wv[j] + (bias ? bv.get(j) : 0.0f));   // the GEMM's 16-bit output
      mine[i] = u32x4{yv.raw.x, yv.raw.y, yv.raw.z, yv.raw.w};
    }
  } else {
    for (int64_t i = (int64_t)r0 * nvec + threadIdx.x; i < (int64_t)r1 * nvec; i += kArThreads) mine[i] = in[i];
  }
  __threadfence_system();
  __syncthreads();
  if ((int)threadIdx.x < world) {
    uint32_t* f = reinterpret_cast<uint32_t*>(peers.base[threadIdx.x]) + ((int64_t)buf * kArBlocks + b) * kArMaxWorld + rank;
    __hip_atomic_store(f, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 2. wait for rows r0 .. r1 of every rank
  if (threadIdx.x == 0) timed_out = 0;
  __syncthreads();
  if ((int)threadIdx.x < world) {
    const uint32_t* f = reinterpret_cast<const uint32_t*>(peers.base[rank]) + ((int64_t)buf * kArBlocks + b) * kArMaxWorld + threadIdx.x;
    const long long t0 = wall_clock64();
    // a launch that already finds the status word raised does not wait at all: after ONE timed-out wait every later collective
    // fails fast (its result is undefined anyway and the host drops the path at its next check), so a dead peer costs one
    // timeout per step sequence, not one per collective
    const long long limit = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 0 : timeout_ticks;
    while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
      if (wall_clock64() - t0 > limit) { timed_out = 1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  if (timed_out) {
    if (threadIdx.x == 0) atomicExch(status, 1);
  } else {
    // 3. row by row: sum in rank order, then the add + norm (+ quant) of rowwise.hip's rms_norm_kernel<T, true, QUANT>
    for (int r = r0; r < r1; ++r) {
      RowVec<T> xv[kArNormMaxVec];
      float ss = 0.0f;
      T* res_row = residual + (int64_t)r * H;
#pragma unroll
      for (int i = 0; i < kArNormMaxVec; ++i) {
        const int c = threadIdx.x + i * kArThreads;
        if (c < nvec) {
          float acc[N];
#pragma unroll
          for (int j = 0; j < N; ++j) acc[j] = 0.0f;
          for (int p = 0; p < world; ++p) {
            const u32x4* src = reinterpret_cast<const u32x4*>(peers.base[p] + kArFlagBytes + (int64_t)buf * slot_bytes);
            const u32x4 v = *reinterpret_cast<const volatile u32x4*>(&src[(int64_t)r * nvec + c]);
            RowVec<T> pv;
            pv.raw = make_uint4(v.x, v.y, v.z, v.w);
#pragma unroll
            for (int j = 0; j < N; ++j) acc[j] += pv.get(j);
          }
          RowVec<T> yv, rv;
#pragma unroll
          for (int j = 0; j < N; ++j) yv.set(j, acc[j]);                       // y = rT(sum): the all-reduce result
          if (out_sum) reinterpret_cast<uint4*>(out_sum + (int64_t)r * H)[c] = yv.raw;
          rv.raw = reinterpret_cast<const uint4*>(res_row)[c];
#pragma unroll
          for (int j = 0; j < N; ++j) xv[i].set(j, yv.get(j) + rv.get(j));     // rT(y + residual)
          reinterpret_cast<uint4*>(res_row)[c] = xv[i].raw;
#pragma unroll
          for (int j = 0; j < N; ++j) { const float x = xv[i].get(j); ss += x * x; }
        }
      }
      ss = block_sum(ss, smem);
      const float inv = 1.0f / sqrtf(ss / (float)H + eps);
      float amax = 0.0f;
#pragma unroll
      for (int i = 0; i < kArNormMaxVec; ++i) {
        const int c = threadIdx.x + i * kArThreads;
        if (c < nvec) {
          RowVec<T> wv;
          wv.raw = reinterpret_cast<const uint4*>(weight)[c];
#pragma unroll
          for (int j = 0; j < N; ++j) {
            const float y = r16<T>(r16<T>(xv[i].get(j) * inv) * wv.get(j));
            xv[i].set(j, y);
            amax = fmaxf(amax, fabsf(y));
          }
          if constexpr (!QUANT) reinterpret_cast<uint4*>(out_norm + (int64_t)r * H)[c] = xv[i].raw;
        }
      }
      if constexpr (QUANT) {
        amax = block_max(amax, smem);
        const float qinv = (amax > 1e-10f) ? 127.0f / amax : 0.0f;
#pragma unroll
        for (int i = 0; i < kArNormMaxVec; ++i) {
          const int c = threadIdx.x + i * kArThreads;
          if (c < nvec) {
            uint32_t pk[2];
#pragma unroll
            for (int j = 0; j < N; j += 4) {
              uint32_t w = 0;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float qv = fmaxf(-127.0f, fminf(127.0f, rintf(xv[i].get(j + e) * qinv)));
                w |= ((uint32_t)(int)qv & 0xffu) << (8 * e);
              }
              pk[j / 4] = w;
            }
            *reinterpret_cast<uint2*>(out_q + (int64_t)r * H + (int64_t)c * N) = make_uint2(pk[0], pk[1]);
          }
        }
        if (threadIdx.x == 0) out_scale[r] = amax / 127.0f;
      }
      __syncthreads();   // smem is reused by the next row's reductions
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(&epoch_state[1], 1u) == gridDim.x - 1) {
      epoch_state[1] = 0;
      __threadfence();
      epoch_state[0] = epoch;
    }
  }
}

}  // namespace xm

using namespace xm;

extern "C" {

size_t xllm_mi355_oneshot_allreduce_buffer_bytes(size_t max_message_bytes) {
  const size_t slot = (max_message_bytes + 255) / 256 * 256;
  return kArFlagBytes + 2 * slot;
}

int xllm_mi355_ipc_alloc(size_t bytes, void** ptr, int* kind) {
  if (!ptr || bytes == 0) return XM_ERR_INVALID;
  // flags and data are polled / read by other agents while a kernel runs: fine-grained (or uncached) device memory
  // *kind on entry: the first kind to try (0 fine-grained, 1 uncached, 2 plain hipMalloc); on return: the kind obtained
  int k = kind ? *kind : 0;
  hipError_t e = hipErrorInvalidValue;
  for (; k <= 2 && e != hipSuccess; ++k) {
    if (k < 0) k = 0;
    e = k == 0 ? hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained)
               : (k == 1 ? hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocUncached) : hipMalloc(ptr, bytes));
    if (e != hipSuccess) (void)hipGetLastError();
  }
  --k;
  if (e != hipSuccess) return XM_ERR_HIP;
  if (hipMemset(*ptr, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return XM_ERR_HIP;
  if (kind) *kind = k;
  return XM_OK;
}

int xllm_mi355_ipc_free(void* ptr) {
  if (!ptr) return XM_ERR_INVALID;
  return hipFree(ptr) == hipSuccess ? XM_OK : XM_ERR_HIP;
}

int xllm_mi355_ipc_get_handle(void* dev_ptr, void* handle64) {
  if (!dev_ptr || !handle64) return XM_ERR_INVALID;
  static_assert(sizeof(hipIpcMemHandle_t) == XLLM_MI355_IPC_HANDLE_BYTES, "handle size");
  hipIpcMemHandle_t h;
  if (hipIpcGetMemHandle(&h, dev_ptr) != hipSuccess) { (void)hipGetLastError(); return XM_ERR_HIP; }
  __builtin_memcpy(handle64, &h, sizeof(h));
  return XM_OK;
}

int xllm_mi355_ipc_open_handle(const void* handle64, void** ptr) {
  if (!handle64 || !ptr) return XM_ERR_INVALID;
  hipIpcMemHandle_t h;
  __builtin_memcpy(&h, handle64, sizeof(h));
  if (hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); return XM_ERR_HIP; }
  return XM_OK;
}

int xllm_mi355_ipc_close_handle(void* ptr) {
  if (!ptr) return XM_ERR_INVALID;
  return hipIpcCloseMemHandle(ptr) == hipSuccess ? XM_OK : XM_ERR_HIP;
}

int xllm_mi355_oneshot_allreduce(void* inout, int64_t count, int dtype, void* const* peer_buffers, int rank, int world,
                                 size_t max_message_bytes, uint32_t* epoch_state, int* status, double timeout_s,
                                 void* stream) {
  if (!inout || !peer_buffers || !epoch_state || !status || count < 0 || world < 1 || world > kArMaxWorld || rank < 0 ||
      rank >= world)
    return XM_ERR_INVALID;
  if (count == 0) return XM_OK;
  const size_t esz = dtype == XM_F32 ? 4 : 2;
  if (dtype != XM_F32 && dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  const size_t bytes = (size_t)count * esz;
  if (bytes % 16 || (uintptr_t)inout % 16) return XM_ERR_UNSUPPORTED;
  if (bytes > max_message_bytes) return XM_ERR_WORKSPACE;
  ArPeers peers;
  for (int p = 0; p < kArMaxWorld; ++p) {
    peers.base[p] = p < world ? (char*)peer_buffers[p] : nullptr;
    if (p < world && !peers.base[p]) return XM_ERR_INVALID;
  }
  const int64_t n_vec = (int64_t)(bytes / 16);
  // enough blocks to keep the incoming links busy, few enough that the flag traffic stays negligible: >= 16 KiB per block
  int64_t grid = (n_vec * 16 + 16383) / 16384;
  grid = grid < 1 ? 1 : (grid > kArPlainBlocks ? kArPlainBlocks : grid);
  const int64_t slot_bytes = (int64_t)((max_message_bytes + 255) / 256 * 256);
  const long long ticks = (long long)((timeout_s > 0 ? timeout_s : 2.0) * 1e8);   // wall_clock64: 100 MHz
  hipStream_t s = (hipStream_t)stream;
#define XM_AR(T)                                                                                                     \
  hipLaunchKernelGGL((oneshot_allreduce_kernel<T>), dim3((unsigned)grid), dim3(kArThreads), 0, s, peers, (T*)inout, \
                     n_vec, rank, world, slot_bytes, epoch_state, status, ticks)
  if (dtype == XM_F32) XM_AR(float);
  else if (dtype == XM_BF16) XM_AR(bf16_t);
  else XM_AR(f16_t);
#undef XM_AR
  return hip_check_launch();
}

static int launch_oneshot_norm(xm::ArSlabs sl, const void* partial, void* residual, const void* norm_weight, float eps,
                               void* out_norm, int8_t* out_q, float* out_q_scale, void* out_sum, int64_t M, int64_t H, int dtype,
                               void* const* peer_buffers, int rank, int world, size_t max_message_bytes, uint32_t* epoch_state,
                               int* status, double timeout_s, int grid_limit, void* stream) {
  if ((!partial && !sl.slabs) || !residual || !norm_weight || !peer_buffers || !epoch_state || !status || M < 0 || H <= 0 ||
      world < 1 || world > kArMaxWorld || rank < 0 || rank >= world)
    return XM_ERR_INVALID;
  if ((out_q != nullptr) == (out_norm != nullptr)) return XM_ERR_INVALID;  // exactly one output form
  if (out_q && !out_q_scale) return XM_ERR_INVALID;
  if (dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (H % 8 || H > (int64_t)kArThreads * kArNormMaxVec * 8 ||
      (((uintptr_t)partial | (uintptr_t)residual | (uintptr_t)norm_weight | (uintptr_t)out_norm | (uintptr_t)out_q |
        (uintptr_t)out_sum) % 16))
    return XM_ERR_UNSUPPORTED;
  if (M == 0) return XM_OK;
  const size_t bytes = (size_t)M * H * 2;
  if (bytes > max_message_bytes) return XM_ERR_WORKSPACE;
  ArPeers peers;
  for (int p = 0; p < kArMaxWorld; ++p) {
    peers.base[p] = p < world ? (char*)peer_buffers[p] : nullptr;
    if (p < world && !peers.base[p]) return XM_ERR_INVALID;
  }
  // whole rows per block; at most `grid_limit` (<= kArBlocks) blocks (one flag row each)
  if (grid_limit < 0 || grid_limit > kArBlocks) return XM_ERR_INVALID;
  const int max_blocks = grid_limit > 0 ? grid_limit : kArPlainBlocks;
  const int rpb = (int)((M + max_blocks - 1) / max_blocks);
  const int grid = (int)((M + rpb - 1) / rpb);
  const int64_t slot_bytes = (int64_t)((max_message_bytes + 255) / 256 * 256);
  const long long ticks = (long long)((timeout_s > 0 ? timeout_s : 2.0) * 1e8);
  hipStream_t s = (hipStream_t)stream;
#define XM_ARN(T, Q)                                                                                                  \
  hipLaunchKernelGGL((oneshot_allreduce_norm_kernel<T, Q>), dim3((unsigned)grid), dim3(kArThreads), 0, s, peers,     \
                     sl, (const T*)partial, (T*)residual, (const T*)norm_weight, eps, (T*)out_norm, out_q, out_q_scale, \
                     (T*)out_sum, (int)M, (int)H, rpb, rank, world, slot_bytes, epoch_state, status, ticks)
  if (dtype == XM_BF16) { if (out_q) XM_ARN(bf16_t, true); else XM_ARN(bf16_t, false); }
  else { if (out_q) XM_ARN(f16_t, true); else XM_ARN(f16_t, false); }
#undef XM_ARN
  return hip_check_launch();
}

int xllm_mi355_oneshot_allreduce_add_rms_norm(const void* partial, void* residual, const void* norm_weight, float eps,
                                              void* out_norm, int8_t* out_q, float* out_q_scale, void* out_sum, int64_t M,
                                              int64_t H, int dtype, void* const* peer_buffers, int rank, int world,
                                              size_t max_message_bytes, uint32_t* epoch_state, int* status, double timeout_s,
                                              int grid_limit, void* stream) {
  if (!partial) return XM_ERR_INVALID;
  return launch_oneshot_norm(xm::ArSlabs{nullptr, 0, 0, nullptr, nullptr, nullptr}, partial, residual, norm_weight, eps, out_norm,
                             out_q, out_q_scale, out_sum, M, H, dtype, peer_buffers, rank, world, max_message_bytes, epoch_state,
                             status, timeout_s, grid_limit, stream);
}

int xllm_mi355_scaled_matmul_oneshot_allreduce_add_rms_norm(
    const int8_t* a, const int8_t* w_packed, const float* a_scale, const float* w_scale, const void* bias, void* residual,
    const void* norm_weight, float eps, void* out_norm, int8_t* out_q, float* out_q_scale, void* out_sum, int64_t M, int64_t N,
    int64_t K, int dtype, void* workspace, size_t ws_bytes, void* const* peer_buffers, int rank, int world,
    size_t max_message_bytes, uint32_t* epoch_state, int* status, double timeout_s, int grid_limit, void* stream) {
  if (!a || !w_packed || !a_scale || !w_scale || !workspace || M < 0 || N <= 0 || K <= 0) return XM_ERR_INVALID;
  if (dtype != XM_BF16 && dtype != XM_F16) return XM_ERR_UNSUPPORTED;
  if (M == 0) return XM_OK;
  if ((size_t)M * N * 2 > max_message_bytes) return XM_ERR_WORKSPACE;
  if (N % 8 || N > 16384 || ((uintptr_t)w_scale % 16) || (bias && (uintptr_t)bias % 16) || M > 512)
    return XM_ERR_UNSUPPORTED;   // (the fused consumer's envelope, checked before the GEMM is launched: a decline has no side effect)
  // the row-parallel GEMM leaves its exact int32 K-slice sums in `workspace` (no dequant pass) ...
  xm::GemmEpi epi{a_scale, M, w_scale, N, bias, nullptr, nullptr, dtype == XM_BF16, nullptr, 0};
  epi.defer = 1;
  int n_slabs = 0;
  const int rc = xm::launch_gemm_ws_i8(a, w_packed, M, N, K, epi, workspace, ws_bytes, &n_slabs, (hipStream_t)stream);
  if (rc != XM_OK) return rc;
  // ... and step 1 of the one-shot kernel turns them into this rank's 16-bit partial on its way into the exchange slot
  return launch_oneshot_norm(xm::ArSlabs{reinterpret_cast<const int32_t*>(workspace), n_slabs, M * N, a_scale, w_scale, bias},
                             nullptr, residual, norm_weight, eps, out_norm, out_q, out_q_scale, out_sum, M, N, dtype,
                             peer_buffers, rank, world, max_message_bytes, epoch_state, status, timeout_s, grid_limit, stream);
}

}  // extern "C"
