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

namespace xm {

constexpr int kArMaxWorld = 8;
constexpr int kArBlocks = 64;                                  // grid upper bound = flag rows per slot
constexpr size_t kArFlagBytes = 2 * kArBlocks * kArMaxWorld * sizeof(uint32_t);   // 4 KiB
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
    while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
      if (wall_clock64() - t0 > timeout_ticks) { timed_out = 1; break; }
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
  grid = grid < 1 ? 1 : (grid > kArBlocks ? kArBlocks : grid);
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

}  // extern "C"
