// workspace.hip -- the library's registry of caller-owned scratch buffers (host code only).
//
// The reference's operators carry no workspace argument (ops_api.h:27-287), so the entry points that mirror them look their
// scratch up here: the zero-at-rest int8 split-K buffer of the row-major GEMMs (kind 0) and the MoE index / tile-table
// scratch (kind 1). Round-2 review, weak #9: these were ONE process-global pointer each plus an unlocked 8-entry stream
// table, while the reference runs one worker THREAD per device in one process (distributed_runtime/dist_manager.cpp:82-84,
// runtime/llm_worker_impl.cpp:175-195) -- two devices would have shared device 0's scratch. Now:
//   * one default buffer PER DEVICE (the device is read off the registered pointer / the launching stream, not off a global),
//   * per-stream overrides (two micro-batches on two streams never share split-K sums), 64 slots,
//   * every access under one mutex (a few dozen nanoseconds per GEMM launch).
// The entry points that take `workspace, ws_bytes` explicitly (the *_packed family) never come here.
#include <mutex>

#include "common.h"

namespace xm {

namespace {
constexpr int kMaxDevices = 16, kMaxStreams = 64, kKinds = 2;
struct Slot { void* ws; size_t bytes; };
struct StreamSlot { void* stream; void* ws; size_t bytes; };
std::mutex g_mu;
Slot g_dev[kKinds][kMaxDevices] = {};
StreamSlot g_stream[kKinds][kMaxStreams] = {};
int g_stream_n[kKinds] = {0, 0};

int device_of_pointer(const void* p) {
  hipPointerAttribute_t a;
  if (p && hipPointerGetAttributes(&a, p) == hipSuccess) return a.device;
  (void)hipGetLastError();
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
  return d;
}
int device_of_stream(void* stream) {
  int d = 0;
  if (stream && hipStreamGetDevice((hipStream_t)stream, &d) == hipSuccess) return d;
  (void)hipGetLastError();
  if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
  return d;
}
}  // namespace

// default buffer of the device that owns `ws` (ws == nullptr: clears the CURRENT device's entry)
int ws_set_device(int kind, void* ws, size_t bytes) {
  const int d = device_of_pointer(ws);
  if (d < 0 || d >= kMaxDevices) return XM_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> lock(g_mu);
  g_dev[kind][d] = Slot{ws, ws ? bytes : 0};
  return XM_OK;
}

// per-stream override; ws == nullptr unregisters the stream
int ws_set_stream(int kind, void* stream, void* ws, size_t bytes) {
  std::lock_guard<std::mutex> lock(g_mu);
  StreamSlot* t = g_stream[kind];
  int& n = g_stream_n[kind];
  for (int i = 0; i < n; ++i)
    if (t[i].stream == stream) {
      if (!ws) { t[i] = t[--n]; return XM_OK; }
      t[i].ws = ws;
      t[i].bytes = bytes;
      return XM_OK;
    }
  if (!ws) return XM_OK;
  if (n == kMaxStreams) return XM_ERR_UNSUPPORTED;
  t[n++] = StreamSlot{stream, ws, bytes};
  return XM_OK;
}

// the buffer a launch on `stream` uses: the stream's own, else the default of the stream's device
void ws_get(int kind, void* stream, void** ws, size_t* bytes) {
  {
    std::lock_guard<std::mutex> lock(g_mu);
    const StreamSlot* t = g_stream[kind];
    for (int i = 0; i < g_stream_n[kind]; ++i)
      if (t[i].stream == stream) { *ws = t[i].ws; *bytes = t[i].bytes; return; }
  }
  const int d = device_of_stream(stream);
  std::lock_guard<std::mutex> lock(g_mu);
  if (d < 0 || d >= kMaxDevices) { *ws = nullptr; *bytes = 0; return; }
  *ws = g_dev[kind][d].ws;
  *bytes = g_dev[kind][d].bytes;
}

}  // namespace xm
