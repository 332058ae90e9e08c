// Launch metering (measurement aid behind bench.py's `roofline` object; include/ape_hip.h ape_hip_meter_*).
// While metering is on for the calling thread, every kernel the library launches goes out through hipExtLaunchKernelGGL with its OWN
// (start, stop) event pair: the pair carries the dispatch's begin / end timestamps -- the same two numbers rocprofv3's kernel trace
// reports -- so hipEventElapsedTime(start, stop) is the kernel's duration without the command-processor gaps an event pair RECORDED
// around a launch includes (~4.5 us per launch on MI355X: BENCH_r04 71.98 us vs rocprofv3 67.49 us for the same kernel).
// Not for captured streams (a capture records no timestamps); off by default; no cost when off beyond one thread-local test per launch.
#include <hip/hip_runtime.h>

#include <vector>

void ape_set_error(const char* fmt, ...);

namespace {
struct Rec {
  const char* name;   // string literal (the kernel expression at the launch site)
  hipEvent_t ev[2];
};
thread_local bool g_on = false;
thread_local std::vector<Rec> g_recs;                 // launches of the current session, in order
thread_local std::vector<hipEvent_t> g_pool;          // events owned by this thread, reused across sessions
thread_local size_t g_pool_used = 0;

hipEvent_t take_event() {
  if (g_pool_used == g_pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_pool.push_back(e);
  }
  return g_pool[g_pool_used++];
}
}  // namespace

// called by APE_LAUNCH (common.h): the event pair of the launch about to be issued, or nullptr when metering is off
hipEvent_t* ape_meter_pair(const char* kernel, hipStream_t stream) {
  if (!g_on) return nullptr;
  // a stream under capture records no timestamps, and hipExtLaunchKernelGGL with events would fail or invalidate the capture: such a
  // launch goes out plain and simply is not metered
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cs) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  if (cs != hipStreamCaptureStatusNone) return nullptr;
  Rec r;
  r.name = kernel;
  const size_t used = g_pool_used;
  r.ev[0] = take_event();
  r.ev[1] = take_event();
  if (r.ev[0] == nullptr || r.ev[1] == nullptr) {
    g_pool_used = used;             // roll back: the slot taken for the first event is not lost when the second cannot be created
    (void)hipGetLastError();
    return nullptr;
  }
  g_recs.push_back(r);
  return g_recs.back().ev;          // valid until the next push_back: APE_LAUNCH uses it at once
}

extern "C" int ape_hip_meter_begin(void) {
  g_recs.clear();
  g_pool_used = 0;
  g_on = true;
  return 0;
}

extern "C" int ape_hip_meter_count(void) { return (int)g_recs.size(); }

extern "C" int ape_hip_meter_end(void) {
  g_on = false;
  return (int)g_recs.size();
}

// launch i of the session: its kernel expression and duration in milliseconds (waits for that launch to finish)
extern "C" int ape_hip_meter_read(int i, const char** name, float* ms) {
  if (i < 0 || i >= (int)g_recs.size()) {
    ape_set_error("ape_hip_meter_read: launch %d of %d", i, (int)g_recs.size());
    return -1;
  }
  const Rec& r = g_recs[i];
  if (name != nullptr) *name = r.name;
  if (ms != nullptr) {
    hipError_t rc = hipEventSynchronize(r.ev[1]);
    if (rc == hipSuccess) rc = hipEventElapsedTime(ms, r.ev[0], r.ev[1]);
    if (rc != hipSuccess) {
      ape_set_error("ape_hip_meter_read: %s (launch %d, %s)", hipGetErrorString(rc), i, r.name);
      return -2;
    }
  }
  return 0;
}

// Zero-fill of a device buffer on a stream (hipMemsetAsync: a memset node when the stream is being captured).  The forward pass has two
// operand buffers whose padding columns must be finite -- the transposed V of the ViT and of the decoder's self-attention -- and cleared
// them with the tensor library's fill kernel; this entry makes that the library's own call (ape_amd.ops.zeros).
extern "C" int ape_hip_zero(void* ptr, size_t nbytes, void* stream) {
  if (nbytes == 0) return 0;
  if (ptr == nullptr) {
    ape_set_error("ape_hip_zero: null pointer");
    return -1;
  }
  const hipError_t rc = hipMemsetAsync(ptr, 0, nbytes, (hipStream_t)stream);
  if (rc != hipSuccess) {
    ape_set_error("ape_hip_zero: %s", hipGetErrorString(rc));
    return -2;
  }
  return 0;
}
