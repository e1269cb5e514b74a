// Error reporting and version for the C-ABI (include/lead_yolo_hip.h).  Host-only translation unit.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void ly_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ly_last_error(void) { return g_err; }

extern "C" int ly_abi_version(void) { return 2; }

// ---- stream events that cross a hipGraph boundary (data-parallel step) ---------------------------------
// A replayed graph of forward + backward must release the gradient exchange of a bucket the moment the bucket's last gradient kernel
// has run, while the rest of the backward is still executing.  `ly_event_record` on a capturing stream adds an EVENT-RECORD NODE behind
// the stream's current capture dependencies (hipGraphAddEventRecordNode + hipStreamUpdateCaptureDependencies); every replay then
// records the event when that node's dependencies have executed, and a stream outside the graph that calls `ly_stream_wait_event`
// after the replay was launched waits for exactly that.  Measured on MI355X / ROCm 7.2 (tools/ext_event_probe.py,
// profiles/r03_ext_event_probe.txt): the side stream is released 18.5 ms into a 73 ms graph and always sees the node's predecessors.
// (hipEventRecordWithFlags(..., hipEventRecordExternal) returns hipErrorInvalidValue on this runtime, and torch refuses
// `Event(external=True)` on ROCm — hence the explicit node.)  On a stream that is not capturing it is a plain hipEventRecord.
#include <hip/hip_runtime.h>

#define LY_HIP_OK(call, what)                                                            \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      ly_set_error("%s: %s", what, hipGetErrorString(e_));                               \
      (void)hipGetLastError();                                                           \
      return -1;                                                                         \
    }                                                                                    \
  } while (0)

extern "C" int ly_event_create(void** event) {
  if (!event) { ly_set_error("ly_event_create: null"); return -1; }
  hipEvent_t ev;
  LY_HIP_OK(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "ly_event_create");
  *event = ev;
  return 0;
}

extern "C" int ly_event_destroy(void* event) {
  if (event) LY_HIP_OK(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)), "ly_event_destroy");
  return 0;
}

extern "C" int ly_event_record(void* event, void* stream) {
  if (!event) { ly_set_error("ly_event_record: null event"); return -1; }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipEvent_t ev = reinterpret_cast<hipEvent_t>(event);
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  LY_HIP_OK(hipStreamGetCaptureInfo_v2(st, &status, &id, &graph, &deps, &ndeps), "ly_event_record: capture info");
  if (status != hipStreamCaptureStatusActive) {
    LY_HIP_OK(hipEventRecord(ev, st), "ly_event_record");
    return 0;
  }
  hipGraphNode_t node = nullptr;
  LY_HIP_OK(hipGraphAddEventRecordNode(&node, graph, deps, ndeps, ev), "ly_event_record: event-record node");
  LY_HIP_OK(hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies), "ly_event_record: capture dependencies");
  return 0;
}

extern "C" int ly_stream_wait_event(void* stream, void* event) {
  if (!event) { ly_set_error("ly_stream_wait_event: null event"); return -1; }
  LY_HIP_OK(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), reinterpret_cast<hipEvent_t>(event), 0), "ly_stream_wait_event");
  return 0;
}
