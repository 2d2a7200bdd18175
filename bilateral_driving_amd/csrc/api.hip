// libbds.so: ABI version + error strings.
#include "bds_common.h"

extern "C" int bds_abi_version(void) { return BDS_ABI_VERSION; }

extern "C" const char *bds_strerror(int code) {
  switch (code) {
    case BDS_OK: return "ok";
    case BDS_EINVAL: return "invalid argument (null/misaligned pointer, bad shape or unsupported parameter)";
    case BDS_EWORKSPACE: return "workspace too small";
    case BDS_ELAUNCH: return "HIP launch or copy failed";
    case BDS_ECAPACITY: return "more intersections than the caller's buffers hold";
    default: return "unknown bds error";
  }
}

namespace bds {
static int g_options[kOptCount] = {/*cap_launch*/ 1, 0, 0, /*debug*/ 0, /*short_sort*/ 1, 0, /*packed*/ 1, /*cells*/ 3, /*sched_bins*/ 1};
int option_get(int which) { return (which >= 0 && which < kOptCount) ? g_options[which] : 0; }
}  // namespace bds

extern "C" int bds_set_option(int which, int value) {
  if (which < 0 || which >= bds::kOptCount) return BDS_EINVAL;
  bds::g_options[which] = value;
  return BDS_OK;
}
extern "C" int bds_get_option(int which) { return bds::option_get(which); }

// ---- timing marks that survive graph capture -------------------------------------------------------------------------------
// An event recorded on a capturing stream with the plain API only orders the capture (it never holds a timestamp); recorded with
// hipEventRecordExternal it becomes an event-record NODE of the graph and is re-recorded by every replay.  bench.py brackets the
// roofline kernel with a pair of these inside the captured view.
static thread_local int g_last_hip_error = 0;
// HIP error code of the last failed runtime call made by the timing-mark entry points (diagnostics)
extern "C" int bds_last_hip_error(void) { return g_last_hip_error; }

extern "C" void *bds_timer_create(void) {
  hipEvent_t ev = nullptr;
  const hipError_t rc = hipEventCreate(&ev);
  if (rc != hipSuccess) { g_last_hip_error = (int)rc; (void)hipGetLastError(); return nullptr; }
  return ev;
}
extern "C" int bds_timer_destroy(void *timer) {
  if (timer && hipEventDestroy(static_cast<hipEvent_t>(timer)) != hipSuccess) { (void)hipGetLastError(); return BDS_EINVAL; }
  return BDS_OK;
}
extern "C" int bds_timer_mark(void *timer, bds_stream_t stream) {
  BDS_REQUIRE(timer);
  hipStream_t st = bds::as_stream(stream);
  hipEvent_t ev = static_cast<hipEvent_t>(timer);
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
  hipError_t rc;
  if (cs != hipStreamCaptureStatusActive) {
    rc = hipEventRecord(ev, st);
  } else {
    // an explicit event-record node behind everything captured so far, which the rest of the capture then depends on
    // (hipEventRecordWithFlags(.., hipEventRecordExternal) does the same in newer runtimes; the HIP runtime PyTorch-ROCm 7.0 ships
    // rejects that flag)
    unsigned long long id = 0;
    hipGraph_t g = nullptr;
    const hipGraphNode_t *deps = nullptr;
    size_t nd = 0;
    hipGraphNode_t node = nullptr;
    rc = hipStreamGetCaptureInfo_v2(st, &cs, &id, &g, &deps, &nd);
    if (rc == hipSuccess) rc = hipGraphAddEventRecordNode(&node, g, deps, nd, ev);
    if (rc == hipSuccess) rc = hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies);
  }
  if (rc != hipSuccess) { g_last_hip_error = (int)rc; (void)hipGetLastError(); return BDS_ELAUNCH; }
  return BDS_OK;
}
// milliseconds between two marks; both must have completed (negative = not available)
extern "C" float bds_timer_elapsed_ms(void *start, void *stop) {
  float ms = -1.f;
  if (!start || !stop) return ms;
  if (hipEventSynchronize(static_cast<hipEvent_t>(stop)) != hipSuccess) { (void)hipGetLastError(); return -1.f; }
  if (hipEventElapsedTime(&ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)) != hipSuccess) { (void)hipGetLastError(); return -1.f; }
  return ms;
}
