// Runtime glue behind the C ABI: thread-local error string, hipGraph capture/replay,
// HIP events on the caller's stream.
#include <stdarg.h>

#include "common.h"

namespace mdm {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int hip_fail(hipError_t e, const char* what) {
    set_error("%s: %s (%d)", what, hipGetErrorString(e), (int)e);
    return -2;
}
}  // namespace mdm
using namespace mdm;

extern "C" const char* mdm_last_error(void) { return g_err; }
extern "C" int mdm_version(void) { return 1; }
extern "C" int mdm_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return hip_fail(e, "hipGetDeviceCount");
    return n;
}

extern "C" int mdm_graph_begin(void* stream) {
    MDM_CHECK_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return 0;
}
extern "C" int mdm_graph_end(void* stream, void** graph_exec_out) {
    MDM_REQUIRE(graph_exec_out, "graph_end: null output");
    hipGraph_t g = nullptr;
    MDM_CHECK_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return hip_fail(e, "hipGraphInstantiate");
    *graph_exec_out = ex;
    return 0;
}
extern "C" int mdm_graph_launch(void* graph_exec, void* stream) {
    MDM_CHECK_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return 0;
}
extern "C" int mdm_graph_destroy(void* graph_exec) {
    if (graph_exec) MDM_CHECK_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return 0;
}

extern "C" int mdm_event_create(void** ev) {
    MDM_REQUIRE(ev, "event_create: null output");
    hipEvent_t e;
    MDM_CHECK_HIP(hipEventCreate(&e));
    *ev = e;
    return 0;
}
extern "C" int mdm_event_record(void* ev, void* stream) {
    MDM_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return 0;
}
extern "C" int mdm_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out) {
    MDM_CHECK_HIP(hipEventSynchronize((hipEvent_t)ev_stop));
    MDM_CHECK_HIP(hipEventElapsedTime(ms_out, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
    return 0;
}
extern "C" int mdm_event_destroy(void* ev) {
    if (ev) MDM_CHECK_HIP(hipEventDestroy((hipEvent_t)ev));
    return 0;
}
extern "C" int mdm_stream_sync(void* stream) {
    MDM_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
