// dm3d_api.hip — version / error text / device probe / HIP-graph capture helpers of the C ABI (include/dm3d.h).
#include "dm3d_common.h"
#include <string.h>

char* dm3d_err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

extern "C" int dm3d_version(void) { return DM3D_VERSION; }

extern "C" const char* dm3d_last_error(void) { return dm3d_err_buf(); }

extern "C" int dm3d_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 0; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// One denoising step (U-Net forward + posterior update + counter decrement) is ~250 launches with fixed pointers; the
// caller captures it once on its stream and replays it T times (the loop index lives in device memory).
extern "C" int dm3d_graph_begin(void* stream) {
    DM3D_HIP(hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeThreadLocal));
    return DM3D_OK;
}

extern "C" int dm3d_graph_end(void* stream, void** graph_exec_out) {
    DM3D_REQUIRE(graph_exec_out != nullptr, "graph_end: null output");
    hipGraph_t graph = nullptr;
    DM3D_HIP(hipStreamEndCapture(static_cast<hipStream_t>(stream), &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return dm3d_fail(DM3D_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    *graph_exec_out = exec;
    return DM3D_OK;
}

extern "C" int dm3d_graph_launch(void* graph_exec, void* stream) {
    DM3D_REQUIRE(graph_exec != nullptr, "graph_launch: null graph");
    DM3D_HIP(hipGraphLaunch(static_cast<hipGraphExec_t>(graph_exec), static_cast<hipStream_t>(stream)));
    return DM3D_OK;
}

extern "C" int dm3d_graph_destroy(void* graph_exec) {
    if (graph_exec) DM3D_HIP(hipGraphExecDestroy(static_cast<hipGraphExec_t>(graph_exec)));
    return DM3D_OK;
}
