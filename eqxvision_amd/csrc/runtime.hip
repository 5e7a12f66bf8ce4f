// Runtime plumbing behind the C ABI: thread-local error text, flags, the per-device zero page,
// hipGraph capture and HIP-event timing.  No compute here.
#include <stdarg.h>

#include <map>
#include <mutex>
#include <string>

#include "common.h"

namespace mv {

static thread_local char g_err[512] = "";
static thread_local char g_kernel[128] = "";
static std::mutex g_mu;
// A/B and test switches.  Thread-local like the error text: a thread that flips a switch cannot change what another
// thread's launches dispatch to, and the hot path takes no lock to read them.
static thread_local std::map<std::string, int> g_flags;
static thread_local int g_flags_epoch = 0;
static std::map<int, void*> g_zero;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
void set_kernel_name(const char* name) {
    strncpy(g_kernel, name, sizeof(g_kernel) - 1);
    g_kernel[sizeof(g_kernel) - 1] = 0;
}
void append_kernel_name(const char* suffix) {
    strncat(g_kernel, suffix, sizeof(g_kernel) - strlen(g_kernel) - 1);
}
// One launch's scratch memory, handed over by the host thread that issues the launch (mv_set_scratch): the library never allocates.
static thread_local struct { void* ptr; size_t bytes; hipStream_t stream; } g_scratch = {nullptr, 0, nullptr};
void* take_scratch(hipStream_t stream, size_t need) {
    if (!g_scratch.ptr || g_scratch.stream != stream || g_scratch.bytes < need) return nullptr;
    void* p = g_scratch.ptr;
    g_scratch.ptr = nullptr;
    return p;
}
void* peek_scratch(hipStream_t stream, size_t* bytes) {       // what is on offer for this stream, without taking it
    if (!g_scratch.ptr || g_scratch.stream != stream) return nullptr;
    if (bytes) *bytes = g_scratch.bytes;
    return g_scratch.ptr;
}
int get_flag(const char* name) {
    if (g_flags.empty()) return 0;
    auto it = g_flags.find(name);
    return it == g_flags.end() ? 0 : it->second;
}

// 4 KiB of zeros per device: the source every out-of-image / out-of-range LDS-DMA lane reads.
const void* zero_page(hipStream_t) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_zero.find(dev);
    if (it != g_zero.end()) return it->second;
    void* p = nullptr;
    if (hipMalloc(&p, 4096) != hipSuccess) return nullptr;
    if (hipMemset(p, 0, 4096) != hipSuccess) return nullptr;  // synchronous, outside any capture
    g_zero[dev] = p;
    return p;
}

}  // namespace mv

extern "C" {

int mv_abi_version(void) { return MV_ABI_VERSION; }
const char* mv_last_error(void) { return mv::g_err; }
const char* mv_last_kernel(void) { return mv::g_kernel; }

int mv_set_flag(const char* name, int value) {
    if (!name) return MV_E_INVALID;
    if (value == 0) mv::g_flags.erase(name);
    else mv::g_flags[name] = value;
    // "epoch" = a hash of the current switch settings (std::map iterates in key order): setting a switch and setting it back
    // gives the first value again, so a recorded launch list is re-used exactly when the switches it was recorded under are back
    unsigned h = 2166136261u;
    for (const auto& kv : mv::g_flags) {
        for (char c : kv.first) h = (h ^ (unsigned char)c) * 16777619u;
        h = (h ^ (unsigned)kv.second) * 16777619u;
        h = (h ^ 0xffu) * 16777619u;
    }
    mv::g_flags_epoch = mv::g_flags.empty() ? 0 : (int)(h & 0x7fffffffu);
    return MV_OK;
}
int mv_flags_epoch(void) { return mv::g_flags_epoch; }
int mv_device_status(int clear, unsigned* status) {
    MV_CHECK_ARG(status, "device_status: NULL pointer");
    return mv::device_status(clear, status);
}
int mv_get_flag(const char* name) { return name ? mv::get_flag(name) : 0; }

int mv_set_scratch(void* ptr, int64_t bytes, mv_stream_t stream) {
    mv::g_scratch.ptr = bytes > 0 ? ptr : nullptr;
    mv::g_scratch.bytes = bytes > 0 ? (size_t)bytes : 0;
    mv::g_scratch.stream = (hipStream_t)stream;
    return MV_OK;
}
int64_t mv_splitk_scratch_bytes(int64_t M, int64_t N, int64_t K_reduction) { return (int64_t)mv::splitk_scratch_bytes(M, N, K_reduction); }

int mv_graph_begin_capture(mv_stream_t stream) {
    if (!mv::zero_page((hipStream_t)stream)) {  // must exist before capture (hipMalloc is illegal inside)
        mv::set_error("zero page allocation failed");
        return MV_E_OOM;
    }
    MV_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return MV_OK;
}
int mv_graph_end_capture(mv_stream_t stream, void** graph_exec) {
    MV_CHECK_ARG(graph_exec, "graph_exec is NULL");
    hipGraph_t g = nullptr;
    MV_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
    hipGraphExec_t ex = nullptr;
    hipError_t e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        mv::set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
        return (int)e;
    }
    *graph_exec = (void*)ex;
    return MV_OK;
}
int mv_graph_launch(void* graph_exec, mv_stream_t stream) {
    MV_CHECK_ARG(graph_exec, "graph_exec is NULL");
    MV_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
    return MV_OK;
}
int mv_graph_destroy(void* graph_exec) {
    if (graph_exec) MV_HIP(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
    return MV_OK;
}

int mv_event_create(void** ev) {
    MV_CHECK_ARG(ev, "ev is NULL");
    hipEvent_t e;
    MV_HIP(hipEventCreate(&e));
    *ev = (void*)e;
    return MV_OK;
}
int mv_event_record(void* ev, mv_stream_t stream) {
    MV_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return MV_OK;
}
int mv_event_elapsed_ms(void* start, void* stop, float* ms) {
    MV_CHECK_ARG(ms, "ms is NULL");
    MV_HIP(hipEventSynchronize((hipEvent_t)stop));
    MV_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return MV_OK;
}
int mv_event_destroy(void* ev) {
    if (ev) MV_HIP(hipEventDestroy((hipEvent_t)ev));
    return MV_OK;
}

}  // extern "C"
