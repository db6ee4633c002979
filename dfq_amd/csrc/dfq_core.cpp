// Error plumbing and version entry points of libdfq_hip.
#include <stdarg.h>

#include <mutex>

#include "dfq_common.hpp"

namespace dfq {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail_arg(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return DFQ_ERR_ARG;
}

int fail_hip(hipError_t e, const char* what, const char* file, int line) {
    char buf[640];
    snprintf(buf, sizeof(buf), "HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what,
             file, line);
    g_last_error = buf;
    return DFQ_ERR_HIP;
}

// ---- SpinGuard (see dfq_common.hpp) ----
namespace {
std::mutex g_spin_mu;
hipEvent_t g_spin_event = nullptr;
hipStream_t g_spin_stream = nullptr;
bool g_spin_pending = false;
}  // namespace

// The mutex is held from the constructor to the destructor: a second host thread that enqueued its batch between
// this thread's wait and this thread's record would wait for the batch BEFORE this one and overlap with this one.
SpinGuard::SpinGuard(hipStream_t stream) : stream_(stream) {
    g_spin_mu.lock();
    if (g_spin_pending && g_spin_stream != stream_ && g_spin_event)
        (void)hipStreamWaitEvent(stream_, g_spin_event, 0);
}

SpinGuard::~SpinGuard() {
    if (!g_spin_event && hipEventCreateWithFlags(&g_spin_event, hipEventDisableTiming) != hipSuccess) g_spin_event = nullptr;
    if (g_spin_event && hipEventRecord(g_spin_event, stream_) == hipSuccess) {
        g_spin_stream = stream_;
        g_spin_pending = true;
    }
    g_spin_mu.unlock();
}

}  // namespace dfq

extern "C" {

int dfq_version(void) { return DFQ_HIP_VERSION; }

const char* dfq_last_error(void) { return dfq::g_last_error.c_str(); }

int dfq_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return dfq::fail_hip(e, "hipGetDeviceCount", __FILE__, __LINE__);
    return n;
}

}  // extern "C"
