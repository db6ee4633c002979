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
// One {event, stream, pending} record per DEVICE ordinal (an event belongs to the device it was created on: recording an event
// of device A on a stream of device B fails), one process-wide mutex for all of them.
namespace {
constexpr int kMaxDevices = 64;
struct SpinState {
    hipEvent_t event = nullptr;
    hipStream_t stream = nullptr;
    bool pending = false;
};
std::mutex g_spin_mu;
SpinState g_spin[kMaxDevices];
}  // namespace

// The mutex is held from the constructor to the destructor: a second host thread that enqueued its batch between
// this thread's wait and this thread's record would wait for the batch BEFORE this one and overlap with this one.
SpinGuard::SpinGuard(hipStream_t stream) : stream_(stream), device_(0) {
    g_spin_mu.lock();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) { (void)hipGetLastError(); dev = 0; }
    device_ = dev;
    SpinState& st = g_spin[device_];
    if (st.pending && st.stream != stream_ && st.event) {
        // a failed wait must not linger as the thread's sticky error (the next launch check would report it as a launch
        // failure): clear it and fall back to the blunt tool
        if (hipStreamWaitEvent(stream_, st.event, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(st.stream); }
    }
}

SpinGuard::~SpinGuard() {
    SpinState& st = g_spin[device_];
    if (!st.event && hipEventCreateWithFlags(&st.event, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); st.event = nullptr; }
    if (st.event && hipEventRecord(st.event, stream_) == hipSuccess) {
        st.stream = stream_;
        st.pending = true;
    } else {
        // no event to wait for: make the batch itself finish before anybody else may enqueue one
        (void)hipGetLastError();
        (void)hipStreamSynchronize(stream_);
        st.pending = false;
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
