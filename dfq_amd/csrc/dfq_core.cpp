// Error plumbing and version entry points of libdfq_hip.
#include <stdarg.h>

#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <vector>

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

// ---- device memory of plans (see dfq_common.hpp) ----
namespace {
struct DevPool {
    std::mutex m;
    std::unordered_map<unsigned long long, std::vector<void*>> free_lists;   // key: device << 48 | rounded size / 256
    std::unordered_map<void*, unsigned long long> live;                      // block -> key, for blocks that may return to a list
    size_t cached = 0, cap = 512u << 20;
    unsigned long long devices = 0;                                           // bit d: a block of device d was ever handed out
    DevPool() {
        const char* e = getenv("DFQ_POOL_MB");
        if (e && *e) cap = (size_t)std::max(0, atoi(e)) << 20;
    }
};
DevPool& dev_pool() { static DevPool* p = new DevPool(); return *p; }         // never destroyed: blocks may outlive static teardown
constexpr size_t kPoolMaxBlock = 64u << 20;
}  // namespace

hipError_t dev_malloc(void** out, size_t bytes) {
    const size_t rounded = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    DevPool& P = dev_pool();
    if (P.cap == 0 || rounded > kPoolMaxBlock) return hipMalloc(out, rounded);
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long key = ((unsigned long long)dev << 48) | (rounded >> 8);
    {
        std::lock_guard<std::mutex> g(P.m);
        if (dev >= 0 && dev < 64) P.devices |= 1ull << dev;
        auto it = P.free_lists.find(key);
        if (it != P.free_lists.end() && !it->second.empty()) {
            *out = it->second.back();
            it->second.pop_back();
            P.cached -= rounded;
            P.live[*out] = key;
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, rounded);
    if (e != hipSuccess) {                                // out of memory with blocks parked in the lists: give them back, try again
        std::vector<void*> all;
        {
            std::lock_guard<std::mutex> g(P.m);
            for (auto& kv : P.free_lists) { all.insert(all.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
            P.cached = 0;
        }
        if (all.empty()) return e;
        (void)hipGetLastError();
        for (void* b : all) (void)hipFree(b);
        e = hipMalloc(out, rounded);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> g(P.m);
    P.live[*out] = key;
    return hipSuccess;
}

void dev_free(void* p) {
    if (!p) return;
    DevPool& P = dev_pool();
    {
        std::lock_guard<std::mutex> g(P.m);
        auto it = P.live.find(p);
        if (it != P.live.end()) {
            const unsigned long long key = it->second;
            const size_t rounded = (size_t)(key & ((1ull << 48) - 1)) << 8;
            P.live.erase(it);
            if (P.cached + rounded <= P.cap) {
                P.free_lists[key].push_back(p);
                P.cached += rounded;
                return;
            }
        }
    }
    (void)hipFree(p);
}

// A destroyed plan's blocks go back to the free list of THEIR device, and the next plan of that device may take them at once:
// nothing of the destroyed plan may still be in flight there.  The caller's current device need not be the plan's (several
// GPUs driven from one process), so every device this library ever allocated on is synchronised -- one, in the usual
// one-process-per-GPU arrangement: the same single hipDeviceSynchronize the first hipFree of a plan used to be.
void dev_quiesce() {
    unsigned long long devices;
    {
        DevPool& P = dev_pool();
        std::lock_guard<std::mutex> g(P.m);
        devices = P.devices;
    }
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return; }
    (void)hipDeviceSynchronize();
    bool moved = false;
    for (int d = 0; d < 64; ++d) {
        if (d == cur || !((devices >> d) & 1)) continue;
        if (hipSetDevice(d) == hipSuccess) { (void)hipDeviceSynchronize(); moved = true; } else (void)hipGetLastError();
    }
    if (moved) (void)hipSetDevice(cur);
}

size_t dev_pool_trim() {
    DevPool& P = dev_pool();
    std::vector<void*> all;
    size_t bytes;
    {
        std::lock_guard<std::mutex> g(P.m);
        for (auto& kv : P.free_lists) { all.insert(all.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
        bytes = P.cached;
        P.cached = 0;
    }
    for (void* b : all) (void)hipFree(b);       // hipFree takes any device's pointer
    return bytes;
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

long long dfq_pool_trim(void) { return (long long)dfq::dev_pool_trim(); }

int dfq_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return dfq::fail_hip(e, "hipGetDeviceCount", __FILE__, __LINE__);
    return n;
}

}  // extern "C"
