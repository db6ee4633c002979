// TEST INFRASTRUCTURE ONLY -- a tiny single-threaded emulation of the HIP constructs used by
// dfq_amd/csrc/*.hip so that the *unmodified* kernel sources can be compiled with g++ and executed
// on the CPU of the build container (which has no GPU).  Every GPU thread of a workgroup is a
// ucontext fiber; __syncthreads() and the wave shuffles are real rendezvous points, so divergent
// barriers and indexing bugs show up here instead of costing GPU minutes.  Workgroups run one after
// another (no inter-workgroup races can be observed -- those are only testable on the GPU).
//
// Nothing under dfq_amd/ knows about this directory; it is reached only through
// `-I tests/emu/include` in tests/emu/build_emu.py, and the resulting library is loaded only by
// tests.  The product library is always built by hipcc against the real <hip/hip_runtime.h>.
#pragma once

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <cmath>
#include <functional>
#include <vector>

#define DFQ_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct int2 { int x, y; };
struct float4 {
    float x, y, z, w;
};

typedef int hipError_t;
#define hipSuccess 0
#define hipErrorUnknown 999
typedef struct emu_stream* hipStream_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };

namespace emu {

struct ThreadCtx {
    dim3 tid, bid, bdim, gdim;
    unsigned char* smem = nullptr;      // dynamic shared memory of the block (concurrent launches only)
    uint64_t slot;
    uint32_t rl_val = 0;
    const void* rl_where = nullptr;
    bool rl_valid = false;
};
extern ThreadCtx* cur;

void launch(dim3 grid, dim3 block, const std::function<void()>& body);
// every workgroup of the grid alive at once (kernels whose workgroups wait for each other in cycles over time: the
// resident equalisation kernel); `smem_bytes` of dynamic shared memory per workgroup; poll loops yield in s_sleep
void launch_concurrent(dim3 grid, dim3 block, size_t smem_bytes, const std::function<void()>& body);
void poll_yield();
void visibility_point();
void sync_block();
void sync_wave();
uint64_t peer_rl(int src_lane, bool* valid);
uint64_t peer_slot(int lane_xor_mask, bool* valid);
uint64_t peer_slot_abs(int src_lane, bool* valid);

template <typename T>
inline T shfl_xor(T v, int mask) {
    static_assert(sizeof(T) <= 8, "shuffle payload");
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    cur->slot = bits;
    sync_wave();
    bool ok = false;
    uint64_t got = peer_slot(mask, &ok);
    sync_wave();
    if (!ok) return v;
    T out;
    memcpy(&out, &got, sizeof(T));
    return out;
}

}  // namespace emu

#define threadIdx (emu::cur->tid)
#define blockIdx (emu::cur->bid)
#define blockDim (emu::cur->bdim)
#define gridDim (emu::cur->gdim)

inline void __syncthreads() { emu::sync_block(); }
namespace emu { uint64_t peer_slot_lane(int src_lane, bool* valid); }
template <typename T>
inline T __shfl(T v, int src_lane) {
    uint64_t bits = 0;
    memcpy(&bits, &v, sizeof(T));
    emu::cur->slot = bits;
    emu::sync_wave();
    bool ok = false;
    const uint64_t got = emu::peer_slot_lane(src_lane, &ok);
    emu::sync_wave();
    if (!ok) return v;
    T out;
    memcpy(&out, &got, sizeof(T));
    return out;
}
template <typename T>
inline T __shfl_xor(T v, int mask) { return emu::shfl_xor(v, mask); }

// v_readlane broadcast.  Kernels read many lanes of ONE register in a row; the register is
// identified by the address of the variable passed in (the macro below), so the wave rendezvous
// happens once per register instead of once per read.  All lanes execute the same sequence of call
// sites, hence they agree on when a new register starts.  Limitation (checked): the variable must not
// change between two reads of the same run.
namespace emu {
inline int readlane_ref(const void* where, int v0, int src_lane) {
    const uint32_t v = (uint32_t)v0;
    if (!(cur->rl_valid && cur->rl_where == where)) {
        if (cur->rl_valid) sync_wave();        // every lane has finished reading the previous register
        cur->rl_val = v;
        cur->rl_where = where;
        cur->rl_valid = true;
        sync_wave();
    } else if (cur->rl_val != v) {
        fprintf(stderr, "emu: readlane source changed between reads of one run\n");
        abort();
    }
    bool ok = false;
    const uint64_t got = peer_rl(src_lane, &ok);
    return ok ? (int)(uint32_t)got : v0;
}
}  // namespace emu
#define __builtin_amdgcn_readlane(v, lane) emu::readlane_ref(&(v), (int)(v), (lane))
// only ever applied to wave-uniform values in these kernels: the lane's own value is the first lane's
#define __builtin_amdgcn_readfirstlane(v) (v)

inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
inline unsigned atomicMin(unsigned* p, unsigned v) { unsigned o = *p; if (v < o) *p = v; return o; }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
inline double atomicAdd(double* p, double v) { double o = *p; *p = o + v; return o; }

#define __HIP_MEMORY_SCOPE_AGENT 4
// A device-scope load is where another workgroup's progress becomes visible.  In a launch whose workgroups run concurrently
// the emulation lets OTHER workgroups run at a pseudo-random quarter of these loads, so that the threads of one workgroup
// can see different values of a word that is changing -- as waves do on the hardware.  (A workgroup that lets every thread
// decide "is the counter there yet?" for itself passed every emulated test and mismatched its barriers on the GPU.)
#define __hip_atomic_load(ptr, order, scope) (emu::visibility_point(), *(ptr))
#define __hip_atomic_store(ptr, val, order, scope) (*(ptr) = (val))
#define __hip_atomic_fetch_add(ptr, val, order, scope) __atomic_fetch_add((ptr), (val), __ATOMIC_SEQ_CST)
inline void __builtin_amdgcn_s_sleep(int) { emu::poll_yield(); }
inline void __builtin_amdgcn_s_waitcnt(int) {}
inline long long clock64() { return 0; }
inline long long wall_clock64() { return 0; }
inline unsigned __builtin_amdgcn_s_getreg(int) { return 0u; }
inline void __builtin_amdgcn_s_setprio(int) {}
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }

using std::max;
using std::min;

// ---- runtime API subset -------------------------------------------------------------------------
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "emulated error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) {
    *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256);
    if (*p) memset(*p, 0xA5, n);   // poison: kernels must not rely on zeroed allocations
    return *p ? hipSuccess : hipErrorUnknown;
}
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
namespace emu { extern std::vector<std::function<void()>>* capture; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
    if (emu::capture) { emu::capture->push_back([=]() { memset(d, v, n); }); return hipSuccess; }
    memset(d, v, n);
    return hipSuccess;
}

typedef struct emu_event* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
#define hipEventDisableTiming 2
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)1; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }

namespace emu {
extern const void* kernarg_ptr;
extern std::vector<std::function<void()>>* capture;     // non-null while a stream capture is open
template <typename K, typename A0, typename... As>
inline void run_k(dim3 grid, dim3 block, K kernel, A0& a0, As&... as) {
    kernarg_ptr = &a0;                                   // the first argument sits at kernarg offset 0
    launch(grid, block, [&]() { kernel(a0, as...); });
    kernarg_ptr = nullptr;
}
template <typename K, typename A0, typename... As>
inline void launch_k(dim3 grid, dim3 block, K kernel, A0 a0, As... as) {
    if (capture) {
        capture->push_back([=]() mutable { run_k(grid, block, kernel, a0, as...); });
        return;
    }
    run_k(grid, block, kernel, a0, as...);
}
}  // namespace emu
inline const void* __builtin_amdgcn_kernarg_segment_ptr() { return emu::kernarg_ptr; }

// ---- stream capture / graphs: a graph is the recorded list of launches ----
typedef std::vector<std::function<void()>>* hipGraph_t;
typedef std::vector<std::function<void()>>* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
#define hipStreamNonBlocking 1
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)8; return hipSuccess; }   // (launches run at once, whatever the stream)
inline hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) {
    emu::capture = new std::vector<std::function<void()>>();
    return hipSuccess;
}
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = emu::capture; emu::capture = nullptr; return hipSuccess; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t g, void*, void*, size_t) {
    *e = new std::vector<std::function<void()>>(*g);
    return hipSuccess;
}
inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) { for (auto& f : *e) f(); return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch_k((grid), (block), kernel, __VA_ARGS__)

namespace emu {
template <typename K, typename A0, typename... As>
inline void launch_concurrent_k(dim3 grid, dim3 block, size_t smem, K kernel, A0 a0, As... as) {
    kernarg_ptr = &a0;
    launch_concurrent(grid, block, smem, [&]() { kernel(a0, as...); });
    kernarg_ptr = nullptr;
}
}  // namespace emu
// residency queries: the emulation keeps any grid alive
inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 8; return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorUnknown; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 256; return hipSuccess; }
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
