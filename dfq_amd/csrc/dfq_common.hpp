// Shared device/host helpers of libdfq_hip (gfx950, wave64).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <string>
#include <tuple>
#include <vector>
#include <algorithm>

#include "../../include/dfq_hip.h"

namespace dfq {

// Pointers fetched from descriptor tables in memory have no address space the compiler can see, so
// plain dereferences become flat_load/flat_store.  Casting to the global address space gives
// global_load/global_store (no LDS-aperture check, no lgkmcnt coupling with LDS traffic).  The CPU
// test emulation builds with -DDFQ_GLOBAL_AS= (an empty qualifier).
#ifndef DFQ_GLOBAL_AS
#define DFQ_GLOBAL_AS __attribute__((address_space(1)))
#endif
#ifndef DFQ_CONSTANT_AS
#define DFQ_CONSTANT_AS __attribute__((address_space(4)))     // kernarg segment / constant memory
#endif
typedef DFQ_GLOBAL_AS float gfloat;
typedef DFQ_GLOBAL_AS unsigned int guint;
typedef float fvec4 __attribute__((vector_size(16)));   // native 16-byte vector (dwordx4 loads/stores)
typedef DFQ_GLOBAL_AS fvec4 gfvec4;

// dynamic shared memory of a kernel and the launch of a kernel whose workgroups must all be alive at once (the CPU test
// emulation gives every workgroup its own buffer and runs the grid concurrently, tests/emu)
#ifdef DFQ_EMU
#define DFQ_DYN_SMEM(name) unsigned char* name = emu::cur->smem
#define DFQ_LAUNCH_RESIDENT(kernel, grid, block, smem, stream, ...) emu::launch_concurrent_k((grid), (block), (smem), kernel, __VA_ARGS__)
#define DFQ_LAUNCH_RESIDENT_PLAIN(kernel, grid, block, smem, stream, ...) emu::launch_concurrent_k((grid), (block), (smem), kernel, __VA_ARGS__)
#define DFQ_LAUNCH_SPINNING(kernel, grid, block, smem, stream, ...) emu::launch_concurrent_k((grid), (block), (smem), kernel, __VA_ARGS__)
#else
#define DFQ_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define DFQ_LAUNCH_RESIDENT(kernel, grid, block, smem, stream, ...) ::dfq::launch_resident(kernel, grid, block, smem, stream, __VA_ARGS__)
#define DFQ_LAUNCH_RESIDENT_PLAIN(kernel, grid, block, smem, stream, ...) ::dfq::launch_resident_opt(false, kernel, grid, block, smem, stream, __VA_ARGS__)
// an ORDINARY launch of a grid whose workgroups wait for each other: the caller sizes it to be co-resident (well below the
// kernel's occupancy limit) and holds a SpinGuard; cheaper than the cooperative launch (~40 us), for small kernels launched often
#define DFQ_LAUNCH_SPINNING(kernel, grid, block, smem, stream, ...) hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__)
#endif

// Streaming accesses: the non-temporal hint ("nt" on the global load / store) keeps a line that is touched once per launch
// from displacing others on its way through the caches.  Measured on the sweep kernel (batch of 32, 0.99 GB per launch):
// hint on the stores -3 %, on loads and stores -4 % of the launch time (tools/ab_variants.sh).
#if defined(__HIP_DEVICE_COMPILE__)
#define DFQ_NT_LOAD(ptr) __builtin_nontemporal_load(ptr)
#define DFQ_NT_STORE(val, ptr) __builtin_nontemporal_store((val), (ptr))
#else
#define DFQ_NT_LOAD(ptr) (*(ptr))
#define DFQ_NT_STORE(val, ptr) (*(ptr) = (val))
#endif

// tuning switch for the read-only reduction kernels (per-tensor / per-sample / per-layer min-max): hint on their loads.
// Off: no gain for the reduction itself (55 us either way on 308 MB), and the quantise pass that reads the same tensor next
// then misses the Infinity Cache (98 -> 105 us).
#ifndef DFQ_READ_NT
#define DFQ_READ_NT 0
#endif
constexpr int kReadNt = DFQ_READ_NT;

constexpr int kBlock = 256;   // 4 wavefronts of 64 lanes
constexpr int kWave = 64;

// ---- error plumbing -----------------------------------------------------------------------
void set_error(const std::string& msg);
int fail_arg(const char* fmt, ...);
int fail_hip(hipError_t e, const char* what, const char* file, int line);

#define DFQ_HIP_TRY(expr)                                                       \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) return ::dfq::fail_hip(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define DFQ_CHECK_LAUNCH()                                                      \
    do {                                                                        \
        hipError_t _e = hipGetLastError();                                      \
        if (_e != hipSuccess) return ::dfq::fail_hip(_e, "kernel launch", __FILE__, __LINE__); \
    } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---- device memory of plans (dfq_core.cpp) ---------------------------------------------------
// A plan owns some thirty small device tables; a service that calibrates batch after batch creates and destroys plans of the
// same shapes over and over, and hipMalloc / hipFree (30 us each, hipFree synchronising the device every time) were a third
// of what creating a plan costs.  Blocks released by a plan go to a per-device free list keyed on their (256-byte rounded)
// size and are handed out again as they are -- nothing is assumed about their contents, exactly like hipMalloc.  A plan's
// destroy synchronises the device ONCE (dev_quiesce) before it releases its blocks, which is what the first hipFree used to do.
// DFQ_POOL_MB caps what the free lists hold (default 512; 0 = every release is a hipFree); blocks above 64 MB bypass the lists.
// DFQ_PLAN_TIMING=1: where creating a plan spends its host time (stderr, one line per plan)
struct PlanTimer {
    const char* name;
    bool on;
    double t0, last;
    std::string line;
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    explicit PlanTimer(const char* n) : name(n) { const char* e = getenv("DFQ_PLAN_TIMING"); on = e && *e == '1'; t0 = last = on ? now() : 0; }
    void tick(const char* label) {
        if (!on) return;
        const double t = now();
        char buf[96];
        snprintf(buf, sizeof(buf), " %s %.3f", label, t - last);
        line += buf;
        last = t;
    }
    ~PlanTimer() { if (on) fprintf(stderr, "[dfq] %s: total %.3f ms:%s\n", name, now() - t0, line.c_str()); }
};

hipError_t dev_malloc(void** out, size_t bytes);
void dev_free(void* p);
void dev_quiesce();
size_t dev_pool_trim();
// The tables of ONE plan, carved out of a few 4 MB blocks (a table larger than that gets a block of its own): two or three
// dev_malloc calls per plan instead of thirty, also when no block of a destroyed plan is waiting in the free lists (a service
// that keeps several batches in flight).  Tables start on 256-byte boundaries; release() returns every block.
struct DevSlab {
    static constexpr size_t kSlab = 4u << 20;
    std::vector<void*> blocks;
    char* cur = nullptr;
    size_t left = 0;
    hipError_t alloc(void** out, size_t bytes) {
        const size_t need = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
        if (need > left) {
            void* b = nullptr;
            const hipError_t e = dev_malloc(&b, std::max(need, kSlab));
            if (e != hipSuccess) return e;
            blocks.push_back(b);
            if (need >= kSlab) { *out = b; return hipSuccess; }          // its own block; the current one keeps its rest
            cur = (char*)b;
            left = kSlab;
        }
        *out = cur;
        cur += need;
        left -= need;
        return hipSuccess;
    }
    void release() {
        for (void* b : blocks) dev_free(b);
        blocks.clear();
        cur = nullptr;
        left = 0;
    }
};

#ifndef DFQ_EMU
// Launch of a kernel whose workgroups wait for each other in cycles (all of them must be resident at once): a COOPERATIVE
// launch -- the runtime itself then guarantees co-residency or refuses the launch (hipErrorCooperativeLaunchTooLarge), instead
// of the plan's own occupancy arithmetic being the only safeguard.  If the runtime refuses (or does not support cooperative
// launches), the ordinary launch the plan sized for residency is used, as before round 3; DFQ_COOPERATIVE=0 forces that.
// `coop_default`: what happens without DFQ_COOPERATIVE in the environment (1 / 0 there force either).  The persistent
// equalisation launch (dfq_le_resident.hip) passes false since round 5: a cooperative launch costs 20-40 us of a 0.65 ms pass
// (measured: tools/gpu_r05_ab.sh), its plan admits only three quarters of the occupancy limit anyway, and that launch stores ALL
// OR NOTHING -- an abandoned wait leaves the network untouched and the pass is repeated on per-level launches (dfq_le_run).
template <typename... Params, typename... Args>
inline void launch_resident_opt(bool coop_default, void (*kernel)(Params...), dim3 grid, dim3 block, size_t smem, hipStream_t stream, Args... args) {
    static const char* env = getenv("DFQ_COOPERATIVE");
    const bool coop = (env && (env[0] == '0' || env[0] == '1')) ? env[0] == '1' : coop_default;
    if (coop) {
        std::tuple<Params...> vals{args...};
        void* ptrs[sizeof...(Params)];
        int i = 0;
        std::apply([&](auto&... v) { ((ptrs[i++] = (void*)&v), ...); }, vals);
        const hipError_t e = hipLaunchCooperativeKernel((const void*)kernel, grid, block, ptrs, (unsigned)smem, stream);
        if (e == hipSuccess) return;
        (void)hipGetLastError();                       // refused: fall through to the ordinary launch
    }
    hipLaunchKernelGGL(kernel, grid, block, smem, stream, args...);
}
template <typename... Params, typename... Args>
inline void launch_resident(void (*kernel)(Params...), dim3 grid, dim3 block, size_t smem, hipStream_t stream, Args... args) {
    launch_resident_opt(true, kernel, grid, block, smem, stream, args...);
}
#endif

// ---- several small buffers zeroed by ONE launch -------------------------------------------------------------------
// (hipMemsetAsync is a launch of its own per buffer -- two when the size is not a multiple of 16 bytes -- with 4-12 us
// between consecutive ones: the three clears in front of a single-network equalisation cost ~30 us of its ~100.)
struct ClearArgs {
    uint32_t* p[4];
    long long words[4];
};
static __global__ void clear_kernel(ClearArgs a) {
    const long long step = (long long)gridDim.x * blockDim.x;
    for (int k = 0; k < 4; ++k)
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.words[k]; i += step) a.p[k][i] = 0u;
}
// up to four (pointer, bytes) pairs; sizes are multiples of 4
inline void clear_buffers(hipStream_t st, void* p0, size_t b0, void* p1 = nullptr, size_t b1 = 0, void* p2 = nullptr, size_t b2 = 0,
                          void* p3 = nullptr, size_t b3 = 0) {
    ClearArgs a;
    void* ps[4] = {p0, p1, p2, p3};
    const size_t bs[4] = {b0, b1, b2, b3};
    long long most = 0;
    for (int k = 0; k < 4; ++k) {
        a.p[k] = (uint32_t*)ps[k];
        a.words[k] = ps[k] ? (long long)(bs[k] / 4) : 0;
        if (a.words[k] > most) most = a.words[k];
    }
    if (most == 0) return;
    const int grid = (int)((most + 1023) / 1024 < 512 ? (most + 1023) / 1024 : 512);
    hipLaunchKernelGGL(clear_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, st, a);
}

// ---- kernels with in-launch waits never overlap across streams -------------------------------------
// The one-launch sweep (le_level_kernel), the resident equalisation kernel and the one-launch bias-correction
// chain contain workgroups that wait for other workgroups of the SAME launch.  That is deadlock-free as long as
// the waited-for workgroups are resident or will become resident -- true for a kernel that has the chip to itself
// or shares it with kernels that finish on their own, NOT for two such kernels from different streams, each
// holding the slots the other's producers need.  So the library serialises them: SpinGuard's
// constructor makes `stream` wait for the last waiting-kernel batch enqueued on a DIFFERENT stream, its destructor
// records the end of this batch; host threads enqueue such batches one at a time (a process-wide mutex held between the two).  One event record per enqueue call, nothing per launch; kernels without in-launch
// waits (per-level launches, bootstrap, quantisers, ...) are not affected.  One record per device ordinal, one mutex per process.
class SpinGuard {
public:
    explicit SpinGuard(hipStream_t stream);
    ~SpinGuard();
    SpinGuard(const SpinGuard&) = delete;
    SpinGuard& operator=(const SpinGuard&) = delete;
private:
    hipStream_t stream_;
    int device_;
};

// ---- order-preserving float <-> uint32 encoding for atomic min/max ----------------------------
// enc is monotone in the float order (-inf < ... < -0 < +0 < ... < +inf).  A "max slot" holds
// atomicMax(enc(x)), a "min slot" holds atomicMax(~enc(x)); both have identity 0, so one memset
// initialises any number of slots.
// Both directions are written as one XOR with a mask built from the arithmetic-shifted sign -- no
// select and no `& 0x7fffffff`: this compiler pattern-matches those into a v_cndmask with an abs
// source modifier, and that fold produced wrong code in one of the LE tile kernels (the result
// changed with unrelated edits to the function; see DESIGN.md "toolchain notes").
__device__ __forceinline__ uint32_t enc_ord(float f) {
    const uint32_t u = __float_as_uint(f);
    const uint32_t sign = (uint32_t)((int32_t)u >> 31);          // all ones for negative floats
    return u ^ (sign | 0x80000000u);                             // negative: ~u, else: set the top bit
}
__device__ __forceinline__ float dec_ord(uint32_t e) {
    const uint32_t top = (uint32_t)((int32_t)e >> 31);           // all ones if the top bit is set
    return __uint_as_float(e ^ (~top | 0x80000000u));            // top set: clear it, else: ~e
}
__device__ __forceinline__ float slot_max(uint32_t slot) { return dec_ord(slot); }
__device__ __forceinline__ float slot_min(uint32_t slot) { return dec_ord(~slot); }

// ---- min / max without the canonicalisation prologue ----------------------------------------------
// fminf / fmaxf make the compiler quiet both operands first (v_max_f32 x, x, x) because an operand might be
// a signalling NaN: three instructions per min or max.  In the hot loops (eight per register slot, plus the
// shuffle butterflies) that was a quarter of all vector instructions.  v_min_f32 / v_max_f32 themselves
// already return the non-NaN operand, which is all these reductions rely on.
__device__ __forceinline__ float vmin_raw(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return fminf(a, b);
#endif
}
__device__ __forceinline__ float vmax_raw(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
#else
    return fmaxf(a, b);
#endif
}

// ---- butterfly steps that stay out of the LDS ------------------------------------------------------
// __shfl_xor is a ds_bpermute_b32: every step goes through the LDS crossbar, which all waves of a CU share -- a tile
// that reduces 16 register slots over 64 lanes issues ~200 of them, and eight waves doing so take microseconds.
// xor_lane<M>(x) returns lane (id ^ M)'s x with register-file moves only: DPP quad permutes (M = 1, 2), a pair of DPP
// row shifts with complementary bank masks (M = 4, 8), gfx950's v_permlane16_swap / v_permlane32_swap (M = 16, 32: after
// the swap of two copies of x, one of them holds the own value and the other the partner's; for the commutative min / max
// it does not matter which).  tools/litmus/lane_xor.hip checks all six against __shfl_xor on the hardware.
#if defined(__HIP_DEVICE_COMPILE__)
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ uint32_t dpp_move(uint32_t old, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xf, BANK_MASK, false);
}
#endif
template <int M>
__device__ __forceinline__ void xor_lane_minmax(float& mn, float& mx) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t a = __float_as_uint(mn), b = __float_as_uint(mx);
    uint32_t pa, pb;
    if (M == 1) { pa = dpp_move<0xB1, 0xf>(a, a); pb = dpp_move<0xB1, 0xf>(b, b); }              // quad_perm [1,0,3,2]
    else if (M == 2) { pa = dpp_move<0x4E, 0xf>(a, a); pb = dpp_move<0x4E, 0xf>(b, b); }         // quad_perm [2,3,0,1]
    else if (M == 4) {      // lanes 0-3 / 8-11 of a row take lane + 4 (row_shl:4), lanes 4-7 / 12-15 take lane - 4 (row_shr:4)
        pa = dpp_move<0x114, 0xA>(dpp_move<0x104, 0x5>(a, a), a);
        pb = dpp_move<0x114, 0xA>(dpp_move<0x104, 0x5>(b, b), b);
    } else if (M == 8) {
        pa = dpp_move<0x118, 0xC>(dpp_move<0x108, 0x3>(a, a), a);
        pb = dpp_move<0x118, 0xC>(dpp_move<0x108, 0x3>(b, b), b);
    } else if (M == 16) {
        const auto ra = __builtin_amdgcn_permlane16_swap(a, a, false, false);
        const auto rb = __builtin_amdgcn_permlane16_swap(b, b, false, false);
        mn = vmin_raw(__uint_as_float(ra[0]), __uint_as_float(ra[1]));
        mx = vmax_raw(__uint_as_float(rb[0]), __uint_as_float(rb[1]));
        return;
    } else {
        const auto ra = __builtin_amdgcn_permlane32_swap(a, a, false, false);
        const auto rb = __builtin_amdgcn_permlane32_swap(b, b, false, false);
        mn = vmin_raw(__uint_as_float(ra[0]), __uint_as_float(ra[1]));
        mx = vmax_raw(__uint_as_float(rb[0]), __uint_as_float(rb[1]));
        return;
    }
    mn = vmin_raw(mn, __uint_as_float(pa));
    mx = vmax_raw(mx, __uint_as_float(pb));
#else
    mn = vmin_raw(mn, __shfl_xor(mn, M));
    mx = vmax_raw(mx, __shfl_xor(mx, M));
#endif
}

// ---- wavefront (64-lane) butterflies: every lane ends with the result ------------------------
__device__ __forceinline__ void wave_minmax(float& mn, float& mx) {
    xor_lane_minmax<1>(mn, mx); xor_lane_minmax<2>(mn, mx); xor_lane_minmax<4>(mn, mx);
    xor_lane_minmax<8>(mn, mx); xor_lane_minmax<16>(mn, mx); xor_lane_minmax<32>(mn, mx);
}
__device__ __forceinline__ float wave_min(float v) {
    float other = v;
    wave_minmax(v, other);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    float other = v;
    wave_minmax(other, v);
    return v;
}
// Wave-wide min / max of FOUR independent (mn, mx) pairs at once.  Four full butterflies are 4 x 6 exchange steps; here the
// first two steps each halve the number of pairs a lane still carries (lanes whose id differs in the step's bit keep
// different pairs: v_permlane32_swap of pair j against pair j + 2, v_permlane16_swap of the two that are left, move exactly
// the halves that have to meet), the last four are an ordinary butterfly on the one pair that remains: 2 + 1 + 4 = 7 steps'
// worth of moves instead of 24.  Lane L ends with the result of pair s(L) = 2 * bit5(L) + bit4(L), the same in the sixteen
// lanes of its row.  min / max are exact: bit-identical to four separate butterflies (the emulation below does exactly those).
__device__ __forceinline__ int wave_minmax4_slot(int lane) { return ((lane >> 5) & 1) * 2 + ((lane >> 4) & 1); }
__device__ __forceinline__ void wave_minmax4(const float (&mn)[4], const float (&mx)[4], float& rmn, float& rmx) {
#if defined(__HIP_DEVICE_COMPILE__)
    float n2[2], x2[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {          // bit 5: lanes < 32 keep pairs 0, 1; lanes >= 32 pairs 2, 3
        const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(mn[j]), __float_as_uint(mn[j + 2]), false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx[j]), __float_as_uint(mx[j + 2]), false, false);
        n2[j] = vmin_raw(__uint_as_float(a[0]), __uint_as_float(a[1]));
        x2[j] = vmax_raw(__uint_as_float(b[0]), __uint_as_float(b[1]));
    }
    // bit 4: even rows of 16 lanes keep pair 0 (of their half), odd rows pair 1
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(n2[0]), __float_as_uint(n2[1]), false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x2[0]), __float_as_uint(x2[1]), false, false);
    float kn = vmin_raw(__uint_as_float(a[0]), __uint_as_float(a[1]));
    float kx = vmax_raw(__uint_as_float(b[0]), __uint_as_float(b[1]));
    xor_lane_minmax<8>(kn, kx); xor_lane_minmax<4>(kn, kx); xor_lane_minmax<2>(kn, kx); xor_lane_minmax<1>(kn, kx);
    rmn = kn; rmx = kx;
#else
    float an[4], ax[4];
    for (int j = 0; j < 4; ++j) { an[j] = mn[j]; ax[j] = mx[j]; wave_minmax(an[j], ax[j]); }
    const int s = wave_minmax4_slot((int)(threadIdx.x % kWave));
    rmn = an[s]; rmx = ax[s];
#endif
}

// v += lane (id ^ M)'s v, with the same register-file moves (both 32-bit halves travel the same way)
template <int M>
__device__ __forceinline__ void xor_lane_add(double& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
    uint32_t plo, phi;
    if (M == 1) { plo = dpp_move<0xB1, 0xf>(lo, lo); phi = dpp_move<0xB1, 0xf>(hi, hi); }
    else if (M == 2) { plo = dpp_move<0x4E, 0xf>(lo, lo); phi = dpp_move<0x4E, 0xf>(hi, hi); }
    else if (M == 4) { plo = dpp_move<0x114, 0xA>(dpp_move<0x104, 0x5>(lo, lo), lo); phi = dpp_move<0x114, 0xA>(dpp_move<0x104, 0x5>(hi, hi), hi); }
    else if (M == 8) { plo = dpp_move<0x118, 0xC>(dpp_move<0x108, 0x3>(lo, lo), lo); phi = dpp_move<0x118, 0xC>(dpp_move<0x108, 0x3>(hi, hi), hi); }
    else {
        // after the swap of two copies, [0] and [1] hold the own and the partner's word (which is which depends on the lane's
        // row, the same way for both halves); the sum is commutative
        uint32_t a0, a1, b0, b1;
        if (M == 16) {
            const auto r = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
            const auto q = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
            a0 = r[0]; a1 = r[1]; b0 = q[0]; b1 = q[1];
        } else {
            const auto r = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
            const auto q = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
            a0 = r[0]; a1 = r[1]; b0 = q[0]; b1 = q[1];
        }
        const double x = __longlong_as_double((long long)(((unsigned long long)b0 << 32) | a0));
        const double y = __longlong_as_double((long long)(((unsigned long long)b1 << 32) | a1));
        v = x + y;
        return;
    }
    v += __longlong_as_double((long long)(((unsigned long long)phi << 32) | plo));
#else
    v += __shfl_xor(v, M);
#endif
}
__device__ __forceinline__ double wave_sum(double v) {
    xor_lane_add<32>(v); xor_lane_add<16>(v); xor_lane_add<8>(v); xor_lane_add<4>(v); xor_lane_add<2>(v); xor_lane_add<1>(v);
    return v;
}

// Deterministic block-wide sum of doubles (fixed butterfly + fixed wave order).  `sh` is a
// __shared__ double[kBlock / kWave].  Result valid in every thread.
__device__ __forceinline__ double block_sum(double v, double* sh) {
    v = wave_sum(v);
    const int wave = threadIdx.x / kWave;
    __syncthreads();   // protect `sh` against a previous use
    if ((threadIdx.x % kWave) == 0) sh[wave] = v;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) t += sh[w];
    return t;
}

// scipy.stats.norm.pdf / cdf (dfq.py:182-183, layer_transform.py:405-406): evaluated in float64 on the
// float32 argument, each rounded to float32.  cdf follows cephes ndtr (what scipy calls).
__device__ __forceinline__ void normal_pdf_cdf(float t, float& pdf, float& cdf) {
    const double x = (double)t;
    const double pdf_d = exp(-(x * x) / 2.0) / 2.5066282746310002;   // sqrt(2*pi)
    const double z = x * 0.70710678118654752440;
    const double az = fabs(z);
    double cdf_d;
    if (az < 0.70710678118654752440) {
        cdf_d = 0.5 + 0.5 * erf(z);
    } else {
        cdf_d = 0.5 * erfc(az);
        if (z > 0) cdf_d = 1.0 - cdf_d;
    }
    pdf = (float)pdf_d;
    cdf = (float)cdf_d;
}

// ---- fake-quant parameters (utils/quantize.py:49-66) -------------------------------------------
struct QParams {
    float qmin, qmax, neg_min, scale, min_value;
};

// Recipe A: min/max are Python floats in the reference -> all scalar arithmetic in double, each
// operand cast to float32 where torch casts a Python scalar to the tensor dtype.
__host__ __device__ inline QParams qparams_double(double mn, double mx, int num_bits, int symmetric) {
    double qmin, qmax, scale;
    if (symmetric) {
        qmin = -(double)(1ll << (num_bits - 1));
        qmax = (double)((1ll << (num_bits - 1)) - 1);
        mx = mx < 0 ? -mx : mx;
        mn = mn < 0 ? -mn : mn;
        if (mx < mn) mx = mn;
        scale = mx / qmax;
        mn = 0.0;
    } else {
        qmin = 0.0;
        qmax = (double)(1ll << num_bits) - 1.0;
        scale = (mx - mn) / (qmax - qmin);
    }
    if (1e-8 > scale) scale = 1e-8;   // Python max(scale, 1e-8): NaN is kept
    QParams p;
    p.qmin = (float)qmin;
    p.qmax = (float)qmax;
    p.neg_min = (float)(-mn);
    p.scale = (float)scale;
    p.min_value = (float)mn;
    return p;
}

// Recipe B: min/max are 0-dim float32 tensors in the reference (min_value=None path,
// quantize.py:24-35) -> the whole recipe stays in float32.
__host__ __device__ inline QParams qparams_float(float mn, float mx, int num_bits, int symmetric) {
    float qmin, qmax, scale, mn_used;
    if (symmetric) {
        qmin = -(float)(1ll << (num_bits - 1));
        qmax = (float)((1ll << (num_bits - 1)) - 1);
        mx = mx < 0 ? -mx : mx;
        mn = mn < 0 ? -mn : mn;
        if (mx < mn) mx = mn;
        scale = mx / qmax;
        mn_used = 0.0f;
    } else {
        qmin = 0.0f;
        qmax = (float)((double)(1ll << num_bits) - 1.0);
        const float span = mx - mn;
        scale = span / (qmax - qmin);
        mn_used = mn;
    }
    if (scale < 1e-8f) scale = 1e-8f;
    QParams p;
    p.qmin = qmin;
    p.qmax = qmax;
    p.neg_min = -mn_used;
    p.scale = scale;
    p.min_value = mn_used;
    return p;
}

// quantize.py:70-74 -- five separately rounded float32 operations (the build uses
// -ffp-contract=off so the last two never fuse into an FMA).  Returns the dequantised value and
// the integer code (as float).
__device__ __forceinline__ float fake_quant_one(float x, const QParams& p, float* code) {
    float q = x + p.neg_min;
    q = q / p.scale;
    q = (q < p.qmin) ? p.qmin : q;   // NaN-propagating clamp, like torch.clamp_
    q = (q > p.qmax) ? p.qmax : q;
    q = rintf(q);                    // round half to even == torch.round_
    *code = q;
    float y = q * p.scale;
    y = y + p.min_value;
    return y;
}

}  // namespace dfq
