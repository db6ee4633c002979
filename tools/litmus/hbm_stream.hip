// What the memory system of this chip delivers to the SIMPLEST streaming kernels (tuning aid, run on the GPU box):
//   read      16-byte loads, eight in flight per lane, min/max reduction           (4 B per element)
//   copy      16-byte load + 16-byte store to a second buffer                      (8 B per element)
//   scale     in place: 16-byte load, multiply, 16-byte store to the same address  (8 B per element)
// each with plain and with non-temporal accesses, over a 1 GiB buffer (four times the Infinity Cache).  The equalisation
// sweep is a `scale` with statistics: these numbers are its practical ceiling.
//   hipcc --offload-arch=gfx950 -O3 -o tools/litmus/hbm_stream tools/litmus/hbm_stream.hip && tools/litmus/hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float fvec4 __attribute__((vector_size(16)));
constexpr int kBlock = 256;
constexpr int kUnroll = 8;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool NT> __device__ __forceinline__ fvec4 ld(const fvec4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(fvec4* p, fvec4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// every workgroup owns one contiguous span of kUnroll * kBlock vectors per trip
template <bool NT> __global__ __launch_bounds__(kBlock) void k_read(const fvec4* x, size_t n4, float* out) {
    float m = -INFINITY;
    for (size_t base = (size_t)blockIdx.x * kUnroll * kBlock; base < n4; base += (size_t)gridDim.x * kUnroll * kBlock) {
        fvec4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = ld<NT>(x + base + u * kBlock + threadIdx.x);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) m = fmaxf(m, fmaxf(fmaxf(v[u][0], v[u][1]), fmaxf(v[u][2], v[u][3])));
    }
    if (m == 12345.0f) out[0] = m;
}
template <bool NTL, bool NTS> __global__ __launch_bounds__(kBlock) void k_copy(const fvec4* x, fvec4* y, size_t n4, float s) {
    for (size_t base = (size_t)blockIdx.x * kUnroll * kBlock; base < n4; base += (size_t)gridDim.x * kUnroll * kBlock) {
        fvec4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = ld<NTL>(x + base + u * kBlock + threadIdx.x);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) st<NTS>(y + base + u * kBlock + threadIdx.x, v[u] * s);
    }
}

template <class F> static float timed(F launch, int reps) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch(); launch();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)1 << 30;
    const size_t n4 = bytes / 16;
    fvec4 *x = nullptr, *y = nullptr;
    float* out = nullptr;
    CHECK(hipMalloc((void**)&x, bytes));
    CHECK(hipMalloc((void**)&y, bytes));
    CHECK(hipMalloc((void**)&out, 4));
    CHECK(hipMemset(x, 0, bytes));
    CHECK(hipMemset(y, 0, bytes));
    const int reps = 10;
    for (int wgs_per_span = 1; wgs_per_span <= 1; ++wgs_per_span) {
        for (int grid : {2048, 8192, (int)(n4 / (kUnroll * kBlock))}) {
            const float r0 = timed([&] { hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(kBlock), 0, 0, x, n4, out); }, reps);
            const float r1 = timed([&] { hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(kBlock), 0, 0, x, n4, out); }, reps);
            const float c0 = timed([&] { hipLaunchKernelGGL((k_copy<false, false>), dim3(grid), dim3(kBlock), 0, 0, x, y, n4, 1.0f); }, reps);
            const float c1 = timed([&] { hipLaunchKernelGGL((k_copy<false, true>), dim3(grid), dim3(kBlock), 0, 0, x, y, n4, 1.0f); }, reps);
            const float c2 = timed([&] { hipLaunchKernelGGL((k_copy<true, true>), dim3(grid), dim3(kBlock), 0, 0, x, y, n4, 1.0f); }, reps);
            const float s0 = timed([&] { hipLaunchKernelGGL((k_copy<false, false>), dim3(grid), dim3(kBlock), 0, 0, x, x, n4, 1.0f); }, reps);
            const float s1 = timed([&] { hipLaunchKernelGGL((k_copy<false, true>), dim3(grid), dim3(kBlock), 0, 0, x, x, n4, 1.0f); }, reps);
            const float s2 = timed([&] { hipLaunchKernelGGL((k_copy<true, true>), dim3(grid), dim3(kBlock), 0, 0, x, x, n4, 1.0f); }, reps);
            auto tbs = [&](float ms, double b_per_elem16) { return b_per_elem16 * (double)n4 / (ms * 1e-3) / 1e12; };
            printf("grid %7d | read %.2f / nt %.2f TB/s | copy %.2f / nt-store %.2f / nt-both %.2f TB/s | scale in place %.2f / nt-store %.2f / nt-both %.2f TB/s\n",
                   grid, tbs(r0, 16), tbs(r1, 16), tbs(c0, 32), tbs(c1, 32), tbs(c2, 32), tbs(s0, 32), tbs(s1, 32), tbs(s2, 32));
        }
    }
    (void)hipFree(x); (void)hipFree(y); (void)hipFree(out);
    return 0;
}
