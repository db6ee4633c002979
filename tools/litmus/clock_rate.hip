// What clock does a SHORT kernel run at?  The single-network equalisation is one ~0.7 ms launch on an otherwise idle chip; this
// probe measures the shader clock (s_memtime ticks per 100 MHz s_memrealtime tick) inside kernels of 0.2 ... 20 ms, launched
// cold (after an idle gap) and warm (right behind a long busy kernel), on 1 and on 768 workgroups.
//   hipcc -O2 --offload-arch=gfx950 tools/litmus/clock_rate.hip -o tools/litmus/clock_rate && tools/litmus/clock_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void spin_kernel(long long iters, long long* out) {
    const long long c0 = clock64(), w0 = wall_clock64();
    float x = (float)threadIdx.x;
    for (long long i = 0; i < iters; ++i) x = x * 1.0000001f + 0.5f;     // a dependent chain: ~2 instructions per iteration
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = c1 - c0; out[blockIdx.x * 4 + 1] = w1 - w0; out[blockIdx.x * 4 + 2] = (long long)x; }
}

static void run(const char* what, int blocks, long long iters, long long* d, bool idle_first) {
    std::vector<long long> h(4 * blocks);
    if (idle_first) std::this_thread::sleep_for(std::chrono::milliseconds(300));
    else { hipLaunchKernelGGL(spin_kernel, dim3(1024), dim3(256), 0, 0, 4000000LL, d); }    // ~20 ms of load in front
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, 0, iters, d);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, sizeof(long long) * 4 * blocks, hipMemcpyDeviceToHost);
    double mhz = 0, us = 0;
    for (int b = 0; b < blocks; ++b) { mhz += 100.0 * (double)h[4 * b] / (double)h[4 * b + 1]; us += (double)h[4 * b + 1] / 100.0; }
    printf("%-34s blocks %4d  kernel %8.1f us  shader clock %7.1f MHz  (%.2f cycles per iteration)\n", what, blocks, us / blocks, mhz / blocks,
           (double)h[0] / (double)iters);
}

int main() {
    long long* d;
    hipMalloc(&d, sizeof(long long) * 4 * 1024);
    for (int rep = 0; rep < 2; ++rep) {
        for (long long iters : {20000LL, 100000LL, 1000000LL}) {
            run("cold, 1 workgroup", 1, iters, d, true);
            run("cold, 768 workgroups", 768, iters, d, true);
            run("warm, 768 workgroups", 768, iters, d, false);
        }
    }
    hipFree(d);
    return 0;
}
