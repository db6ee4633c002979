// Litmus test for the "one launch per sweep" plan: can a workgroup publish words with
// device-scope atomics, bump a counter, and have workgroups on OTHER XCDs that spin on the counter read those
// words correctly -- without release/acquire fences (which would write back / invalidate whole L2s)?
//   producers (blocks 0..P-1):   atomicMax(word[i], value(iter));  s_waitcnt vmcnt(0);  atomicAdd(counter, 1)
//   consumers (blocks P..P+C-1): spin until counter >= P*(iter+1) (device-scope atomic load), then read every word
//                                with a device-scope atomic load and compare with value(iter).
// Many iterations reuse the same words (stale-line hazard).  Prints the number of mismatches and give-ups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

constexpr int kWords = 4096;

__global__ void litmus(unsigned* words, unsigned* counter, unsigned* errors, int iter, int n_prod, int mode) {
    const unsigned want = 1000u + (unsigned)iter;
    if ((int)blockIdx.x < n_prod) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kWords; i += n_prod * blockDim.x) {
            if (mode == 0) atomicMax(&words[i], want);
            else words[i] = want;                                   // plain store: expected to FAIL without fences
        }
        if (mode == 2) __threadfence();                              // plain stores + full fence
        __builtin_amdgcn_s_waitcnt(0);                               // all counters to zero (vmcnt included)
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(counter, 1u);
    } else {
        const unsigned target = (unsigned)n_prod * (unsigned)(iter + 1);
        if (threadIdx.x == 0) {
            long spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > 20000000) { atomicAdd(&errors[1], 1u); break; }
            }
        }
        __syncthreads();
        if (mode == 2) __threadfence();
        unsigned bad = 0;
        for (int i = threadIdx.x; i < kWords; i += blockDim.x) {
            const unsigned v = (mode == 0) ? __hip_atomic_load(&words[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : words[i];
            if (v != want) ++bad;
        }
        if (bad) atomicAdd(&errors[0], bad);
    }
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    const int n_prod = 64, n_cons = 192;
    unsigned *words, *counter, *errors;
    hipMalloc(&words, kWords * 4); hipMalloc(&counter, 4); hipMalloc(&errors, 8);
    hipMemset(words, 0, kWords * 4); hipMemset(counter, 0, 4); hipMemset(errors, 0, 8);
    for (int it = 0; it < iters; ++it) litmus<<<n_prod + n_cons, 256>>>(words, counter, errors, it, n_prod, mode);
    hipDeviceSynchronize();
    unsigned h[2];
    hipMemcpy(h, errors, 8, hipMemcpyDeviceToHost);
    printf("mode %d (%s): %d launches of %d producers + %d consumers: %u stale words, %u give-ups\n", mode,
           mode == 0 ? "atomics + sc1 loads, no fence" : mode == 1 ? "plain stores/loads, no fence" : "plain + __threadfence", iters,
           n_prod, n_cons, h[0], h[1]);
    return 0;
}
