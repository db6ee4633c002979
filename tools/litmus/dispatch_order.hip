// Litmus test for the assumption behind the in-launch waits of le_level_kernel and bc_chain_kernel: a
// workgroup only ever waits for workgroups with LOWER indices, which is deadlock-free as long as the hardware dispatches a
// 1-D grid in index order -- a waiting workgroup then never holds the slot a producer still needs.  HIP does not promise
// that order; this program measures it: every workgroup takes a ticket when it starts (one atomicAdd), does a little work so
// that slots stay occupied, and the host replays the tickets: when workgroup b started, how far behind was the lowest index
// that had NOT started yet?  "0 inversions beyond the resident window" is what the engine relies on: a workgroup that has
// started can have unstarted predecessors only among the workgroups dispatched in the same wave of slots, never further back
// than the number of workgroups the chip holds at once.
//   dispatch_order [grid] [threads] [spin]     prints: grid, max depth of an inversion, workgroups resident at once (estimate)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void take_ticket(unsigned* counter, unsigned* ticket, int spin) {
    __shared__ unsigned t;
    if (threadIdx.x == 0) t = atomicAdd(counter, 1u);
    __syncthreads();
    if (threadIdx.x == 0) ticket[blockIdx.x] = t;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);       // keep the slot for a while, like a tile does
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 21280;               // the sweep launch of the benchmark batch
    const int threads = argc > 2 ? atoi(argv[2]) : 256;
    const int spin = argc > 3 ? atoi(argv[3]) : 200;
    unsigned *counter, *ticket;
    hipMalloc(&counter, 4); hipMalloc(&ticket, 4 * (size_t)grid);
    int worst = 0;
    for (int rep = 0; rep < 5; ++rep) {
        hipMemset(counter, 0, 4);
        take_ticket<<<grid, threads>>>(counter, ticket, spin);
        hipDeviceSynchronize();
        std::vector<unsigned> t(grid);
        hipMemcpy(t.data(), ticket, 4 * (size_t)grid, hipMemcpyDeviceToHost);
        std::vector<int> by_ticket(grid, -1);
        for (int b = 0; b < grid; ++b) if (t[b] < (unsigned)grid) by_ticket[t[b]] = b;
        std::vector<char> started(grid, 0);
        int lowest_missing = 0;
        for (int k = 0; k < grid; ++k) {
            const int b = by_ticket[k];
            if (b < 0) { printf("ticket %d missing\n", k); return 2; }
            started[b] = 1;
            while (lowest_missing < grid && started[lowest_missing]) ++lowest_missing;
            // b started while `lowest_missing` (< b) had not: depth of the inversion
            if (lowest_missing < b) worst = std::max(worst, b - lowest_missing);
        }
    }
    int occ = 0, dev = 0;
    hipDeviceProp_t prop;
    hipGetDevice(&dev); hipGetDeviceProperties(&prop, dev);
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)take_ticket, threads, 0);
    const long resident = (long)occ * prop.multiProcessorCount;
    printf("grid %d x %d threads: deepest inversion %d workgroups; resident at once <= %ld (%d per CU x %d CUs): %s\n", grid, threads,
           worst, resident, occ, prop.multiProcessorCount, worst < resident ? "dispatch is index-ordered within the resident window" : "OUT OF ORDER");
    return worst < resident ? 0 : 1;
}
