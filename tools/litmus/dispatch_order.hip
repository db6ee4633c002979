// Litmus test for the assumption behind the in-launch waits of le_level_kernel and bc_chain_kernel: a
// workgroup only ever waits for workgroups with LOWER indices.  That is deadlock-free if every XCD dispatches ITS share of a
// 1-D grid in index order: take the lowest-index workgroup b that has not finished; everything it waits for has finished, so
// it completes if it is resident; and if it is not, every workgroup its XCD dispatched before it has a lower index, has
// therefore finished and freed its slot, so b is dispatched next.  (The XCDs may run arbitrarily far apart from one another --
// they do, see the "whole chip" figure below -- without breaking the argument.)  HIP promises no dispatch order; this program
// measures it: every workgroup takes a ticket when it starts (one atomicAdd) and records the XCD it runs on (XCC_ID), keeps
// its slot for a while like a tile does, and the host replays the tickets XCD by XCD: when workgroup b started, how many
// workgroups of the same XCD with lower indices had not?  A ticket is taken a little after the dispatch, so workgroups in
// flight on the XCD's CUs can overtake one another; the engine relies on nothing overtaking by more than that -- the deepest
// inversion stays below the number of workgroups one XCD holds at once.  (With spin 0 the workgroups end as they start and the
// figure measures start-up latency instead -- a CU fetching the kernel's code for the first time while its neighbours retire
// hundreds of workgroups -- so the verdict is only drawn for spin > 0.)
// The second half is the property itself in its worst case: a CHAIN, every workgroup waits (bounded, 2 s) until its
// predecessor has finished -- the deepest dependency the index-ordered waits allow, with all but one resident workgroup
// waiting at any time.  An XCD that dispatched a workgroup past an unfinished stretch longer than it can hold would stall it.
//   dispatch_order [grid] [threads] [spin]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void take_ticket(unsigned* counter, unsigned* ticket, unsigned* xcd, int spin) {
    if (threadIdx.x == 0) {
        ticket[blockIdx.x] = atomicAdd(counter, 1u);
        xcd[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u;       // HW_REG_XCC_ID
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8);       // keep the slot for a while, like a tile does
}

__global__ void chain(unsigned* done, unsigned* timeouts) {
    if (threadIdx.x == 0 && blockIdx.x > 0) {
        const long long t0 = wall_clock64();                         // 100 MHz
        while (__hip_atomic_load(done + blockIdx.x - 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) {
            if (wall_clock64() - t0 > 200000000ll) { atomicAdd(timeouts, 1u); break; }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(done + blockIdx.x, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// ids of one XCD's workgroups in index order + their tickets: the deepest "b started while an earlier one had not", counted in
// workgroups of that XCD
static int deepest_inversion(const std::vector<int>& ids, const std::vector<unsigned>& t) {
    const int n = (int)ids.size();
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return t[ids[a]] < t[ids[b]]; });
    std::vector<char> started(n, 0);
    int lowest_missing = 0, worst = 0;
    for (int k = 0; k < n; ++k) {
        const int pos = order[k];
        started[pos] = 1;
        while (lowest_missing < n && started[lowest_missing]) ++lowest_missing;
        if (lowest_missing < pos) worst = std::max(worst, pos - lowest_missing);
    }
    return worst;
}

int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 21280;               // the sweep launch of the benchmark batch
    const int threads = argc > 2 ? atoi(argv[2]) : 256;
    const int spin = argc > 3 ? atoi(argv[3]) : 200;
    unsigned *counter, *ticket, *xcd;
    (void)hipMalloc(&counter, 4); (void)hipMalloc(&ticket, 4 * (size_t)grid); (void)hipMalloc(&xcd, 4 * (size_t)grid);
    int worst_xcd = 0, worst_chip = 0, n_xcd = 0, round_robin = 1;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipMemset(counter, 0, 4);
        take_ticket<<<grid, threads>>>(counter, ticket, xcd, spin);
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
        std::vector<unsigned> t(grid), x(grid);
        (void)hipMemcpy(t.data(), ticket, 4 * (size_t)grid, hipMemcpyDeviceToHost);
        (void)hipMemcpy(x.data(), xcd, 4 * (size_t)grid, hipMemcpyDeviceToHost);
        std::vector<std::vector<int>> share(16);
        std::vector<int> all(grid);
        for (int b = 0; b < grid; ++b) { share[x[b]].push_back(b); all[b] = b; }
        n_xcd = 0;
        for (auto& ids : share) if (!ids.empty()) { ++n_xcd; worst_xcd = std::max(worst_xcd, deepest_inversion(ids, t)); }
        for (int b = 0; b + n_xcd < grid; ++b) round_robin &= x[b + n_xcd] == x[b];
        worst_chip = std::max(worst_chip, deepest_inversion(all, t));
    }
    int occ = 0, dev = 0;
    hipDeviceProp_t prop;
    (void)hipGetDevice(&dev); (void)hipGetDeviceProperties(&prop, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)take_ticket, threads, 0);
    const int window = occ * prop.multiProcessorCount / std::max(n_xcd, 1);
    // the chain
    unsigned* timeouts = counter;
    (void)hipMemset(ticket, 0, 4 * (size_t)grid); (void)hipMemset(timeouts, 0, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    chain<<<grid, threads>>>(ticket, timeouts);
    (void)hipEventRecord(e1, 0);
    if (hipDeviceSynchronize() != hipSuccess) { printf("chain launch failed\n"); return 2; }
    unsigned stalled = 0;
    float ms = 0;
    (void)hipMemcpy(&stalled, timeouts, 4, hipMemcpyDeviceToHost);
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("chain of %d workgroups, each waiting for the one before it: %u stalled, %.1f ms (%.2f us per hand-off)\n", grid, stalled, ms, 1e3 * ms / grid);
    const bool ok = (spin == 0 || worst_xcd < window) && stalled == 0;
    printf("grid %d x %d threads, %d XCDs (%s): deepest inversion within an XCD %d of its workgroups, one XCD holds <= %d at once (%d per CU); "
           "whole chip %d workgroups: %s\n", grid, threads, n_xcd, round_robin ? "workgroup i on XCD i mod n" : "NOT round-robin", worst_xcd, window, occ,
           worst_chip, spin == 0 ? "start-up latency, no verdict (spin 0)" : worst_xcd < window ? "every XCD dispatches its share in index order" : "OUT OF ORDER");
    return ok ? 0 : 1;
}
