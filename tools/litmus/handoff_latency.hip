// What does a statistics hand-off between workgroups COST on this chip?  The resident equalisation kernel (dfq_le_resident.hip)
// spends 70 % of its wave cycles parked: a sweep of a tile is a handful of dependent trips through the memory system.  This
// probe measures the primitives those trips are made of, between workgroups on the same XCD and on different XCDs:
//
//   pingpong   : A stores a tagged word (device-scope atomic store), B polls it (device-scope loads) and answers -- the one-way
//                latency of a single-producer ("relaxed") hand-off;
//   allreduce  : P workgroups merge W channels x 2 words into shared tagged words with atomicMax, wait until their atomics have
//                been performed (s_waitcnt 0), bump a counter (eight copies), poll it until all P have arrived and read ALL
//                W x 2 x 2 words back -- the "strict" publication of a layer's column statistics, the way the kernel does it;
//                per phase: publish (atomics + wait), poll (until the last sibling is in), fetch.
//
//   hipcc -O2 --offload-arch=gfx950 tools/litmus/handoff_latency.hip -o tools/litmus/handoff_latency
//   tools/litmus/handoff_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
constexpr int kStride = 16;      // u64 per 128-byte line

__device__ __forceinline__ u64 ld_agent(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u); }

// ---------------------------------------------------------------------------------------------------------------------------
// pingpong between workgroup `a` and workgroup `b` of the launch (all others leave at once)
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void pingpong(u64* words, int a, int b, int iters, int nap, long long* out) {
    const int me = (int)blockIdx.x;
    if (threadIdx.x == 0) out[8 + me] = xcc_id();
    if (me != a && me != b) return;
    u64* mine = words + (me == a ? 0 : kStride);
    u64* theirs = words + (me == a ? kStride : 0);
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        for (int k = 1; k <= iters; ++k) {
            if (me == a) st_agent(mine, (u64)k);
            long spins = 0;
            while (ld_agent(theirs) < (u64)k) { if (nap) __builtin_amdgcn_s_sleep(1); if (++spins > 100000000) break; }
            if (me == b) st_agent(mine, (u64)k);
        }
        const long long t1 = wall_clock64();
        if (me == a) out[0] = t1 - t0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// allreduce among the workgroups with blockIdx % stride == 0 (P of them)
// ---------------------------------------------------------------------------------------------------------------------------
template <int kLoad>      // 0: two 8-byte device-scope atomic loads per (min, max) pair; 1: one 16-byte volatile load per pair
__global__ void allreduce(u64* stats, u64* counters, int stride, int P, int W, int iters, long long* out, unsigned* errors) {
    const int me = (int)blockIdx.x;
    if (me % stride != 0 || me / stride >= P) return;
    const int rank = me / stride;
    __shared__ int sh_flag;
    long long t_pub = 0, t_poll = 0, t_fetch = 0;
    const int tid = threadIdx.x;
    unsigned bad = 0;
    const long long t_begin = wall_clock64();
    for (int k = 0; k < iters; ++k) {
        const u64 tag = (u64)(k + 1) << 32;
        u64* arena = stats + (size_t)(k & 1) * (size_t)W * 4;            // two parities; per channel: r1 (min, max), r2 (min, max)
        const long long t0 = wall_clock64();
        // publish: this workgroup's contribution to the r2 words of every channel (the column statistics of its rows)
        for (int c = tid; c < W; c += blockDim.x) {
            atomicMax(arena + 4 * (size_t)c + 2, tag | (u64)(unsigned)(1000 + rank + c));
            atomicMax(arena + 4 * (size_t)c + 3, tag | (u64)(unsigned)(2000 + rank + c));
            if (rank == 0) {                                              // the r1 words have one producer (another layer's tile)
                st_agent(arena + 4 * (size_t)c + 0, tag | 7ull);
                st_agent(arena + 4 * (size_t)c + 1, tag | 9ull);
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid < 8) atomicAdd(counters + (size_t)tid * kStride, 1ull);
        const long long t1 = wall_clock64();
        if (tid == 0) {
            const u64 want = (u64)P * (u64)(k + 1);
            long spins = 0;
            while (ld_agent(counters + (size_t)(me & 7) * kStride) < want) { __builtin_amdgcn_s_sleep(1); if (++spins > 300000) { ++bad; break; } }
            sh_flag = 1;
        }
        __syncthreads();
        const long long t2 = wall_clock64();
        // fetch: every word of every channel (what a tile of complete rows needs to solve all its column scales)
        u64 acc = 0;
        for (int c0 = 0; c0 < W; c0 += 4 * blockDim.x) {
            u64 w[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = min(c0 + j * (int)blockDim.x + tid, W - 1);
                const u64* p = arena + 4 * (size_t)c;
                if (kLoad == 0) {
                    w[j][0] = ld_agent(p); w[j][1] = ld_agent(p + 1); w[j][2] = ld_agent(p + 2); w[j][3] = ld_agent(p + 3);
                } else {
                    typedef unsigned uv4 __attribute__((ext_vector_type(4)));
                    const uv4 lo = *(const volatile uv4*)p, hi = *(const volatile uv4*)(p + 2);
                    w[j][0] = ((u64)lo[1] << 32) | lo[0]; w[j][1] = ((u64)lo[3] << 32) | lo[2];
                    w[j][2] = ((u64)hi[1] << 32) | hi[0]; w[j][3] = ((u64)hi[3] << 32) | hi[2];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = min(c0 + j * (int)blockDim.x + tid, W - 1);
                // complete?  the r2 words must carry this sweep's tag AND the maximum over all P contributions
                if ((w[j][2] >> 32) != (u64)(k + 1) || (unsigned)w[j][2] != (unsigned)(1000 + (P - 1) + c)) ++bad;
                if ((w[j][3] >> 32) != (u64)(k + 1) || (unsigned)w[j][3] != (unsigned)(2000 + (P - 1) + c)) ++bad;
                acc += w[j][0] + w[j][1];
            }
        }
        if (acc == 0x1234567ull) ++bad;
        __syncthreads();
        const long long t3 = wall_clock64();
        t_pub += t1 - t0; t_poll += t2 - t1; t_fetch += t3 - t2;
    }
    const long long t_end = wall_clock64();
    if (bad) atomicAdd(errors, bad);
    if (tid == 0) {
        out[rank * 4 + 0] = t_end - t_begin; out[rank * 4 + 1] = t_pub; out[rank * 4 + 2] = t_poll; out[rank * 4 + 3] = t_fetch;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// the same all-reduce WITHOUT atomics and without a counter: every participant STORES its W x 2 partial words into its own slot
// (fire and forget), then reduces ITS slice of the channels over all P slots (polling the tagged words until every sibling's
// have arrived), stores the merged words, and finally reads all W x 2 merged words (+ the W x 2 single-producer words) back,
// polling their tags.  Two dependent hand-offs of the pingpong kind instead of atomics -> performed -> counter -> poll -> fetch.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ void slot_allreduce(u64* slots, u64* stats, int stride, int P, int W, int iters, long long* out, unsigned* errors) {
    const int me = (int)blockIdx.x;
    if (me % stride != 0 || me / stride >= P) return;
    const int rank = me / stride;
    __shared__ unsigned sh_red[2 * 1024];
    long long t_pub = 0, t_red = 0, t_fetch = 0;
    const int tid = threadIdx.x;
    const int per = (W + P - 1) / P;                      // channels this participant reduces
    const int c_lo = min(rank * per, W), c_n = min(per, W - c_lo);
    unsigned bad = 0;
    const long long t_begin = wall_clock64();
    for (int k = 0; k < iters; ++k) {
        const u64 tag = (u64)(k + 1) << 32;
        u64* arena = stats + (size_t)(k & 1) * (size_t)W * 4;
        u64* slot = slots + (size_t)(k & 1) * (size_t)P * W * 2;
        const long long t0 = wall_clock64();
        for (int c = tid; c < W; c += blockDim.x) {
            st_agent(slot + ((size_t)rank * W + c) * 2 + 0, tag | (u64)(unsigned)(1000 + rank + c));
            st_agent(slot + ((size_t)rank * W + c) * 2 + 1, tag | (u64)(unsigned)(2000 + rank + c));
            if (rank == 0) {
                st_agent(arena + 4 * (size_t)c + 0, tag | 7ull);
                st_agent(arena + 4 * (size_t)c + 1, tag | 9ull);
            }
        }
        for (int i = tid; i < 2 * c_n; i += blockDim.x) sh_red[i] = 0u;
        __syncthreads();
        const long long t1 = wall_clock64();
        // reduce my slice over all P slots: item = (participant p, channel j of the slice), two words each
        for (int it = tid; it < P * c_n; it += blockDim.x) {
            const int p = it / c_n, j = it - p * c_n;
            const u64* src = slot + ((size_t)p * W + c_lo + j) * 2;
            u64 a, b;
            long spins = 0;
            for (;;) {
                a = ld_agent(src); b = ld_agent(src + 1);
                if ((a >> 32) == (u64)(k + 1) && (b >> 32) == (u64)(k + 1)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 300000) { ++bad; break; }
            }
            atomicMax(&sh_red[2 * j], (unsigned)a);
            atomicMax(&sh_red[2 * j + 1], (unsigned)b);
        }
        __syncthreads();
        for (int j = tid; j < c_n; j += blockDim.x) {
            st_agent(arena + 4 * (size_t)(c_lo + j) + 2, tag | (u64)sh_red[2 * j]);
            st_agent(arena + 4 * (size_t)(c_lo + j) + 3, tag | (u64)sh_red[2 * j + 1]);
        }
        const long long t2 = wall_clock64();
        // fetch everything, polling the tags (each thread its own words)
        u64 acc = 0;
        for (int c0 = 0; c0 < W; c0 += 4 * blockDim.x) {
            u64 w[4][4];
            long spins = 0;
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = min(c0 + j * (int)blockDim.x + tid, W - 1);
                    const u64* p = arena + 4 * (size_t)c;
                    w[j][0] = ld_agent(p); w[j][1] = ld_agent(p + 1); w[j][2] = ld_agent(p + 2); w[j][3] = ld_agent(p + 3);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    for (int q = 0; q < 4; ++q) ok = ok && (w[j][q] >> 32) == (u64)(k + 1);
                if (ok) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 300000) { ++bad; break; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = min(c0 + j * (int)blockDim.x + tid, W - 1);
                if ((unsigned)w[j][2] != (unsigned)(1000 + (P - 1) + c)) ++bad;
                if ((unsigned)w[j][3] != (unsigned)(2000 + (P - 1) + c)) ++bad;
                acc += w[j][0] + w[j][1];
            }
        }
        if (acc == 0x1234567ull) ++bad;
        __syncthreads();
        const long long t3 = wall_clock64();
        t_pub += t1 - t0; t_red += t2 - t1; t_fetch += t3 - t2;
    }
    const long long t_end = wall_clock64();
    if (bad) atomicAdd(errors, bad);
    if (tid == 0) {
        out[rank * 4 + 0] = t_end - t_begin; out[rank * 4 + 1] = t_pub; out[rank * 4 + 2] = t_red; out[rank * 4 + 3] = t_fetch;
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    u64 *words, *stats, *counters;
    long long* out;
    unsigned* errors;
    const int kMaxW = 1024, kMaxP = 160;
    hipMalloc(&words, 4096);
    hipMalloc(&stats, sizeof(u64) * 2 * 4 * kMaxW);
    hipMalloc(&counters, sizeof(u64) * 8 * kStride);
    hipMalloc(&out, sizeof(long long) * 4 * 1024);
    hipMalloc(&errors, 4);
    std::vector<long long> h(4 * 1024);

    // ---- pingpong: blocks 0 and 8 share an XCD (i mod 8), blocks 0 and 1 do not ----
    for (int nap = 0; nap <= 1; ++nap)
        for (int b : {8, 1, 4}) {
            hipMemset(words, 0, 4096);
            const int iters = 5000;
            hipLaunchKernelGGL(pingpong, dim3(16), dim3(256), 0, 0, words, 0, b, iters, nap, out);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, sizeof(long long) * 32, hipMemcpyDeviceToHost);
            printf("pingpong blocks 0 (XCD %lld) <-> %d (XCD %lld)%s: %.3f us per round trip, %.3f us one way\n", h[8], b, h[8 + b],
                   nap ? ", s_sleep(1) between polls" : "", (double)h[0] / 100.0 / iters, (double)h[0] / 200.0 / iters);
        }

    // ---- allreduce: P participants, W channels; spread over the XCDs (stride 1) or all on XCD 0 (stride 8) ----
    struct Case { int stride, P, W; };
    const Case cases[] = {{1, 40, 960}, {8, 40, 960}, {1, 80, 320}, {8, 80, 320}, {1, 8, 960}, {8, 8, 960}, {1, 2, 960}, {1, 160, 64}};
    for (int load = 0; load <= 1; ++load)
        for (const Case& c : cases) {
            if (c.P > kMaxP || c.W > kMaxW) continue;
            hipMemset(stats, 0, sizeof(u64) * 2 * 4 * kMaxW);
            hipMemset(counters, 0, sizeof(u64) * 8 * kStride);
            hipMemset(errors, 0, 4);
            const int iters = 400;
            const int grid = c.stride * c.P;
            if (grid > 8 * 32 * 3) { printf("skip: grid %d is not co-resident\n", grid); continue; }
            if (load == 0) hipLaunchKernelGGL(allreduce<0>, dim3(grid), dim3(256), 0, 0, stats, counters, c.stride, c.P, c.W, iters, out, errors);
            else hipLaunchKernelGGL(allreduce<1>, dim3(grid), dim3(256), 0, 0, stats, counters, c.stride, c.P, c.W, iters, out, errors);
            hipDeviceSynchronize();
            unsigned bad = 0;
            hipMemcpy(&bad, errors, 4, hipMemcpyDeviceToHost);
            hipMemcpy(h.data(), out, sizeof(long long) * 4 * c.P, hipMemcpyDeviceToHost);
            double tot = 0, pub = 0, poll = 0, fetch = 0;
            for (int r = 0; r < c.P; ++r) { tot += h[4 * r]; pub += h[4 * r + 1]; poll += h[4 * r + 2]; fetch += h[4 * r + 3]; }
            const double n = 100.0 * iters * c.P;
            printf("allreduce P=%3d W=%4d %-22s %-14s: %6.2f us per round = publish %5.2f + poll %5.2f + fetch %5.2f   (%u incomplete words)\n",
                   c.P, c.W, c.stride == 1 ? "spread over the XCDs" : "all on one XCD", load ? "16-byte loads" : "8-byte loads",
                   tot / n, pub / n, poll / n, fetch / n, bad);
        }
    // ---- the same rounds through per-participant slots: stores and tagged polls only ----
    u64* slots;
    hipMalloc(&slots, sizeof(u64) * 2 * 2 * (size_t)kMaxP * kMaxW);
    for (const Case& c : cases) {
        hipMemset(stats, 0, sizeof(u64) * 2 * 4 * kMaxW);
        hipMemset(slots, 0, sizeof(u64) * 2 * 2 * (size_t)kMaxP * kMaxW);
        hipMemset(errors, 0, 4);
        const int iters = 400;
        const int grid = c.stride * c.P;
        if (grid > 8 * 32 * 3) continue;
        hipLaunchKernelGGL(slot_allreduce, dim3(grid), dim3(256), 0, 0, slots, stats, c.stride, c.P, c.W, iters, out, errors);
        hipDeviceSynchronize();
        unsigned bad = 0;
        hipMemcpy(&bad, errors, 4, hipMemcpyDeviceToHost);
        hipMemcpy(h.data(), out, sizeof(long long) * 4 * c.P, hipMemcpyDeviceToHost);
        double tot = 0, pub = 0, red = 0, fetch = 0;
        for (int r = 0; r < c.P; ++r) { tot += h[4 * r]; pub += h[4 * r + 1]; red += h[4 * r + 2]; fetch += h[4 * r + 3]; }
        const double n = 100.0 * iters * c.P;
        printf("slot allreduce P=%3d W=%4d %-22s: %6.2f us per round = store %5.2f + reduce slice %5.2f + fetch merged %5.2f   (%u wrong words)\n",
               c.P, c.W, c.stride == 1 ? "spread over the XCDs" : "all on one XCD", tot / n, pub / n, red / n, fetch / n, bad);
    }
    return 0;
}
