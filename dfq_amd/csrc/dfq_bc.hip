// Analytic bias correction (dfq.py:173-293) for gfx950.
//
// Two stages on one stream:
//   1. per-tensor min/max of every corrected layer            (one launch, all layers)
//   2. the sequential chain (ONE launch); a step forms the quant-error row sums of its rows,
//        eps[o, i] = sum_k (Q(W)-W)[o,i,k]  (dfq.py:216-219),  in registers straight from W -- until round 3 a
//        separate launch wrote eps[O, I/g] for all layers and the chain read it back: for 1x1 layers a buffer the size
//        of the weights, 8 B per (o, i) pair of traffic; now a pass moves 8 B per weight (min/max read + this read) --
//        then:
//        E[x] from the BN proxies (ReLU moment matching dfq.py:182-184,238-242; add/cat merge
//        :244-270)  ->  bias[g] = eps[g] . E[g] (:281-287)  ->  b -= bias (:290-292)  ->
//        next BN's beta~ += -bias (:204-206, 293).
// Stage 3 is inherently serial across layers (each correction feeds the next expectation, and every
// output row needs the whole expectation vector); the stream order carries that dependency, no host
// synchronisation is involved.  The float64 pdf/cdf evaluation of the ReLU moment is done ONCE per
// BN channel: a BN's beta~ changes exactly once (when the layer in front of it is corrected), so the
// thread that applies that update also refreshes the channel's cached E[ReLU(.)]; a step then only
// gathers cached values.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <unordered_map>
#include <memory>
#include <vector>

#include "dfq_common.hpp"
#include "dfq_le_shared.hpp"

namespace dfq {

#if DFQ_BC_TRACE
constexpr int kBcTraceWgs = 32768, kBcTraceWords = 8;
__device__ long long g_bc_trace[kBcTraceWgs * kBcTraceWords];
#define BC_STAMP(k) do { if (chained && threadIdx.x == 0 && blockIdx.x < kBcTraceWgs) g_bc_trace[blockIdx.x * kBcTraceWords + (k)] = (long long)wall_clock64(); } while (0)
#define BC_STAMP_VAL(k, v) do { if (chained && threadIdx.x == 0 && blockIdx.x < kBcTraceWgs) g_bc_trace[blockIdx.x * kBcTraceWords + (k)] = (long long)(v); } while (0)
#else
#define BC_STAMP(k) do { } while (0)
#define BC_STAMP_VAL(k, v) do { } while (0)
#endif

constexpr int kMmChunk = kBlock * 64;   // floats per workgroup of the min/max pass: a read-only stream wants long runs (4 trips of 4 loads)
constexpr int kQePairs = 4;            // (o, i) pairs per thread of the quant-error kernel for khkw == 1 layers
constexpr int kExpectMax = 8192;       // floats of E[x] kept in LDS (32 KiB)
constexpr int kExpectSmall = 2048;     // ... by the kernel variant used when every step of a launch fits (8 KiB)
constexpr int kRowsPerBlock = kBlock / kWave;   // output rows of the matvec per workgroup: one per wave
constexpr int kBcRegs = 24;            // eps values a lane preloads (rows up to 1536 inputs per group)
constexpr double kBcSkewDefault = 0.0; // chain positions a network of a batch runs behind the network in front of it (DFQ_BC_SKEW; see the plan)

struct BcLayerDev {
    const float* w;
    float* eps;              // [O * I/g]
    int64_t n;               // O * I/g * khkw
    int64_t pairs;           // O * I/g
    int32_t khkw;
    int32_t pad;
};

constexpr int kStepSources = 6;        // sources of a step carried in the kernarg

struct BcSourceDev {
    const float* fw;
    const float* fb;
    const float* cache;      // relu_mean(fw, fb) per channel, kept current by the producing step
    int32_t channels, relu, concat;
    int32_t tag_off;         // first tagged-value slot of this BN when a step of the plan rewrites it, else -1 (see BcDep)
};

struct BcStepDev {
    const float* w;          // the layer's weights [O, I/g, khkw]
    const uint32_t* mm;      // its (min, max) slots, filled by bc_minmax_kernel
    int32_t khkw;
    int32_t fold;            // index of the depthwise step folded into this step's per-row tail (BcFoldDev table), or -1
    const float* eps;        // debug copy of the row sums (DFQ_BC_EPS=1: written by bc_quant_error_kernel, read by nobody)
    float* bias;             // layer bias [O], in/out
    float* next_bn_bias;     // [O] or null
    const float* next_bn_weight;   // gamma~ of that BN (null if nobody reads its ReLU moment)
    float* next_cache;       // its relu_mean cache
    float* corr;             // [O] out: the correction `bias` of dfq.py:285-287
    int32_t out_ch, in_per_group, source_begin, source_count, expect_len, inline_sources;
    int32_t lg_lanes, chunks, rows_per_block;          // work split, see bc_step_kernel
    int32_t next_tag_off;    // tagged-value slots of the next BN, or -1
    int32_t mm_index, mm_blocks, wait_cache;   // one-launch correction: the layer's arrival counter, how many min/max blocks feed it; 1: a
                                               // source is a never-rewritten BN read through a ReLU (its moment is refreshed by blocks of the launch)
    BcSourceDev src[kStepSources];   // copy of sources[source_begin ...] when source_count <= kStepSources
};

// A depthwise step folded into the per-row tail of the step that produces its only source (round 4).  A layer with ONE input
// channel per group and as many groups as outputs corrects output channel o with eps[o] * E[o] alone (dfq.py:281-287 with
// I/g = 1), and E[o] is the beta~ / ReLU moment of channel o of the BN the previous step has just rewritten -- a value the
// thread that owns row o of THAT step holds in a register.  So that thread also performs the depthwise layer's correction of
// channel o: its nine taps and per-row operands are requested with the step's own, the row sum of the quantisation error is
// formed while the expectation is assembled, and after the step's own update the thread does the one multiplication, the
// two updates of dfq.py:290-293 and the next BN's ReLU moment.  A hand-over through the memory system (3.8 us for a single
// network, 7.5 us in a batch of 32) per depthwise layer disappears: MobileNetV2's chain has 35 dependent steps instead of 52.
// Same operations in the same order as the step would perform: bit-identical (DFQ_BC_FOLD=0 keeps the step).
constexpr int kFoldTaps = 9;            // taps of a folded layer a thread preloads (3 x 3 depthwise kernels)
struct BcFoldDev {
    const float* w;              // [O, 1, khkw]
    const uint32_t* mm;          // (min, max) slots of the folded layer
    float* bias;                 // [O] in/out
    float* next_bn_bias;         // [O] or null
    const float* next_bn_weight; // gamma~ of that BN (null if nobody reads its ReLU moment)
    float* next_cache;
    float* corr;                 // [O] out
    float* eps;                  // debug copy target is written by bc_quant_error_kernel; unused here
    int32_t khkw, next_tag_off, src_relu, pad;
    int32_t mm_index, mm_blocks;           // (as in BcStepDev)
};

struct BcCacheSeg {          // one BN whose ReLU moment is cached
    const float* fw;
    const float* fb;
    float* cache;
    int32_t channels, begin; // begin: first global channel index of this segment in the init launch
};

__device__ __forceinline__ int bc_find(const int32_t* __restrict__ begin, int n, int block) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (begin[mid] <= block) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ float relu_mean(float w, float b);
struct BcCacheSeg;
__device__ __forceinline__ void bc_cache_init_block(const BcCacheSeg* __restrict__ segs, int n_segs, int total, int block, bool device_scope);

// per-tensor (min, max): one block of `mm_chunk` floats of one layer, merged into the layer's two slots
__device__ __forceinline__ void bc_minmax_block(const BcLayerDev* __restrict__ layers, const int32_t* __restrict__ block_begin, int n_layers,
                                                uint32_t* __restrict__ slots, int mm_chunk, int block, uint32_t* arrive) {
    __shared__ float sh_mn[kBlock / kWave];
    __shared__ float sh_mx[kBlock / kWave];
    const int l = bc_find(block_begin, n_layers, block);
    const BcLayerDev L = layers[l];
    const int64_t b = (int64_t)(block - block_begin[l]) * mm_chunk;
    const int64_t e = (b + mm_chunk < L.n) ? b + mm_chunk : L.n;
    float mn = INFINITY, mx = -INFINITY;
    if ((((uintptr_t)L.w) & 15u) == 0) {
        // 16-byte vectors over the aligned body of the chunk (chunk starts are multiples of 4 floats)
        const int64_t e4 = b + ((e - b) & ~(int64_t)3);
        // four independent 16-byte loads per trip (a read-only pass with one load in flight per lane leaves most of the
        // memory pipeline idle), raw v_min / v_max (no canonicalisation prologue)
        int64_t i = b + 4 * (int64_t)threadIdx.x;
        for (; i + 12 * kBlock < e4; i += 16 * kBlock) {
            fvec4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = kReadNt ? DFQ_NT_LOAD((const fvec4*)(L.w + i + u * 4 * kBlock)) : *(const fvec4*)(L.w + i + u * 4 * kBlock);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mn = vmin_raw(vmin_raw(mn, v[u][0]), vmin_raw(v[u][1], vmin_raw(v[u][2], v[u][3])));
                mx = vmax_raw(vmax_raw(mx, v[u][0]), vmax_raw(v[u][1], vmax_raw(v[u][2], v[u][3])));
            }
        }
        for (; i < e4; i += 4 * kBlock) {
            const fvec4 v = *(const fvec4*)(L.w + i);
            mn = vmin_raw(vmin_raw(mn, v[0]), vmin_raw(v[1], vmin_raw(v[2], v[3])));
            mx = vmax_raw(vmax_raw(mx, v[0]), vmax_raw(v[1], vmax_raw(v[2], v[3])));
        }
        for (int64_t i = e4 + threadIdx.x; i < e; i += kBlock) {
            const float v = L.w[i];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    } else {
        for (int64_t i = b + threadIdx.x; i < e; i += kBlock) {
            const float v = L.w[i];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if ((threadIdx.x % kWave) == 0) { sh_mn[threadIdx.x / kWave] = mn; sh_mx[threadIdx.x / kWave] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = sh_mn[0], c = sh_mx[0];
        for (int w = 1; w < kBlock / kWave; ++w) { a = fminf(a, sh_mn[w]); c = fmaxf(c, sh_mx[w]); }
        if (a <= c) {
            atomicMax(slots + 2 * l + 0, ~enc_ord(a));
            atomicMax(slots + 2 * l + 1, enc_ord(c));
        }
        // one-launch correction: the steps of THIS launch wait for their layer's blocks.  The two merges above are performed (s_waitcnt)
        // before the counter moves -- plain device-scope atomics: a RELEASE here is a write-back of the whole L2 per block, an ACQUIRE
        // in the waiters' polls an invalidation per poll, and the correction of a batch took 0.98 instead of 0.41 ms with them.
        if (arrive) {
            __builtin_amdgcn_s_waitcnt(0);
            atomicAdd(arrive + l, 1u);
        }
    }
}

// (the workgroups behind the `n_mm_blocks` min/max ones refresh the cached ReLU moments of every BN that a step reads through a
// ReLU from the current proxies -- bc_cache_init_block: until round 4 a launch of its own)
__global__ __launch_bounds__(kBlock) void bc_minmax_kernel(const BcLayerDev* __restrict__ layers,
                                                           const int32_t* __restrict__ block_begin, int n_layers,
                                                           uint32_t* __restrict__ slots, int n_mm_blocks,
                                                           const BcCacheSeg* __restrict__ segs, int n_segs, int cache_total, int mm_chunk) {
    if ((int)blockIdx.x >= n_mm_blocks) { bc_cache_init_block(segs, n_segs, cache_total, (int)blockIdx.x - n_mm_blocks, false); return; }
    bc_minmax_block(layers, block_begin, n_layers, slots, mm_chunk, (int)blockIdx.x, nullptr);
}

// non-temporal hint on the 16-byte loads (1) / stores (2) of bc_quant_error_kernel (stores only: -1.5 % of a batch's
// bias-correction time; both: the same within noise)
#ifndef DFQ_BC_NT
#define DFQ_BC_NT 2
#endif
constexpr int kBcNt = DFQ_BC_NT;

// one thread per (o, i) pair: sequential float32 sum over kH*kW of (Q(w) - w)
__global__ __launch_bounds__(kBlock) void bc_quant_error_kernel(const BcLayerDev* __restrict__ layers,
                                                                const int32_t* __restrict__ block_begin,
                                                                int n_layers, const uint32_t* __restrict__ slots,
                                                                int num_bits, int symmetric) {
    const int l = bc_find(block_begin, n_layers, blockIdx.x);
    const BcLayerDev L = layers[l];
    const QParams p = qparams_double((double)slot_min(slots[2 * l + 0]), (double)slot_max(slots[2 * l + 1]),
                                     num_bits, symmetric);
    if (L.khkw == 1) {
        // 1x1 / linear layers (most of the weights): eps has the weight's shape, four pairs per thread as one
        // 16-byte load and one 16-byte store when the pointers allow it
        const int64_t pair = ((int64_t)(blockIdx.x - block_begin[l]) * kBlock + threadIdx.x) * kQePairs;
        if (pair >= L.pairs) return;
        float code;
        if (pair + kQePairs <= L.pairs && ((((uintptr_t)L.w) | ((uintptr_t)L.eps)) & 15u) == 0) {
            const fvec4 v = (kBcNt & 1) ? DFQ_NT_LOAD((const fvec4*)(L.w + pair)) : *(const fvec4*)(L.w + pair);
            fvec4 d;
#pragma unroll
            for (int k = 0; k < 4; ++k) d[k] = 0.0f + (fake_quant_one(v[k], p, &code) - v[k]);
            if (kBcNt & 2) DFQ_NT_STORE(d, (fvec4*)(L.eps + pair));
            else *(fvec4*)(L.eps + pair) = d;
        } else {
            for (int64_t q = pair; q < pair + kQePairs && q < L.pairs; ++q) {
                const float v = L.w[q];
                L.eps[q] = 0.0f + (fake_quant_one(v, p, &code) - v);
            }
        }
        return;
    }
    const int64_t pair = (int64_t)(blockIdx.x - block_begin[l]) * kBlock + threadIdx.x;
    if (pair >= L.pairs) return;
    const float* w = L.w + pair * L.khkw;
    float acc = 0.0f;
    for (int k = 0; k < L.khkw; ++k) {
        float code;
        const float v = w[k];
        const float d = fake_quant_one(v, p, &code) - v;
        acc = acc + d;
    }
    L.eps[pair] = acc;
}

// dfq.py:182-184: gamma*pdf(-beta/gamma) + beta*(1 - cdf(-beta/gamma)), clipped at 0 (NaN stays NaN).
// DFQ_BC_F32_MOMENT=1 (tuning): pdf / cdf from the float32 expf / erfcf instead of float64 exp / erf rounded to float32 (what
// scipy gives the reference).  They differ by a few float32 ulps (<= 5e-7 absolute on a moment of order one, far inside the
// 1e-5 contract of bias correction) and cost a fifth of the instructions of a step's per-row tail -- which bought 4.5 % of the
// single-network correction (measured: 0.287 -> 0.274 ms for the five stream operations of a MobileNetV2's pass): the tail is
// not what a step's 4 us are made of (the hand-over is), so the default stays the reference's float64 recipe.
#ifndef DFQ_BC_F32_MOMENT
#define DFQ_BC_F32_MOMENT 0
#endif
// Measurement builds only (results wrong by construction; tools/gpu_r05_bc_ablate.sh): what a chain position's ~3.6 us are made
// of.  1: no ReLU moment (pdf / cdf), 2: no matvec, 4: polls take whatever the slot holds (no dependency: the chain's
// positions run side by side -- what is left is launch, weight stream and quantiser), 8: no quantiser, 16: no second barrier pair
#ifndef DFQ_BC_ABLATE
#define DFQ_BC_ABLATE 0
#endif
// -DDFQ_BC_TRACE=1 (measurement builds; tools/bc_trace.py): thread 0 of every workgroup of the one-launch chain leaves five
// timestamps (wall_clock64: 100 MHz) -- entry, weights quantised (in front of the wait), expectation assembled, matvec done,
// tail done -- and its step index; dfq_bc_debug_trace copies them out.
#ifndef DFQ_BC_TRACE
#define DFQ_BC_TRACE 0
#endif
// 1: the batch body also settles each slot's place in sh_E and its predicate before the wait (81 instead of 72 VGPRs: the batch of
// 32 measured 0.416 against 0.400 ms over three alternating rounds, tools/gpu_r05_bench_ab.sh -- residency ahead of the chain's
// front is worth more to a batch than a shorter row sum)
#ifndef DFQ_BC_BATCH_HOIST
#define DFQ_BC_BATCH_HOIST 0
#endif
__device__ __forceinline__ float relu_mean(float w, float b) {
    if (DFQ_BC_ABLATE & 1) return b + w;
    const float t = (-b) / w;
    float pdf, cdf;
    if (DFQ_BC_F32_MOMENT) {
        pdf = expf(-(t * t) * 0.5f) * 0.3989422804014327f;
        cdf = 0.5f * erfcf(-t * 0.70710678118654752f);
    } else {
        normal_pdf_cdf(t, pdf, cdf);
    }
    const float a = w * pdf;
    const float one_m = 1.0f - cdf;
    const float c = b * one_m;
    float e = a + c;
    if (e < 0.0f) e = 0.0f;        // expect[expect < 0] = 0; NaN stays NaN
    return e;
}

// ReLU moments of every cached BN from the current beta~ (once per run, all BNs: the trailing workgroups of the min/max launch)
__device__ __forceinline__ void bc_cache_init_block(const BcCacheSeg* __restrict__ segs, int n_segs, int total, int block, bool device_scope) {
    const int i = block * kBlock + threadIdx.x;
    if (i >= total) return;
    int lo = 0, hi = n_segs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].begin <= i) lo = mid; else hi = mid - 1;
    }
    const BcCacheSeg sg = segs[lo];
    const int c = i - sg.begin;
    const float m = relu_mean(sg.fw[c], sg.fb[c]);
    if (device_scope) __hip_atomic_store((uint32_t*)(sg.cache + c), __float_as_uint(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // read by workgroups of the same launch
    else sg.cache[c] = m;
}

constexpr int kStepWords = (int)(sizeof(BcStepDev) / 4);
static_assert(sizeof(BcStepDev) % 4 == 0 && kStepWords <= 2 * kWave, "step descriptor must fit two wave-wide loads");

__device__ __forceinline__ int bc_small_div(int a, int b) {     // exact for 0 <= a < 2^20 (see dfq_le.hip)
    return (int)(((float)a + 0.5f) * __builtin_amdgcn_rcpf((float)b));
}

// One launch = the j-th correction step of every network of the plan (grid.y = networks; a single
// network passes its descriptor by value).  Batched descriptors live in a table and are fetched with
// two wave-wide loads + v_readlane, like the equalisation kernel's.
//
// Work split (chosen by the plan, carried in the descriptor): `lanes` = 2^lg_lanes lanes share an
// output row (64 for rows of >= 64 inputs, fewer for narrow / depthwise rows, so one wave covers
// 64/lanes rows per register slot); a row longer than `lanes` takes `chunks` slots; a wave owns
// rows_per_block/4 consecutive rows, all of whose eps values are in registers (<= kBcRegs slots)
// before the expectation vector is assembled.  The per-row tail -- bias update, next BN's beta~ and the
// float64 ReLU moment behind it -- runs one row per THREAD after a barrier, so its long dependent
// chain is paid once per workgroup instead of once per row.
// One workgroup of a chain launch (the whole correction chain of every network in ONE launch): which step, which
// workgroup of it, which counter to wait for (the previous step of the same network) and which to bump.
struct BcChainRef {
    int32_t step;          // index into the launch-major step table
    int32_t block;         // workgroup inside the step
    int32_t wait_idx;      // counter of the step this one depends on, or -1
    int32_t wait_blocks;   // its workgroups
};
constexpr int kBcDepStride = 32;        // one counter per 128-byte line

// values that another workgroup of the SAME launch may have written (a BN's beta~ and its cached ReLU moment):
// device-scope accesses, see dfq_le.hip / tools/litmus
__device__ __forceinline__ float ld_shared_f32(const float* p) {
    return __uint_as_float(__hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_shared_f32(float* p, float v) {
    __hip_atomic_store((uint32_t*)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Two hand-over protocols inside the one-launch chain.  Counters (tags == null): a step's workgroups wait until the counter
// of the previous step of their network has reached its number of workgroups, then read beta~ / the ReLU moment with
// device-scope loads; a producer makes its stores visible (s_waitcnt 0, barrier) and bumps its counter -- per step: store
// -> ack -> barrier -> atomic -> poll -> barrier -> load, about five dependent trips through the memory system.
// Tagged values (tags != null, the default): every BN channel a step of the plan rewrites has two 64-bit slots {run epoch :
// float bits} for beta~ and for its ReLU moment; the producing thread writes them with ONE 64-bit device-scope store each,
// a consumer thread polls exactly the slots it needs until they carry this run's epoch -- store -> poll, and no counter,
// barrier or wait for the acknowledgement on the producing side.  A step then depends on the steps that produce its sources,
// not on its predecessor in the list.
struct BcDep {            // null counters: every step is its own launch (dependencies are kernel boundaries)
    uint32_t* counters;
    uint32_t* err;
    int32_t wait_idx, wait_blocks, bump_idx, pad;
    unsigned long long* tags;
    uint32_t epoch;
    int32_t symmetric;    // dfq.py:173 `signed`: the quantiser of the row sums
    int32_t spin_limit;   // polls after which a wait is abandoned (DFQ_SPIN_LIMIT)
    int32_t mm_off;       // words from a step's `mm` to the (min, max) slots of THIS run (the slots exist in two parities, below)
    const uint32_t* mm_arrive;   // one-launch correction: per layer, the min/max blocks of THIS launch that have merged; else null
    const uint32_t* cache_arrive;   // ... and the blocks that have refreshed the cached moments of the never-rewritten BNs
    int32_t cache_blocks;
};
// The error word.  Counter protocol: cleared by the run's clear launch, any non-zero value = a wait of this run was abandoned.
// Tagged protocol (no clear launch): it holds the EPOCH of the latest run in which a wait was abandoned (atomicMax: epochs
// only grow), so an old failure is no failure of this run.
__device__ __forceinline__ bool bc_err_raised(const BcDep& dep) {
    const uint32_t e = __hip_atomic_load(dep.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return dep.tags ? (e == dep.epoch) : (e != 0u);
}
__device__ __forceinline__ void bc_raise_err(const BcDep& dep) { atomicMax(dep.err, dep.tags ? dep.epoch : 1u); }

// kOneGroup: every step of the launch gives a wave ONE group of rows (n_slots == chunks: the split of a single network, where
// the chain is latency and nothing else) -- the body then spends two dozen registers more on settling the row sum's operands
// before the wait.  A batch keeps the leaner body: its chain is bound by how many workgroups are resident ahead of the front
// (80 against 119 VGPRs: the batch of 32 measured 0.46 against 0.49 ms).
template <int kExp, bool kOneGroup>
__device__ __forceinline__ void bc_step_body(const BcStepDev& st, int blk, const BcSourceDev* __restrict__ sources,
                                             const BcFoldDev* __restrict__ folds, const BcDep& dep, float* sh_E, float* sh_corr, int* sh_flag) {
    const int tid = threadIdx.x;
    const int lane = tid % kWave;
    const int wave = tid / kWave;
    const bool chained = dep.counters != nullptr;
    const int rpb = st.rows_per_block;
    const int in = st.in_per_group;
    const int lanes = 1 << st.lg_lanes;
    const int rps = kWave >> st.lg_lanes;                          // rows per register slot
    const int chunks = st.chunks;
    const int rw = rpb / kRowsPerBlock;                            // rows of this wave
    const int sub = lane >> st.lg_lanes;                           // row inside a slot
    const int ln = lane & (lanes - 1);
    const int row0 = blk * rpb + wave * rw;
    const int n_slots = min(kBcRegs, ((rw + rps - 1) >> (6 - st.lg_lanes)) * chunks);
    BC_STAMP(0);
    BC_STAMP_VAL(7, ((long long)dep.bump_idx << 32) | (uint32_t)blk);
    // ---- this wave's quant-error row sums go into registers first: eps[o, i] = sum_k (Q(w) - w) (dfq.py:216-219, 8 bit,
    //      per-tensor range from the min/max launch; sequential float32 sum over k from 0.0f like the reference's .sum(-1)).
    //      The weights are requested before anything else and the quantiser runs while the expectation's sources arrive ----
    const int khkw = st.khkw;
    float ev[kBcRegs];
    {
        int rg = 0, c = 0;
        if (khkw == 1) {
#pragma unroll
            for (int u = 0; u < kBcRegs; ++u) {
                if (u < n_slots) {
                    const int row = min(row0 + rg * rps + sub, st.out_ch - 1);
                    const int col = min(c * lanes + ln, in - 1);
                    ev[u] = st.w[(int64_t)row * in + col];
                    if (++c == chunks) { c = 0; ++rg; }
                }
            }
        }
    }
    const bool tagged = chained && dep.tags != nullptr;
    if (tagged) {
        if (tid == 0) *sh_flag = 1;
        __syncthreads();
    }
    // One-launch correction (dep.mm_arrive): the layer's (min, max) are formed by min/max blocks of THIS launch, which lie in
    // front of this workgroup in the grid; wait until all of them have merged (the weights requested above arrive meanwhile)
    // and read the slots past the caches of this XCD.
    auto mm_wait = [&](const uint32_t* counter, int blocks) {
        long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)blocks) {
            __builtin_amdgcn_s_sleep(2);
            ++spins;
            if (spins > dep.spin_limit || ((spins & 255) == 0 && bc_err_raised(dep))) {
                bc_raise_err(dep);
                *sh_flag = 0;
                break;
            }
        }
    };
    auto mm_slot = [&](const uint32_t* q) {
        return dep.mm_arrive ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
    };
    if (dep.mm_arrive) {
        if (tid == 0) {
            mm_wait(dep.mm_arrive + st.mm_index, st.mm_blocks);
            if (st.wait_cache) mm_wait(dep.cache_arrive, dep.cache_blocks);
        }
        __syncthreads();
        if (*sh_flag == 0) return;           // abandoned: nothing has been stored
    }
    const QParams qp = qparams_double((double)slot_min(mm_slot(st.mm + dep.mm_off + 0)), (double)slot_max(mm_slot(st.mm + dep.mm_off + 1)), 8, dep.symmetric);
    // Everything of the matvec that does not depend on the expectation is settled here, before the wait: where in sh_E each
    // slot's factor will lie (a byte offset < 64 Ki, two per register) and which slots take part at all (a slot outside the
    // row or the layer holds eps = 0: its product is a zero of either sign, and adding one to a sum that started at +0.0
    // changes no bit -- the row it would poison with 0 x inf is a row that is discarded, or one the clamped element belongs to).
    // Until round 5 the index arithmetic, the predicate and the LDS read of every slot sat between the arrival of the
    // expectation and the row's sum, one slot after the other, and the compiler had sunk the quantiser there too: a third
    // of a MobileNetV2's dependent chain (profiles/r05_bc_chain.txt).
    const int num_group = st.expect_len / in;
    const int step_o = st.out_ch / num_group;
    uint32_t eo[kBcRegs / 2];
#pragma unroll
    for (int u = 0; u < kBcRegs / 2; ++u) eo[u] = 0u;
    {
        int rg = 0, c = 0;
        uint32_t off0 = 0u;
#pragma unroll
        for (int u = 0; u < kBcRegs; ++u) {
            if (u < n_slots) {
                float code;
                const int r_local = rg * rps + sub;
                const int row_u = row0 + r_local;
                const int col_u = c * lanes + ln;
                const bool ok = r_local < rw && row_u < st.out_ch && col_u < in;
                if (kOneGroup || DFQ_BC_BATCH_HOIST) {
                    const int g = (num_group == 1) ? 0 : bc_small_div(min(row_u, st.out_ch - 1), step_o);
                    const uint32_t off = (uint32_t)(g * in + min(col_u, in - 1)) * 4u;
                    if (u == 0) off0 = off;
                    eo[u >> 1] |= off << (16 * (u & 1));
                }
                if (khkw == 1) {
                    const float v = ev[u];
                    ev[u] = (DFQ_BC_ABLATE & 8) ? v : 0.0f + (fake_quant_one(v, qp, &code) - v);
                } else {
                    const int row = min(row_u, st.out_ch - 1);
                    const int col = min(col_u, in - 1);
                    const float* wp = st.w + ((int64_t)row * in + col) * khkw;
                    float acc = 0.0f;
                    for (int k = 0; k < khkw; ++k) {
                        const float v = wp[k];
                        acc = (DFQ_BC_ABLATE & 8) ? acc + v : acc + (fake_quant_one(v, qp, &code) - v);
                    }
                    ev[u] = acc;
                }
                if ((kOneGroup || DFQ_BC_BATCH_HOIST) && chunks <= kBcRegs && !ok) ev[u] = 0.0f;
                // the quantiser's arithmetic belongs in FRONT of the wait: without this the compiler sinks it to the first use
                // of ev[u] -- the matvec, behind the arrival of the expectation, on the chain's critical path
                asm volatile("" : "+v"(ev[u]));
                if (++c == chunks) { c = 0; ++rg; }
            } else if (kOneGroup) {
                ev[u] = 0.0f;                                    // a slot nobody owns: factor = this lane's first one
                eo[u >> 1] |= off0 << (16 * (u & 1));
            }
        }
    }
    // the per-row operands of the tail (this layer's bias, the next BN's beta~ and gamma~) do not depend on the previous
    // step either: requested now, they are back before the matvec is done
    const int o_tail = blk * rpb + tid;
    const bool tail_on = tid < rpb && o_tail < st.out_ch;
    float pre_bias = 0.0f, pre_nb = 0.0f, pre_nw = 0.0f;
    if (tail_on) {
        pre_bias = st.bias[o_tail];
        if (st.next_bn_bias) pre_nb = st.next_bn_bias[o_tail];
        if (st.next_cache) pre_nw = st.next_bn_weight[o_tail];
    }
    // the folded depthwise layer (BcFoldDev): this thread's row of it -- taps, bias, the BN behind it -- is requested now too,
    // and the row sum of its quantisation error (sequential float32 sum from 0.0f, like a step's) is ready before the tail
    const bool fold_on = st.fold >= 0 && tail_on;
    float f_eps = 0.0f, f_bias = 0.0f, f_nb = 0.0f, f_nw = 0.0f;
    if (fold_on) {
        const BcFoldDev F = folds[st.fold];
        float taps[kFoldTaps];
#pragma unroll
        for (int k = 0; k < kFoldTaps; ++k) taps[k] = F.w[(int64_t)o_tail * F.khkw + min(k, F.khkw - 1)];
        f_bias = F.bias[o_tail];
        if (F.next_bn_bias) f_nb = F.next_bn_bias[o_tail];
        if (F.next_cache) f_nw = F.next_bn_weight[o_tail];
        if (dep.mm_arrive) mm_wait(dep.mm_arrive + F.mm_index, F.mm_blocks);      // (the few threads that own a row of it; an abandoned wait is noticed behind the merges)
        const QParams fq = qparams_double((double)slot_min(mm_slot(F.mm + dep.mm_off + 0)), (double)slot_max(mm_slot(F.mm + dep.mm_off + 1)), 8, dep.symmetric);
        float acc = 0.0f, code;
#pragma unroll
        for (int k = 0; k < kFoldTaps; ++k)
            if (k < F.khkw) acc = acc + (fake_quant_one(taps[k], fq, &code) - taps[k]);
        f_eps = acc;
        asm volatile("" : "+v"(f_eps));                           // (in front of the wait, like ev[])
    }
    BC_STAMP(1);
    if (chained && !tagged && dep.wait_idx >= 0) {
        // the previous layer's correction feeds this expectation: wait for all its workgroups (the eps values
        // requested above arrive meanwhile)
        if (tid == 0) {
            long spins = 0;
            int ok = 1;
            while (__hip_atomic_load(dep.counters + (int64_t)dep.wait_idx * kBcDepStride, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)dep.wait_blocks) {
                __builtin_amdgcn_s_sleep(2);               // few waiters here (one step's workgroups): poll briskly
                ++spins;
                if (spins > dep.spin_limit ||
                    ((spins & 255) == 0 && bc_err_raised(dep))) {
                    bc_raise_err(dep);
                    ok = 0;
                    break;
                }
            }
            *sh_flag = ok;
        }
        __syncthreads();
        // abandoned wait: leave before anything is stored (no bias is corrected with a stale expectation) and without
        // bumping this step's counter -- the later steps give up at once through `err`, the status call reports it
        if (*sh_flag == 0) return;
    }
    // ---- E[x]: first source assigns, 'cat' appends, anything else adds (dfq.py:229-270) ----
    int cur_len = 0;
    auto merge_source = [&](const BcSourceDev& s, int m) {
        const bool assign = (m == 0) || (s.concat != 0);
        const int base = (m == 0) ? 0 : (s.concat ? cur_len : 0);
        const float* val = s.relu ? s.cache : s.fb;      // E[ReLU(N(beta, gamma^2))] or beta
        const bool poll = tagged && s.tag_off >= 0;      // rewritten by an earlier step of this launch: wait for THIS run's value
        const unsigned long long* slot = poll ? dep.tags + 2 * (int64_t)s.tag_off + (s.relu ? 1 : 0) : nullptr;
        if (!kOneGroup && poll) {
            // The batch body asks for a thread's slots four at a time and for the stale ones again together.  The workgroups of a
            // batch's large layers are dispatched when the steps they depend on have long finished: every slot is fresh, and
            // a source of 1280 channels was five round trips to the memory side one after the other in a workgroup whose whole
            // life is ten microseconds (profiles/r05_bc_chain.txt).  (A single network's chain, where the poll IS the hand-over,
            // measured 4 % slower with it: 0.130 against 0.125 ms.)
            constexpr int kPollBatch = 4;
            for (int i0 = tid; i0 < s.channels; i0 += kBlock * kPollBatch) {
                unsigned long long w[kPollBatch];
                bool stale = false;
#pragma unroll
                for (int j = 0; j < kPollBatch; ++j) {
                    const int i = i0 + j * kBlock;
                    w[j] = (i < s.channels) ? __hip_atomic_load(slot + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                            : ((unsigned long long)dep.epoch << 32);
                }
#pragma unroll
                for (int j = 0; j < kPollBatch; ++j) stale |= !(DFQ_BC_ABLATE & 4) && (uint32_t)(w[j] >> 32) != dep.epoch;
                long spins = 0;
                while (stale) {
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
                    if (spins > dep.spin_limit ||
                        ((spins & 255) == 0 && bc_err_raised(dep))) {
                        bc_raise_err(dep);
                        *sh_flag = 0;                      // (every thread that gives up writes the same value)
                        break;
                    }
                    stale = false;
#pragma unroll
                    for (int j = 0; j < kPollBatch; ++j) {
                        const int i = i0 + j * kBlock;
                        if ((uint32_t)(w[j] >> 32) != dep.epoch) {
                            w[j] = __hip_atomic_load(slot + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            stale |= (uint32_t)(w[j] >> 32) != dep.epoch;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < kPollBatch; ++j) {
                    const int i = i0 + j * kBlock;
                    if (i < s.channels) {
                        const float e = __uint_as_float((uint32_t)w[j]);
                        if (assign) sh_E[base + i] = e;
                        else sh_E[i] = sh_E[i] + e;
                    }
                }
            }
        } else
        for (int i = tid; i < s.channels; i += kBlock) {
            float e;
            if (poll) {
                unsigned long long w = __hip_atomic_load(slot + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                long spins = 0;
                while (!(DFQ_BC_ABLATE & 4) && (uint32_t)(w >> 32) != dep.epoch) {
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
                    if (spins > dep.spin_limit ||
                        ((spins & 255) == 0 && bc_err_raised(dep))) {
                        bc_raise_err(dep);
                        *sh_flag = 0;                      // (every thread that gives up writes the same value)
                        break;
                    }
                    w = __hip_atomic_load(slot + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                e = __uint_as_float((uint32_t)w);
            } else {
                e = (chained && (!tagged || dep.mm_arrive)) ? ld_shared_f32(val + i) : val[i];
            }
            if (assign) sh_E[base + i] = e;
            else sh_E[i] = sh_E[i] + e;
        }
        cur_len = (m == 0) ? s.channels : (s.concat ? cur_len + s.channels : cur_len);
        __syncthreads();
    };
    if (st.inline_sources) {
        // compile-time indices only: `st.src[m]` with a run-time m would force the whole descriptor into
        // scratch memory (it did: 336 B of private segment per lane, written by every wave of every workgroup)
#pragma unroll
        for (int m = 0; m < kStepSources; ++m)
            if (m < st.source_count) merge_source(st.src[m], m);
    } else {
        for (int m = 0; m < st.source_count; ++m) {
            const BcSourceDev s = sources[st.source_begin + m];
            merge_source(s, m);
        }
    }
    // (every merge ends with a barrier) an abandoned poll: leave before anything is stored, like an abandoned counter wait
    if (tagged && *sh_flag == 0) return;
    BC_STAMP(2);
    // ---- grouped matvec (dfq.py:281-287), float64 accumulation rounded once per row ----
    if (chunks > kBcRegs) {
        // very long rows (> 1536 inputs): one row per wave, the tail streams from memory
        const int o = min(row0, st.out_ch - 1);
        const float* ex = sh_E + bc_small_div(o, step_o) * in;
        const float* wr = st.w + (int64_t)o * in * khkw;
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < kBcRegs; ++u) {
            const int i = lane + u * kWave;
            acc += (i < in) ? (double)ev[u] * (double)ex[i] : 0.0;
        }
        for (int i = lane + kBcRegs * kWave; i < in; i += kWave) {
            float er = 0.0f, code;
            for (int k = 0; k < khkw; ++k) {
                const float v = wr[(int64_t)i * khkw + k];
                er = er + (fake_quant_one(v, qp, &code) - v);
            }
            acc += (double)er * (double)ex[i];
        }
        acc = wave_sum(acc);
        if (lane == 0) sh_corr[wave * rw] = (float)acc;
    } else if (DFQ_BC_ABLATE & 2) {
        if (tid < rpb) sh_corr[tid] = ev[0] * sh_E[0];
    } else {
        if (kOneGroup) {
            // the factors of all slots are requested at once (eight, sixteen or all twenty-four), then summed in slot order, in
            // groups of eight without a decision per slot (a slot nobody owns adds a zero, above), one butterfly, one store:
            // 0.35 us where the general loop below -- which computes each slot's place, asks whether the slot exists and
            // whether it ends a group, 24 times -- took 1.0 us even for a single slot (profiles/r05_bc_chain.txt).  Same
            // additions in the same order.
            float e[kBcRegs];
            auto factor = [&](int u) { return *(const float*)((const char*)sh_E + ((eo[u >> 1] >> (16 * (u & 1))) & 0xffffu)); };
#pragma unroll
            for (int u = 0; u < 8; ++u) e[u] = factor(u);
            if (n_slots > 8) {
#pragma unroll
                for (int u = 8; u < 16; ++u) e[u] = factor(u);
            }
            if (n_slots > 16) {
#pragma unroll
                for (int u = 16; u < kBcRegs; ++u) e[u] = factor(u);
            }
            double acc = 0.0;
            BC_STAMP(5);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (double)ev[u] * (double)e[u];
            if (n_slots > 8) {
#pragma unroll
                for (int u = 8; u < 16; ++u) acc += (double)ev[u] * (double)e[u];
            }
            if (n_slots > 16) {
#pragma unroll
                for (int u = 16; u < kBcRegs; ++u) acc += (double)ev[u] * (double)e[u];
            }
            // segmented butterfly over the `lanes` lanes of a row, high mask first (register-file moves, dfq_common.hpp)
            if (lanes > 32) xor_lane_add<32>(acc);
            if (lanes > 16) xor_lane_add<16>(acc);
            if (lanes > 8) xor_lane_add<8>(acc);
            if (lanes > 4) xor_lane_add<4>(acc);
            if (lanes > 2) xor_lane_add<2>(acc);
            if (lanes > 1) xor_lane_add<1>(acc);
            if (ln == 0 && sub < rw) sh_corr[wave * rw + sub] = (float)acc;
        } else {
            int rg = 0, c = 0;
            double acc = 0.0;
#pragma unroll
            for (int u = 0; u < kBcRegs; ++u) {
                if (u < n_slots) {
                    const int r_local = rg * rps + sub;
                    if (DFQ_BC_BATCH_HOIST) {
                        const float e = *(const float*)((const char*)sh_E + ((eo[u >> 1] >> (16 * (u & 1))) & 0xffffu));
                        acc += (double)ev[u] * (double)e;
                    } else {
                        const int row = row0 + r_local;
                        const int col = c * lanes + ln;
                        const bool ok = r_local < rw && row < st.out_ch && col < in;
                        const int g = bc_small_div(min(row, st.out_ch - 1), step_o);
                        const float e = sh_E[g * in + min(col, in - 1)];
                        acc += ok ? (double)ev[u] * (double)e : 0.0;
                    }
                    if (++c == chunks) {                                   // the rows of this slot group are complete
                        if (lanes > 32) xor_lane_add<32>(acc);
                        if (lanes > 16) xor_lane_add<16>(acc);
                        if (lanes > 8) xor_lane_add<8>(acc);
                        if (lanes > 4) xor_lane_add<4>(acc);
                        if (lanes > 2) xor_lane_add<2>(acc);
                        if (lanes > 1) xor_lane_add<1>(acc);
                        if (ln == 0 && r_local < rw) sh_corr[wave * rw + r_local] = (float)acc;
                        acc = 0.0; c = 0; ++rg;
                    }
                }
            }
        }
    }
    BC_STAMP(6);
    __syncthreads();
    BC_STAMP(3);
    // ---- one row per thread: dfq.py:290-293 and the refreshed ReLU moment of the next BN ----
    const int o = o_tail;
    // publication of one row's update (this step's, then the folded step's): the three hand-over protocols
    auto publish = [&](float* bn_bias, float* cache, int tag_off, float nb, float moment) {
        if (tagged) {
            bn_bias[o] = nb;                                      // the final state; consumers inside the launch read the slots
            if (cache) cache[o] = moment;
            if (tag_off >= 0) {
                unsigned long long* slot = dep.tags + 2 * ((int64_t)tag_off + o);
                const unsigned long long hi = (unsigned long long)dep.epoch << 32;
                __hip_atomic_store(slot + 0, hi | __float_as_uint(nb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cache)
                    __hip_atomic_store(slot + 1, hi | __float_as_uint(moment), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (chained) {
            st_shared_f32(bn_bias + o, nb);
            if (cache) st_shared_f32(cache + o, moment);
        } else {
            bn_bias[o] = nb;
            if (cache) cache[o] = moment;
        }
    };
    if (tail_on) {
        const float corr = sh_corr[tid];
        const float neg = -corr;
        st.corr[o] = corr;
        st.bias[o] = pre_bias + neg;                              // dfq.py:292
        float nb = 0.0f, moment = 0.0f;
        if (st.next_bn_bias) {
            nb = pre_nb + neg;                                    // dfq.py:204-206, 293 (a BN's beta~ changes once)
            moment = st.next_cache ? relu_mean(pre_nw, nb) : 0.0f;
            publish(st.next_bn_bias, st.next_cache, st.next_tag_off, nb, moment);
        }
        if (fold_on) {
            // the folded depthwise step, channel o: E[o] is what this thread has just produced; one input per group, so the
            // "matvec" of dfq.py:281-287 is one product, accumulated in float64 and rounded once like a step's row
            const BcFoldDev F = folds[st.fold];
            const float e = F.src_relu ? moment : nb;
            const float fcorr = (float)((double)f_eps * (double)e);
            const float fneg = -fcorr;
            F.corr[o] = fcorr;
            F.bias[o] = f_bias + fneg;
            if (F.next_bn_bias) {
                const float fnb = f_nb + fneg;
                const float fmoment = F.next_cache ? relu_mean(f_nw, fnb) : 0.0f;
                publish(F.next_bn_bias, F.next_cache, F.next_tag_off, fnb, fmoment);
            }
        }
    }
    BC_STAMP(4);
    if (chained && !tagged) {
        __builtin_amdgcn_s_waitcnt(0);                            // the stores above have been performed
        __syncthreads();
        if (tid == 0) atomicAdd(dep.counters + (int64_t)dep.bump_idx * kBcDepStride, 1u);
    }
}

// the two wave-wide loads + v_readlane that fetch a step descriptor from the table
__device__ __forceinline__ void bc_load_step(const BcStepDev* __restrict__ entry, uint32_t (&u)[kStepWords]) {
    const int lane = threadIdx.x % kWave;
    const guint* src = (const guint*)entry;
    const uint32_t w0 = src[min(lane, kStepWords - 1)];
    const uint32_t w1 = src[min(kWave + lane, kStepWords - 1)];
#pragma unroll
    for (int i = 0; i < kStepWords; ++i)
        u[i] = (i < kWave) ? __builtin_amdgcn_readlane(w0, i % kWave) : __builtin_amdgcn_readlane(w1, i % kWave);
}

// one launch per chain position (DFQ_BC_MERGED=0): grid (workgroups of the largest step, networks)
template <int kExp>
__global__ __launch_bounds__(kBlock) void bc_step_kernel(BcStepDev st_inline, const BcStepDev* __restrict__ table,
                                                         const BcSourceDev* __restrict__ sources, const BcFoldDev* __restrict__ folds, int symmetric) {
    __shared__ float sh_E[kExp];
    __shared__ float sh_corr[kBlock];
    __shared__ int sh_flag;
    union { BcStepDev st; uint32_t u[kStepWords]; } desc;
    if (table) bc_load_step(table + blockIdx.y, desc.u);
    else desc.st = st_inline;
    if ((int)blockIdx.x * desc.st.rows_per_block >= desc.st.out_ch) return;     // grid.x is sized for the largest step of the launch
    bc_step_body<kExp, false>(desc.st, blockIdx.x, sources, folds, BcDep{nullptr, nullptr, -1, 0, -1, 0, nullptr, 0u, symmetric, 0, 0, nullptr, nullptr, 0}, sh_E, sh_corr, &sh_flag);
}

// the whole chain of every network in one launch: 1-D grid over (step, workgroup) in chain order; a workgroup waits
// for the previous step of its network (lower indices only -> no deadlock, see dfq_le.hip)
// One-launch correction (round 5): the per-tensor min/max blocks are workgroups of the chain launch.  The plan interleaves them with
// the chain's positions -- the blocks of the layers a position needs lie a few positions in front of it in the grid -- so a step
// waits only for blocks with lower indices (no deadlock), the large late layers are reduced while the chain's early positions
// hand over (a phase that leaves the memory system idle), and the launch boundary between the two kernels is gone.
// INVARIANT of everything that is handed over behind these arrival counters (and behind the chain's own step counters, BcDep): the
// producer writes it with device-scope atomics (the min / max merges, the cached moments) or sc1 stores and orders it before its
// arrival with s_waitcnt(0) only; the consumer polls with relaxed device-scope loads and reads the payload through mm_slot /
// ld_shared_f32 (device-scope loads).  There is NO release / acquire pair: a release would write back the XCD's whole L2 per block and
// an acquire invalidate it per poll (0.98 instead of 0.41 ms for a batch).  A PLAIN load or store added on this path would be wrong
// across XCDs -- silently, and invisibly to the CPU emulation (its atomics are sequentially consistent): keep every access on these
// helpers.  tests/test_engine_parity.py::test_one_launch_correction_under_stress repeats one-launch runs with the look-ahead at 0.
struct BcFusedMm {
    const BcLayerDev* layers;
    const int32_t* block_begin;
    uint32_t* slots;        // this run's parity
    uint32_t* arrive;       // per layer: merged blocks of this run (same parity region, zero at the start); null: min/max was its own launch
    int32_t n_layers, mm_chunk;
    const BcCacheSeg* segs; // the never-rewritten BNs read through a ReLU: their cached moments are refreshed by the launch's first blocks
    uint32_t* cache_arrive;
    int32_t n_segs, cache_total, cache_blocks, pad;
};

// (the batch body at 64 VGPRs -- amdgpu_waves_per_eu(8, 8): 17 registers spilled to scratch -- measured 0.44 against 0.40 ms)
template <int kExp, bool kOneGroup>
__global__ __launch_bounds__(kBlock) void bc_chain_kernel(const BcStepDev* __restrict__ table,
                                                          const BcChainRef* __restrict__ refs,
                                                          const BcSourceDev* __restrict__ sources, const BcFoldDev* __restrict__ folds,
                                                          uint32_t* counters, uint32_t* err, unsigned long long* tags, uint32_t epoch,
                                                          int symmetric, int spin_limit, int mm_off, uint32_t* slots_clear, int n_slots_clear,
                                                          BcFusedMm fm) {
    __shared__ float sh_E[kExp];
    __shared__ float sh_corr[kBlock];
    __shared__ int sh_flag;
    typedef int ivec4 __attribute__((vector_size(16)));
    // the tagged chain has no clear launch in front of it: its first workgroup clears the (min, max) slots the NEXT run's
    // min/max launch accumulates into (the other parity: nobody touches it in this run)
    if (blockIdx.x == 0 && slots_clear)
        for (int i = threadIdx.x; i < n_slots_clear; i += kBlock) slots_clear[i] = 0u;
    const ivec4 ref = *(const DFQ_GLOBAL_AS ivec4*)(refs + blockIdx.x);
    const int step = __builtin_amdgcn_readfirstlane(ref[0]);
    const int blk = __builtin_amdgcn_readfirstlane(ref[1]);
    if (step == -1) {                // a min/max block of the one-launch correction
        bc_minmax_block(fm.layers, fm.block_begin, fm.n_layers, fm.slots, fm.mm_chunk, blk, fm.arrive);
        return;
    }
    if (step == -2) {                // ... a block of cached ReLU moments
        bc_cache_init_block(fm.segs, fm.n_segs, fm.cache_total, blk, true);
        __builtin_amdgcn_s_waitcnt(0);              // the device-scope stores have been performed
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(fm.cache_arrive, 1u);
        return;
    }
    union { BcStepDev st; uint32_t u[kStepWords]; } desc;
    bc_load_step(table + step, desc.u);
    bc_step_body<kExp, kOneGroup>(desc.st, blk, sources, folds,
                       BcDep{counters, err, __builtin_amdgcn_readfirstlane(ref[2]), __builtin_amdgcn_readfirstlane(ref[3]), step, 0,
                             tags, epoch, symmetric, spin_limit, mm_off, fm.arrive, fm.cache_arrive, fm.cache_blocks},
                       sh_E, sh_corr, &sh_flag);
}

}  // namespace dfq

using namespace dfq;

struct dfq_bc_plan {
    dfq::DevSlab mem;                 // every device table below lives in here

    int n_steps = 0;
    int minmax_blocks = 0, qerr_blocks = 0;
    int mm_chunk = kMmChunk;               // floats per workgroup of the min/max launch
    bool one_group = false;                // every live step gives a wave one group of rows: the chain kernel's latency variant
    bool fused_mm = false;                 // tagged runs are ONE launch: the min/max blocks are workgroups of the chain launch
    BcChainRef* d_refs_fused = nullptr;    // its workgroup table (min/max blocks a few positions in front of the steps that need them)
    int chain_blocks_fused = 0;
    BcCacheSeg* d_cache_segs_launch = nullptr;   // the never-rewritten BNs among the cached ones (refreshed by blocks of the one launch)
    int n_cache_segs_launch = 0, cache_total_launch = 0;
    int64_t weight_elems = 0, eps_elems = 0;
    std::vector<BcStepDev> steps;          // host copies in the caller's order
    struct Launch { int begin, n, max_blocks, max_expect; };
    std::vector<Launch> launches;          // launch j = j-th step of every network
    std::vector<BcStepDev> launch_steps;   // launch-major copy (kernel argument by value for 1-step launches)
    BcStepDev* d_steps = nullptr;          // launch_steps on the device
    BcChainRef* d_refs = nullptr;          // workgroup table of the one-launch chain
    uint32_t* d_counters = nullptr;        // per step: finished workgroups (padded), + error flag
    unsigned long long* d_tags = nullptr;  // tagged-value slots {epoch : float} x 2 per rewritten BN channel; null: counter protocol
    uint32_t epoch = 1;                    // run counter carried by the slots; the first tagged run has epoch 2: the error word of a failed
                                           // COUNTER-protocol run is 1 and must not read as "this tagged run failed" 
    int slot_parity = 0;                   // which half of d_slots the next tagged run's min/max launch accumulates into (it is zero)
    bool last_tagged = false;              // the last run used the tagged protocol (how dfq_bc_plan_status reads the error word)
    int chain_blocks = 0, max_expect = 0;
    bool merged = true;                    // one launch for the whole chain (false: one per chain position, DFQ_BC_MERGED=0)
    std::vector<const float*> eps_ptr;
    BcLayerDev* d_layers = nullptr;
    int32_t* d_mm_begin = nullptr;
    int32_t* d_qe_begin = nullptr;
    BcSourceDev* d_sources = nullptr;
    BcFoldDev* d_folds = nullptr;          // depthwise steps folded into their predecessor's tail (see BcFoldDev)
    int n_folds = 0;
    uint32_t* d_slots = nullptr;
    float* d_eps = nullptr;                // debug (DFQ_BC_EPS=1): all eps matrices, back to back
    bool keep_eps = false;
    float* d_corr = nullptr;               // all correction vectors, back to back
    float* d_cache = nullptr;              // ReLU moments of the BNs that some step reads through a ReLU
    BcCacheSeg* d_cache_segs = nullptr;
    int n_cache_segs = 0, cache_total = 0;
    hipStream_t capture_stream = nullptr;  // private stream used only to record the graph
    hipGraphExec_t exec[2] = {nullptr, nullptr};   // recorded run, by `symmetric`
};

extern "C" {

void dfq_bc_plan_destroy(dfq_bc_plan* p) {
    if (!p) return;
    dfq::dev_quiesce();                                  // nothing in flight may still use the blocks released below
    p->mem.release();
    for (auto& e : p->exec) if (e) (void)hipGraphExecDestroy(e);
    if (p->capture_stream) (void)hipStreamDestroy(p->capture_stream);
    delete p;
}

int dfq_bc_plan_create_replicated(const dfq_layer* layers, int32_t n_layers, const dfq_bc_step* steps, int32_t n_steps,
                                  const dfq_bc_source* sources, int32_t n_sources, const void* const* bases, int32_t n_nets,
                                  dfq_bc_plan** out_plan) {
    if (!layers || n_layers <= 0 || !steps || n_steps <= 0 || !sources || n_sources <= 0 || !bases || n_nets < 1 || !out_plan)
        return fail_arg("dfq_bc_plan_create_replicated: bad argument");
    if ((int64_t)n_layers * n_nets > INT32_MAX || (int64_t)n_steps * n_nets > INT32_MAX || (int64_t)n_sources * n_nets > INT32_MAX)
        return fail_arg("dfq_bc_plan_create_replicated: too many layers");
    std::vector<dfq_layer> L((size_t)n_layers * n_nets);
    std::vector<dfq_bc_step> S((size_t)n_steps * n_nets);
    std::vector<dfq_bc_source> C((size_t)n_sources * n_nets);
    const intptr_t b0 = (intptr_t)bases[0];
    auto moved = [](const float* p, intptr_t by) { return p ? (const float*)((const char*)p + by) : nullptr; };
    for (int n = 0; n < n_nets; ++n) {
        if (!bases[n]) return fail_arg("dfq_bc_plan_create_replicated: network %d has no base address", n);
        const intptr_t by = (intptr_t)bases[n] - b0;
        for (int l = 0; l < n_layers; ++l) {
            dfq_layer& d = L[(size_t)n * n_layers + l];
            d = layers[l];
            d.weight = (float*)moved(d.weight, by);
            d.bias = (float*)moved(d.bias, by);
        }
        for (int k = 0; k < n_steps; ++k) {
            dfq_bc_step& d = S[(size_t)n * n_steps + k];
            d = steps[k];
            if (d.net != 0) return fail_arg("dfq_bc_plan_create_replicated: the step table must describe ONE network (step %d)", k);
            d.layer += n * n_layers;
            d.source_begin += n * n_sources;
            d.next_bn_bias = (float*)moved(d.next_bn_bias, by);
            d.net = n;
        }
        for (int k = 0; k < n_sources; ++k) {
            dfq_bc_source& d = C[(size_t)n * n_sources + k];
            d = sources[k];
            d.fake_weight = moved(d.fake_weight, by);
            d.fake_bias = moved(d.fake_bias, by);
        }
    }
    return dfq_bc_plan_create(L.data(), n_layers * n_nets, S.data(), n_steps * n_nets, C.data(), n_sources * n_nets, out_plan);
}

int dfq_bc_plan_create(const dfq_layer* layers, int32_t n_layers, const dfq_bc_step* steps, int32_t n_steps,
                       const dfq_bc_source* sources, int32_t n_sources, dfq_bc_plan** out_plan) {
    if (!layers || n_layers <= 0 || !steps || n_steps <= 0 || !sources || n_sources <= 0 || !out_plan)
        return fail_arg("dfq_bc_plan_create: bad argument");
    PlanTimer timer("dfq_bc_plan_create");
    // ---- validate & size ----
    int64_t eps_total = 0, eps_true = 0, corr_total = 0, mm_blocks = 0, qe_blocks = 0;
    // floats per workgroup of the min/max launch (a multiple of 4 * kBlock): DFQ_BC_MM_CHUNK, tuning
    int mm_chunk = kMmChunk;
    if (const char* ce = getenv("DFQ_BC_MM_CHUNK")) { const int v = atoi(ce); if (v >= 4 * kBlock && v % (4 * kBlock) == 0) mm_chunk = v; }
    std::vector<int> expect_len(n_steps, 0);
    for (int s = 0; s < n_steps; ++s) {
        const dfq_bc_step& st = steps[s];
        if (st.layer < 0 || st.layer >= n_layers) return fail_arg("dfq_bc_plan_create: step %d: bad layer index", s);
        if (st.net < 0 || (s > 0 && st.net < steps[s - 1].net))
            return fail_arg("dfq_bc_plan_create: step %d: steps must be listed network by network", s);
        const dfq_layer& L = layers[st.layer];
        if (!L.weight || !L.bias) return fail_arg("dfq_bc_plan_create: step %d: layer needs weight and bias", s);
        if (L.out_ch <= 0 || L.in_per_group <= 0 || L.khkw <= 0 || L.out_ch >= (1 << 20) || L.in_per_group >= (1 << 20))
            return fail_arg("dfq_bc_plan_create: step %d: layer dimensions must be in [1, 2^20)", s);
        if (st.source_count <= 0 || st.source_begin < 0 || st.source_begin + st.source_count > n_sources)
            return fail_arg("dfq_bc_plan_create: step %d: bad source range", s);
        int len = 0;
        for (int m = 0; m < st.source_count; ++m) {
            const dfq_bc_source& src = sources[st.source_begin + m];
            if (!src.fake_bias || (src.relu && !src.fake_weight) || src.channels <= 0)
                return fail_arg("dfq_bc_plan_create: step %d source %d: null BN proxy", s, m);
            if (m == 0) len = src.channels;
            else if (src.concat) len += src.channels;
            else if (src.channels != len)
                return fail_arg("dfq_bc_plan_create: step %d source %d: add of %d channels onto %d", s, m, src.channels, len);
        }
        if (len > kExpectMax) return fail_arg("dfq_bc_plan_create: step %d: expectation of %d channels > %d", s, len, kExpectMax);
        if (len % L.in_per_group != 0 || L.out_ch % (len / L.in_per_group) != 0)
            return fail_arg("dfq_bc_plan_create: step %d: expectation length %d does not fit I/g=%d, O=%d", s, len,
                            L.in_per_group, L.out_ch);
        expect_len[s] = len;
        const int64_t pairs = (int64_t)L.out_ch * L.in_per_group;
        eps_total += (pairs + 3) & ~(int64_t)3;      // every eps matrix starts 16-byte aligned
        eps_true += pairs;
        corr_total += L.out_ch;
        mm_blocks += (pairs * L.khkw + mm_chunk - 1) / mm_chunk;
        qe_blocks += (L.khkw == 1) ? (pairs + kBlock * kQePairs - 1) / (kBlock * kQePairs) : (pairs + kBlock - 1) / kBlock;
    }
    if (mm_blocks > 0x7fffffff || qe_blocks > 0x7fffffff) return fail_arg("dfq_bc_plan_create: too large");

    dfq_bc_plan* p = new dfq_bc_plan();
    p->n_steps = n_steps;
    p->mm_chunk = mm_chunk;
    p->eps_elems = eps_true;
    hipError_t e;
    auto fail_alloc = [&](hipError_t err) { dfq_bc_plan_destroy(p); return fail_hip(err, "bc plan allocation", __FILE__, __LINE__); };
    {   // debug: also materialise eps[O, I/g] of every layer (dfq_bc_plan_eps); the chain does not read it
        const char* de = getenv("DFQ_BC_EPS");
        p->keep_eps = de && de[0] == '1';
    }
    timer.tick("validate");
    if (p->keep_eps && (e = p->mem.alloc((void**)&p->d_eps, sizeof(float) * eps_total)) != hipSuccess) return fail_alloc(e);
    if ((e = p->mem.alloc((void**)&p->d_corr, sizeof(float) * corr_total)) != hipSuccess) return fail_alloc(e);
    if ((e = p->mem.alloc((void**)&p->d_layers, sizeof(BcLayerDev) * n_steps)) != hipSuccess) return fail_alloc(e);
    if ((e = p->mem.alloc((void**)&p->d_mm_begin, sizeof(int32_t) * (n_steps + 1))) != hipSuccess) return fail_alloc(e);
    if ((e = p->mem.alloc((void**)&p->d_qe_begin, sizeof(int32_t) * (n_steps + 1))) != hipSuccess) return fail_alloc(e);
    if ((e = p->mem.alloc((void**)&p->d_sources, sizeof(BcSourceDev) * n_sources)) != hipSuccess) return fail_alloc(e);
    // per parity: (min, max) of every step's layer, then one arrival counter per layer (one-launch correction)
    // (+ one counter for the blocks that refresh cached moments inside the launch)
    if ((e = p->mem.alloc((void**)&p->d_slots, sizeof(uint32_t) * 2 * (3 * (size_t)n_steps + 1))) != hipSuccess) return fail_alloc(e);      // two parities
    if ((e = hipMemset(p->d_slots, 0, sizeof(uint32_t) * 2 * (3 * (size_t)n_steps + 1))) != hipSuccess) return fail_alloc(e);

    std::vector<BcLayerDev> hl(n_steps);
    std::vector<int32_t> mmb(n_steps + 1), qeb(n_steps + 1);
    // BNs read through a ReLU get a cache of their moment E[ReLU(N(beta~, gamma~^2))], keyed by beta~
    std::unordered_map<const float*, int> cache_of;
    cache_of.reserve(2 * (size_t)n_sources);
    std::vector<BcCacheSeg> segs;
    int cache_total = 0;
    for (int i = 0; i < n_sources; ++i) {
        if (!sources[i].relu || cache_of.count(sources[i].fake_bias)) continue;
        BcCacheSeg sg;
        sg.fw = sources[i].fake_weight; sg.fb = sources[i].fake_bias; sg.cache = nullptr;
        sg.channels = sources[i].channels; sg.begin = cache_total;
        cache_of[sources[i].fake_bias] = (int)segs.size();
        segs.push_back(sg);
        cache_total += sources[i].channels;
    }
    if (cache_total > 0) {
        if ((e = p->mem.alloc((void**)&p->d_cache, sizeof(float) * cache_total)) != hipSuccess) return fail_alloc(e);
        if ((e = p->mem.alloc((void**)&p->d_cache_segs, sizeof(BcCacheSeg) * segs.size())) != hipSuccess) return fail_alloc(e);
        for (BcCacheSeg& sg : segs) sg.cache = p->d_cache + sg.begin;
        if ((e = hipMemcpy(p->d_cache_segs, segs.data(), sizeof(BcCacheSeg) * segs.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
    }
    p->n_cache_segs = (int)segs.size();
    p->cache_total = cache_total;
    // tagged-value slots (see BcDep): one pair per channel of every BN that a step rewrites.  Only if every source of every step
    // is either never rewritten or rewritten by an EARLIER step (the reference's sequential loop reads the old value otherwise)
    // and no BN is rewritten twice; else the counter protocol, which orders whole steps, stays.
    std::unordered_map<const float*, int> tag_of, writer_of;
    tag_of.reserve(2 * (size_t)n_steps);
    writer_of.reserve(2 * (size_t)n_steps);
    int tag_total = 0;
    bool tagged_ok = true;
    {
        const char* te = getenv("DFQ_BC_TAGGED");
        if (te && te[0] == '0') tagged_ok = false;
    }
    for (int s2 = 0; s2 < n_steps && tagged_ok; ++s2) {
        const float* nb = steps[s2].next_bn_bias;
        if (!nb) continue;
        if (tag_of.count(nb)) { tagged_ok = false; break; }
        tag_of[nb] = tag_total;
        writer_of[nb] = s2;
        tag_total += layers[steps[s2].layer].out_ch;
    }
    for (int s2 = 0; s2 < n_steps && tagged_ok; ++s2)
        for (int m = 0; m < steps[s2].source_count; ++m) {
            auto it = writer_of.find(sources[steps[s2].source_begin + m].fake_bias);
            if (it != writer_of.end() && it->second >= s2) tagged_ok = false;
            if (it != writer_of.end() && layers[steps[it->second].layer].out_ch != sources[steps[s2].source_begin + m].channels) tagged_ok = false;
        }
    if (!tagged_ok) { tag_of.clear(); tag_total = 0; }
    std::vector<BcSourceDev> hs(n_sources);
    for (int i = 0; i < n_sources; ++i) {
        hs[i].fw = sources[i].fake_weight; hs[i].fb = sources[i].fake_bias;
        hs[i].cache = sources[i].relu ? segs[cache_of[sources[i].fake_bias]].cache : nullptr;
        hs[i].channels = sources[i].channels; hs[i].relu = sources[i].relu; hs[i].concat = sources[i].concat;
        { auto it = tag_of.find(sources[i].fake_bias); hs[i].tag_off = (it != tag_of.end()) ? it->second : -1; }
    }
    p->steps.resize(n_steps);
    p->eps_ptr.resize(n_steps);
    int64_t eps_off = 0, corr_off = 0, mb = 0, qb = 0;
    for (int s = 0; s < n_steps; ++s) {
        const dfq_layer& L = layers[steps[s].layer];
        const int64_t pairs = (int64_t)L.out_ch * L.in_per_group;
        hl[s].w = L.weight; hl[s].eps = p->keep_eps ? p->d_eps + eps_off : nullptr; hl[s].n = pairs * L.khkw; hl[s].pairs = pairs;
        hl[s].khkw = L.khkw; hl[s].pad = 0;
        mmb[s] = (int32_t)mb; qeb[s] = (int32_t)qb;
        mb += (hl[s].n + mm_chunk - 1) / mm_chunk;
        qb += (L.khkw == 1) ? (pairs + kBlock * kQePairs - 1) / (kBlock * kQePairs) : (pairs + kBlock - 1) / kBlock;
        BcStepDev& d = p->steps[s];
        d.w = L.weight; d.mm = p->d_slots + 2 * s; d.khkw = L.khkw; d.fold = -1;
        d.mm_index = s; d.mm_blocks = (int32_t)((hl[s].n + mm_chunk - 1) / mm_chunk); d.wait_cache = 0;
        d.eps = p->keep_eps ? p->d_eps + eps_off : nullptr; d.bias = L.bias; d.next_bn_bias = steps[s].next_bn_bias; d.corr = p->d_corr + corr_off;
        d.out_ch = L.out_ch; d.in_per_group = L.in_per_group; d.source_begin = steps[s].source_begin;
        d.source_count = steps[s].source_count; d.expect_len = expect_len[s];
        d.inline_sources = d.source_count <= kStepSources ? 1 : 0;
        {
            // work split: lanes per row, slots per row, rows per wave (everything preloaded: <= kBcRegs slots)
            int lg = 0;
            while ((1 << lg) < std::min(L.in_per_group, kWave)) ++lg;
            const int lanes = 1 << lg, rps = kWave / lanes;
            d.lg_lanes = lg;
            d.chunks = (L.in_per_group + lanes - 1) / lanes;
            int rw = (d.chunks > kBcRegs) ? 1 : rps * (kBcRegs / d.chunks);
            rw = std::min(rw, kBlock / kRowsPerBlock);                       // one tail row per thread
            rw = std::min(rw, std::max(1, (L.out_ch + kRowsPerBlock - 1) / kRowsPerBlock));
            d.rows_per_block = rw * kRowsPerBlock;                           // upper bound; trimmed per launch below
            { auto it = steps[s].next_bn_bias ? tag_of.find(steps[s].next_bn_bias) : tag_of.end(); d.next_tag_off = (it != tag_of.end()) ? it->second : -1; }
        }
        for (int m = 0; m < kStepSources; ++m)
            d.src[m] = (d.inline_sources && m < d.source_count) ? hs[d.source_begin + m] : BcSourceDev();
        d.next_bn_weight = nullptr; d.next_cache = nullptr;
        if (d.next_bn_bias) {
            auto it = cache_of.find(d.next_bn_bias);
            if (it != cache_of.end()) {
                if (segs[it->second].channels != L.out_ch) {
                    dfq_bc_plan_destroy(p);
                    return fail_arg("dfq_bc_plan_create: step %d: next BN has %d channels, layer has %d", s, segs[it->second].channels, L.out_ch);
                }
                d.next_bn_weight = segs[it->second].fw;
                d.next_cache = segs[it->second].cache;
            }
        }
        p->eps_ptr[s] = d.eps;
        p->weight_elems += hl[s].n;
        eps_off += (pairs + 3) & ~(int64_t)3; corr_off += L.out_ch;
    }
    mmb[n_steps] = (int32_t)mb; qeb[n_steps] = (int32_t)qb;
    timer.tick("tables");
    // ---- depthwise steps folded into the tail of the step in front of them (BcFoldDev).  Step D folds into P = D - 1 (same
    //      network) when D has one input channel per group and as many groups as outputs (every output o needs E[o] only), at
    //      most kFoldTaps taps, and its ONLY source is the BN that P rewrites, channel for channel; P must launch itself (a
    //      folded step has no tail to host another one).  Nothing lies between P and D in the sequential order of
    //      dfq.py:197-293, so every later step sees the same state as without the fold. ----
    std::vector<int> folded_into(n_steps, -1);
    std::vector<BcFoldDev> folds;
    {
        const char* fe = getenv("DFQ_BC_FOLD");
        const bool fold_on = !(fe && fe[0] == '0');
        for (int D = 1; D < n_steps && fold_on; ++D) {
            const int P = D - 1;
            if (steps[P].net != steps[D].net || folded_into[P] >= 0) continue;
            const dfq_layer& LD = layers[steps[D].layer];
            const dfq_layer& LP = layers[steps[P].layer];
            const dfq_bc_source& src = sources[steps[D].source_begin];
            if (LD.in_per_group != 1 || expect_len[D] != LD.out_ch || LD.khkw > kFoldTaps || steps[D].source_count != 1) continue;
            if (!steps[P].next_bn_bias || src.fake_bias != steps[P].next_bn_bias || src.channels != LD.out_ch || LP.out_ch != LD.out_ch) continue;
            if (steps[D].next_bn_bias == steps[P].next_bn_bias || LD.bias == LP.bias) continue;
            if (src.relu && !p->steps[P].next_cache) continue;           // (cannot happen: a source read through a ReLU has a cache)
            BcFoldDev F;
            const BcStepDev& d = p->steps[D];
            F.w = d.w; F.mm = d.mm; F.bias = d.bias; F.next_bn_bias = d.next_bn_bias; F.next_bn_weight = d.next_bn_weight;
            F.next_cache = d.next_cache; F.corr = d.corr; F.eps = nullptr; F.khkw = d.khkw; F.next_tag_off = d.next_tag_off;
            F.src_relu = src.relu ? 1 : 0; F.pad = 0; F.mm_index = d.mm_index; F.mm_blocks = d.mm_blocks;
            p->steps[P].fold = (int)folds.size();
            folds.push_back(F);
            folded_into[D] = P;
        }
        p->n_folds = (int)folds.size();
        if (!folds.empty()) {
            if ((e = p->mem.alloc((void**)&p->d_folds, sizeof(BcFoldDev) * folds.size())) != hipSuccess) return fail_alloc(e);
            if ((e = hipMemcpy(p->d_folds, folds.data(), sizeof(BcFoldDev) * folds.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
        }
    }
    const int n_live = n_steps - p->n_folds;       // steps that are launched
    // launch j = the j-th LIVE step of every network (steps arrive network by network, in graph order)
    {
        std::vector<int> ordinal(n_steps, -1);
        int n_launch = 0;
        for (int s2 = 0, cur_net = -1, k = 0; s2 < n_steps; ++s2) {
            if (steps[s2].net != cur_net) { cur_net = steps[s2].net; k = 0; }
            if (folded_into[s2] >= 0) continue;
            ordinal[s2] = k++;
            n_launch = std::max(n_launch, k);
        }
        p->launches.assign(n_launch, dfq_bc_plan::Launch{0, 0, 0, 0});
        for (int s2 = 0; s2 < n_steps; ++s2) if (ordinal[s2] >= 0) p->launches[ordinal[s2]].n += 1;
        int off = 0;
        for (auto& L : p->launches) { L.begin = off; off += L.n; L.n = 0; }
        p->launch_steps.resize(n_live);
        // A launch wants enough workgroups to occupy the chip (a single network is latency-bound: one
        // row per wave, every row in flight at once) but no more than that (each workgroup rebuilds the
        // expectation vector and pays the descriptor round trip): rows per wave shrink until the launch
        // has about `target` workgroups.
        const char* te = getenv("DFQ_BC_BLOCKS");
        const int target = (te && atoi(te) > 0) ? atoi(te) : 512;
        std::vector<int> n_in_launch(n_launch, 0);
        for (int s2 = 0; s2 < n_steps; ++s2) if (ordinal[s2] >= 0) n_in_launch[ordinal[s2]] += 1;
        for (int s2 = 0; s2 < n_steps; ++s2) {
            if (ordinal[s2] < 0) continue;
            BcStepDev& d = p->steps[s2];
            const int rps = kWave >> d.lg_lanes;
            int rw = d.rows_per_block / kRowsPerBlock;
            const int64_t rows_launch = (int64_t)d.out_ch * n_in_launch[ordinal[s2]];
            const int want = (int)std::max<int64_t>(1, rows_launch / ((int64_t)kRowsPerBlock * target));
            rw = std::min(rw, std::max(want, std::min(rps, rw)));            // never below one full register slot of rows
            d.rows_per_block = rw * kRowsPerBlock;
        }
        p->one_group = !(getenv("DFQ_BC_ONE_GROUP") && getenv("DFQ_BC_ONE_GROUP")[0] == '0');
        for (int s2 = 0; s2 < n_steps; ++s2) {
            if (ordinal[s2] < 0) continue;
            if (p->steps[s2].rows_per_block / kRowsPerBlock > (kWave >> p->steps[s2].lg_lanes)) p->one_group = false;
        }
        for (int s2 = 0; s2 < n_steps; ++s2) {
            if (ordinal[s2] < 0) continue;
            auto& L = p->launches[ordinal[s2]];
            p->launch_steps[L.begin + L.n++] = p->steps[s2];
            L.max_blocks = std::max(L.max_blocks, (p->steps[s2].out_ch + p->steps[s2].rows_per_block - 1) / p->steps[s2].rows_per_block);
            L.max_expect = std::max(L.max_expect, p->steps[s2].expect_len);
        }
        if ((e = p->mem.alloc((void**)&p->d_steps, sizeof(BcStepDev) * n_live)) != hipSuccess) return fail_alloc(e);
        if ((e = hipMemcpy(p->d_steps, p->launch_steps.data(), sizeof(BcStepDev) * n_live, hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
        // workgroup table of the one-launch chain: chain position after chain position; a step waits for the previous LIVE
        // step of its network (steps arrive network by network in graph order)
        std::vector<int> pos(n_steps, -1), fill(n_launch, 0);
        for (int s2 = 0; s2 < n_steps; ++s2) if (ordinal[s2] >= 0) pos[s2] = p->launches[ordinal[s2]].begin + fill[ordinal[s2]]++;
        std::vector<int> wait_of(n_live, -1), blocks_of(n_live, 0);
        for (int s2 = 0, prev_live = -1, cur_net = -1; s2 < n_steps; ++s2) {
            if (steps[s2].net != cur_net) { cur_net = steps[s2].net; prev_live = -1; }
            if (ordinal[s2] < 0) continue;
            blocks_of[pos[s2]] = (p->steps[s2].out_ch + p->steps[s2].rows_per_block - 1) / p->steps[s2].rows_per_block;
            if (prev_live >= 0) wait_of[pos[s2]] = pos[prev_live];
            prev_live = s2;
        }
        std::vector<BcChainRef> refs;
        for (int q = 0; q < n_live; ++q) {
            p->max_expect = std::max(p->max_expect, p->launch_steps[q].expect_len);
            for (int b = 0; b < blocks_of[q]; ++b)
                refs.push_back(BcChainRef{q, b, wait_of[q], wait_of[q] >= 0 ? blocks_of[wait_of[q]] : 0});
        }
        // ---- a batch's networks SKEWED against each other (late round 6).  In position-major order every network of a batch is at
        //      the same chain position at the same time: ~33 small positions that hand over at latency pace while the memory system
        //      idles, then the three large ones (84 % of the weights) that stream while nothing hands over (profiles/r05_bc_chain.txt:
        //      215 + 100 us for a batch of 32).  The workgroups are therefore listed by position + skew x network: network n runs
        //      `skew` positions behind network n-1, so at any time some networks stream their large layers while others hand over.
        //      Within a network the order of positions is what it was -- every workgroup's producers still have lower grid indices
        //      (the in-launch waits stay deadlock-free) -- and a workgroup computes what it computed: results are bit-identical.
        //      DFQ_BC_SKEW sets the skew (positions per network; 0: position-major). ----
        std::vector<int> P_of(n_live, 0), net_rank_of(n_live, 0);
        double skew = 0.0;
        {
            std::vector<int> nets_seen;
            for (int s2 = 0; s2 < n_steps; ++s2) {
                if (ordinal[s2] < 0) continue;
                P_of[pos[s2]] = ordinal[s2];
                auto it = std::find(nets_seen.begin(), nets_seen.end(), (int)steps[s2].net);
                if (it == nets_seen.end()) { nets_seen.push_back((int)steps[s2].net); it = nets_seen.end() - 1; }
                net_rank_of[pos[s2]] = (int)(it - nets_seen.begin());
            }
            const char* se = getenv("DFQ_BC_SKEW");
            skew = se ? atof(se) : kBcSkewDefault;
            if (nets_seen.size() < 2 || !(skew > 0.0)) skew = 0.0;
        }
        auto key_of = [&](int q) { return (double)P_of[q] + skew * (double)net_rank_of[q]; };
        if (skew > 0.0)
            std::stable_sort(refs.begin(), refs.end(), [&](const BcChainRef& a, const BcChainRef& b) { return key_of(a.step) < key_of(b.step); });
        p->chain_blocks = (int)refs.size();
        if ((e = p->mem.alloc((void**)&p->d_refs, sizeof(BcChainRef) * std::max<size_t>(1, refs.size()))) != hipSuccess) return fail_alloc(e);
        if ((e = hipMemcpy(p->d_refs, refs.data(), sizeof(BcChainRef) * refs.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
        // ---- one-launch correction: the same table with the min/max blocks woven in.  The blocks of the layers that chain position
        //      P needs (its steps' own layers and the depthwise layers folded into them) lie `ahead` positions in front of P's
        //      workgroups; the first `ahead` positions' blocks open the grid.  A never-rewritten BN read through a ReLU is covered
        //      too: its cached moment is refreshed by the launch's first blocks and its readers wait for them (wait_cache /
        //      cache_arrive, the step == -2 blocks below).  Not for plans that keep the eps matrices. ----
        {
            // Default: for a BATCH only.  Measured (tools/gpu_r05_bench_env_ab.sh, two alternating rounds): the batch of 32 0.410-0.415 ->
            // 0.375-0.379 ms -- its large late layers are reduced while the early positions hand over -- but ONE network 0.1265 -> 0.131-
            // 0.136 ms, ResNet-18 0.082 -> 0.112-0.117: every workgroup of a single network's chain is resident from the first
            // microsecond and polls its sources' slots (180 000 threads, a device-scope load each every 64 clocks), and min/max blocks
            // that stream next to that finish at 40 us instead of 5 (tools/bc_trace.py: `quantised` of position 0).  DFQ_BC_ONE_LAUNCH=1 / 0
            // forces it on / off.
            const char* fe = getenv("DFQ_BC_ONE_LAUNCH");
            const bool want = fe ? (fe[0] != '0') : !p->one_group;
            bool ok = tagged_ok && tag_total > 0 && !p->keep_eps && want;
            std::vector<int> wait_cache(n_steps, 0);
            for (int s2 = 0; s2 < n_steps && ok; ++s2) {
                if (p->steps[s2].chunks > kBcRegs) ok = false;      // (nothing in the way; its rows stream from memory behind the wait: keep the plain pair of launches)
                for (int m = 0; m < steps[s2].source_count; ++m) {
                    const BcSourceDev& src = hs[steps[s2].source_begin + m];
                    if (src.relu && src.tag_off < 0) wait_cache[s2] = 1;          // a never-rewritten BN read through a ReLU
                }
            }
            // the cached moments of never-rewritten BNs are refreshed by the first blocks of the launch (the rewritten BNs' are
            // produced by the steps themselves and read through the tagged slots)
            std::vector<BcCacheSeg> segs_launch;
            int cache_total_launch = 0;
            for (const BcCacheSeg& sg : segs) {
                if (tag_of.count(sg.fb)) continue;
                BcCacheSeg c = sg;
                c.begin = cache_total_launch;
                cache_total_launch += sg.channels;
                segs_launch.push_back(c);
            }
            const int cache_blocks_launch = (cache_total_launch + kBlock - 1) / kBlock;
            if (ok) {
                const char* ae = getenv("DFQ_BC_MM_AHEAD");
                const int ahead = (ae && atoi(ae) >= 0) ? atoi(ae) : 2;
                // min/max blocks per chain position (launch), in step order
                std::vector<std::vector<int>> mm_of(n_launch);
                for (int s2 = 0; s2 < n_steps; ++s2) {
                    const int host = ordinal[s2] >= 0 ? s2 : folded_into[s2];     // a folded step's layer is needed where its host runs
                    if (host < 0 || ordinal[host] < 0) { ok = false; break; }
                    for (int b = mmb[s2]; b < mmb[s2 + 1]; ++b) mm_of[ordinal[host]].push_back(b);
                }
                if (ok) {
                    std::vector<BcChainRef> fused;
                    fused.reserve(refs.size() + (size_t)mmb[n_steps] + cache_blocks_launch);
                    auto emit_mm = [&](int P) { if (P < n_launch) for (int b : mm_of[P]) fused.push_back(BcChainRef{-1, b, -1, 0}); };
                    for (int b = 0; b < cache_blocks_launch; ++b) fused.push_back(BcChainRef{-2, b, -1, 0});
                    size_t r = 0;
                    if (skew > 0.0) {
                        // skewed networks (above): a layer's min/max blocks lie `ahead` positions in front of the workgroups of ITS
                        // network's step that needs them -- merge the two sorted lists
                        struct MmItem { double key; int block; };
                        std::vector<MmItem> mm_items;
                        for (int s2 = 0; s2 < n_steps; ++s2) {
                            const int host = ordinal[s2] >= 0 ? s2 : folded_into[s2];
                            const double k = key_of(pos[host]) - (double)ahead;
                            for (int b = mmb[s2]; b < mmb[s2 + 1]; ++b) mm_items.push_back(MmItem{k, b});
                        }
                        std::stable_sort(mm_items.begin(), mm_items.end(), [](const MmItem& a, const MmItem& b) { return a.key < b.key; });
                        size_t m = 0;
                        for (; r < refs.size(); ++r) {
                            const double k = key_of(refs[r].step);
                            for (; m < mm_items.size() && mm_items[m].key <= k; ++m) fused.push_back(BcChainRef{-1, mm_items[m].block, -1, 0});
                            fused.push_back(refs[r]);
                        }
                        for (; m < mm_items.size(); ++m) fused.push_back(BcChainRef{-1, mm_items[m].block, -1, 0});
                    } else {
                        for (int P = 0; P < std::min(ahead, n_launch); ++P) emit_mm(P);
                        for (int P = 0; P < n_launch; ++P) {
                            emit_mm(P + ahead);
                            const int q_end = p->launches[P].begin + p->launches[P].n;
                            for (; r < refs.size() && refs[r].step < q_end; ++r) fused.push_back(refs[r]);
                        }
                    }
                    if (r == refs.size() && fused.size() == refs.size() + (size_t)mmb[n_steps] + cache_blocks_launch) {
                        p->chain_blocks_fused = (int)fused.size();
                        if (!segs_launch.empty()) {
                            if ((e = p->mem.alloc((void**)&p->d_cache_segs_launch, sizeof(BcCacheSeg) * segs_launch.size())) != hipSuccess) return fail_alloc(e);
                            if ((e = hipMemcpy(p->d_cache_segs_launch, segs_launch.data(), sizeof(BcCacheSeg) * segs_launch.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
                        }
                        p->n_cache_segs_launch = (int)segs_launch.size();
                        p->cache_total_launch = cache_total_launch;
                        // (the step table is on the device already: the flag goes into both copies)
                        for (int s2 = 0; s2 < n_steps; ++s2) {
                            if (!wait_cache[s2] || ordinal[s2] < 0) continue;
                            p->steps[s2].wait_cache = 1;
                            p->launch_steps[pos[s2]].wait_cache = 1;
                        }
                        if ((e = hipMemcpy(p->d_steps, p->launch_steps.data(), sizeof(BcStepDev) * n_live, hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
                        if ((e = p->mem.alloc((void**)&p->d_refs_fused, sizeof(BcChainRef) * fused.size())) != hipSuccess) return fail_alloc(e);
                        if ((e = hipMemcpy(p->d_refs_fused, fused.data(), sizeof(BcChainRef) * fused.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
                        p->fused_mm = true;
                    }
                }
            }
        }
        if ((e = p->mem.alloc((void**)&p->d_counters, sizeof(uint32_t) * ((size_t)n_steps * kBcDepStride + 1))) != hipSuccess) return fail_alloc(e);
        if ((e = hipMemset(p->d_counters, 0, sizeof(uint32_t) * ((size_t)n_steps * kBcDepStride + 1))) != hipSuccess) return fail_alloc(e);
        if (tag_total > 0) {
            if ((e = p->mem.alloc((void**)&p->d_tags, sizeof(unsigned long long) * 2 * (size_t)tag_total)) != hipSuccess) return fail_alloc(e);
            if ((e = hipMemset(p->d_tags, 0, sizeof(unsigned long long) * 2 * (size_t)tag_total)) != hipSuccess) return fail_alloc(e);
        }
        const char* me = getenv("DFQ_BC_MERGED");
        p->merged = !(me && me[0] == '0');
    }
    p->minmax_blocks = (int)mb; p->qerr_blocks = (int)qb;
    timer.tick("launches");
    if ((e = hipMemcpy(p->d_layers, hl.data(), sizeof(BcLayerDev) * n_steps, hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
    if ((e = hipMemcpy(p->d_mm_begin, mmb.data(), sizeof(int32_t) * (n_steps + 1), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
    if ((e = hipMemcpy(p->d_qe_begin, qeb.data(), sizeof(int32_t) * (n_steps + 1), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
    if ((e = hipMemcpy(p->d_sources, hs.data(), sizeof(BcSourceDev) * n_sources, hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
    timer.tick("uploads");
    if ((e = hipDeviceSynchronize()) != hipSuccess) return fail_alloc(e);
    *out_plan = p;
    return DFQ_OK;
}

static int bc_run_direct(dfq_bc_plan* p, int32_t symmetric, hipStream_t st);

int dfq_bc_plan_run(dfq_bc_plan* p, int32_t symmetric, void* stream) {
    if (!p) return fail_arg("dfq_bc_plan_run: null plan");
    hipStream_t st = as_stream(stream);
    const char* ge = getenv("DFQ_GRAPH");
    if (!(ge && ge[0] == '1')) return bc_run_direct(p, symmetric, st);     // opt-in, see dfq_le.hip
    // ~(n_layers + 4) dependent launches with fixed arguments: record once, replay as one graph launch
    hipGraphExec_t& exec = p->exec[symmetric ? 1 : 0];
    if (!exec) {
        if (!p->capture_stream) DFQ_HIP_TRY(hipStreamCreate(&p->capture_stream));
        DFQ_HIP_TRY(hipStreamBeginCapture(p->capture_stream, hipStreamCaptureModeThreadLocal));
        const int rc = bc_run_direct(p, symmetric, p->capture_stream);
        hipGraph_t graph = nullptr;
        const hipError_t ee = hipStreamEndCapture(p->capture_stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (ee != hipSuccess) return fail_hip(ee, "hipStreamEndCapture", __FILE__, __LINE__);
        DFQ_HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
    }
    DFQ_HIP_TRY(hipGraphLaunch(exec, st));
    return DFQ_OK;
}

static int bc_run_direct(dfq_bc_plan* p, int32_t symmetric, hipStream_t st) {
    const bool chain = p->merged && p->chain_blocks > 0;
    // (the NULL stream is a caller's stream like any other: `capture_stream` is null until a graph is recorded, and until round 5
    //  a run on the NULL stream compared equal to it -- counter protocol, no guard: 250 instead of 130 us for a MobileNetV2)
    const bool capturing = p->capture_stream != nullptr && st == p->capture_stream;
    const bool tagged_run = chain && p->d_tags != nullptr && !capturing;
    // (min, max) slots: two parities.  A tagged run accumulates into parity `slot_parity` (zero: cleared by the previous tagged
    // run's chain launch, or at creation) and has NO clear launch; any other run clears both parities and its counters + error
    // word with one launch and uses parity 0.
    int parity = 0;
    if (tagged_run) {
        parity = p->slot_parity;
        p->slot_parity ^= 1;
    } else {
        clear_buffers(st, p->d_slots, sizeof(uint32_t) * 2 * (3 * (size_t)p->n_steps + 1),
                      chain ? p->d_counters : nullptr, sizeof(uint32_t) * ((size_t)p->n_steps * kBcDepStride + 1));
        DFQ_CHECK_LAUNCH();
        p->slot_parity = 1;                 // parity 0 is dirty after this run, parity 1 clean
    }
    p->last_tagged = tagged_run;
    const int mm_off = parity * (3 * p->n_steps + 1);
    uint32_t* slots = p->d_slots + mm_off;
    const int cache_blocks = (p->cache_total + kBlock - 1) / kBlock;
    // one-launch correction: the min/max blocks are workgroups of the tagged chain launch below
    const bool one_launch = tagged_run && p->fused_mm;
    if (!one_launch) {
        hipLaunchKernelGGL(bc_minmax_kernel, dim3(p->minmax_blocks + cache_blocks), dim3(kBlock), 0, st, (const BcLayerDev*)p->d_layers,
                           (const int32_t*)p->d_mm_begin, p->n_steps, slots, p->minmax_blocks,
                           (const BcCacheSeg*)p->d_cache_segs, p->n_cache_segs, p->cache_total, p->mm_chunk);
        DFQ_CHECK_LAUNCH();
    }
    if (p->keep_eps) {
        hipLaunchKernelGGL(bc_quant_error_kernel, dim3(p->qerr_blocks), dim3(kBlock), 0, st, (const BcLayerDev*)p->d_layers,
                           (const int32_t*)p->d_qe_begin, p->n_steps, (const uint32_t*)slots, 8, (int)symmetric);
        DFQ_CHECK_LAUNCH();
    }
    if (chain) {
        uint32_t* err = p->d_counters + (size_t)p->n_steps * kBcDepStride;
        // the chain kernel contains in-launch waits: never concurrent with another stream's (dfq_common.hpp)
        std::unique_ptr<SpinGuard> guard;
        if (!capturing) guard.reset(new SpinGuard(st));
        // a recorded graph replays its arguments, so the run epoch of the tagged slots cannot advance: counters there
        unsigned long long* tags = capturing ? nullptr : p->d_tags;
        if (tags && ++p->epoch < 2u) p->epoch = 2u;
        const int spin_limit = spin_limit_from_env(20000000);
        BcFusedMm fm;
        fm.layers = (const BcLayerDev*)p->d_layers; fm.block_begin = (const int32_t*)p->d_mm_begin; fm.slots = slots;
        fm.arrive = one_launch ? slots + 2 * p->n_steps : nullptr; fm.n_layers = p->n_steps; fm.mm_chunk = p->mm_chunk;
        fm.segs = (const BcCacheSeg*)p->d_cache_segs_launch; fm.cache_arrive = one_launch ? slots + 3 * p->n_steps : nullptr;
        fm.n_segs = p->n_cache_segs_launch; fm.cache_total = p->cache_total_launch; fm.cache_blocks = (p->cache_total_launch + kBlock - 1) / kBlock; fm.pad = 0;
        const int grid_blocks = one_launch ? p->chain_blocks_fused : p->chain_blocks;
        const BcChainRef* ref_table = one_launch ? p->d_refs_fused : p->d_refs;
#define DFQ_BC_CHAIN_LAUNCH(EXP, ONE)                                                                                                  \
        hipLaunchKernelGGL((bc_chain_kernel<EXP, ONE>), dim3(grid_blocks), dim3(kBlock), 0, st, (const BcStepDev*)p->d_steps,             \
                           ref_table, (const BcSourceDev*)p->d_sources, (const BcFoldDev*)p->d_folds, p->d_counters,                    \
                           err, tags, p->epoch, (int)symmetric, spin_limit, mm_off,                                                     \
                           tagged_run ? p->d_slots + (parity ^ 1) * (3 * p->n_steps + 1) : nullptr, 3 * p->n_steps + 1, fm)
        if (p->max_expect <= kExpectSmall) { if (p->one_group) DFQ_BC_CHAIN_LAUNCH(kExpectSmall, true); else DFQ_BC_CHAIN_LAUNCH(kExpectSmall, false); }
        else { if (p->one_group) DFQ_BC_CHAIN_LAUNCH(kExpectMax, true); else DFQ_BC_CHAIN_LAUNCH(kExpectMax, false); }
#undef DFQ_BC_CHAIN_LAUNCH
        DFQ_CHECK_LAUNCH();
        return DFQ_OK;
    }
    for (const auto& L : p->launches) {
        const BcStepDev* table = (L.n == 1) ? nullptr : p->d_steps + L.begin;
        if (L.max_expect <= kExpectSmall)
            hipLaunchKernelGGL(bc_step_kernel<kExpectSmall>, dim3(L.max_blocks, L.n), dim3(kBlock), 0, st,
                               p->launch_steps[L.begin], table, (const BcSourceDev*)p->d_sources, (const BcFoldDev*)p->d_folds, (int)symmetric);
        else
            hipLaunchKernelGGL(bc_step_kernel<kExpectMax>, dim3(L.max_blocks, L.n), dim3(kBlock), 0, st,
                               p->launch_steps[L.begin], table, (const BcSourceDev*)p->d_sources, (const BcFoldDev*)p->d_folds, (int)symmetric);
        DFQ_CHECK_LAUNCH();
    }
    return DFQ_OK;
}

// Synchronises `stream` and reports whether a workgroup of the one-launch chain gave up waiting for the step it
// depends on (a logic error surfaces here instead of as a hung GPU)
int dfq_bc_plan_status(dfq_bc_plan* p, void* stream) {
    if (!p) return fail_arg("dfq_bc_plan_status: null plan");
    hipStream_t st = as_stream(stream);
    uint32_t gave_up = 0;
    DFQ_HIP_TRY(hipMemcpyAsync(&gave_up, p->d_counters + (size_t)p->n_steps * kBcDepStride, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    DFQ_HIP_TRY(hipStreamSynchronize(st));
    // (tagged protocol: the word holds the epoch of the latest failed run, an older failure is none of this run's)
    if (p->last_tagged ? (gave_up == p->epoch) : (gave_up != 0u)) {
        set_error("dfq_bc_plan_status: a workgroup gave up waiting for the previous correction step (results are invalid)");
        return DFQ_ERR_ABANDONED;
    }
    return DFQ_OK;
}

const float* dfq_bc_plan_eps(const dfq_bc_plan* p, int32_t step) {
    return (p && step >= 0 && step < p->n_steps) ? p->eps_ptr[step] : nullptr;
}
const float* dfq_bc_plan_correction(const dfq_bc_plan* p, int32_t step) {
    return (p && step >= 0 && step < p->n_steps) ? p->steps[step].corr : nullptr;
}
int64_t dfq_bc_plan_weight_elements(const dfq_bc_plan* p) { return p ? p->weight_elems : 0; }
int32_t dfq_bc_plan_tagged(const dfq_bc_plan* p) { return (p && p->merged && p->d_tags) ? 1 : 0; }
// the chain as one launch contains in-launch waits (dfq_bc_plan_status can report DFQ_ERR_ABANDONED); safe mode: one launch per
// chain position from now on -- nothing waits, nothing can be abandoned
int32_t dfq_bc_plan_has_waits(const dfq_bc_plan* p) { return (p && p->merged && p->chain_blocks > 0) ? 1 : 0; }
int dfq_bc_plan_set_safe_mode(dfq_bc_plan* p) {
    if (!p) return fail_arg("dfq_bc_plan_set_safe_mode: null plan");
    p->merged = false;
    // the error word of an abandoned chain launch is none of the per-position launches' (they have no waits and do not clear it)
    DFQ_HIP_TRY(hipMemset(p->d_counters + (size_t)p->n_steps * kBcDepStride, 0, sizeof(uint32_t)));
    p->last_tagged = false;
    return DFQ_OK;
}
int32_t dfq_bc_plan_last_run_tagged(const dfq_bc_plan* p) { return (p && p->last_tagged) ? 1 : 0; }
int32_t dfq_bc_plan_one_launch(const dfq_bc_plan* p) { return (p && p->merged && p->fused_mm) ? 1 : 0; }

// measurement builds (-DDFQ_BC_TRACE=1): the timestamps of the latest chain launch, 8 words per workgroup; 0 words otherwise
int64_t dfq_bc_debug_trace(long long* out, int64_t words) {
#if DFQ_BC_TRACE
    if (!out || words <= 0) return 0;
    const int64_t n = std::min<int64_t>(words, (int64_t)kBcTraceWgs * kBcTraceWords);
    if (hipDeviceSynchronize() != hipSuccess) return 0;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bc_trace), sizeof(long long) * n, 0, hipMemcpyDeviceToHost) != hipSuccess) return 0;
    return n;
#else
    (void)out; (void)words;
    return 0;
#endif
}
int64_t dfq_bc_plan_eps_elements(const dfq_bc_plan* p) { return p ? p->eps_elems : 0; }
int32_t dfq_bc_plan_folded(const dfq_bc_plan* p) { return p ? p->n_folds : 0; }
int32_t dfq_bc_plan_chain_steps(const dfq_bc_plan* p) { return p ? (int32_t)p->launches.size() : 0; }

}  // extern "C"
