// Per-tensor range reductions and the fake-quant round trip (utils/quantize.py:23-76, :102-119;
// utils/layer_transform.py:279-296; dfq.py:8-25) as HIP kernels for gfx950.
//
// All kernels here are streaming, HBM/L2-bound: one pass for min/max, one read+write pass for the
// quantiser.  Reductions use 64-lane butterflies, one LDS hop across the 4 waves of a workgroup
// and one order-preserving atomicMax per workgroup (min/max are order independent, so the result
// is deterministic).
#include <vector>

#include <algorithm>

#include "dfq_common.hpp"
#include "dfq_le_shared.hpp"   // spin_limit_from_env

namespace dfq {

constexpr int kChunk = kBlock * 16;   // elements one workgroup owns in the multi-tensor kernels

// ---------------------------------------------------------------------------------------------
// device bodies
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void thread_minmax_range(const float* __restrict__ x, int64_t begin,
                                                    int64_t end, float& mn, float& mx) {
    const int tid = threadIdx.x;
    const float* p = x + begin;
    const int64_t len = end - begin;
    if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
        const int64_t n4 = len >> 2;
        const float4* p4 = reinterpret_cast<const float4*>(p);
        // four independent 16-byte loads per trip (a read-only pass with one load in flight per lane leaves most of
        // the memory pipeline idle), raw v_min / v_max (no canonicalisation prologue)
        int64_t i = tid;
        for (; i + 7 * kBlock < n4; i += 8 * kBlock) {              // eight 16-byte loads in flight per lane
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (kReadNt) { const fvec4 t = DFQ_NT_LOAD((const fvec4*)(p4 + i + u * kBlock)); v[u].x = t[0]; v[u].y = t[1]; v[u].z = t[2]; v[u].w = t[3]; }
                else v[u] = p4[i + u * kBlock];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                mn = vmin_raw(vmin_raw(mn, v[u].x), vmin_raw(v[u].y, vmin_raw(v[u].z, v[u].w)));
                mx = vmax_raw(vmax_raw(mx, v[u].x), vmax_raw(v[u].y, vmax_raw(v[u].z, v[u].w)));
            }
        }
        for (; i + 3 * kBlock < n4; i += 4 * kBlock) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p4[i + u * kBlock];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                mn = vmin_raw(vmin_raw(mn, v[u].x), vmin_raw(v[u].y, vmin_raw(v[u].z, v[u].w)));
                mx = vmax_raw(vmax_raw(mx, v[u].x), vmax_raw(v[u].y, vmax_raw(v[u].z, v[u].w)));
            }
        }
        for (; i < n4; i += kBlock) {
            const float4 v = p4[i];
            mn = vmin_raw(vmin_raw(mn, v.x), vmin_raw(v.y, vmin_raw(v.z, v.w)));
            mx = vmax_raw(vmax_raw(mx, v.x), vmax_raw(v.y, vmax_raw(v.z, v.w)));
        }
        for (int64_t i = (n4 << 2) + tid; i < len; i += kBlock) {
            const float v = p[i];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    } else {
        for (int64_t i = tid; i < len; i += kBlock) {
            const float v = p[i];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    }
}

// block-wide min/max -> one pair of atomics on the (max slot, min slot) pair
__device__ __forceinline__ void block_publish_minmax(float mn, float mx, uint32_t* slot_pair) {
    __shared__ float sh_mn[kBlock / kWave];
    __shared__ float sh_mx[kBlock / kWave];
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int wave = threadIdx.x / kWave;
    if ((threadIdx.x % kWave) == 0) {
        sh_mn[wave] = mn;
        sh_mx[wave] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = sh_mn[0], b = sh_mx[0];
#pragma unroll
        for (int w = 1; w < kBlock / kWave; ++w) {
            a = fminf(a, sh_mn[w]);
            b = fmaxf(b, sh_mx[w]);
        }
        if (a <= b) {   // false only if the range was empty
            atomicMax(slot_pair + 0, ~enc_ord(a));   // min slot
            atomicMax(slot_pair + 1, enc_ord(b));    // max slot
        }
    }
}

__device__ __forceinline__ int find_segment(const int32_t* __restrict__ block_begin, int n_segs, int block) {
    int lo = 0, hi = n_segs - 1;   // largest s with block_begin[s] <= block
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (block_begin[mid] <= block) lo = mid; else hi = mid - 1;
    }
    return lo;
}

struct SegDev {
    float* data;
    int32_t* codes;
    int64_t n;
    int32_t num_bits;
    int32_t symmetric;
};

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void minmax_kernel(const float* __restrict__ x, int64_t n,
                                                        uint32_t* __restrict__ slot_pair) {
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t c = blockIdx.x; c * kChunk < n; c += gridDim.x) {
        const int64_t b = c * kChunk;
        const int64_t e = (b + kChunk < n) ? b + kChunk : n;
        thread_minmax_range(x, b, e, mn, mx);
    }
    block_publish_minmax(mn, mx, slot_pair);
}

__global__ void slots_decode_kernel(const uint32_t* __restrict__ slots, float* __restrict__ out, int n_pairs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pairs) {
        out[2 * i + 0] = slot_min(slots[2 * i + 0]);
        out[2 * i + 1] = slot_max(slots[2 * i + 1]);
    }
}

__global__ __launch_bounds__(kBlock) void seg_minmax_kernel(const SegDev* __restrict__ segs,
                                                            const int32_t* __restrict__ block_begin,
                                                            int n_segs, uint32_t* __restrict__ slots) {
    const int s = find_segment(block_begin, n_segs, blockIdx.x);
    const SegDev sg = segs[s];
    const int64_t b = (int64_t)(blockIdx.x - block_begin[s]) * kChunk;
    const int64_t e = (b + kChunk < sg.n) ? b + kChunk : sg.n;
    float mn = INFINITY, mx = -INFINITY;
    thread_minmax_range(sg.data, b, e, mn, mx);
    block_publish_minmax(mn, mx, slots + 2 * s);
}

// x may alias y (in-place), so no __restrict__ here
__device__ __forceinline__ void thread_fake_quant_range(const float* x, float* y,
                                                        int32_t* __restrict__ codes, int64_t begin,
                                                        int64_t end, const QParams& p) {
    for (int64_t i = begin + threadIdx.x; i < end; i += kBlock) {
        float code;
        const float v = fake_quant_one(x[i], p, &code);
        y[i] = v;
        if (codes) codes[i] = (int32_t)code;
    }
}

__global__ __launch_bounds__(kBlock) void seg_fake_quant_kernel(const SegDev* __restrict__ segs,
                                                                const int32_t* __restrict__ block_begin,
                                                                int n_segs,
                                                                const uint32_t* __restrict__ slots,
                                                                float* __restrict__ minmax_out) {
    const int s = find_segment(block_begin, n_segs, blockIdx.x);
    const SegDev sg = segs[s];
    const int chunk = blockIdx.x - block_begin[s];
    const float mn = slot_min(slots[2 * s + 0]);
    const float mx = slot_max(slots[2 * s + 1]);
    if (chunk == 0 && threadIdx.x == 0) {
        minmax_out[2 * s + 0] = mn;
        minmax_out[2 * s + 1] = mx;
    }
    const QParams p = qparams_double((double)mn, (double)mx, sg.num_bits, sg.symmetric);
    const int64_t b = (int64_t)chunk * kChunk;
    const int64_t e = (b + kChunk < sg.n) ? b + kChunk : sg.n;
    thread_fake_quant_range(sg.data, sg.data, sg.codes, b, e, p);
}

// non-temporal hint on the 16-byte loads (1) / stores (2) of fake_quant_kernel (dfq_common.hpp).  Measured on a
// [64,96,112,112] activation: 113 us without, 107 us with both, 97 us with the stores only.
#ifndef DFQ_QUANT_NT
#define DFQ_QUANT_NT 2
#endif
constexpr int kQuantNt = DFQ_QUANT_NT;

// range_mode 0: `p0` is final.  1: double recipe from minmax_dev.  2: float32 recipe from minmax_dev.
__global__ __launch_bounds__(kBlock) void fake_quant_kernel(const float* x, float* y,
                                                            int64_t n, QParams p0, int num_bits, int symmetric,
                                                            int range_mode, const float* __restrict__ minmax_dev,
                                                            int32_t* __restrict__ codes) {
    QParams p = p0;
    if (range_mode == 1) p = qparams_double((double)minmax_dev[0], (double)minmax_dev[1], num_bits, symmetric);
    else if (range_mode == 2) p = qparams_float(minmax_dev[0], minmax_dev[1], num_bits, symmetric);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    if (!codes && ((((uintptr_t)x) | ((uintptr_t)y)) & 15u) == 0) {
        // 16-byte vectors over the body (in place or out of place: every element is read before it is written)
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
            const fvec4 v = (kQuantNt & 1) ? DFQ_NT_LOAD((const fvec4*)(x + 4 * i)) : *(const fvec4*)(x + 4 * i);
            fvec4 r;
            float code;
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = fake_quant_one(v[k], p, &code);
            if (kQuantNt & 2) DFQ_NT_STORE(r, (fvec4*)(y + 4 * i));
            else *(fvec4*)(y + 4 * i) = r;
        }
        for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
            float code;
            y[i] = fake_quant_one(x[i], p, &code);
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float code;
        const float v = fake_quant_one(x[i], p, &code);
        y[i] = v;
        if (codes) codes[i] = (int32_t)code;
    }
}

// per-sample min/max: grid (spans, samples); a workgroup streams ONE contiguous span of a sample (tens of KB: a read-only
// pass needs long runs per workgroup -- with 16 KB chunks the launch was 16 000 workgroups of one load trip each and ran at
// 1.7 TB/s)
__global__ __launch_bounds__(kBlock) void sample_minmax_kernel(const float* __restrict__ x, int64_t sample_len, int64_t span,
                                                               uint32_t* __restrict__ slots) {
    const int smp = blockIdx.y;
    const float* xs = x + (int64_t)smp * sample_len;
    const int64_t b = (int64_t)blockIdx.x * span;
    const int64_t e = (b + span < sample_len) ? b + span : sample_len;
    float mn = INFINITY, mx = -INFINITY;
    if (b < e) thread_minmax_range(xs, b, e, mn, mx);
    block_publish_minmax(mn, mx, slots + 2 * smp);
}

// mean over samples of the per-sample extrema (float64 accumulation in a fixed order, rounded once); valid in every thread
__device__ __forceinline__ void sample_mean(const uint32_t* __restrict__ slots, int n_samples, double* sh, float& mn, float& mx) {
    double smn = 0.0, smx = 0.0;
    for (int i = threadIdx.x; i < n_samples; i += kBlock) {
        smn += (double)slot_min(slots[2 * i + 0]);
        smx += (double)slot_max(slots[2 * i + 1]);
    }
    smn = block_sum(smn, sh);
    smx = block_sum(smx, sh);
    // one sample: the mean of one element is the element itself
    mn = (n_samples == 1) ? slot_min(slots[0]) : (float)(smn / (double)n_samples);
    mx = (n_samples == 1) ? slot_max(slots[1]) : (float)(smx / (double)n_samples);
}

// mean over samples + running_min/max update (quantize.py:106-107)
__global__ __launch_bounds__(kBlock) void sample_mean_kernel(const uint32_t* __restrict__ slots, int n_samples,
                                                             float* __restrict__ out2, float* __restrict__ running2) {
    __shared__ double sh[kBlock / kWave];
    float mn, mx;
    sample_mean(slots, n_samples, sh, mn, mx);
    if (threadIdx.x == 0) {
        out2[0] = mn;
        out2[1] = mx;
        if (running2) {
            // Python min(running_min, v) keeps running_min unless v < running_min (NaN keeps it)
            if (mn < running2[0]) running2[0] = mn;
            if (mx > running2[1]) running2[1] = mx;
        }
    }
}

// QuantMeasure.forward with update_stat (utils/quantize.py:102-119) behind the per-sample extrema: EVERY workgroup forms the
// mean of the extrema (the same fixed-order float64 sums -> the same two numbers), folds it into the running range
// (quantize.py:106-107; min(min(r, m), m) == min(r, m), so it does not matter whether a workgroup reads the running pair
// before or after workgroup 0 stored the folded one) and quantises its share of x with that range -- the separate one-
// workgroup launch for the mean and the memset before the extrema launch are gone: workgroup 0 also clears the slots the
// NEXT call accumulates into (the caller alternates between two slot buffers).
__global__ __launch_bounds__(kBlock) void measured_fake_quant_kernel(const float* x, float* y, int64_t n, const uint32_t* __restrict__ slots,
                                                                     int n_samples, float* running2, uint32_t* __restrict__ slots_next,
                                                                     int num_bits) {
    __shared__ double sh[kBlock / kWave];
    float mn, mx;
    sample_mean(slots, n_samples, sh, mn, mx);
    const float r0 = running2[0], r1 = running2[1];
    const float lo = (mn < r0) ? mn : r0;            // Python min(running_min, v): keeps running_min unless v < it (NaN keeps it)
    const float hi = (mx > r1) ? mx : r1;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) { running2[0] = lo; running2[1] = hi; }
        for (int i = threadIdx.x; i < 2 * n_samples; i += kBlock) slots_next[i] = 0u;
    }
    const QParams p = qparams_double((double)lo, (double)hi, num_bits, 0);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    if (((((uintptr_t)x) | ((uintptr_t)y)) & 15u) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += stride) {
            const fvec4 v = (kQuantNt & 1) ? DFQ_NT_LOAD((const fvec4*)(x + 4 * i)) : *(const fvec4*)(x + 4 * i);
            fvec4 r;
            float code;
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = fake_quant_one(v[k], p, &code);
            if (kQuantNt & 2) DFQ_NT_STORE(r, (fvec4*)(y + 4 * i));
            else *(fvec4*)(y + 4 * i) = r;
        }
        for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
            float code;
            y[i] = fake_quant_one(x[i], p, &code);
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float code;
        y[i] = fake_quant_one(x[i], p, &code);
    }
}


// QuantMeasure.forward with update_stat in ONE launch (round 4).  config 5 runs 74 of these per distilled batch, most of them on
// small activations: two launches each meant 148 launch boundaries per batch, and the quantise pass of a mid-sized activation
// re-read from HBM what the extrema pass had just streamed.  Here a grid of persistent workgroups (sized to be co-resident:
// <= 4 per CU of a kernel that could run 8) does both: phase A, the per-sample extrema of its spans (device-scope atomicMax
// into the per-sample slots), an arrival on a monotonic counter, a bounded wait until the whole grid has arrived; phase B, the
// mean of the extrema read back with device-scope loads (the same fixed-order float64 sums in every workgroup), the running
// range, and the quantisation of its share of x -- which, up to the size of the L2s and the 256 MB Infinity Cache, is still
// on the chip.  Same numbers as the two-launch form (dfq_quant_measure), bit for bit.  All shared memory is dynamic (the
// kernel's workgroups wait for each other).
struct QmArgs {
    const float* x;
    float* y;
    int64_t n, sample_len, span;
    uint32_t* slots_cur;
    uint32_t* slots_next;
    float* running2;
    unsigned long long* sync;        // [0] arrivals (monotonic over the calls on this scratch), [1] error word
    unsigned long long target;       // arrivals when the whole grid of THIS call has arrived
    int32_t n_samples, spans_per_sample, num_bits, spin_limit;
};
constexpr size_t kQmSmem = 128;

__global__ __launch_bounds__(kBlock) void quant_measure_fused_kernel(QmArgs a) {
    DFQ_DYN_SMEM(smem);
    float* sh_mn = (float*)smem;                         // [kBlock / kWave]
    float* sh_mx = sh_mn + kBlock / kWave;
    double* sh_d = (double*)(smem + 32);                 // [kBlock / kWave]
    int* sh_ok = (int*)(smem + 96);
    const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
    // ---- phase A: extrema of this workgroup's spans ----
    const int items = a.n_samples * a.spans_per_sample;
    for (int it = blockIdx.x; it < items; it += gridDim.x) {
        const int smp = it / a.spans_per_sample;
        const int64_t b = (int64_t)(it - smp * a.spans_per_sample) * a.span;
        const int64_t e = (b + a.span < a.sample_len) ? b + a.span : a.sample_len;
        float mn = INFINITY, mx = -INFINITY;
        if (b < e) thread_minmax_range(a.x + (int64_t)smp * a.sample_len, b, e, mn, mx);
        mn = wave_min(mn);
        mx = wave_max(mx);
        if (lane == 0) { sh_mn[wave] = mn; sh_mx[wave] = mx; }
        __syncthreads();
        if (tid == 0) {
            float lo = sh_mn[0], hi = sh_mx[0];
#pragma unroll
            for (int w = 1; w < kBlock / kWave; ++w) { lo = fminf(lo, sh_mn[w]); hi = fmaxf(hi, sh_mx[w]); }
            if (lo <= hi) {   // false only if the range was empty
                atomicMax(a.slots_cur + 2 * smp + 0, ~enc_ord(lo));
                atomicMax(a.slots_cur + 2 * smp + 1, enc_ord(hi));
            }
        }
        __syncthreads();                                  // sh_mn / sh_mx are rewritten by the next item
    }
    // ---- the whole grid has published: arrival (after this workgroup's atomics have been performed), bounded wait ----
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
        atomicAdd(a.sync, 1ull);
        long spins = 0;
        int ok = 1;
        while (__hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.target) {
            __builtin_amdgcn_s_sleep(2);
            ++spins;
            if (spins > a.spin_limit ||
                ((spins & 255) == 0 && __hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull)) {
                atomicMax(a.sync + 1, 1ull);
                ok = 0;
                break;
            }
        }
        *sh_ok = ok;
    }
    __syncthreads();
    if (*sh_ok == 0) return;                             // abandoned: nothing is quantised, the status call reports it
    // ---- phase B: mean over samples of the per-sample extrema (float64, fixed order, rounded once: sample_mean) ----
    double smn = 0.0, smx = 0.0;
    for (int i = tid; i < a.n_samples; i += kBlock) {
        smn += (double)slot_min(__hip_atomic_load(a.slots_cur + 2 * i + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        smx += (double)slot_max(__hip_atomic_load(a.slots_cur + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    smn = block_sum(smn, sh_d);
    smx = block_sum(smx, sh_d);
    float mn, mx;
    if (a.n_samples == 1) {
        mn = slot_min(__hip_atomic_load(a.slots_cur + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        mx = slot_max(__hip_atomic_load(a.slots_cur + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    } else {
        mn = (float)(smn / (double)a.n_samples);
        mx = (float)(smx / (double)a.n_samples);
    }
    const float r0 = a.running2[0], r1 = a.running2[1];
    const float lo = (mn < r0) ? mn : r0;            // Python min(running_min, v): keeps running_min unless v < it (NaN keeps it)
    const float hi = (mx > r1) ? mx : r1;
    if (blockIdx.x == 0) {
        if (tid == 0) { a.running2[0] = lo; a.running2[1] = hi; }
        for (int i = tid; i < 2 * a.n_samples; i += kBlock) a.slots_next[i] = 0u;
    }
    const QParams p = qparams_double((double)lo, (double)hi, a.num_bits, 0);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const float* x = a.x;
    float* y = a.y;
    const int64_t n = a.n;
    if (((((uintptr_t)x) | ((uintptr_t)y)) & 15u) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * kBlock + tid; i < n4; i += stride) {
            const fvec4 v = *(const fvec4*)(x + 4 * i);   // (plain load: this is the second read, from the caches where it fits)
            fvec4 r;
            float code;
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = fake_quant_one(v[k], p, &code);
            if (kQuantNt & 2) DFQ_NT_STORE(r, (fvec4*)(y + 4 * i));
            else *(fvec4*)(y + 4 * i) = r;
        }
        for (int64_t i = 4 * n4 + (int64_t)blockIdx.x * kBlock + tid; i < n; i += stride) {
            float code;
            y[i] = fake_quant_one(x[i], p, &code);
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * kBlock + tid; i < n; i += stride) {
        float code;
        y[i] = fake_quant_one(x[i], p, &code);
    }
}

// ---- _quantize_error (dfq.py:8-25) ----------------------------------------------------------
__global__ __launch_bounds__(kBlock) void quant_error_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                             int64_t n, int num_bits, int symmetric,
                                                             const uint32_t* __restrict__ slot_pair) {
    const QParams p = qparams_double((double)slot_min(slot_pair[0]), (double)slot_max(slot_pair[1]), num_bits, symmetric);
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float code;
        const float v = x[i];
        out[i] = fake_quant_one(v, p, &code) - v;
    }
}

// rows of `row_len` elements; one wave per row (round robin); per-row value by `reduction`,
// accumulated per block into partial[blockIdx.x] (float64, fixed order).
__global__ __launch_bounds__(kBlock) void quant_error_rows_kernel(const float* __restrict__ x, int64_t n,
                                                                  int64_t rows, int64_t row_len, int num_bits,
                                                                  int symmetric, int reduction,
                                                                  const uint32_t* __restrict__ slot_pair,
                                                                  double* __restrict__ partial) {
    __shared__ double sh[kBlock / kWave];
    const QParams p = qparams_double((double)slot_min(slot_pair[0]), (double)slot_max(slot_pair[1]), num_bits, symmetric);
    const int lane = threadIdx.x % kWave;
    const int64_t wave_global = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    const int64_t n_waves = (int64_t)gridDim.x * (kBlock / kWave);
    double acc = 0.0;
    for (int64_t r = wave_global; r < rows; r += n_waves) {
        const int64_t b = r * row_len;
        const int64_t e = (b + row_len < n) ? b + row_len : n;
        double s = 0.0;
        for (int64_t i = b + lane; i < e; i += kWave) {
            float code;
            const float v = x[i];
            const float d = fake_quant_one(v, p, &code) - v;
            s += (reduction == 1) ? (double)fabsf(d) : (double)d;
        }
        s = wave_sum(s);
        if (reduction >= 3) s = fabs(s);
        acc += s;
    }
    const double t = block_sum((lane == 0) ? acc : 0.0, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

__global__ void quant_error_final_kernel(const double* __restrict__ partial, int n_partial, double denom,
                                         float* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < n_partial; ++i) t += partial[i];
        out[0] = (float)(t / denom);
    }
}

static int grid_for(int64_t n, int per_block, int cap) {
    int64_t g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

__global__ void seg_decode_kernel(const uint32_t* __restrict__ slots, int n_segs, float* __restrict__ minmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_segs) return;
    minmax[2 * i + 0] = slot_min(slots[2 * i + 0]);
    minmax[2 * i + 1] = slot_max(slots[2 * i + 1]);
}

}  // namespace dfq

using namespace dfq;

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int dfq_tensor_minmax(const float* x, int64_t n, float* out2, uint32_t* scratch2, void* stream) {
    if (!x || !out2 || !scratch2 || n <= 0) return fail_arg("dfq_tensor_minmax: bad argument (n=%lld)", (long long)n);
    hipStream_t st = as_stream(stream);
    DFQ_HIP_TRY(hipMemsetAsync(scratch2, 0, 2 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(minmax_kernel, dim3(grid_for(n, kChunk, 2048)), dim3(kBlock), 0, st, x, n, scratch2);
    DFQ_CHECK_LAUNCH();
    hipLaunchKernelGGL(slots_decode_kernel, dim3(1), dim3(64), 0, st, (const uint32_t*)scratch2, out2, 1);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_fake_quant(const float* x, float* y, int64_t n, int32_t num_bits, int32_t symmetric,
                   int32_t range_mode, double min_value, double max_value, const float* minmax_dev,
                   int32_t* codes, void* stream) {
    if (!x || !y || n < 0) return fail_arg("dfq_fake_quant: bad argument");
    if (num_bits < 1 || num_bits > 30) return fail_arg("dfq_fake_quant: num_bits=%d out of range", num_bits);
    if (range_mode < 0 || range_mode > 2) return fail_arg("dfq_fake_quant: range_mode=%d", range_mode);
    if (range_mode != 0 && !minmax_dev) return fail_arg("dfq_fake_quant: range_mode %d needs minmax_dev", range_mode);
    if (n == 0) return DFQ_OK;
    QParams p0 = qparams_double(min_value, max_value, num_bits, symmetric);
    hipLaunchKernelGGL(fake_quant_kernel, dim3(grid_for(n, kBlock * 8, 4096)), dim3(kBlock), 0, as_stream(stream),
                       x, y, n, p0, (int)num_bits, (int)symmetric, (int)range_mode, minmax_dev, codes);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_sample_minmax_mean(const float* x, int32_t n_samples, int64_t sample_len, float* out2,
                           float* running2, uint32_t* scratch, void* stream) {
    if (!x || !out2 || !scratch || n_samples <= 0 || sample_len <= 0)
        return fail_arg("dfq_sample_minmax_mean: bad argument");
    if (n_samples > 65535) return fail_arg("dfq_sample_minmax_mean: n_samples=%d > 65535", n_samples);
    hipStream_t st = as_stream(stream);
    DFQ_HIP_TRY(hipMemsetAsync(scratch, 0, 2 * sizeof(uint32_t) * (size_t)n_samples, st));
    // ~2048 workgroups in all (8 per CU), each a span that is a multiple of 1024 floats (16-byte vectors stay aligned)
    int64_t gx = std::max<int64_t>(1, 2048 / n_samples);
    int64_t span = (sample_len + gx - 1) / gx;
    span = std::max<int64_t>(4096, (span + 1023) / 1024 * 1024);
    gx = (sample_len + span - 1) / span;
    hipLaunchKernelGGL(sample_minmax_kernel, dim3((unsigned)gx, n_samples), dim3(kBlock), 0, st, x, sample_len, span, scratch);
    DFQ_CHECK_LAUNCH();
    hipLaunchKernelGGL(sample_mean_kernel, dim3(1), dim3(kBlock), 0, st, (const uint32_t*)scratch, (int)n_samples,
                       out2, running2);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_quant_measure(const float* x, float* y, int32_t n_samples, int64_t sample_len, int32_t num_bits, float* running2,
                      uint32_t* scratch, int32_t parity, void* stream) {
    if (!x || !y || !running2 || !scratch || n_samples <= 0 || sample_len <= 0 || (parity != 0 && parity != 1))
        return fail_arg("dfq_quant_measure: bad argument");
    if (n_samples > 65535) return fail_arg("dfq_quant_measure: n_samples=%d > 65535", n_samples);
    if (num_bits < 1 || num_bits > 30) return fail_arg("dfq_quant_measure: num_bits=%d out of range", num_bits);
    hipStream_t st = as_stream(stream);
    uint32_t* cur = scratch + (size_t)parity * 2 * n_samples;
    uint32_t* nxt = scratch + (size_t)(parity ^ 1) * 2 * n_samples;
    int64_t gx = std::max<int64_t>(1, 2048 / n_samples);
    int64_t span = (sample_len + gx - 1) / gx;
    span = std::max<int64_t>(4096, (span + 1023) / 1024 * 1024);
    gx = (sample_len + span - 1) / span;
    hipLaunchKernelGGL(sample_minmax_kernel, dim3((unsigned)gx, n_samples), dim3(kBlock), 0, st, x, sample_len, span, cur);
    DFQ_CHECK_LAUNCH();
    const int64_t n = (int64_t)n_samples * sample_len;
    hipLaunchKernelGGL(measured_fake_quant_kernel, dim3(grid_for(n, kBlock * 8, 4096)), dim3(kBlock), 0, st, x, y, n,
                       (const uint32_t*)cur, (int)n_samples, running2, nxt, (int)num_bits);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}


static void qm_split(int32_t n_samples, int64_t sample_len, int64_t* span_out, int64_t* spans_out) {
    int64_t gx = std::max<int64_t>(1, 2048 / n_samples);
    int64_t span = (sample_len + gx - 1) / gx;
    span = std::max<int64_t>(4096, (span + 1023) / 1024 * 1024);
    *span_out = span;
    *spans_out = (sample_len + span - 1) / span;
}

int32_t dfq_quant_measure_fused_grid(int32_t n_samples, int64_t sample_len) {
    if (n_samples <= 0 || sample_len <= 0) return 0;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                  ? prop.multiProcessorCount : 256;
    }
    int64_t span, spans;
    qm_split(n_samples, sample_len, &span, &spans);
    const int64_t n = (int64_t)n_samples * sample_len;
    const int64_t want = std::max<int64_t>(spans * n_samples, grid_for(n, kBlock * 8, 4096));
    return (int32_t)std::min<int64_t>(want, 4 * (int64_t)cus);     // co-resident: the kernel's occupancy limit is 8 workgroups per CU
}

int dfq_quant_measure_fused(const float* x, float* y, int32_t n_samples, int64_t sample_len, int32_t num_bits, float* running2,
                            uint32_t* scratch, int32_t parity, int64_t arrivals_before, void* stream) {
    if (!x || !y || !running2 || !scratch || n_samples <= 0 || sample_len <= 0 || (parity != 0 && parity != 1) || arrivals_before < 0)
        return fail_arg("dfq_quant_measure_fused: bad argument");
    if (n_samples > 65535) return fail_arg("dfq_quant_measure_fused: n_samples=%d > 65535", n_samples);
    if (num_bits < 1 || num_bits > 30) return fail_arg("dfq_quant_measure_fused: num_bits=%d out of range", num_bits);
    if ((uintptr_t)(scratch + 4 * (size_t)n_samples) & 7u) return fail_arg("dfq_quant_measure_fused: scratch must be 8-byte aligned");
    hipStream_t st = as_stream(stream);
    QmArgs a;
    a.x = x; a.y = y; a.n = (int64_t)n_samples * sample_len; a.sample_len = sample_len;
    int64_t spans;
    qm_split(n_samples, sample_len, &a.span, &spans);
    a.spans_per_sample = (int32_t)spans;
    a.slots_cur = scratch + (size_t)parity * 2 * n_samples;
    a.slots_next = scratch + (size_t)(parity ^ 1) * 2 * n_samples;
    a.running2 = running2;
    a.sync = (unsigned long long*)(scratch + 4 * (size_t)n_samples);
    const int grid = dfq_quant_measure_fused_grid(n_samples, sample_len);
    a.target = (unsigned long long)arrivals_before + (unsigned long long)grid;
    a.n_samples = n_samples; a.num_bits = num_bits;
    a.spin_limit = dfq::spin_limit_from_env(4000000);
    SpinGuard guard(st);             // workgroups of this launch wait for each other: never next to another such kernel
    DFQ_LAUNCH_SPINNING(quant_measure_fused_kernel, dim3(grid), dim3(kBlock), kQmSmem, st, a);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_quant_measure_fused_status(const uint32_t* scratch, int32_t n_samples, void* stream) {
    if (!scratch || n_samples <= 0) return fail_arg("dfq_quant_measure_fused_status: bad argument");
    hipStream_t st = as_stream(stream);
    unsigned long long err = 0;
    DFQ_HIP_TRY(hipMemcpyAsync(&err, (const unsigned long long*)(scratch + 4 * (size_t)n_samples) + 1, sizeof(err), hipMemcpyDeviceToHost, st));
    DFQ_HIP_TRY(hipStreamSynchronize(st));
    if (err) { set_error("dfq_quant_measure_fused: a workgroup gave up waiting for the rest of its grid (the output is invalid)"); return DFQ_ERR_ABANDONED; }
    return DFQ_OK;
}

// ---- multi-tensor plan ------------------------------------------------------------------------
struct dfq_quant_plan {
    int n_segs = 0;
    int n_blocks = 0;
    SegDev* d_segs = nullptr;
    int32_t* d_block_begin = nullptr;
    uint32_t* d_slots = nullptr;
    float* d_minmax = nullptr;
};

int dfq_quant_plan_create(const dfq_segment* segs, int32_t n_segs, dfq_quant_plan** out_plan) {
    if (!segs || n_segs <= 0 || !out_plan) return fail_arg("dfq_quant_plan_create: bad argument");
    std::vector<SegDev> h(n_segs);
    std::vector<int32_t> bb(n_segs + 1);
    int64_t blocks = 0;
    for (int i = 0; i < n_segs; ++i) {
        if (!segs[i].data || segs[i].n <= 0) return fail_arg("dfq_quant_plan_create: segment %d is empty", i);
        if (segs[i].num_bits < 1 || segs[i].num_bits > 30) return fail_arg("dfq_quant_plan_create: segment %d bits", i);
        h[i].data = segs[i].data;
        h[i].codes = segs[i].codes;
        h[i].n = segs[i].n;
        h[i].num_bits = segs[i].num_bits;
        h[i].symmetric = segs[i].symmetric;
        bb[i] = (int32_t)blocks;
        blocks += (segs[i].n + kChunk - 1) / kChunk;
        if (blocks > 0x7fffffff) return fail_arg("dfq_quant_plan_create: too many elements");
    }
    bb[n_segs] = (int32_t)blocks;
    dfq_quant_plan* p = new dfq_quant_plan();
    p->n_segs = n_segs;
    p->n_blocks = (int)blocks;
    hipError_t e;
    if ((e = dfq::dev_malloc((void**)&p->d_segs, sizeof(SegDev) * n_segs)) != hipSuccess ||
        (e = dfq::dev_malloc((void**)&p->d_block_begin, sizeof(int32_t) * (n_segs + 1))) != hipSuccess ||
        (e = dfq::dev_malloc((void**)&p->d_slots, sizeof(uint32_t) * 2 * n_segs)) != hipSuccess ||
        (e = dfq::dev_malloc((void**)&p->d_minmax, sizeof(float) * 2 * n_segs)) != hipSuccess ||
        (e = hipMemcpy(p->d_segs, h.data(), sizeof(SegDev) * n_segs, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->d_block_begin, bb.data(), sizeof(int32_t) * (n_segs + 1), hipMemcpyHostToDevice)) != hipSuccess) {
        dfq_quant_plan_destroy(p);
        return fail_hip(e, "quant plan allocation", __FILE__, __LINE__);
    }
    *out_plan = p;
    return DFQ_OK;
}

void dfq_quant_plan_destroy(dfq_quant_plan* p) {
    if (!p) return;
    dfq::dev_quiesce();                                  // nothing in flight may still use the blocks released below
    if (p->d_segs) dfq::dev_free(p->d_segs);
    if (p->d_block_begin) dfq::dev_free(p->d_block_begin);
    if (p->d_slots) dfq::dev_free(p->d_slots);
    if (p->d_minmax) dfq::dev_free(p->d_minmax);
    delete p;
}

int dfq_quant_plan_run(dfq_quant_plan* p, void* stream) {
    if (!p) return fail_arg("dfq_quant_plan_run: null plan");
    hipStream_t st = as_stream(stream);
    DFQ_HIP_TRY(hipMemsetAsync(p->d_slots, 0, sizeof(uint32_t) * 2 * p->n_segs, st));
    hipLaunchKernelGGL(seg_minmax_kernel, dim3(p->n_blocks), dim3(kBlock), 0, st, (const SegDev*)p->d_segs,
                       (const int32_t*)p->d_block_begin, p->n_segs, p->d_slots);
    DFQ_CHECK_LAUNCH();
    hipLaunchKernelGGL(seg_fake_quant_kernel, dim3(p->n_blocks), dim3(kBlock), 0, st, (const SegDev*)p->d_segs,
                       (const int32_t*)p->d_block_begin, p->n_segs, (const uint32_t*)p->d_slots, p->d_minmax);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

// min/max only: the first launch of dfq_quant_plan_run plus a decode of the slots (calibration-table writers need
// the ranges of the weights, not quantised weights)
int dfq_quant_plan_measure(dfq_quant_plan* p, void* stream) {
    if (!p) return fail_arg("dfq_quant_plan_measure: null plan");
    hipStream_t st = as_stream(stream);
    DFQ_HIP_TRY(hipMemsetAsync(p->d_slots, 0, sizeof(uint32_t) * 2 * p->n_segs, st));
    hipLaunchKernelGGL(seg_minmax_kernel, dim3(p->n_blocks), dim3(kBlock), 0, st, (const SegDev*)p->d_segs,
                       (const int32_t*)p->d_block_begin, p->n_segs, p->d_slots);
    DFQ_CHECK_LAUNCH();
    hipLaunchKernelGGL(seg_decode_kernel, dim3((p->n_segs + kBlock - 1) / kBlock), dim3(kBlock), 0, st,
                       (const uint32_t*)p->d_slots, p->n_segs, p->d_minmax);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

const float* dfq_quant_plan_minmax(const dfq_quant_plan* p) { return p ? p->d_minmax : nullptr; }

// ---- _quantize_error ---------------------------------------------------------------------------
static const int kQerrPartials = 1024;

size_t dfq_quant_error_scratch_bytes(int64_t, int64_t) { return 16 + sizeof(double) * kQerrPartials; }

int dfq_quant_error(const float* x, int64_t n, int64_t rows, int32_t num_bits, int32_t symmetric,
                    int32_t reduction, float* out, void* scratch, void* stream) {
    if (!x || !out || !scratch || n <= 0) return fail_arg("dfq_quant_error: bad argument");
    if (reduction < 0 || reduction > 4) return fail_arg("dfq_quant_error: reduction=%d", reduction);
    hipStream_t st = as_stream(stream);
    uint32_t* slots = reinterpret_cast<uint32_t*>(scratch);
    double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(scratch) + 16);
    DFQ_HIP_TRY(hipMemsetAsync(slots, 0, 2 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(minmax_kernel, dim3(grid_for(n, kChunk, 2048)), dim3(kBlock), 0, st, x, n, slots);
    DFQ_CHECK_LAUNCH();
    if (reduction == 0) {
        hipLaunchKernelGGL(quant_error_kernel, dim3(grid_for(n, kBlock * 8, 4096)), dim3(kBlock), 0, st, x, out, n,
                           (int)num_bits, (int)symmetric, (const uint32_t*)slots);
        DFQ_CHECK_LAUNCH();
        return DFQ_OK;
    }
    int64_t r = rows, row_len;
    if (reduction <= 2) {   // scalar over the whole tensor: rows are just work units
        row_len = 4096;
        r = (n + row_len - 1) / row_len;
    } else {
        if (rows <= 0 || n % rows != 0) return fail_arg("dfq_quant_error: rows=%lld does not divide n=%lld", (long long)rows, (long long)n);
        row_len = n / rows;
    }
    const int grid = grid_for(r, kBlock / kWave, kQerrPartials);
    hipLaunchKernelGGL(quant_error_rows_kernel, dim3(grid), dim3(kBlock), 0, st, x, n, r, row_len, (int)num_bits,
                       (int)symmetric, (int)reduction, (const uint32_t*)slots, partial);
    DFQ_CHECK_LAUNCH();
    hipLaunchKernelGGL(quant_error_final_kernel, dim3(1), dim3(64), 0, st, (const double*)partial, grid,
                       reduction == 2 ? (double)n : 1.0, out);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

}  // extern "C"
