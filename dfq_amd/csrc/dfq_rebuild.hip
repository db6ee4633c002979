// Batched rebuild of a network's paired tensors from their pristine values and the cumulative per-relation scale
// vectors (utils/relation.py:20-24):  W = diag(S_out) . W0 . diag(1 / S_in),  b = b0 . S_out,  BN proxies likewise.
//
// This is the replicated write of the sharded equalisation (dfq_amd/sharded.py): after the all_gather of
// the scale vectors EVERY rank -- the owner of a component included -- runs this one launch on identical inputs
// (W0, gathered S), so all ranks end with bit-identical tensors, whatever the number of ranks.  Per element two
// separately rounded float32 operations in a fixed order:  t = fl(w0 * s_out[o]);  w = fl(t / s_in[ch])  (or fl(t * r_in[ch])
// when the caller holds reciprocals: the lazy-scale equalisation, dfq_le_lazy.hip).
//
// One launch covers every item of the plan: a workgroup owns a span of kSpan consecutive elements of one item
// (16-byte vectors when the item's rows allow it); read once, written once -> 8 B per element.
#include <vector>

#include "dfq_common.hpp"

namespace dfq {

constexpr int kSpan = 4096;   // elements per workgroup: 256 threads x 4 vectors of 4

struct RebuildItemDev {
    const float* src;
    float* dst;
    const float* s_out;
    const float* s_in;
    int64_t n;             // elements
    int32_t row_len;       // cols * khkw
    int32_t khkw;
    int32_t cols;          // inputs per group
    int32_t out_per_group;
    int32_t in_mul;        // s_in holds reciprocals: multiply
    int32_t pad;
};

struct RebuildBlock {
    int32_t item;
    int32_t pad;
    int64_t first;         // first element of the span
};

__device__ __forceinline__ float rebuild_one(float w, int64_t e, const RebuildItemDev& it) {
    const int64_t o = e / it.row_len;
    if (it.s_out) w = w * it.s_out[o];
    if (it.s_in) {
        const int r = (int)(e - o * it.row_len);
        const int i = r / it.khkw;
        const int ch = (int)(o / it.out_per_group) * it.cols + i;
        w = it.in_mul ? w * it.s_in[ch] : w / it.s_in[ch];
    }
    return w;
}

__global__ __launch_bounds__(kBlock) void rebuild_kernel(const RebuildItemDev* __restrict__ items,
                                                         const RebuildBlock* __restrict__ blocks) {
    const RebuildBlock blk = blocks[blockIdx.x];
    const RebuildItemDev it = items[blk.item];
    const int64_t end = (blk.first + kSpan < it.n) ? blk.first + kSpan : it.n;
    const bool vec = ((it.row_len & 3) == 0) && ((((uintptr_t)it.src | (uintptr_t)it.dst) & 15) == 0);
    if (vec) {
        // a vector never crosses a row (row_len % 4 == 0, spans start at multiples of 4): one output channel per vector
        fvec4 v[4];
        int64_t e0[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            e0[k] = blk.first + 4 * ((int64_t)k * kBlock + threadIdx.x);
            if (e0[k] < end) v[k] = *(const fvec4*)(it.src + e0[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (e0[k] >= end) continue;
            const int64_t o = e0[k] / it.row_len;
            const int r = (int)(e0[k] - o * it.row_len);
            fvec4 x = v[k];
            if (it.s_out) {
                const float so = it.s_out[o];
                x[0] = x[0] * so; x[1] = x[1] * so; x[2] = x[2] * so; x[3] = x[3] * so;
            }
            if (it.s_in) {
                const float* sg = it.s_in + (int64_t)(o / it.out_per_group) * it.cols;
                float f[4];
                if (it.khkw == 1) { f[0] = sg[r]; f[1] = sg[r + 1]; f[2] = sg[r + 2]; f[3] = sg[r + 3]; }
                else { f[0] = sg[r / it.khkw]; f[1] = sg[(r + 1) / it.khkw]; f[2] = sg[(r + 2) / it.khkw]; f[3] = sg[(r + 3) / it.khkw]; }
                if (it.in_mul) { x[0] = x[0] * f[0]; x[1] = x[1] * f[1]; x[2] = x[2] * f[2]; x[3] = x[3] * f[3]; }
                else { x[0] = x[0] / f[0]; x[1] = x[1] / f[1]; x[2] = x[2] / f[2]; x[3] = x[3] / f[3]; }
            }
            *(fvec4*)(it.dst + e0[k]) = x;
        }
    } else {
        for (int64_t e = blk.first + threadIdx.x; e < end; e += kBlock) it.dst[e] = rebuild_one(it.src[e], e, it);
    }
}

struct RebuildPlan {
    RebuildItemDev* d_items = nullptr;
    RebuildBlock* d_blocks = nullptr;
    int n_blocks = 0;
    int64_t elements = 0;
};

}  // namespace dfq

using namespace dfq;

extern "C" {

int dfq_rebuild_plan_create(const dfq_rebuild_item* items, int32_t n_items, dfq_rebuild_plan** out_plan) {
    if (!items || n_items <= 0 || !out_plan) return fail_arg("dfq_rebuild_plan_create: bad argument");
    std::vector<RebuildItemDev> dev(n_items);
    std::vector<RebuildBlock> blocks;
    int64_t total = 0;
    for (int i = 0; i < n_items; ++i) {
        const dfq_rebuild_item& a = items[i];
        if (!a.src || !a.dst || a.rows <= 0 || a.cols <= 0 || a.khkw <= 0 || a.groups <= 0 || a.rows % a.groups != 0)
            return fail_arg("dfq_rebuild_plan_create: item %d: bad geometry", i);
        const int64_t row_len = (int64_t)a.cols * a.khkw;
        if (row_len > 0x7fffffff) return fail_arg("dfq_rebuild_plan_create: item %d: row too long", i);
        RebuildItemDev& d = dev[i];
        d.src = a.src; d.dst = a.dst; d.s_out = a.s_out; d.s_in = a.s_in;
        d.n = (int64_t)a.rows * row_len;
        d.row_len = (int32_t)row_len; d.khkw = a.khkw; d.cols = a.cols; d.out_per_group = a.rows / a.groups;
        d.in_mul = a.in_reciprocal ? 1 : 0; d.pad = 0;
        for (int64_t f = 0; f < d.n; f += kSpan) blocks.push_back(RebuildBlock{i, 0, f});
        total += d.n;
    }
    if (blocks.size() > 0x7fffffffu) return fail_arg("dfq_rebuild_plan_create: too many elements");
    RebuildPlan* p = new RebuildPlan();
    p->n_blocks = (int)blocks.size();
    p->elements = total;
    hipError_t e = dfq::dev_malloc((void**)&p->d_items, dev.size() * sizeof(RebuildItemDev));
    if (e == hipSuccess) e = dfq::dev_malloc((void**)&p->d_blocks, blocks.size() * sizeof(RebuildBlock));
    if (e == hipSuccess) e = hipMemcpy(p->d_items, dev.data(), dev.size() * sizeof(RebuildItemDev), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_blocks, blocks.data(), blocks.size() * sizeof(RebuildBlock), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        dfq_rebuild_plan_destroy((dfq_rebuild_plan*)p);
        return fail_hip(e, "rebuild plan tables", __FILE__, __LINE__);
    }
    *out_plan = (dfq_rebuild_plan*)p;
    return DFQ_OK;
}

void dfq_rebuild_plan_destroy(dfq_rebuild_plan* plan) {
    RebuildPlan* p = (RebuildPlan*)plan;
    if (!p) return;
    dfq::dev_quiesce();                                  // nothing in flight may still use the blocks released below
    if (p->d_items) dfq::dev_free(p->d_items);
    if (p->d_blocks) dfq::dev_free(p->d_blocks);
    delete p;
}

int64_t dfq_rebuild_plan_elements(const dfq_rebuild_plan* plan) { return plan ? ((const RebuildPlan*)plan)->elements : 0; }

int dfq_rebuild_plan_run(dfq_rebuild_plan* plan, void* stream) {
    RebuildPlan* p = (RebuildPlan*)plan;
    if (!p) return fail_arg("dfq_rebuild_plan_run: null plan");
    hipLaunchKernelGGL(rebuild_kernel, dim3(p->n_blocks), dim3(kBlock), 0, as_stream(stream),
                       (const RebuildItemDev*)p->d_items, (const RebuildBlock*)p->d_blocks);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

}  // extern "C"
