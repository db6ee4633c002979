// Cross-layer weight equalisation (dfq.py:28-75 _layer_equalization, dfq.py:78-117 sweep loop)
// for gfx950.
//
// Work decomposition
//   relation  = (first layer W1 [O1, row_len], second layer W2 [O2, I2/g, khkw]);  channel c of
//               W1 pairs with input channel ii = c % gi of group g = c / gi of W2 (dfq.py:29-46).
//   tile      = `tc` consecutive paired channels of one relation = one 256-thread workgroup.
//               The W1 side of a tile is ONE contiguous run of tc*row_len floats, the W2 side is
//               `go` runs of tc*khkw floats (contiguous across consecutive channels of a group),
//               so both sides are read with unit-stride lanes.
//   level     = set of relations that share no layer (Gauss-Seidel order of dfq.py:85 kept
//               between levels) = one kernel launch.
// A tile does everything for its channels in one launch: ranges of the rows and of the strided
// columns (order-preserving LDS atomics), the scale solve with the reference's Python clamp
// semantics, the in-place rescale of W1 rows / b1 / BN proxies / W2 columns, the cumulative S, and
// the convergence statistic sum|W - W_prev| as a per-tile float64 partial.  Each paired element is
// read twice (second read is an L1/L2 hit) and written once.
//
// Convergence bookkeeping (dfq.py:105-115): a layer touched once per sweep needs no snapshot
// (|new - old| is known in registers); a layer touched twice (second of one relation, first of the
// next) stores its pre-sweep value into a snapshot arena at the first touch and reads it back at
// the last.  Untouched layers contribute exactly 0.  A single-workgroup control kernel reduces the
// partials per layer in a fixed order, forms mean -> float32 -> float64 sum in graph order and
// advances the reference's (diff, count) state machine on the device; every level kernel starts
// with a uniform load of `done` and exits if the loop has ended, so the host can enqueue sweeps
// ahead without synchronising.
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "dfq_common.hpp"

namespace dfq {

constexpr int kTcMax = 64;      // max paired channels per tile (LDS arrays below)

enum DiffMode : int32_t { DIFF_DIRECT = 0, DIFF_SAVE = 1, DIFF_FROM_PREV = 2, DIFF_NONE = 3 };

struct LeRelDev {
    float* w1;
    float* w2;
    float* b1;
    float* bnw;
    float* bnb;
    float* s_cum;
    float* prev1;    // snapshot arena of the first layer (same indexing as w1) or null
    float* prev2;    // snapshot arena of the second layer or null
    int32_t o1, row_len;
    int32_t gi, go, i2g, khkw;
    int32_t tc, n_tiles;
    int32_t diff1, diff2;
    int32_t partial_base;   // first tile slot of this relation in the partial array
    int32_t tile_begin;     // first workgroup of this relation inside its level launch
};

struct LeParams {
    float s_lo, s_hi, inv_lo, inv_hi, eps;
    int32_t hi_gt_lo, signed_range;
};

struct LeState {
    double diff;
    double last_diff_tmp;
    int32_t count;
    int32_t sweeps;
    int32_t done;
    int32_t pad;
};

struct LeLayerDiff {
    int32_t partial_begin;   // tile slot of the first partial of the diff-producing touch, -1: untouched
    int32_t n_partials;
    int32_t side;            // 0: row side of that relation, 1: column side
    int32_t pad;
    double n_elems;
};

// dfq.py:58-59 with Python's max/min semantics on a 0-dim float32 tensor (see oracle.le_solve).
__device__ __forceinline__ void le_solve(float r1, float r2, const LeParams& p, float& s_out, float& inv_out) {
    const float a = r1 + p.eps;
    const float recip = 1.0f / a;
    const float prod = r1 * r2;
    const float rad = prod + p.eps;
    const float root = sqrtf(rad);
    const float s = recip * root;
    const bool keep_hi = s < p.s_hi;                 // False for NaN -> hi
    const float t = keep_hi ? s : p.s_hi;
    const bool keep_lo = keep_hi ? (t > p.s_lo) : (p.hi_gt_lo != 0);
    s_out = keep_lo ? t : p.s_lo;
    inv_out = keep_lo ? (keep_hi ? (1.0f / s_out) : p.inv_hi) : p.inv_lo;
}

__device__ __forceinline__ float range_of(uint32_t mn_slot, uint32_t mx_slot, int signed_range) {
    const float mn = slot_min(mn_slot);
    const float mx = slot_max(mx_slot);
    if (signed_range) return fmaxf(fabsf(mn), fabsf(mx));
    return mx - mn;
}

__global__ __launch_bounds__(kBlock) void le_level_kernel(const LeRelDev* __restrict__ rels, int n_rels,
                                                          LeParams p, const LeState* __restrict__ state,
                                                          double* __restrict__ partials) {
    if (state->done) return;   // wave-uniform: the reference loop has already exited

    __shared__ uint32_t sh_mn1[kTcMax], sh_mx1[kTcMax], sh_mn2[kTcMax], sh_mx2[kTcMax];
    __shared__ float sh_s[kTcMax], sh_inv[kTcMax];
    __shared__ double sh_red[kBlock / kWave];

    int r = 0;
    while (r + 1 < n_rels && (int)blockIdx.x >= rels[r + 1].tile_begin) ++r;
    const LeRelDev R = rels[r];
    const int tid = threadIdx.x;
    const int tile = blockIdx.x - R.tile_begin;
    const int c0 = tile * R.tc;
    const int nc = (R.o1 - c0 < R.tc) ? (R.o1 - c0) : R.tc;

    if (tid < nc) {
        sh_mn1[tid] = 0u; sh_mx1[tid] = 0u; sh_mn2[tid] = 0u; sh_mx2[tid] = 0u;
    }
    __syncthreads();

    // ---- W1 side: one contiguous run of nc*row_len floats; lanes unit-stride ---------------------
    float* const rows = R.w1 + (int64_t)c0 * R.row_len;
    const int row_total = nc * R.row_len;
    const int q0 = tid / R.row_len;            // channel (within the tile) of this thread's first element
    const int rem0 = tid - q0 * R.row_len;
    const int dq = kBlock / R.row_len;
    const int dr = kBlock - dq * R.row_len;
    {
        int q = q0, rem = rem0, cur = q0;
        float mn = INFINITY, mx = -INFINITY;
        for (int e = tid; e < row_total; e += kBlock) {
            if (q != cur) {
                atomicMax(&sh_mn1[cur], ~enc_ord(mn));
                atomicMax(&sh_mx1[cur], enc_ord(mx));
                mn = INFINITY; mx = -INFINITY; cur = q;
            }
            const float v = rows[e];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
            q += dq; rem += dr;
            if (rem >= R.row_len) { rem -= R.row_len; ++q; }
        }
        if (mn <= mx) {
            atomicMax(&sh_mn1[cur], ~enc_ord(mn));
            atomicMax(&sh_mx1[cur], enc_ord(mx));
        }
    }

    // ---- W2 side: positions p = (channel-in-tile, k) are contiguous in memory within a group;
    //      thread -> (row lane jl, position p); rows j = jl, jl+JL, ... ------------------------------
    const int P = nc * R.khkw;
    const int JL = (P >= kBlock) ? 1 : (kBlock / P);
    const int jl = (P >= kBlock) ? 0 : (tid / P);
    const int p_first = (P >= kBlock) ? tid : (tid - jl * P);
    const bool col_active = jl < JL;
    const int64_t col_stride = (int64_t)R.i2g * R.khkw;
    if (col_active) {
        for (int pp = p_first; pp < P; pp += kBlock) {
            const int ct = pp / R.khkw;
            const int k = pp - ct * R.khkw;
            const int c = c0 + ct;
            const int g = c / R.gi;
            const int ii = c - g * R.gi;
            const float* col = R.w2 + ((int64_t)g * R.go * R.i2g + ii) * R.khkw + k;
            float mn = INFINITY, mx = -INFINITY;
            for (int j = jl; j < R.go; j += JL) {
                const float v = col[(int64_t)j * col_stride];
                mn = fminf(mn, v);
                mx = fmaxf(mx, v);
            }
            if (mn <= mx) {
                atomicMax(&sh_mn2[ct], ~enc_ord(mn));
                atomicMax(&sh_mx2[ct], enc_ord(mx));
            }
        }
    }
    __syncthreads();

    // ---- solve + per-channel vectors ---------------------------------------------------------------
    if (tid < nc) {
        const float r1 = range_of(sh_mn1[tid], sh_mx1[tid], p.signed_range);
        const float r2 = range_of(sh_mn2[tid], sh_mx2[tid], p.signed_range);
        float s, inv;
        le_solve(r1, r2, p, s, inv);
        sh_s[tid] = s;
        sh_inv[tid] = inv;
        const int c = c0 + tid;
        R.s_cum[c] = R.s_cum[c] * s;                  // Relation.set_scale_vec (relation.py:20-24)
        if (R.bnw) R.bnw[c] = R.bnw[c] * s;           // dfq.py:64-65
        if (R.bnb) R.bnb[c] = R.bnb[c] * s;           // dfq.py:67-68
        if (R.b1) R.b1[c] = R.b1[c] * s;              // dfq.py:70-71
    }
    __syncthreads();

    // ---- apply: W1 rows *= s (dfq.py:62) ------------------------------------------------------------
    double acc1 = 0.0, acc2 = 0.0;
    {
        float* const prev = R.prev1 ? R.prev1 + (int64_t)c0 * R.row_len : nullptr;
        int q = q0, rem = rem0;
        for (int e = tid; e < row_total; e += kBlock) {
            const float v = rows[e];
            const float nv = v * sh_s[q];
            rows[e] = nv;
            if (R.diff1 == DIFF_DIRECT) acc1 += (double)fabsf(nv - v);
            else if (R.diff1 == DIFF_SAVE) prev[e] = v;
            else if (R.diff1 == DIFF_FROM_PREV) acc1 += (double)fabsf(nv - prev[e]);
            q += dq; rem += dr;
            if (rem >= R.row_len) { rem -= R.row_len; ++q; }
        }
    }
    // ---- apply: W2 columns *= 1/s (dfq.py:73) ------------------------------------------------------
    if (col_active) {
        for (int pp = p_first; pp < P; pp += kBlock) {
            const int ct = pp / R.khkw;
            const int k = pp - ct * R.khkw;
            const int c = c0 + ct;
            const int g = c / R.gi;
            const int ii = c - g * R.gi;
            const int64_t base = ((int64_t)g * R.go * R.i2g + ii) * R.khkw + k;
            float* const col = R.w2 + base;
            float* const prev = R.prev2 ? R.prev2 + base : nullptr;
            const float inv = sh_inv[ct];
            for (int j = jl; j < R.go; j += JL) {
                const int64_t off = (int64_t)j * col_stride;
                const float v = col[off];
                const float nv = v * inv;
                col[off] = nv;
                if (R.diff2 == DIFF_DIRECT) acc2 += (double)fabsf(nv - v);
                else if (R.diff2 == DIFF_SAVE) prev[off] = v;
                else if (R.diff2 == DIFF_FROM_PREV) acc2 += (double)fabsf(nv - prev[off]);
            }
        }
    }
    // ---- per-tile partials of sum|W - W_prev| (fixed reduction order -> deterministic) ---------------
    const double t1 = block_sum(acc1, sh_red);
    const double t2 = block_sum(acc2, sh_red);
    if (tid == 0) {
        partials[2 * (int64_t)(R.partial_base + tile) + 0] = t1;
        partials[2 * (int64_t)(R.partial_base + tile) + 1] = t2;
    }
}

// dfq.py:105-115 on the device.  One workgroup; wave w reduces layers w, w+4, ...
__global__ __launch_bounds__(kBlock) void le_control_kernel(const LeLayerDiff* __restrict__ layers, int n_layers,
                                                            const double* __restrict__ partials,
                                                            double* __restrict__ layer_mean,
                                                            LeState* __restrict__ state, double converge_thres,
                                                            int converge_count, int max_sweeps) {
    if (state->done) return;
    const int lane = threadIdx.x % kWave;
    const int wave = threadIdx.x / kWave;
    for (int l = wave; l < n_layers; l += kBlock / kWave) {
        const LeLayerDiff L = layers[l];
        double s = 0.0;
        if (L.partial_begin >= 0) {
            for (int i = lane; i < L.n_partials; i += kWave) s += partials[2 * (int64_t)(L.partial_begin + i) + L.side];
            s = wave_sum(s);
        }
        if (lane == 0) {
            // float(torch.mean(torch.abs(W - W_prev))): float32 mean, widened to double (dfq.py:108)
            layer_mean[l] = (L.partial_begin >= 0) ? (double)(float)(s / L.n_elems) : 0.0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double diff_tmp = 0.0;
        for (int l = 0; l < n_layers; ++l) diff_tmp += layer_mean[l];   // graph order
        double diff = state->diff;
        int count = state->count;
        if (fabs(diff - diff_tmp) > 1e-9) { count = 0; diff = diff_tmp; }
        else { count += 1; }
        const int sweeps = state->sweeps + 1;
        const bool go_on = (diff > converge_thres) && (count < converge_count) &&
                           (max_sweeps < 0 || sweeps < max_sweeps);
        state->diff = diff;
        state->count = count;
        state->sweeps = sweeps;
        state->last_diff_tmp = diff_tmp;
        state->done = go_on ? 0 : 1;
    }
}

__global__ void le_reset_kernel(LeState* state, double converge_thres, int converge_count, int max_sweeps) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        state->diff = 10.0;          // dfq.py:81
        state->count = 0;            // dfq.py:82
        state->sweeps = 0;
        state->last_diff_tmp = 0.0;
        const bool go_on = (10.0 > converge_thres) && (0 < converge_count) && (max_sweeps != 0);
        state->done = go_on ? 0 : 1;
    }
}

// ---------------------------------------------------------------------------------------------
// host side: the plan
// ---------------------------------------------------------------------------------------------
struct LevelLaunch {
    int rel_begin = 0;      // range in the level-sorted device relation table
    int n_rels = 0;
    int n_blocks = 0;
    int64_t paired = 0;     // elements n1+n2 of the relations in this level
    int64_t snapshot = 0;   // snapshot-arena elements written or read in this level
};

}  // namespace dfq

using namespace dfq;

struct dfq_le_plan {
    int n_layers = 0, n_rels = 0;
    std::vector<LevelLaunch> levels;
    int64_t paired_total = 0, snapshot_total = 0;
    int total_tiles = 0;
    LeRelDev* d_rels = nullptr;
    LeLayerDiff* d_layer_diff = nullptr;
    double* d_partials = nullptr;
    double* d_layer_mean = nullptr;
    LeState* d_state = nullptr;
    std::vector<float*> arenas;   // snapshot arenas (hipMalloc)
};

static int pick_tc(int64_t per_channel, int o1, int khkw2) {
    static const int target = []() {
        const char* e = getenv("DFQ_LE_TILE_ELEMS");
        const int v = e ? atoi(e) : 0;
        return v > 0 ? v : 8192;
    }();
    int64_t tc = target / (per_channel > 0 ? per_channel : 1);
    if (tc < 1) tc = 1;
    if (khkw2 == 1 && tc < 8) tc = 8;      // 1x1 second layer: keep >= 32 B of each W2 row per tile
    if (tc > kTcMax) tc = kTcMax;
    if (tc > o1) tc = o1;
    return (int)tc;
}

extern "C" {

void dfq_le_plan_destroy(dfq_le_plan* p) {
    if (!p) return;
    if (p->d_rels) (void)hipFree(p->d_rels);
    if (p->d_layer_diff) (void)hipFree(p->d_layer_diff);
    if (p->d_partials) (void)hipFree(p->d_partials);
    if (p->d_layer_mean) (void)hipFree(p->d_layer_mean);
    if (p->d_state) (void)hipFree(p->d_state);
    for (float* a : p->arenas) (void)hipFree(a);
    delete p;
}

int dfq_le_plan_create(const dfq_layer* layers, int32_t n_layers, const dfq_relation* relations,
                       int32_t n_relations, dfq_le_plan** out_plan) {
    if (!layers || n_layers <= 0 || !out_plan || n_relations < 0 || (n_relations > 0 && !relations))
        return fail_arg("dfq_le_plan_create: bad argument");
    // ---- validate geometry (dfq.py:29-35) ----
    for (int r = 0; r < n_relations; ++r) {
        const dfq_relation& rr = relations[r];
        if (rr.first < 0 || rr.first >= n_layers || rr.second < 0 || rr.second >= n_layers || rr.first == rr.second)
            return fail_arg("dfq_le_plan_create: relation %d has bad layer indices (%d, %d)", r, rr.first, rr.second);
        const dfq_layer& A = layers[rr.first];
        const dfq_layer& B = layers[rr.second];
        if (!A.weight || !B.weight || !rr.scale_cum) return fail_arg("dfq_le_plan_create: relation %d has a null pointer", r);
        if (A.out_ch <= 0 || A.in_per_group <= 0 || A.khkw <= 0 || B.out_ch <= 0 || B.in_per_group <= 0 || B.khkw <= 0)
            return fail_arg("dfq_le_plan_create: relation %d has an empty layer", r);
        const int o1 = A.out_ch, i2g = B.in_per_group;
        const int G = (o1 != i2g) ? (o1 / i2g) : 1;
        if (G < 1 || o1 != G * i2g || B.out_ch % G != 0)
            return fail_arg("dfq_le_plan_create: relation %d: unsupported pairing O1=%d, I2/g=%d, O2=%d", r, o1, i2g, B.out_ch);
        if ((int64_t)A.in_per_group * A.khkw > 0x7fffffff / kTcMax)
            return fail_arg("dfq_le_plan_create: relation %d: row too long", r);
    }
    dfq_le_plan* p = new dfq_le_plan();
    p->n_layers = n_layers;
    p->n_rels = n_relations;

    // ---- touches per layer, in sweep order -> diff modes ----
    std::vector<int> touches(n_layers, 0), seen(n_layers, 0);
    for (int r = 0; r < n_relations; ++r) { touches[relations[r].first]++; touches[relations[r].second]++; }
    std::vector<float*> arena(n_layers, nullptr);
    auto fail_alloc = [&](hipError_t e) { dfq_le_plan_destroy(p); return fail_hip(e, "le plan allocation", __FILE__, __LINE__); };
    for (int l = 0; l < n_layers; ++l) {
        if (touches[l] >= 2) {
            const int64_t n = (int64_t)layers[l].out_ch * layers[l].in_per_group * layers[l].khkw;
            float* a = nullptr;
            hipError_t e = hipMalloc((void**)&a, sizeof(float) * n);
            if (e != hipSuccess) return fail_alloc(e);
            p->arenas.push_back(a);
            arena[l] = a;
        }
    }
    auto mode_for = [&](int l) -> int32_t {
        const int k = ++seen[l];
        if (touches[l] == 1) return DIFF_DIRECT;
        if (k == 1) return DIFF_SAVE;
        if (k == touches[l]) return DIFF_FROM_PREV;
        return DIFF_NONE;
    };

    // ---- dependency levels: relations sharing a layer keep their list order ----
    std::vector<int> level(n_relations, 0), last_level(n_layers, -1);
    int n_levels = 0;
    for (int r = 0; r < n_relations; ++r) {
        const int lv = std::max(last_level[relations[r].first], last_level[relations[r].second]) + 1;
        level[r] = lv;
        last_level[relations[r].first] = lv;
        last_level[relations[r].second] = lv;
        n_levels = std::max(n_levels, lv + 1);
    }

    // ---- per-relation device descriptors ----
    std::vector<LeRelDev> h(n_relations);
    std::vector<LeLayerDiff> ld(n_layers);
    for (int l = 0; l < n_layers; ++l) {
        ld[l].partial_begin = -1; ld[l].n_partials = 0; ld[l].side = 0; ld[l].pad = 0;
        ld[l].n_elems = (double)((int64_t)layers[l].out_ch * layers[l].in_per_group * layers[l].khkw);
    }
    int tile_slot = 0;
    for (int r = 0; r < n_relations; ++r) {
        const dfq_relation& rr = relations[r];
        const dfq_layer& A = layers[rr.first];
        const dfq_layer& B = layers[rr.second];
        LeRelDev& d = h[r];
        d.w1 = A.weight; d.w2 = B.weight; d.b1 = A.bias; d.bnw = rr.bn_weight; d.bnb = rr.bn_bias; d.s_cum = rr.scale_cum;
        d.o1 = A.out_ch; d.row_len = A.in_per_group * A.khkw;
        d.i2g = B.in_per_group; d.khkw = B.khkw;
        const int G = (d.o1 != d.i2g) ? (d.o1 / d.i2g) : 1;
        d.gi = d.o1 / G; d.go = B.out_ch / G;
        const int64_t per_channel = (int64_t)d.row_len + (int64_t)d.go * d.khkw;
        d.tc = pick_tc(per_channel, d.o1, d.khkw);
        d.n_tiles = (d.o1 + d.tc - 1) / d.tc;
        d.diff1 = mode_for(rr.first);
        d.diff2 = mode_for(rr.second);
        d.prev1 = (d.diff1 == DIFF_SAVE || d.diff1 == DIFF_FROM_PREV) ? arena[rr.first] : nullptr;
        d.prev2 = (d.diff2 == DIFF_SAVE || d.diff2 == DIFF_FROM_PREV) ? arena[rr.second] : nullptr;
        d.partial_base = tile_slot;
        d.tile_begin = 0;
        tile_slot += d.n_tiles;
        if (d.diff1 == DIFF_DIRECT || d.diff1 == DIFF_FROM_PREV) {
            ld[rr.first].partial_begin = d.partial_base; ld[rr.first].n_partials = d.n_tiles; ld[rr.first].side = 0;
        }
        if (d.diff2 == DIFF_DIRECT || d.diff2 == DIFF_FROM_PREV) {
            ld[rr.second].partial_begin = d.partial_base; ld[rr.second].n_partials = d.n_tiles; ld[rr.second].side = 1;
        }
        const int64_t n1 = (int64_t)d.o1 * d.row_len;
        const int64_t n2 = (int64_t)B.out_ch * d.i2g * d.khkw;
        p->paired_total += n1 + n2;
    }
    p->total_tiles = tile_slot;

    // ---- sort relations by level (stable) and lay out the launches ----
    std::vector<int> order(n_relations);
    for (int r = 0; r < n_relations; ++r) order[r] = r;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return level[a] < level[b]; });
    std::vector<LeRelDev> sorted(n_relations);
    p->levels.assign(n_levels, LevelLaunch());
    for (int i = 0; i < n_relations; ++i) {
        const int r = order[i];
        LevelLaunch& L = p->levels[level[r]];
        if (L.n_rels == 0) L.rel_begin = i;
        sorted[i] = h[r];
        sorted[i].tile_begin = L.n_blocks;
        L.n_blocks += h[r].n_tiles;
        L.n_rels += 1;
        const dfq_layer& A = layers[relations[r].first];
        const dfq_layer& B = layers[relations[r].second];
        const int64_t n1 = (int64_t)A.out_ch * A.in_per_group * A.khkw;
        const int64_t n2 = (int64_t)B.out_ch * B.in_per_group * B.khkw;
        L.paired += n1 + n2;
        if (h[r].prev1) L.snapshot += n1;
        if (h[r].prev2) L.snapshot += n2;
    }
    for (const LevelLaunch& L : p->levels) p->snapshot_total += L.snapshot;

    hipError_t e;
    const size_t n_part = (size_t)std::max(1, p->total_tiles) * 2;
    if ((e = hipMalloc((void**)&p->d_rels, sizeof(LeRelDev) * std::max(1, n_relations))) != hipSuccess) return fail_alloc(e);
    if ((e = hipMalloc((void**)&p->d_layer_diff, sizeof(LeLayerDiff) * n_layers)) != hipSuccess) return fail_alloc(e);
    if ((e = hipMalloc((void**)&p->d_partials, sizeof(double) * n_part)) != hipSuccess) return fail_alloc(e);
    if ((e = hipMalloc((void**)&p->d_layer_mean, sizeof(double) * n_layers)) != hipSuccess) return fail_alloc(e);
    if ((e = hipMalloc((void**)&p->d_state, sizeof(LeState))) != hipSuccess) return fail_alloc(e);
    if (n_relations > 0 &&
        (e = hipMemcpy(p->d_rels, sorted.data(), sizeof(LeRelDev) * n_relations, hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
    if ((e = hipMemcpy(p->d_layer_diff, ld.data(), sizeof(LeLayerDiff) * n_layers, hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
    if ((e = hipMemset(p->d_partials, 0, sizeof(double) * n_part)) != hipSuccess) return fail_alloc(e);
    if ((e = hipMemset(p->d_state, 0, sizeof(LeState))) != hipSuccess) return fail_alloc(e);
    if ((e = hipDeviceSynchronize()) != hipSuccess) return fail_alloc(e);
    *out_plan = p;
    return DFQ_OK;
}

int32_t dfq_le_plan_levels(const dfq_le_plan* p) { return p ? (int32_t)p->levels.size() : 0; }
int64_t dfq_le_plan_paired_elements(const dfq_le_plan* p) { return p ? p->paired_total : 0; }
int64_t dfq_le_plan_snapshot_elements(const dfq_le_plan* p) { return p ? p->snapshot_total : 0; }

int32_t dfq_le_plan_level_launches(const dfq_le_plan* p, int32_t level, int64_t* paired_elems,
                                   int64_t* snapshot_elems, int32_t* n_workgroups) {
    if (!p || level < 0 || level >= (int32_t)p->levels.size()) return fail_arg("dfq_le_plan_level_launches: bad level");
    const LevelLaunch& L = p->levels[level];
    if (paired_elems) *paired_elems = L.paired;
    if (snapshot_elems) *snapshot_elems = L.snapshot;
    if (n_workgroups) *n_workgroups = L.n_blocks;
    return L.n_rels;
}

static LeParams make_params(const dfq_le_config* c) {
    LeParams q;
    q.s_lo = c->s_lo; q.s_hi = c->s_hi; q.inv_lo = c->inv_lo; q.inv_hi = c->inv_hi; q.eps = c->eps;
    q.hi_gt_lo = c->hi_gt_lo; q.signed_range = c->signed_range;
    return q;
}

int dfq_le_enqueue(dfq_le_plan* p, const dfq_le_config* cfg, int32_t n_sweeps, int32_t restart, void* stream) {
    if (!p || !cfg || n_sweeps < 0) return fail_arg("dfq_le_enqueue: bad argument");
    hipStream_t st = as_stream(stream);
    const LeParams q = make_params(cfg);
    if (restart) {
        hipLaunchKernelGGL(le_reset_kernel, dim3(1), dim3(64), 0, st, p->d_state, cfg->converge_thres,
                           (int)cfg->converge_count, (int)cfg->max_sweeps);
        DFQ_CHECK_LAUNCH();
    }
    for (int s = 0; s < n_sweeps; ++s) {
        for (const LevelLaunch& L : p->levels) {
            if (L.n_blocks == 0) continue;
            hipLaunchKernelGGL(le_level_kernel, dim3(L.n_blocks), dim3(kBlock), 0, st,
                               (const LeRelDev*)(p->d_rels + L.rel_begin), L.n_rels, q,
                               (const LeState*)p->d_state, p->d_partials);
            DFQ_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL(le_control_kernel, dim3(1), dim3(kBlock), 0, st, (const LeLayerDiff*)p->d_layer_diff,
                           p->n_layers, (const double*)p->d_partials, p->d_layer_mean, p->d_state,
                           cfg->converge_thres, (int)cfg->converge_count, (int)cfg->max_sweeps);
        DFQ_CHECK_LAUNCH();
    }
    return DFQ_OK;
}

int dfq_le_profile(dfq_le_plan* p, const dfq_le_config* cfg, int32_t n_sweeps, void* stream, double* level_ms,
                   double* control_ms, int32_t* n_level_launches) {
    if (!p || !cfg || n_sweeps <= 0 || !level_ms) return fail_arg("dfq_le_profile: bad argument");
    hipStream_t st = as_stream(stream);
    const LeParams q = make_params(cfg);
    const int n_levels = (int)p->levels.size();
    const int per_sweep = n_levels + 1;
    std::vector<hipEvent_t> ev((size_t)2 * per_sweep * n_sweeps);
    for (auto& e : ev) DFQ_HIP_TRY(hipEventCreate(&e));
    hipLaunchKernelGGL(le_reset_kernel, dim3(1), dim3(64), 0, st, p->d_state, cfg->converge_thres,
                       (int)cfg->converge_count, (int)cfg->max_sweeps);
    DFQ_CHECK_LAUNCH();
    size_t k = 0;
    for (int s = 0; s < n_sweeps; ++s) {
        for (const LevelLaunch& L : p->levels) {
            DFQ_HIP_TRY(hipEventRecord(ev[k++], st));
            if (L.n_blocks > 0)
                hipLaunchKernelGGL(le_level_kernel, dim3(L.n_blocks), dim3(kBlock), 0, st,
                                   (const LeRelDev*)(p->d_rels + L.rel_begin), L.n_rels, q,
                                   (const LeState*)p->d_state, p->d_partials);
            DFQ_HIP_TRY(hipEventRecord(ev[k++], st));
        }
        DFQ_HIP_TRY(hipEventRecord(ev[k++], st));
        hipLaunchKernelGGL(le_control_kernel, dim3(1), dim3(kBlock), 0, st, (const LeLayerDiff*)p->d_layer_diff,
                           p->n_layers, (const double*)p->d_partials, p->d_layer_mean, p->d_state,
                           cfg->converge_thres, (int)cfg->converge_count, (int)cfg->max_sweeps);
        DFQ_HIP_TRY(hipEventRecord(ev[k++], st));
    }
    DFQ_CHECK_LAUNCH();
    DFQ_HIP_TRY(hipStreamSynchronize(st));
    for (int l = 0; l < n_levels; ++l) level_ms[l] = 0.0;
    double ctl = 0.0;
    k = 0;
    for (int s = 0; s < n_sweeps; ++s) {
        for (int l = 0; l <= n_levels; ++l) {
            float ms = 0.0f;
            DFQ_HIP_TRY(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
            k += 2;
            if (l < n_levels) level_ms[l] += ms; else ctl += ms;
        }
    }
    if (control_ms) *control_ms = ctl;
    if (n_level_launches) *n_level_launches = n_levels * n_sweeps;
    for (auto& e : ev) (void)hipEventDestroy(e);
    return DFQ_OK;
}

int dfq_le_query(dfq_le_plan* p, void* stream, dfq_le_result* out, int32_t* done) {
    if (!p) return fail_arg("dfq_le_query: null plan");
    LeState h;
    hipStream_t st = as_stream(stream);
    DFQ_HIP_TRY(hipMemcpyAsync(&h, p->d_state, sizeof(LeState), hipMemcpyDeviceToHost, st));
    DFQ_HIP_TRY(hipStreamSynchronize(st));
    if (out) {
        out->sweeps = h.sweeps;
        out->stall_count = h.count;
        out->diff = h.diff;
        out->last_diff_tmp = h.last_diff_tmp;
    }
    if (done) *done = h.done;
    return DFQ_OK;
}

int dfq_le_run(dfq_le_plan* p, const dfq_le_config* cfg, void* stream, dfq_le_result* out) {
    if (!p || !cfg) return fail_arg("dfq_le_run: bad argument");
    int rc = dfq_le_enqueue(p, cfg, 0, 1, stream);
    if (rc) return rc;
    int32_t done = 0;
    dfq_le_result res;
    int chunk = 8;
    if (cfg->max_sweeps >= 0) {
        // the sweep count is known: enqueue all of it, one synchronisation at the end
        rc = dfq_le_enqueue(p, cfg, cfg->max_sweeps, 0, stream);
        if (rc) return rc;
        rc = dfq_le_query(p, stream, &res, &done);
        if (rc) return rc;
    } else {
        for (;;) {
            rc = dfq_le_query(p, stream, &res, &done);
            if (rc) return rc;
            if (done) break;
            rc = dfq_le_enqueue(p, cfg, chunk, 0, stream);
            if (rc) return rc;
            if (chunk < 32) chunk *= 2;
        }
    }
    if (out) *out = res;
    return DFQ_OK;
}

}  // extern "C"
