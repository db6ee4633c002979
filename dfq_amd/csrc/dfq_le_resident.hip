// Register-resident cross-layer equalisation: the WHOLE data-dependent loop of dfq.py:78-117 for one network as
// ONE persistent launch (gfx950).
//
// A single network is tiny for this chip: MobileNetV2's paired layers are 13.9 MB, the register files of 256 CUs
// hold 128 MB.  The streaming kernel of dfq_le.hip re-reads and re-writes every weight every sweep and pays, per
// sweep, one launch boundary plus a chain of dependent tile latencies (descriptor -> data -> statistics -> store:
// ~6 us per dependency level, 47 sweeps x 5 levels).  Here every workgroup loads ONE [rows x columns] tile of ONE
// paired layer into registers once, keeps it there for all sweeps, and only the per-channel statistics travel
// between workgroups; the weights are written back once, after the loop has stopped.  A sweep then costs the
// latency of its dependency chain of statistics hand-offs (a few microseconds) and nothing else.
//
// What a tile of layer L does in sweep k (A = the relation whose SECOND layer is L, B = the relation whose FIRST
// layer is L; either may be absent; dfq.py:85-101 processes A before B):
//   phase 1 (A): wait for A's row statistics of this sweep and L's column statistics of the previous sweep;
//                solve s_A per input channel (dfq.py:58-59); the row statistics of t = fl(w / s_A) (dfq.py:73) are
//                merged over the tiles of the row block and published for B -- no store, t is recomputed below;
//   phase 2 (B): wait for L's merged row statistics and for the column statistics of B's second layer (previous
//                sweep); solve s_B per row; the tile that owns the rows updates b, gamma~, beta~, S (dfq.py:62-71);
//   phase 3    : w <- fl(fl(w / s_A) * s_B) in registers (the reference's two roundings in its order), |dW| summed
//                in float64, column statistics of the new values published for A of the NEXT sweep (row
//                statistics for B of the next sweep when L is a chain start).
// Statistics words are 64-bit {sweep tag : order-preserving float bits}, merged with device-scope atomicMax: a newer
// sweep always wins, so nothing is ever cleared; two parities (tag & 1) keep a sweep's readers and the next sweep's
// writers apart.  "All tiles of layer X have published" is one monotonic counter per layer and statistic kind.
// Convergence (dfq.py:105-115): every tile leaves one float64 partial, the LAST tile to arrive (ticket from an
// atomicAdd) sums them per layer in a fixed order, advances the reference's (diff, count) state machine and
// publishes the decision; everybody waits for it before the next sweep.
//
// All workgroups of the launch must be resident at once (they wait for each other in cycles over the sweeps): the
// plan refuses networks that do not fit (the caller then uses the streaming kernel), the library never runs two
// kernels with in-launch waits concurrently (SpinGuard), every wait is bounded, and a workgroup that abandons a wait
// stores NOTHING: a failed launch leaves the weights exactly as they were and is reported by the next query.
// Results are bit-identical to the streaming kernel and to the oracle (same IEEE operations; min/max are exact).
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "dfq_common.hpp"
#include "dfq_le_shared.hpp"
#include "dfq_le_resident.hpp"

namespace dfq {

constexpr int kResTab = 2048;        // LDS table entries of a tile: (groups x input channels) it spans
constexpr int kResRows = 1024;       // rows of a tile that needs per-row tables (4 owner rows per thread at most)
constexpr int kResOwn = kResRows / kBlock;
constexpr int kResStride = 16;       // one 64-bit counter per 128-byte line
constexpr long kResSpinLimit = 8000000;
constexpr int kResMaxTiles = 3072;   // partials staged in LDS by the deciding workgroup
constexpr int kResMaxLayers = 1024;

typedef unsigned long long u64;

struct ResTile {                     // one workgroup
    float* w;                        // weight base of the layer
    int32_t n_rows, row_len, khkw;   // layer geometry: rows, floats per row, taps per input channel
    int32_t go, i2g;                 // as a SECOND layer: rows per group, input channels per group (paired channel = g * i2g + ii)
    int32_t r0, nr, c0, nc;          // tile: rows [r0, r0 + nr), row positions [c0, c0 + nc)
    int32_t vec;                     // 4: float4 slots (row_len, c0, nc multiples of 4, base 16-byte aligned), 1: scalar slots
    int32_t relA, relB;              // relation whose second / first layer this is, or -1
    int32_t layer;                   // paired-layer index: counters cnt_r / cnt_c
    int32_t a_layer, b_layer;        // paired-layer index of A's first layer / B's second layer
    int32_t nt_self, nt_a, nt_b;     // tiles of this layer / of those two
    int32_t owner;                   // holds column block 0: updates the [O] vectors of relation B
    int32_t pad;
};

struct ResRel {
    float* b1;
    float* bnw;
    float* bnb;
    float* s_cum;
    int64_t r1_off, r2_off;          // first u64 word of this relation's row / column statistics (parity 0)
    int32_t channels;
    int32_t pad;
};

struct ResLayerDiff {                // one targ layer of the network, graph order
    int32_t tile_begin, n_tiles;     // its tiles (contiguous), n_tiles == 0: untouched -> contributes exactly 0
    double n_elems;
};

struct ResArgs {
    const ResTile* tiles;
    const ResRel* rels;
    const ResLayerDiff* layer_diff;
    u64* stats;                      // r1 arena then r2 arena; each [2 parities][channels][2 words]
    int64_t parity_stride;           // u64 words between the parities of an arena
    u64* cnt_r;                      // per paired layer (x kResStride): tiles that published row statistics
    u64* cnt_c;                      //   "   column statistics
    u64* done_cnt;                   // tiles that finished a sweep
    u64* seq;                        // (sweeps finished in this launch << 1) | stop
    u64* err;
    double* partials;                // one per tile
    LeState* state;
    int32_t n_tiles, n_layers;       // n_layers: targ layers of the network (layer_diff entries)
    int32_t n_sweeps;                // sweeps this launch may run
    int32_t max_sweeps;              // cfg: total cap (< 0: none)
    int32_t converge_count;
    int32_t pad;
    double converge_thres;
};

// ---- waits ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool res_wait(const u64* word, u64 target, u64 shift, u64* err, int* sh_flag) {
    if (threadIdx.x == 0) {
        long spins = 0;
        int ok = 1;
        while ((__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> shift) < target) {
            __builtin_amdgcn_s_sleep(2);
            ++spins;
            if (spins > kResSpinLimit ||
                ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull)) {
                atomicMax(err, 1ull);
                ok = 0;
                break;
            }
        }
        *sh_flag = ok;
    }
    __syncthreads();
    const bool ok = *sh_flag != 0;
    __syncthreads();                 // sh_flag may be rewritten by the next wait
    return ok;
}

__device__ __forceinline__ u64 ld_word(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish_max(u64* p, uint32_t tag, uint32_t slot) {
    atomicMax(p, ((u64)tag << 32) | (u64)slot);
}
// (min, max) of channel c for consumption tag `tag`; a word that still carries an older tag reads as "empty range"
__device__ __forceinline__ void read_range(const u64* arena, int64_t off, int64_t parity_stride, uint32_t tag, int c,
                                           float& mn, float& mx) {
    const u64* p = arena + off + (int64_t)(tag & 1u) * parity_stride + 2 * (int64_t)c;
    const u64 a = ld_word(p), b = ld_word(p + 1);
    mn = ((uint32_t)(a >> 32) == tag) ? slot_min((uint32_t)a) : INFINITY;
    mx = ((uint32_t)(b >> 32) == tag) ? slot_max((uint32_t)b) : -INFINITY;
}

// everything performed -> one arrival on the counter
__device__ __forceinline__ void arrive(u64* counter) {
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(counter, 1ull);
}

__device__ __forceinline__ void opaque(int& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
}

// ---- the tile in registers ------------------------------------------------------------------------------
// Slot u of thread t holds VEC consecutive floats of tile row `row`, positions `pos .. pos + VEC - 1`:
//   q = u * 256 + t,  row = q / (nc / VEC),  pos = c0 + (q % (nc / VEC)) * VEC.
template <int VEC>
struct Geo {
    int tcv;                         // vectors per tile row
    int n_vec;                       // vectors of the tile
    int g_lo, nci, i0;               // LDS table of phase 1 / column statistics: (group - g_lo) * nci + (ii - i0)
};

template <int VEC>
__device__ __forceinline__ void slot_coords(const ResTile& T, const Geo<VEC>& G, int u, int& row, int& pos, bool& on) {
    int q = u * kBlock + (int)threadIdx.x;
    // The coordinates of a slot never change, so the compiler would hoist them out of the sweep loop and keep three
    // integers per slot alive next to the data (it did: 400+ registers, one wave per SIMD).  They cost a handful of
    // instructions to recompute; the empty asm makes `q` opaque so that they are recomputed where they are used.
    opaque(q);
    on = q < G.n_vec;
    const int qq = on ? q : 0;
    row = small_div(qq, G.tcv);
    pos = T.c0 + (qq - row * G.tcv) * VEC;
}

// table index of (tile row, position + k)
template <int VEC>
__device__ __forceinline__ int tab_index(const ResTile& T, const Geo<VEC>& G, int row, int pos_k) {
    const int g = small_div(T.r0 + row, T.go) - G.g_lo;
    const int ii = small_div(pos_k, T.khkw) - G.i0;
    return g * G.nci + ii;
}

// Row statistics (per tile row) of the values `get(u, k, row, pos)` into sh_row[2 * row + {0: min slot, 1: max slot}]
// (identity 0).  A wave whose 64 lanes sit in one row reduces with a butterfly and issues one LDS atomic; otherwise
// every lane issues its own.
template <int VEC, int NS, typename Get>
__device__ __forceinline__ void tile_row_stats(const ResTile& T, const Geo<VEC>& G, uint32_t* sh_row, Get get) {
    const int lane = threadIdx.x % kWave;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        if (u * kBlock >= G.n_vec) continue;                       // uniform
        int row, pos; bool on;
        slot_coords<VEC>(T, G, u, row, pos, on);
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float x = get(u, k, row, pos);
            mn = vmin_raw(mn, on ? x : INFINITY);
            mx = vmax_raw(mx, on ? x : -INFINITY);
        }
        const int r_first = __shfl(row, 0), r_last = __shfl(row, kWave - 1);
        const int on_all = __shfl((int)on, kWave - 1);             // lanes are ordered: the last one decides
        if (on_all && r_first == r_last) {
            mn = wave_min(mn); mx = wave_max(mx);
            if (lane == 0) { atomicMax(&sh_row[2 * row], ~enc_ord(mn)); atomicMax(&sh_row[2 * row + 1], enc_ord(mx)); }
        } else if (on) {
            atomicMax(&sh_row[2 * row], ~enc_ord(mn));
            atomicMax(&sh_row[2 * row + 1], enc_ord(mx));
        }
    }
}

// Column statistics of x into sh_col[2 * table index + {0, 1}] (identity 0)
template <int VEC, int NS>
__device__ __forceinline__ void tile_col_stats(const ResTile& T, const Geo<VEC>& G, const float (&x)[NS][VEC], uint32_t* sh_col) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        if (u * kBlock >= G.n_vec) continue;
        int row, pos; bool on;
        slot_coords<VEC>(T, G, u, row, pos, on);
        if (!on) continue;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int idx = tab_index<VEC>(T, G, row, pos + k);
            atomicMax(&sh_col[2 * idx], ~enc_ord(x[u][k]));
            atomicMax(&sh_col[2 * idx + 1], enc_ord(x[u][k]));
        }
    }
}

// sh_row -> global row statistics of relation B (rows r0 .. r0 + nr), tagged
__device__ __forceinline__ void publish_rows(const ResArgs& a, const ResTile& T, const ResRel& RB, const uint32_t* sh_row, uint32_t tag) {
    u64* dst = a.stats + RB.r1_off + (int64_t)(tag & 1u) * a.parity_stride;
    for (int i = threadIdx.x; i < T.nr; i += kBlock) {
        publish_max(dst + 2 * (int64_t)(T.r0 + i), tag, sh_row[2 * i]);
        publish_max(dst + 2 * (int64_t)(T.r0 + i) + 1, tag, sh_row[2 * i + 1]);
    }
}
// sh_col -> global column statistics of relation A, tagged
template <int VEC>
__device__ __forceinline__ void publish_cols(const ResArgs& a, const ResTile& T, const Geo<VEC>& G, int g_n, const ResRel& RA,
                                             const uint32_t* sh_col, uint32_t tag) {
    u64* dst = a.stats + RA.r2_off + (int64_t)(tag & 1u) * a.parity_stride;
    for (int idx = threadIdx.x; idx < g_n * G.nci; idx += kBlock) {
        const int gq = small_div(idx, G.nci);
        const int c = (G.g_lo + gq) * T.i2g + G.i0 + (idx - gq * G.nci);
        if (sh_col[2 * idx + 1] != 0u) {              // a channel no element of this tile belongs to stays untouched
            publish_max(dst + 2 * (int64_t)c, tag, sh_col[2 * idx]);
            publish_max(dst + 2 * (int64_t)c + 1, tag, sh_col[2 * idx + 1]);
        }
    }
}

// dfq.py:105-115 by the last tile to arrive.  sh_d: kResMaxTiles doubles (the tile's own LDS tables are dead here)
__device__ __forceinline__ void decide(const ResArgs& a, int k, double* sh_d, double* sh_mean) {
    const int tid = threadIdx.x;
    for (int i = tid; i < a.n_tiles; i += kBlock)
        sh_d[i] = __longlong_as_double((long long)ld_word((const u64*)a.partials + i));
    __syncthreads();
    for (int l = tid; l < a.n_layers; l += kBlock) {
        const ResLayerDiff L = a.layer_diff[l];
        double s = 0.0;
        for (int i = 0; i < L.n_tiles; ++i) s += sh_d[L.tile_begin + i];          // fixed order
        // float(torch.mean(torch.abs(W - W_prev))): float32 mean, widened to double (dfq.py:108)
        sh_mean[l] = (L.n_tiles > 0) ? (double)(float)(s / L.n_elems) : 0.0;
    }
    __syncthreads();
    if (tid == 0) {
        double diff_tmp = 0.0;
        for (int l = 0; l < a.n_layers; ++l) diff_tmp += sh_mean[l];               // graph order, like Python's sum
        // the deciding workgroup changes from sweep to sweep (different XCDs, L2s not coherent): device-scope accesses
        LeState* st = a.state;
        double diff = __longlong_as_double((long long)ld_word((const u64*)&st->diff));
        int count = (int)__hip_atomic_load((const uint32_t*)&st->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (fabs(diff - diff_tmp) > 1e-9) { count = 0; diff = diff_tmp; }
        else { count += 1; }
        const int sweeps = (int)__hip_atomic_load((const uint32_t*)&st->sweeps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        const bool go_on = (diff > a.converge_thres) && (count < a.converge_count) && (a.max_sweeps < 0 || sweeps < a.max_sweeps);
        __hip_atomic_store((u64*)&st->diff, (u64)__double_as_longlong(diff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((u64*)&st->last_diff_tmp, (u64)__double_as_longlong(diff_tmp), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((uint32_t*)&st->count, (uint32_t)count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((uint32_t*)&st->sweeps, (uint32_t)sweeps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((uint32_t*)&st->done, go_on ? 0u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);
        __hip_atomic_store(a.seq, ((u64)(k + 1) << 1) | (go_on ? 0ull : 1ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int VEC, int NS>
__device__ __forceinline__ void res_tile_body(const ResArgs& a, const LeParams& p, const ResTile& T, unsigned char* smem) {
    // LDS: [inv: kResTab f32][s: kResRows f32][row stats: 2 * kResRows u32][col stats: 2 * kResTab u32][flag]
    float* sh_inv = (float*)smem;
    float* sh_s = sh_inv + kResTab;
    uint32_t* sh_row = (uint32_t*)(sh_s + kResRows);
    uint32_t* sh_col = sh_row + 2 * kResRows;
    int* sh_flag = (int*)(sh_col + 2 * kResTab);
    const int tid = threadIdx.x;
    const bool hasA = T.relA >= 0, hasB = T.relB >= 0;
    const ResRel RA = a.rels[hasA ? T.relA : 0];
    const ResRel RB = a.rels[hasB ? T.relB : 0];
    Geo<VEC> G;
    G.tcv = T.nc / VEC;
    G.n_vec = T.nr * G.tcv;
    G.i0 = small_div(T.c0, T.khkw);
    G.nci = small_div(T.c0 + T.nc - 1, T.khkw) - G.i0 + 1;
    G.g_lo = small_div(T.r0, T.go);
    const int g_n = small_div(T.r0 + T.nr - 1, T.go) - G.g_lo + 1;
    gfloat* const wt = (gfloat*)T.w;

    // ---- load the tile (once) ----
    float v[NS][VEC];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        int row, pos; bool on;
        slot_coords<VEC>(T, G, u, row, pos, on);
        const gfloat* src = wt + ((int64_t)(T.r0 + row) * T.row_len + pos);
        if (u * kBlock < G.n_vec) {
            if (VEC == 4) {
                const fvec4 t4 = *(const gfvec4*)src;
                v[u][0] = t4[0]; v[u][1 % VEC] = t4[1]; v[u][2 % VEC] = t4[2]; v[u][3 % VEC] = t4[3];
            } else {
                v[u][0] = *src;
            }
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[u][k] = 0.0f;
        }
    }
    // the [O] vectors of relation B for the rows this thread owns
    float o_cum[kResOwn], o_bnw[kResOwn], o_bnb[kResOwn], o_b1[kResOwn];
    const bool owner = hasB && T.owner != 0;
#pragma unroll
    for (int j = 0; j < kResOwn; ++j) {
        const int i = tid + j * kBlock;
        o_cum[j] = 1.0f; o_bnw[j] = 0.0f; o_bnb[j] = 0.0f; o_b1[j] = 0.0f;
        if (owner && i < T.nr) {
            const int c = T.r0 + i;
            o_cum[j] = RB.s_cum[c];
            if (RB.bnw) o_bnw[j] = RB.bnw[c];
            if (RB.bnb) o_bnb[j] = RB.bnb[c];
            if (RB.b1) o_b1[j] = RB.b1[c];
        }
    }

    // ---- statistics of the untouched weights: consumption tag 1 (sweep 0) ----
    const bool chain_start = hasB && !hasA;
    if (hasA) {
        for (int i = tid; i < 2 * g_n * G.nci; i += kBlock) sh_col[i] = 0u;
        __syncthreads();
        tile_col_stats<VEC, NS>(T, G, v, sh_col);
        __syncthreads();
        publish_cols<VEC>(a, T, G, g_n, RA, sh_col, 1u);
        arrive(a.cnt_c + (int64_t)T.layer * kResStride);
    }
    if (chain_start) {
        for (int i = tid; i < 2 * T.nr; i += kBlock) sh_row[i] = 0u;
        __syncthreads();
        tile_row_stats<VEC, NS>(T, G, sh_row, [&](int u, int kk, int, int) { return v[u][kk]; });
        __syncthreads();
        publish_rows(a, T, RB, sh_row, 1u);
        arrive(a.cnt_r + (int64_t)T.layer * kResStride);
    }

    int k = 0;
    bool failed = false;
    for (;; ++k) {
        const uint32_t tag = (uint32_t)k + 1u;                  // what this sweep consumes
        const u64 round = (u64)k + 1ull;
        // ---- phase 1: s_A per (group, input channel) of the tile ----
        if (hasA) {
            if (!res_wait(a.cnt_r + (int64_t)T.a_layer * kResStride, (u64)T.nt_a * round, 0, a.err, sh_flag) ||
                !res_wait(a.cnt_c + (int64_t)T.layer * kResStride, (u64)T.nt_self * round, 0, a.err, sh_flag)) { failed = true; break; }
            for (int idx = tid; idx < g_n * G.nci; idx += kBlock) {
                const int gq = small_div(idx, G.nci);
                const int c = (G.g_lo + gq) * T.i2g + G.i0 + (idx - gq * G.nci);
                float mn1, mx1, mn2, mx2, s, inv;
                read_range(a.stats, RA.r1_off, a.parity_stride, tag, c, mn1, mx1);
                read_range(a.stats, RA.r2_off, a.parity_stride, tag, c, mn2, mx2);
                le_solve(range_of(mn1, mx1, p.signed_range), range_of(mn2, mx2, p.signed_range), p, s, inv);
                sh_inv[idx] = inv;
            }
            if (hasB) for (int i = tid; i < 2 * T.nr; i += kBlock) sh_row[i] = 0u;
            __syncthreads();
            if (hasB) {
                // row statistics of t = fl(w * 1/s_A) for relation B of this same sweep (t is not kept: phase 3 recomputes it)
                tile_row_stats<VEC, NS>(T, G, sh_row, [&](int u, int kk, int row, int pos) {
                    return v[u][kk] * sh_inv[tab_index<VEC>(T, G, row, pos + kk)];
                });
                __syncthreads();
                publish_rows(a, T, RB, sh_row, tag);
                arrive(a.cnt_r + (int64_t)T.layer * kResStride);
            }
        }
        // ---- phase 2: s_B per row ----
        if (hasB) {
            if (!res_wait(a.cnt_r + (int64_t)T.layer * kResStride, (u64)T.nt_self * round, 0, a.err, sh_flag) ||
                !res_wait(a.cnt_c + (int64_t)T.b_layer * kResStride, (u64)T.nt_b * round, 0, a.err, sh_flag)) { failed = true; break; }
#pragma unroll
            for (int j = 0; j < kResOwn; ++j) {
                const int i = tid + j * kBlock;
                if (i < T.nr) {
                    const int c = T.r0 + i;
                    float mn1, mx1, mn2, mx2, s, inv;
                    read_range(a.stats, RB.r1_off, a.parity_stride, tag, c, mn1, mx1);
                    read_range(a.stats, RB.r2_off, a.parity_stride, tag, c, mn2, mx2);
                    le_solve(range_of(mn1, mx1, p.signed_range), range_of(mn2, mx2, p.signed_range), p, s, inv);
                    sh_s[i] = s;
                    o_cum[j] = o_cum[j] * s;                      // relation.py:20-24
                    o_bnw[j] = o_bnw[j] * s;                      // dfq.py:64-65
                    o_bnb[j] = o_bnb[j] * s;                      // dfq.py:67-68
                    o_b1[j] = o_b1[j] * s;                        // dfq.py:70-71
                }
            }
        }
        // ---- phase 3: the new values, |dW|, statistics for the next sweep ----
        if (hasA) for (int i = tid; i < 2 * g_n * G.nci; i += kBlock) sh_col[i] = 0u;
        if (chain_start) for (int i = tid; i < 2 * T.nr; i += kBlock) sh_row[i] = 0u;
        __syncthreads();
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            if (u * kBlock >= G.n_vec) continue;
            int row, pos; bool on;
            slot_coords<VEC>(T, G, u, row, pos, on);
            const float s = hasB ? sh_s[row] : 1.0f;
            double part = 0.0;
#pragma unroll
            for (int kk = 0; kk < VEC; ++kk) {
                const float inv = hasA ? sh_inv[tab_index<VEC>(T, G, row, pos + kk)] : 1.0f;
                const float tt = v[u][kk] * inv;                  // dfq.py:73 (rounded), then
                const float nv = tt * s;                          // dfq.py:62
                const float d = nv - v[u][kk];
                part += (double)__uint_as_float(__float_as_uint(d) & 0x7fffffffu);
                v[u][kk] = nv;
            }
            acc += on ? part : 0.0;
        }
        if (hasA) tile_col_stats<VEC, NS>(T, G, v, sh_col);
        if (chain_start) tile_row_stats<VEC, NS>(T, G, sh_row, [&](int u, int kk, int, int) { return v[u][kk]; });
        __syncthreads();
        if (hasA) {
            publish_cols<VEC>(a, T, G, g_n, RA, sh_col, tag + 1u);
            arrive(a.cnt_c + (int64_t)T.layer * kResStride);
        }
        if (chain_start) {
            publish_rows(a, T, RB, sh_row, tag + 1u);
            arrive(a.cnt_r + (int64_t)T.layer * kResStride);
        }
        // ---- convergence: one partial per tile (fixed butterfly + fixed wave order), the last arrival decides ----
        {
            double* sh_w = (double*)sh_inv;                        // tables are dead until the next sweep
            __syncthreads();
            const double tsum = block_sum(acc, sh_w);
            if (tid == 0) {
                __hip_atomic_store((u64*)a.partials + blockIdx.x, (u64)__double_as_longlong(tsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_s_waitcnt(0);
                const u64 ticket = atomicAdd(a.done_cnt, 1ull);
                *sh_flag = (ticket == (u64)a.n_tiles * round - 1ull) ? 1 : 0;
            }
            __syncthreads();
            const bool last = *sh_flag != 0;
            __syncthreads();
            if (last) decide(a, k, (double*)smem, (double*)smem + kResMaxTiles);
        }
        if (!res_wait(a.seq, round, 1, a.err, sh_flag)) { failed = true; break; }
        const u64 sq = ld_word(a.seq);
        if ((sq & 1ull) != 0ull || k + 1 >= a.n_sweeps) { ++k; break; }
    }
    if (failed) return;                     // nothing is stored: the weights stay as they were before the launch
    // ---- write the tile back (once) ----
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        int row, pos; bool on;
        slot_coords<VEC>(T, G, u, row, pos, on);
        gfloat* dst = wt + ((int64_t)(T.r0 + row) * T.row_len + pos);
        if (on) {
            if (VEC == 4) {
                fvec4 t4;
                t4[0] = v[u][0]; t4[1] = v[u][1 % VEC]; t4[2] = v[u][2 % VEC]; t4[3] = v[u][3 % VEC];
                *(gfvec4*)dst = t4;
            } else {
                *dst = v[u][0];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kResOwn; ++j) {
        const int i = tid + j * kBlock;
        if (owner && i < T.nr) {
            const int c = T.r0 + i;
            RB.s_cum[c] = o_cum[j];
            if (RB.bnw) RB.bnw[c] = o_bnw[j];
            if (RB.bnb) RB.bnb[c] = o_bnb[j];
            if (RB.b1) RB.b1[c] = o_b1[j];
        }
    }
}

constexpr size_t kResSmemBytes = sizeof(float) * (kResTab + kResRows) + sizeof(uint32_t) * 2 * (kResRows + kResTab) + 64;
static_assert(kResSmemBytes >= sizeof(double) * (kResMaxTiles + kResMaxLayers), "the deciding workgroup stages the partials in the tile's LDS");

// NS4 float4 slots per thread (tiles of 1024 * NS4 floats); scalar tiles always hold 32 floats per thread
template <int NS4>
__global__ __launch_bounds__(kBlock) void le_resident_kernel(ResArgs a, LeParams p) {
    DFQ_DYN_SMEM(smem);
    if (a.state->done) return;              // already stopped (uniform over the launch: written before it started)
    const ResTile T = a.tiles[blockIdx.x];
    if (T.vec == 4) res_tile_body<4, NS4>(a, p, T, smem);
    else res_tile_body<1, 32>(a, p, T, smem);
}

}  // namespace dfq

using namespace dfq;

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct dfq::LeResident {
    int n_tiles = 0, n_pl = 0, n_rels = 0, n_layers = 0, ns4 = 8;
    ResTile* d_tiles = nullptr;
    ResRel* d_rels = nullptr;
    ResLayerDiff* d_layer_diff = nullptr;
    u64* d_stats = nullptr;
    int64_t stat_words = 0;          // u64 words of both arenas, both parities
    int64_t parity_stride = 0;
    u64* d_sync = nullptr;           // cnt_r | cnt_c | done | seq
    size_t sync_words = 0;
    double* d_partials = nullptr;
    int64_t elements = 0;            // floats held in registers
};

namespace {

int ceil_div_i(int a, int b) { return (a + b - 1) / b; }

struct Shape { int tr, tc; };

// [tr x tc] tiling of an [R x C] layer holding at most `cap` floats per tile; cost = global statistics atomics per sweep
Shape pick_shape(int R, int C, int vec, int khkw, int go, int i2g, bool need_row, bool need_col, int cap) {
    Shape best{0, 0};
    double best_cost = 1e300;
    std::vector<int> cands;
    for (int tc = vec; tc < C; tc *= 2) cands.push_back(tc);
    cands.push_back(C);
    for (int tc : cands) {
        if (tc % vec) continue;
        int tr = std::min(R, cap / tc);
        if (tr < 1) continue;
        if (need_row) tr = std::min(tr, kResRows);
        // LDS table of a tile with column duty: (groups spanned by its rows) x (input channels spanned by its columns)
        const int nci = std::min((tc + khkw - 1) / khkw + 1, i2g);
        if (need_col) {
            while (tr > 1 && ((tr + go - 1) / go + 1) * nci > kResTab) tr = (tr + 1) / 2;
            if (((tr + go - 1) / go + 1) * nci > kResTab) continue;
        }
        const int n_rb = ceil_div_i(R, tr), n_cb = ceil_div_i(C, tc);
        double cost = (double)n_rb * n_cb * 0.02;                       // a tile is a workgroup: mild pressure for fewer
        if (need_row) cost += (double)R * n_cb;
        if (need_col) cost += (double)(C / khkw) * n_rb;
        if (cost < best_cost) { best_cost = cost; best = Shape{tr, tc}; }
    }
    return best;
}

}  // namespace

namespace dfq {

void le_resident_destroy(LeResident* r) {
    if (!r) return;
    if (r->d_tiles) (void)hipFree(r->d_tiles);
    if (r->d_rels) (void)hipFree(r->d_rels);
    if (r->d_layer_diff) (void)hipFree(r->d_layer_diff);
    if (r->d_stats) (void)hipFree(r->d_stats);
    if (r->d_sync) (void)hipFree(r->d_sync);
    if (r->d_partials) (void)hipFree(r->d_partials);
    delete r;
}

int le_resident_tiles(const LeResident* r) { return r ? r->n_tiles : 0; }
int64_t le_resident_elements(const LeResident* r) { return r ? r->elements : 0; }

LeResident* le_resident_create(const dfq_layer* layers, int n_layers, const dfq_relation* relations, int n_relations,
                               std::string* why_not) {
    auto refuse = [&](const std::string& m) -> LeResident* { if (why_not) *why_not = m; return nullptr; };
    const char* env = getenv("DFQ_LE_RESIDENT");
    if (env && env[0] == '0') return refuse("disabled by DFQ_LE_RESIDENT=0");
    if (n_relations <= 0) return refuse("no relations");
    if (n_layers > kResMaxLayers) return refuse("too many layers");
    std::vector<int> as_first(n_layers, -1), as_second(n_layers, -1);
    for (int r = 0; r < n_relations; ++r) { as_first[relations[r].first] = r; as_second[relations[r].second] = r; }
    // paired layers, graph order
    std::vector<int> pl_of(n_layers, -1);
    int n_pl = 0;
    for (int l = 0; l < n_layers; ++l) if (as_first[l] >= 0 || as_second[l] >= 0) pl_of[l] = n_pl++;
    int64_t total = 0;
    for (int l = 0; l < n_layers; ++l)
        if (pl_of[l] >= 0) total += (int64_t)layers[l].out_ch * layers[l].in_per_group * layers[l].khkw;

    // residency: how many workgroups of this kernel the chip keeps alive
    int dev = 0, cus = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return refuse("no device properties");
    cus = prop.multiProcessorCount;
    int ns4 = 0, capacity = 0;
    std::vector<ResTile> tiles;
    std::vector<int> tile_begin(n_layers, 0), tile_count(n_layers, 0);
    for (int cand : {8, 16}) {
        int occ = 0;
        hipError_t e = (cand == 8)
            ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)le_resident_kernel<8>, kBlock, kResSmemBytes)
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)le_resident_kernel<16>, kBlock, kResSmemBytes);
        if (e != hipSuccess || occ < 1) continue;
        // the hardware may admit one workgroup per CU fewer than the API says (sgpr granularity): stay one below, and
        // never above 90 % of that
        const int cap_tiles = (int)(0.9 * (double)std::max(1, occ - 1) * cus);
        tiles.clear();
        bool ok = true;
        std::string why;
        for (int l = 0; l < n_layers && ok; ++l) {
            tile_begin[l] = (int)tiles.size();
            tile_count[l] = 0;
            if (pl_of[l] < 0) continue;
            const dfq_layer& L = layers[l];
            const int R = L.out_ch, C = L.in_per_group * L.khkw;
            const int relA = as_second[l], relB = as_first[l];
            int go = R, i2g = L.in_per_group;
            if (relA >= 0) {
                const int o1 = layers[relations[relA].first].out_ch;
                const int Gp = (o1 != i2g) ? (o1 / i2g) : 1;
                go = R / Gp;
            }
            const int vec = (C % 4 == 0 && ((uintptr_t)L.weight & 15u) == 0) ? 4 : 1;
            const int cap = (vec == 4) ? 1024 * cand : 32 * kBlock;
            const Shape sh = pick_shape(R, C, vec, L.khkw, go, i2g, relB >= 0, relA >= 0, cap);
            if (sh.tr < 1) { ok = false; why = "a layer does not tile"; break; }
            const int n_rb = ceil_div_i(R, sh.tr), n_cb = ceil_div_i(C, sh.tc);
            for (int rb = 0; rb < n_rb; ++rb)
                for (int cb = 0; cb < n_cb; ++cb) {
                    ResTile T;
                    memset(&T, 0, sizeof(T));
                    T.w = L.weight;
                    T.n_rows = R; T.row_len = C; T.khkw = L.khkw; T.go = go; T.i2g = i2g;
                    T.r0 = rb * sh.tr; T.nr = std::min(sh.tr, R - T.r0);
                    T.c0 = cb * sh.tc; T.nc = std::min(sh.tc, C - T.c0);
                    T.vec = vec;
                    T.relA = relA; T.relB = relB;
                    T.layer = pl_of[l];
                    T.a_layer = relA >= 0 ? pl_of[relations[relA].first] : -1;
                    T.b_layer = relB >= 0 ? pl_of[relations[relB].second] : -1;
                    T.owner = cb == 0 ? 1 : 0;
                    tiles.push_back(T);
                }
            tile_count[l] = n_rb * n_cb;
        }
        if (!ok) { if (why_not) *why_not = why; continue; }
        if ((int)tiles.size() <= cap_tiles && (int)tiles.size() <= kResMaxTiles) { ns4 = cand; capacity = cap_tiles; break; }
        if (why_not) *why_not = "the network does not fit the register files: " + std::to_string(tiles.size()) + " tiles > " +
                                std::to_string(cap_tiles) + " resident workgroups";
    }
    if (!ns4) return nullptr;
    (void)capacity;
    for (ResTile& T : tiles) {
        // tiles per layer (by paired-layer index)
        for (int l = 0; l < n_layers; ++l) {
            if (pl_of[l] == T.layer) T.nt_self = tile_count[l];
            if (pl_of[l] >= 0 && pl_of[l] == T.a_layer) T.nt_a = tile_count[l];
            if (pl_of[l] >= 0 && pl_of[l] == T.b_layer) T.nt_b = tile_count[l];
        }
    }
    LeResident* r = new LeResident();
    r->n_tiles = (int)tiles.size(); r->n_pl = n_pl; r->n_rels = n_relations; r->n_layers = n_layers; r->ns4 = ns4; r->elements = total;
    // statistics arenas: per relation `channels` = O1 entries of 2 words, two parities; r1 arena then r2 arena
    std::vector<ResRel> hr(n_relations);
    int64_t ch_total = 0;
    for (int q = 0; q < n_relations; ++q) ch_total += layers[relations[q].first].out_ch;
    r->parity_stride = 2 * ch_total;
    int64_t off = 0;
    for (int q = 0; q < n_relations; ++q) {
        const dfq_relation& rr = relations[q];
        hr[q].b1 = layers[rr.first].bias; hr[q].bnw = rr.bn_weight; hr[q].bnb = rr.bn_bias; hr[q].s_cum = rr.scale_cum;
        hr[q].channels = layers[rr.first].out_ch;
        hr[q].r1_off = off;
        hr[q].r2_off = 2 * r->parity_stride + off;
        hr[q].pad = 0;
        off += 2 * (int64_t)hr[q].channels;
    }
    r->stat_words = 4 * r->parity_stride;
    std::vector<ResLayerDiff> ld(n_layers);
    for (int l = 0; l < n_layers; ++l) {
        ld[l].tile_begin = tile_begin[l];
        ld[l].n_tiles = tile_count[l];
        ld[l].n_elems = (double)((int64_t)layers[l].out_ch * layers[l].in_per_group * layers[l].khkw);
    }
    r->sync_words = (size_t)(2 * n_pl + 3) * kResStride;
    bool ok = hipMalloc((void**)&r->d_tiles, sizeof(ResTile) * tiles.size()) == hipSuccess &&
              hipMalloc((void**)&r->d_rels, sizeof(ResRel) * n_relations) == hipSuccess &&
              hipMalloc((void**)&r->d_layer_diff, sizeof(ResLayerDiff) * n_layers) == hipSuccess &&
              hipMalloc((void**)&r->d_stats, sizeof(u64) * (size_t)r->stat_words) == hipSuccess &&
              hipMalloc((void**)&r->d_sync, sizeof(u64) * r->sync_words) == hipSuccess &&
              hipMalloc((void**)&r->d_partials, sizeof(double) * tiles.size()) == hipSuccess &&
              hipMemcpy(r->d_tiles, tiles.data(), sizeof(ResTile) * tiles.size(), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(r->d_rels, hr.data(), sizeof(ResRel) * n_relations, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(r->d_layer_diff, ld.data(), sizeof(ResLayerDiff) * n_layers, hipMemcpyHostToDevice) == hipSuccess;
    if (ok && kResSmemBytes > 48 * 1024) {
        ok = hipFuncSetAttribute((const void*)le_resident_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kResSmemBytes) == hipSuccess &&
             hipFuncSetAttribute((const void*)le_resident_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kResSmemBytes) == hipSuccess;
    }
    if (!ok) { le_resident_destroy(r); return refuse("device allocation failed"); }
    return r;
}

int le_resident_enqueue(LeResident* r, const dfq_le_config* cfg, LeState* d_state, unsigned long long* d_err, int n_sweeps,
                        hipStream_t st) {
    if (!r || !cfg || !d_state || !d_err) return fail_arg("le_resident_enqueue: bad argument");
    if (n_sweeps <= 0) return DFQ_OK;
    // every launch is self-contained: statistics are re-derived from the weights it loads, tags and counters start at zero
    DFQ_HIP_TRY(hipMemsetAsync(r->d_stats, 0, sizeof(u64) * (size_t)r->stat_words, st));
    DFQ_HIP_TRY(hipMemsetAsync(r->d_sync, 0, sizeof(u64) * r->sync_words, st));
    ResArgs a;
    memset(&a, 0, sizeof(a));
    a.tiles = r->d_tiles; a.rels = r->d_rels; a.layer_diff = r->d_layer_diff;
    a.stats = r->d_stats; a.parity_stride = r->parity_stride;
    a.cnt_r = r->d_sync;
    a.cnt_c = r->d_sync + (size_t)r->n_pl * kResStride;
    a.done_cnt = r->d_sync + (size_t)2 * r->n_pl * kResStride;
    a.seq = a.done_cnt + kResStride;
    a.err = d_err;
    a.partials = r->d_partials;
    a.state = d_state;
    a.n_tiles = r->n_tiles; a.n_layers = r->n_layers;
    a.n_sweeps = n_sweeps;
    a.max_sweeps = cfg->max_sweeps;
    a.converge_count = cfg->converge_count;
    a.converge_thres = cfg->converge_thres;
    const LeParams q = make_params(cfg);
    SpinGuard guard(st);
    if (r->ns4 == 8) DFQ_LAUNCH_RESIDENT(le_resident_kernel<8>, dim3(r->n_tiles), dim3(kBlock), kResSmemBytes, st, a, q);
    else DFQ_LAUNCH_RESIDENT(le_resident_kernel<16>, dim3(r->n_tiles), dim3(kBlock), kResSmemBytes, st, a, q);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

}  // namespace dfq
