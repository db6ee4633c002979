// Free-running segments of the streaming equalisation engine (dfq.py:28-117) -- included by dfq_le.hip inside namespace dfq,
// after the tile helpers (vload / vstore / slot_abs_diff / fetch_words).
//
// Statistics that need no exchange.  A layer that a relation scales UNIFORMLY along the axis a statistic reduces over has that
// statistic in closed form: multiplication by s > 0 and float32 rounding are monotone, so max_i fl(w_i * s) == fl(max_i w_i * s),
// bit for bit, sweep after sweep (tests/test_closed_form_statistics.py: against the oracle and against the unmodified
// reference's _layer_equalization).  That covers
//   * the rows of a chain's first layer (rows * s: dfq.py:62),
//   * the columns of a chain's last layer (columns * 1/s: dfq.py:73),
//   * every statistic of a depthwise layer between two relations (channel k is scaled by 1/s_A[k], then by s_B[k]).
// A SEGMENT is a chain start followed by relations linked through such depthwise layers; if its last layer is a chain end, every
// range its relations consume follows from a few scalars per channel taken once from the untouched weights: the scale factors of
// ALL sweeps are an independent recurrence per channel that never reads a weight ("free-running").  MobileNetV2: the expand ->
// depthwise -> project triple of fourteen of its sixteen inverted-residual blocks; ResNet-18: every conv1 -> conv2 pair.
//
// What a sweep still owes the reference for such layers is sum |W - W_prev| (dfq.py:105-108) -- every element, every sweep --
// but nobody waits for it, and with the factors known ahead it does not take a pass per sweep:
//   * le_cf_solve (workgroups of the convergence launch at every G-th sweep, G = the group depth; a launch of its own at a
//     restart) advances the recurrence by G sweeps and leaves s and 1/s of every channel and sweep in the relation's factor ring
//     ([2G sweeps][s, 1/s][channel]);
//   * le_lean_kernel runs at the first sweep k of a group only.  A tile reads its elements ONCE, takes them through the G-1
//     factors of the previous group that are not in the stored values yet (one rounded multiplication each -- the very
//     operations the skipped stores would have performed), applies sweep k and STORES (sweep k is known to happen: the launch
//     saw the verdict of sweep k-1), then carries the values on through sweeps k+1 .. k+G-1 for their |dW| sums only: G partial
//     sums per wave, one per sweep, which the convergence launches of those sweeps pick up.  8 B per element and G sweeps
//     instead of 8 (or, with deferred stores, 5) per sweep; no statistics words, no min/max, no atomics, no in-launch wait.
//   * a loop that stops inside a group leaves up to G-1 sweeps that are not in the stored values: le_flush_kernel applies them
//     (as for the deferred stores of the general tiles) and le_cf_settle_kernel does the same for the relation's [O1] vectors
//     and then sets the applied ring entries to 1, so that the next group's replay changes nothing.
//   * BACKGROUND mode (late round 6; opt-in, DFQ_LE_CF_BG=1: built, bit-identical, measured NO FASTER -- a batch of 32 on one
//     stream 1.783-1.786e10 weights/s against 1.777-1.780e10, with two batches in flight 1.62-1.64e10 against 1.80e10: the
//     second queue's kernel shares the memory system with the sweep launch instead of waiting for its gaps, whatever the
//     stream's priority, and the cross-queue waits cost the two-stream pipeline more than that returns; profiles/r06_experiments.txt 9).
//     The lean launch above sits IN the sweep's chain of launches:
//     ~100 us per 8 sweeps of a batch of 32 during which no general tile runs, while every sweep leaves ~20 us of launch
//     boundaries (sweep launch -> convergence launch -> next sweep launch) in which the chip idles.  Nothing forces that order
//     but the |dW| sums: so the tile looks ahead TWO groups.  The launch of group start k (kLeanBg) still stores the values
//     after sweep k -- known to happen -- but leaves the |dW| sums of sweeps k+G .. k+2G-1; those of k .. k+G-1 were left by
//     the launch before it (the first launch of a run, kLeanFirst, leaves 2G of them).  Its deadline is therefore the
//     convergence launch of sweep k+G, a whole group away, and it runs on a second, low-priority stream next to that group's
//     sweep launches (dfq_le.hip: le_launch_lean, le_bg_deadline, le_bg_join).  The factor ring holds four groups (the
//     launch reads sweeps k-G+1 .. k+2G-1 while the solver may already be writing k+2G .. k+3G-1), the solver runs one group
//     further ahead, the partial sums exist 2G times.  Same float32 operations per element in the same order, same float64
//     partials per wave: bit-identical to the in-line mode, whatever the two streams' interleaving.
// Every value, every |dW| term and every cumulative scale is bit-identical to the general tiles' (same float32 operations in the
// same order per element); only the order in which the float64 partial sums are formed differs, as it does between tile shapes.
#pragma once

constexpr int kCfGroupMax = 8;                  // largest group depth G (a power of two)
constexpr int kCfTab = 256;                     // entries per table (rows of a row tile / (groups x channels) of a column tile)
static_assert(kCfTab <= kBlock, "a lean column tile fills its tables with one entry per thread");
constexpr int kCfMaxRel = 4;                    // relations per segment (longer chains of depthwise layers stay on the general path)

// one relation of a free-running segment
struct LeCfRel {
    float* ring;               // [4G][2][o1]: s and 1/s of every channel for the sweeps of four consecutive groups (cf_slot)
    float* state;              // [4][o1]: (min, max) of the first layer's rows (segment start only), (min, max) of the second layer's columns
    const uint32_t* boot_r1;   // the bootstrap launch's words (parity 0), [o1][2] order-preserving (min, max); null unless segment start
    const uint32_t* boot_r2;
    float* s_cum;              // the relation's [O1] vectors (le_cf_settle_kernel)
    float* bnw;
    float* bnb;
    float* b1;
    int32_t o1;
    int32_t net;
};
struct LeCfSeg {
    int32_t n_rel, o1, net, pad;
    int32_t rel[kCfMaxRel];    // indices into the LeCfRel table, in sweep order
};

// slot of sweep `sweep` (may be negative by less than 4G) in a ring of 4G entries
__device__ __forceinline__ int cf_slot(int sweep, int G) { return (sweep + 4 * G) & (4 * G - 1); }

// what a lean launch does around its group's first sweep k (the stored values have seen the sweeps up to k-G):
//   kLeanInline  replay k-G+1 .. k-1, apply k and store, look ahead k+1 .. k+G-1:   |dW| sums of k .. k+G-1
//   kLeanFirst   (k = 0 of a background plan) ... look ahead 1 .. 2G-1:             |dW| sums of 0 .. 2G-1
//   kLeanBg      ... look ahead k+1 .. k+2G-1:                                      |dW| sums of k+G .. k+2G-1
enum { kLeanInline = 0, kLeanFirst = 1, kLeanBg = 2 };
template <int G, int MODE>
struct LeanShape {
    static constexpr int LOOK = MODE == kLeanInline ? G : 2 * G;     // sweeps from k on (k included)
    static constexpr int ACC0 = MODE == kLeanBg ? G : 0;             // the first of them whose |dW| sum is kept
    static constexpr int NA = LOOK - ACC0;                           // sums kept
    static constexpr int NT = G - 1 + LOOK;                          // factor sets: sweeps k-G+1 .. k+LOOK-1
};

// The recurrence: thread = one channel of one segment, `n_sweeps` sweeps starting at `k0`.  init: the scalars come from the
// bootstrap launch's statistics words instead of the state arrays.
__device__ __forceinline__ void cf_solve_channel(const LeCfSeg& S, const LeCfRel* __restrict__ rels, int c, const LeParams& p,
                                                 int k0, int n_sweeps, int G, bool init) {
    float mn1 = 0.f, mx1 = 0.f, mn2[kCfMaxRel], mx2[kCfMaxRel];
    const LeCfRel* R[kCfMaxRel];
#pragma unroll
    for (int i = 0; i < kCfMaxRel; ++i) R[i] = rels + (i < S.n_rel ? S.rel[i] : S.rel[0]);      // (static indices: the segment stays in registers)
    // every scalar requested before the first use
    if (init) {
        const uint32_t a0 = R[0]->boot_r1[2 * c], a1 = R[0]->boot_r1[2 * c + 1];
        uint32_t b0[kCfMaxRel], b1[kCfMaxRel];
#pragma unroll
        for (int i = 0; i < kCfMaxRel; ++i) { b0[i] = R[i]->boot_r2[2 * c]; b1[i] = R[i]->boot_r2[2 * c + 1]; }
        mn1 = slot_min(a0); mx1 = slot_max(a1);
#pragma unroll
        for (int i = 0; i < kCfMaxRel; ++i) { mn2[i] = slot_min(b0[i]); mx2[i] = slot_max(b1[i]); }
    } else {
        mn1 = R[0]->state[c]; mx1 = R[0]->state[S.o1 + c];
#pragma unroll
        for (int i = 0; i < kCfMaxRel; ++i) { mn2[i] = R[i]->state[2 * S.o1 + c]; mx2[i] = R[i]->state[3 * S.o1 + c]; }
    }
    for (int j = 0; j < n_sweeps; ++j) {
        const int slot = cf_slot(k0 + j, G);
        // the first layer's rows at this relation's turn: the segment start's own rows, then the depthwise layer in between
        // (the previous relation's second layer after ITS column rescale)
        float a_mn = mn1, a_mx = mx1;
#pragma unroll
        for (int i = 0; i < kCfMaxRel; ++i) {
            if (i < S.n_rel) {
                float s, inv;
                le_solve(range_of(a_mn, a_mx, p.signed_range), range_of(mn2[i], mx2[i], p.signed_range), p, s, inv);
                float* ring = R[i]->ring + (int64_t)(2 * slot) * S.o1 + c;
                ring[0] = s;
                ring[S.o1] = inv;
                // dfq.py:62 on the first layer's extrema: the segment start's rows, or the depthwise layer's channel (which is
                // the previous relation's column statistic of the next sweep)
                if (i == 0) { mn1 = a_mn * s; mx1 = a_mx * s; }
                else { mn2[i > 0 ? i - 1 : 0] = a_mn * s; mx2[i > 0 ? i - 1 : 0] = a_mx * s; }
                // dfq.py:73 on the second layer's column extrema: the next relation's rows, or -- the segment's last layer --
                // this relation's own statistic of the next sweep
                a_mn = mn2[i] * inv; a_mx = mx2[i] * inv;
                if (i == S.n_rel - 1) { mn2[i] = a_mn; mx2[i] = a_mx; }
            }
        }
    }
    R[0]->state[c] = mn1; R[0]->state[S.o1 + c] = mx1;
#pragma unroll
    for (int i = 0; i < kCfMaxRel; ++i) {
        if (i < S.n_rel) { R[i]->state[2 * S.o1 + c] = mn2[i]; R[i]->state[3 * S.o1 + c] = mx2[i]; }
    }
}

// workgroup b of the solver: segment block_map[2 b], channels block_map[2 b + 1] * blockDim.x ...
__device__ __forceinline__ void cf_solve_block(const LeCfSeg* __restrict__ segs, const LeCfRel* __restrict__ rels,
                                               const int32_t* __restrict__ block_map, int b, const LeParams& p, int k0,
                                               int n_sweeps, int G, bool init, const LeState* __restrict__ state) {
    const int seg = block_map[2 * b], chunk = block_map[2 * b + 1];
    const LeCfSeg S = segs[seg];
    const int c = chunk * (int)blockDim.x + (int)threadIdx.x;
    if (c >= S.o1) return;
    if (!init && state[S.net].done) return;
    cf_solve_channel(S, rels, c, p, k0, n_sweeps, G, init);
}

__global__ __launch_bounds__(1024) void le_cf_solve_kernel(const LeCfSeg* __restrict__ segs, const LeCfRel* __restrict__ rels,
                                                           const int32_t* __restrict__ block_map, LeParams p, int k0, int n_sweeps,
                                                           int G, int init, const LeState* __restrict__ state) {
    cf_solve_block(segs, rels, block_map, (int)blockIdx.x, p, k0, n_sweeps, G, init != 0, state);
}

// every ring entry 1 (a restart: the first group's replay must change nothing)
__global__ __launch_bounds__(kBlock) void le_cf_ring_reset_kernel(const LeCfRel* __restrict__ rels, int G) {
    const LeCfRel& R = rels[blockIdx.x];
    const int64_t n = (int64_t)8 * G * R.o1;
    for (int64_t i = threadIdx.x; i < n; i += kBlock) R.ring[i] = 1.0f;
}

// sweeps of a network that are not in the stored values of its free-running layers yet: the group of the last sweep began at
// g0 (and that sweep stored), sweeps g0+1 .. T-1 are pending
__device__ __forceinline__ int cf_pending(int sweeps_done, int G, int* first_slot) {
    if (sweeps_done <= 0) { *first_slot = 0; return 0; }
    const int last = sweeps_done - 1;
    const int g0 = last & ~(G - 1);
    *first_slot = cf_slot(g0 + 1, G);
    return last - g0;
}

// After le_flush_kernel: the pending factors go into the relation's [O1] vectors (relation.py:20-24, dfq.py:64-71) and the ring
// entries of the applied sweeps are set to 1.  One workgroup per free-running relation.
__global__ __launch_bounds__(kBlock) void le_cf_settle_kernel(const LeCfRel* __restrict__ rels, const LeState* __restrict__ state, int G) {
    const LeCfRel R = rels[blockIdx.x];
    int slot0;
    const int pend = cf_pending(state[R.net].sweeps, G, &slot0);
    if (pend == 0) return;
    for (int c = threadIdx.x; c < R.o1; c += kBlock) {
        float f[kCfGroupMax];
#pragma unroll
        for (int j = 0; j < kCfGroupMax; ++j) f[j] = (j < pend) ? R.ring[(int64_t)(2 * (slot0 + j)) * R.o1 + c] : 1.0f;
        float v_cum = R.s_cum[c];
        float v_bnw = R.bnw ? R.bnw[c] : 0.f, v_bnb = R.bnb ? R.bnb[c] : 0.f, v_b1 = R.b1 ? R.b1[c] : 0.f;
#pragma unroll
        for (int j = 0; j < kCfGroupMax; ++j) {
            if (j < pend) { v_cum = v_cum * f[j]; v_bnw = v_bnw * f[j]; v_bnb = v_bnb * f[j]; v_b1 = v_b1 * f[j]; }
        }
        R.s_cum[c] = v_cum;
        if (R.bnw) R.bnw[c] = v_bnw;
        if (R.bnb) R.bnb[c] = v_bnb;
        if (R.b1) R.b1[c] = v_b1;
#pragma unroll
        for (int j = 0; j < kCfGroupMax; ++j) {
            if (j < pend) {
                R.ring[(int64_t)(2 * (slot0 + j)) * R.o1 + c] = 1.0f;
                R.ring[(int64_t)(2 * (slot0 + j) + 1) * R.o1 + c] = 1.0f;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// lean tiles
// ---------------------------------------------------------------------------------------------
// One tile of a free-running layer: self-contained (ONE wave-wide load, then v_readlane), nothing else is looked up.
enum { kLeanRow4 = 0, kLeanRow1 = 1, kLeanShort0 = 2, kLeanCol4 = 3, kLeanCol1 = 4, kLeanShort1 = 5 };
struct alignas(128) LeLeanRef {
    float* w;                // first element of the tile (thread-per-row kinds: first row of the tile)
    const float* ring;       // the relation's factor ring
    const float* ring_prev;  // kLeanShort0 of a depthwise layer between two relations: the previous relation's ring (its 1/s goes first), else null
    float* s_cum;            // tiles that own their rows' [O1] entries (first slab of a row tile, kLeanShort0): the relation's vectors, else null
    float* bnw;
    float* bnb;
    float* b1;
    int32_t kind;
    int32_t net;
    int32_t nr, np;          // rows and floats per row of the tile (thread-per-row kinds: np = the row's length)
    int32_t stride;          // floats between two rows of the layer
    int32_t r0, p0;          // first row, first position in the row
    int32_t o1;              // channels of the relation (ring stride)
    int32_t go, gi, khkw;    // column kinds: element (o, p) is scaled by channel (o / go) * gi + p / khkw
    int32_t o1_prev;         // channels of the previous relation (stride of ring_prev)
    int32_t slot;            // the tile's partial-sum slot
    int32_t pc_gi;           // kLeanShort0 behind a relation: row o is channel o * pc_gi of it
    int32_t pad[4];
};
static_assert(sizeof(LeLeanRef) == 128, "one lean tile reference per 128-byte line");
constexpr int kLeanWords = 28;

struct LeanArgs {
    int32_t k;               // the group's first sweep (sweeps since the last restart)
    int32_t pad;
    int64_t part_stride;     // doubles between the partial arrays of two sweeps (2G arrays: sweep j uses array j mod 2G)
};

// the [O1] entries of channel c for the sweeps up to and including k (the later ones are not known to happen):
// relation.py:20-24, dfq.py:64-71; f[0 .. G-2] = the previous group's pending sweeps, f[G-1] = sweep k.  Requested early
// (lean_vectors_load, in front of the tile's elements: memory returns in order), finished when the factors are there.
struct LeanVec { float cum, bnw, bnb, b1; };
__device__ __forceinline__ LeanVec lean_vectors_load(const LeLeanRef& T, int c) {
    LeanVec v;
    v.cum = T.s_cum[c];
    v.bnw = T.bnw ? T.bnw[c] : 0.f; v.bnb = T.bnb ? T.bnb[c] : 0.f; v.b1 = T.b1 ? T.b1[c] : 0.f;
    return v;
}
template <int G, int NT>
__device__ __forceinline__ void lean_vectors_finish(const LeLeanRef& T, int c, LeanVec v, const float (&f)[NT]) {
#pragma unroll
    for (int j = 0; j < G; ++j) { v.cum = v.cum * f[j]; v.bnw = v.bnw * f[j]; v.bnb = v.bnb * f[j]; v.b1 = v.b1 * f[j]; }
    T.s_cum[c] = v.cum;
    if (T.bnw) T.bnw[c] = v.bnw;
    if (T.bnb) T.bnb[c] = v.bnb;
    if (T.b1) T.b1[c] = v.b1;
}

// rows * s: [nr x np] block, lanes along the row, a thread walks down the rows (as row_tile)
template <int VEC, int G, int MODE>
__device__ __forceinline__ void lean_row(const LeLeanRef& T, const LeanArgs& A, float* sh_f, double (&acc)[LeanShape<G, MODE>::NA]) {
    typedef LeanShape<G, MODE> SH;
    constexpr int NV = kSlotsVec4;
    constexpr int NT = SH::NT;
    const int tid = threadIdx.x;
    const int nr = T.nr;
    const int npv = T.np / VEC;
    const int JL = small_div(kBlock, npv);
    const int jl_raw = small_div(tid, npv);
    const bool lane_on = jl_raw < JL;
    const int jl = lane_on ? jl_raw : 0;
    const int pos = lane_on ? (tid - jl_raw * npv) * VEC : 0;
    const int n_own = lane_on ? small_div(nr - jl + JL - 1, JL) : 0;
    const int n_max = min(NV, small_div(nr + JL - 1, JL));
    gfloat* const w = (gfloat*)T.w + pos;
    // the rows' factors are requested IN FRONT of the elements (memory returns in order: behind them, the table could only be
    // filled once the whole tile had arrived)
    const int c = T.r0 + min(tid, nr - 1);
    float f[NT];
    LeanVec vec{0.f, 0.f, 0.f, 0.f};
    if (tid < nr) {
        const gfloat* const ring = (const gfloat*)T.ring + c;
#pragma unroll
        for (int j = 0; j < NT; ++j) f[j] = ring[(int64_t)(2 * cf_slot(A.k - G + 1 + j, G)) * T.o1];
        if (T.s_cum) vec = lean_vectors_load(T, c);
    }
    float v[NV][VEC];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        if (u < n_max) vload<VEC>(w + min(jl + u * JL, nr - 1) * T.stride, v[u]);
    }
    if (tid < nr) {
#pragma unroll
        for (int j = 0; j < NT; ++j) sh_f[j * kCfTab + tid] = f[j];
        if (T.s_cum) lean_vectors_finish<G, NT>(T, c, vec, f);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        if (u >= n_max) continue;
        const int r = min(jl + u * JL, nr - 1);
        const bool ok = u < n_own;
        float h[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) h[j] = sh_f[j * kCfTab + r];
        float x[VEC], nv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) x[e] = v[u][e];
#pragma unroll
        for (int j = 0; j < G - 1; ++j) {                      // the previous group's sweeps the stored values have not seen
#pragma unroll
            for (int e = 0; e < VEC; ++e) x[e] = x[e] * h[j];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) nv[e] = x[e] * h[G - 1];  // sweep k: dfq.py:62
        if (ok) vstore<VEC>(w + r * T.stride, nv);
        if (SH::ACC0 == 0) acc[0] += slot_abs_diff<VEC>(ok, nv, x);
#pragma unroll
        for (int i = 1; i < SH::LOOK; ++i) {                   // the sweeps behind k: their |dW| only (from ACC0 on)
            float y[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) y[e] = nv[e] * h[G - 1 + i];
            if (i >= SH::ACC0) acc[i - SH::ACC0] += slot_abs_diff<VEC>(ok, y, nv);
#pragma unroll
            for (int e = 0; e < VEC; ++e) nv[e] = y[e];
        }
    }
}

// columns * 1/s: [nr x np] block, G2 = pow2 >= np / VEC lanes share a row (as col_tile)
template <int VEC, int G, int MODE>
__device__ __forceinline__ void lean_col(const LeLeanRef& T, const LeanArgs& A, float* sh_f, int* sh_tab, double (&acc)[LeanShape<G, MODE>::NA]) {
    typedef LeanShape<G, MODE> SH;
    constexpr int NV = kSlotsVec4;
    constexpr int NT = SH::NT;
    const int tid = threadIdx.x;
    const int nr = T.nr, np = T.np;
    const int npv = np / VEC;
    int G2 = 1, lg = 0;
    while (G2 < npv) { G2 <<= 1; ++lg; }
    const int n_rowslots = kBlock >> lg;
    const int grp = tid >> lg;
    const int ln = tid - grp * G2;
    const bool lane_on = ln < npv;
    const int pos = min(ln, npv - 1) * VEC;
    const int n_max = min(NV, (nr + n_rowslots - 1) >> (8 - lg));
    gfloat* const w = (gfloat*)T.w + pos;
    float v[NV][VEC];
    // 1/s tables: (groups spanned by the rows) x (input channels spanned by the columns), one table per sweep
    const int i0 = small_div(T.p0, T.khkw);
    const int nci = small_div(T.p0 + np - 1, T.khkw) - i0 + 1;
    const int g_lo = small_div(T.r0, T.go);
    const int g_n = small_div(T.r0 + nr - 1, T.go) - g_lo + 1;
    int ci[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) ci[e] = small_div(T.p0 + pos + e, T.khkw) - i0;
    if (g_n == 1) {
        // one group (every ungrouped layer): a thread's columns are the same in all of its rows -- their factors come straight
        // from the ring into registers, no table, no barrier; requested in front of the elements (memory returns in order)
        const int c0 = g_lo * T.gi + i0;
        float h[NT][VEC];
        // a pointwise layer's four columns are four consecutive channels: one 16-byte request per sweep
        const bool wide = VEC == 4 && T.khkw == 1 && (T.o1 & 3) == 0 && ((c0 + ci[0]) & 3) == 0 && (((uintptr_t)T.ring) & 15u) == 0;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const gfloat* const ring = (const gfloat*)T.ring + (int64_t)(2 * cf_slot(A.k - G + 1 + j, G) + 1) * T.o1 + c0;
            if (wide) {
                const fvec4 t = *(const gfvec4*)(ring + ci[0]);
#pragma unroll
                for (int e = 0; e < VEC; ++e) h[j][e] = t[e % 4];
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) h[j][e] = ring[ci[e]];
            }
        }
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            if (u < n_max) vload<VEC>(w + min(grp + u * n_rowslots, nr - 1) * T.stride, v[u]);
        }
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            if (u >= n_max) continue;
            const int r_raw = grp + u * n_rowslots;
            const int r = min(r_raw, nr - 1);
            const bool ok = lane_on && r_raw < nr;
            float x[VEC], nv[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) x[e] = v[u][e];
#pragma unroll
            for (int j = 0; j < G - 1; ++j) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) x[e] = x[e] * h[j][e];
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) nv[e] = x[e] * h[G - 1][e];       // sweep k: dfq.py:73
            if (ok) vstore<VEC>(w + r * T.stride, nv);
            if (SH::ACC0 == 0) acc[0] += slot_abs_diff<VEC>(ok, nv, x);
#pragma unroll
            for (int i = 1; i < SH::LOOK; ++i) {
                float y[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) y[e] = nv[e] * h[G - 1 + i][e];
                if (i >= SH::ACC0) acc[i - SH::ACC0] += slot_abs_diff<VEC>(ok, y, nv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) nv[e] = y[e];
            }
        }
        return;
    }
    {
        // (<= kCfTab = kBlock entries by plan: one per thread; requested in front of the elements)
        const int idx = min(tid, g_n * nci - 1);
        const int gq = small_div(idx, nci);
        const int c = (g_lo + gq) * T.gi + i0 + (idx - gq * nci);
        const gfloat* const ring = (const gfloat*)T.ring + T.o1 + c;
        float f[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) f[j] = ring[(int64_t)(2 * cf_slot(A.k - G + 1 + j, G)) * T.o1];
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            if (u < n_max) vload<VEC>(w + min(grp + u * n_rowslots, nr - 1) * T.stride, v[u]);
        }
        if (tid < g_n * nci) {
#pragma unroll
            for (int j = 0; j < NT; ++j) sh_f[j * kCfTab + idx] = f[j];
        }
    }
    if (tid < nr) sh_tab[tid] = (small_div(T.r0 + tid, T.go) - g_lo) * nci;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        if (u >= n_max) continue;
        const int r_raw = grp + u * n_rowslots;
        const int r = min(r_raw, nr - 1);
        const bool ok = lane_on && r_raw < nr;
        const int t0 = sh_tab[r];
        float x[VEC], nv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) x[e] = v[u][e];
#pragma unroll
        for (int j = 0; j < G - 1; ++j) {
            const float* tab = sh_f + j * kCfTab + t0;
#pragma unroll
            for (int e = 0; e < VEC; ++e) x[e] = x[e] * tab[ci[e]];
        }
        {
            const float* tab = sh_f + (G - 1) * kCfTab + t0;   // sweep k: dfq.py:73
#pragma unroll
            for (int e = 0; e < VEC; ++e) nv[e] = x[e] * tab[ci[e]];
            if (ok) vstore<VEC>(w + r * T.stride, nv);
            if (SH::ACC0 == 0) acc[0] += slot_abs_diff<VEC>(ok, nv, x);
        }
#pragma unroll
        for (int i = 1; i < SH::LOOK; ++i) {
            const float* tab = sh_f + (G - 1 + i) * kCfTab + t0;
            float y[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) y[e] = nv[e] * tab[ci[e]];
            if (i >= SH::ACC0) acc[i - SH::ACC0] += slot_abs_diff<VEC>(ok, y, nv);
#pragma unroll
            for (int e = 0; e < VEC; ++e) nv[e] = y[e];
        }
    }
}

// one THREAD per row (depthwise k x k kernels).  side 0: first layer of the relation (rows * s; behind a relation also * 1/s of
// that one first: dfq.py:73 then dfq.py:62, two roundings); side 1: depthwise second layer at a chain's end (rows * 1/s).
template <int side, int G, int MODE>
__device__ __forceinline__ void lean_short(const LeLeanRef& T, const LeanArgs& A, double (&acc)[LeanShape<G, MODE>::NA]) {
    typedef LeanShape<G, MODE> SH;
    constexpr int NT = SH::NT;
    const int tid = threadIdx.x;
    const bool ok = tid < T.nr;
    const int o = T.r0 + min(tid, T.nr - 1);
    const int len = T.np;
    gfloat* const w = (gfloat*)T.w + (int64_t)min(tid, T.nr - 1) * len;
    const bool fused = side == 0 && T.ring_prev != nullptr;
    const int c = side == 0 ? o : small_div(o, T.go) * T.gi;
    float f[NT], pf[NT];
    {
        const gfloat* const ring = (const gfloat*)T.ring + (side == 0 ? 0 : T.o1) + c;
        const gfloat* const prev = fused ? (const gfloat*)T.ring_prev + T.o1_prev + o * T.pc_gi : (const gfloat*)T.ring;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int slot = cf_slot(A.k - G + 1 + j, G);
            f[j] = ring[(int64_t)(2 * slot) * T.o1];
            pf[j] = fused ? prev[(int64_t)(2 * slot) * T.o1_prev] : 1.0f;                      // (* 1.0f is exact)
        }
    }
    if (side == 0 && ok && T.s_cum) lean_vectors_finish<G, NT>(T, c, lean_vectors_load(T, c), f);
    for (int k0 = 0; k0 < len; k0 += kShortChunk) {
        float x[kShortChunk];
#pragma unroll
        for (int e = 0; e < kShortChunk; ++e) x[e] = w[min(k0 + e, len - 1)];
#pragma unroll
        for (int e = 0; e < kShortChunk; ++e) {
            const bool in = ok && k0 + e < len;
            float nv = x[e];
#pragma unroll
            for (int j = 0; j < G - 1; ++j) { nv = nv * pf[j]; nv = nv * f[j]; }
#pragma unroll
            for (int i = 0; i < SH::LOOK; ++i) {
                const float y = (nv * pf[G - 1 + i]) * f[G - 1 + i];
                if (i == 0 && in) w[k0 + e] = y;
                if (i >= SH::ACC0) acc[i - SH::ACC0] += (double)abs_diff_if(in, y, nv);
                nv = y;
            }
        }
    }
}

// tile `idx` of the lean table.  sh_f: LeanShape::NT * kCfTab floats, sh_tab: kTileRowsMax ints of LDS.
template <int G, int MODE = kLeanInline>
__device__ __forceinline__ void lean_tile_run(const LeLeanRef* __restrict__ refs, int idx, const LeanArgs& A, const LeState* __restrict__ state,
                                              double* __restrict__ partials, float* sh_f, int* sh_tab) {
    const int lane = threadIdx.x % kWave;
    const uint32_t word = fetch_words(refs + idx, kLeanWords, lane);
    LeLeanRef T;
    {
        auto ptr = [&](int i) {
            const uint64_t lo = (uint32_t)__builtin_amdgcn_readlane(word, i), hi = (uint32_t)__builtin_amdgcn_readlane(word, i + 1);
            return (uintptr_t)(lo | (hi << 32));
        };
        T.w = (float*)ptr(0); T.ring = (const float*)ptr(2); T.ring_prev = (const float*)ptr(4);
        T.s_cum = (float*)ptr(6); T.bnw = (float*)ptr(8); T.bnb = (float*)ptr(10); T.b1 = (float*)ptr(12);
        T.kind = __builtin_amdgcn_readlane(word, 14); T.net = __builtin_amdgcn_readlane(word, 15);
        T.nr = __builtin_amdgcn_readlane(word, 16); T.np = __builtin_amdgcn_readlane(word, 17);
        T.stride = __builtin_amdgcn_readlane(word, 18); T.r0 = __builtin_amdgcn_readlane(word, 19);
        T.p0 = __builtin_amdgcn_readlane(word, 20); T.o1 = __builtin_amdgcn_readlane(word, 21);
        T.go = __builtin_amdgcn_readlane(word, 22); T.gi = __builtin_amdgcn_readlane(word, 23);
        T.khkw = __builtin_amdgcn_readlane(word, 24); T.o1_prev = __builtin_amdgcn_readlane(word, 25);
        T.slot = __builtin_amdgcn_readlane(word, 26); T.pc_gi = __builtin_amdgcn_readlane(word, 27);
    }
    typedef LeanShape<G, MODE> SH;
    // Does sweep k happen?  In line, the launch sits behind the convergence launch of sweep k-1: `done` is what that one left.
    // A background launch runs next to LATER convergence launches, which may raise `done` while some of its workgroups have
    // not started yet: it asks for the token the convergence launch of sweep k-1 left instead ("sweep k happens": `happen`
    // only grows during a run).  (uniform)
    if (MODE == kLeanBg ? state[T.net].happen < A.k : state[T.net].done != 0) return;
    double acc[SH::NA];
#pragma unroll
    for (int i = 0; i < SH::NA; ++i) acc[i] = 0.0;
    switch (T.kind) {
        case kLeanRow4: lean_row<4, G, MODE>(T, A, sh_f, acc); break;
        case kLeanRow1: lean_row<1, G, MODE>(T, A, sh_f, acc); break;
        case kLeanShort0: lean_short<0, G, MODE>(T, A, acc); break;
        case kLeanCol4: lean_col<4, G, MODE>(T, A, sh_f, sh_tab, acc); break;
        case kLeanCol1: lean_col<1, G, MODE>(T, A, sh_f, sh_tab, acc); break;
        default: lean_short<1, G, MODE>(T, A, acc); break;
    }
    // one partial per wave and sweep (fixed butterfly order -> deterministic); sweep j is read by ITS convergence launch
#pragma unroll
    for (int i = 0; i < SH::NA; ++i) {
        const double t = wave_sum(acc[i]);
        if (lane == 0)
            partials[(int64_t)((A.k + SH::ACC0 + i) & (2 * G - 1)) * A.part_stride + (int64_t)T.slot * (kBlock / kWave) + threadIdx.x / kWave] = t;
    }
}

// One launch per group of G sweeps: grid = the lean tiles of every free-running layer of the plan.  (With the default depth the
// lean tiles are instead woven into the sweep's own launch at the group's first sweep -- le_level_kernel -- where their
// arithmetic overlaps the general tiles' memory traffic; this kernel serves the other depths and the per-level launches.)
template <int G, int MODE>
__global__ __launch_bounds__(kBlock) void le_lean_kernel(const LeLeanRef* __restrict__ refs, LeanArgs A, const LeState* __restrict__ state,
                                                         double* __restrict__ partials) {
    __shared__ float sh_f[LeanShape<G, MODE>::NT * kCfTab];
    __shared__ int sh_tab[kTileRowsMax];
    lean_tile_run<G, MODE>(refs, (int)blockIdx.x, A, state, partials, sh_f, sh_tab);
}
