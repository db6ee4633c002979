// Cross-layer weight equalisation (dfq.py:28-75 _layer_equalization, dfq.py:78-117 sweep loop)
// for gfx950.
//
// The reference visits every paired channel c of a relation (W1 = first layer [O1, row_len],
// W2 = second layer [O2, I2/g, khkw]) and does: r1 = range(W1[c]), r2 = range(W2[:, c]),
// s = solve(r1, r2), W1[c] *= s, W2[:, c] *= 1/s.  Computing a range needs a full row / a full
// strided column, scaling needs the scale: two dependent passes over the data per relation.
//
// Stat forwarding.  fl(v * s) is monotone in v for s > 0, and every write of a layer is done by
// exactly one of these kernels, so the ranges a relation will need are produced as a by-product of
// the PREVIOUS write of that layer:
//   * the column rescale of W2 (which is the first layer of the next relation in its chain) emits
//     the new per-row min/max of W2                       -> row stats R1 of the next relation;
//   * the row rescale of W1 (second layer of the previous relation in its chain) emits the new
//     per-input-channel min/max of W1                     -> column stats R2 of the previous
//     relation, consumed one sweep later;
//   * a layer that is only ever scaled one way (chain start / chain end) has its stats forwarded
//     arithmetically: min' = fl(min * s), max' = fl(max * s) -- exact by monotonicity.
// A sweep is therefore ONE streaming launch: every weight is read and written once (interior layers
// are read once more, see below), tiles are small 2-D blocks (unit-stride 16-byte lanes), and the
// scale of a channel is re-derived from four stat words by every tile that needs it (identical IEEE
// operations -> identical value everywhere).  Results are bit-identical to the two-pass formulation
// because min/max are exact.
//
//   level     = relations that share no layer (Gauss-Seidel order of dfq.py:85 kept between
//               levels); the workgroups of a sweep are listed level after level and a workgroup
//               waits for the tiles of the earlier relation it depends on (le_level_kernel);
//   row tile  = [rt_rows x rt_cols] block of W1, scaled per row, emits per-input-channel stats;
//   col tile  = [ct_rows x ct_cols] block of W2, scaled per input channel, emits per-row stats.
//   In both, lanes run along the contiguous row positions and every thread walks down the rows
//   with all of its elements loaded into registers before the scale solve.
//
// Partial stats of a tile are merged with order-preserving atomicMax words (identity 0); the
// buffers alternate between two parities and the control kernel clears the one that is about to
// be accumulated into.
//
// Interior layers.  A layer that is the second layer of relation A and the first layer of relation B
// (every layer inside a chain) is rescaled twice per sweep: columns by 1/s_A, then rows by s_B, and s_B
// needs the row ranges of the column-rescaled matrix.  It is NOT written twice: in A's level a
// read-only pass computes the row statistics of t = fl(w * 1/s_A) (4 B per element), in B's level one
// pass recomputes t and stores fl(t * s_B) -- the reference's two roundings, in its order -- while
// emitting the new column statistics (8 B per element).  12 B per interior element and sweep instead
// of 24 B for two read-write passes plus a pre-sweep snapshot.
//
// Deferred stores.  A layer that is only ever scaled one way (chain start: rows by s; chain end: columns by 1/s) is read
// by nobody but its own tile between two sweeps -- its ranges are forwarded arithmetically -- so its store can wait: with
// depth D (LeParams::defer, a power of two; DFQ_LE_DEFER; default 4 for batched plans, 1 = off for single networks) sweep k with k % D != D-1 only reads the
// tile, forms nv = fl(t * s_k) for the |dW| sum and remembers s_k per channel (`hold`), where t is the stored value taken
// through the remembered factors of the skipped sweeps one rounding at a time -- the very float32 operations the
// reference performs, in its order, so every value and every |dW| term is bit-identical; sweep k % D == D-1 replays the
// same way and stores.  4 (D + 1) / D bytes per element and sweep instead of 8.  A loop that ends between two stores
// leaves up to D-1 pending factors: le_flush_kernel applies them (launched at the end of every enqueue call, gated per
// network on its sweep count) and le_hold_reset_kernel sets the remembered factors back to 1.
//
// Convergence (dfq.py:105-115): every element is written exactly once per sweep, by a pass that has
// its pre-sweep value in a register, so each tile leaves float64 partials of sum|W - W_prev|; they are
// reduced in a fixed order by a one-workgroup control kernel that also advances the reference's
// (diff, count) state machine on the device.  Level kernels start by reading `done`, so the host enqueues sweeps
// ahead without synchronising.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "dfq_common.hpp"
#include "dfq_le_shared.hpp"
#include "dfq_le_resident.hpp"

namespace dfq {

// register slots (vectors) a thread preloads: 4 x float4 or 8 x float.  Kept small on purpose: every
// slot is unrolled code, and a cold workgroup pays instruction-fetch latency for every line it walks.
#ifndef DFQ_LE_SLOTS
#define DFQ_LE_SLOTS 8
#endif
constexpr int kSlotsVec4 = DFQ_LE_SLOTS;
constexpr int kSlotsVec1 = DFQ_LE_SLOTS;
static_assert(kSlotsVec4 == kSlotsVec1, "the tile functions take one register-array shape");
constexpr int kTileRowsMax = 256;      // rows of a tile (one LDS entry per row)
constexpr int kRowTileColsMax = 128;   // positions of a row tile (x2 for float4 tiles)
constexpr int kColTileLanes = 64;      // vector positions of a col tile (one wave per row at most)
constexpr int kSlotMax = 1024;         // LDS table entries of a tile (stat slots / 1/s table)
constexpr int kShortChunk = 9;          // taps a thread-per-row tile keeps in flight (one 3x3 kernel)
constexpr int kHoldMax = 3;             // remembered factors per channel: deferred stores of depth <= 4
constexpr int kBootTc = 256;           // channels per bootstrap tile (1 KB of a pointwise second layer's row)
#ifndef DFQ_BOOT_ABLATE
#define DFQ_BOOT_ABLATE 0
#endif
constexpr int kBootAblate = DFQ_BOOT_ABLATE;   // tuning builds: 1 no row pass, 2 no wide column pass, 4 no generic column pass
constexpr int kBootWork = 16384;       // elements per pass a bootstrap workgroup streams at most (larger blocks are split)
constexpr int kCtlBlock = 1024;        // threads of the control kernel
constexpr int kCtlStage = 6144;        // partials staged in LDS by the control kernel


// One relation.  The descriptors of a launch are passed BY VALUE (kernarg segment) and selected with
// blockIdx.y, so a workgroup reaches its descriptor with the kernel's very first scalar loads: no
// table lookup, no dependent fetch.
struct LeRelDev {
    float* w1;
    float* w2;
    float* b1;
    float* bnw;
    float* bnb;
    float* s_cum;
    float* hold;         // remembered factors of the skipped sweeps (see `defer` below), or null
    uint32_t* prev_r1;   // W1 is interior: row stats (R1) of the relation whose second layer it is, else null
    uint32_t* r1;        // parity 0 of the row stats of W1: [o1][2] = (min slot, max slot); parity 1 is
    uint32_t* r2;        // `stat_stride` words further.  r2: column stats of W2 per paired channel
    uint32_t* out_cols;  // R2 of the relation whose SECOND layer is our W1, or null: forward r1
    uint32_t* out_rows;  // R1 of the relation whose FIRST layer is our W2, or null: forward r2
    int64_t stat_stride; // words between the two parities of a stat arena
    int32_t o1, row_len, khkw1;
    int32_t pc_go, pc_gi;                // W1 element (o, i) -> channel (o / pc_go) * pc_gi + i of out_cols
    int32_t o2, gi, go, i2g, khkw;       // W2 geometry; paired channel c = g*gi + ii
    int32_t rt_rows, rt_cols, rt_slabs, rt_vec, n_row_tiles;   // *_vec: 4 float4 tiles, 1 scalar tiles, 0 thread-per-row
    int32_t ct_rows, ct_cols, ct_slabs, ct_vec, n_col_tiles;
    int32_t w1_interior;    // W1 is also the second layer of an earlier relation: its row pass applies both rescales
    int32_t w2_interior;    // W2 is also the first layer of a later relation: its column pass only takes statistics
    int32_t partial_base;   // first partial slot of this relation (row tiles, then col tiles)
    int32_t boot_begin;     // first workgroup inside the bootstrap launch
    int32_t boot_tiles;     // channel blocks x boot_split
    int32_t boot_split;     // workgroups that share one block of kBootTc channels (long rows / tall columns are cut up)
    int32_t net;            // which network of a batched plan (index into the loop-state array)
    // one launch per sweep: the tiles of this relation may start once the column tiles of the relation that
    // produces its row statistics have finished -- dep_counter >= dep_tiles * (sweep + 1)
    int32_t dep_idx;        // index of that relation's counter, or -1
    int32_t dep_tiles;      // its column tiles per sweep
    int32_t counter_idx;    // own counter (bumped by this relation's column tiles if somebody waits for them), or -1
    // deferred stores (see "Deferred stores" in the header comment): bit 0 = W1, bit 1 = W2 is scaled one way only and its
    // store is skipped in all but every `LeParams::defer`-th sweep; hold[(2 j + side) * o1 + c] = factor of channel c in the
    // j-th skipped sweep since the last store
    int32_t defer;
    // thread-per-row W1 that is interior: the thread holds its whole row, so it takes the row range of t = fl(w / s_prev) itself
    // instead of reading the row statistics a read-only pass of the previous relation would have published -- that pass (one
    // workgroup per 256 rows, a wait inside the launch, 4 B per element) is then not launched at all (round 4)
    // The row statistics such a relation's COLUMN tiles solve their 1/s from are then published by its own row tiles (device-scope
    // atomics into r1, like the pass they replace) and the column tiles wait for THEM: cdep_idx / cdep_tiles replace dep_idx /
    // dep_tiles on the column side, rcounter_idx is the counter the row tiles bump (-1 everywhere else: both sides share dep_idx).
    int32_t local_r1;
    int32_t cdep_idx, cdep_tiles, rcounter_idx;
};

// Tuning builds only (tools/ablate.sh): -DDFQ_LE_ABLATE=bits switches parts of the tile kernels off at compile
// time to see what a launch's time is made of.  1: no data stores, 2: no |dW| accumulation, 4: no statistics
// emission, 8: no data loads.  The product library is built with 0.
#ifndef DFQ_LE_ABLATE
#define DFQ_LE_ABLATE 0
#endif
constexpr int kAblate = DFQ_LE_ABLATE;
// 1 = the tiles' 16-byte element loads carry the non-temporal hint, 2 = their stores do, 3 = both (dfq_common.hpp);
// +4 = not the loads of the read-only passes over interior layers (the row pass of the next relation reads them again)
#ifndef DFQ_LE_NT
#define DFQ_LE_NT 3
#endif
constexpr int kNonTemporal = DFQ_LE_NT;

struct LeLayerDiff {
    int32_t partial_begin;   // -1: layer untouched by any relation (contributes exactly 0)
    int32_t n_partials;
    double n_elems;
};

// one working workgroup of a launch: which relation (index into the descriptor table), which tile of it, which
// network (loop-state index)
struct LeBlockRef {
    int32_t rel;
    int32_t tile;     // row tiles first, then col tiles
    int32_t net;
    int32_t pad;
};
// A launch is a 1-D grid of exactly the working workgroups: workgroup b reads blocks[b] (16 bytes) and then, with
// ONE wave-wide load, its descriptor table[rel] and the loop state of its network.  (A rectangular grid over
// hundreds of relations of very different sizes would be mostly empty workgroups that each pay a global round
// trip to find that out.)

// optional per-phase cycle stamps of one workgroup (dfq_le_trace, tuning aid; null in production)
struct LeTrace {
    long long* out;     // device [16] or null
    int32_t block;
    int32_t flat;       // index of the running tile in the launch's table (set by the kernel)
};
__device__ __forceinline__ void stamp(const LeTrace& tr, int slot) {
    if (!tr.out || threadIdx.x != 0) return;
    const int flat = tr.flat;
    if (tr.block == -1) {
        // every workgroup: [0] entry, [1] exit (100 MHz wall clock), [2] XCC_ID << 32 | HW_ID
        if (slot == 0) {
            tr.out[3 * (int64_t)flat + 0] = wall_clock64();
            tr.out[3 * (int64_t)flat + 2] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                            (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        } else if (slot == 7) {
            tr.out[3 * (int64_t)flat + 1] = wall_clock64();
        }
    } else if (flat == tr.block) {
        tr.out[slot] = clock64();
    }
}

// Row statistics are produced and consumed inside ONE launch (by workgroups on different XCDs, whose L2s are not
// coherent for plain accesses within a kernel): they are published with device-scope atomics and read with
// device-scope loads (tools/litmus/xcd_flag.hip checks exactly this protocol on the hardware).  Column
// statistics cross a kernel boundary and use plain loads.
__device__ __forceinline__ uint32_t ld_stat(const guint* p) {
    return __hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// scale of paired channel c from the four stat words of the current parity
__device__ __forceinline__ void channel_scale(const LeRelDev& R, const LeParams& p, int cur, int c, float& s,
                                              float& inv, float& mn1, float& mx1, float& mn2, float& mx2) {
    const guint* a = (const guint*)R.r1 + (int64_t)cur * R.stat_stride + 2 * c;
    const guint* b = (const guint*)R.r2 + (int64_t)cur * R.stat_stride + 2 * c;
    const uint32_t a0 = ld_stat(a), a1 = ld_stat(a + 1), b0 = b[0], b1 = b[1];
    mn1 = slot_min(a0); mx1 = slot_max(a1);
    mn2 = slot_min(b0); mx2 = slot_max(b1);
    le_solve(range_of(mn1, mx1, p.signed_range), range_of(mn2, mx2, p.signed_range), p, s, inv);
}

__device__ __forceinline__ void scale_from_words(const LeParams& p, uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1,
                                                 float& s, float& inv, float& mn1, float& mx1, float& mn2, float& mx2) {
    mn1 = slot_min(a0); mx1 = slot_max(a1);
    mn2 = slot_min(b0); mx2 = slot_max(b1);
    le_solve(range_of(mn1, mx1, p.signed_range), range_of(mn2, mx2, p.signed_range), p, s, inv);
}

// |a - b| if `on`, else 0 -- the contribution of one element to sum|W - W_prev|.  The select is done
// on the float32 difference and the magnitude taken afterwards by clearing the sign bit: the pattern
// select(on, (double)fabsf(d), 0.0) made this compiler fold the abs source modifier into one half of
// the 64-bit select ("Illegal instruction detected: Operand has incorrect register class").
__device__ __forceinline__ float abs_diff_if(bool on, float a, float b) {
    const float d = on ? (a - b) : 0.0f;
    return __uint_as_float(__float_as_uint(d) & 0x7fffffffu);
}

// Make four just-requested words count as "arrived" from here on.  The compiler tracks outstanding vector-memory results
// per register and, where two paths join, assumes the younger request: a register requested early on one path and late
// on the other is then waited for as if it were the youngest request of all.  The late path calls this.
__device__ __forceinline__ void landed(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#endif
}

// dependency of a workgroup inside a one-launch sweep (see le_level_kernel): where the counters and the error word are, how
// patient a wait is.  Two carriers with one interface: LeDep holds the values (le_sweep_kernel); LeDepCold (below, behind
// LevelArgs) re-reads them from the kernel's argument block where a wait uses them -- they are touched by one thread of a
// workgroup, once or twice in its life, and six scalar registers that stay live across a whole tile function are six more values
// the compiler parks in vector-register lanes and moves back and forth (`cold`, dfq_le_resident.hip).
struct LeDep {
    const unsigned long long* counters_;   // 64-bit: tiles x sweeps outgrows 32 bits on runs with a large sweep cap
    unsigned long long* err_;
    int32_t sweep;
    int32_t naps_, limit_;
    __device__ __forceinline__ unsigned long long* counters() const { return (unsigned long long*)counters_; }
    __device__ __forceinline__ unsigned long long* err() const { return err_; }
    __device__ __forceinline__ int naps() const { return naps_; }
    __device__ __forceinline__ int limit() const { return limit_; }
};
constexpr int kDepStride = 16;          // one 64-bit counter per 128-byte line: hundreds of waiting workgroups poll them
constexpr int kLocalSlabs = 3;          // LeRelDev::local_r1: row tiles that merge their rows' statistics over the slabs of a row block inside the launch
// Returns false (for the whole workgroup) when the wait was abandoned: the caller then leaves WITHOUT storing anything,
// so a failed run never rescales weights with stale statistics -- it only stops short, and `err` makes every later
// waiter of the plan give up at once (they poll it every 256 spins) and the next query / run report DFQ_ERR_STATE.
// Memory order: the payload a consumer reads after this wait (row statistics) was written with device-scope atomics,
// which are performed at the coherence point, and is read with device-scope (sc1) loads, so the counter itself can be
// relaxed; a release on the add would be a `buffer_wbl2` of every dirty weight line of the XCD per tile.  The producer
// side orders "statistics performed" before "counter incremented" with s_waitcnt(0) + a workgroup barrier.
template <class Dep>
__device__ __forceinline__ bool counter_wait(const Dep& dep, int idx, int tiles, int* sh_flag) {
    if (idx < 0) return true;                        // uniform
    if (threadIdx.x == 0) {
        const unsigned long long target = (unsigned long long)tiles * (unsigned long long)(dep.sweep + 1);
        const unsigned long long* const counter = dep.counters() + (int64_t)idx * kDepStride;
        unsigned long long* const err = dep.err();
        const int naps = dep.naps(), spin_limit = dep.limit();
        long spins = 0;
        int ok = 1;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            for (int k = 0; k < naps; ++k) __builtin_amdgcn_s_sleep(8);   // default 2 naps: ~0.5 us between polls
            ++spins;
            if (spins > spin_limit ||
                ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull)) {
                atomicMax(err, 1ull);
                ok = 0;
                break;
            }
        }
        *sh_flag = ok;
    }
    __syncthreads();
    return *sh_flag != 0;
}
template <class Dep>
__device__ __forceinline__ bool dep_wait(const LeRelDev& R, const Dep& dep, int* sh_flag) {
    return counter_wait(dep, R.dep_idx, R.dep_tiles, sh_flag);
}
constexpr double kTileAbandoned = -1.0;   // a tile function's return value after a failed wait (sums of |dW| are >= 0)

// sum_k |a[k] - b[k]| of one register slot in float64, or 0 when the slot is a clamped duplicate (`on` false).
// One select per slot on the float64 sum; the magnitude is taken on the float32 difference by clearing the sign bit
// (see abs_diff_if for why not fabs()).
template <int VEC>
__device__ __forceinline__ double slot_abs_diff(bool on, const float (&a)[VEC], const float (&b)[VEC]) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const float d = a[k] - b[k];
        t += (double)__uint_as_float(__float_as_uint(d) & 0x7fffffffu);
    }
    return on ? t : 0.0;
}

template <int VEC>
__device__ __forceinline__ void vload(const gfloat* p, float (&x)[VEC], bool keep = false) {
    if (kAblate & 8) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) x[k] = 1.0f + (float)k;
        return;
    }
    if (VEC == 4) {
        const fvec4 t = ((kNonTemporal & 1) && !((kNonTemporal & 4) && keep)) ? DFQ_NT_LOAD((const gfvec4*)p) : *(const gfvec4*)p;
        x[0] = t[0]; x[1 % VEC] = t[1]; x[2 % VEC] = t[2]; x[3 % VEC] = t[3];
    } else {
        x[0] = *p;
    }
}
template <int VEC>
__device__ __forceinline__ void vstore(gfloat* p, const float (&x)[VEC]) {
    if ((kAblate & 1) && x[0] != 12345.678f) return;      // data-dependent so the producers stay alive
    if (VEC == 4) {
        fvec4 t;
        t[0] = x[0]; t[1] = x[1 % VEC]; t[2] = x[2 % VEC]; t[3] = x[3 % VEC];
        if (kNonTemporal & 2) DFQ_NT_STORE(t, (gfvec4*)p);
        else *(gfvec4*)p = t;
    } else {
        *p = x[0];
    }
}

// ---------------------------------------------------------------------------------------------
// Tiles.  Written once for VEC = 1 (any geometry) and VEC = 4 (rows that are a multiple of 4
// floats: every 1x1 / linear / dense conv layer of the benchmark networks): a thread then moves
// 16-byte vectors, which quarters the number of vector-memory instructions -- a CU retires roughly
// one wave-wide store instruction per ~36 cycles regardless of its width, so dword stores, not
// bytes, limit a scalar version.  Loads are unconditional (indices clamped into the tile), only
// stores are predicated; unused register slots are skipped with block-uniform branches.
// ---------------------------------------------------------------------------------------------

// row tile: W1[r0:r0+nr, p0:p0+np] *= s[row]   (+ column stats of the new values)
// PRE: the tile's elements are already in `v` (le_sweep_kernel requested them one tile ahead); `ready`: the dependency
// is known to be satisfied; `mid()` is called right after the tile's own statistics requests have been issued -- the
// sweep kernel requests the NEXT tile's data there, so that waiting for the statistics does not wait for that data
// (vector memory returns in order).
template <int VEC, bool PRE, class Mid, class Dep>
__device__ __forceinline__ double row_tile(const LeRelDev& R, const LeParams& p, int tile, int cur, const Dep& dep, bool ready,
                                           float (&v_in)[kSlotsVec4][VEC], Mid mid,
                                           float* sh_s, uint32_t* sh_slot, int* sh_g, float* sh_pinv, uint32_t* sh_rs, int* sh_flag, const LeTrace& tr) {
    constexpr int NV = (VEC == 4) ? kSlotsVec4 : kSlotsVec1;   // vectors per thread
    float v_own[NV][VEC];                                      // (an array of the caller used here when !PRE tripled the registers)
    float (&v)[NV][VEC] = *(PRE ? &v_in : &v_own);
    const int tid = threadIdx.x;
    const int rblk = small_div(tile, R.rt_slabs);
    const int slab = tile - rblk * R.rt_slabs;
    const int r0 = rblk * R.rt_rows;
    const int nr = min(R.rt_rows, R.o1 - r0);
    const int p0 = slab * R.rt_cols;
    const int np = min(R.rt_cols, R.row_len - p0);
    const int npv = np / VEC;                  // vector positions (np % VEC == 0 by plan)
    const int nxt = cur ^ 1;
    const int JL = small_div(kBlock, npv);
    const int jl_raw = small_div(tid, npv);
    const bool lane_on = jl_raw < JL;
    const int jl = lane_on ? jl_raw : 0;
    const int pos = p0 + (lane_on ? (tid - jl_raw * npv) * VEC : 0);
    const int n_own = lane_on ? small_div(nr - jl + JL - 1, JL) : 0;   // rows this thread owns (<= NV by plan)
    const int n_max = min(NV, small_div(nr + JL - 1, JL));             // register slots in use (block-uniform)
    gfloat* const w = (gfloat*)R.w1 + ((int64_t)r0 * R.row_len + pos);
    const bool fused = R.w1_interior != 0;     // the column rescale of the previous relation is applied here too
    // local_r1 (plan: VEC == 4, fused, the tile spans FULL rows): the tile takes the row ranges of t = fl(w / s_prev) itself -- a
    // pass over its registers, a reduction per row through LDS -- instead of reading the statistics a read-only pass of the
    // previous relation over this layer would have published (that pass is not launched: 4 B per element and a workgroup per
    // 8192 elements less), and publishes them for its relation's column tiles
    const bool local = VEC == 4 && R.local_r1 != 0;
    // deferred store (plan: never together with `fused` or `emit`): `phase` skipped sweeps precede this one since the
    // last store; their factors were written by earlier launches and are requested before everything else
    const bool defer = (R.defer & 1) != 0;
    const int phase = defer ? (dep.sweep & (p.defer - 1)) : 0;
    const bool keep = defer && phase != p.defer - 1;           // this sweep does not store either
    float hv[kHoldMax];
#pragma unroll
    for (int j = 0; j < kHoldMax; ++j) hv[j] = (j < phase && tid < nr) ? R.hold[(int64_t)(2 * j) * R.o1 + r0 + tid] : 1.0f;

    // ---- load order.  No dependency: the four statistics words of this thread's row go first -- they are back long
    //      before the data and the scale solve then overlaps the data's flight instead of queueing behind it (memory
    //      returns in order).  With a dependency the statistics do not exist yet: the data (which nobody writes before
    //      this workgroup does) is requested first and arrives while the workgroup waits for its producers. ----
    const bool waits = R.dep_idx >= 0 && !ready;
    // column-stat slot geometry of this tile
    const bool emit = R.out_cols != nullptr && !(kAblate & 4);
    int g0 = 0, i0 = 0, nci = 1, n_slots = 0;
    if (emit) {
        g0 = small_div(r0, R.pc_go);
        i0 = small_div(p0, R.khkw1);
        nci = small_div(p0 + np - 1, R.khkw1) - i0 + 1;
        n_slots = (small_div(r0 + nr - 1, R.pc_go) - g0 + 1) * nci;
    }
    uint32_t wa0 = 0u, wa1 = 0u, wb0 = 0u, wb1 = 0u;
    const guint* const st_a = (const guint*)R.r1 + (int64_t)cur * R.stat_stride + 2 * (r0 + min(tid, nr - 1));
    const guint* const st_b = (const guint*)R.r2 + (int64_t)cur * R.stat_stride + 2 * (r0 + min(tid, nr - 1));
    // first entry of this thread in the table of 1/s of the previous relation (see below): requested with the row's own
    // statistics; threads from the top of the workgroup so that the row-scale solves (threads 0..nr-1) run next to them
    const guint* const a_base = fused ? (const guint*)R.prev_r1 + (int64_t)cur * R.stat_stride : nullptr;
    const guint* const b_base = fused ? (const guint*)R.out_cols + (int64_t)cur * R.stat_stride : nullptr;
    const int sl0 = kBlock - 1 - tid;
    const bool has_sl0 = fused && sl0 < n_slots;
    int c_sl0 = 0;
    if (has_sl0) { const int gq = small_div(sl0, nci); c_sl0 = (g0 + gq) * R.pc_gi + i0 + (sl0 - gq * nci); }
    uint32_t pa0 = 0u, pa1 = 0u, pb0 = 0u, pb1 = 0u;
    if (!waits) {
        if (tid < nr) { if (!local) { wa0 = ld_stat(st_a); wa1 = ld_stat(st_a + 1); } wb0 = st_b[0]; wb1 = st_b[1]; }
        if (has_sl0) { pa0 = ld_stat(a_base + 2 * c_sl0); pa1 = ld_stat(a_base + 2 * c_sl0 + 1); pb0 = b_base[2 * c_sl0]; pb1 = b_base[2 * c_sl0 + 1]; }
    }
    if (!PRE) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            if (u < n_max) {
                const int r = min(jl + u * JL, nr - 1);
                vload<VEC>(w + r * R.row_len, v[u]);
            }
        }
    }
    mid();
    if (waits) {
        if (!dep_wait(R, dep, sh_flag)) return kTileAbandoned;
        if (tid < nr) { if (!local) { wa0 = ld_stat(st_a); wa1 = ld_stat(st_a + 1); } wb0 = st_b[0]; wb1 = st_b[1]; }
        if (has_sl0) { pa0 = ld_stat(a_base + 2 * c_sl0); pa1 = ld_stat(a_base + 2 * c_sl0 + 1); pb0 = b_base[2 * c_sl0]; pb1 = b_base[2 * c_sl0 + 1]; }
        if (PRE) { landed(wa0, wa1, wb0, wb1); landed(pa0, pa1, pb0, pb1); }
    }

    if (emit) for (int i = tid; i < 2 * n_slots; i += kBlock) sh_slot[i] = 0u;
    if (fused) {
        // 1/s of the previous relation for every (group, input channel) this tile spans: the same slot
        // geometry as the column statistics
        if (has_sl0) {
            float s, inv;
            le_solve(range_of(slot_min(pa0), slot_max(pa1), p.signed_range), range_of(slot_min(pb0), slot_max(pb1), p.signed_range),
                     p, s, inv);
            sh_pinv[sl0] = inv;
        }
        for (int sl = sl0 + kBlock; sl < n_slots; sl += kBlock) {      // tables of more than 256 entries
            const int gq = small_div(sl, nci);
            const int c = (g0 + gq) * R.pc_gi + i0 + (sl - gq * nci);
            const uint32_t a0 = ld_stat(a_base + 2 * c), a1 = ld_stat(a_base + 2 * c + 1), b0 = b_base[2 * c], b1 = b_base[2 * c + 1];
            float s, inv;
            le_solve(range_of(slot_min(a0), slot_max(a1), p.signed_range), range_of(slot_min(b0), slot_max(b1), p.signed_range),
                     p, s, inv);
            sh_pinv[sl] = inv;
        }
    }
    if (local) {
        // the rows' statistics of t: every thread folds its slots' four values into its row's pair (LDS atomics: the lanes of a
        // row may sit in different waves)
        if (tid < nr) { sh_rs[2 * tid] = 0u; sh_rs[2 * tid + 1] = 0u; sh_g[tid] = (small_div(r0 + tid, R.pc_go) - g0) * nci; }
        __syncthreads();                                   // sh_pinv, sh_g, the cleared pairs
        int ci0[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) ci0[k] = small_div(pos + k, R.khkw1) - i0;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            if (u >= n_max) continue;
            const int r = min(jl + u * JL, nr - 1);
            const int g = sh_g[r];
            float mn = INFINITY, mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float t = v[u][k] * sh_pinv[g + ci0[k]];     // dfq.py:73 of the previous relation (rounded)
                mn = vmin_raw(mn, t); mx = vmax_raw(mx, t);
            }
            if (u < n_own) { atomicMax(&sh_rs[2 * r], ~enc_ord(mn)); atomicMax(&sh_rs[2 * r + 1], enc_ord(mx)); }
        }
        __syncthreads();
        if (tid < nr) {
            wa0 = sh_rs[2 * tid]; wa1 = sh_rs[2 * tid + 1];
            // for this relation's column tiles (same launch, other XCDs: atomics, see ld_stat)
            uint32_t* dst = R.r1 + (int64_t)cur * R.stat_stride + 2 * (r0 + tid);
            atomicMax(dst + 0, wa0);
            atomicMax(dst + 1, wa1);
        }
        if (R.local_r1 == kLocalSlabs) {
            // The tile holds a SLAB of its rows; the rest of them sits in the row block's other slabs, whose workgroups are doing
            // exactly this.  So: the partial ranges above merge in the relation's statistics words, the tile arrives on the
            // relation's row-tile counter (statistics performed -> s_waitcnt 0 -> barrier -> one atomicAdd), waits until EVERY row
            // tile of the relation has arrived -- with its elements in registers -- and reads its rows' merged ranges back.  The
            // layer is then read ONCE per sweep: no read-only pass of the relation in front over it (8 instead of 12 B per element).
            // Row tiles of one relation and network are adjacent in the launch's table and dispatched in index order, so the ones
            // a tile waits for are resident or the next to become resident (le_level_kernel, "Dependencies inside a launch").
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (tid == 0) atomicAdd(dep.counters() + (int64_t)R.rcounter_idx * kDepStride, 1ull);
            if (!counter_wait(dep, R.rcounter_idx, R.n_row_tiles, sh_flag)) return kTileAbandoned;
            if (tid < nr) {
                const guint* src = (const guint*)R.r1 + (int64_t)cur * R.stat_stride + 2 * (r0 + tid);
                wa0 = ld_stat(src); wa1 = ld_stat(src + 1);
            }
        }
    }
    stamp(tr, 2);
    if (tid < nr) {
        const int c = r0 + tid;
        const bool own = slab == 0;                        // one tile per row owns the [O1] vectors
        float o_cum = 0.f, o_bnw = 0.f, o_bnb = 0.f, o_b1 = 0.f;
        if (own) {
            o_cum = R.s_cum[c];
            if (R.bnw) o_bnw = R.bnw[c];
            if (R.bnb) o_bnb = R.bnb[c];
            if (R.b1) o_b1 = R.b1[c];
        }
        float s, inv, mn1, mx1, mn2, mx2;
        scale_from_words(p, wa0, wa1, wb0, wb1, s, inv, mn1, mx1, mn2, mx2);
        sh_s[tid] = s;
        if (emit) sh_g[tid] = (small_div(c, R.pc_go) - g0) * nci;
        if (defer) {
#pragma unroll
            for (int j = 0; j < kHoldMax; ++j) if (j < phase) sh_pinv[j * kTileRowsMax + tid] = hv[j];
            if (own && keep) R.hold[(int64_t)(2 * phase) * R.o1 + c] = s;
        }
        if (own) {
            R.s_cum[c] = o_cum * s;                       // relation.py:20-24
            if (R.bnw) R.bnw[c] = o_bnw * s;              // dfq.py:64-65
            if (R.bnb) R.bnb[c] = o_bnb * s;              // dfq.py:67-68
            if (R.b1) R.b1[c] = o_b1 * s;                 // dfq.py:70-71
            if (!emit) {                                   // W1 is never column-scaled: forward its row stats
                guint* f = (guint*)R.r1 + (int64_t)nxt * R.stat_stride + 2 * c;
                f[0] = ~enc_ord(mn1 * s);
                f[1] = enc_ord(mx1 * s);
            }
        }
    }
    stamp(tr, 3);
    __syncthreads();
    stamp(tr, 4);

    double acc = 0.0;
    int ci[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) ci[k] = emit ? (small_div(pos + k, R.khkw1) - i0) : 0;
    float cmn[VEC], cmx[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { cmn[k] = INFINITY; cmx[k] = -INFINITY; }
    int cur_g = -1;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        if (u >= n_max) continue;
        const int r = min(jl + u * JL, nr - 1);
        const bool ok = u < n_own;
        const float s = sh_s[r];
        const int g = emit ? sh_g[r] : 0;                  // slot row of this row's group
        // 1/s of the previous relation for this row's columns: all reads issued before the first use (a
        // conditional read per element made the compiler wait for every LDS round trip separately); 1.0f when
        // the layer is not interior -- multiplying by one is exact
        float pin[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) pin[k] = 1.0f;
        if (fused) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) pin[k] = sh_pinv[g + ci[k]];
        }
        float nv[VEC];
        if (!defer) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float t = v[u][k] * pin[k];               // dfq.py:73 of the previous relation (rounded), then
                nv[k] = t * s;                                  // dfq.py:62 of this one
            }
            if (ok) vstore<VEC>(w + r * R.row_len, nv);
            if (!(kAblate & 2)) acc += slot_abs_diff<VEC>(ok, nv, v[u]);
        } else {
            // the stored value is `phase` sweeps old: take it through the remembered factors (one rounding each, as the
            // skipped stores would have), then this sweep's
            float t[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) t[k] = v[u][k];
#pragma unroll
            for (int j = 0; j < kHoldMax; ++j) {
                if (j < phase) {
                    const float h = sh_pinv[j * kTileRowsMax + r];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) t[k] = t[k] * h;
                }
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) nv[k] = t[k] * s;
            if (ok && !keep) vstore<VEC>(w + r * R.row_len, nv);
            if (!(kAblate & 2)) acc += slot_abs_diff<VEC>(ok, nv, t);
        }
        if (emit && ok) {
            if (g != cur_g) {
                if (cur_g >= 0) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        atomicMax(&sh_slot[2 * (cur_g + ci[k]) + 0], ~enc_ord(cmn[k]));
                        atomicMax(&sh_slot[2 * (cur_g + ci[k]) + 1], enc_ord(cmx[k]));
                        cmn[k] = INFINITY; cmx[k] = -INFINITY;
                    }
                }
                cur_g = g;
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) { cmn[k] = vmin_raw(cmn[k], nv[k]); cmx[k] = vmax_raw(cmx[k], nv[k]); }
        }
    }
    if (emit && cur_g >= 0) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            atomicMax(&sh_slot[2 * (cur_g + ci[k]) + 0], ~enc_ord(cmn[k]));
            atomicMax(&sh_slot[2 * (cur_g + ci[k]) + 1], enc_ord(cmx[k]));
        }
    }
    stamp(tr, 5);
    if (emit) {
        __syncthreads();
        for (int sl = tid; sl < n_slots; sl += kBlock) {
            const uint32_t a = sh_slot[2 * sl + 0], b = sh_slot[2 * sl + 1];
            if (b != 0u) {
                const int gq = small_div(sl, nci);
                const int g = g0 + gq;
                const int i = i0 + sl - gq * nci;
                guint* dst = (guint*)R.out_cols + (int64_t)nxt * R.stat_stride + 2 * ((int64_t)g * R.pc_gi + i);
                atomicMax((unsigned*)dst + 0, a);
                atomicMax((unsigned*)dst + 1, b);
            }
        }
    }
    return acc;
}

// col tile: W2[r0:r0+nr, p0:p0+np] *= 1/s[input channel]   (+ row stats of the new values)
// G = pow2 >= np/VEC lanes share a row; 256/G rows are in flight per register slot.
template <int VEC, bool PRE, class Mid, class Dep>
__device__ __forceinline__ double col_tile(const LeRelDev& R, const LeParams& p, int tile, int cur, const Dep& dep, bool ready,
                                           float (&v_in)[kSlotsVec4][VEC], Mid mid,
                                           float* sh_inv, uint32_t* sh_row, int* sh_tab, float* sh_hold, int* sh_flag, const LeTrace& tr) {
    constexpr int NV = (VEC == 4) ? kSlotsVec4 : kSlotsVec1;
    float v_own[NV][VEC];
    float (&v)[NV][VEC] = *(PRE ? &v_in : &v_own);
    const int tid = threadIdx.x;
    const int rblk = small_div(tile, R.ct_slabs);
    const int slab = tile - rblk * R.ct_slabs;
    const int r0 = rblk * R.ct_rows;
    const int nr = min(R.ct_rows, R.o2 - r0);
    const int row_len2 = R.i2g * R.khkw;
    const int p0 = slab * R.ct_cols;
    const int np = min(R.ct_cols, row_len2 - p0);
    const int npv = np / VEC;                              // <= kColTileLanes by plan
    int G = 1, lgG = 0;
    while (G < npv) { G <<= 1; ++lgG; }
    const int n_rowslots = kBlock >> lgG;                  // rows in flight
    const int grp = tid >> lgG;
    const int ln = tid - grp * G;
    const bool lane_on = ln < npv;
    const int pos = p0 + min(ln, npv - 1) * VEC;
    const int n_max = min(NV, (nr + n_rowslots - 1) >> (8 - lgG));     // register slots in use (block-uniform)
    const int nxt = cur ^ 1;
    const bool stat_only = R.w2_interior != 0;              // the write happens in the next relation's row pass
    const bool emit = R.out_rows != nullptr && !(kAblate & 4);
    gfloat* const w = (gfloat*)R.w2 + ((int64_t)r0 * row_len2 + pos);

    // 1/s table of the tile: (groups spanned by the rows) x (input channels spanned by the columns)
    const int i0 = small_div(p0, R.khkw);
    const int nci = small_div(p0 + np - 1, R.khkw) - i0 + 1;
    const int g_lo = small_div(r0, R.go);
    const int g_n = small_div(r0 + nr - 1, R.go) - g_lo + 1;
    // ---- load order as in row_tile: statistics of this thread's table entry first, unless they are still being
    //      produced -- then the data first, the wait, the statistics ----
    const bool waits = R.dep_idx >= 0 && !ready;
    uint32_t wa0 = 0u, wa1 = 0u, wb0 = 0u, wb1 = 0u;
    const bool has_entry = tid < g_n * nci;
    const int e_gq = small_div(min(tid, g_n * nci - 1), nci);
    const int e_c = (g_lo + e_gq) * R.gi + i0 + (min(tid, g_n * nci - 1) - e_gq * nci);
    const guint* const st_a = (const guint*)R.r1 + (int64_t)cur * R.stat_stride + 2 * e_c;
    const guint* const st_b = (const guint*)R.r2 + (int64_t)cur * R.stat_stride + 2 * e_c;
    // deferred store (plan: never with `stat_only` or `emit`), see row_tile: the remembered 1/s of the skipped sweeps, one
    // table per skipped sweep laid out like sh_inv (the first in sh_hold, the others in the unused row-statistics words)
    const bool defer = (R.defer & 2) != 0;
    const int phase = defer ? (dep.sweep & (p.defer - 1)) : 0;
    const bool keep = defer && phase != p.defer - 1;
    if ((kAblate & 16) && defer) {
        // tuning builds only: what a deferred chain-end column tile would cost with its factors handed to it (no wait, no
        // statistics words, no solve, no table, no barrier) -- results WRONG by construction
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            if (u < n_max) vload<VEC>(w + min(grp + u * n_rowslots, nr - 1) * row_len2, v[u], false);
        }
        const float f = 1.0f + 1e-7f * (float)(dep.sweep & 3);
        double a = 0.0;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            if (u >= n_max) continue;
            const int r_raw = grp + u * n_rowslots;
            const bool ok = lane_on && r_raw < nr;
            float nv[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) nv[k] = v[u][k] * f;
            if (ok && !keep) vstore<VEC>(w + min(r_raw, nr - 1) * row_len2, nv);
            a += slot_abs_diff<VEC>(ok, nv, v[u]);
        }
        return a;
    }
    auto hold_tab = [&](int j) { return j == 0 ? sh_hold : (float*)sh_row + (j - 1) * kSlotMax; };
    float hv[kHoldMax];
#pragma unroll
    for (int j = 0; j < kHoldMax; ++j) hv[j] = (j < phase && has_entry) ? R.hold[(int64_t)(2 * j + 1) * R.o1 + e_c] : 1.0f;
    if (!waits && has_entry) { wa0 = ld_stat(st_a); wa1 = ld_stat(st_a + 1); wb0 = st_b[0]; wb1 = st_b[1]; }
    if (!PRE) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            if (u < n_max) {
                const int r = min(grp + u * n_rowslots, nr - 1);
                vload<VEC>(w + r * row_len2, v[u], stat_only);
            }
        }
    }
    mid();
    if (waits) {
        if (!dep_wait(R, dep, sh_flag)) return kTileAbandoned;
        if (has_entry) { wa0 = ld_stat(st_a); wa1 = ld_stat(st_a + 1); wb0 = st_b[0]; wb1 = st_b[1]; }
        if (PRE) landed(wa0, wa1, wb0, wb1);
    }
    stamp(tr, 2);
    for (int idx = tid; idx < g_n * nci; idx += kBlock) {      // one entry per thread for every plan-made tile
        const int gq = small_div(idx, nci);
        const int g = g_lo + gq;
        const int c = g * R.gi + i0 + (idx - gq * nci);
        float s, inv, mn1, mx1, mn2, mx2;
        if (idx == tid) scale_from_words(p, wa0, wa1, wb0, wb1, s, inv, mn1, mx1, mn2, mx2);
        else channel_scale(R, p, cur, c, s, inv, mn1, mx1, mn2, mx2);
        sh_inv[idx] = inv;
        const bool first_row = g * R.go >= r0 && g * R.go < r0 + nr;
        // W2 is never row-scaled: forward its column stats (the tile holding the group's first row does it)
        if (!emit && first_row) {
            guint* f = (guint*)R.r2 + (int64_t)nxt * R.stat_stride + 2 * c;
            f[0] = ~enc_ord(mn2 * inv);
            f[1] = enc_ord(mx2 * inv);
        }
        if (defer) {
#pragma unroll
            for (int j = 0; j < kHoldMax; ++j)
                if (j < phase) hold_tab(j)[idx] = (idx == tid) ? hv[j] : R.hold[(int64_t)(2 * j + 1) * R.o1 + c];
            if (keep && first_row) R.hold[(int64_t)(2 * phase + 1) * R.o1 + c] = inv;
        }
    }
    if (tid < nr) sh_tab[tid] = (small_div(r0 + tid, R.go) - g_lo) * nci;
    stamp(tr, 3);
    __syncthreads();
    stamp(tr, 4);

    double acc = 0.0;
    int ci[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) ci[k] = small_div(pos + k, R.khkw) - i0;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        if (u >= n_max) continue;
        const int r_raw = grp + u * n_rowslots;
        const int r = min(r_raw, nr - 1);
        const bool ok = lane_on && r_raw < nr;
        const int t0 = sh_tab[r];
        float nv[VEC];
        if (!defer) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) nv[k] = v[u][k] * sh_inv[t0 + ci[k]];      // dfq.py:73
            if (!stat_only) {
                if (ok) vstore<VEC>(w + r * row_len2, nv);
                if (!(kAblate & 2)) acc += slot_abs_diff<VEC>(ok, nv, v[u]);
            }
        } else {
            float t[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) t[k] = v[u][k];
#pragma unroll
            for (int j = 0; j < kHoldMax; ++j) {
                if (j < phase) {
                    const float* tab = hold_tab(j) + t0;
#pragma unroll
                    for (int k = 0; k < VEC; ++k) t[k] = t[k] * tab[ci[k]];
                }
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k) nv[k] = t[k] * sh_inv[t0 + ci[k]];
            if (ok && !keep) vstore<VEC>(w + r * row_len2, nv);
            if (!(kAblate & 2)) acc += slot_abs_diff<VEC>(ok, nv, t);
        }
        if (emit) {                                        // block-uniform: every lane reaches the shuffles
            float rmn = INFINITY, rmx = -INFINITY;
            if (ok) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) { rmn = vmin_raw(rmn, nv[k]); rmx = vmax_raw(rmx, nv[k]); }
            }
            // register-file butterfly steps behind uniform guards (xor_lane_minmax, dfq_common.hpp): a run-time mask makes
            // every step a ds_bpermute through the LDS crossbar that all waves of the CU share
            if (G > 1) xor_lane_minmax<1>(rmn, rmx);
            if (G > 2) xor_lane_minmax<2>(rmn, rmx);
            if (G > 4) xor_lane_minmax<4>(rmn, rmx);
            if (G > 8) xor_lane_minmax<8>(rmn, rmx);
            if (G > 16) xor_lane_minmax<16>(rmn, rmx);
            if (G > 32) xor_lane_minmax<32>(rmn, rmx);
            if (ln == 0 && r_raw < nr) {                   // one writer per row of the tile
                sh_row[2 * r + 0] = ~enc_ord(rmn);
                sh_row[2 * r + 1] = enc_ord(rmx);
            }
        }
    }
    stamp(tr, 5);
    if (emit) {
        __syncthreads();
        // row stats go into the SAME sweep's parity (consumed by a later level of this sweep)
        // (always atomics, also for complete rows: a consumer on another XCD reads them in this same launch)
        guint* dst = (guint*)R.out_rows + (int64_t)cur * R.stat_stride + 2 * r0;
        for (int i = tid; i < 2 * nr; i += kBlock) atomicMax((unsigned*)dst + i, sh_row[i]);
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// short-row tile: one THREAD per row (depthwise k x k kernels: 9 or 25 floats per row).  The row's
// scale depends on a single paired channel, its statistics stay in the thread's registers: no LDS,
// no barrier, no cross-lane traffic.  side 0: W1 rows (scale s, row == paired channel, emits the
// per-row stats the previous relation needs as its column stats); side 1: W2 rows of a depthwise
// second layer (scale 1/s of its single input channel, emits row stats for the next relation).
// ---------------------------------------------------------------------------------------------
template <int side>
__device__ __forceinline__ double short_tile(const LeRelDev& R, const LeParams& p, int tile, int cur) {
    const int tid = threadIdx.x;
    const int nxt = cur ^ 1;
    const int n_rows = side == 0 ? R.o1 : R.o2;
    const int len = side == 0 ? R.row_len : R.khkw;          // side 1: I2/g == 1, row = khkw floats
    const int o = min(tile * kBlock + tid, n_rows - 1);
    const bool ok = tile * kBlock + tid < n_rows;
    gfloat* const w = (gfloat*)(side == 0 ? R.w1 : R.w2) + (int64_t)o * len;
    const bool fused = side == 0 && R.w1_interior != 0;       // also apply the previous relation's 1/s (per row here)
    const bool stat_only = side == 1 && R.w2_interior != 0;   // statistics of the rescaled row only, no store
    const int c = side == 0 ? o : small_div(o, R.go) * R.gi;  // paired channel that scales this row

    float o_cum = 0.f, o_bnw = 0.f, o_bnb = 0.f, o_b1 = 0.f;
    if (side == 0) {
        o_cum = R.s_cum[c];
        if (R.bnw) o_bnw = R.bnw[c];
        if (R.bnb) o_bnb = R.bnb[c];
        if (R.b1) o_b1 = R.b1[c];
    }
    float pinv = 1.0f;
    if (fused) {              // row o of W1 is channel o * pc_gi of the previous relation (pc_go == 1 by plan)
        const int cp = o * R.pc_gi;
        const guint* a = (const guint*)R.prev_r1 + (int64_t)cur * R.stat_stride + 2 * cp;
        const guint* b = (const guint*)R.out_cols + (int64_t)cur * R.stat_stride + 2 * cp;
        const uint32_t a0 = ld_stat(a), a1 = ld_stat(a + 1), b0 = b[0], b1 = b[1];
        float ps;
        le_solve(range_of(slot_min(a0), slot_max(a1), p.signed_range), range_of(slot_min(b0), slot_max(b1), p.signed_range),
                 p, ps, pinv);
    }
    float s, inv, mn1, mx1, mn2, mx2;
    const bool local = side == 0 && R.local_r1 != 0;          // (plan: fused, len <= kShortChunk)
    float x0[kShortChunk];
    if (local) {
        // the row's own statistics: range of t = fl(w * 1/s_prev) over its taps (what the previous relation's read-only pass
        // over this layer computed with the same multiplication), then the scale from them and the stored column statistics
#pragma unroll
        for (int k = 0; k < kShortChunk; ++k) x0[k] = w[min(k, len - 1)];
        const guint* b = (const guint*)R.r2 + (int64_t)cur * R.stat_stride + 2 * c;
        const uint32_t b0 = b[0], b1 = b[1];
        mn1 = INFINITY; mx1 = -INFINITY;
#pragma unroll
        for (int k = 0; k < kShortChunk; ++k) {
            const float t = x0[k] * pinv;                          // clamped duplicates of the last tap are harmless
            mn1 = vmin_raw(mn1, t); mx1 = vmax_raw(mx1, t);
        }
        mn2 = slot_min(b0); mx2 = slot_max(b1);
        le_solve(range_of(mn1, mx1, p.signed_range), range_of(mn2, mx2, p.signed_range), p, s, inv);
        if (ok) {            // for this relation's column tiles (same launch, other XCDs: atomics, see ld_stat)
            uint32_t* dst = R.r1 + (int64_t)cur * R.stat_stride + 2 * c;
            atomicMax(dst + 0, ~enc_ord(mn1));
            atomicMax(dst + 1, enc_ord(mx1));
        }
    } else {
        channel_scale(R, p, cur, c, s, inv, mn1, mx1, mn2, mx2);
    }
    const float f = side == 0 ? s : inv;
    if (side == 0 && ok) {
        R.s_cum[c] = o_cum * s;                       // relation.py:20-24
        if (R.bnw) R.bnw[c] = o_bnw * s;              // dfq.py:64-65
        if (R.bnb) R.bnb[c] = o_bnb * s;              // dfq.py:67-68
        if (R.b1) R.b1[c] = o_b1 * s;                 // dfq.py:70-71
    }
    double acc = 0.0;
    float rmn = INFINITY, rmx = -INFINITY;
    // the row is walked in chunks of kShortChunk floats with every load of a chunk in flight at once (a
    // plain one-float-at-a-time loop pays one global round trip per tap: ~9 in a row for a 3x3 kernel)
    for (int k0 = 0; k0 < len; k0 += kShortChunk) {
        float x[kShortChunk];
#pragma unroll
        for (int k = 0; k < kShortChunk; ++k) x[k] = local ? x0[k] : w[min(k0 + k, len - 1)];
#pragma unroll
        for (int k = 0; k < kShortChunk; ++k) {
            const bool in = k0 + k < len;                          // uniform
            float t = x[k];
            if (fused) t = t * pinv;                               // dfq.py:73 of the previous relation, rounded
            const float nv = t * f;                                // dfq.py:62 / dfq.py:73
            if (!stat_only) {
                if (ok && in) w[k0 + k] = nv;
                acc += (double)abs_diff_if(ok && in, nv, x[k]);
            }
            rmn = vmin_raw(rmn, nv);                               // clamped duplicates of the last tap are harmless
            rmx = vmax_raw(rmx, nv);
        }
    }
    if (ok) {
        if (side == 0) {
            if (R.out_cols) {      // row o of W1 is channel o * pc_gi of the relation that has W1 as second layer
                uint32_t* dst = R.out_cols + (int64_t)nxt * R.stat_stride + 2 * ((int64_t)o * R.pc_gi);
                dst[0] = ~enc_ord(rmn);
                dst[1] = enc_ord(rmx);
            } else {               // chain start: forward the row stats
                uint32_t* fwd = R.r1 + (int64_t)nxt * R.stat_stride + 2 * c;
                fwd[0] = ~enc_ord(mn1 * s);
                fwd[1] = enc_ord(mx1 * s);
            }
        } else {
            if (R.out_rows) {      // consumed in this same launch: atomics (see ld_stat)
                uint32_t* dst = R.out_rows + (int64_t)cur * R.stat_stride + 2 * o;
                atomicMax(dst + 0, ~enc_ord(rmn));
                atomicMax(dst + 1, enc_ord(rmx));
            } else if (o == small_div(o, R.go) * R.go) {   // chain end: first row of the group forwards
                uint32_t* fwd = R.r2 + (int64_t)nxt * R.stat_stride + 2 * c;
                fwd[0] = ~enc_ord(mn2 * inv);
                fwd[1] = enc_ord(mx2 * inv);
            }
        }
    }
    return acc;
}

// lanes 0..n_words-1 fetch one word each (the others repeat the last one: no predicate, so the request is unconditional
// for the compiler's bookkeeping of outstanding loads, see le_sweep_kernel)
__device__ __forceinline__ uint32_t fetch_words(const void* base, int n_words, int lane) {
    return ((const guint*)base)[min(lane, n_words - 1)];
}
#include "dfq_le_cf.hpp"

constexpr size_t kLevelSmem = sizeof(float) * (2 * kSlotMax + 2 * kSlotMax + kTileRowsMax + 2 * kTileRowsMax + 1);   // le_level_kernel's shared memory
constexpr int kDescWords = (int)(sizeof(LeRelDev) / 4);
static_assert(sizeof(LeRelDev) % 4 == 0 && kDescWords + 1 <= kWave, "descriptor must fit one wave-wide load");

// One launch = any contiguous slice of the sweep's workgroup table: the whole sweep (default) or one dependency
// level (DFQ_LE_MERGED=0).  `sweep` = sweeps since the last restart (parity = sweep & 1).
//
// Dependencies inside a launch.  The tiles of a relation whose first layer is interior need the row statistics
// that the column tiles of ONE earlier relation (its predecessor in the chain) publish, and must not write that
// layer before those tiles have read it.  The table is in level order, workgroups are dispatched in index
// order and only ever wait for lower indices, so a spinning workgroup never blocks its producers.  Producer:
// statistics atomics -> s_waitcnt 0 -> barrier -> one atomicAdd on the relation's counter.  Consumer: one
// thread spins (device-scope loads, s_sleep) until the counter reaches tiles x (sweep + 1), then a barrier.
// The spin is bounded: a consumer that gives up raises `err[0]` and carries on, so a logic error shows up as
// an error code from the query call instead of a hung GPU.
#ifndef DFQ_LE_MIN_WAVES
#define DFQ_LE_MIN_WAVES 1
#endif
// kTrace: the tuning instantiation that honours `tr_arg` (dfq_le_trace*); the production one sees a constant null trace, so the
// eight stamp sites and the four scalar registers of the argument vanish.
// -DDFQ_LE_WEAVE=1 (opt-in build): the lean tiles of the free-running layers ride in the sweep's own launch at a group's first
// sweep, listed evenly among the general tiles, instead of in a launch of their own.  Built and measured in round 6 (batch of 32
// MobileNetV2, two alternating rounds, tools/gpu_r06_weave.sh): 1.701 / 1.705e10 weights/s woven against 1.688 / 1.690e10 with
// le_lean_kernel's own launch -- the sweep's wall time is the same (141.5-143.6 against 139.9-140.8 us): the lean tiles' arithmetic
// does not hide behind the general tiles' memory traffic, it queues for the same issue slots -- and the woven body takes
// le_level_kernel from 78 to 86 vector registers (six -> five workgroups per CU) and from 94 to 127 scalar spills, which costs the
// launches WITHOUT lean tiles 2 % (1.66e10 with the lean launch on the 86-register build).  Not the default.
#ifndef DFQ_LE_WEAVE
#define DFQ_LE_WEAVE 0
#endif
constexpr bool kWeave = DFQ_LE_WEAVE != 0;
constexpr int kWeaveGroup = 4;          // the group depth whose lean tiles ride in the sweep's launch (the default depth)
static_assert((2 * kWeaveGroup - 1) * kCfTab <= 2 * kSlotMax, "the woven lean tile's factor tables live in the column-statistics slots");
// The kernel's ONE argument: what a tile needs all the time is read through `a`, what it needs once or twice in its life (the
// wait's counters, error word and patience) through cold(a) at the point of use.
struct LevelArgs {
    const LeRelDev* table;
    const LeBlockRef* blocks;
    LeParams p;
    int32_t sweep;
    int32_t pad;
    const LeState* state;
    double* partials;
    unsigned long long* dep_counters;
    unsigned long long* err;
    LeTrace tr;
    const LeLeanRef* lean;
    int64_t part_stride;
};
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ const LevelArgs DFQ_CONSTANT_AS& cold(const LevelArgs&) {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();      // LevelArgs is the kernel's first (and only) argument
    asm volatile("" : "+s"(p));
    return *(const LevelArgs DFQ_CONSTANT_AS*)p;
}
#else
__device__ __forceinline__ const LevelArgs& cold(const LevelArgs& a) { return a; }
#endif
struct LeDepCold {
    const LevelArgs& a;
    int32_t sweep;
    __device__ __forceinline__ unsigned long long* counters() const { return cold(a).dep_counters; }
    __device__ __forceinline__ unsigned long long* err() const { return cold(a).err; }
    __device__ __forceinline__ int naps() const { return cold(a).p.poll_naps; }
    __device__ __forceinline__ int limit() const { return cold(a).p.spin_limit; }
};
template <bool kTrace>
__global__ __launch_bounds__(kBlock, DFQ_LE_MIN_WAVES) void le_level_kernel(LevelArgs a) {
    const LeRelDev* const table = a.table;
    const LeBlockRef* const blocks = a.blocks;
    const LeParams& p = a.p;
    const int sweep = a.sweep;
    const LeState* const state = a.state;
    double* const partials = a.partials;
    LeTrace tr = kTrace ? a.tr : LeTrace{nullptr, 0, 0};
    tr.flat = (int)blockIdx.x;
    stamp(tr, 0);
    // (dynamic shared memory, carved by hand: the slab tiles keep their tables across an in-launch wait, and the CPU emulation
    // gives every workgroup of a concurrent launch a buffer of its own only for dynamic shared memory)
    DFQ_DYN_SMEM(smem);
    float* const sh_f = (float*)smem;                           // [kSlotMax] row tile: scales; col tile: 1/s table
    uint32_t* const sh_u = (uint32_t*)(sh_f + kSlotMax);        // [2 kSlotMax] row tile: column-stat slots; col tile: row stats
    int* const sh_g = (int*)(sh_u + 2 * kSlotMax);              // [kTileRowsMax] per-row table offsets
    float* const sh_p = (float*)(sh_g + kTileRowsMax);          // [kSlotMax] row tile of an interior layer: 1/s of the previous relation
    uint32_t* const sh_rs = (uint32_t*)(sh_p + kSlotMax);       // [2 kTileRowsMax] row tile that takes its rows' statistics itself (local_r1)
    int& sh_flag = *(int*)(sh_rs + 2 * kTileRowsMax);           // outcome of the dependency wait
    const int lane = threadIdx.x % kWave;
    // one 16-byte load of the workgroup's entry, then ONE wave-wide load that fetches the descriptor
    // (lanes 0..kDescWords-1) and the loop state of the network (lane kDescWords) together; v_readlane
    // broadcasts the words (letting the compiler index a descriptor struct makes it fetch the fields piecemeal
    // at first use: one dependent round trip per group of fields)
    typedef int ivec4 __attribute__((vector_size(16)));
    const ivec4 ref = *(const DFQ_GLOBAL_AS ivec4*)(blocks + blockIdx.x);
    const int rel = __builtin_amdgcn_readfirstlane(ref[0]);
    const int net = __builtin_amdgcn_readfirstlane(ref[2]);
    const int tile = __builtin_amdgcn_readfirstlane(ref[1]);
    if (kWeave && rel < 0) {
        // a lean tile of a free-running layer (dfq_le_cf.hpp), woven into the launch of its group's first sweep: multiplies and
        // |dW| chains for kWeaveGroup sweeps per element loaded -- arithmetic that overlaps the general tiles' memory traffic
        lean_tile_run<kWeaveGroup>(a.lean, tile, LeanArgs{sweep, 0, a.part_stride}, state, partials, (float*)sh_u, sh_g);
        return;
    }
    uint32_t word = 0u;
    {
        const guint* src = (lane < kDescWords) ? (const guint*)(table + rel) + lane : (const guint*)&state[net].done;
        if (lane <= kDescWords) word = *src;
    }
    // Only the fields the tile's side reads are broadcast into scalar registers (the others stay zero constants): the whole
    // descriptor is 57 of the ~100 scalar registers a wave has, and every field beyond the budget costs a spill move per use.
    union { LeRelDev R; uint32_t u[kDescWords]; } desc;
#pragma unroll
    for (int i = 0; i < kDescWords; ++i) desc.u[i] = 0u;
#define DFQ_TAKE(f)                                                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < (int)(sizeof(desc.R.f) / 4); ++i_)                                               \
        desc.u[offsetof(LeRelDev, f) / 4 + i_] = __builtin_amdgcn_readlane(word, (int)(offsetof(LeRelDev, f) / 4) + i_)
    DFQ_TAKE(n_row_tiles); DFQ_TAKE(n_col_tiles); DFQ_TAKE(partial_base); DFQ_TAKE(counter_idx);
    DFQ_TAKE(dep_idx); DFQ_TAKE(dep_tiles); DFQ_TAKE(r1); DFQ_TAKE(r2); DFQ_TAKE(stat_stride); DFQ_TAKE(hold); DFQ_TAKE(defer); DFQ_TAKE(o1);
    const uint32_t done = __builtin_amdgcn_readlane(word, kDescWords);
    const LeRelDev& R = desc.R;
    const int cur = sweep & 1;
    if (done || tile >= R.n_row_tiles + R.n_col_tiles) return;   // uniform
    stamp(tr, 1);

    const LeDepCold dep{a, sweep};
    auto nothing = [] {};

    double acc;
    const bool col_side = tile >= R.n_row_tiles;
    if (!col_side) {
        DFQ_TAKE(w1); DFQ_TAKE(b1); DFQ_TAKE(bnw); DFQ_TAKE(bnb); DFQ_TAKE(s_cum); DFQ_TAKE(prev_r1); DFQ_TAKE(out_cols);
        DFQ_TAKE(row_len); DFQ_TAKE(khkw1); DFQ_TAKE(pc_go); DFQ_TAKE(pc_gi);
        DFQ_TAKE(rt_rows); DFQ_TAKE(rt_cols); DFQ_TAKE(rt_slabs); DFQ_TAKE(rt_vec); DFQ_TAKE(w1_interior); DFQ_TAKE(local_r1); DFQ_TAKE(rcounter_idx);
    } else {
        DFQ_TAKE(w2); DFQ_TAKE(out_rows); DFQ_TAKE(o2); DFQ_TAKE(gi); DFQ_TAKE(go); DFQ_TAKE(i2g); DFQ_TAKE(khkw);
        DFQ_TAKE(ct_rows); DFQ_TAKE(ct_cols); DFQ_TAKE(ct_slabs); DFQ_TAKE(ct_vec); DFQ_TAKE(w2_interior);
        DFQ_TAKE(cdep_idx); DFQ_TAKE(cdep_tiles);
        if ((int)desc.R.cdep_idx >= 0) { desc.R.dep_idx = desc.R.cdep_idx; desc.R.dep_tiles = desc.R.cdep_tiles; }   // (see LeRelDev::local_r1)
    }
#undef DFQ_TAKE
    if (!col_side) {
        if (R.rt_vec == 0) acc = dep_wait(R, dep, &sh_flag) ? short_tile<0>(R, p, tile, cur) : kTileAbandoned;
        else if (R.rt_vec == 4) { float v[kSlotsVec4][4]; acc = row_tile<4, false>(R, p, tile, cur, dep, false, v, nothing, sh_f, sh_u, sh_g, sh_p, sh_rs, &sh_flag, tr); }
        else { float v[kSlotsVec4][1]; acc = row_tile<1, false>(R, p, tile, cur, dep, false, v, nothing, sh_f, sh_u, sh_g, sh_p, sh_rs, &sh_flag, tr); }
    } else {
        if (R.ct_vec == 0) acc = dep_wait(R, dep, &sh_flag) ? short_tile<1>(R, p, tile - R.n_row_tiles, cur) : kTileAbandoned;
        else if (R.ct_vec == 4) { float v[kSlotsVec4][4]; acc = col_tile<4, false>(R, p, tile - R.n_row_tiles, cur, dep, false, v, nothing, sh_f, sh_u, sh_g, sh_p, &sh_flag, tr); }
        else { float v[kSlotsVec4][1]; acc = col_tile<1, false>(R, p, tile - R.n_row_tiles, cur, dep, false, v, nothing, sh_f, sh_u, sh_g, sh_p, &sh_flag, tr); }
    }
    stamp(tr, 6);
    if (acc < 0.0) return;          // abandoned wait (uniform): nothing was stored, the counter is not bumped
    if (col_side && R.counter_idx >= 0) {
        // every statistics atomic of this workgroup has been performed before the counter moves
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(dep.counters() + (int64_t)R.counter_idx * kDepStride, 1ull);
    }
    if (!col_side && R.rcounter_idx >= 0 && R.local_r1 != kLocalSlabs) {   // row tiles that published their rows' statistics (local_r1; the slab kind arrived long ago)
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(dep.counters() + (int64_t)R.rcounter_idx * kDepStride, 1ull);
    }
    // one partial per wave (fixed butterfly order -> deterministic), no workgroup barrier
    const double t = wave_sum(acc);
    if (lane == 0) partials[(int64_t)(R.partial_base + tile) * (kBlock / kWave) + threadIdx.x / kWave] = t;
    stamp(tr, 7);
}

// ---------------------------------------------------------------------------------------------
// le_sweep_kernel: the same sweep as ONE launch of persistent workgroups.  Workgroup g walks the tiles g, g + G,
// g + 2G, ... of the sweep's table (G = gridDim.x <= the number of workgroups the chip holds at once) and keeps the
// NEXT tile's data in flight while it works on the current one: everything needed to request a tile's elements is a
// 64-byte LeTileRef, fetched two tiles ahead; the relation descriptor and the dependency counter of a tile are fetched
// one tile ahead together with its elements.  A workgroup of le_level_kernel spends most of its 10-15 us on the chain
// table entry -> descriptor -> statistics -> solve -> elements -> stores, with its loads in flight for a fraction of that
// time; here the chain of tile i+1 overlaps the arithmetic and the stores of tile i.
//
// Order of the requests inside an iteration (vector memory returns in order): the statistics of tile i first, then the
// elements of tile i+1 -- all eight slots unconditionally, so that the compiler can count them and the wait for the
// statistics does not become a wait for everything.
//
// Progress: a tile only ever waits for tiles with a lower index; the workgroups hold their tiles in increasing order and
// are all resident (the host sizes the grid with the occupancy query), so the lowest unfinished tile is always some
// workgroup's current tile and can finish.  The wait is bounded as in le_level_kernel.
// ---------------------------------------------------------------------------------------------
struct alignas(64) LeTileRef {
    float* w;            // element (first row, first position) of the tile
    int32_t stride;      // floats between two rows of the layer
    int32_t nr;          // rows
    int32_t npv;         // 16-byte vectors per row of the tile; 0: the tile fetches its own elements (scalar / thread-per-row)
    int32_t lanes;       // threads along a row: npv for row tiles, the next power of two for column tiles
    int32_t rel;         // relation (index into the level-sorted table)
    int32_t tile;        // tile of the relation: row tiles first, then column tiles
    int32_t net;         // network (loop-state index)
    int32_t dep_idx;     // counter this tile waits for, or -1
    int32_t dep_tiles;   // column tiles per sweep behind that counter
};
static_assert(sizeof(LeTileRef) == 64, "one tile reference per 64-byte line");
constexpr int kRefWords = 11;
constexpr int kSweepMaxNets = 4096;     // done flags of a launch's networks, one LDS byte each

__device__ __forceinline__ void ref_from_word(uint32_t word, LeTileRef& T) {
    const uint64_t lo = (uint32_t)__builtin_amdgcn_readlane(word, 0), hi = (uint32_t)__builtin_amdgcn_readlane(word, 1);
    T.w = (float*)(uintptr_t)(lo | (hi << 32));
    T.stride = __builtin_amdgcn_readlane(word, 2);
    T.nr = __builtin_amdgcn_readlane(word, 3);
    T.npv = __builtin_amdgcn_readlane(word, 4);
    T.lanes = __builtin_amdgcn_readlane(word, 5);
    T.rel = __builtin_amdgcn_readlane(word, 6);
    T.tile = __builtin_amdgcn_readlane(word, 7);
    T.net = __builtin_amdgcn_readlane(word, 8);
    T.dep_idx = __builtin_amdgcn_readlane(word, 9);
    T.dep_tiles = __builtin_amdgcn_readlane(word, 10);
}
// request the elements of a tile made of 16-byte vectors: the same (row, position) per thread and slot as row_tile<4> /
// col_tile<4> compute for themselves
__device__ __forceinline__ void request_tile(const LeTileRef& T, float (&x)[kSlotsVec4][4]) {
    const int tid = threadIdx.x;
    const int rf = small_div(kBlock, T.lanes);               // rows in flight
    const int trow = small_div(tid, T.lanes);
    const bool on = trow < rf;
    const int row0 = on ? trow : 0;
    const int ln = on ? min(tid - trow * T.lanes, T.npv - 1) : 0;
    const gfloat* w = (const gfloat*)T.w + ln * 4;
#pragma unroll
    for (int u = 0; u < kSlotsVec4; ++u) {
        const int r = min(row0 + u * rf, T.nr - 1);          // slots past the tile repeat its last row (a cache hit)
        vload<4>(w + r * T.stride, x[u]);
    }
}

#ifndef DFQ_LE_SWEEP_MIN_WAVES
#define DFQ_LE_SWEEP_MIN_WAVES 4
#endif
__global__ __launch_bounds__(kBlock, DFQ_LE_SWEEP_MIN_WAVES) void le_sweep_kernel(const LeRelDev* __restrict__ table, const LeTileRef* __restrict__ tiles,
                                                          int n_tiles, LeParams p, int sweep, const LeState* __restrict__ state,
                                                          int n_nets, double* __restrict__ partials,
                                                          unsigned long long* dep_counters, unsigned long long* err, LeTrace tr) {
    __shared__ float sh_f[kSlotMax];
    __shared__ uint32_t sh_u[2 * kSlotMax];
    __shared__ int sh_g[kTileRowsMax];
    __shared__ float sh_p[kSlotMax];
    __shared__ uint32_t sh_rs[2 * kTileRowsMax];
    __shared__ int sh_flag;
    __shared__ unsigned char sh_done[kSweepMaxNets];
    __shared__ int sh_ready[2];                     // is the dependency of the tile of an even / odd iteration satisfied already?
    const int tid = threadIdx.x;
    const int lane = tid % kWave;
    const int G = (int)gridDim.x;
    const int cur = sweep & 1;
    const LeDep dep{dep_counters, err, sweep, p.poll_naps, p.spin_limit};

    // which networks have stopped (the flags only change between launches)
    for (int n = tid; n < n_nets; n += kBlock) sh_done[n] = state[n].done != 0;
    __syncthreads();
    auto net_done = [&](int net) { return sh_done[net] != 0; };

    int t = (int)blockIdx.x;
    if (t >= n_tiles) return;
    // prologue: reference of the first tile, its descriptor / counter / elements, and the reference of the second
    // (only the fetched words are carried around the loop, the fields are re-broadcast from them every iteration)
    uint32_t ref_cur = fetch_words(tiles + t, kRefWords, lane);
    uint32_t ref_next = fetch_words(tiles + min(t + G, n_tiles - 1), kRefWords, lane);
    uint32_t rel_word;
    unsigned long long cnt;
    float v[kSlotsVec4][4];
    {
        LeTileRef F;
        ref_from_word(ref_cur, F);
        rel_word = fetch_words(table + F.rel, kDescWords, lane);
        cnt = __hip_atomic_load(dep_counters + (int64_t)max(F.dep_idx, 0) * kDepStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!(F.npv > 0 && !net_done(F.net))) { F.w = (float*)tiles; F.stride = 0; F.nr = 1; F.npv = 1; F.lanes = 1; }
        request_tile(F, v);
        // ONE thread's view of the counter decides for the workgroup: the waves read it at different times, and a
        // workgroup whose waves disagree on whether to wait would disagree on the number of barriers
        if (tid == 0) sh_ready[0] = F.dep_idx < 0 || cnt >= (unsigned long long)F.dep_tiles * (unsigned long long)(sweep + 1);
        __syncthreads();
    }

    for (int iter = 0;; ++iter) {
        tr.flat = t;
        stamp(tr, 0);
        const bool has_next = t + G < n_tiles;                  // uniform
        LeTileRef T, N;
        ref_from_word(ref_cur, T);
        ref_from_word(ref_next, N);                             // (the last reference again when there is no next tile)
        const bool live = !net_done(T.net);
        const bool next_live = has_next && !net_done(N.net);
        float vn[kSlotsVec4][4];
        uint32_t rel_next = 0u;
        unsigned long long cnt_next = 0ull;
        // the requests for the next tile, issued by the current tile right after its own statistics requests
        // (every request unconditional; with nothing to fetch the eight element requests all hit one 16-byte word)
        const uint32_t ref_next_old = ref_next;
        LeTileRef Q = N;
        if (!(next_live && N.npv > 0)) { Q.w = (float*)tiles; Q.stride = 0; Q.nr = 1; Q.npv = 1; Q.lanes = 1; }
        auto ahead = [&]() __attribute__((always_inline)) {
            request_tile(Q, vn);
            rel_next = fetch_words(table + N.rel, kDescWords, lane);
            cnt_next = __hip_atomic_load(dep_counters + (int64_t)max(N.dep_idx, 0) * kDepStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ref_next = fetch_words(tiles + min(t + 2 * G, n_tiles - 1), kRefWords, lane);
        };
        if (live) {
            union { LeRelDev R; uint32_t u[kDescWords]; } desc;
#pragma unroll
            for (int i = 0; i < kDescWords; ++i) desc.u[i] = __builtin_amdgcn_readlane(rel_word, i);
            // column tiles of a relation whose row tiles publish their rows' statistics wait for THOSE (LeRelDev::local_r1; the
            // tile reference already carries that counter for the look-ahead)
            if (T.tile >= desc.R.n_row_tiles && desc.R.cdep_idx >= 0) { desc.R.dep_idx = desc.R.cdep_idx; desc.R.dep_tiles = desc.R.cdep_tiles; }
            const LeRelDev& R = desc.R;
            stamp(tr, 1);
            // the counter was read one tile ago: if it had already reached its target the producers are done
            const bool ready = sh_ready[iter & 1] != 0;
            const int tile = T.tile;
            const bool col_side = tile >= R.n_row_tiles;
            double acc;
            if (!col_side) {
                if (R.rt_vec == 4) acc = row_tile<4, true>(R, p, tile, cur, dep, ready, v, ahead, sh_f, sh_u, sh_g, sh_p, sh_rs, &sh_flag, tr);
                else {
                    ahead();
                    if (!ready && !dep_wait(R, dep, &sh_flag)) acc = kTileAbandoned;
                    else if (R.rt_vec == 0) acc = short_tile<0>(R, p, tile, cur);
                    else { float v1[kSlotsVec4][1]; acc = row_tile<1, false>(R, p, tile, cur, dep, true, v1, [] {}, sh_f, sh_u, sh_g, sh_p, sh_rs, &sh_flag, tr); }
                }
            } else {
                if (R.ct_vec == 4) acc = col_tile<4, true>(R, p, tile - R.n_row_tiles, cur, dep, ready, v, ahead, sh_f, sh_u, sh_g, sh_p, &sh_flag, tr);
                else {
                    ahead();
                    if (!ready && !dep_wait(R, dep, &sh_flag)) acc = kTileAbandoned;
                    else if (R.ct_vec == 0) acc = short_tile<1>(R, p, tile - R.n_row_tiles, cur);
                    else { float v1[kSlotsVec4][1]; acc = col_tile<1, false>(R, p, tile - R.n_row_tiles, cur, dep, true, v1, [] {}, sh_f, sh_u, sh_g, sh_p, &sh_flag, tr); }
                }
            }
            stamp(tr, 6);
            if (acc < 0.0) return;      // abandoned wait (uniform): nothing was stored; `err` stops every other workgroup
            if (col_side && R.counter_idx >= 0) {
                __builtin_amdgcn_s_waitcnt(0);
                __syncthreads();
                if (tid == 0) atomicAdd(dep_counters + (int64_t)R.counter_idx * kDepStride, 1ull);
            }
            if (!col_side && R.rcounter_idx >= 0) {
                __builtin_amdgcn_s_waitcnt(0);
                __syncthreads();
                if (tid == 0) atomicAdd(dep_counters + (int64_t)R.rcounter_idx * kDepStride, 1ull);
            }
            const double ts = wave_sum(acc);
            if (lane == 0) partials[(int64_t)(R.partial_base + tile) * (kBlock / kWave) + tid / kWave] = ts;
            stamp(tr, 7);
        } else {
            ahead();
        }
        if (!has_next) return;
        // rotate
        ref_cur = ref_next_old;
        rel_word = rel_next;
        cnt = cnt_next;
#pragma unroll
        for (int u = 0; u < kSlotsVec4; ++u) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[u][k] = vn[u][k];
        }
        if (tid == 0) sh_ready[(iter + 1) & 1] = N.dep_idx < 0 || cnt >= (unsigned long long)N.dep_tiles * (unsigned long long)(sweep + 1);
        t += G;
        __syncthreads();               // the LDS tables of this tile are dead before the next one fills them
    }
}

// Stats of the untouched weights, once per run: R1 (rows of W1) for chain-start relations and R2
// (columns of W2) for every relation, parity 0.  One workgroup per `kBootTc` paired channels.
__global__ __launch_bounds__(kBlock) void le_bootstrap_kernel(const LeRelDev* __restrict__ rels,
                                                              const int32_t* __restrict__ rel_of_block) {
    __shared__ uint32_t sh_mn1[kBootTc], sh_mx1[kBootTc], sh_mn2[kBootTc], sh_mx2[kBootTc];
    const LeRelDev R = rels[rel_of_block[blockIdx.x]];
    const int tid = threadIdx.x;
    // a block of kBootTc channels is shared by boot_split workgroups: each takes a slice of the rows' elements and a slice of
    // the second layer's rows, and merges into the (zeroed) statistics words with atomicMax
    const int local = (int)blockIdx.x - R.boot_begin;
    const int cb = local / R.boot_split, sp = local - cb * R.boot_split;
    const int c0 = cb * kBootTc;
    const int nc = min(kBootTc, R.o1 - c0);
    if (tid < nc) { sh_mn1[tid] = 0u; sh_mx1[tid] = 0u; sh_mn2[tid] = 0u; sh_mx2[tid] = 0u; }
    __syncthreads();
    const bool need_rows = R.out_cols == nullptr;      // chain start: nobody else produces R1
    if (need_rows && !(kBootAblate & 1)) {
        // Rows of the first layer: G = pow2 >= (vectors per row) lanes, at most one wave, share a row; 256/G rows are in
        // flight per trip and a row's min/max is reduced inside its lanes with register-file butterflies -- one plain LDS
        // store per row.  (The first version walked the block's elements linearly and merged into LDS with an atomicMax pair
        // whenever a lane's row changed: for 16..160-float rows that is every element, and the row pass was 210 of the
        // launch's 280 us for a batch of 32.)  The boot_split workgroups of a block take disjoint rows.
        const int r_piece = (nc + R.boot_split - 1) / R.boot_split;
        const int r_lo = sp * r_piece, r_hi = min(nc, r_lo + r_piece);
        const bool vec = (R.row_len % 4) == 0 && (((uintptr_t)R.w1) & 15u) == 0;
        const int npv = vec ? R.row_len / 4 : R.row_len;              // positions of a row
        int lgG = 0;
        while ((1 << lgG) < npv && lgG < 6) ++lgG;
        const int G = 1 << lgG, n_rg = kBlock >> lgG;
        const int ln = tid & (G - 1), rg = tid >> lgG;
        const float* rows = R.w1 + (int64_t)c0 * R.row_len;
        if (npv <= G) {
            // short rows: eight rows per trip
            const bool on = ln < npv;
            const int pos = (on ? ln : 0) * (vec ? 4 : 1);
            for (int rb = r_lo; rb < r_hi; rb += 8 * n_rg) {          // uniform trip count: the butterflies below are wave-wide
                const int r0 = rb + rg;
                float mn[8], mx[8];
                if (vec) {
                    fvec4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = *(const fvec4*)(rows + (int64_t)min(r0 + u * n_rg, r_hi - 1) * R.row_len + pos);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        mn[u] = vmin_raw(vmin_raw(v[u][0], v[u][1]), vmin_raw(v[u][2], v[u][3]));
                        mx[u] = vmax_raw(vmax_raw(v[u][0], v[u][1]), vmax_raw(v[u][2], v[u][3]));
                    }
                } else {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = rows[(int64_t)min(r0 + u * n_rg, r_hi - 1) * R.row_len + pos];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { mn[u] = v[u]; mx[u] = v[u]; }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (!on) { mn[u] = INFINITY; mx[u] = -INFINITY; }
                    if (G > 1) xor_lane_minmax<1>(mn[u], mx[u]);
                    if (G > 2) xor_lane_minmax<2>(mn[u], mx[u]);
                    if (G > 4) xor_lane_minmax<4>(mn[u], mx[u]);
                    if (G > 8) xor_lane_minmax<8>(mn[u], mx[u]);
                    if (G > 16) xor_lane_minmax<16>(mn[u], mx[u]);
                    if (G > 32) xor_lane_minmax<32>(mn[u], mx[u]);
                    const int r = r0 + u * n_rg;
                    if (ln == 0 && r < r_hi) { sh_mn1[r] = ~enc_ord(mn[u]); sh_mx1[r] = enc_ord(mx[u]); }
                }
            }
        } else {
            // long rows (more than one wave of positions): a wave per row, eight positions per trip
            for (int r = r_lo + rg; r < r_hi; r += n_rg) {
                const float* row = rows + (int64_t)r * R.row_len;
                float mn = INFINITY, mx = -INFINITY;
                for (int p0 = ln; p0 < npv; p0 += 8 * G) {
                    if (vec) {
                        fvec4 v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = *(const fvec4*)(row + 4 * min(p0 + u * G, npv - 1));
#pragma unroll
                        for (int u = 0; u < 8; ++u) {          // positions past the row repeat its last vector: harmless
                            mn = vmin_raw(vmin_raw(mn, v[u][0]), vmin_raw(v[u][1], vmin_raw(v[u][2], v[u][3])));
                            mx = vmax_raw(vmax_raw(mx, v[u][0]), vmax_raw(v[u][1], vmax_raw(v[u][2], v[u][3])));
                        }
                    } else {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = row[min(p0 + u * G, npv - 1)];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { mn = vmin_raw(mn, v[u]); mx = vmax_raw(mx, v[u]); }
                    }
                }
                xor_lane_minmax<1>(mn, mx); xor_lane_minmax<2>(mn, mx); xor_lane_minmax<4>(mn, mx);
                xor_lane_minmax<8>(mn, mx); xor_lane_minmax<16>(mn, mx); xor_lane_minmax<32>(mn, mx);
                if (ln == 0) { sh_mn1[r] = ~enc_ord(mn); sh_mx1[r] = enc_ord(mx); }
            }
        }
    }
    // this workgroup's rows of the second layer (of every group's go rows)
    const int j_piece = (R.go + R.boot_split - 1) / R.boot_split;
    const int j_lo = sp * j_piece, j_hi = min(R.go, j_lo + j_piece);
    // (the block's channels must lie in ONE group of the second layer: always true for an ungrouped layer)
    const bool wide_cols = R.khkw == 1 && c0 / R.gi == (c0 + nc - 1) / R.gi && (R.i2g % 4) == 0 && (nc % 4) == 0 &&
                           ((c0 - (c0 / R.gi) * R.gi) % 4) == 0 && (((uintptr_t)R.w2) & 15u) == 0;
    if (wide_cols && (kBootAblate & 2)) {
    } else if (wide_cols) {
        // pointwise / linear second layer: G = pow2 >= nc/4 lanes x float4 cover the block's input channels of one row (up to
        // 1 KB contiguous; a layer with <= 256 input channels is read front to back), 256/G row groups per workgroup, eight
        // 16-byte loads in flight per lane.  (One float per lane and row ran at ~1.7 TB/s; 64-channel blocks = 256-byte
        // pieces at 2.3 TB/s whatever the number of workgroups.)
        const int g = c0 / R.gi;
        const int ii0 = c0 - g * R.gi;
        const float* base = R.w2 + ((int64_t)g * R.go * R.i2g + ii0);
        int lgG = 0;
        while ((4 << lgG) < nc) ++lgG;
        const int n_rg = kBlock >> lgG;
        const int lane4 = tid & ((1 << lgG) - 1), rg = tid >> lgG;
        const bool on = 4 * lane4 < nc;
        const float* colp = base + 4 * (on ? lane4 : 0);
        float cmn[4], cmx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { cmn[k] = INFINITY; cmx[k] = -INFINITY; }
        int j = j_lo + rg;
        for (; j + 7 * n_rg < j_hi; j += 8 * n_rg) {
            fvec4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *(const fvec4*)(colp + (int64_t)(j + u * n_rg) * R.i2g);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { cmn[k] = vmin_raw(cmn[k], v[u][k]); cmx[k] = vmax_raw(cmx[k], v[u][k]); }
            }
        }
        for (; j < j_hi; j += n_rg) {
            const fvec4 v = *(const fvec4*)(colp + (int64_t)j * R.i2g);
#pragma unroll
            for (int k = 0; k < 4; ++k) { cmn[k] = vmin_raw(cmn[k], v[k]); cmx[k] = vmax_raw(cmx[k], v[k]); }
        }
        if (on && j_lo + rg < j_hi) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { atomicMax(&sh_mn2[4 * lane4 + k], ~enc_ord(cmn[k])); atomicMax(&sh_mx2[4 * lane4 + k], enc_ord(cmx[k])); }
        }
    } else if (!(kBootAblate & 4)) {
        const int P = nc * R.khkw;
        const int JL = (P >= kBlock) ? 1 : (kBlock / P);
        const int jl = (P >= kBlock) ? 0 : (tid / P);
        const int p_first = (P >= kBlock) ? tid : (tid - jl * P);
        const int64_t col_stride = (int64_t)R.i2g * R.khkw;
        if (jl < JL) {
            for (int pp = p_first; pp < P; pp += kBlock) {
                const int ct = pp / R.khkw;
                const int k = pp - ct * R.khkw;
                const int c = c0 + ct;
                const int g = c / R.gi;
                const int ii = c - g * R.gi;
                const float* col = R.w2 + ((int64_t)g * R.go * R.i2g + ii) * R.khkw + k;
                float mn = INFINITY, mx = -INFINITY;
                // eight rows per trip: the loads are independent, so eight are in flight instead of one
                int j = j_lo + jl;
                for (; j + 7 * JL < j_hi; j += 8 * JL) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = col[(int64_t)(j + u * JL) * col_stride];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { mn = fminf(mn, v[u]); mx = fmaxf(mx, v[u]); }
                }
                for (; j < j_hi; j += JL) {
                    const float v = col[(int64_t)j * col_stride];
                    mn = fminf(mn, v); mx = fmaxf(mx, v);
                }
                if (mn <= mx) { atomicMax(&sh_mn2[ct], ~enc_ord(mn)); atomicMax(&sh_mx2[ct], enc_ord(mx)); }
            }
        }
    }
    __syncthreads();
    if (tid < nc) {
        const int c = c0 + tid;
        if (R.boot_split == 1) {
            if (need_rows) { R.r1[2 * c + 0] = sh_mn1[tid]; R.r1[2 * c + 1] = sh_mx1[tid]; }
            R.r2[2 * c + 0] = sh_mn2[tid]; R.r2[2 * c + 1] = sh_mx2[tid];
        } else {
            if (need_rows) { atomicMax(&R.r1[2 * c + 0], sh_mn1[tid]); atomicMax(&R.r1[2 * c + 1], sh_mx1[tid]); }
            atomicMax(&R.r2[2 * c + 0], sh_mn2[tid]); atomicMax(&R.r2[2 * c + 1], sh_mx2[tid]);
        }
    }
}

struct LeNetDesc {           // one network of a (possibly batched) plan
    int32_t layer_begin, n_layers;     // its targ layers, in graph order, in the layer table
    int32_t tile_begin, n_tiles;       // its contiguous range of tile slots in the partial array
};

// dfq.py:105-115 on the device.  One 16-wave workgroup per network: its per-wave partials and layer
// table are staged into LDS with every load in flight at once, wave w then reduces layers w, w+16,
// ... in a fixed order.  Extra workgroups (blockIdx >= n_nets) clear the stat buffers the next sweep
// accumulates into: parity `cur` of the column stats (R2) and parity `nxt` of the in-sweep row stats
// (leading part of the R1 arena).
__global__ __launch_bounds__(kCtlBlock) void le_control_kernel(const LeLayerDiff* __restrict__ layers,
                                                               const LeNetDesc* __restrict__ nets, int n_nets,
                                                               const double* __restrict__ partials,
                                                               double* __restrict__ layer_mean,
                                                               uint32_t* __restrict__ r2_arena, int64_t r2_words,
                                                               uint32_t* __restrict__ r1_arena, int64_t r1_words,
                                                               int64_t r1_zero_words, int parity,
                                                               LeState* __restrict__ states, double converge_thres,
                                                               int converge_count, int max_sweeps, int uni_layers, int uni_tiles,
                                                               int n_clear, const LeCfSeg* __restrict__ cf_segs,
                                                               const LeCfRel* __restrict__ cf_rels, const int32_t* __restrict__ cf_map,
                                                               LeParams cf_p, int cf_k0, int cf_group) {
    __shared__ double sh_part[kCtlStage];
    __shared__ double sh_mean[1024];
    __shared__ LeLayerDiff sh_layer[1024];
    const int tid = threadIdx.x;
    const int lane = tid % kWave;
    const int wave = tid / kWave;
    const int cur = parity;          // sweep index & 1, from the host
    if ((int)blockIdx.x >= n_nets + n_clear) {
        // free-running segments (dfq_le_cf.hpp): at the last sweep of a group these workgroups advance the recurrence by the next
        // group's sweeps -- a thread per channel, no weight is read -- next to the verdicts and the clearing
        cf_solve_block(cf_segs, cf_rels, cf_map, (int)blockIdx.x - n_nets - n_clear, cf_p, cf_k0, cf_group, cf_group, false, states);
        return;
    }
    if ((int)blockIdx.x >= n_nets) {
        // helper workgroups: clear the stat words the next sweep accumulates into (harmless once a
        // network has converged: its stats are never read again)
        // (16 bytes per store, up to one helper per CU: 6.3 MB per sweep for the benchmark's batch of 32 -- with 32 helpers
        // storing 4 bytes per thread this kernel took 8 us, 5.5 % of a sweep; both bases are 16-byte aligned, see stat_words)
        const int64_t nz = n_clear, z = blockIdx.x - n_nets;
        auto clear = [&](uint32_t* base, int64_t n) {
            const int64_t n4 = n >> 2;
            float4* const b4 = reinterpret_cast<float4*>(base);            // all-zero bits either way
            float4 zero;
            zero.x = zero.y = zero.z = zero.w = 0.f;
            for (int64_t i = z * kCtlBlock + tid; i < n4; i += nz * kCtlBlock) b4[i] = zero;
            for (int64_t i = (n4 << 2) + z * kCtlBlock + tid; i < n; i += nz * kCtlBlock) base[i] = 0u;
        };
        clear(r2_arena + (int64_t)cur * r2_words, r2_words);
        clear(r1_arena + (int64_t)(cur ^ 1) * r1_words, r1_zero_words);
        return;
    }
    LeState* const state = states + blockIdx.x;
    // a batch of like networks (uni_layers > 0): the descriptor is arithmetic, and the partials, the layer table and the state are
    // requested at once instead of behind the descriptor's round trip through the memory system
    LeNetDesc nd;
    if (uni_layers > 0) {
        nd.layer_begin = (int)blockIdx.x * uni_layers; nd.n_layers = uni_layers;
        nd.tile_begin = (int)blockIdx.x * uni_tiles; nd.n_tiles = uni_tiles;
    } else {
        nd = nets[blockIdx.x];
    }
    layers += nd.layer_begin;
    layer_mean += nd.layer_begin;
    const int n_layers = nd.n_layers;
    // every global read of the kernel is issued before the first wait: partials, layer table, state
    const int waves_per_tile = kBlock / kWave;
    const int64_t part0 = (int64_t)nd.tile_begin * waves_per_tile;
    const int n_stage = min(nd.n_tiles * waves_per_tile, kCtlStage);
    const LeState before = *state;       // requested with the partials: the verdict at the end does not wait for it again
    for (int i = tid; i < n_stage; i += kCtlBlock) sh_part[i] = partials[part0 + i];
    for (int i = tid; i < min(n_layers, 1024); i += kCtlBlock) sh_layer[i] = layers[i];
    if (before.done) return;
    __syncthreads();
    for (int l = wave; l < n_layers; l += kCtlBlock / kWave) {
        const LeLayerDiff L = (l < 1024) ? sh_layer[l] : layers[l];
        double s = 0.0;
        if (L.partial_begin >= 0) {
            // every tile left one partial per wave
            const int rel0 = (L.partial_begin - nd.tile_begin) * waves_per_tile;     // offset inside the staged range
            // (same order of additions as one value per trip; eight loads in flight for the part of a large network that
            // did not fit the staging buffer -- ResNet-18's 10 868 partials made this kernel 14.7 us instead of 6)
            const int n_l = L.n_partials * waves_per_tile;
            for (int i = lane; i < n_l; i += 8 * kWave) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = rel0 + i + u * kWave;
                    x[u] = (i + u * kWave < n_l) ? ((idx < n_stage) ? sh_part[idx] : partials[part0 + idx]) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) s += x[u];
            }
            s = wave_sum(s);
        }
        if (lane == 0) {
            // float(torch.mean(torch.abs(W - W_prev))): float32 mean, widened to double (dfq.py:108)
            const double m = (L.partial_begin >= 0) ? (double)(float)(s / L.n_elems) : 0.0;
            if (l < 1024) sh_mean[l] = m; else layer_mean[l] = m;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    // diff_tmp = sum of the layer means IN GRAPH ORDER (Python's left-to-right float64 sum).  Lane l
    // fetches mean[l] with one LDS instruction per 64 layers; the values then travel lane by lane
    // through a shuffle so that lane 0 adds them in order (adding +0.0 for missing layers is exact).
    // (every lane reads the same word: a broadcast; until round 5 lane l fetched mean[l] and the values travelled to lane 0 through
    // 64 shuffles per 64 layers -- same additions in the same order, a third of the instructions)
    double diff_tmp = 0.0;
    {
        const int n_lds = min(n_layers, 1024);
        int l = 0;
        for (; l + 8 <= n_lds; l += 8) {
            double m[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) m[u] = sh_mean[l + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) diff_tmp += m[u];
        }
        for (; l < n_lds; ++l) diff_tmp += sh_mean[l];
        for (; l < n_layers; ++l) diff_tmp += layer_mean[l];
    }
    if (tid == 0) {
        double diff = before.diff;
        int count = before.count;
        if (fabs(diff - diff_tmp) > 1e-9) { count = 0; diff = diff_tmp; }
        else { count += 1; }
        const int sweeps = before.sweeps + 1;
        const bool go_on = (diff > converge_thres) && (count < converge_count) &&
                           (max_sweeps < 0 || sweeps < max_sweeps);
        state->diff = diff;
        state->count = count;
        state->sweeps = sweeps;
        state->last_diff_tmp = diff_tmp;
        state->done = go_on ? 0 : 1;
        if (go_on) state->happen = sweeps;                   // sweep `sweeps` (the next one) happens
        if (before.log && before.sweeps < before.log_cap) before.log[before.sweeps] = diff_tmp;
    }
}

__global__ void le_reset_kernel(LeState* states, int n_nets, double converge_thres, int converge_count, int max_sweeps,
                                unsigned long long* err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && err) *err = 0ull;              // the plan's "a wait was abandoned" word
    if (i < n_nets) {
        LeState* state = states + i;
        state->diff = 10.0;          // dfq.py:81
        state->count = 0;            // dfq.py:82
        state->sweeps = 0;
        state->last_diff_tmp = 0.0;
        const bool go_on = (10.0 > converge_thres) && (0 < converge_count) && (max_sweeps != 0);
        state->done = go_on ? 0 : 1;
        state->happen = go_on ? 0 : -1;
    }
}

// restart of a streaming plan in ONE launch: the loop state of every network (dfq.py:81-82) and the clearing of the dependency
// counters + error word and of every statistics word (until round 4: le_reset_kernel + clear_kernel)
__global__ void le_prepare_kernel(ClearArgs a, LeState* states, int n_nets, double converge_thres, int converge_count, int max_sweeps,
                                  float* ones, long long n_ones) {
    const long long step = (long long)gridDim.x * blockDim.x;
    const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < 4; ++k)
        for (long long i = i0; i < a.words[k]; i += step) a.p[k][i] = 0u;
    // the factor rings of the free-running segments (dfq_le_cf.hpp): every entry 1, so that the first group's replay changes nothing
    for (long long i = i0; i < n_ones; i += step) ones[i] = 1.0f;
    for (long long i = i0; i < n_nets; i += step) {
        LeState* state = states + i;
        state->diff = 10.0;          // dfq.py:81
        state->count = 0;            // dfq.py:82
        state->sweeps = 0;
        state->last_diff_tmp = 0.0;
        const bool go_on = (10.0 > converge_thres) && (0 < converge_count) && (max_sweeps != 0);
        state->done = go_on ? 0 : 1;
        state->happen = go_on ? 0 : -1;
    }
}

// Deferred stores: apply the factors still pending for a network (sweeps % depth of them) to its one-way-scaled layers.
// One workgroup per span of kFlushSpan elements of one layer; workgroups of networks with nothing pending leave at once.
// A span's reference is self-contained (64 bytes, fetched with ONE wave-wide load and broadcast with v_readlane): the first
// version went span -> relation descriptor (fields fetched piecemeal) -> loop state -> data, three dependent round trips before
// the first element was requested, with four vectors per thread in flight: 258 us for the 516 MB of a batch of 32 (2 TB/s).
constexpr int kFlushSpan = 8192;
struct alignas(128) LeFlushRef {
    float* w;            // first element of the span
    const float* hold;   // remembered factors of this side: factor j of channel c is hold[2 j o1 + c]
    int32_t span;        // elements
    int32_t row_len;     // of the layer
    int32_t row0;        // row and position in that row of the first element
    int32_t pos0;
    int32_t side;        // 0: W1 (rows by s), 1: W2 (columns by 1/s), 2: a depthwise layer between two free-running relations
                         //    (row o by 1/s of the relation in front -- hold2 -- and then by s of its own -- hold --, sweep by sweep)
    int32_t net;
    int32_t go, gi, khkw, o1;
    int32_t vec;         // 1: the span is made of aligned 16-byte vectors that never cross a row
    int32_t fr;          // layer of a free-running segment (dfq_le_cf.hpp): `hold` is the relation's factor ring, the pending
                         //    sweeps are the ones after the group's first (cf_pending)
    const float* hold2;  // side 2: the ring of the relation in front, at its 1/s
    int32_t o1_2;        // channels of that relation
    int32_t pc_gi;       // side 2: row o is channel o * pc_gi of it
    int32_t pad[12];
};
static_assert(sizeof(LeFlushRef) == 128, "one span reference per 128-byte line");
__global__ __launch_bounds__(kBlock) void le_flush_kernel(const LeFlushRef* __restrict__ refs, const LeState* __restrict__ state, int depth,
                                                          int cf_group) {
    const int lane = threadIdx.x % kWave;
    const uint32_t word = fetch_words(refs + blockIdx.x, 20, lane);
    LeFlushRef ref;
    {
        const uint64_t w_lo = (uint32_t)__builtin_amdgcn_readlane(word, 0), w_hi = (uint32_t)__builtin_amdgcn_readlane(word, 1);
        const uint64_t h_lo = (uint32_t)__builtin_amdgcn_readlane(word, 2), h_hi = (uint32_t)__builtin_amdgcn_readlane(word, 3);
        ref.w = (float*)(uintptr_t)(w_lo | (w_hi << 32));
        ref.hold = (const float*)(uintptr_t)(h_lo | (h_hi << 32));
        ref.span = __builtin_amdgcn_readlane(word, 4); ref.row_len = __builtin_amdgcn_readlane(word, 5);
        ref.row0 = __builtin_amdgcn_readlane(word, 6); ref.pos0 = __builtin_amdgcn_readlane(word, 7);
        ref.side = __builtin_amdgcn_readlane(word, 8); ref.net = __builtin_amdgcn_readlane(word, 9);
        ref.go = __builtin_amdgcn_readlane(word, 10); ref.gi = __builtin_amdgcn_readlane(word, 11);
        ref.khkw = __builtin_amdgcn_readlane(word, 12); ref.o1 = __builtin_amdgcn_readlane(word, 13);
        ref.vec = __builtin_amdgcn_readlane(word, 14); ref.fr = __builtin_amdgcn_readlane(word, 15);
        const uint64_t g_lo = (uint32_t)__builtin_amdgcn_readlane(word, 16), g_hi = (uint32_t)__builtin_amdgcn_readlane(word, 17);
        ref.hold2 = (const float*)(uintptr_t)(g_lo | (g_hi << 32));
        ref.o1_2 = __builtin_amdgcn_readlane(word, 18); ref.pc_gi = __builtin_amdgcn_readlane(word, 19);
    }
    int slot0 = 0;
    const int pend = ref.fr ? cf_pending(state[ref.net].sweeps, cf_group, &slot0) : (state[ref.net].sweeps & (depth - 1));
    if (pend == 0) return;
    if (ref.side == 2) {
        // depthwise layer between two free-running relations (a few thousand floats): per sweep dfq.py:73 of the relation in
        // front, then dfq.py:62 of its own -- the two roundings the lean tile would have performed
        const int row_len = ref.row_len;
        gfloat* const w2 = (gfloat*)ref.w;
        for (int off = (int)threadIdx.x; off < ref.span; off += kBlock) {
            const int o = ref.row0 + small_div(ref.pos0 + off, row_len);
            float x = w2[off];
            for (int j = 0; j < pend; ++j) {
                x = x * ref.hold2[(int64_t)(2 * (slot0 + j)) * ref.o1_2 + o * ref.pc_gi];
                x = x * ref.hold[(int64_t)(2 * (slot0 + j)) * ref.o1 + o];
            }
            w2[off] = x;
        }
        return;
    }
    const int side = ref.side, row_len = ref.row_len, span = ref.span;
    gfloat* const w = (gfloat*)ref.w;
    const float* const hold = ref.hold;
    // channel of the element `off` floats into the span (off < kFlushSpan, positions < 2^20: small_div is exact)
    auto channel = [&](int off) {
        const int p = ref.pos0 + off;
        const int dr = small_div(p, row_len);
        const int o = ref.row0 + dr;
        return side == 0 ? o : small_div(o, ref.go) * ref.gi + small_div(p - dr * row_len, ref.khkw);
    };
    if (ref.vec) {
        constexpr int NV = kFlushSpan / (4 * kBlock);                 // vectors per thread, all requested before the first use
        fvec4 x[NV];
        int off[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            off[k] = 4 * (k * kBlock + (int)threadIdx.x);
            if (off[k] < span) x[k] = DFQ_NT_LOAD((const gfvec4*)(w + off[k]));
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (off[k] >= span) continue;
            // the factors of a vector's four elements: ONE load per pending sweep when they share a channel (rows by s: a
            // vector never crosses a row) or sit on consecutive channels (columns of a 1x1 layer, 16-byte aligned), four
            // otherwise -- as four gathers per vector and sweep this loop kept the address units busier than the data did
            const int c0 = channel(off[k]);
            if (side == 0) {
                for (int j = 0; j < pend; ++j) {
                    const float hj = hold[(int64_t)(2 * (slot0 + j)) * ref.o1 + c0];
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[k][e] = x[k][e] * hj;
                }
            } else if (ref.khkw == 1 && (((uintptr_t)(hold + c0)) & 15) == 0 && ((2 * ref.o1) & 3) == 0) {
                for (int j = 0; j < pend; ++j) {
                    const fvec4 hj = *(const fvec4*)(hold + (int64_t)(2 * (slot0 + j)) * ref.o1 + c0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[k][e] = x[k][e] * hj[e];
                }
            } else {
                const int c1 = channel(off[k] + 1), c2 = channel(off[k] + 2), c3 = channel(off[k] + 3);
                for (int j = 0; j < pend; ++j) {
                    const float* h = hold + (int64_t)(2 * (slot0 + j)) * ref.o1;
                    x[k][0] = x[k][0] * h[c0]; x[k][1] = x[k][1] * h[c1]; x[k][2] = x[k][2] * h[c2]; x[k][3] = x[k][3] * h[c3];
                }
            }
            DFQ_NT_STORE(x[k], (gfvec4*)(w + off[k]));
        }
    } else {
        for (int off = (int)threadIdx.x; off < span; off += kBlock) {
            const int c = channel(off);
            float x = w[off];
            for (int j = 0; j < pend; ++j) x = x * hold[(int64_t)(2 * (slot0 + j)) * ref.o1 + c];
            w[off] = x;
        }
    }
}

// remembered factors back to 1 (so that a later sweep that replays them changes nothing): every relation with deferred
// layers of the networks that had something pending (`force`: all of them, at a restart)
__global__ __launch_bounds__(kBlock) void le_hold_reset_kernel(const LeRelDev* __restrict__ table, const int32_t* __restrict__ rels,
                                                               const LeState* __restrict__ state, int depth, int force) {
    const LeRelDev& R = table[rels[blockIdx.x]];
    if (!force && (state[R.net].sweeps & (depth - 1)) == 0) return;
    const int64_t n = (int64_t)2 * (depth - 1) * R.o1;
    for (int64_t i = threadIdx.x; i < n; i += kBlock) R.hold[i] = 1.0f;
}

// ---------------------------------------------------------------------------------------------
// host side: the plan
// ---------------------------------------------------------------------------------------------
struct LevelLaunch {
    int rel_begin = 0;      // range in the level-sorted device relation table
    int n_rels = 0;
    int n_blocks = 0;       // workgroups (all of them do work)
    int block_begin = 0;    // first entry of this level in the sweep's workgroup table
    int64_t rw_elems = 0;   // elements read AND written by this launch (8 B each)
    int64_t ro_elems = 0;   // elements only read by it: statistics pass over interior layers (4 B each)
};

}  // namespace dfq

using namespace dfq;

struct dfq_le_plan {
    dfq::DevSlab mem;                 // every device table below lives in here

    int n_layers = 0, n_rels = 0, n_nets = 1;
    LeNetDesc* d_nets = nullptr;
    int uni_layers = 0, uni_tiles = 0;       // > 0: every network of the plan has this many layers / tiles, laid out back to back
    int32_t* d_boot_map = nullptr;         // bootstrap workgroup -> relation
    std::vector<LevelLaunch> levels;
    int64_t paired_total = 0;              // sum over relations of n1 + n2 (SURVEY 8d's Sigma_rel)
    int64_t rw_total = 0, ro_total = 0;    // per sweep: elements read+written / only read
    // deferred stores: depth (1 = off), elements of the layers concerned (part of rw_total: read every sweep, written every
    // `defer`-th), the remembered factors, the workgroup tables of the flush / reset launches
    int defer = 1;
    int64_t deferred_total = 0;
    float* d_hold = nullptr;
    LeFlushRef* d_flush = nullptr;
    int32_t* d_hold_rels = nullptr;
    int n_flush = 0, n_hold_rels = 0;
    int total_tiles = 0, boot_blocks = 0;
    int64_t stat_words = 0;                // per parity, per arena
    int64_t r1_zero_words = 0;             // leading part of the R1 arena that is accumulated with atomics
    int64_t sweep_index = 0;               // sweeps enqueued since the last restart (parity = & 1)
    hipStream_t capture_stream = nullptr;  // private stream used only to record graphs
    struct CachedGraph {
        std::vector<unsigned char> key;    // (n_sweeps, restart, start parity, config bytes)
        hipGraphExec_t exec;
    };
    std::vector<CachedGraph> graphs;
    LeRelDev* d_rels = nullptr;
    LeBlockRef* d_blocks = nullptr;        // workgroup table of a sweep: level after level
    unsigned long long* d_dep = nullptr;   // per-relation counters of finished column tiles (padded) + error flag
    bool merged = true;                    // one launch per sweep (false: one per level, DFQ_LE_MERGED=0)
    bool resident_off = false;             // the resident engine abandoned a wait once (nothing stored): this plan streams from now on
    int degraded = 0;                      // runs that were repeated on the per-level launches after such an abandon (dfq_le_plan_degraded)
    LeLayerDiff* d_layer_diff = nullptr;
    double* d_partials = nullptr;
    double* d_layer_mean = nullptr;
    LeState* d_state = nullptr;
    uint32_t* d_stats = nullptr;           // R2 arena [2][stat_words], then R1 arena [2][stat_words]
    // single networks that fit the LDS of the chip run the whole loop as ONE persistent launch (dfq_le_resident.hip);
    // null: the streaming one-launch-per-sweep kernel above (batched plans, networks too large, DFQ_LE_RESIDENT=0)
    dfq::LeResident* resident = nullptr;
    std::string resident_why;
    std::vector<LeRelDev> h_rels;          // host copies of the two tables (dfq_le_plan_block_info)
    std::vector<LeBlockRef> h_blocks;
    // DFQ_LE_PERSIST=1: one-launch sweeps run as persistent workgroups that keep the next tile's data in flight
    // (le_sweep_kernel; an experiment that measured slower than one workgroup per tile)
    LeTileRef* d_tiles = nullptr;
    int sweep_grid = 0;                    // workgroups of le_sweep_kernel; 0: le_level_kernel
    // free-running segments (dfq_le_cf.hpp): their layers are not in the sweep's workgroup table -- le_lean_kernel handles a
    // group of cf_group sweeps at the group's first sweep, the solver workgroups of the convergence launch keep the factor rings
    int cf_group = 1;                      // G; 1: no free-running segments in this plan
    int n_cf_rels = 0, n_cf_segs = 0, n_cf_blocks = 0, n_lean = 0;
    LeCfRel* d_cf_rels = nullptr;
    LeCfSeg* d_cf_segs = nullptr;
    int32_t* d_cf_map = nullptr;           // solver workgroup -> (segment, block of kCtlBlock channels)
    LeLeanRef* d_lean = nullptr;
    float* d_cf_ring = nullptr;            // every relation's ring, back to back (set to 1 by the restart's first launch)
    int64_t cf_ring_floats = 0;
    int64_t fr_total = 0;                  // elements of free-running layers: read and written once per cf_group sweeps
    int64_t part_stride = 0;               // doubles of one sweep's partial sums (the array exists 2 * cf_group times: sweep j uses array j mod 2G)
    // background mode of the lean launches (dfq_le_cf.hpp): a second, low-priority stream next to the group's sweep launches
    bool cf_bg = false;
    hipStream_t bg_stream = nullptr;
    hipEvent_t bg_fork = nullptr;          // recorded on the caller's stream where a background launch may begin
    hipEvent_t bg_done[2] = {nullptr, nullptr};   // recorded behind the background launch of group g in bg_done[g & 1] ...
    int64_t bg_wait_at[2] = {-1, -1};      // ... which the convergence launch of THIS sweep (and nothing earlier) has to wait for; -1: none
    std::vector<int> lean_info;            // per lean tile: kind, rows, floats per row (dfq_le_plan_lean_info)
    // the sweep's workgroup table with the lean tiles woven in evenly (group depth kWeaveGroup, one launch per sweep): the launch
    // of a group's first sweep; null: the lean tiles get a launch of their own (le_lean_kernel)
    LeBlockRef* d_blocks_woven = nullptr;
    int n_woven = 0;
    bool has_slab_tiles = false;           // some row tiles wait for their row block's other slabs (kLocalSlabs): also a per-level launch contains in-launch waits
};

// elements per tile; DFQ_LE_TILE_ELEMS overrides (tuning / tests).  A workgroup's fixed cost (workgroup table ->
// descriptor -> statistics -> solve -> barrier: about four dependent global round trips) is the same for
// any tile size, and the register budget admits only 5-6 workgroups per CU to hide it: throughput-oriented
// (batched) plans therefore use the largest tile the register slots hold, a single network keeps smaller
// tiles because there the number of workgroups in flight, not their cost, bounds a launch.
static int tile_target(bool batched) {
    const char* e = getenv("DFQ_LE_TILE_ELEMS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : (batched ? 8192 : 4096);
}
// columns of a row tile that publishes column statistics.  Every row block of a layer merges its column
// minima/maxima into the same global words with atomicMax, so the number of atomics per sweep is
// (rows / tile rows) x columns: tall narrow tiles.  (A 960x960 layer cut into 17-row tiles issued 57
// atomics per statistics word and sweep, and that, not the data, set the launch's duration.)
static int emit_cols_max(int vec) {
    const char* e = getenv("DFQ_LE_EMIT_COLS");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : (vec == 4 ? 64 : 32);
}

// columns of a row tile that publishes nothing (a row's scale is all it needs): DFQ_LE_ROW_COLS overrides (tuning; at
// most 1024 floats = 256 lanes x 16 bytes)
static int plain_row_cols_max(int vec) {
    const char* e = getenv("DFQ_LE_ROW_COLS");
    const int v = e ? atoi(e) : 0;
    if (v > 0) return std::min(v, vec == 4 ? 1024 : 256);
    return kRowTileColsMax * (vec == 4 ? 2 : 1);
}
// columns of a col tile: every column costs the tile four statistics words (and the remembered factors of deferred stores)
// whatever its height, so narrow tall tiles read fewer of them per element; DFQ_LE_COL_COLS overrides (tuning)
// (batch of 32 MobileNetV2: 256 / 128 / 64 / 32 floats -> 175 / 174 / 168 / 177 us per forced sweep; a 32 x 240 tile reads
// 6.7 KB of statistics and factors for 30 KB of elements, a 128 x 64 tile 1.8 KB)
static int col_cols_max(int vec, bool batched) {
    const char* e = getenv("DFQ_LE_COL_COLS");
    const int v = e ? atoi(e) : 0;
    if (v > 0) return std::min(v, kColTileLanes * vec);
    return (batched && vec == 4) ? 64 : kColTileLanes * vec;
}
static int ceil_div(int a, int b) { return (a + b - 1) / b; }
// group depth of the free-running segments (dfq_le_cf.hpp): DFQ_LE_CF=0 or DFQ_LE_CF_GROUP=1 keep every layer on the general
// tiles; 2 / 4 / 8 sweeps per pass over a free-running layer
// Default: 8 for a batched plan (batch of 32 MobileNetV2, two alternating rounds, tools/gpu_r06_g8.sh: 1.741 / 1.743e10 weights/s at
// depth 4, 1.788 / 1.801e10 at depth 8 -- the lean launch 73 us per 4 sweeps against 104 us per 8), 4 for a single network (a
// ResNet-18 stops after two sweeps: every sweep a lean tile looks ahead beyond the loop's end is arithmetic for nothing).
// A single network SMALL enough for the chip's caches (MobileNetV2, DeepLab: the sharded pass, the repeat of an abandoned resident
// launch) keeps every layer on the general tiles unless DFQ_LE_CF_GROUP asks: its sweep is two launch latencies, not bandwidth,
// and the lean launches only add to them (tools/lat.py, DFQ_LE_RESIDENT=0, two alternating rounds: MobileNetV2 1.87 ms with
// segments of depth 4 against 1.82 without, DeepLab 1.22 against 1.14 -- profiles/r06_experiments.txt 8).
static int cf_group_from_env(bool batched, bool large) {
    const char* off = getenv("DFQ_LE_CF");
    if (off && off[0] == '0') return 1;
    const char* e = getenv("DFQ_LE_CF_GROUP");
    const int v = e ? atoi(e) : (batched ? 8 : large ? 4 : 1);
    return v >= 8 ? 8 : v >= 4 ? 4 : v >= 2 ? 2 : 1;
}
// rows of a full-row tile (row_tile's local mode): what the register slots hold, at most the tile target
static int local_rows(int n_rows, int row_len, int target) {
    const int cap = kSlotsVec4 * (kBlock / (row_len / 4));
    return std::max(1, std::min(std::min(std::min(kTileRowsMax, n_rows), cap), std::max(1, target / row_len)));
}

// [rows x cols] tiling of a [n_rows, row_len] matrix whose tiles move `vec`-wide vectors:
// cols = slab width (multiple of vec, <= cols_max), rows so that a thread holds <= its register slots.
static void tile_shape(int n_rows, int row_len, int vec, int cols_max, bool pow2_lanes, int target,
                       int* rows, int* cols, int* slabs) {
    int s = ceil_div(row_len, cols_max);
    int c = ceil_div(ceil_div(row_len, s), vec) * vec;
    s = ceil_div(row_len, c);
    int lanes = c / vec;                              // threads along a row
    if (pow2_lanes) { int g = 1; while (g < lanes) g <<= 1; lanes = g; }
    const int rows_in_flight = kBlock / lanes;
    if (vec == 1) target = std::min(target, 2048);    // scalar tiles: one memory instruction per float
    int r = std::max(1, target / c);
    r = std::min(r, (vec == 4 ? kSlotsVec4 : kSlotsVec1) * rows_in_flight);  // register preload capacity
    r = std::min(r, std::min(kTileRowsMax, n_rows));
    *rows = r; *cols = c; *slabs = s;
}

extern "C" {

void dfq_le_plan_destroy(dfq_le_plan* p) {
    if (!p) return;
    dfq::dev_quiesce();                                  // nothing in flight may still use the blocks released below
    p->mem.release();
    for (auto& g : p->graphs) (void)hipGraphExecDestroy(g.exec);
    if (p->capture_stream) (void)hipStreamDestroy(p->capture_stream);
    if (p->bg_stream) (void)hipStreamDestroy(p->bg_stream);
    if (p->bg_fork) (void)hipEventDestroy(p->bg_fork);
    for (auto& e : p->bg_done) if (e) (void)hipEventDestroy(e);
    if (p->resident) le_resident_destroy(p->resident);
    delete p;
}

int dfq_le_plan_create(const dfq_layer* layers, int32_t n_layers, const dfq_relation* relations,
                       int32_t n_relations, dfq_le_plan** out_plan) {
    return dfq_le_plan_create_batch(layers, n_layers, nullptr, 1, relations, n_relations, out_plan);
}

int dfq_le_plan_create_replicated(const dfq_layer* layers, int32_t n_layers, const dfq_relation* relations, int32_t n_relations,
                                  const void* const* bases, int32_t n_nets, dfq_le_plan** out_plan) {
    if (!layers || n_layers <= 0 || n_relations < 0 || (n_relations > 0 && !relations) || !bases || n_nets < 1 || !out_plan)
        return fail_arg("dfq_le_plan_create_replicated: bad argument");
    if ((int64_t)n_layers * n_nets > INT32_MAX || (int64_t)n_relations * n_nets > INT32_MAX)
        return fail_arg("dfq_le_plan_create_replicated: too many layers");
    std::vector<dfq_layer> L((size_t)n_layers * n_nets);
    std::vector<dfq_relation> R((size_t)std::max(1, n_relations) * n_nets);
    std::vector<int32_t> net((size_t)n_layers * n_nets);
    const intptr_t b0 = (intptr_t)bases[0];
    auto moved = [](const float* p, intptr_t by) { return p ? (const float*)((const char*)p + by) : nullptr; };
    for (int n = 0; n < n_nets; ++n) {
        if (!bases[n]) return fail_arg("dfq_le_plan_create_replicated: network %d has no base address", n);
        const intptr_t by = (intptr_t)bases[n] - b0;
        for (int l = 0; l < n_layers; ++l) {
            dfq_layer& d = L[(size_t)n * n_layers + l];
            d = layers[l];
            d.weight = (float*)moved(d.weight, by);
            d.bias = (float*)moved(d.bias, by);
            net[(size_t)n * n_layers + l] = n;
        }
        for (int r = 0; r < n_relations; ++r) {
            dfq_relation& d = R[(size_t)n * n_relations + r];
            d = relations[r];
            d.first += n * n_layers;
            d.second += n * n_layers;
            d.bn_weight = (float*)moved(d.bn_weight, by);
            d.bn_bias = (float*)moved(d.bn_bias, by);
            d.scale_cum = (float*)moved(d.scale_cum, by);
        }
    }
    return dfq_le_plan_create_batch(L.data(), n_layers * n_nets, net.data(), n_nets, R.data(), n_relations * n_nets, out_plan);
}

int dfq_le_plan_create_batch(const dfq_layer* layers, int32_t n_layers, const int32_t* layer_net, int32_t n_nets,
                             const dfq_relation* relations, int32_t n_relations, dfq_le_plan** out_plan) {
    if (!layers || n_layers <= 0 || !out_plan || n_relations < 0 || (n_relations > 0 && !relations) || n_nets < 1 ||
        (n_nets > 1 && !layer_net))
        return fail_arg("dfq_le_plan_create: bad argument");
    PlanTimer timer("dfq_le_plan_create");
    auto net_of = [&](int l) { return layer_net ? layer_net[l] : 0; };
    for (int l = 0; l < n_layers; ++l) {
        if (net_of(l) < 0 || net_of(l) >= n_nets || (l > 0 && net_of(l) < net_of(l - 1)))
            return fail_arg("dfq_le_plan_create: layers must be listed network by network (layer %d)", l);
    }
    for (int r = 0; r < n_relations; ++r) {
        const dfq_relation& rr = relations[r];
        if (rr.first < 0 || rr.first >= n_layers || rr.second < 0 || rr.second >= n_layers) continue;   // reported below
        if (net_of(rr.first) != net_of(rr.second))
            return fail_arg("dfq_le_plan_create: relation %d pairs layers of two networks", r);
        if (r > 0 && relations[r - 1].first >= 0 && relations[r - 1].first < n_layers &&
            net_of(rr.first) < net_of(relations[r - 1].first))
            return fail_arg("dfq_le_plan_create: relations must be listed network by network (relation %d)", r);
    }
    // ---- validate geometry (dfq.py:29-35) and the chain structure create_relation guarantees ----
    std::vector<int> as_first(n_layers, -1), as_second(n_layers, -1);
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
        if ((int64_t)A.in_per_group * A.khkw >= (1 << 20) || (int64_t)B.in_per_group * B.khkw >= (1 << 20) ||
            A.out_ch >= (1 << 20) || B.out_ch >= (1 << 20))
            return fail_arg("dfq_le_plan_create: relation %d: layer dimension >= 2^20", r);
        if (as_first[rr.first] >= 0 || as_second[rr.second] >= 0)
            return fail_arg("dfq_le_plan_create: relation %d: a layer may be first in one relation and second in one "
                            "relation only (utils/relation.py:57-67)", r);
        as_first[rr.first] = r;
        as_second[rr.second] = r;
    }
    for (int l = 0; l < n_layers; ++l)
        if (as_first[l] >= 0 && as_second[l] >= 0 && as_second[l] > as_first[l])
            return fail_arg("dfq_le_plan_create: layer %d is rescaled as a first layer (relation %d) before it is "
                            "rescaled as a second layer (relation %d); unsupported order", l, as_first[l], as_second[l]);
    dfq_le_plan* p = new dfq_le_plan();
    p->n_layers = n_layers;
    p->n_rels = n_relations;
    p->n_nets = n_nets;
    auto fail_alloc = [&](hipError_t e) { dfq_le_plan_destroy(p); return fail_hip(e, "le plan allocation", __FILE__, __LINE__); };

    timer.tick("validate");
    // ---- dependency levels: relations sharing a layer keep their list order ----
    std::vector<int> level(n_relations, 0), last_level(n_layers, -1);
    for (int r = 0; r < n_relations; ++r) {
        const int lv = std::max(last_level[relations[r].first], last_level[relations[r].second]) + 1;
        level[r] = lv;
        last_level[relations[r].first] = lv;
        last_level[relations[r].second] = lv;
    }

    timer.tick("levels");
    // ---- stat arenas.  R2 (column stats) of every relation; R1 (row stats) with the relations whose
    //      R1 is accumulated in-sweep (their first layer is someone's second layer) laid out first so
    //      the control kernel can clear exactly that part. ----
    std::vector<int64_t> r2_off(n_relations, 0), r1_off(n_relations, 0);
    int64_t words = 0;
    for (int r = 0; r < n_relations; ++r) { r2_off[r] = words; words += 2 * (int64_t)layers[relations[r].first].out_ch; }
    words = (words + 3) & ~(int64_t)3;     // every parity of every arena starts on a 16-byte boundary (the clearing stores)
    p->stat_words = words;
    int64_t w1 = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int r = 0; r < n_relations; ++r) {
            const bool interior = as_second[relations[r].first] >= 0;
            if (interior == (pass == 0)) { r1_off[r] = w1; w1 += 2 * (int64_t)layers[relations[r].first].out_ch; }
        }
        if (pass == 0) p->r1_zero_words = w1;
    }
    hipError_t e;
    if ((e = p->mem.alloc((void**)&p->d_stats, sizeof(uint32_t) * std::max<int64_t>(1, 4 * words))) != hipSuccess) return fail_alloc(e);
    uint32_t* r2_base = p->d_stats;
    uint32_t* r1_base = p->d_stats + 2 * words;

    timer.tick("arenas");
    // ---- per-relation device descriptors ----
    std::vector<LeRelDev> h(n_relations);
    std::vector<LeLayerDiff> ld(n_layers);
    for (int l = 0; l < n_layers; ++l) {
        ld[l].partial_begin = -1; ld[l].n_partials = 0;
        ld[l].n_elems = (double)((int64_t)layers[l].out_ch * layers[l].in_per_group * layers[l].khkw);
    }
    // (a single network large enough to stream -- ResNet-18: 11 M paired elements, 2 717 tiles of 4 096 -- has workgroups to spare
    // and takes the batch's tile: its pass 0.166 -> 0.161 ms, two alternating rounds of tools/lat.py)
    int64_t paired_elems = 0;
    for (int r = 0; r < n_relations; ++r) {
        const dfq_layer& A = layers[relations[r].first];
        const dfq_layer& B = layers[relations[r].second];
        paired_elems += (int64_t)A.out_ch * A.in_per_group * A.khkw + (int64_t)B.out_ch * B.in_per_group * B.khkw;
    }
    const int target = tile_target(n_nets > 1 || paired_elems >= (int64_t)6 << 20);
    for (int r = 0; r < n_relations; ++r) {
        const dfq_relation& rr = relations[r];
        const dfq_layer& A = layers[rr.first];
        const dfq_layer& B = layers[rr.second];
        LeRelDev& d = h[r];
        d = LeRelDev();
        d.w1 = A.weight; d.w2 = B.weight; d.b1 = A.bias; d.bnw = rr.bn_weight; d.bnb = rr.bn_bias; d.s_cum = rr.scale_cum;
        d.o1 = A.out_ch; d.row_len = A.in_per_group * A.khkw; d.khkw1 = A.khkw;
        d.o2 = B.out_ch; d.i2g = B.in_per_group; d.khkw = B.khkw;
        const int G = (d.o1 != d.i2g) ? (d.o1 / d.i2g) : 1;
        d.gi = d.o1 / G; d.go = B.out_ch / G;
        d.r1 = r1_base + r1_off[r];
        d.r2 = r2_base + r2_off[r];
        d.stat_stride = words;
        // interior layers: W1 that was the second layer of an earlier relation gets both rescales in this
        // relation's row pass; W2 that is the first layer of a later relation is only measured here
        d.w1_interior = as_second[rr.first] >= 0 ? 1 : 0;
        d.w2_interior = as_first[rr.second] >= 0 ? 1 : 0;
        const int row_len2 = d.i2g * d.khkw;
        d.rt_vec = (d.row_len % 4 == 0 && ((uintptr_t)A.weight & 15u) == 0) ? 4 : 1;
        d.ct_vec = (row_len2 % 4 == 0 && ((uintptr_t)B.weight & 15u) == 0) ? 4 : 1;
        const bool emits_cols = as_second[rr.first] >= 0;
        tile_shape(d.o1, d.row_len, d.rt_vec,
                   emits_cols ? emit_cols_max(d.rt_vec) : plain_row_cols_max(d.rt_vec), false, target,
                   &d.rt_rows, &d.rt_cols, &d.rt_slabs);
        tile_shape(d.o2, row_len2, d.ct_vec, col_cols_max(d.ct_vec, n_nets > 1), true, target,
                   &d.ct_rows, &d.ct_cols, &d.ct_slabs);
        const bool allow_short = getenv("DFQ_LE_NO_SHORT") == nullptr;
        if (allow_short && d.i2g == 1 && d.khkw <= 32 && d.khkw != 1) {     // depthwise second layer: one thread per row
            d.ct_vec = 0; d.ct_rows = kBlock; d.ct_cols = row_len2; d.ct_slabs = 1;
        }
        // 1/s table of a col tile: (#groups spanned by its rows) x (#input channels spanned by its columns)
        const int nci2 = ceil_div(d.ct_cols, d.khkw) + 1;
        while (d.ct_rows > 1 && (ceil_div(d.ct_rows, d.go) + 1) * nci2 > kSlotMax) d.ct_rows = (d.ct_rows + 1) / 2;
        if ((ceil_div(d.ct_rows, d.go) + 1) * nci2 > kSlotMax)
            return (dfq_le_plan_destroy(p), fail_arg("dfq_le_plan_create: relation %d: kernel size too small for its width", r));
        {
            // a workgroup of the bootstrap launch streams ~kBootWork elements at most: a block of 64 channels of a 3x3 layer
            // with 512 of them is 295 000 elements per pass -- 30 workgroups, 227 us for the whole of ResNet-18 before the split
            const int nc = std::min(kBootTc, d.o1);
            const int64_t rows_work = (as_second[rr.first] < 0) ? (int64_t)nc * d.row_len : 0;
            const int64_t cols_work = (int64_t)nc * d.khkw * d.go;
            const char* be = getenv("DFQ_LE_BOOT_WORK");          // tests: split small layers too
            // (a single network is bound by the number of workgroups in flight: half the span -- ResNet-18's restart + two
            // sweeps 0.119 -> 0.105 ms)
            const int64_t unit = (be && atoi(be) > 0) ? atoi(be) : (n_nets == 1 ? kBootWork / 2 : kBootWork);
            const int64_t want = (std::max(rows_work, cols_work) + unit - 1) / unit;
            d.boot_split = (int)std::max<int64_t>(1, std::min<int64_t>(want, std::min<int64_t>(64, std::max(1, be ? d.go : d.go / 8))));
        }
        d.boot_tiles = ceil_div(d.o1, kBootTc) * d.boot_split;
        d.net = net_of(rr.first);
    }
    // ---- free-running segments (dfq_le_cf.hpp): a chain start, relations linked through depthwise layers, a chain end ----
    std::vector<int> fr(n_relations, 0);            // 1 + position in its segment
    std::vector<int> fr_last(n_relations, 0);       // last relation of its segment (its second layer is the chain's end)
    std::vector<std::vector<int>> segments;
    p->cf_group = cf_group_from_env(n_nets > 1, paired_elems >= (int64_t)6 << 20);
    {
        const bool no_short = getenv("DFQ_LE_NO_SHORT") != nullptr;
        // first layer handled by one thread per row (the rule of the second pass below)
        auto short_first = [&](int r, int j_prev) {
            const LeRelDev& d = h[r];
            return !no_short && d.khkw1 == d.row_len && d.row_len <= 32 && d.row_len != 1 && (j_prev < 0 || h[j_prev].go == 1);
        };
        for (int r0 = 0; r0 < n_relations && p->cf_group > 1; ++r0) {
            if (as_second[relations[r0].first] >= 0) continue;              // not a chain start
            std::vector<int> seg{r0};
            bool ok = true;
            for (int cur = r0;;) {
                const int nx = as_first[relations[cur].second];
                if (nx < 0) break;                                          // the chain ends here: every statistic is closed-form
                // the layer in between must be scaled uniformly along BOTH statistics' axes: one thread-per-row channel per paired channel
                const bool diag = short_first(nx, cur) && h[cur].go == 1 && h[cur].gi == 1 && h[nx].o1 == h[cur].o1 && (int)seg.size() < kCfMaxRel;
                if (!diag) { ok = false; break; }
                seg.push_back(nx);
                cur = nx;
            }
            if (!ok) continue;
            // the last layer's column tiles keep one table entry per (group, input channel) they span and sweep
            LeRelDev& last = h[seg.back()];
            // (an UNGROUPED layer -- every row in the one group -- needs no table at all: lean_col keeps a thread's factors in
            // registers.  Until late in round 6 the bound was applied to those too and refused every chain end whose column tile
            // is wider than 127 channels: a single MobileNetV2 on the streaming engine -- the sharded pass, the repeat of an
            // abandoned resident launch -- had no free-running segment at all.)
            if (last.ct_vec != 0 && last.go < last.o2) {
                const int nci2 = last.khkw == 1 ? last.ct_cols : ceil_div(last.ct_cols, last.khkw) + 1;
                while (last.ct_rows > 1 && (ceil_div(last.ct_rows, last.go) + 1) * nci2 > kCfTab) last.ct_rows = (last.ct_rows + 1) / 2;
                if ((ceil_div(last.ct_rows, last.go) + 1) * nci2 > kCfTab) continue;
            }
            for (size_t i = 0; i < seg.size(); ++i) fr[seg[i]] = 1 + (int)i;
            fr_last[seg.back()] = 1;
            segments.push_back(seg);
        }
        if (segments.empty()) p->cf_group = 1;
        // background mode of the lean launches (dfq_le_cf.hpp): OPT-IN, DFQ_LE_CF_BG=1 -- bit-identical and measured no faster
        // (profiles/r06_experiments.txt 9); not with recorded graphs (a second stream inside a capture) and not with the woven build
        const char* be = getenv("DFQ_LE_CF_BG");
        const char* ge = getenv("DFQ_GRAPH");
        p->cf_bg = p->cf_group > 1 && be && be[0] == '1' && !(ge && ge[0] == '1') && !kWeave;
    }
    // producer links + slot limits need every relation's geometry, so a second pass
    int tile_slot = 0;
    // Thread-per-row W1 that is interior (a depthwise layer inside a chain): its threads take the row range of t = fl(w / s_prev)
    // themselves (LeRelDev::local_r1), so the read-only pass of the previous relation over this layer is not launched at all --
    // that relation keeps its row tiles only (DFQ_LE_LOCAL_R1=0 keeps the pass).
    std::vector<int> skip_cols(n_relations, 0), local_r1(n_relations, 0);
    {
        const bool on = !(getenv("DFQ_LE_LOCAL_R1") && getenv("DFQ_LE_LOCAL_R1")[0] == '0') && getenv("DFQ_LE_NO_SHORT") == nullptr;
        const char* lre = getenv("DFQ_LE_LOCAL_ROW");
        const int local_row_max = lre ? atoi(lre) : 0;
        const char* fe = getenv("DFQ_LE_FUSE");
        const char* pe0 = getenv("DFQ_LE_PERSIST");           // (persistent workgroups walk the table one tile at a time: no tile may wait for a later one)
        const bool fuse_slabs = (fe && fe[0] == '1') && !(pe0 && pe0[0] == '1');
        for (int r = 0; r < n_relations && on; ++r) {
            const LeRelDev& d = h[r];
            const int j_prev = as_second[relations[r].first];
            if (fr[r]) continue;                      // free-running: its tiles are le_lean_kernel's
            if (j_prev < 0 || as_first[relations[j_prev].second] != r) continue;
            const bool short_rows = d.khkw1 == d.row_len && d.row_len <= kShortChunk && d.row_len != 1 && h[j_prev].go == 1;
            if (short_rows && h[j_prev].ct_vec == 0) { local_r1[r] = 1; skip_cols[j_prev] = 1; continue; }
            // ... and, OPT-IN (DFQ_LE_LOCAL_ROW = longest row in floats, e.g. 512; default 0 = off), a layer of 16-byte-vector rows
            // tiled in FULL rows: row_tile's local mode.  Built for VERDICT round 3 item 4 and measured at batch 32 (MobileNetV2:
            // the 1280 x 320 layer in [24 x 320] tiles; with 1024 also the 320 x 960 layer in [8 x 960] tiles): 7 % fewer bytes per
            // sweep, but 162-164 us per forced sweep against 159-160 us without it -- a full-row tile solves five times as many
            // column scales and issues five times as many column-statistics atomics as a [128 x 64] one, and this kernel is bound
            // by instruction issue and workgroup residency as much as by bytes.  Its tile shape is fixed below.
            const bool short_kind = d.khkw1 == d.row_len && d.row_len <= 32 && d.row_len != 1 && h[j_prev].go == 1;   // (thread-per-row, > 9 taps)
            if (!short_kind && d.rt_vec == 4 && d.row_len <= local_row_max && d.row_len / 4 <= kBlock) {
                const int nci = ceil_div(d.row_len, d.khkw1) + 1;
                const int rows = local_rows(d.o1, d.row_len, target);
                if ((ceil_div(rows, h[j_prev].go) + 1) * nci <= kSlotMax) { local_r1[r] = 2; skip_cols[j_prev] = 1; }
            }
            // ... and the same without full rows (round 6; OPT-IN, DFQ_LE_FUSE=1): the tiles keep their tall narrow shape -- few
            // column-statistics atomics per word -- and merge their rows' ranges over the slabs of the row block INSIDE the launch
            // (kLocalSlabs, row_tile): 12 -> 8 B per element of a layer scaled along both axes and no read-only workgroups, at the
            // price of a second in-launch wait with the tile's elements in registers.  Bit-identical (test_deferred_stores_are_
            // invisible[...streaming-fused], the GPU suite) and measured NO FASTER at batch 32 (tools/gpu_r06_fuse.sh, two alternating
            // rounds: 1.725 / 1.723e10 weights/s against 1.732 / 1.730e10, 416 instead of 508 MB per launch in the same 99 us): a slab
            // tile lives 26 us where the read-only pass and the rescaling pass lived 11 + 10 -- it holds its slot and its registers
            // while the relation's other row tiles trickle in, and the layer behind it waits for all of them in turn; the launch is
            // bound by that chain and by workgroup residency, not by the bytes saved (profiles/r06_level_kernel_by_kind_fused.txt).
            if (local_r1[r] == 0 && fuse_slabs && !short_kind && d.rt_vec == 4) { local_r1[r] = kLocalSlabs; skip_cols[j_prev] = 1; p->has_slab_tiles = true; }
        }
    }
    for (int r = 0; r < n_relations; ++r) {
        const dfq_relation& rr = relations[r];
        LeRelDev& d = h[r];
        const int j_prev = as_second[rr.first];     // relation whose second layer is our W1
        const int j_next = as_first[rr.second];     // relation whose first layer is our W2
        d.prev_r1 = (j_prev >= 0) ? h[j_prev].r1 : nullptr;
        if (j_prev >= 0) {
            d.out_cols = h[j_prev].r2;
            d.pc_go = h[j_prev].go; d.pc_gi = h[j_prev].gi;
            // LDS slots of a row tile: (#groups spanned) * (#channels spanned)
            const int nci = ceil_div(d.rt_cols, d.khkw1) + 1;
            while (d.rt_rows > 1 && (ceil_div(d.rt_rows, d.pc_go) + 1) * nci > kSlotMax) d.rt_rows = (d.rt_rows + 1) / 2;
            if ((ceil_div(d.rt_rows, d.pc_go) + 1) * nci > kSlotMax)
                return (dfq_le_plan_destroy(p), fail_arg("dfq_le_plan_create: relation %d: row tile does not fit", r));
        } else {
            d.out_cols = nullptr; d.pc_go = 1; d.pc_gi = 1;
        }
        d.out_rows = (j_next >= 0) ? h[j_next].r1 : nullptr;
        // depthwise-like first layer (one input channel per row, k x k kernel): one thread per row, provided
        // every row is its own channel of the stat consumer
        if (getenv("DFQ_LE_NO_SHORT") == nullptr && d.khkw1 == d.row_len && d.row_len <= 32 && d.row_len != 1 && (j_prev < 0 || d.pc_go == 1)) {
            d.rt_vec = 0; d.rt_rows = kBlock; d.rt_cols = d.row_len; d.rt_slabs = 1;
        }
        if (local_r1[r] == 2) {                       // full rows: row_len / 4 lanes per row, as many rows as the register slots hold
            d.rt_cols = d.row_len; d.rt_slabs = 1;
            d.rt_rows = local_rows(d.o1, d.row_len, target);
        }
        d.n_row_tiles = ceil_div(d.o1, d.rt_rows) * d.rt_slabs;
        d.n_col_tiles = (skip_cols[r] || (fr[r] && !fr_last[r])) ? 0 : ceil_div(d.o2, d.ct_rows) * d.ct_slabs;   // (no pass over a depthwise layer inside a free-running segment)
        d.local_r1 = local_r1[r] == kLocalSlabs ? kLocalSlabs : (local_r1[r] ? 1 : 0);
        d.partial_base = tile_slot;
        ld[rr.first].partial_begin = d.partial_base;
        ld[rr.first].n_partials = d.n_row_tiles;                       // last touch of W1 this sweep
        if (!d.w2_interior) {
            ld[rr.second].partial_begin = d.partial_base + d.n_row_tiles;
            ld[rr.second].n_partials = d.n_col_tiles;
        }
        tile_slot += d.n_row_tiles + d.n_col_tiles;
        p->paired_total += (int64_t)d.o1 * d.row_len + (int64_t)d.o2 * d.i2g * d.khkw;
    }
    p->total_tiles = tile_slot;
    timer.tick("descriptors");
    // ---- deferred stores: layers scaled one way only, handled by the register-tile functions ----
    {
        // default: 4 for a batched plan (bound by what a sweep moves: 1.22e10 -> 1.35e10 -> 1.40e10 weights/s at depth 1 / 2 / 4
        // for 32 MobileNetV2), 1 for a single network (bound by launch latencies: ResNet-18's two sweeps 0.113 / 0.123 /
        // 0.154 ms -- the launches that bring the weights up to date cost more than the skipped stores save)
        const char* de = getenv("DFQ_LE_DEFER");
        const int want = de ? atoi(de) : (n_nets > 1 ? 4 : 1);
        p->defer = (want >= 4) ? 4 : (want >= 2) ? 2 : 1;
        int64_t hold_floats = 0;
        std::vector<int64_t> hold_off(n_relations, -1);
        for (int r = 0; r < n_relations && p->defer > 1; ++r) {
            LeRelDev& d = h[r];
            if (fr[r]) continue;
            if (!d.w1_interior && d.rt_vec != 0) { d.defer |= 1; p->deferred_total += (int64_t)d.o1 * d.row_len; }
            if (!d.w2_interior && d.ct_vec != 0) { d.defer |= 2; p->deferred_total += (int64_t)d.o2 * d.i2g * d.khkw; }
            if (d.defer) { hold_off[r] = hold_floats; hold_floats += (int64_t)2 * (p->defer - 1) * d.o1; }
        }
        if (hold_floats > 0) {
            if ((e = p->mem.alloc((void**)&p->d_hold, sizeof(float) * hold_floats)) != hipSuccess) return fail_alloc(e);
            for (int r = 0; r < n_relations; ++r) if (hold_off[r] >= 0) h[r].hold = p->d_hold + hold_off[r];
        } else {
            p->defer = 1;
        }
    }
    std::vector<LeNetDesc> nets(n_nets);
    for (int n = 0; n < n_nets; ++n) { nets[n].layer_begin = 0; nets[n].n_layers = 0; nets[n].tile_begin = 0; nets[n].n_tiles = 0; }
    for (int l = n_layers - 1; l >= 0; --l) { nets[net_of(l)].layer_begin = l; nets[net_of(l)].n_layers += 1; }
    for (int r = n_relations - 1; r >= 0; --r) {
        LeNetDesc& nd = nets[h[r].net];
        nd.tile_begin = h[r].partial_base;
        nd.n_tiles += h[r].n_row_tiles + h[r].n_col_tiles;
    }

    {
        bool uni = n_nets > 1 && nets[0].n_layers > 0 && nets[0].n_tiles > 0 && !(getenv("DFQ_LE_UNIFORM") && getenv("DFQ_LE_UNIFORM")[0] == '0');
        for (int n = 0; n < n_nets && uni; ++n)
            uni = nets[n].n_layers == nets[0].n_layers && nets[n].n_tiles == nets[0].n_tiles &&
                  nets[n].layer_begin == n * nets[0].n_layers && nets[n].tile_begin == n * nets[0].n_tiles;
        p->uni_layers = uni ? nets[0].n_layers : 0;
        p->uni_tiles = uni ? nets[0].n_tiles : 0;
    }
    timer.tick("deferred");
    // ---- sort relations by level (stable) and lay out the launches ----
    std::vector<int> order(n_relations);
    for (int r = 0; r < n_relations; ++r) order[r] = r;
    // Inside a level the relations of the longest chains come first: in a one-launch sweep the tiles of the next
    // level wait for exactly those, so the critical chain must not queue behind the independent short ones
    // (MobileNetV2's five-relation chain sits at the END of the relation list).
    std::vector<int> height(n_relations, 0);
    for (int r = n_relations - 1; r >= 0; --r) {
        const int j_next = as_first[relations[r].second];
        height[r] = (j_next >= 0) ? height[j_next] + 1 : 0;
    }
    // ... for ONE network.  A batched plan keeps the list order (network after network inside a level): its launch is bound
    // by throughput, not by one network's critical chain, and sorting by chain height lines up all networks' tiles of one
    // kind -- a phase of nothing but small depthwise tiles fills the workgroup slots while moving few bytes (batch of 32:
    // 188 vs 194 us per launch).  DFQ_LE_CHAIN_FIRST=0/1 overrides.
    const char* cf = getenv("DFQ_LE_CHAIN_FIRST");
    const bool chain_first = cf ? (cf[0] != '0') : (n_nets == 1);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (level[a] != level[b]) return level[a] < level[b];
        return chain_first && height[a] > height[b];
    });
    std::vector<LeRelDev> sorted(n_relations);
    int boot = 0, prev_level = -1;
    for (int i = 0; i < n_relations; ++i) {
        const int r = order[i];
        if (level[r] != prev_level) {                                             // new level
            p->levels.push_back(LevelLaunch());
            p->levels.back().rel_begin = i;
            prev_level = level[r];
        }
        LevelLaunch& L = p->levels.back();
        sorted[i] = h[r];
        sorted[i].boot_begin = boot;              // bootstrap launch walks the same (sorted) table
        boot += h[r].boot_tiles;
        L.n_rels += 1;
        const int64_t n1 = (int64_t)h[r].o1 * h[r].row_len;
        const int64_t n2 = (int64_t)h[r].o2 * h[r].i2g * h[r].khkw;
        if (fr[r]) {                              // free-running: not in the sweep's workgroup table (le_lean_kernel)
            p->fr_total += n1 + (fr_last[r] ? n2 : 0);
            continue;
        }
        L.n_blocks += h[r].n_row_tiles + h[r].n_col_tiles;
        L.rw_elems += n1 + (h[r].w2_interior ? 0 : n2);
        L.ro_elems += (h[r].w2_interior && h[r].n_col_tiles > 0) ? n2 : 0;      // (no read-only pass where the next relation's rows are local)
    }
    // local_r1: the row tiles of such a relation depend on nothing its own level produces and nobody else touches their layer, so
    // they are listed (and counted) with the PREVIOUS relation of the chain, a level earlier -- right where that relation's read-
    // only pass used to be; the relation's column tiles, which wait for them, then find them long finished (listed in the same
    // level they waited for the workgroups just in front of them: 178 instead of 168 us per sweep at batch 32)
    std::vector<int> lvl_index(n_relations, 0);
    {
        int li = -1, pl = -1;
        for (int i = 0; i < n_relations; ++i) {
            if (level[order[i]] != pl) { ++li; pl = level[order[i]]; }
            lvl_index[order[i]] = li;
        }
        for (int r = 0; r < n_relations; ++r) {
            if (!h[r].local_r1) continue;
            const int j_prev = as_second[relations[r].first];
            const int64_t n1 = (int64_t)h[r].o1 * h[r].row_len;
            p->levels[lvl_index[r]].n_blocks -= h[r].n_row_tiles;
            p->levels[lvl_index[r]].rw_elems -= n1;
            p->levels[lvl_index[j_prev]].n_blocks += h[r].n_row_tiles;
            p->levels[lvl_index[j_prev]].rw_elems += n1;
        }
    }
    for (const LevelLaunch& L : p->levels) { p->rw_total += L.rw_elems; p->ro_total += L.ro_elems; }
    p->boot_blocks = boot;
    std::vector<int32_t> boot_map(std::max(1, boot));
    for (int i = 0; i < n_relations; ++i)
        for (int b = 0; b < sorted[i].boot_tiles; ++b) boot_map[sorted[i].boot_begin + b] = i;

    const size_t n_part = (size_t)std::max(1, p->total_tiles) * (kBlock / kWave);
    timer.tick("layout");
    if ((e = p->mem.alloc((void**)&p->d_rels, sizeof(LeRelDev) * std::max(1, n_relations))) != hipSuccess) return fail_alloc(e);
    if ((e = p->mem.alloc((void**)&p->d_layer_diff, sizeof(LeLayerDiff) * n_layers)) != hipSuccess) return fail_alloc(e);
    p->part_stride = (int64_t)n_part;                // one array of partial sums per sweep of a group (dfq_le_cf.hpp)
    const size_t part_ring = p->cf_group > 1 ? (size_t)2 * p->cf_group : 1;
    if ((e = p->mem.alloc((void**)&p->d_partials, sizeof(double) * n_part * part_ring)) != hipSuccess) return fail_alloc(e);
    if ((e = p->mem.alloc((void**)&p->d_layer_mean, sizeof(double) * n_layers)) != hipSuccess) return fail_alloc(e);
    if ((e = p->mem.alloc((void**)&p->d_state, sizeof(LeState) * n_nets)) != hipSuccess) return fail_alloc(e);
    if ((e = p->mem.alloc((void**)&p->d_nets, sizeof(LeNetDesc) * n_nets)) != hipSuccess) return fail_alloc(e);
    if ((e = p->mem.alloc((void**)&p->d_boot_map, sizeof(int32_t) * boot_map.size())) != hipSuccess) return fail_alloc(e);
    if ((e = hipMemcpy(p->d_nets, nets.data(), sizeof(LeNetDesc) * n_nets, hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
    if ((e = hipMemcpy(p->d_boot_map, boot_map.data(), sizeof(int32_t) * boot_map.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
    if ((e = hipMemcpy(p->d_layer_diff, ld.data(), sizeof(LeLayerDiff) * n_layers, hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
    if ((e = hipMemset(p->d_partials, 0, sizeof(double) * n_part * part_ring)) != hipSuccess) return fail_alloc(e);
    if ((e = hipMemset(p->d_state, 0, sizeof(LeState) * n_nets)) != hipSuccess) return fail_alloc(e);
    {
        // dependency links (see le_level_kernel): position of every relation in the level-sorted table
        std::vector<int> pos(n_relations, -1);
        for (int i = 0; i < n_relations; ++i) pos[order[i]] = i;
        for (int i = 0; i < n_relations; ++i) {
            const int r = order[i];
            const int j_prev = as_second[relations[r].first];
            const int j_next = as_first[relations[r].second];
            // the relation whose column tiles produce what this relation's tiles need inside the launch: its predecessor in the
            // chain -- or, where that one's read-only pass is not launched (local_r1), the predecessor's own predecessor, whose
            // column tiles publish the row statistics this relation's 1/s_prev is solved from
            // ... inside the launch: its predecessor in the chain.  Where that one's read-only pass is not launched (this relation
            // is local_r1) what remains to wait for is whoever publishes the predecessor's ROW statistics (this relation's
            // 1/s_prev is solved from them): the predecessor's own row tiles if it is local_r1 too, else the column tiles of the
            // relation before it.
            int j_dep = j_prev, dep_counter = -1, dep_tiles = 0;
            if (j_dep >= 0 && h[j_dep].n_col_tiles > 0) {
                dep_counter = pos[j_dep]; dep_tiles = h[j_dep].n_col_tiles;
            } else if (j_dep >= 0 && h[j_dep].local_r1) {
                dep_counter = n_relations + 1 + pos[j_dep]; dep_tiles = h[j_dep].n_row_tiles;
            } else if (j_dep >= 0) {
                j_dep = as_second[relations[j_dep].first];
                if (j_dep >= 0) { dep_counter = pos[j_dep]; dep_tiles = h[j_dep].n_col_tiles; }
            }
            sorted[i].dep_idx = dep_counter;
            sorted[i].dep_tiles = dep_tiles;
            // local_r1: the column tiles of the relation wait for its own row tiles (their counter sits behind the error word)
            sorted[i].cdep_idx = -1; sorted[i].cdep_tiles = 0; sorted[i].rcounter_idx = -1;
            if (h[r].local_r1) {
                sorted[i].rcounter_idx = n_relations + 1 + i;
                sorted[i].cdep_idx = n_relations + 1 + i;
                sorted[i].cdep_tiles = h[r].n_row_tiles;
            }
            sorted[i].counter_idx = (j_next >= 0) ? i : -1;
            if (j_dep >= 0 && pos[j_dep] >= i)
                return (dfq_le_plan_destroy(p), fail_arg("dfq_le_plan_create: relation %d precedes the relation it depends on", r));
        }
        if (n_relations > 0 &&
            (e = hipMemcpy(p->d_rels, sorted.data(), sizeof(LeRelDev) * n_relations, hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
        // workgroup table of a sweep: the tiles of a level in relation order (a relation's tiles stay adjacent, so do
        // its rows in memory), level after level
        std::vector<LeBlockRef> blocks;
        for (LevelLaunch& L : p->levels) {
            L.block_begin = (int)blocks.size();
            for (int i = 0; i < L.n_rels; ++i) {
                const LeRelDev& d = sorted[L.rel_begin + i];
                if (fr[order[L.rel_begin + i]]) continue;
                for (int t = d.local_r1 ? d.n_row_tiles : 0; t < d.n_row_tiles + d.n_col_tiles; ++t) blocks.push_back(LeBlockRef{L.rel_begin + i, t, d.net, 0});
                // ... followed by the row tiles of the next relation of the chain where those take their rows' statistics themselves
                const int j_next = as_first[relations[order[L.rel_begin + i]].second];
                if (j_next >= 0 && h[j_next].local_r1) {
                    const int q = pos[j_next];
                    for (int t = 0; t < sorted[q].n_row_tiles; ++t) blocks.push_back(LeBlockRef{q, t, sorted[q].net, 0});
                }
            }
            if ((int)blocks.size() - L.block_begin != L.n_blocks)
                return (dfq_le_plan_destroy(p), fail_arg("dfq_le_plan_create: internal: level of %d workgroups listed as %d", L.n_blocks, (int)blocks.size() - L.block_begin));
        }
        if ((e = p->mem.alloc((void**)&p->d_blocks, sizeof(LeBlockRef) * std::max<size_t>(1, blocks.size()))) != hipSuccess) return fail_alloc(e);
        if (!blocks.empty() &&
            (e = hipMemcpy(p->d_blocks, blocks.data(), sizeof(LeBlockRef) * blocks.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
        p->h_rels = sorted;
        p->h_blocks = blocks;
        {
            std::vector<LeFlushRef> refs;
            std::vector<int32_t> hold_rels;
            auto add_spans = [&](float* base, int64_t n, int64_t rl, const float* hold, int side, const LeRelDev& d, int is_fr,
                                 const float* hold2, int o1_2) {
                for (int64_t f = 0; f < n; f += kFlushSpan) {
                    LeFlushRef fr_;
                    memset(&fr_, 0, sizeof(fr_));
                    fr_.w = base + f;
                    fr_.hold = hold;
                    fr_.span = (int32_t)std::min<int64_t>(kFlushSpan, n - f);
                    fr_.row_len = (int32_t)rl;
                    fr_.row0 = (int32_t)(f / rl); fr_.pos0 = (int32_t)(f % rl);
                    fr_.side = side; fr_.net = d.net;
                    fr_.go = d.go; fr_.gi = d.gi; fr_.khkw = d.khkw; fr_.o1 = d.o1;
                    fr_.vec = ((rl & 3) == 0 && (((uintptr_t)fr_.w) & 15) == 0) ? 1 : 0;
                    fr_.fr = is_fr;
                    fr_.hold2 = hold2; fr_.o1_2 = o1_2; fr_.pc_gi = d.pc_gi;
                    refs.push_back(fr_);
                }
            };
            for (int i = 0; i < n_relations && p->defer > 1; ++i) {
                const LeRelDev& d = sorted[i];
                if (!d.defer) continue;
                hold_rels.push_back(i);
                for (int side = 0; side < 2; ++side) {
                    if (!(d.defer & (1 << side))) continue;
                    const int64_t n = side == 0 ? (int64_t)d.o1 * d.row_len : (int64_t)d.o2 * d.i2g * d.khkw;
                    const int64_t rl = side == 0 ? d.row_len : (int64_t)d.i2g * d.khkw;
                    add_spans(side == 0 ? d.w1 : d.w2, n, rl, d.hold + (int64_t)side * d.o1, side, d, 0, nullptr, 0);
                }
            }
            // ---- free-running segments: factor rings + recurrence state, the solver's tables, the lean tiles, their flush spans ----
            if (p->cf_group > 1) {
                const int G = p->cf_group;
                std::vector<int> cf_index(n_relations, -1);          // original relation -> LeCfRel
                std::vector<LeCfRel> cf_rels;
                std::vector<LeCfSeg> cf_segs;
                std::vector<int32_t> cf_map;
                int64_t ring_floats = 0, state_floats = 0;
                for (const auto& seg : segments)
                    for (int r : seg) { ring_floats += (int64_t)8 * G * h[r].o1; state_floats += (int64_t)4 * h[r].o1; }   // (ring: 4G sweeps x {s, 1/s})
                float* d_ring = nullptr;
                float* d_state = nullptr;
                if ((e = p->mem.alloc((void**)&d_ring, sizeof(float) * ring_floats)) != hipSuccess) return fail_alloc(e);
                if ((e = p->mem.alloc((void**)&d_state, sizeof(float) * state_floats)) != hipSuccess) return fail_alloc(e);
                p->d_cf_ring = d_ring; p->cf_ring_floats = ring_floats;
                int64_t ring_off = 0, state_off = 0;
                for (const auto& seg : segments) {
                    LeCfSeg S;
                    memset(&S, 0, sizeof(S));
                    S.n_rel = (int)seg.size(); S.o1 = h[seg[0]].o1; S.net = h[seg[0]].net;
                    for (size_t i = 0; i < seg.size(); ++i) {
                        const int r = seg[i];
                        const LeRelDev& d = sorted[pos[r]];
                        LeCfRel c;
                        memset(&c, 0, sizeof(c));
                        c.ring = d_ring + ring_off; ring_off += (int64_t)8 * G * d.o1;
                        c.state = d_state + state_off; state_off += (int64_t)4 * d.o1;
                        c.boot_r1 = i == 0 ? d.r1 : nullptr;             // (parity 0: what the bootstrap launch writes)
                        c.boot_r2 = d.r2;
                        c.s_cum = d.s_cum; c.bnw = d.bnw; c.bnb = d.bnb; c.b1 = d.b1;
                        c.o1 = d.o1; c.net = d.net;
                        cf_index[r] = (int)cf_rels.size();
                        S.rel[i] = (int)cf_rels.size();
                        cf_rels.push_back(c);
                    }
                    for (int ch = 0; ch < ceil_div(S.o1, kCtlBlock); ++ch) { cf_map.push_back((int32_t)cf_segs.size()); cf_map.push_back(ch); }
                    cf_segs.push_back(S);
                }
                // the lean tiles: the same shapes the general tiles of these layers would have had, listed network after network
                std::vector<LeLeanRef> lean;
                auto lean_base = [&](const LeRelDev& d, int r) {
                    LeLeanRef t;
                    memset(&t, 0, sizeof(t));
                    t.ring = cf_rels[cf_index[r]].ring;
                    t.net = d.net; t.o1 = d.o1; t.go = d.go; t.gi = d.gi; t.khkw = d.khkw; t.pc_gi = 1; t.o1_prev = 1;
                    return t;
                };
                for (int i = 0; i < n_relations; ++i) {
                    const int r = order[i];
                    if (!fr[r]) continue;
                    const LeRelDev& d = sorted[i];
                    const int j_prev = as_second[relations[r].first];
                    // first layer: rows * s
                    for (int tile = 0; tile < d.n_row_tiles; ++tile) {
                        LeLeanRef t = lean_base(d, r);
                        t.slot = d.partial_base + tile;
                        if (d.rt_vec == 0) {
                            t.kind = kLeanShort0;
                            t.r0 = tile * kBlock; t.nr = std::min(kBlock, d.o1 - t.r0); t.np = d.row_len; t.stride = d.row_len; t.p0 = 0;
                            t.w = d.w1 + (int64_t)t.r0 * d.row_len;
                            if (j_prev >= 0) { t.ring_prev = cf_rels[cf_index[j_prev]].ring; t.o1_prev = h[j_prev].o1; t.pc_gi = d.pc_gi; }
                            t.s_cum = d.s_cum; t.bnw = d.bnw; t.bnb = d.bnb; t.b1 = d.b1;
                        } else {
                            const int rblk = tile / d.rt_slabs, slab = tile - rblk * d.rt_slabs;
                            t.kind = d.rt_vec == 4 ? kLeanRow4 : kLeanRow1;
                            t.r0 = rblk * d.rt_rows; t.nr = std::min(d.rt_rows, d.o1 - t.r0);
                            t.p0 = slab * d.rt_cols; t.np = std::min(d.rt_cols, d.row_len - t.p0); t.stride = d.row_len;
                            t.w = d.w1 + ((int64_t)t.r0 * d.row_len + t.p0);
                            if (slab == 0) { t.s_cum = d.s_cum; t.bnw = d.bnw; t.bnb = d.bnb; t.b1 = d.b1; }
                        }
                        lean.push_back(t);
                        p->lean_info.push_back(t.kind); p->lean_info.push_back(t.nr); p->lean_info.push_back(t.np);
                    }
                    // the chain's last layer: columns * 1/s
                    const int row_len2 = d.i2g * d.khkw;
                    for (int tile = 0; tile < d.n_col_tiles; ++tile) {
                        LeLeanRef t = lean_base(d, r);
                        t.slot = d.partial_base + d.n_row_tiles + tile;
                        if (d.ct_vec == 0) {
                            t.kind = kLeanShort1;
                            t.r0 = tile * kBlock; t.nr = std::min(kBlock, d.o2 - t.r0); t.np = row_len2; t.stride = row_len2; t.p0 = 0;
                            t.w = d.w2 + (int64_t)t.r0 * row_len2;
                        } else {
                            const int rblk = tile / d.ct_slabs, slab = tile - rblk * d.ct_slabs;
                            t.kind = d.ct_vec == 4 ? kLeanCol4 : kLeanCol1;
                            t.r0 = rblk * d.ct_rows; t.nr = std::min(d.ct_rows, d.o2 - t.r0);
                            t.p0 = slab * d.ct_cols; t.np = std::min(d.ct_cols, row_len2 - t.p0); t.stride = row_len2;
                            t.w = d.w2 + ((int64_t)t.r0 * row_len2 + t.p0);
                        }
                        lean.push_back(t);
                        p->lean_info.push_back(t.kind); p->lean_info.push_back(t.nr); p->lean_info.push_back(t.np);
                    }
                    // what a loop that stops inside a group leaves pending
                    const float* ring = cf_rels[cf_index[r]].ring;
                    if (j_prev >= 0)
                        add_spans(d.w1, (int64_t)d.o1 * d.row_len, d.row_len, ring, 2, d, 1, cf_rels[cf_index[j_prev]].ring + h[j_prev].o1, h[j_prev].o1);
                    else
                        add_spans(d.w1, (int64_t)d.o1 * d.row_len, d.row_len, ring, 0, d, 1, nullptr, 0);
                    if (fr_last[r]) add_spans(d.w2, (int64_t)d.o2 * row_len2, row_len2, ring + d.o1, 1, d, 1, nullptr, 0);
                }
                p->n_cf_rels = (int)cf_rels.size(); p->n_cf_segs = (int)cf_segs.size(); p->n_cf_blocks = (int)cf_map.size() / 2;
                p->n_lean = (int)lean.size();
                if ((e = p->mem.alloc((void**)&p->d_cf_rels, sizeof(LeCfRel) * cf_rels.size())) != hipSuccess) return fail_alloc(e);
                if ((e = p->mem.alloc((void**)&p->d_cf_segs, sizeof(LeCfSeg) * cf_segs.size())) != hipSuccess) return fail_alloc(e);
                if ((e = p->mem.alloc((void**)&p->d_cf_map, sizeof(int32_t) * cf_map.size())) != hipSuccess) return fail_alloc(e);
                if ((e = p->mem.alloc((void**)&p->d_lean, sizeof(LeLeanRef) * lean.size())) != hipSuccess) return fail_alloc(e);
                if ((e = hipMemcpy(p->d_cf_rels, cf_rels.data(), sizeof(LeCfRel) * cf_rels.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
                if ((e = hipMemcpy(p->d_cf_segs, cf_segs.data(), sizeof(LeCfSeg) * cf_segs.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
                if ((e = hipMemcpy(p->d_cf_map, cf_map.data(), sizeof(int32_t) * cf_map.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
                if ((e = hipMemcpy(p->d_lean, lean.data(), sizeof(LeLeanRef) * lean.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
                hipLaunchKernelGGL(le_cf_ring_reset_kernel, dim3(p->n_cf_rels), dim3(kBlock), 0, nullptr, (const LeCfRel*)p->d_cf_rels, G);
                // the lean tiles woven evenly into the sweep's workgroup table (they wait for nobody and nobody waits for them, so
                // the order among the general tiles -- which only ever wait for lower indices -- is kept); DFQ_LE_CF_WEAVE=0: own launch
                const char* we = getenv("DFQ_LE_CF_WEAVE");
                if (kWeave && G == kWeaveGroup && !(we && we[0] == '0') && !lean.empty()) {
                    std::vector<LeBlockRef> woven;
                    woven.reserve(blocks.size() + lean.size());
                    const size_t n_g = blocks.size(), n_l = lean.size();
                    size_t gi = 0, li = 0;
                    while (gi < n_g || li < n_l) {
                        // next lean tile whenever the lean share of what has been listed falls behind its share of the whole
                        if (li < n_l && (gi >= n_g || li * (n_g + n_l) <= (gi + li) * n_l)) {
                            woven.push_back(LeBlockRef{-1, (int32_t)li, lean[li].net, 0});
                            ++li;
                        } else {
                            woven.push_back(blocks[gi++]);
                        }
                    }
                    p->n_woven = (int)woven.size();
                    if ((e = p->mem.alloc((void**)&p->d_blocks_woven, sizeof(LeBlockRef) * woven.size())) != hipSuccess) return fail_alloc(e);
                    if ((e = hipMemcpy(p->d_blocks_woven, woven.data(), sizeof(LeBlockRef) * woven.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
                }
            }
            p->n_flush = (int)refs.size();
            p->n_hold_rels = (int)hold_rels.size();
            if (!refs.empty()) {
                if ((e = p->mem.alloc((void**)&p->d_flush, sizeof(LeFlushRef) * refs.size())) != hipSuccess) return fail_alloc(e);
                if ((e = hipMemcpy(p->d_flush, refs.data(), sizeof(LeFlushRef) * refs.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
            }
            if (!hold_rels.empty()) {
                if ((e = p->mem.alloc((void**)&p->d_hold_rels, sizeof(int32_t) * hold_rels.size())) != hipSuccess) return fail_alloc(e);
                if ((e = hipMemcpy(p->d_hold_rels, hold_rels.data(), sizeof(int32_t) * hold_rels.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
                hipLaunchKernelGGL(le_hold_reset_kernel, dim3(p->n_hold_rels), dim3(kBlock), 0, nullptr, (const LeRelDev*)p->d_rels,
                                   (const int32_t*)p->d_hold_rels, (const LeState*)p->d_state, p->defer, 1);
            }
        }
        // counters: [relation: its column tiles][error word + padding][relation: its row tiles (local_r1)]
        if ((e = p->mem.alloc((void**)&p->d_dep, sizeof(unsigned long long) * ((size_t)(2 * n_relations + 1) * kDepStride + 1))) != hipSuccess) return fail_alloc(e);
        if ((e = hipMemset(p->d_dep, 0, sizeof(unsigned long long) * ((size_t)(2 * n_relations + 1) * kDepStride + 1))) != hipSuccess) return fail_alloc(e);
        const char* me = getenv("DFQ_LE_MERGED");
        p->merged = !(me && me[0] == '0');
        const char* pe = getenv("DFQ_LE_PERSIST");
        if (p->merged && pe && pe[0] == '1' && n_nets <= kSweepMaxNets && !blocks.empty()) {
            std::vector<LeTileRef> refs(blocks.size());
            for (size_t i = 0; i < blocks.size(); ++i) {
                const LeRelDev& d = sorted[blocks[i].rel];
                const bool col = blocks[i].tile >= d.n_row_tiles;
                const int tile = col ? blocks[i].tile - d.n_row_tiles : blocks[i].tile;
                const int vec = col ? d.ct_vec : d.rt_vec;
                const int n_rows = col ? d.o2 : d.o1, row_len = col ? d.i2g * d.khkw : d.row_len;
                const int t_rows = col ? d.ct_rows : d.rt_rows, t_cols = col ? d.ct_cols : d.rt_cols, slabs = col ? d.ct_slabs : d.rt_slabs;
                const int rblk = tile / slabs, slab = tile - rblk * slabs;
                const int r0 = rblk * t_rows, p0 = slab * t_cols;
                LeTileRef& t = refs[i];
                memset(&t, 0, sizeof(t));
                t.w = (col ? d.w2 : d.w1) + ((int64_t)r0 * row_len + p0);
                t.stride = row_len;
                t.nr = std::min(t_rows, n_rows - r0);
                t.npv = vec == 4 ? std::min(t_cols, row_len - p0) / 4 : 0;
                t.lanes = std::max(1, t.npv);
                if (col) { int g = 1; while (g < t.npv) g <<= 1; t.lanes = g; }
                t.rel = blocks[i].rel; t.tile = blocks[i].tile; t.net = blocks[i].net;
                t.dep_idx = (col && d.cdep_idx >= 0) ? d.cdep_idx : d.dep_idx;
                t.dep_tiles = (col && d.cdep_idx >= 0) ? d.cdep_tiles : d.dep_tiles;
            }
            if ((e = p->mem.alloc((void**)&p->d_tiles, sizeof(LeTileRef) * refs.size())) != hipSuccess) return fail_alloc(e);
            if ((e = hipMemcpy(p->d_tiles, refs.data(), sizeof(LeTileRef) * refs.size(), hipMemcpyHostToDevice)) != hipSuccess) return fail_alloc(e);
            // every workgroup must be resident at once (a tile may wait for a tile another workgroup holds)
            int dev = 0, occ = 0;
            hipDeviceProp_t prop;
            if ((e = hipGetDevice(&dev)) != hipSuccess || (e = hipGetDeviceProperties(&prop, dev)) != hipSuccess) return fail_alloc(e);
            if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)le_sweep_kernel, kBlock, 0)) != hipSuccess) return fail_alloc(e);
            // ... with a quarter of the chip left free: at exactly the occupancy limit the launch never became fully
            // resident while a second stream kept the chip busy (measured: 1024 of 1024 slots gave up, 768 ran)
            int64_t cap = std::max<int64_t>(1, (int64_t)std::max(1, occ) * std::max(1, prop.multiProcessorCount) * 3 / 4);
#ifdef DFQ_EMU
            cap = 24;                      // the CPU emulation keeps every workgroup of such a launch alive as fibers
#endif
            const char* we = getenv("DFQ_LE_SWEEP_WGS");
            if (we && atoi(we) > 0) cap = std::min<int64_t>(cap, atoi(we));
            p->sweep_grid = (int)std::min<int64_t>(cap, (int64_t)blocks.size());
        }
    }
    timer.tick("uploads");
    if (n_nets == 1 && n_relations > 0) p->resident = le_resident_create(layers, n_layers, relations, n_relations, &p->resident_why);
    else p->resident_why = n_nets > 1 ? "batched plan" : "no relations";
    timer.tick("resident");
    if ((e = hipDeviceSynchronize()) != hipSuccess) return fail_alloc(e);
    timer.tick("sync");
    if (timer.on) fprintf(stderr, "[dfq] le plan: %d networks, %d relations, stat words %lld (x4 parities/kinds), r1 zero words %lld, partials %d\n",
                          n_nets, n_relations, (long long)p->stat_words, (long long)p->r1_zero_words, (int)n_part);
    *out_plan = p;
    return DFQ_OK;
}

// workgroups (= LDS-resident tiles) of the persistent whole-loop launch, 0 when the plan streams
static inline bool res_on(const dfq_le_plan* p) { return p && p->resident && !p->resident_off; }
int32_t dfq_le_plan_resident_tiles(const dfq_le_plan* p) { return res_on(p) ? le_resident_tiles(p->resident) : 0; }
int32_t dfq_le_plan_degraded(const dfq_le_plan* p) { return p ? p->degraded : 0; }
int32_t dfq_le_plan_uniform(const dfq_le_plan* p) { return (p && p->uni_layers > 0) ? 1 : 0; }
int32_t dfq_le_plan_has_waits(const dfq_le_plan* p) {
    return (p && !res_on(p) && p->n_rels > 0 && !p->h_blocks.empty() && ((p->merged && p->levels.size() > 1) || p->has_slab_tiles)) ? 1 : 0;
}
int dfq_le_plan_set_safe_mode(dfq_le_plan* p) {
    if (!p) return fail_arg("dfq_le_plan_set_safe_mode: null plan");
    if (p->has_slab_tiles) { set_error("dfq_le_plan_set_safe_mode: a plan with slab tiles (DFQ_LE_FUSE=1) waits inside every launch"); return DFQ_ERR_STATE; }
    p->resident_off = true;
    p->merged = false;
    p->sweep_grid = 0;
    if (p->resident_why.empty() || p->resident) p->resident_why = "safe mode: one launch per dependency level, no in-launch waits";
    return DFQ_OK;
}
const char* dfq_le_plan_resident_reason(const dfq_le_plan* p) { return p ? p->resident_why.c_str() : ""; }
int dfq_le_resident_stats(dfq_le_plan* p, void* stream, int64_t* out5) {
    if (!res_on(p)) return fail_arg("dfq_le_resident_stats: not a resident plan");
    return le_resident_stats(p->resident, as_stream(stream), out5);
}
// persistent workgroups of a streaming sweep launch (le_sweep_kernel), 0 when the plan launches one workgroup per tile
int32_t dfq_le_plan_sweep_workgroups(const dfq_le_plan* p) { return (p && !res_on(p)) ? p->sweep_grid : 0; }

// Tuning aid: restart, run `n_sweeps` sweeps of the persistent launch with per-tile phase stamps (100 MHz wall clock;
// [tile][6 sweeps][8 points]: 0 sweep start, 1 s_A solved, 2 row statistics published, 3 s_B solved, 4 new values +
// statistics published, 5 ticket taken, 6 decision seen; [7] of sweep 0 = layer << 32 | rows << 16 | columns).  Synchronises.
int dfq_le_resident_trace(dfq_le_plan* p, const dfq_le_config* cfg, int32_t n_sweeps, void* stream, int64_t* out, int64_t capacity) {
    if (!p || !cfg || !out || !res_on(p)) return fail_arg("dfq_le_resident_trace: not a resident plan");
    const int64_t words = le_resident_trace_words(p->resident);
    if (capacity < words) return fail_arg("dfq_le_resident_trace: need room for %lld words", (long long)words);
    hipStream_t st = as_stream(stream);
    long long* d = nullptr;
    DFQ_HIP_TRY(dfq::dev_malloc((void**)&d, words * sizeof(long long)));
    DFQ_HIP_TRY(hipMemsetAsync(d, 0, words * sizeof(long long), st));
    unsigned long long* err = p->d_dep + (size_t)p->n_rels * kDepStride;
    hipLaunchKernelGGL(le_reset_kernel, dim3(1), dim3(64), 0, st, p->d_state, 1, cfg->converge_thres, (int)cfg->converge_count,
                       (int)cfg->max_sweeps, err);
    int rc = le_resident_enqueue(p->resident, cfg, p->d_state, err, n_sweeps, st, d);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(out, d, words * sizeof(long long), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail_hip(e, "trace copy", __FILE__, __LINE__);
    }
    dfq::dev_free(d);
    return rc;
}
int64_t dfq_le_resident_trace_words(const dfq_le_plan* p) { return (p && p->resident) ? le_resident_trace_words(p->resident) : 0; }

// launches of a sweep (the convergence kernel not counted): 1, or the number of dependency levels with DFQ_LE_MERGED=0
int32_t dfq_le_plan_levels(const dfq_le_plan* p) { return p ? (p->merged ? (p->levels.empty() ? 0 : 1) : (int32_t)p->levels.size()) : 0; }
int32_t dfq_le_plan_depth(const dfq_le_plan* p) { return p ? (int32_t)p->levels.size() : 0; }
int64_t dfq_le_plan_paired_elements(const dfq_le_plan* p) { return p ? p->paired_total : 0; }
int64_t dfq_le_plan_rw_elements(const dfq_le_plan* p) { return p ? p->rw_total + (res_on(p) ? p->fr_total : 0) : 0; }   // (the persistent launch holds every paired layer)
int64_t dfq_le_plan_ro_elements(const dfq_le_plan* p) { return p ? p->ro_total : 0; }
// deferred stores of the streaming engine: elements (a part of rw_elements) that are read every sweep but written only
// every `depth`-th; depth 1 = every sweep (resident plans, DFQ_LE_DEFER=1)
int64_t dfq_le_plan_deferred_elements(const dfq_le_plan* p) { return (p && !res_on(p)) ? p->deferred_total : 0; }
int32_t dfq_le_plan_defer_depth(const dfq_le_plan* p) { return (p && !res_on(p)) ? p->defer : 1; }
// free-running segments of the streaming engine (dfq_le_cf.hpp): elements (NOT part of rw_elements) of layers whose every
// statistic is closed-form -- read and written once per `group` sweeps by le_lean_kernel; group 1 = none
int64_t dfq_le_plan_free_running_elements(const dfq_le_plan* p) { return (p && !res_on(p)) ? p->fr_total : 0; }
int32_t dfq_le_plan_free_running_group(const dfq_le_plan* p) { return (p && !res_on(p)) ? p->cf_group : 1; }
// 1: the lean launches of the free-running layers run in the background (a second stream, two groups of look-ahead: dfq_le_cf.hpp)
int32_t dfq_le_plan_lean_background(const dfq_le_plan* p) { return (p && !res_on(p) && p->cf_bg) ? 1 : 0; }
int32_t dfq_le_plan_lean_tiles(const dfq_le_plan* p) { return (p && !res_on(p)) ? p->n_lean : 0; }
// tile `tile` of the lean launch: out3 = kind (0/1 rows of 16-byte vectors / floats, 2 thread per row, 3/4 columns, 5 thread per
// row of a chain's last layer), rows, floats per row
int dfq_le_plan_lean_info(const dfq_le_plan* p, int32_t tile, int64_t* out3) {
    if (!p || !out3 || tile < 0 || tile >= p->n_lean) return fail_arg("dfq_le_plan_lean_info: bad argument");
    for (int i = 0; i < 3; ++i) out3[i] = p->lean_info[(size_t)3 * tile + i];
    return DFQ_OK;
}

// the slice of the sweep's workgroup table that launch `launch` covers
static bool launch_slice(const dfq_le_plan* p, int launch, int* begin, int* count, int* n_rels, int64_t* rw, int64_t* ro) {
    if (!p || launch < 0 || launch >= dfq_le_plan_levels(p)) return false;
    if (p->merged) {
        *begin = 0; *count = (int)p->h_blocks.size(); *n_rels = p->n_rels; *rw = p->rw_total; *ro = p->ro_total;
    } else {
        const LevelLaunch& L = p->levels[launch];
        *begin = L.block_begin; *count = L.n_blocks; *n_rels = L.n_rels; *rw = L.rw_elems; *ro = L.ro_elems;
    }
    return true;
}

int dfq_le_plan_level_grid(const dfq_le_plan* p, int32_t level, int32_t* grid_x, int32_t* grid_y) {
    int b, c, n; int64_t rw, ro;
    if (!launch_slice(p, level, &b, &c, &n, &rw, &ro)) return fail_arg("dfq_le_plan_level_grid: bad level");
    if (grid_x) *grid_x = c;
    if (grid_y) *grid_y = 1;
    return DFQ_OK;
}

int dfq_le_plan_block_info(const dfq_le_plan* p, int32_t launch, int32_t block, int64_t* out8) {
    int b, c, n; int64_t rw, ro;
    if (!out8 || !launch_slice(p, launch, &b, &c, &n, &rw, &ro) || block < 0 || block >= c) return fail_arg("dfq_le_plan_block_info: bad argument");
    const LeBlockRef& ref = p->h_blocks[(size_t)b + block];
    const LeRelDev& R = p->h_rels[ref.rel];
    const bool col = ref.tile >= R.n_row_tiles;
    const int tile = col ? ref.tile - R.n_row_tiles : ref.tile;
    const int vec = col ? R.ct_vec : R.rt_vec;
    const int n_rows = col ? R.o2 : R.o1, row_len = col ? R.i2g * R.khkw : R.row_len;
    const int t_rows = col ? R.ct_rows : R.rt_rows, t_cols = col ? R.ct_cols : R.rt_cols, slabs = col ? R.ct_slabs : R.rt_slabs;
    const int rblk = tile / slabs, slab = tile - rblk * slabs;
    const int nr = std::min(t_rows, n_rows - rblk * t_rows), nc = std::min(t_cols, row_len - slab * t_cols);
    const bool stat_only = col && R.w2_interior;
    out8[0] = (col ? 3 : 0) + (vec == 4 ? 0 : vec == 1 ? 1 : 2);
    out8[1] = nr;
    out8[2] = nc;
    out8[3] = stat_only ? 0 : (int64_t)nr * nc;
    out8[4] = stat_only ? (int64_t)nr * nc : 0;
    out8[5] = R.dep_idx >= 0 ? 1 : 0;
    out8[6] = col ? (R.out_rows != nullptr) : (R.out_cols != nullptr);
    out8[7] = ref.rel;
    return DFQ_OK;
}

int32_t dfq_le_plan_level_launches(const dfq_le_plan* p, int32_t level, int64_t* rw_elems,
                                   int64_t* ro_elems, int32_t* n_workgroups) {
    int b, c, n; int64_t rw, ro;
    if (!launch_slice(p, level, &b, &c, &n, &rw, &ro)) return fail_arg("dfq_le_plan_level_launches: bad level");
    if (rw_elems) *rw_elems = rw;
    if (ro_elems) *ro_elems = ro;
    if (n_workgroups) *n_workgroups = c;
    return n;
}

}  // extern "C"

static LeParams plan_params(const dfq_le_plan* p, const dfq_le_config* cfg) {
    LeParams q = make_params(cfg);
    q.defer = p->defer;
    return q;
}

static int le_bg_join(dfq_le_plan* p, hipStream_t st);
// reset the loop state, clear every stat word, recompute the stats of the untouched weights
static int le_restart(dfq_le_plan* p, const dfq_le_config* cfg, hipStream_t st) {
    int rcj = le_bg_join(p, st);
    if (rcj) return rcj;
    {
        ClearArgs ca;
        ca.p[0] = (uint32_t*)p->d_dep; ca.words[0] = (long long)(2 * ((size_t)(2 * p->n_rels + 1) * kDepStride + 1));
        ca.p[1] = p->n_rels > 0 ? p->d_stats : nullptr; ca.words[1] = p->n_rels > 0 ? (long long)(4 * p->stat_words) : 0;
        ca.p[2] = nullptr; ca.words[2] = 0; ca.p[3] = nullptr; ca.words[3] = 0;
        const long long most = std::max(std::max(ca.words[0], (long long)p->cf_ring_floats), std::max(ca.words[1], (long long)p->n_nets));
        const int grid = (int)std::max<long long>(1, std::min<long long>((most + 1023) / 1024, 512));
        hipLaunchKernelGGL(le_prepare_kernel, dim3(grid), dim3(256), 0, st, ca, p->d_state, p->n_nets, cfg->converge_thres,
                           (int)cfg->converge_count, (int)cfg->max_sweeps, p->d_cf_ring, (long long)p->cf_ring_floats);
        DFQ_CHECK_LAUNCH();
    }
    p->sweep_index = 0;
    if (p->defer > 1 && p->n_hold_rels > 0) {
        hipLaunchKernelGGL(le_hold_reset_kernel, dim3(p->n_hold_rels), dim3(kBlock), 0, st, (const LeRelDev*)p->d_rels,
                           (const int32_t*)p->d_hold_rels, (const LeState*)p->d_state, p->defer, 1);
        DFQ_CHECK_LAUNCH();
    }
    if (p->n_rels > 0) {
        hipLaunchKernelGGL(le_bootstrap_kernel, dim3(p->boot_blocks), dim3(kBlock), 0, st,
                           (const LeRelDev*)p->d_rels, (const int32_t*)p->d_boot_map);
        DFQ_CHECK_LAUNCH();
    }
    if (p->cf_group > 1) {
        // the factors of the first group's sweeps (background mode: of the first TWO groups'), from the scalars the bootstrap launch has just taken
        hipLaunchKernelGGL(le_cf_solve_kernel, dim3(p->n_cf_blocks), dim3(kCtlBlock), 0, st, (const LeCfSeg*)p->d_cf_segs,
                           (const LeCfRel*)p->d_cf_rels, (const int32_t*)p->d_cf_map, plan_params(p, cfg), 0, (p->cf_bg ? 2 : 1) * p->cf_group, p->cf_group, 1,
                           (const LeState*)p->d_state);
        DFQ_CHECK_LAUNCH();
    }
    return DFQ_OK;
}

// deferred stores: bring the weights up to date with the sweeps run so far (a no-op for networks whose last sweep stored)
static int le_flush(dfq_le_plan* p, hipStream_t st) {
    int rcj = le_bg_join(p, st);                     // (a background lean launch stores the values the write-back starts from)
    if (rcj) return rcj;
    if (p->n_flush == 0) return DFQ_OK;
    hipLaunchKernelGGL(le_flush_kernel, dim3(p->n_flush), dim3(kBlock), 0, st, (const LeFlushRef*)p->d_flush, (const LeState*)p->d_state,
                       p->defer, p->cf_group);
    DFQ_CHECK_LAUNCH();
    if (p->n_hold_rels > 0) {
        hipLaunchKernelGGL(le_hold_reset_kernel, dim3(p->n_hold_rels), dim3(kBlock), 0, st, (const LeRelDev*)p->d_rels,
                           (const int32_t*)p->d_hold_rels, (const LeState*)p->d_state, p->defer, 0);
        DFQ_CHECK_LAUNCH();
    }
    if (p->cf_group > 1) {
        // free-running segments: the pending sweeps into the [O1] vectors, their ring entries back to 1
        hipLaunchKernelGGL(le_cf_settle_kernel, dim3(p->n_cf_rels), dim3(kBlock), 0, st, (const LeCfRel*)p->d_cf_rels,
                           (const LeState*)p->d_state, p->cf_group);
        DFQ_CHECK_LAUNCH();
    }
    return DFQ_OK;
}

// partial sums of the sweep being enqueued (one array per sweep of a group: the lean tiles of a group's first sweep leave the
// later sweeps' sums in theirs, dfq_le_cf.hpp)
static double* sweep_partials(const dfq_le_plan* p) { return p->d_partials + (p->cf_group > 1 ? (p->sweep_index & (2 * p->cf_group - 1)) : 0) * p->part_stride; }

// the lean tiles of the free-running layers: at the first sweep of a group only
// are the lean tiles part of the sweep's own launch?  (one launch per sweep on le_level_kernel only)
static bool lean_woven(const dfq_le_plan* p) { return p->d_blocks_woven != nullptr && p->merged && p->sweep_grid == 0; }
template <int MODE>
static void le_lean_dispatch(const dfq_le_plan* p, const LeanArgs& a, hipStream_t st) {
    if (p->cf_group == 2)
        hipLaunchKernelGGL((le_lean_kernel<2, MODE>), dim3(p->n_lean), dim3(kBlock), 0, st, (const LeLeanRef*)p->d_lean, a, (const LeState*)p->d_state, p->d_partials);
    else if (p->cf_group == 4)
        hipLaunchKernelGGL((le_lean_kernel<4, MODE>), dim3(p->n_lean), dim3(kBlock), 0, st, (const LeLeanRef*)p->d_lean, a, (const LeState*)p->d_state, p->d_partials);
    else
        hipLaunchKernelGGL((le_lean_kernel<8, MODE>), dim3(p->n_lean), dim3(kBlock), 0, st, (const LeLeanRef*)p->d_lean, a, (const LeState*)p->d_state, p->d_partials);
}
// `in_line`: a background plan's launch on the caller's stream after all (the profiling entry points bracket every launch with
// events; an in-order stream meets every deadline by itself)
static int le_launch_lean(dfq_le_plan* p, hipStream_t st, bool in_line = false) {
    if (p->cf_group <= 1 || p->n_lean == 0 || (p->sweep_index & (p->cf_group - 1)) != 0 || lean_woven(p)) return DFQ_OK;
    LeanArgs a;
    a.k = (int32_t)p->sweep_index; a.pad = 0; a.part_stride = p->part_stride;
    if (!p->cf_bg) {
        le_lean_dispatch<kLeanInline>(p, a, st);
    } else if (p->sweep_index == 0) {
        le_lean_dispatch<kLeanFirst>(p, a, st);                 // the |dW| sums of the first two groups: nothing to overlap with yet
    } else if (in_line) {
        le_lean_dispatch<kLeanBg>(p, a, st);
    } else {
        // next to the sweep launches of this group, on the plan's own stream: it may begin where the caller's stream stands now
        // (the verdict of sweep k-1, the factors, an earlier call's write-back), and the convergence launch of sweep k+G waits
        // for it (le_bg_deadline) -- as does everything that touches the free-running layers or the rings (le_bg_join)
        if (!p->bg_stream) {
            int lo = 0, hi = 0;
            DFQ_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));        // (lo: the numerically largest = the lowest priority)
            const char* pe = getenv("DFQ_LE_CF_BG_PRIO");
            const int prio = (pe && pe[0] == 'h') ? hi : (pe && pe[0] == 'n') ? (lo + hi) / 2 : lo;     // (A/B: high / normal / low, the default)
            DFQ_HIP_TRY(hipStreamCreateWithPriority(&p->bg_stream, hipStreamNonBlocking, prio));
            DFQ_HIP_TRY(hipEventCreateWithFlags(&p->bg_fork, hipEventDisableTiming));
            for (auto& e : p->bg_done) DFQ_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        const int which = (int)((p->sweep_index / p->cf_group) & 1);
        DFQ_HIP_TRY(hipEventRecord(p->bg_fork, st));
        DFQ_HIP_TRY(hipStreamWaitEvent(p->bg_stream, p->bg_fork, 0));
        le_lean_dispatch<kLeanBg>(p, a, p->bg_stream);
        DFQ_HIP_TRY(hipEventRecord(p->bg_done[which], p->bg_stream));
        p->bg_wait_at[which] = p->sweep_index + p->cf_group;
    }
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}
// the convergence launch of the sweep being enqueued reads |dW| sums a background launch leaves: wait for that one
static int le_bg_deadline(dfq_le_plan* p, hipStream_t st) {
    for (int i = 0; i < 2; ++i)
        if (p->bg_wait_at[i] >= 0 && p->bg_wait_at[i] <= p->sweep_index) {
            DFQ_HIP_TRY(hipStreamWaitEvent(st, p->bg_done[i], 0));
            p->bg_wait_at[i] = -1;
        }
    return DFQ_OK;
}
// ... and so does whatever touches the free-running layers, the rings or the [O1] vectors next (write-back, restart)
static int le_bg_join(dfq_le_plan* p, hipStream_t st) {
    for (int i = 0; i < 2; ++i)
        if (p->bg_wait_at[i] >= 0) {
            DFQ_HIP_TRY(hipStreamWaitEvent(st, p->bg_done[i], 0));
            p->bg_wait_at[i] = -1;
        }
    return DFQ_OK;
}

// launch number `launch` of a sweep (see dfq_le_plan_levels)
static int le_launch_level(dfq_le_plan* p, int launch, const LeParams& q, hipStream_t st,
                           LeTrace tr = LeTrace{nullptr, 0, 0}) {
    int begin, count, n; int64_t rw, ro;
    if (!launch_slice(p, launch, &begin, &count, &n, &rw, &ro)) return fail_arg("le_launch_level: bad launch");
    const LeBlockRef* table = p->d_blocks + begin;
    if (lean_woven(p) && (p->sweep_index & (p->cf_group - 1)) == 0) { table = p->d_blocks_woven; count = p->n_woven; }   // a group's first sweep
    if (count == 0) return DFQ_OK;
    if (p->sweep_grid > 0) {
        DFQ_LAUNCH_RESIDENT(le_sweep_kernel, dim3(p->sweep_grid), dim3(kBlock), 0, st, (const LeRelDev*)p->d_rels,
                            (const LeTileRef*)p->d_tiles, count, q, (int)p->sweep_index, (const LeState*)p->d_state, p->n_nets,
                            sweep_partials(p), p->d_dep, p->d_dep + (size_t)p->n_rels * kDepStride, tr);
        DFQ_CHECK_LAUNCH();
        return DFQ_OK;
    }
    LevelArgs la;
    la.table = (const LeRelDev*)p->d_rels; la.blocks = table; la.p = q; la.sweep = (int32_t)p->sweep_index; la.pad = 0;
    la.state = (const LeState*)p->d_state; la.partials = sweep_partials(p); la.dep_counters = p->d_dep;
    la.err = p->d_dep + (size_t)p->n_rels * kDepStride; la.tr = tr; la.lean = (const LeLeanRef*)p->d_lean; la.part_stride = p->part_stride;
#ifdef DFQ_EMU
    if (p->has_slab_tiles) {           // tiles wait for LATER tiles of their row block: the CPU emulation must keep the launch's workgroups alive together
        DFQ_LAUNCH_SPINNING(le_level_kernel<false>, dim3(count), dim3(kBlock), kLevelSmem, st, la);
        return DFQ_OK;
    }
#endif
    if (tr.out) hipLaunchKernelGGL(le_level_kernel<true>, dim3(count), dim3(kBlock), kLevelSmem, st, la);
    else hipLaunchKernelGGL(le_level_kernel<false>, dim3(count), dim3(kBlock), kLevelSmem, st, la);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

#ifdef DFQ_EMU
constexpr int kCtlHelpers = 8;
#else
constexpr int kCtlHelpers = 224;           // workgroups that clear the statistics arenas next to the per-network verdicts
#endif
static int le_launch_control(dfq_le_plan* p, const dfq_le_config* cfg, hipStream_t st) {
    const int n_clear = (int)std::min<int64_t>(kCtlHelpers, (p->stat_words + p->r1_zero_words + 8 * kCtlBlock - 1) / (8 * kCtlBlock));
    // at the last sweep of a group: the solver workgroups of the free-running segments leave the next group's factors -- in
    // background mode those of the group after the next (dfq_le_cf.hpp)
    const bool solve = p->cf_group > 1 && ((p->sweep_index + 1) & (p->cf_group - 1)) == 0;
    const int n_helpers = std::max(1, n_clear);
    int rcb = le_bg_deadline(p, st);
    if (rcb) return rcb;
    hipLaunchKernelGGL(le_control_kernel, dim3(p->n_nets + n_helpers + (solve ? p->n_cf_blocks : 0)), dim3(kCtlBlock), 0, st,
                       (const LeLayerDiff*)p->d_layer_diff, (const LeNetDesc*)p->d_nets, p->n_nets,
                       (const double*)sweep_partials(p), p->d_layer_mean, p->d_stats,
                       (int64_t)p->stat_words, p->d_stats + 2 * p->stat_words, (int64_t)p->stat_words,
                       (int64_t)p->r1_zero_words, (int)(p->sweep_index & 1),
                       p->d_state, cfg->converge_thres, (int)cfg->converge_count, (int)cfg->max_sweeps, p->uni_layers, p->uni_tiles,
                       n_helpers, (const LeCfSeg*)p->d_cf_segs, (const LeCfRel*)p->d_cf_rels, (const int32_t*)p->d_cf_map,
                       plan_params(p, cfg), (int)(p->sweep_index + 1 + (p->cf_bg ? p->cf_group : 0)), p->cf_group);
    DFQ_CHECK_LAUNCH();
    p->sweep_index += 1;             // the control launch closes a sweep
    return DFQ_OK;
}

extern "C" {

static int le_enqueue_direct(dfq_le_plan* p, const dfq_le_config* cfg, int n_sweeps, int restart, hipStream_t st) {
    const LeParams q = plan_params(p, cfg);
    int rc;
    if (restart && (rc = le_restart(p, cfg, st))) return rc;
    if (n_sweeps <= 0) return DFQ_OK;
    // one-launch sweeps contain in-launch waits: never concurrent with another stream's (dfq_common.hpp); not while
    // the stream is being captured (the graph launch is guarded instead)
    // The guard is taken around the whole loop of sweeps.  Round 5 tried it PER SWEEP LAUNCH (DFQ_LE_GUARD_PER_LAUNCH=1): only the
    // sweep kernel waits inside a launch, so with two batches in flight the sweep launches of the two streams could alternate and
    // one batch's control launch, launch gap and ramp would run next to the other batch's sweep kernel.  Measured (batch of 32, two
    // streams, two rounds): 1.437e10 weights/s against 1.472e10 -- an event record and a cross-stream wait per launch cost more than
    // the overlap returns.  Kept as a switch.
    static const bool guard_loop = !(getenv("DFQ_LE_GUARD_PER_LAUNCH") && getenv("DFQ_LE_GUARD_PER_LAUNCH")[0] == '1');
    // (a plan whose every layer is free-running -- ResNet-18 -- launches no sweep kernel at all: nothing waits, nothing to guard;
    // the guard's event record showed as 6 us between the last convergence launch and the write-back of its 0.1 ms pass)
    const bool guarded = (p->merged || p->has_slab_tiles) && !p->h_blocks.empty() &&
                         !(p->capture_stream != nullptr && st == p->capture_stream);      // the NULL stream is a caller's stream too
    std::unique_ptr<SpinGuard> guard;
    if (guarded && guard_loop) guard.reset(new SpinGuard(st));
    for (int s = 0; s < n_sweeps; ++s) {
        if ((rc = le_launch_lean(p, st))) return rc;
        for (int l = 0; l < dfq_le_plan_levels(p); ++l) {
            std::unique_ptr<SpinGuard> g1;
            if (guarded && !guard_loop) g1.reset(new SpinGuard(st));
            if ((rc = le_launch_level(p, l, q, st))) return rc;
        }
        if ((rc = le_launch_control(p, cfg, st))) return rc;
    }
    // the write-back of the deferred stores waits for nobody: it runs outside the guard, next to the first sweeps of a batch
    // another stream has in flight
    guard.reset();
    return le_flush(p, st);
}

// hipGraph replay is opt-in (DFQ_GRAPH=1): on ROCm 7.2 replaying these few-microsecond kernels as graph
// nodes is slower than plain launches on both sides (MobileNetV2: 2.60 ms vs 2.26 ms of GPU time per
// 47-sweep run, 1.85 vs 2.2 ms of host time), see DESIGN.md section 7.
static bool graphs_enabled() {
    const char* e = getenv("DFQ_GRAPH");
    return e && e[0] == '1';
}

int dfq_le_enqueue(dfq_le_plan* p, const dfq_le_config* cfg, int32_t n_sweeps, int32_t restart, void* stream) {
    if (!p || !cfg || n_sweeps < 0) return fail_arg("dfq_le_enqueue: bad argument");
    hipStream_t st = as_stream(stream);
    if (res_on(p)) {
        // one persistent launch runs up to n_sweeps sweeps from the state in d_state (it re-derives the statistics from
        // the weights it loads, so there is nothing to bootstrap and nothing to carry between calls)
        unsigned long long* err = p->d_dep + (size_t)p->n_rels * kDepStride;
        return le_resident_enqueue(p->resident, cfg, p->d_state, err, n_sweeps, st, nullptr, restart ? 1 : 0);
    }
    if (!graphs_enabled() || n_sweeps < 2) return le_enqueue_direct(p, cfg, n_sweeps, restart, st);
    // A whole run of sweeps is a few hundred dependent launches with arguments that only depend on
    // (config, index of the first sweep -- the dependency counters count up from the last restart): record it
    // once on a private stream, replay it as one graph launch.
    const int64_t start_index = restart ? 0 : p->sweep_index;
    std::vector<unsigned char> key(sizeof(int32_t) * 3 + sizeof(dfq_le_config));
    const int32_t head[3] = {n_sweeps, restart ? 1 : 0, (int32_t)start_index};
    memcpy(key.data(), head, sizeof(head));
    memcpy(key.data() + sizeof(head), cfg, sizeof(dfq_le_config));
    hipGraphExec_t exec = nullptr;
    for (auto& g : p->graphs) if (g.key == key) exec = g.exec;
    if (!exec) {
        if (!p->capture_stream) DFQ_HIP_TRY(hipStreamCreate(&p->capture_stream));
        if (!restart) p->sweep_index = start_index;
        DFQ_HIP_TRY(hipStreamBeginCapture(p->capture_stream, hipStreamCaptureModeThreadLocal));
        const int rc = le_enqueue_direct(p, cfg, n_sweeps, restart, p->capture_stream);
        hipGraph_t graph = nullptr;
        const hipError_t ee = hipStreamEndCapture(p->capture_stream, &graph);
        if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (ee != hipSuccess) return fail_hip(ee, "hipStreamEndCapture", __FILE__, __LINE__);
        DFQ_HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        (void)hipGraphDestroy(graph);
        if (p->graphs.size() >= 16) { (void)hipGraphExecDestroy(p->graphs.front().exec); p->graphs.erase(p->graphs.begin()); }
        p->graphs.push_back({key, exec});
    }
    p->sweep_index = start_index + n_sweeps;
    SpinGuard guard(st);
    DFQ_HIP_TRY(hipGraphLaunch(exec, st));
    return DFQ_OK;
}

int dfq_le_profile(dfq_le_plan* p, const dfq_le_config* cfg, int32_t n_sweeps, void* stream, double* level_ms,
                   double* control_ms, int32_t* n_level_launches, double* empty_bracket_ms, double* lean_ms, int32_t* n_lean_launches) {
    if (!p || !cfg || n_sweeps <= 0 || !level_ms) return fail_arg("dfq_le_profile: bad argument");
    hipStream_t st = as_stream(stream);
    const LeParams q = plan_params(p, cfg);
    const int n_levels = dfq_le_plan_levels(p);
    const int per_sweep = n_levels + 2;
    std::vector<hipEvent_t> ev((size_t)2 * per_sweep * n_sweeps);
    for (auto& e : ev) DFQ_HIP_TRY(hipEventCreate(&e));
    int rc = le_restart(p, cfg, st);
    if (rc) return rc;
    // what a pair of event records costs with nothing between them (subtracted by the caller)
    const int n_cal = 32;
    std::vector<hipEvent_t> cal((size_t)2 * n_cal);
    for (auto& e : cal) DFQ_HIP_TRY(hipEventCreate(&e));
    for (int i = 0; i < n_cal; ++i) {
        DFQ_HIP_TRY(hipEventRecord(cal[2 * i], st));
        DFQ_HIP_TRY(hipEventRecord(cal[2 * i + 1], st));
    }
    size_t k = 0;
    std::vector<char> had_lean(n_sweeps, 0);
    for (int s = 0; s < n_sweeps; ++s) {
        // the lean tiles of the free-running layers (first sweep of a group only, dfq_le_cf.hpp): a bracket of their own
        had_lean[s] = p->cf_group > 1 && p->n_lean > 0 && (p->sweep_index & (p->cf_group - 1)) == 0 && !lean_woven(p);
        DFQ_HIP_TRY(hipEventRecord(ev[k++], st));
        if ((rc = le_launch_lean(p, st, true))) return rc;
        DFQ_HIP_TRY(hipEventRecord(ev[k++], st));
        for (int l = 0; l < n_levels; ++l) {
            DFQ_HIP_TRY(hipEventRecord(ev[k++], st));
            if ((rc = le_launch_level(p, l, q, st))) return rc;
            DFQ_HIP_TRY(hipEventRecord(ev[k++], st));
        }
        DFQ_HIP_TRY(hipEventRecord(ev[k++], st));
        if ((rc = le_launch_control(p, cfg, st))) return rc;
        DFQ_HIP_TRY(hipEventRecord(ev[k++], st));
    }
    if ((rc = le_flush(p, st))) return rc;
    DFQ_HIP_TRY(hipStreamSynchronize(st));
    for (int l = 0; l < n_levels; ++l) level_ms[l] = 0.0;
    double ctl = 0.0, lean = 0.0;
    int n_lean = 0;
    k = 0;
    for (int s = 0; s < n_sweeps; ++s) {
        for (int l = -1; l <= n_levels; ++l) {
            float ms = 0.0f;
            DFQ_HIP_TRY(hipEventElapsedTime(&ms, ev[k], ev[k + 1]));
            k += 2;
            if (l < 0) { if (had_lean[s]) { lean += ms; n_lean += 1; } }
            else if (l < n_levels) level_ms[l] += ms;
            else ctl += ms;
        }
    }
    if (control_ms) *control_ms = ctl;
    if (n_level_launches) *n_level_launches = n_levels * n_sweeps;
    if (lean_ms) *lean_ms = lean;
    if (n_lean_launches) *n_lean_launches = n_lean;
    double cal_ms = 0.0;
    for (int i = 0; i < n_cal; ++i) {
        float ms = 0.0f;
        DFQ_HIP_TRY(hipEventElapsedTime(&ms, cal[2 * i], cal[2 * i + 1]));
        cal_ms += ms;
    }
    if (empty_bracket_ms) *empty_bracket_ms = cal_ms / n_cal;
    for (auto& e : cal) (void)hipEventDestroy(e);
    for (auto& e : ev) (void)hipEventDestroy(e);
    return DFQ_OK;
}

int dfq_le_trace(dfq_le_plan* p, const dfq_le_config* cfg, int32_t launch, int32_t block, void* stream,
                 int64_t* stamps16) {
    if (!p || !cfg || !stamps16 || launch < 0 || launch >= dfq_le_plan_levels(p)) return fail_arg("dfq_le_trace: bad argument");
    hipStream_t st = as_stream(stream);
    const LeParams q = plan_params(p, cfg);
    long long* d = nullptr;
    DFQ_HIP_TRY(dfq::dev_malloc((void**)&d, 16 * sizeof(long long)));
    DFQ_HIP_TRY(hipMemsetAsync(d, 0, 16 * sizeof(long long), st));
    int rc = le_restart(p, cfg, st);
    const char* te = getenv("DFQ_TRACE_SWEEP");    // default: the second sweep (steady-state stat flow)
    const int traced = (te && atoi(te) >= 0) ? atoi(te) : 1;
    for (int s = 0; s <= traced && !rc; ++s) {
        rc = le_launch_lean(p, st, true);
        for (int l = 0; l < dfq_le_plan_levels(p) && !rc; ++l)
            rc = le_launch_level(p, l, q, st, (s == traced && l == launch) ? LeTrace{d, block, 0} : LeTrace{nullptr, 0, 0});
        if (!rc) rc = le_launch_control(p, cfg, st);
    }
    if (!rc) rc = le_flush(p, st);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(stamps16, d, 16 * sizeof(long long), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail_hip(e, "trace copy", __FILE__, __LINE__);
    }
    dfq::dev_free(d);
    return rc;
}

int dfq_le_trace_blocks(dfq_le_plan* p, const dfq_le_config* cfg, int32_t launch, void* stream, int64_t* out,
                        int64_t capacity_blocks) {
    int tb, tc, tn; int64_t trw, tro;
    if (!cfg || !out || !launch_slice(p, launch, &tb, &tc, &tn, &trw, &tro)) return fail_arg("dfq_le_trace_blocks: bad argument");
    const int64_t n_blocks = tc;
    if (capacity_blocks < n_blocks) return fail_arg("dfq_le_trace_blocks: need room for %lld workgroups", (long long)n_blocks);
    hipStream_t st = as_stream(stream);
    const LeParams q = plan_params(p, cfg);
    long long* d = nullptr;
    DFQ_HIP_TRY(dfq::dev_malloc((void**)&d, 3 * n_blocks * sizeof(long long)));
    DFQ_HIP_TRY(hipMemsetAsync(d, 0, 3 * n_blocks * sizeof(long long), st));
    int rc = le_restart(p, cfg, st);
    // trace the third sweep (steady state, code and tables warm); DFQ_TRACE_SWEEP picks another one (with deferred stores
    // of depth 2 the third sweep does not store, the fourth does)
    const char* te = getenv("DFQ_TRACE_SWEEP");
    const int traced = (te && atoi(te) >= 0) ? atoi(te) : 2;
    for (int s = 0; s <= traced && !rc; ++s) {
        rc = le_launch_lean(p, st, true);
        for (int l = 0; l < dfq_le_plan_levels(p) && !rc; ++l)
            rc = le_launch_level(p, l, q, st, (s == traced && l == launch) ? LeTrace{d, -1, 0} : LeTrace{nullptr, 0, 0});
        if (!rc) rc = le_launch_control(p, cfg, st);
    }
    if (!rc) rc = le_flush(p, st);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(out, d, 3 * n_blocks * sizeof(long long), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail_hip(e, "trace copy", __FILE__, __LINE__);
    }
    dfq::dev_free(d);
    return rc;
}

int dfq_le_query_all(dfq_le_plan* p, void* stream, dfq_le_result* out, int32_t* all_done) {
    if (!p) return fail_arg("dfq_le_query: null plan");
    std::vector<LeState> h(p->n_nets);
    hipStream_t st = as_stream(stream);
    unsigned long long gave_up = 0;
    DFQ_HIP_TRY(hipMemcpyAsync(h.data(), p->d_state, sizeof(LeState) * p->n_nets, hipMemcpyDeviceToHost, st));
    DFQ_HIP_TRY(hipMemcpyAsync(&gave_up, p->d_dep + (size_t)p->n_rels * kDepStride, sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    DFQ_HIP_TRY(hipStreamSynchronize(st));
    if (gave_up) {
        set_error("dfq_le_query: a workgroup gave up waiting for the tiles it depends on (results are invalid)");
        return DFQ_ERR_ABANDONED;
    }
    int32_t done = 1;
    for (int n = 0; n < p->n_nets; ++n) {
        if (out) {
            out[n].sweeps = h[n].sweeps;
            out[n].stall_count = h[n].count;
            out[n].diff = h[n].diff;
            out[n].last_diff_tmp = h[n].last_diff_tmp;
        }
        done &= h[n].done ? 1 : 0;
    }
    if (all_done) *all_done = done;
    return DFQ_OK;
}

int dfq_le_query(dfq_le_plan* p, void* stream, dfq_le_result* out, int32_t* done) {
    if (!p) return fail_arg("dfq_le_query: null plan");
    std::vector<dfq_le_result> all(p->n_nets);
    const int rc = dfq_le_query_all(p, stream, all.data(), done);
    if (rc) return rc;
    if (out) *out = all[0];
    return DFQ_OK;
}

int32_t dfq_le_plan_nets(const dfq_le_plan* p) { return p ? p->n_nets : 0; }

// ---- a stopping rule that spans several plans (the sharded pass, dfq_amd/sharded.py) ----
int dfq_le_set_diff_log(dfq_le_plan* p, double* log_device, int32_t capacity) {
    if (!p || capacity < 0 || (capacity > 0 && !log_device)) return fail_arg("dfq_le_set_diff_log: bad argument");
    if (p->n_nets != 1) return fail_arg("dfq_le_set_diff_log: single-network plans only");
    LeState h;
    DFQ_HIP_TRY(hipMemcpy(&h, p->d_state, sizeof(LeState), hipMemcpyDeviceToHost));
    h.log = capacity > 0 ? log_device : nullptr;
    h.log_cap = capacity;
    DFQ_HIP_TRY(hipMemcpy(p->d_state, &h, sizeof(LeState), hipMemcpyHostToDevice));
    return DFQ_OK;
}

}  // extern "C"

// dfq.py:110-115 over `n` values of diff_tmp that are sums over ALL participants: ext = { diff, count, sweeps, done } as doubles
__global__ void le_shared_verdict_kernel(const double* __restrict__ reduced, int n, double* __restrict__ ext, dfq::LeState* states,
                                         int n_nets, double converge_thres, int converge_count, int max_sweeps) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double diff = ext[0];
    int count = (int)ext[1], sweeps = (int)ext[2];
    bool done = ext[3] != 0.0;
    for (int i = 0; i < n && !done; ++i) {
        const double diff_tmp = reduced[i];
        if (fabs(diff - diff_tmp) > 1e-9) { count = 0; diff = diff_tmp; }
        else { count += 1; }
        sweeps += 1;
        done = !((diff > converge_thres) && (count < converge_count) && (max_sweeps < 0 || sweeps < max_sweeps));
    }
    ext[0] = diff; ext[1] = (double)count; ext[2] = (double)sweeps; ext[3] = done ? 1.0 : 0.0;
    if (done) for (int k = 0; k < n_nets; ++k) states[k].done = 1;        // sweeps already enqueued behind this become no-ops
}

extern "C" {

int dfq_le_shared_verdict(dfq_le_plan* p, const double* reduced_device, int32_t n, double* ext4_device, double converge_thres,
                          int32_t converge_count, int32_t max_sweeps, void* stream) {
    if (!reduced_device || !ext4_device || n < 0) return fail_arg("dfq_le_shared_verdict: bad argument");
    hipLaunchKernelGGL(le_shared_verdict_kernel, dim3(1), dim3(64), 0, as_stream(stream), reduced_device, (int)n, ext4_device,
                       p ? p->d_state : nullptr, p ? p->n_nets : 0, converge_thres, (int)converge_count, (int)max_sweeps);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_le_run(dfq_le_plan* p, const dfq_le_config* cfg, void* stream, dfq_le_result* out) {
    if (!p || !cfg) return fail_arg("dfq_le_run: bad argument");
    int rc;
    int32_t done = 0;
    dfq_le_result res;
    int chunk = 8;
    bool streamed = !res_on(p);
    if (!streamed) {
        // the kernel stops by itself where the reference's loop stops; the reset of the loop state rides on the launch in front of it
        rc = dfq_le_enqueue(p, cfg, cfg->max_sweeps >= 0 ? cfg->max_sweeps : (1 << 30), 1, stream);
        if (rc) return rc;
        rc = dfq_le_query(p, stream, &res, &done);
        if (rc == DFQ_ERR_ABANDONED) {
            // A workgroup of the persistent launch gave up a wait (DFQ_SPIN_LIMIT: the chip was not the library's alone for seconds,
            // or the dispatch the in-launch waits count on did not happen).  The launch stores all or nothing (dfq_le_resident.hip,
            // "all or nothing"): if no tile stored, the network is exactly as the caller passed it, and the pass is simply run
            // again -- on ONE LAUNCH PER LEVEL, which waits for nothing inside a launch -- instead of failing.  The plan stays
            // on that engine (dfq_le_plan_degraded counts the repeats; dfq_le_plan_resident_reason says why).
            int64_t stored = -1;
            if (le_resident_stored_tiles(p->resident, as_stream(stream), &stored) != DFQ_OK || stored != 0) return rc;
            p->resident_off = true;
            p->merged = false;
            p->sweep_grid = 0;                   // (the persistent-workgroup variant of the sweep walks the WHOLE table: per-level launches are le_level_kernel's)
            p->degraded += 1;
            p->resident_why = "an in-launch wait of the persistent launch was abandoned (nothing had been stored): this plan now runs one launch per level";
            streamed = true;
        } else if (rc) {
            return rc;
        }
    }
    if (!streamed) {
        // (done above)
    } else if ((rc = dfq_le_enqueue(p, cfg, 0, 1, stream)) != 0) {
        return rc;
    } else if (cfg->max_sweeps >= 0) {
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
