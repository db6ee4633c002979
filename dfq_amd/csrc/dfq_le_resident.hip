// LDS-resident cross-layer equalisation: the WHOLE data-dependent loop of dfq.py:78-117 for one network as ONE persistent
// (cooperative) launch (gfx950).
//
// A single network is tiny for this chip: MobileNetV2's paired layers are 13.9 MB, the LDS of 256 CUs holds 40 MB.  The
// streaming kernel of dfq_le.hip re-reads and re-writes every weight every sweep and pays, per sweep, one launch boundary
// plus a chain of dependent tile latencies.  Here every workgroup loads ONE [rows x columns] tile of ONE paired layer into
// its LDS once (32 KB; the first version kept it in registers -- see "the tile: three layouts" below for why not), keeps it
// there for all sweeps, and only the per-channel statistics travel between workgroups; the weights are written back once,
// after the loop has stopped.  A sweep then costs the latency of its dependency chain of statistics hand-offs.
//
// What a tile of layer L does in sweep k (A = the relation whose SECOND layer is L, B = the relation whose FIRST
// layer is L; either may be absent; dfq.py:85-101 processes A before B):
//   top      : ONE poll of the counters of everything other tiles produce for this sweep, then every statistics word the
//              tile needs is requested in one trip through the memory system (words of phase 2 too if they are there);
//   phase 1 (A): solve s_A per input channel (dfq.py:58-59); the row statistics of t = fl(w / s_A) (dfq.py:73) are
//                published for B -- no store, t is recomputed below;
//   phase 2 (B): solve s_B per row (rows complete in the tile: from LDS; else merged over the row block's tiles);
//   phase 3    : |dW| in float64 and the statistics of the values the sweep WILL produce, from the pending factors;
//                column statistics published for A of the NEXT sweep (row statistics for a chain start);
//   commit     : w <- fl(fl(w / s_A) * s_B) in LDS (the reference's two roundings in its order), once the verdict of
//                sweep k-1 says that sweep k happens at all.
// Statistics words are 64-bit {sweep tag : order-preserving float bits}, merged with device-scope atomicMax: a newer
// sweep always wins, so nothing is ever cleared; two parities (tag & 1) keep a sweep's readers and the next sweep's
// writers apart.  "The tiles of layer X have published" is a monotonic counter per layer and statistic kind (eight copies,
// see `arrive`): strict (all contributions performed) for statistics merged from several tiles, relaxed (issued) for
// single-producer statistics, whose readers validate the tag of every word they read.
// Convergence (dfq.py:105-115): every tile leaves its float64 partial as two tagged words; ONE small tile reads them,
// sums them per layer in a fixed order and publishes diff_tmp as two tagged words; every workgroup advances its own copy
// of the reference's (diff, count) state machine with it.
//
// All workgroups of the launch must be resident at once (they wait for each other in cycles over the sweeps): the launch is
// cooperative (the runtime guarantees co-residency or refuses; the plan additionally refuses networks beyond three quarters
// of the occupancy limit and the caller then streams), the library never runs two kernels with in-launch waits concurrently
// (SpinGuard), every wait is bounded (DFQ_SPIN_LIMIT), and the launch stores ALL OR NOTHING: a tile writes its result back only
// once every tile has finished the loop, so after an abandoned wait (reported as DFQ_ERR_ABANDONED by the query) the network is
// exactly as the caller passed it -- le_resident_stored_tiles == 0 -- and dfq_le_run repeats the pass at once on one launch per
// level (dfq_le.hip; the plan stays there, dfq_le_plan_degraded).  The one case that still leaves an undefined network is a
// give-up inside that last "every tile has finished" wait itself (some tiles stored, others did not: DFQ_ERR_STATE; that wait is
// at least 200 000 polls patient whatever the limit says).  Since round 5 the launch is an ordinary one by default
// (DFQ_COOPERATIVE=1 restores the cooperative launch).  Results are bit-identical to the streaming kernel and to the oracle
// (same IEEE operations; min/max exact).
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "dfq_common.hpp"
#include "dfq_le_shared.hpp"
#include "dfq_le_resident.hpp"

namespace dfq {

constexpr int kResTileFloats = 8192;  // the tile in LDS (32 KB)
constexpr int kResTab = 1024;        // LDS table entries of a tile: (groups x input channels) it spans
constexpr int kResRows = 256;        // rows of a tile that needs per-row tables: one per thread (512 until round 4: the second row per thread cost eleven vector registers
                                     // the kernel does not have -- three workgroups per CU -- and no tile of the BASELINE networks used it)
constexpr int kResOwn = kResRows / kBlock;
constexpr int kResStride = 16;       // one 64-bit counter per 128-byte line
constexpr int kResMaxTiles = 1536;   // partials staged in LDS when the sweep's verdict is drawn
constexpr int kResMaxLayers = 512;
#ifndef DFQ_RES_LATE_ARRIVE
#define DFQ_RES_LATE_ARRIVE 0         // 1: a strict arrival is made after the sweep's tail instead of right behind the publication (A/B on one
                                     // box, three runs each: 0.759-0.80 vs 0.729-0.766 ms for MobileNetV2, 0.474-0.491 vs 0.457-0.470 ms for
                                     // DeepLab -- the counter moves later than it could and every consumer of the layer with it)
#endif
#ifndef DFQ_RES_ABLATE
#define DFQ_RES_ABLATE 0             // tuning aid, NEVER in the product build: compile work OUT of a sweep to see what its time is made of
                                     // (tools/gpu_r05_ablate.sh; results are wrong, run with a pinned sweep count).  1: no float64 |dW| chain,
                                     // 2: no scale solves (s = 1), 4: no row-statistics pass, 8: no phase-3 pass, 16: no statistics fetch,
                                     // 32: no factor log, 64: no statistics publication (use with 16), 128: no checkpoints
#endif
// Round 5, built, parity-green and measured NO faster (tools/gpu_r05_ab.sh, MobileNetV2 47 sweeps / DeepLab 60, two rounds on one box:
// both on 0.72-0.75 / 0.48 ms, prefetch off 0.69-0.71 / 0.48, both off 0.675-0.69 / 0.48): (1) single-producer COLUMN statistics
// validated by their tags alone, no poll of the layer's counter (relax_c bits 1, 2); (2) a thread's first statistics words requested
// BEFORE the sweep's first poll, so that poll and fetch are one trip.  256 threads re-reading tagged words are a heavier poll than one
// thread on a counter line, and the trips were never what paces a sweep (profiles/r05_resident_ablation.txt).  Opt-in switches:
#ifndef DFQ_RES_NAP
#define DFQ_RES_NAP 1                // s_sleep units (64 clocks) between two looks of a waiting wave
#endif
#ifndef DFQ_RES_PRIO
#define DFQ_RES_PRIO 1               // 1: a wave raises its issue priority for the element passes (the waves it shares a SIMD with mostly poll)
#endif
#ifndef DFQ_RES_SPLIT_AB
#define DFQ_RES_SPLIT_AB 0             // 1: the tile body compiled per (A, B) case (measured: MobileNetV2 -2 %, DeepLab +2 % -- noise; three times the code)
#endif
#ifndef DFQ_RES_HOT
#define DFQ_RES_HOT 1                // 0: every [256 x float4] tile runs the generic phase-3 loop (round 4)
#endif
#ifndef DFQ_RES_COMMIT
#define DFQ_RES_COMMIT 1             // 0 (A/B only): tiles store as soon as they have finished themselves, as until round 4 (an abandoned wait then leaves
                                     // an undefined network behind and dfq_le_run must not repeat the pass)
#endif
#ifndef DFQ_RES_PIPE
#define DFQ_RES_PIPE 2               // slots of a [256 x float4] tile pass whose LDS reads are in flight together (1: the rolled loop of round 4; measured on one box,
                                     // two rounds, MobileNetV2 47 sweeps / DeepLab 60: 1 -> 0.69 / 0.48 ms, 2 -> 0.65 / 0.453, 4 -> 0.68 / 0.46)
#endif
#ifndef DFQ_RES_PREFETCH
#define DFQ_RES_PREFETCH 0
#endif
#ifndef DFQ_RES_DIRECT_COLS
#define DFQ_RES_DIRECT_COLS 0
#endif
#ifndef DFQ_RES_TOPWAIT
#define DFQ_RES_TOPWAIT 0            // 1: the sweep's first poll also WAITS for phase 2's counter (A/B; slower: see the loop)
#endif

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
    int32_t layout;                  // kLayGeneral / kLayFixed / kLayShort
    int32_t relax_r;                 // (bit 2: see le_resident_create -- the layer's row counter has no reader)
                                     // bit 0: every row of this layer lives in ONE tile -> its row statistics have a single producer;
                                     // bit 1: the same holds for relation A's FIRST layer (this tile reads THOSE row statistics: it polls the tagged
                                     // words themselves instead of that layer's counter -- one trip through the memory system less per hand-off)
    int32_t relax_c;                 // bit 0: every input channel of this layer lives in ONE tile -> likewise for its column statistics;
                                     // bit 1: the same holds for relation B's SECOND layer (this tile reads THOSE column statistics in phase 2);
                                     // bit 2: bit 0 and the tile may poll its own column words' tags instead of the layer's counter (DFQ_RES_DIRECT)
                                     // bit 3: a chain END (relation A only): the next sweep's column statistics follow from this sweep's in closed
                                     // form -- see "closed-form column statistics" in res_tile_body (round 6)
    int32_t slot;                    // logical index of the tile (partial-sum slot, checkpoint slot): the table itself is in LAUNCH order
    int32_t log_off;                 // first float of this tile inside an entry of the factor log: 1/s_A per table entry, then s_B per row
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
    u64* cnt_r;                      // [paired layer][8 copies] (x kResStride): tiles that published row statistics (see arrive)
    u64* cnt_c;                      //   "   column statistics
    u64* prog;                       // [8 copies] (x kResStride): {sweeps that happen, once the loop has stopped : verdicts drawn}
    u64* err;
    double* partials;                // [part_ring][tiles][4 waves][2]: {sweep + 1 : half of the wave's float64 sum of |dW|}
    float* log;                      // [log_ring][log_total]: the factors of the latest sweeps (see "speculation past the verdict")
    float* ckpt;                     // [2][tiles][kCkptFloats]: alternating checkpoints of the LDS tiles
    int64_t log_total;               // floats of one log entry (all tiles)
    LeState* state;
    int32_t n_tiles, n_layers;       // n_layers: targ layers of the network (layer_diff entries)
    int32_t n_sweeps;                // sweeps this launch may run
    int32_t max_sweeps;              // cfg: total cap (< 0: none)
    int32_t converge_count;
    int32_t spec;                    // sweeps a tile may run ahead of the verdicts (0: none)
    int32_t ckpt_every;              // sweeps between two checkpoints (>= spec)
    int32_t log_ring, part_ring;     // ckpt_every + spec; spec + 2
    int32_t pad;
    double converge_thres;
    long long* trace;                // tuning aid (null in production): [tile][kTraceSweeps][kTracePoints] wall-clock stamps
};

constexpr int kTraceSweeps = 6;
constexpr int kTracePoints = 16;
// Cold launch arguments -- everything a sweep touches at most once (the verdict's buffers and thresholds, the trace buffer,
// tables read in the prologue): read from the kernarg segment WHERE THEY ARE USED, through a pointer the compiler cannot
// prove loop-invariant, instead of living in scalar registers for the whole loop (the kernel's scarce resource: descriptor
// fields that stay live across the loop are spilled to vector-register lanes, and every use becomes a v_readlane).
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ const ResArgs DFQ_CONSTANT_AS& cold(const ResArgs&) {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();      // ResArgs is the kernel's first argument
    asm volatile("" : "+s"(p));
    return *(const ResArgs DFQ_CONSTANT_AS*)p;
}
#else
__device__ __forceinline__ const ResArgs& cold(const ResArgs& a) { return a; }
#endif
template <bool kTrace>
__device__ __forceinline__ void res_stamp(const ResArgs& a, int k, int point) {
    if (kTrace && threadIdx.x == 0 && k < kTraceSweeps)
        cold(a).trace[((int64_t)blockIdx.x * kTraceSweeps + k) * kTracePoints + point] = wall_clock64();
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

// the same in two steps: request the two words of a channel, decode them later (several channels in flight)
struct RangeWords { u64 a, b; };
__device__ __forceinline__ RangeWords load_range(const u64* arena, int64_t off, int64_t parity_stride, uint32_t tag, int c) {
    const u64* p = arena + off + (int64_t)(tag & 1u) * parity_stride + 2 * (int64_t)c;
    RangeWords w;
    w.a = ld_word(p); w.b = ld_word(p + 1);
    return w;
}
__device__ __forceinline__ bool tagged(const RangeWords& w, uint32_t tag) {
    return (uint32_t)(w.a >> 32) == tag && (uint32_t)(w.b >> 32) == tag;
}
__device__ __forceinline__ void decode_range(const RangeWords& w, uint32_t tag, float& mn, float& mx) {
    mn = ((uint32_t)(w.a >> 32) == tag) ? slot_min((uint32_t)w.a) : INFINITY;
    mx = ((uint32_t)(w.b >> 32) == tag) ? slot_max((uint32_t)w.b) : -INFINITY;
}

// One arrival of this tile on a layer's counter.  The counter exists in eight copies, each on its own 128-byte line (a layer
// cut into 160 tiles has 160 workgroups polling it, and polls of ONE address are served one after the other, ~12 ns each:
// a waiter polls copy blockIdx % 8, the arriving tile bumps all eight with one instruction).
//   strict : statistics merged from several tiles (atomicMax over the row / column blocks of a layer) -- a reader must not
//            look before ALL contributions have been performed: s_waitcnt 0, barrier, then the arrival.
//   relaxed: statistics with a single producer tile -- the arrival only says "issued" (barrier, no wait for the atomics:
//            one trip through the memory system less on the producer's side of every hand-off); the reader checks the
//            sweep tag every statistics word carries and reads again while a word still shows an older one.
__device__ __forceinline__ u64* cnt_line(u64* base, int layer, int copy) { return base + ((int64_t)layer * 8 + copy) * kResStride; }
__device__ __forceinline__ void arrive(u64* base, int layer, bool strict) {
    if (strict) __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x < 8) atomicAdd(cnt_line(base, layer, threadIdx.x), 1ull);
}

__device__ __forceinline__ void opaque(int& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
}

// ---- the tile: three layouts ------------------------------------------------------------------------------
// The tile lives in the workgroup's LDS for the whole launch (kResTileFloats floats; element = vector q * VEC + k with
// q = slot u * 256 + thread t, so consecutive threads touch consecutive 16-byte words: conflict-free), and every element
// loop is an ordinary rolled loop over the slots.  (The first version kept the tile in REGISTERS, which needs fully
// unrolled slot loops: 99 000 instructions for the sweep loop, 20 % of them scalar-register spill moves, far more code than
// the instruction cache holds -- every phase ran at instruction-fetch speed, 5-7 us for 64 multiplies per thread.  LDS reads
// cost a fraction of that, the code is a few kilobytes, and the registers are free for the reductions.)
// A layout says which elements a thread's slots hold and provides the element loops of a sweep:
//   LayFixed   float4 slots, a tile row of nc / 4 <= 256 vectors on the next power of two of lanes: a thread keeps ONE column
//              position for all its slots (slot u = row u * rps + t / tcv), so 1/s_A of its four columns and its column
//              minima / maxima live in registers and the row reduction is a butterfly over the tcv lanes of a row;
//   LayShort   rows of <= 32 floats (depthwise k x k kernels, the stem): one THREAD per row, row statistics without
//              any cross-lane traffic;
//   LayGeneral anything else: slot u of thread t holds vector q = u * 256 + t of the tile, statistics through LDS
//              atomics (correct for every geometry, slow for wide rows: the plan avoids it where it can).
struct TileGeo {
    int g_lo, g_n, nci, i0;          // LDS table of phase 1 / column statistics: (group - g_lo) * nci + (ii - i0)
};
__device__ __forceinline__ TileGeo tile_geo(const ResTile& T) {
    TileGeo G;
    G.i0 = small_div(T.c0, T.khkw);
    G.nci = small_div(T.c0 + T.nc - 1, T.khkw) - G.i0 + 1;
    G.g_lo = small_div(T.r0, T.go);
    G.g_n = small_div(T.r0 + T.nr - 1, T.go) - G.g_lo + 1;
    return G;
}
__device__ __forceinline__ float abs_f32(float d) { return __uint_as_float(__float_as_uint(d) & 0x7fffffffu); }
__device__ __forceinline__ void lds_minmax(uint32_t* pair, float mn, float mx) {
    atomicMax(pair, ~enc_ord(mn));
    atomicMax(pair + 1, enc_ord(mx));
}

template <int VEC_>
struct LayGeneral {
    static constexpr int VEC = VEC_;
    static constexpr bool kFusedCols = false;
    static constexpr bool kFusedRows = false;
    __device__ __forceinline__ double update_rows(const ResTile&, const TileGeo&, float*, bool, bool, const float*, const float*, uint32_t*) const { return 0.0; }
    int tcv, n_vec, n_slots;
    __device__ __forceinline__ void init(const ResTile& T, const TileGeo&) {
        tcv = T.nc / VEC; n_vec = T.nr * tcv; n_slots = (n_vec + kBlock - 1) / kBlock;
    }
    __device__ __forceinline__ int tab(const ResTile& T, const TileGeo& G, int row, int pos_k) const {
        return (small_div(T.r0 + row, T.go) - G.g_lo) * G.nci + small_div(pos_k, T.khkw) - G.i0;
    }
    template <typename F>     // f(x, row, pos, on) for every slot in use (uniform trip count); x = pointer to the VEC floats
    __device__ __forceinline__ void slots(const ResTile& T, float* tile, F f) const {
        for (int u = 0; u < n_slots; ++u) {
            const int q = u * kBlock + (int)threadIdx.x;
            const bool on = q < n_vec;
            const int qq = on ? q : 0;
            const int row = small_div(qq, tcv);
            const int pos = T.c0 + (qq - row * tcv) * VEC;
            f(tile + (int64_t)qq * VEC, row, pos, on);
        }
    }
    __device__ __forceinline__ void load(const ResTile& T, float* tile) const {
        const gfloat* wt = (const gfloat*)T.w;
        slots(T, tile, [&](float* x, int row, int pos, bool on) {
            if (!on) return;
            const gfloat* src = wt + ((int64_t)(T.r0 + row) * T.row_len + pos);
            if (VEC == 4) *(fvec4*)x = *(const gfvec4*)src;
            else x[0] = *src;
        });
    }
    __device__ __forceinline__ void store(const ResTile& T, float* tile) const {
        gfloat* wt = (gfloat*)T.w;
        slots(T, tile, [&](float* x, int row, int pos, bool on) {
            if (!on) return;
            gfloat* dst = wt + ((int64_t)(T.r0 + row) * T.row_len + pos);
            if (VEC == 4) *(gfvec4*)dst = *(const fvec4*)x;
            else *dst = x[0];
        });
    }
    // the value an element WILL have: fl(fl(w * 1/s_A) * s_B) with the factors in use (dfq.py:73 then :62, both rounded)
    __device__ __forceinline__ float val(const ResTile& T, const TileGeo& G, float w, bool useA, bool useB, const float* sh_inv,
                                         const float* sh_s, int row, int pos_k) const {
        const float tt = useA ? w * sh_inv[tab(T, G, row, pos_k)] : w;
        return useB ? tt * sh_s[row] : tt;
    }
    // row statistics of the (pending) values into sh_row (zeroed by the caller)
    __device__ __forceinline__ void row_stats(const ResTile& T, const TileGeo& G, float* tile, bool useA, bool useB,
                                              const float* sh_inv, const float* sh_s, uint32_t* sh_row) const {
        const int lane = threadIdx.x % kWave;
        slots(T, tile, [&](float* x, int row, int pos, bool on) {
            float mn = INFINITY, mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float y = val(T, G, x[k], useA, useB, sh_inv, sh_s, row, pos + k);
                mn = vmin_raw(mn, on ? y : INFINITY);
                mx = vmax_raw(mx, on ? y : -INFINITY);
            }
            const int r_first = __shfl(row, 0), r_last = __shfl(row, kWave - 1);
            const int on_all = __shfl((int)on, kWave - 1);             // lanes are ordered: the last one decides
            if (on_all && r_first == r_last) {
                wave_minmax(mn, mx);
                if (lane == 0) lds_minmax(sh_row + 2 * row, mn, mx);
            } else if (on) {
                lds_minmax(sh_row + 2 * row, mn, mx);
            }
        });
    }
    __device__ __forceinline__ void col_stats(const ResTile& T, const TileGeo& G, float* tile, bool useA, bool useB,
                                              const float* sh_inv, const float* sh_s, uint32_t* sh_col) const {
        slots(T, tile, [&](float* x, int row, int pos, bool on) {
            if (!on) return;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float y = val(T, G, x[k], useA, useB, sh_inv, sh_s, row, pos + k);
                lds_minmax(sh_col + 2 * tab(T, G, row, pos + k), y, y);
            }
        });
    }
    __device__ __forceinline__ double diff_and_cols(const ResTile&, const TileGeo&, float*, bool, bool, const float*, const float*, uint32_t*, bool = false, bool = true) const { return 0.0; }
    // w <- new; kDiff: returns the thread's sum of |new - old| in float64 (else 0: the replay of a rollback)
    template <bool kDiff>
    __device__ __forceinline__ double update(const ResTile& T, const TileGeo& G, float* tile, bool useA, bool useB,
                                             const float* sh_inv, const float* sh_s) const {
        double acc = 0.0;
        slots(T, tile, [&](float* x, int row, int pos, bool on) {
            double part = 0.0;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float nv = val(T, G, x[k], useA, useB, sh_inv, sh_s, row, pos + k);
                if (kDiff) part += (double)abs_f32(nv - x[k]);
                if (on) x[k] = nv;
            }
            acc += on ? part : 0.0;
        });
        return acc;
    }
};

struct LayFixed {
    static constexpr int VEC = 4;
    static constexpr bool kFusedCols = true;
    int lg_tcv, tcv, rps, rsub, pos, n_used;
    int tabk[4];                       // table offset of the thread's four columns inside a group row
    bool one_group, lane_on;
    __device__ __forceinline__ void init(const ResTile& T, const TileGeo& G) {
        // a row of nc / 4 vectors occupies the next power of two of lanes (tcv); the lanes past its end hold duplicates of
        // its last vector (harmless for min / max, excluded from the stores and from |dW| through `lane_on`)
        const int real = T.nc / 4;
        lg_tcv = (real == 1) ? 0 : 32 - __builtin_clz((unsigned)(real - 1));
        tcv = 1 << lg_tcv;
        rps = kBlock >> lg_tcv;
        rsub = (int)threadIdx.x >> lg_tcv;
        const int colv = (int)threadIdx.x & (tcv - 1);
        lane_on = colv < real;
        pos = T.c0 + min(colv, real - 1) * 4;
        n_used = (T.nr + rps - 1) >> (8 - lg_tcv);
        one_group = G.g_n == 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) tabk[k] = small_div(pos + k, T.khkw) - G.i0;
    }
    __device__ __forceinline__ int group_row(const ResTile& T, const TileGeo& G, int row) const {
        return one_group ? 0 : (small_div(T.r0 + row, T.go) - G.g_lo) * G.nci;
    }
    template <typename F>     // f(x, row, on): x = pointer to the slot's float4 in the LDS tile
    __device__ __forceinline__ void slots(const ResTile& T, float* tile, F f) const {

        for (int u = 0; u < n_used; ++u) {
            int row = u * rps + rsub;
            const bool row_ok = row < T.nr;
            const bool on = row_ok && lane_on;           // padded lanes keep their OWN row (they duplicate its last vector)
            row = row_ok ? row : T.nr - 1;
            f(tile + (u * kBlock + (int)threadIdx.x) * 4, row, on);
        }
    }
    // The element passes of a sweep, kPipe slots at a time: the LDS reads of all of them (tile vectors and per-row factors) are
    // issued BEFORE the first is used.  (Round 5: a rolled loop with one ds_read_b128 -> s_waitcnt -> arithmetic -> ds_write per slot
    // paid the LDS latency eight times per pass, and a wave of this kernel has nothing else to issue meanwhile: 71 % of its cycles
    // were spent parked -- profiles/r05_resident_pmc.txt.)  f(x, xv, sr, row, on) per slot in use, in slot order.
    static constexpr int kPipe = DFQ_RES_PIPE;
    template <typename F>
    __device__ __forceinline__ void slots_piped(const ResTile& T, float* tile, bool useB, const float* sh_s, F f) const {
        for (int u0 = 0; u0 < n_used; u0 += kPipe) {
            fvec4 xv[kPipe];
            float sr[kPipe];
            int rowj[kPipe];
            bool onj[kPipe];
#pragma unroll
            for (int j = 0; j < kPipe; ++j) {
                const int u = min(u0 + j, n_used - 1);                 // past the end: the last slot again (read, not used)
                int row = u * rps + rsub;
                const bool row_ok = row < T.nr;
                onj[j] = row_ok && lane_on;
                rowj[j] = row_ok ? row : T.nr - 1;
                xv[j] = *(const fvec4*)(tile + (u * kBlock + (int)threadIdx.x) * 4);
                sr[j] = useB ? sh_s[rowj[j]] : 1.0f;
            }
#pragma unroll
            for (int j = 0; j < kPipe; ++j)
                if (u0 + j < n_used) f(tile + ((u0 + j) * kBlock + (int)threadIdx.x) * 4, xv[j], sr[j], rowj[j], onj[j]);     // (uniform)
        }
    }
    __device__ __forceinline__ void load(const ResTile& T, float* tile) const {
        const gfloat* wt = (const gfloat*)T.w;
        slots(T, tile, [&](float* x, int row, bool) { *(fvec4*)x = *(const gfvec4*)(wt + ((int64_t)(T.r0 + row) * T.row_len + pos)); });
    }
    __device__ __forceinline__ void store(const ResTile& T, float* tile) const {
        gfloat* wt = (gfloat*)T.w;
        slots(T, tile, [&](float* x, int row, bool on) { if (on) *(gfvec4*)(wt + ((int64_t)(T.r0 + row) * T.row_len + pos)) = *(const fvec4*)x; });
    }
    __device__ __forceinline__ void inv4(const ResTile& T, const TileGeo& G, const float* sh_inv, int row, float (&iv)[4]) const {
        const int gr = group_row(T, G, row);
#pragma unroll
        for (int k = 0; k < 4; ++k) iv[k] = sh_inv[gr + tabk[k]];
    }
    // kCommit (round 5, a chain start's phase 3): the pass also WRITES the values it takes the statistics of and returns the thread's
    // float64 sum of |new - old| -- w <- new, |dW| and the row statistics of the new values in ONE pass over the tile instead of
    // update<true> followed by row_stats (two generic passes: 3.4 us of a [64 x 96] tile's ~10 us sweep)
    template <bool kCommit>
    __device__ __forceinline__ double row_stats_t(const ResTile& T, const TileGeo& G, float* tile, bool useA, bool useB,
                                                  const float* sh_inv, const float* sh_s, uint32_t* sh_row) const {
        double acc = 0.0;
        const int lane = threadIdx.x % kWave;
        const int w = tcv < kWave ? tcv : kWave;                    // lanes of a wave that share a row
        float iv[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        if (useA && one_group) inv4(T, G, sh_inv, 0, iv);
        if (w == kWave) {
            // a whole wave per row and slot: the rows of four slots are reduced TOGETHER (wave_minmax4: 7 instead of 24 butterfly
            // steps' worth of moves -- a third of this pass's instructions)
            const bool wave_on = (((int)threadIdx.x & ~(kWave - 1)) & (tcv - 1)) < (T.nc / 4);   // else: duplicates of the row's last vector only
            for (int u0 = 0; u0 < n_used; u0 += 4) {
                float mn4[4], mx4[4];
                fvec4 xv4[4];
                float sr4[4];
                int row4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {                           // all four slots' LDS reads in flight (see slots_piped)
                    const int u = min(u0 + j, n_used - 1);
                    int row = u * rps + rsub;
                    row4[j] = row < T.nr ? row : T.nr - 1;
                    xv4[j] = *(const fvec4*)(tile + (u * kBlock + (int)threadIdx.x) * 4);
                    sr4[j] = useB ? sh_s[row4[j]] : 1.0f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mn4[j] = INFINITY; mx4[j] = -INFINITY;
                    if (u0 + j < n_used) {                              // uniform
                        if (useA && !one_group) inv4(T, G, sh_inv, row4[j], iv);
                        const float sr = sr4[j];
                        const fvec4 xv = xv4[j];
                        fvec4 yv;
                        double part = 0.0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float y = (xv[k] * iv[k]) * sr;        // * 1.0f is exact
                            yv[k] = y;
                            if (kCommit) part += (double)abs_f32(y - xv[k]);
                            mn4[j] = vmin_raw(mn4[j], y); mx4[j] = vmax_raw(mx4[j], y);
                        }
                        if (kCommit) {
                            *(fvec4*)(tile + ((u0 + j) * kBlock + (int)threadIdx.x) * 4) = yv;       // the thread's own slot
                            acc += ((u0 + j) * rps + rsub < T.nr && lane_on) ? part : 0.0;           // slot order
                        }
                    }
                }
                float mn, mx;
                wave_minmax4(mn4, mx4, mn, mx);
                const int u = u0 + wave_minmax4_slot(lane);
                const int row = u * rps + rsub;
                if ((lane & 15) == 0 && u < n_used && row < T.nr && wave_on) lds_minmax(sh_row + 2 * row, mn, mx);   // one writer per row and wave
            }
            return acc;
        }
        slots_piped(T, tile, useB, sh_s, [&](float* x, const fvec4& xv, float sr, int row, bool on) {
            if (useA && !one_group) inv4(T, G, sh_inv, row, iv);
            float mn = INFINITY, mx = -INFINITY;
            fvec4 yv;
            double part = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float y = (xv[k] * iv[k]) * sr;                // * 1.0f is exact
                yv[k] = y;
                if (kCommit) part += (double)abs_f32(y - xv[k]);
                mn = vmin_raw(mn, y); mx = vmax_raw(mx, y);
            }
            if (kCommit) { *(fvec4*)x = yv; acc += on ? part : 0.0; }
            // register-file butterflies (xor_lane_minmax, dfq_common.hpp) behind uniform guards
            if (1 < w) xor_lane_minmax<1>(mn, mx);
            if (2 < w) xor_lane_minmax<2>(mn, mx);
            if (4 < w) xor_lane_minmax<4>(mn, mx);
            if (8 < w) xor_lane_minmax<8>(mn, mx);
            if (16 < w) xor_lane_minmax<16>(mn, mx);
            if ((lane & (w - 1)) == 0 && on) lds_minmax(sh_row + 2 * row, mn, mx);   // one writer per row and wave
        });
        return acc;
    }
    __device__ __forceinline__ void row_stats(const ResTile& T, const TileGeo& G, float* tile, bool useA, bool useB,
                                              const float* sh_inv, const float* sh_s, uint32_t* sh_row) const {
        (void)row_stats_t<false>(T, G, tile, useA, useB, sh_inv, sh_s, sh_row);
    }
    static constexpr bool kFusedRows = true;
    __device__ __forceinline__ double update_rows(const ResTile& T, const TileGeo& G, float* tile, bool useA, bool useB,
                                                  const float* sh_inv, const float* sh_s, uint32_t* sh_row) const {
        return row_stats_t<true>(T, G, tile, useA, useB, sh_inv, sh_s, sh_row);
    }
    // cross-lane part of the column statistics: lanes of the wave that hold the same columns (ids differing in bits >= lg tcv)
    __device__ __forceinline__ void cols_finish(float (&cmn)[4], float (&cmx)[4], uint32_t* sh_col) const {
        const int lane = threadIdx.x % kWave;
#define DFQ_COL_STEP(M) if ((M) >= tcv) { _Pragma("unroll") for (int k = 0; k < 4; ++k) xor_lane_minmax<M>(cmn[k], cmx[k]); }
        DFQ_COL_STEP(1) DFQ_COL_STEP(2) DFQ_COL_STEP(4) DFQ_COL_STEP(8) DFQ_COL_STEP(16) DFQ_COL_STEP(32)
#undef DFQ_COL_STEP
        if (lane < tcv && lane_on) {
#pragma unroll
            for (int k = 0; k < 4; ++k) lds_minmax(sh_col + 2 * tabk[k], cmn[k], cmx[k]);
        }
    }
    // The hot form of the phase-3 pass, compiled for its case (round 5): ONE group (the thread's four 1/s_A stay in registers, the
    // column statistics too), relation A present, w <- new in the same pass, and `UB` (relation B present) a template switch --
    // every tile of a pointwise layer behind a depthwise or a pointwise one.  The generic loop below decides all of that per
    // slot with uniform branches and keeps both arms in the loop body: ~90 issued instructions per float4 slot where this
    // needs ~40, and a pass of a [8 x 960] tile took 3.0 us where the bare loop takes 0.7 (tools/litmus/lds_pass.hip).
    // (`cols` false -- a chain end with closed-form column statistics, round 6: w <- new and |dW| only; a uniform branch)
    template <bool UB>
    __device__ __forceinline__ double diff_and_cols_hot(const ResTile& T, const TileGeo& G, float* tile, const float* sh_inv,
                                                        const float* sh_s, uint32_t* sh_col, bool cols) const {
        double acc = 0.0;
        float iv[4];
        inv4(T, G, sh_inv, 0, iv);
        float cmn[4], cmx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { cmn[k] = INFINITY; cmx[k] = -INFINITY; }
        const int tid4 = (int)threadIdx.x * 4;
        const int nr_last = T.nr - 1;
        for (int u0 = 0; u0 < n_used; u0 += 2) {
            const int u1 = min(u0 + 1, n_used - 1);                        // an odd count: the last slot twice (second copy unused)
            const int ra = u0 * rps + rsub, rb = u1 * rps + rsub;
            float* xa = tile + u0 * (kBlock * 4) + tid4;
            float* xb = tile + u1 * (kBlock * 4) + tid4;
            const fvec4 va = *(const fvec4*)xa, vb = *(const fvec4*)xb;
            float sa = 1.0f, sb = 1.0f;
            if (UB) { sa = sh_s[min(ra, nr_last)]; sb = sh_s[min(rb, nr_last)]; }
            const bool on_a = ra < T.nr && lane_on, on_b = rb < T.nr && lane_on && (u0 + 1 < n_used);
            fvec4 na, nb;
            double pa = 0.0, pb = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                na[k] = UB ? (va[k] * iv[k]) * sa : va[k] * iv[k];        // dfq.py:73 then :62, both rounded
                nb[k] = UB ? (vb[k] * iv[k]) * sb : vb[k] * iv[k];
                if (!(DFQ_RES_ABLATE & 1)) { pa += (double)abs_f32(na[k] - va[k]); pb += (double)abs_f32(nb[k] - vb[k]); }
                // (no `on` select: padded lanes / rows hold exact duplicates of valid elements; the unused second copy of an odd
                // count's last slot is such a duplicate too)
                if (UB || cols) {
                    cmn[k] = vmin_raw(vmin_raw(cmn[k], na[k]), nb[k]);
                    cmx[k] = vmax_raw(vmax_raw(cmx[k], na[k]), nb[k]);
                }
            }
            *(fvec4*)xa = na;
            if (u0 + 1 < n_used) *(fvec4*)xb = nb;                         // (uniform)
            acc += on_a ? pa : 0.0;                                        // slot order, as the generic loop sums
            acc += on_b ? pb : 0.0;
        }
        if (UB || cols) cols_finish(cmn, cmx, sh_col);
        return acc;
    }
    // one pass: |dW| and the column statistics of the new values; `commit`: w <- new in the same pass
    __device__ __forceinline__ double diff_and_cols(const ResTile& T, const TileGeo& G, float* tile, bool useA, bool useB,
                                                    const float* sh_inv, const float* sh_s, uint32_t* sh_col, bool commit = false, bool cols = true) const {
        if (DFQ_RES_HOT && one_group && useA && commit)
            return useB ? diff_and_cols_hot<true>(T, G, tile, sh_inv, sh_s, sh_col, true) : diff_and_cols_hot<false>(T, G, tile, sh_inv, sh_s, sh_col, cols);
        double acc = 0.0;
        float iv[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        if (useA && one_group) inv4(T, G, sh_inv, 0, iv);
        float cmn[4], cmx[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { cmn[k] = INFINITY; cmx[k] = -INFINITY; }
        slots_piped(T, tile, useB, sh_s, [&](float* x, const fvec4& xv, float sr, int row, bool on) {
            if (useA && !one_group) inv4(T, G, sh_inv, row, iv);
            const int gr = one_group ? 0 : group_row(T, G, row);
            fvec4 nw;
            double part = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float nv = (xv[k] * iv[k]) * sr;            // dfq.py:73 then :62, both rounded
                nw[k] = nv;
                if (!(DFQ_RES_ABLATE & 1)) part += (double)abs_f32(nv - xv[k]);
                if (!cols) {
                } else if (one_group) {
                    // (no `on` select: the slots of padded lanes / rows hold exact DUPLICATES of valid elements -- loaded as such,
                    // updated with the same factors, checkpointed raw -- and a duplicate changes no minimum or maximum)
                    cmn[k] = vmin_raw(cmn[k], nv);
                    cmx[k] = vmax_raw(cmx[k], nv);
                } else if (on) {
                    lds_minmax(sh_col + 2 * (gr + tabk[k]), nv, nv);
                }
            }
            if (commit) *(fvec4*)x = nw;                          // the thread's own slot (padded lanes hold private duplicates)
            acc += on ? part : 0.0;
        });
        if (one_group && cols) cols_finish(cmn, cmx, sh_col);
        return acc;
    }
    __device__ __forceinline__ void col_stats(const ResTile& T, const TileGeo& G, float* tile, bool useA, bool useB,
                                              const float* sh_inv, const float* sh_s, uint32_t* sh_col) const {
        (void)diff_and_cols(T, G, tile, useA, useB, sh_inv, sh_s, sh_col);
    }
    template <bool kDiff>
    __device__ __forceinline__ double update(const ResTile& T, const TileGeo& G, float* tile, bool useA, bool useB,
                                             const float* sh_inv, const float* sh_s) const {
        double acc = 0.0;
        float iv[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        if (useA && one_group) inv4(T, G, sh_inv, 0, iv);
        slots_piped(T, tile, useB, sh_s, [&](float* x, const fvec4& xv, float s, int row, bool on) {
            if (useA && !one_group) inv4(T, G, sh_inv, row, iv);
            fvec4 nv;
            double part = 0.0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                nv[k] = (xv[k] * iv[k]) * s;                      // dfq.py:73 (rounded), then dfq.py:62
                if (kDiff) part += (double)abs_f32(nv[k] - xv[k]);
            }
            *(fvec4*)x = nv;                                      // the thread's own slot (padded lanes hold private duplicates)
            acc += on ? part : 0.0;
        });
        return acc;
    }
};

// thread t holds rows t, t + 256 of the tile (complete rows of L = row_len <= 32 floats); element e of the thread's row j
// sits at tile[(j * L + e) * 256 + t].  Everything that depends on the row only -- its group's table offset, 1/s_A of a
// depthwise row (one input channel per group: the common case), s_B -- is fetched once per row, so an element costs one LDS
// access and two multiplications (the first version divided twice per element to find its table entry: 40 instructions per
// element, 2.2 us to multiply the 18 elements a thread holds).
struct LayShort {
    static constexpr int VEC = 1;
    static constexpr bool kFusedCols = true;
    static constexpr bool kFusedRows = false;
    __device__ __forceinline__ double update_rows(const ResTile&, const TileGeo&, float*, bool, bool, const float*, const float*, uint32_t*) const { return 0.0; }
    int L, rpt, nci, khkw;
    int rt[kResOwn];                   // table offset of the row's group: (group - g_lo) * nci
    __device__ __forceinline__ void init(const ResTile& T, const TileGeo& G) {
        L = T.row_len; rpt = (T.nr + kBlock - 1) / kBlock; nci = G.nci; khkw = T.khkw;
#pragma unroll
        for (int j = 0; j < kResOwn; ++j) {
            const int row = min(j * kBlock + (int)threadIdx.x, T.nr - 1);
            rt[j] = (small_div(T.r0 + row, T.go) - G.g_lo) * G.nci;              // complete rows: i0 == 0
        }
    }
    // f(j, row, on, base): row j of the thread (base = its first element in the tile; elements are kBlock floats apart)
    template <typename F>
    __device__ __forceinline__ void rows(const ResTile& T, float* tile, F f) const {
#pragma unroll
        for (int j = 0; j < kResOwn; ++j) {
            if (j < rpt) {
                int row = j * kBlock + (int)threadIdx.x;
                const bool on = row < T.nr;
                row = on ? row : T.nr - 1;
                f(j, row, on, tile + (int64_t)j * L * kBlock + threadIdx.x);
            }
        }
    }
    __device__ __forceinline__ void load(const ResTile& T, float* tile) const {
        const gfloat* wt = (const gfloat*)T.w;
        rows(T, tile, [&](int, int row, bool, float* base) {
            for (int e = 0; e < L; ++e) base[e * kBlock] = wt[(int64_t)(T.r0 + row) * L + e];
        });
    }
    __device__ __forceinline__ void store(const ResTile& T, float* tile) const {
        gfloat* wt = (gfloat*)T.w;
        rows(T, tile, [&](int, int row, bool on, float* base) {
            if (on) for (int e = 0; e < L; ++e) wt[(int64_t)(T.r0 + row) * L + e] = base[e * kBlock];
        });
    }
    // g(e, x) for every element of a row, nine LDS reads in flight per trip (a 3 x 3 kernel is one trip; a plain loop pays
    // one LDS round trip per element)
    template <typename Gf>
    __device__ __forceinline__ void elems(const float* base, Gf g) const {
        for (int e0 = 0; e0 < L; e0 += 9) {
            float x[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) x[i] = base[min(e0 + i, L - 1) * kBlock];
#pragma unroll
            for (int i = 0; i < 9; ++i) if (e0 + i < L) g(e0 + i, x[i]);
        }
    }
    // the factors of row j: fa = 1/s_A when the row has ONE input channel (else per element through `inv_at`), fb = s_B
    __device__ __forceinline__ float fa_of(int j, bool useA, const float* sh_inv) const { return (useA && nci == 1) ? sh_inv[rt[j]] : 1.0f; }
    __device__ __forceinline__ float val(float w, int j, int e, bool useA, float fa, float fb, const float* sh_inv) const {
        const float ia = (useA && nci != 1) ? sh_inv[rt[j] + small_div(e, khkw)] : fa;
        return (w * ia) * fb;                                        // dfq.py:73 (rounded), then dfq.py:62; * 1.0f is exact
    }
    __device__ __forceinline__ void row_stats(const ResTile& T, const TileGeo& G, float* tile, bool useA, bool useB,
                                              const float* sh_inv, const float* sh_s, uint32_t* sh_row) const {
        rows(T, tile, [&](int j, int row, bool on, float* base) {
            const float fa = fa_of(j, useA, sh_inv), fb = useB ? sh_s[row] : 1.0f;
            float mn = INFINITY, mx = -INFINITY;
            elems(base, [&](int e, float x) {
                const float y = val(x, j, e, useA, fa, fb, sh_inv);
                mn = vmin_raw(mn, y); mx = vmax_raw(mx, y);
            });
            if (on) { sh_row[2 * row] = ~enc_ord(mn); sh_row[2 * row + 1] = enc_ord(mx); }    // the row's only owner
        });
    }
    // one pass: |dW| and the column statistics of the new values; `commit`: w <- new in the same pass (round 4: the depthwise tile
    // of the longest chain sits on the sweep's critical cycle and used to walk its rows twice here)
    __device__ __forceinline__ double diff_and_cols(const ResTile& T, const TileGeo& G, float* tile, bool useA, bool useB,
                                                    const float* sh_inv, const float* sh_s, uint32_t* sh_col, bool commit = false, bool cols = true) const {
        double acc = 0.0;
        rows(T, tile, [&](int j, int row, bool on, float* base) {
            const float fa = fa_of(j, useA, sh_inv), fb = useB ? sh_s[row] : 1.0f;
            double part = 0.0;
            if (nci == 1) {          // one input channel per group (depthwise): the row's range goes to its group's channel
                float mn = INFINITY, mx = -INFINITY;
                elems(base, [&](int e, float x) {
                    const float nv = val(x, j, e, useA, fa, fb, sh_inv);
                    if (!(DFQ_RES_ABLATE & 1)) part += (double)abs_f32(nv - x);
                    if (commit) base[e * kBlock] = nv;
                    mn = vmin_raw(mn, nv); mx = vmax_raw(mx, nv);
                });
                if (on && cols) lds_minmax(sh_col + 2 * rt[j], mn, mx);
            } else {
                elems(base, [&](int e, float x) {
                    const float nv = val(x, j, e, useA, fa, fb, sh_inv);
                    part += (double)abs_f32(nv - x);
                    if (commit) base[e * kBlock] = nv;
                    if (on && cols) lds_minmax(sh_col + 2 * (rt[j] + small_div(e, khkw)), nv, nv);
                });
            }
            acc += on ? part : 0.0;
        });
        return acc;
    }
    __device__ __forceinline__ void col_stats(const ResTile& T, const TileGeo& G, float* tile, bool useA, bool useB,
                                              const float* sh_inv, const float* sh_s, uint32_t* sh_col) const {
        (void)diff_and_cols(T, G, tile, useA, useB, sh_inv, sh_s, sh_col);
    }
    template <bool kDiff>
    __device__ __forceinline__ double update(const ResTile& T, const TileGeo& G, float* tile, bool useA, bool useB,
                                             const float* sh_inv, const float* sh_s) const {
        double acc = 0.0;
        rows(T, tile, [&](int j, int row, bool on, float* base) {
            const float fa = fa_of(j, useA, sh_inv), fb = useB ? sh_s[row] : 1.0f;
            double part = 0.0;
            elems(base, [&](int e, float x) {
                const float nv = val(x, j, e, useA, fa, fb, sh_inv);
                if (kDiff) part += (double)abs_f32(nv - x);
                base[e * kBlock] = nv;
            });
            acc += on ? part : 0.0;
        });
        return acc;
    }
};

// sh_row -> global row statistics of relation B (rows r0 .. r0 + nr), tagged
__device__ __forceinline__ void publish_rows(const ResArgs& a, const ResTile& T, int64_t r1_off, const uint32_t* sh_row, uint32_t tag) {
    u64* dst = a.stats + r1_off + (int64_t)(tag & 1u) * a.parity_stride;
    if (DFQ_RES_ABLATE & 64) return;
    for (int i = threadIdx.x; i < T.nr; i += kBlock) {
        publish_max(dst + 2 * (int64_t)(T.r0 + i), tag, sh_row[2 * i]);
        publish_max(dst + 2 * (int64_t)(T.r0 + i) + 1, tag, sh_row[2 * i + 1]);
    }
}
// sh_col -> global column statistics of relation A, tagged
// (and the table entry goes back to the identity 0 behind the read: the next sweep's pass accumulates into it without a clearing
// loop and a barrier of its own in front of it -- round 5)
__device__ __forceinline__ void publish_cols(const ResArgs& a, const ResTile& T, const TileGeo& G, int64_t r2_off,
                                             uint32_t* sh_col, uint32_t tag) {
    u64* dst = a.stats + r2_off + (int64_t)(tag & 1u) * a.parity_stride;
    for (int idx = threadIdx.x; idx < G.g_n * G.nci; idx += kBlock) {
        if (DFQ_RES_ABLATE & 64) { sh_col[2 * idx] = 0u; sh_col[2 * idx + 1] = 0u; continue; }
        const int gq = small_div(idx, G.nci);
        const int c = (G.g_lo + gq) * T.i2g + G.i0 + (idx - gq * G.nci);
        const uint32_t lo = sh_col[2 * idx], hi = sh_col[2 * idx + 1];
        if ((DFQ_RES_ABLATE & 8) || hi != 0u) {                               // a channel no element of this tile belongs to stays untouched
            publish_max(dst + 2 * (int64_t)c, tag, lo);
            publish_max(dst + 2 * (int64_t)c + 1, tag, hi);
            sh_col[2 * idx] = 0u; sh_col[2 * idx + 1] = 0u;
        }
    }
}
// closed-form column statistics (see res_tile_body): sh_col holds the extrema this sweep CONSUMED (raw floats, from phase 1), the
// next sweep's are fl(extremum * 1/s_A) -- published by the tile that holds the first row of the channel's group only (every
// tile of the channel holds the same pair: one producer per word), and the table goes back to zero
__device__ __forceinline__ void publish_cols_cf(const ResArgs& a, const ResTile& T, const TileGeo& G, int64_t r2_off,
                                                uint32_t* sh_col, const float* sh_inv, uint32_t tag) {
    u64* dst = a.stats + r2_off + (int64_t)(tag & 1u) * a.parity_stride;
#pragma unroll 1
    for (int idx = threadIdx.x; idx < G.g_n * G.nci; idx += kBlock) {
        const int gq = small_div(idx, G.nci);
        const int first = (G.g_lo + gq) * T.go - T.r0;
        if (first >= 0 && first < T.nr && !(DFQ_RES_ABLATE & 64)) {
            const int c = (G.g_lo + gq) * T.i2g + G.i0 + (idx - gq * G.nci);
            const float inv = sh_inv[idx];
            publish_max(dst + 2 * (int64_t)c, tag, ~enc_ord(__uint_as_float(sh_col[2 * idx]) * inv));
            publish_max(dst + 2 * (int64_t)c + 1, tag, enc_ord(__uint_as_float(sh_col[2 * idx + 1]) * inv));
        }
        sh_col[2 * idx] = 0u; sh_col[2 * idx + 1] = 0u;
    }
}

// The loop state of dfq.py:81-83,110-115: advanced by ONE workgroup, the reducer (res_reducer_body), which publishes what the
// tiles need to know of it in a single 64-bit progress word {number of sweeps that HAPPEN, once known : verdicts drawn}.
struct LoopState {
    double diff, last_diff_tmp;
    int count, sweeps, done;
};

// sum of x[0 .. n) in index order, eight LDS reads in flight per trip (a plain loop pays one LDS round trip per element)
__device__ __forceinline__ double ordered_sum(const double* x, int n) {
    double s = 0.0;
    for (int i = 0; i < n; i += 8) {
        double t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = x[min(i + j, n - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (i + j < n) ? t[j] : 0.0;                 // + 0.0 is exact
    }
    return s;
}

// ---- speculation past the verdict (round 4) -----------------------------------------------------------------------
// dfq.py:105-115 decides after sweep k whether sweep k + 1 happens, from the sum over ALL layers of mean |dW|: a global
// reduction behind every sweep.  Until round 3 a tile applied sweep k to its weights only once the verdict of sweep k - 1 had
// arrived (speculation depth one), and the builder's own trace showed nearly every tile idle 10-20 us of every 18 us sweep
// waiting for exactly that.  Now a sweep is applied AT ONCE (one pass forms the new values, |dW| and the statistics, and writes
// them to the LDS tile) and a tile may run up to `spec` sweeps ahead of the verdicts; nothing on a chain's dependency cycle
// waits for the reduction any more.  The price is a rollback when the verdict finally says "the loop stopped after sweep j"
// and the tile has already applied sweeps j + 1 .. k:
//   * every sweep's factors (1/s_A per input channel of the tile, s_B per row: <= 6 KB) go to a ring in global memory with
//     fire-and-forget stores from the threads that solved them (phases 1 and 2);
//   * every `ckpt` sweeps the LDS tile (raw, 32 KB) and the thread's [O] vector entries go to one of two alternating
//     checkpoint buffers, fire-and-forget as well (checkpoint 0 is the untouched tensor in place: nothing is stored into the
//     caller's weights before the loop has stopped);
//   * rollback = reload the newest checkpoint at or before sweep j + 1 and replay the logged factors: the same two rounded
//     multiplications per element and sweep in the same order, so the result is bit-identical to never having speculated.
// Ring sizes: a tile starts sweep k only when the verdicts of sweeps < k - spec are out, so it is at most `spec` sweeps past
// the stopping point: partial sums spec + 2 deep, factor log ckpt + spec deep, two checkpoints (spec <= ckpt).
// Stopping: the reducer publishes the stop in the progress word; every spin loop of a tile looks at that word too, so a
// tile waiting for statistics of a sweep that will never be produced (its producer has left) leaves as well.
constexpr int kCkptFloats = kResTileFloats + 4 * kResOwn * kBlock;      // raw tile + (s_cum, bn_w, bn_b, b1) x owned rows per thread

__device__ __forceinline__ u64* prog_line(u64* base, int copy) { return base + (int64_t)copy * kResStride; }

// One poll loop by one thread for everything a sweep's start depends on: up to three statistics counters (null = not needed;
// the third one is only looked at unless `need3`), the speculation bound (verdicts out >= need_v) and the stop.
// Returns 0: a wait was abandoned, 1: go (the first two counters are there), 2: go (all three are there), 3: the loop has stopped.
__device__ __forceinline__ int res_wait(const u64* w1, u64 t1, const u64* w2, u64 t2, const u64* w3, u64 t3, bool need3,
                                        const u64* prog, uint32_t need_v, u64* err, int* sh_flag, long kResSpinLimit) {
    if (threadIdx.x == 0) {
        long spins = 0;
        int ok = 1;
        for (;;) {
            const u64 a1 = w1 ? __hip_atomic_load(w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : t1;
            const u64 a2 = w2 ? __hip_atomic_load(w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : t2;
            const u64 a3 = w3 ? __hip_atomic_load(w3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : t3;
            const u64 pg = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(pg >> 32) != 0u) { ok = 3; break; }
            if (a1 >= t1 && a2 >= t2 && (a3 >= t3 || !need3) && (uint32_t)pg >= need_v) { ok = (a3 >= t3) ? 2 : 1; break; }
            __builtin_amdgcn_s_sleep(DFQ_RES_NAP);
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
    return *sh_flag;                 // (no second barrier: consecutive waits of a tile alternate between two flag words -- the callers' `wflag`)
}

// The end of a tile's loop: wait until the verdict of every sweep this tile has applied is out (or the stop is).  Returns the
// number of sweeps to keep, or -1 when the wait was abandoned.
__device__ __forceinline__ int res_final(const u64* prog, int applied, u64* err, int* sh_flag, long kResSpinLimit) {
    if (threadIdx.x == 0) {
        long spins = 0;
        int keep = -1;
        for (;;) {
            const u64 pg = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(pg >> 32) != 0u) { keep = min((int)(uint32_t)(pg >> 32), applied); break; }
            if ((uint32_t)pg >= (uint32_t)applied) { keep = applied; break; }
            __builtin_amdgcn_s_sleep(DFQ_RES_NAP);
            ++spins;
            if (spins > kResSpinLimit ||
                ((spins & 255) == 0 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull)) {
                atomicMax(err, 1ull);
                break;
            }
        }
        *sh_flag = keep;
    }
    __syncthreads();
    const int keep = *sh_flag;
    __syncthreads();
    return keep;
}

// kAB: bit 0 -- the layer is the SECOND layer of a relation (A), bit 1 -- the FIRST layer of one (B).  Template switches (round 5: as
// run-time flags both arms of every `if (hasA)` / `if (hasB)` sat in the sweep loop and were branched around; DFQ_RES_SPLIT_AB=0)
template <typename Lay, bool kTrace, int kAB>
__device__ __forceinline__ void res_tile_body(const ResArgs& a, const LeParams& p, const ResTile& T, unsigned char* smem) {
    // LDS: [tile: kResTileFloats f32][inv: kResTab f32][s: kResRows f32][row stats: 2 * kResRows u32][col stats: 2 * kResTab u32][flags: 16 words]
    //      [b1 of the owned rows: kResRows f32 -- one of the four [O] vectors lives here instead of in registers: the kernel runs AT its
    //       register limit (168 for three workgroups per CU) and the LDS has 2 KB to spare per workgroup]
    float* v = (float*)smem;                       // the tile
    float* sh_inv = v + kResTileFloats;
    float* sh_s = sh_inv + kResTab;
    uint32_t* sh_row = (uint32_t*)(sh_s + kResRows);
    uint32_t* sh_col = sh_row + 2 * kResRows;
    int* sh_flag = (int*)(sh_col + 2 * kResTab);
    int* sh_bad = sh_flag + 1;                     // set by a thread whose statistics words never showed this sweep's tag
    float* sh_ob1 = (float*)(sh_flag + 16);
    if (threadIdx.x == 0) *sh_bad = 0;
    const int tid = threadIdx.x;
    const long kResSpinLimit = p.spin_limit;
    const bool hasA = kAB < 0 ? T.relA >= 0 : (kAB & 1) != 0, hasB = kAB < 0 ? T.relB >= 0 : (kAB & 2) != 0;
    const bool chain_start = hasB && !hasA;
    const bool rows_local = T.nc == T.row_len;       // the tile holds complete rows: its row statistics are final
    // Closed-form column statistics (round 6, bit 3 of relax_c).  A chain END's sweep is w <- fl(w * 1/s_A[c]) and nothing else
    // (dfq.py:73), a positive factor per input channel: rounding is monotonic, so the minimum (maximum) of the channel's new values
    // is fl(old minimum (maximum) * 1/s_A[c]) -- exactly, whichever element attains it.  Every tile of the channel holds both
    // operands after phase 1 (the merged statistics it has just consumed, the factor it has just solved), so the next sweep's
    // statistics are two multiplications per channel instead of a min/max over the tile, a merge of the tiles' results through
    // atomics (1000 x 1280 classifier: 160 tiles x 2048 atomics per sweep) and an arrival that waits for them.  ONE tile per
    // channel publishes (the tile holding the first row of the channel's group): the words have a single producer, readers
    // validate them by their tags as they do all relaxed words, and the arrival need not wait for the atomics.  The FIRST
    // publication (the untouched weights) is still taken from the elements and merged strictly.  Opt-in: see le_resident_create.
    auto cf_cols_f = [&]() -> bool { return hasA && !hasB && (T.relax_c & 8) != 0; };      // (re-derived at each use: scalar registers)
#define cf_cols (cf_cols_f())
    // only the statistics offsets of the two relations stay live through the loop (scalar registers are the scarce resource
    // of this kernel); the [O] vector pointers are re-read where the vectors are loaded and stored
    const int64_t ra_r1 = cold(a).rels[hasA ? T.relA : 0].r1_off, ra_r2 = cold(a).rels[hasA ? T.relA : 0].r2_off;
    const int64_t rb_r1 = cold(a).rels[hasB ? T.relB : 0].r1_off, rb_r2 = cold(a).rels[hasB ? T.relB : 0].r2_off;
    const TileGeo G = tile_geo(T);
    const int n_ch = hasA ? G.g_n * G.nci : 0;
    Lay lay;
    lay.init(T, G);
    // sweeps this launch may run: its own budget and what the loop's cap leaves (the state carries the sweeps of earlier launches)
    int cap;
    {
        const auto& c = cold(a);
        cap = c.n_sweeps;
        if (c.max_sweeps >= 0) cap = min(cap, c.max_sweeps - c.state->sweeps);
    }
    const bool owner = hasB && T.owner != 0;
    float o_cum[kResOwn], o_bnw[kResOwn], o_bnb[kResOwn];      // (b1: sh_ob1[tid + j * kBlock], touched by its own thread only)
    // ---- load the tile and the [O] vectors of relation B for the rows this thread owns (also: checkpoint 0 of a rollback) ----
    auto load_pristine = [&]() {
        lay.load(T, v);
#pragma unroll
        for (int j = 0; j < kResOwn; ++j) {
            const int i = tid + j * kBlock;
            o_cum[j] = 1.0f; o_bnw[j] = 0.0f; o_bnb[j] = 0.0f;
            float b1 = 0.0f;
            if (owner && i < T.nr) {
                const ResRel RB = cold(a).rels[T.relB];
                const int c = T.r0 + i;
                o_cum[j] = RB.s_cum[c];
                if (RB.bnw) o_bnw[j] = RB.bnw[c];
                if (RB.bnb) o_bnb[j] = RB.bnb[c];
                if (RB.b1) b1 = RB.b1[c];
            }
            sh_ob1[i] = b1;
        }
    };
    load_pristine();

    // ---- statistics of the untouched weights: consumption tag 1 (sweep 0) ----
    if (cap > 0) {
        if (hasA) {
            for (int i = tid; i < 2 * G.g_n * G.nci; i += kBlock) sh_col[i] = 0u;
            __syncthreads();
            lay.col_stats(T, G, v, false, false, sh_inv, sh_s, sh_col);
            __syncthreads();
            publish_cols(a, T, G, ra_r2, sh_col, 1u);
            arrive(a.cnt_c, T.layer, !(T.relax_c & 1));
        }
        if (chain_start) {
            for (int i = tid; i < 2 * T.nr; i += kBlock) sh_row[i] = 0u;
            __syncthreads();
            lay.row_stats(T, G, v, false, false, sh_inv, sh_s, sh_row);
            __syncthreads();
            publish_rows(a, T, rb_r1, sh_row, 1u);
            arrive(a.cnt_r, T.layer, !(T.relax_r & 1));
        }
    }

    // this tile's entry of the factor log / its checkpoint slot (see "speculation past the verdict")
    auto log_entry = [&](int sweep) -> float* {
        const auto& c = cold(a);
        return c.log + (int64_t)(sweep % c.log_ring) * c.log_total + T.log_off;
    };
    auto ckpt_slot = [&](int which) -> float* {
        const auto& c = cold(a);
        return c.ckpt + ((int64_t)(which & 1) * c.n_tiles + T.slot) * kCkptFloats;
    };

    int k = 0;                       // sweeps applied to the LDS tile so far == the sweep about to run
    bool failed = false, stopped = false;
    for (; k < cap; ++k) {
        // ---- checkpoint after every `ckpt_every`-th sweep (here, at the start of the next one): the raw LDS tile (each thread its
        //      own 16-byte words, whatever the layout) and the thread's [O] entries, fire-and-forget.  Everything is formed from an
        //      index the compiler cannot see through: nothing of this cold block is precomputed and kept in registers across the
        //      sweep loop (the kernel runs AT its register limit) ----
        if (!(DFQ_RES_ABLATE & 128) && k > 0 && k % cold(a).ckpt_every == 0) {
            float* ck = ckpt_slot(k / cold(a).ckpt_every);
            int ti = tid;
            opaque(ti);
            const unsigned tt = (unsigned)ti;
            gfvec4* gk = (gfvec4*)ck;
#pragma unroll 1
            for (unsigned u = 0; u < (unsigned)(kResTileFloats / (4 * kBlock)); ++u)
                gk[u * kBlock + tt] = *(const fvec4*)(v + (u * kBlock + tt) * 4);
            if (owner) {
                gfloat* go = (gfloat*)ck + kResTileFloats;
#pragma unroll
                for (int j = 0; j < kResOwn; ++j) {
                    go[(4 * j + 0) * kBlock + tt] = o_cum[j]; go[(4 * j + 1) * kBlock + tt] = o_bnw[j];
                    go[(4 * j + 2) * kBlock + tt] = o_bnb[j]; go[(4 * j + 3) * kBlock + tt] = sh_ob1[tt + j * kBlock];
                }
            }
        }
        const uint32_t tag = (uint32_t)k + 1u;                  // what this sweep consumes
        const u64 round = (u64)k + 1ull;
        res_stamp<kTrace>(a, k, 0);
        // ---- what this sweep consumes.  Phase 1 (s_A) needs the row statistics of A's first layer (this sweep: the chain's
        //      hand-off) and this layer's own column statistics (previous sweep); phase 2 (s_B) needs the column statistics of
        //      B's second layer (previous sweep).  ONE poll looks at all three counters, at the speculation bound and at the
        //      stop; it waits only for phase 1's counters and the bound.  If phase 2's are there as well (the usual case: they
        //      are a sweep old), every statistics word of the tile is requested in the same trip through the memory system and
        //      phase 2 never waits; otherwise phase 2 polls and reads later -- the row statistics this tile publishes in phase
        //      1 must not wait for that (the tiles of B's second layer overlap their own work with it).  A chain start paces
        //      itself on its own layer's row counter (its tiles do not otherwise wait for each other, and a tile two
        //      publications ahead of a sibling would make the monotonic counter lie to the layer's consumers). ----
        // Round 5 experiment, OFF by default (see DFQ_RES_PREFETCH / DFQ_RES_DIRECT_COLS at the top: measured no faster).  The idea
        // (tools/litmus/handoff_latency.hip: one dependent trip through the memory system is ~0.4 us on an idle chip and
        // ~1 us inside this kernel, and a sweep of a tile makes two before it can compute anything -- poll, THEN fetch):
        //   * statistics words with a single producer are validated by their sweep tags, so they need no counter at all: the
        //     tile's OWN column statistics when every input channel lives in one tile (relax_c bit 0: no poll of the layer's
        //     counter), those of B's second layer likewise (bit 1), A's row statistics as before (relax_r bit 1);
        //   * and they are REQUESTED BEFORE the poll (of the speculation bound, the stop and whatever counters are left): when
        //     they are there -- the usual case behind a busy neighbour -- the poll and the fetch are ONE trip.  A word that is not
        //     there yet shows an older tag and is read again below, exactly as a relaxed word always was.
        //   Words merged from several tiles (strict) keep counter -> fetch: their tags cannot tell a complete merge from a partial one.
        const bool direct_r = hasA && (T.relax_r & 2) != 0, direct_c = hasA && (T.relax_c & 4) != 0 && DFQ_RES_DIRECT_COLS;
        const bool direct_b = hasB && (T.relax_c & 2) != 0 && DFQ_RES_DIRECT_COLS;
        RangeWords w1[kResTab / kBlock], w2[kResTab / kBlock], v1[kResOwn], v2[kResOwn];
        // (only the words of a thread's first kPreJ channels are requested ahead of the poll: the kernel runs AT its register limit --
        // three workgroups per CU -- and sixteen 64-bit words in flight across the poll spilled; tiles with more than kPreJ * 256 table
        // entries -- complete rows of 960 floats -- request the rest behind the poll, in the same trip as a retry would make)
        constexpr int kPreJ = 2;
        auto fetch_top = [&](int j_lo, int j_hi, bool want1, bool want2, bool wantb) {
#pragma unroll
            for (int j = 0; j < kResTab / kBlock; ++j) {
                if (j >= j_lo && j < j_hi && j * kBlock < n_ch) {     // uniform over the workgroup
                    const int idx = min(tid + j * kBlock, n_ch - 1);
                    const int gq = small_div(idx, G.nci);
                    const int c = (G.g_lo + gq) * T.i2g + G.i0 + (idx - gq * G.nci);
                    if (DFQ_RES_ABLATE & 16) { w1[j].a = w1[j].b = w2[j].a = w2[j].b = ((u64)tag << 32) | (u64)(uint32_t)c; }
                    else {
                    if (want1) w1[j] = load_range(a.stats, ra_r1, a.parity_stride, tag, c);
                    if (want2) w2[j] = load_range(a.stats, ra_r2, a.parity_stride, tag, c);
                    }
                }
            }
            if (wantb) {
#pragma unroll
                for (int j = 0; j < kResOwn; ++j) {
                    const int c = T.r0 + min(tid + j * kBlock, T.nr - 1);
                    if (DFQ_RES_ABLATE & 16) v2[j].a = v2[j].b = ((u64)tag << 32) | (u64)(uint32_t)c;
                    else v2[j] = load_range(a.stats, rb_r2, a.parity_stride, tag, c);
                }
            }
        };
        const bool pre = DFQ_RES_PREFETCH != 0;
        if (pre && (direct_r || direct_c)) fetch_top(0, kPreJ, direct_r, direct_c, false);
        bool have_b = false;
        {
            const int copy = blockIdx.x & 7;
            // (row statistics with a single producer per word are polled directly below: no counter)
            const u64* c1 = hasA ? (direct_r ? nullptr : cnt_line(a.cnt_r, T.a_layer, copy)) : (chain_start ? cnt_line(a.cnt_r, T.layer, copy) : nullptr);
            const u64 t1 = (u64)(hasA ? T.nt_a : T.nt_self) * round;
            const u64* c2 = (hasA && !direct_c) ? cnt_line(a.cnt_c, T.layer, copy) : nullptr;
            const u64* c3 = (hasB && !direct_b) ? cnt_line(a.cnt_c, T.b_layer, copy) : nullptr;
            const int got = res_wait(c1, t1, c2, (u64)T.nt_self * round, c3, (u64)T.nt_b * round, DFQ_RES_TOPWAIT != 0,
                                     prog_line(cold(a).prog, blockIdx.x & 7), (uint32_t)max(k - cold(a).spec, 0), a.err, sh_flag + 2, kResSpinLimit);
            if (!got) { failed = true; break; }
            if (got == 3) { stopped = true; break; }
            have_b = hasB && !direct_b && got == 2;          // counter-guarded words of phase 2 are complete: fetched with phase 1's
        }
        res_stamp<kTrace>(a, k, 15);
        {
            // A word whose producer arrived "relaxed" (see arrive) may still show the previous sweep's tag: read again.  Words
            // merged from several tiles were complete before their counter moved.  Each thread looks after its own words.
            // Phase 2's single-producer words (direct_b) ride along but are NOT waited for here: the row statistics this tile
            // publishes in phase 1 must not wait for them (phase 2 polls them itself).
            long tries = 0;
            bool first = true;
            for (;;) {
                bool ok = true;
                if (first) {
                    fetch_top(0, kPreJ, !(pre && direct_r), !(pre && direct_c), have_b || direct_b);
                    fetch_top(kPreJ, kResTab / kBlock, true, true, false);
                } else {
                    fetch_top(0, kResTab / kBlock, true, true, have_b || direct_b);
                }
                first = false;
#pragma unroll
                for (int j = 0; j < kResTab / kBlock; ++j)
                    if (j * kBlock < n_ch) ok = ok && tagged(w1[j], tag) && tagged(w2[j], tag);
                if (have_b) {
#pragma unroll
                    for (int j = 0; j < kResOwn; ++j) ok = ok && tagged(v2[j], tag);
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(DFQ_RES_NAP);
                ++tries;
                // the producer may have left: the loop has stopped (sweep k does not happen then, whatever this tile makes of it)
                if ((tries & 7) == 0 && (uint32_t)(ld_word(prog_line(cold(a).prog, blockIdx.x & 7)) >> 32) != 0u) { *sh_bad = 2; break; }
                if (tries > kResSpinLimit / 16) { atomicMax(a.err, 1ull); *sh_bad = 1; break; }
            }
        }
        // (the factor log: uniform base + an index the compiler cannot hoist -- see the checkpoint block)
        gfloat* const lg = (gfloat*)log_entry(k);
        int lt_i = tid;
        opaque(lt_i);
        const unsigned lt = (unsigned)lt_i;
        // ---- phase 1: s_A per (group, input channel) of the tile ----
        if (hasA) {
#pragma unroll
            for (int j = 0; j < kResTab / kBlock; ++j) {
                const int idx = tid + j * kBlock;
                if (j * kBlock < n_ch) {
                    float mn1, mx1, mn2, mx2, s, inv;
                    decode_range(w1[j], tag, mn1, mx1);
                    decode_range(w2[j], tag, mn2, mx2);
                    if (DFQ_RES_ABLATE & 2) { s = 1.0f; inv = 1.0f + 0.0f * (mn1 + mx1 + mn2 + mx2); }
                    else le_solve(range_of(mn1, mx1, p.signed_range), range_of(mn2, mx2, p.signed_range), p, s, inv);
                    if (idx < n_ch) { sh_inv[idx] = inv; if (!(DFQ_RES_ABLATE & 32)) lg[lt + (unsigned)(j * kBlock)] = inv; }       // (the log: fire-and-forget)
                    // closed-form column statistics: this sweep's extrema stay in the table (raw floats) for publish_cols_cf
                    if (cf_cols && idx < n_ch) { sh_col[2 * idx] = __float_as_uint(mn2); sh_col[2 * idx + 1] = __float_as_uint(mx2); }
                }
            }
            res_stamp<kTrace>(a, k, 1);
            if (hasB) for (int i = tid; i < 2 * T.nr; i += kBlock) sh_row[i] = 0u;
            __syncthreads();
            if (*sh_bad) break;                                   // (uniform; sorted out behind the loop) nothing is published from stale words
            if (hasB) {
                // row statistics of t = fl(w * 1/s_A) for relation B of this same sweep (t is not kept: phase 3 recomputes it)
                res_stamp<kTrace>(a, k, 12);
                if (DFQ_RES_PRIO) __builtin_amdgcn_s_setprio(3);
                if (!(DFQ_RES_ABLATE & 4)) lay.row_stats(T, G, v, true, false, sh_inv, sh_s, sh_row);
                if (DFQ_RES_PRIO) __builtin_amdgcn_s_setprio(0);
                res_stamp<kTrace>(a, k, 13);
                __syncthreads();
                publish_rows(a, T, rb_r1, sh_row, tag);
                res_stamp<kTrace>(a, k, 14);
                // (bit 2 of relax_r: every reader of these rows polls the tagged words themselves and no tile of the layer waits for
                // its siblings' rows -- nobody ever looks at the layer's row counter: no barrier, no arrival)
                if (!(T.relax_r & 4)) arrive(a.cnt_r, T.layer, !(T.relax_r & 1));
            }
        }
        if (!hasA) __syncthreads();                               // (phase 1 has its own barriers)
        if (*sh_bad) break;
        res_stamp<kTrace>(a, k, 2);
        // ---- phase 2: s_B per row ----
        if (hasB) {
            // a tile of complete rows already has its rows' statistics (sh_row: this sweep's phase 1, or the previous
            // sweep's phase 3 for a chain start); otherwise they are merged over the row block's tiles in global memory
            const bool own_wait = !rows_local && !chain_start;
            const bool cnt_b = !have_b && !direct_b;              // counter-guarded column statistics that were not there at the top
            if (own_wait || cnt_b) {
                const int copy = blockIdx.x & 7;
                const int got = res_wait(own_wait ? cnt_line(a.cnt_r, T.layer, copy) : nullptr, (u64)T.nt_self * round,
                                         cnt_b ? cnt_line(a.cnt_c, T.b_layer, copy) : nullptr, (u64)T.nt_b * round, nullptr, 0, false,
                                         prog_line(cold(a).prog, blockIdx.x & 7), 0u, a.err, sh_flag + 3, kResSpinLimit);
                if (!got) { failed = true; break; }
                if (got == 3) { stopped = true; break; }          // (nothing of sweep k has been applied)
            }
            if (!rows_local || !have_b) {
                // (direct_b: the words requested at the sweep's top are looked at first; only a thread whose words are not there
                // yet reads again)
                long tries = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < kResOwn; ++j) {
                        const int c = T.r0 + min(tid + j * kBlock, T.nr - 1);
                        if (DFQ_RES_ABLATE & 16) { v1[j].a = v1[j].b = v2[j].a = v2[j].b = ((u64)tag << 32) | (u64)(uint32_t)c; continue; }
                        if (!rows_local) v1[j] = load_range(a.stats, rb_r1, a.parity_stride, tag, c);
                        if (!have_b && !(direct_b && tries == 0 && tagged(v2[j], tag))) v2[j] = load_range(a.stats, rb_r2, a.parity_stride, tag, c);
                    }
#pragma unroll
                    for (int j = 0; j < kResOwn; ++j) ok = ok && (rows_local || tagged(v1[j], tag)) && (have_b || tagged(v2[j], tag));
                    if (ok) break;
                    __builtin_amdgcn_s_sleep(DFQ_RES_NAP);
                    ++tries;
                    if ((tries & 7) == 0 && (uint32_t)(ld_word(prog_line(cold(a).prog, blockIdx.x & 7)) >> 32) != 0u) { *sh_bad = 2; break; }
                    if (tries > kResSpinLimit / 16) { atomicMax(a.err, 1ull); *sh_bad = 1; break; }
                }
            }
#pragma unroll
            for (int j = 0; j < kResOwn; ++j) {
                const int i = tid + j * kBlock;
                if (i < T.nr) {
                    float mn1, mx1, mn2, mx2, s, inv;
                    if (rows_local) {
                        mn1 = slot_min(sh_row[2 * i]); mx1 = slot_max(sh_row[2 * i + 1]);
                        if (chain_start) { sh_row[2 * i] = 0u; sh_row[2 * i + 1] = 0u; }     // phase 3 accumulates the next sweep's into it
                    }
                    else decode_range(v1[j], tag, mn1, mx1);
                    decode_range(v2[j], tag, mn2, mx2);
                    if (DFQ_RES_ABLATE & 2) { inv = 1.0f; s = 1.0f + 0.0f * (mn1 + mx1 + mn2 + mx2); }
                    else le_solve(range_of(mn1, mx1, p.signed_range), range_of(mn2, mx2, p.signed_range), p, s, inv);
                    sh_s[i] = s;                                  // also applied to the [O] vectors below
                    if (!(DFQ_RES_ABLATE & 32)) lg[(unsigned)n_ch + lt + (unsigned)(j * kBlock)] = s;     // (the log: fire-and-forget)
                }
            }
        }
        res_stamp<kTrace>(a, k, 3);
        // ---- phase 3: w <- fl(fl(w / s_A) * s_B) (dfq.py:73 then :62, both rounded), |dW| and the statistics of the new values
        //      in ONE pass over the tile.  Applied at once: whether sweep k happens at all is found out later (see
        //      "speculation past the verdict"). ----
        __syncthreads();                                          // sh_s complete; sh_col is all zero since its publication, a chain start's sh_row since phase 2
        if (*sh_bad) break;                                       // a statistics word of phase 2 never showed this sweep's tag (or the loop stopped)
        if (chain_start && !rows_local) {                         // (never with the shapes pick_shape chooses: a chain start whose rows are cut)
            for (int i = tid; i < 2 * T.nr; i += kBlock) sh_row[i] = 0u;
            __syncthreads();
        }
        res_stamp<kTrace>(a, k, 8);
        double acc;
        if (DFQ_RES_PRIO) __builtin_amdgcn_s_setprio(3);
        if (DFQ_RES_ABLATE & 8) {
            acc = 0.0;
        } else if (Lay::kFusedCols && hasA) {
            if (DFQ_RES_ABLATE & 256) { double junk = lay.diff_and_cols(T, G, v, hasA, hasB, sh_inv, sh_s, sh_col, false); if (junk == -1.0) sh_s[0] = 0.0f; }   // the pass TWICE: its marginal cost
            if (DFQ_RES_ABLATE & 512) { lay.row_stats(T, G, v, hasA, hasB, sh_inv, sh_s, sh_row); }                  // + a read-only statistics pass
            acc = lay.diff_and_cols(T, G, v, hasA, hasB, sh_inv, sh_s, sh_col, true, !cf_cols);   // (closed form: the statistics were formed in phase 1)
        } else if (DFQ_RES_HOT && Lay::kFusedRows && chain_start) {
            acc = lay.update_rows(T, G, v, false, true, sh_inv, sh_s, sh_row);         // w <- w * s_B, |dW| and the new rows' statistics: one pass
        } else {
            acc = lay.template update<true>(T, G, v, hasA, hasB, sh_inv, sh_s);
            if (hasA && !cf_cols) lay.col_stats(T, G, v, false, false, sh_inv, sh_s, sh_col);      // of the values just written
            if (chain_start) lay.row_stats(T, G, v, false, false, sh_inv, sh_s, sh_row);
        }
        if (DFQ_RES_PRIO) __builtin_amdgcn_s_setprio(0);
        res_stamp<kTrace>(a, k, 9);
        __syncthreads();
        res_stamp<kTrace>(a, k, 10);
        // (DFQ_RES_LATE_ARRIVE: a STRICT arrival -- statistics merged from several tiles: the counter may move only once this tile's
        // atomics have been performed, a trip through the memory system -- could be made after the sweep's tail, which would then
        // run while the atomics are in flight.  Measured slower: see the switch.)
        const bool strict_c = DFQ_RES_LATE_ARRIVE && hasA && !(T.relax_c & 1) && !cf_cols, strict_r = DFQ_RES_LATE_ARRIVE && chain_start && !(T.relax_r & 1);
        if (hasA) {
            if (cf_cols) publish_cols_cf(a, T, G, ra_r2, sh_col, sh_inv, tag + 1u);
            else publish_cols(a, T, G, ra_r2, sh_col, tag + 1u);
            if (!strict_c) arrive(a.cnt_c, T.layer, !(T.relax_c & 1) && !cf_cols);
        }
        if (chain_start) {
            publish_rows(a, T, rb_r1, sh_row, tag + 1u);
            if (!strict_r) arrive(a.cnt_r, T.layer, !(T.relax_r & 1));
        }
        res_stamp<kTrace>(a, k, 4);
        // ---- convergence: one partial per tile (fixed butterfly + fixed wave order) as two tagged words; the reducer workgroup
        //      reads them until they carry k + 1, sums them per layer and draws the verdict -- off every tile's path ----
        {
            // (round 5: one partial per WAVE -- the fixed butterfly -- instead of one per tile: the four are added by the reducer in
            // wave order from 0.0, exactly what block_sum did here behind two barriers and a trip through the LDS)
            const double wsum = wave_sum(acc);
            if ((tid % kWave) == 0) {
                const auto& c = cold(a);
                u64* dst = (u64*)c.partials + (((int64_t)(k % c.part_ring) * c.n_tiles + T.slot) * (kBlock / kWave) + tid / kWave) * 2;
                const u64 bits = (u64)__double_as_longlong(wsum), tg = (u64)(k + 1) << 32;
                __hip_atomic_store(dst, tg | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 1, tg | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            res_stamp<kTrace>(a, k, 5);
        }
        if (kTrace && tid == 0 && k == 0) cold(a).trace[((int64_t)blockIdx.x * kTraceSweeps) * kTracePoints + 7] = ((long long)T.layer << 32) | (unsigned)T.nr << 16 | (unsigned)(T.nc & 0xffff);
#if defined(__HIP_DEVICE_COMPILE__)
        // where the workgroup runs (tuning aid): XCC_ID (hwreg 20) and HW_ID (hwreg 4: cu_id [11:8], sh_id [12], se_id [15:13])
        if (kTrace && tid == 0 && k == 1)
            cold(a).trace[((int64_t)blockIdx.x * kTraceSweeps + 1) * kTracePoints + 7] =
                ((long long)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u) << 32) | (long long)(__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)));
#endif
        if (hasB) {
#pragma unroll
            for (int j = 0; j < kResOwn; ++j) {
                const float sj = sh_s[min(tid + j * kBlock, T.nr - 1)];     // this thread's own rows (written by itself in phase 2)
                o_cum[j] = o_cum[j] * sj;                         // relation.py:20-24
                o_bnw[j] = o_bnw[j] * sj;                         // dfq.py:64-65
                o_bnb[j] = o_bnb[j] * sj;                         // dfq.py:67-68
                sh_ob1[tid + j * kBlock] = sh_ob1[tid + j * kBlock] * sj;   // dfq.py:70-71
            }
        }
        res_stamp<kTrace>(a, k, 6);
        if (strict_c) arrive(a.cnt_c, T.layer, true);
        if (strict_r) arrive(a.cnt_r, T.layer, true);
        res_stamp<kTrace>(a, k, 11);
    }
#undef cf_cols
    // a statistics spin left the loop through *sh_bad: 1 = abandoned, 2 = the loop has stopped (sweep k was not applied)
    __syncthreads();
    if (*sh_bad == 1) failed = true;
    if (failed) return;                     // this tile stores nothing (others may have: the run reports DFQ_ERR_STATE, see the header)
    // ---- how many of the k applied sweeps happen ----
    (void)stopped;
    const int keep = res_final(prog_line(cold(a).prog, blockIdx.x & 7), k, a.err, sh_flag, kResSpinLimit);
    if (keep < 0) return;
    if (keep < k) {
        // rollback: newest checkpoint at or before `keep` sweeps, then the logged factors of the sweeps up to `keep`
        __builtin_amdgcn_s_waitcnt(0);      // this workgroup's own log / checkpoint stores have been performed
        __syncthreads();
        if (tid == 0) {                     // statistics of the launch (tests, tuning): tiles that rolled back, sweeps undone
            u64* sts = prog_line(cold(a).prog, 8);
            atomicAdd(sts, 1ull);
            atomicAdd(sts + 1, (u64)(k - keep));
            atomicMax(sts + 2, (u64)(k - keep));
        }
        const int ckpt_every = cold(a).ckpt_every;
        const int c0 = (keep / ckpt_every) * ckpt_every;
        if (c0 == 0) {
            load_pristine();
        } else {
            const float* ck = ckpt_slot(c0 / ckpt_every);
#pragma unroll 1
            for (int u = 0; u < kResTileFloats / (4 * kBlock); ++u)
                *(fvec4*)(v + (u * kBlock + tid) * 4) = *(const gfvec4*)(ck + (u * kBlock + tid) * 4);
            if (owner) {
#pragma unroll
                for (int j = 0; j < kResOwn; ++j) {
                    const float* o = ck + kResTileFloats + (4 * j) * kBlock + tid;
                    o_cum[j] = o[0]; o_bnw[j] = o[kBlock]; o_bnb[j] = o[2 * kBlock]; sh_ob1[tid + j * kBlock] = o[3 * kBlock];
                }
            }
        }
        for (int q = c0; q < keep; ++q) {
            const float* lq = log_entry(q);
            __syncthreads();                                      // the previous replay step has read the tables
            for (int idx = tid; idx < n_ch; idx += kBlock) sh_inv[idx] = lq[idx];
            if (hasB) for (int i = tid; i < T.nr; i += kBlock) sh_s[i] = lq[n_ch + i];
            __syncthreads();
            (void)lay.template update<false>(T, G, v, hasA, hasB, sh_inv, sh_s);
            if (hasB) {
#pragma unroll
                for (int j = 0; j < kResOwn; ++j) {
                    const float sj = sh_s[min(tid + j * kBlock, T.nr - 1)];
                    o_cum[j] = o_cum[j] * sj; o_bnw[j] = o_bnw[j] * sj; o_bnb[j] = o_bnb[j] * sj;
                    sh_ob1[tid + j * kBlock] = sh_ob1[tid + j * kBlock] * sj;
                }
            }
        }
        __syncthreads();
    }
    if (keep == 0) return;                  // nothing happened: the tensors stay as they are
    // ---- all or nothing (round 5).  Nothing has been stored into the caller's tensors so far.  A tile stores only once EVERY
    //      tile of the launch has got this far (one arrival per tile, one more bounded wait): a tile that abandoned a wait never
    //      arrives, the others give up here in turn, and the launch leaves the network exactly as it found it -- which is what
    //      lets the host run the pass again on the streaming engine instead of reporting an undefined network (dfq_le_run).
    //      Tiles that do store count themselves, so the host can tell "nothing was stored" from "the last wait itself timed out
    //      in some tiles" (then, and only then, the network is undefined as before). ----
    if (DFQ_RES_COMMIT) {
        u64* commit = prog_line(cold(a).prog, 9);
        if (tid == 0) {
            atomicAdd(commit, 1ull);
            const u64 want = (u64)cold(a).n_tiles;
            // (this wait is at least 200 000 polls patient whatever DFQ_SPIN_LIMIT says: the tiles reach it up to `spec` sweeps
            // apart, and a tile that gave up here while the others go on to store is the one case that still leaves an undefined
            // network behind -- the tests' limit of ONE poll is meant for the waits of the loop)
            const long patience = kResSpinLimit > 200000 ? kResSpinLimit : 200000;
            long spins = 0;
            int ok = 1;
            while (__hip_atomic_load(commit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(DFQ_RES_NAP);
                ++spins;
                if (spins > patience ||
                    ((spins & 255) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull)) {
                    atomicMax(a.err, 1ull);
                    ok = 0;
                    break;
                }
            }
            if (ok) atomicAdd(prog_line(cold(a).prog, 10), 1ull);     // this tile stores
            *sh_flag = ok;
        }
        __syncthreads();
        if (!*sh_flag) return;
    }
    // ---- write the tile back (once) ----
    lay.store(T, v);
#pragma unroll
    for (int j = 0; j < kResOwn; ++j) {
        const int i = tid + j * kBlock;
        if (owner && i < T.nr) {
            const ResRel RB = cold(a).rels[T.relB];
            const int c = T.r0 + i;
            RB.s_cum[c] = o_cum[j];
            if (RB.bnw) RB.bnw[c] = o_bnw[j];
            if (RB.bnb) RB.bnb[c] = o_bnb[j];
            if (RB.b1) RB.b1[c] = sh_ob1[i];
        }
    }
}

// ---- the reducer: one extra workgroup (the launch's last) that holds no tile.  For sweep j = 0, 1, ... it reads the tiles'
//      partial sums until all carry j + 1, forms dfq.py:105-108 -- per layer float(mean |dW|) from the tiles' float64 partials
//      in a fixed order, summed over the layers in graph order like Python's sum -- advances the reference's (diff, count)
//      state machine (dfq.py:110-115) and publishes {sweeps that happen, once the loop has stopped : verdicts drawn} in eight
//      copies of the progress word.  Nobody waits for it inside a dependency chain. ----
__device__ __forceinline__ void res_reducer_body(const ResArgs& a, const LeParams& p, unsigned char* smem) {
    double* sh_d = (double*)smem;                                  // [kResMaxTiles] partial sums of one sweep
    double* sh_mean = sh_d + kResMaxTiles;                         // [kResMaxLayers]
    int* sh_bad = (int*)(sh_mean + kResMaxLayers);
    const int tid = threadIdx.x;
    const long kResSpinLimit = p.spin_limit;
    const auto& c = cold(a);
    const int n_tiles = c.n_tiles, n_layers = c.n_layers;
    LoopState st;
    {
        const LeState* s0 = c.state;
        st.diff = s0->diff; st.last_diff_tmp = s0->last_diff_tmp;
        st.count = s0->count; st.sweeps = s0->sweeps; st.done = 0;
    }
    int cap = c.n_sweeps;
    if (c.max_sweeps >= 0) cap = min(cap, c.max_sweeps - st.sweeps);
    if (tid == 0) *sh_bad = 0;
    __syncthreads();
    for (int j = 0; j < cap; ++j) {
        constexpr int kW = kBlock / kWave;                           // partials per tile: one per wave
        const u64* part = (const u64*)c.partials + (int64_t)(j % c.part_ring) * 2 * kW * n_tiles;
        const u64 want = (u64)(j + 1);
        for (int i = tid; i < n_tiles; i += kBlock) {
            long tries = 0;
            u64 hi[kW], lo[kW];
            for (;;) {
                bool ok = true;
#pragma unroll
                for (int w = 0; w < kW; ++w) { hi[w] = ld_word(part + 2 * (kW * i + w)); lo[w] = ld_word(part + 2 * (kW * i + w) + 1); }
#pragma unroll
                for (int w = 0; w < kW; ++w) ok = ok && (hi[w] >> 32) == want && (lo[w] >> 32) == want;
                if (ok) break;
                __builtin_amdgcn_s_sleep(DFQ_RES_NAP);
                if (++tries > kResSpinLimit ||
                    ((tries & 255) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull)) {
                    atomicMax(a.err, 1ull); *sh_bad = 1; break;
                }
            }
            double t = 0.0;                                           // wave order, from 0.0: block_sum's
#pragma unroll
            for (int w = 0; w < kW; ++w) t += __longlong_as_double((long long)((hi[w] << 32) | (lo[w] & 0xffffffffull)));
            sh_d[i] = t;
        }
        __syncthreads();
        if (*sh_bad) return;                                        // abandoned: the tiles give up through the error word
        for (int l = tid; l < n_layers; l += kBlock) {
            const ResLayerDiff L = c.layer_diff[l];
            const double s = ordered_sum(sh_d + L.tile_begin, L.n_tiles);              // fixed order
            // float(torch.mean(torch.abs(W - W_prev))): float32 mean, widened to double (dfq.py:108)
            sh_mean[l] = (L.n_tiles > 0) ? (double)(float)(s / L.n_elems) : 0.0;
        }
        __syncthreads();
        const double diff_tmp = ordered_sum(sh_mean, n_layers);                        // graph order, like Python's sum
        __syncthreads();                 // sh_d / sh_mean are reused
        // dfq.py:110-115
        if (fabs(st.diff - diff_tmp) > 1e-9) { st.count = 0; st.diff = diff_tmp; }
        else { st.count += 1; }
        if (tid == 0) {                         // (optional per-sweep record for a stopping rule that spans several plans, LeState::log)
            const LeState* s0 = c.state;
            if (s0->log && st.sweeps < s0->log_cap) s0->log[st.sweeps] = diff_tmp;
        }
        st.sweeps += 1;
        st.last_diff_tmp = diff_tmp;
        const bool go_on = (st.diff > c.converge_thres) && (st.count < c.converge_count) && (c.max_sweeps < 0 || st.sweeps < c.max_sweeps);
        st.done = go_on ? 0 : 1;
        if (tid < 8) {
            const u64 word = ((u64)(st.done ? (uint32_t)(j + 1) : 0u) << 32) | (u64)(uint32_t)(j + 1);
            __hip_atomic_store(prog_line(a.prog, tid), word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (st.done) break;
    }
    if (tid == 0) {
        LeState* o = c.state;
        o->diff = st.diff; o->last_diff_tmp = st.last_diff_tmp; o->count = st.count; o->sweeps = st.sweeps; o->done = st.done;
    }
}

// Everything a resident launch needs zeroed -- statistics words, counters, the progress word, the partial-sum ring -- and, for a
// run that starts a new loop, the loop state of dfq.py:81-82 and the plan's error word: ONE launch in front of the cooperative
// one (until round 4: a reset launch and a clear launch).
__global__ void res_prepare_kernel(ClearArgs a, LeState* state, int restart, double converge_thres, int converge_count, int max_sweeps,
                                   unsigned long long* err) {
    const long long step = (long long)gridDim.x * blockDim.x;
    for (int k = 0; k < 4; ++k)
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.words[k]; i += step) a.p[k][i] = 0u;
    if (restart && blockIdx.x == 0 && threadIdx.x == 0) {
        *err = 0ull;
        state->diff = 10.0;          // dfq.py:81
        state->count = 0;            // dfq.py:82
        state->sweeps = 0;
        state->last_diff_tmp = 0.0;
        const bool go_on = (10.0 > converge_thres) && (0 < converge_count) && (max_sweeps != 0);
        state->done = go_on ? 0 : 1;
    }
}

constexpr size_t kResSmemBytes = sizeof(float) * (kResTileFloats + kResTab + kResRows) + sizeof(uint32_t) * 2 * (kResRows + kResTab) + 64 +
                                 sizeof(float) * kResRows;
static_assert(3 * kResSmemBytes <= 160 * 1024, "three workgroups per CU");
static_assert(sizeof(double) * (kResMaxTiles + kResMaxLayers) + 64 <= kResSmemBytes, "the reducer stages a sweep's partial sums in its LDS");

enum { kLayGeneral = 0, kLayFixed = 1, kLayShort = 2 };

template <bool kTrace>      // 3 waves per SIMD: at most 168 registers (three workgroups per CU is what the plan counts on)
__global__ __launch_bounds__(kBlock, 3) void le_resident_kernel(ResArgs a, LeParams p) {
    DFQ_DYN_SMEM(smem);
    if (a.state->done) return;              // already stopped (uniform over the launch: written before it started)
    if ((int)blockIdx.x == a.n_tiles) { res_reducer_body(a, p, smem); return; }
    const ResTile T = a.tiles[blockIdx.x];
#if DFQ_RES_SPLIT_AB
#define DFQ_RES_BODY(LAY)                                                                     \
    do {                                                                                      \
        if (T.relA >= 0 && T.relB >= 0) res_tile_body<LAY, kTrace, 3>(a, p, T, smem);         \
        else if (T.relA >= 0) res_tile_body<LAY, kTrace, 1>(a, p, T, smem);                   \
        else res_tile_body<LAY, kTrace, 2>(a, p, T, smem);                                    \
    } while (0)
#else
#define DFQ_RES_BODY(LAY) res_tile_body<LAY, kTrace, -1>(a, p, T, smem)
#endif
    if (T.layout == kLayFixed) DFQ_RES_BODY(LayFixed);
    else if (T.layout == kLayShort) DFQ_RES_BODY(LayShort);
    else if (T.vec == 4) res_tile_body<LayGeneral<4>, kTrace, -1>(a, p, T, smem);       // (the slow fallback layouts stay generic)
    else res_tile_body<LayGeneral<1>, kTrace, -1>(a, p, T, smem);
#undef DFQ_RES_BODY
}

}  // namespace dfq

using namespace dfq;

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct dfq::LeResident {
    int n_tiles = 0, n_pl = 0, n_rels = 0, n_layers = 0;
    int spec = 2, ckpt_every = 8;    // speculation depth and checkpoint period of the launch (DFQ_RES_SPEC, DFQ_RES_CKPT)
    float* d_log = nullptr;          // [ckpt_every + spec][log_total]
    float* d_ckpt = nullptr;         // [2][n_tiles][kCkptFloats]
    int64_t log_total = 0;
    ResTile* d_tiles = nullptr;
    ResRel* d_rels = nullptr;
    ResLayerDiff* d_layer_diff = nullptr;
    u64* d_stats = nullptr;
    int64_t stat_words = 0;          // u64 words of both arenas, both parities
    int64_t parity_stride = 0;
    u64* d_sync = nullptr;           // cnt_r | cnt_c | done | seq
    size_t sync_words = 0;
    double* d_partials = nullptr;
    int64_t elements = 0;            // floats held in LDS
};

namespace {

int ceil_div_i(int a, int b) { return (a + b - 1) / b; }

struct Shape { int tr, tc; };

int pow2_ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }
// groups (blocks of `go` rows) the row blocks of height tr of an R-row layer span at most -- exactly, not (tr / go + 1): with
// one group (go == R), or row blocks that divide a group, a tile never straddles a boundary, and the bound decides whether a
// tile of complete 960-float rows fits the LDS table at all
int max_groups(int R, int tr, int go) {
    static const bool exact = !(getenv("DFQ_RES_EXACT_GROUPS") && getenv("DFQ_RES_EXACT_GROUPS")[0] == '0');   // A/B switch
    if (!exact) return (tr + go - 1) / go + 1;
    int worst = 1;
    for (int r0 = 0; r0 < R; r0 += tr) {
        const int r1 = std::min(R, r0 + tr) - 1;
        worst = std::max(worst, r1 / go - r0 / go + 1);
    }
    return worst;
}
int layout_of(int vec, int row_len, int nc) {
    if (vec == 1 && row_len <= 32 && nc == row_len) return kLayShort;
    if (vec == 4 && nc / 4 <= kBlock) return kLayFixed;
    return kLayGeneral;
}

// [tr x tc] tiling of an [R x C] layer (a tile holds kResTileFloats floats).  Cost = global statistics atomics per sweep (row
// statistics are merged over the column blocks, column statistics over the row blocks) -- complete rows are favoured
// for layers with row duty: no merge, and the tile does not wait for its own publication -- with a heavy penalty for
// tiles that fall back to the general layout (LDS atomics per element).
// elements a tile may hold: kResTileFloats, or fewer with DFQ_RES_TILE_FLOATS (TESTS: small networks then get layers of several
// tiles -- statistics merged over row blocks, strict arrivals -- on the CPU emulation; multiples of 1024)
int tile_capacity() {
    const char* e = getenv("DFQ_RES_TILE_FLOATS");                          // (read per plan: tests switch it between plans)
    const int v = (e && atoi(e) > 0) ? atoi(e) : kResTileFloats;
    return std::max(4 * kBlock, std::min(kResTileFloats, v / (4 * kBlock) * (4 * kBlock)));
}

Shape pick_shape(int R, int C, int vec, int khkw, int go, int i2g, bool need_row, bool need_col, int short_rpt) {
    const int ns4 = tile_capacity() / (4 * kBlock);         // float4 slots per thread
    if (vec == 1 && C <= 32) {               // thread-per-row tiles: complete rows; `short_rpt` rows per thread if they are <= 16 floats
        int tr = std::min(R, kBlock * (C <= 16 ? std::min(short_rpt, kResOwn) : 1));
        if (need_row) tr = std::min(tr, kResRows);
        if (need_col) while (tr > 1 && max_groups(R, tr, go) * std::min((C + khkw - 1) / khkw + 1, i2g) > kResTab) tr = (tr + 1) / 2;
        return Shape{tr, C};
    }
    Shape best{0, 0};
    double best_cost = 1e300;
    std::vector<int> cands;
    for (int tc = vec; tc < C && tc <= vec * kBlock; tc *= 2) cands.push_back(tc);
    cands.push_back(C);
    for (int tc : cands) {
        if (tc % vec) continue;
        const int lay = layout_of(vec, C, tc);
        int tr;
        if (lay == kLayFixed) tr = ns4 * (kBlock / pow2_ceil(tc / 4));      // rows per slot x slots
        else tr = tile_capacity() / tc;
        tr = std::min(R, tr);
        if (tr < 1) continue;
        if (need_row) tr = std::min(tr, kResRows);
        // LDS table of a tile with column duty: (groups spanned by its rows) x (input channels spanned by its columns)
        const int nci = std::min((tc + khkw - 1) / khkw + 1, i2g);
        if (need_col) {
            while (tr > 1 && max_groups(R, tr, go) * nci > kResTab) tr = (tr + 1) / 2;
            if (max_groups(R, tr, go) * nci > kResTab) continue;
        }
        const int n_rb = ceil_div_i(R, tr), n_cb = ceil_div_i(C, tc);
        // Measured (tools/trace_resident.py): a row merge over column blocks puts one more hand-off (publish -> counter ->
        // re-read, 2-3 us) on the tile's own critical cycle, every sweep; the column statistics are consumed a sweep later and
        // their atomics (a few per thread) vanish in the publication.
        double cost = (double)n_rb * n_cb * 2.0;                        // a tile is a workgroup of a bounded supply
        if (need_row) cost += (double)R * (n_cb > 1 ? 4.0 * n_cb : 0.25);
        if (need_col) cost += (double)(C / khkw) * n_rb * 0.05;
        const int rem = C % tc;
        if (lay == kLayGeneral) cost += 1e7;
        if (rem && layout_of(vec, C, rem) == kLayGeneral) cost += 1e7 * rem / (double)C;
        if (cost < best_cost) { best_cost = cost; best = Shape{tr, tc}; }
    }
    return best;
}

}  // namespace

namespace dfq {

void le_resident_destroy(LeResident* r) {
    if (!r) return;
    if (r->d_tiles) dfq::dev_free(r->d_tiles);
    if (r->d_rels) dfq::dev_free(r->d_rels);
    if (r->d_layer_diff) dfq::dev_free(r->d_layer_diff);
    if (r->d_stats) dfq::dev_free(r->d_stats);
    if (r->d_sync) dfq::dev_free(r->d_sync);
    if (r->d_partials) dfq::dev_free(r->d_partials);
    if (r->d_log) dfq::dev_free(r->d_log);
    if (r->d_ckpt) dfq::dev_free(r->d_ckpt);
    delete r;
}

int le_resident_tiles(const LeResident* r) { return r ? r->n_tiles : 0; }
// of the LAST launch: out[0] tiles that rolled back, out[1] sweeps undone in total, out[2] most sweeps undone by one tile,
// out[3] speculation depth, out[4] checkpoint period.  Synchronises `st`.
int le_resident_stats(const LeResident* r, hipStream_t st, int64_t* out5) {
    if (!r || !out5) return fail_arg("le_resident_stats: bad argument");
    u64 w[3] = {0, 0, 0};
    DFQ_HIP_TRY(hipMemcpyAsync(w, r->d_sync + (size_t)(16 * r->n_pl + 8) * kResStride, sizeof(w), hipMemcpyDeviceToHost, st));
    DFQ_HIP_TRY(hipStreamSynchronize(st));
    out5[0] = (int64_t)w[0]; out5[1] = (int64_t)w[1]; out5[2] = (int64_t)w[2]; out5[3] = r->spec; out5[4] = r->ckpt_every;
    return DFQ_OK;
}
// tiles of the LAST launch that stored their result (0 after an abandoned wait in front of the commit: the network is untouched).
// Synchronises `st`.
int le_resident_stored_tiles(const LeResident* r, hipStream_t st, int64_t* out) {
    if (!r || !out) return fail_arg("le_resident_stored_tiles: bad argument");
    u64 w = 0;
    DFQ_HIP_TRY(hipMemcpyAsync(&w, r->d_sync + (size_t)(16 * r->n_pl + 10) * kResStride, sizeof(w), hipMemcpyDeviceToHost, st));
    DFQ_HIP_TRY(hipStreamSynchronize(st));
    *out = (int64_t)w;
    return DFQ_OK;
}
int le_resident_trace_words(const LeResident* r) { return r ? r->n_tiles * kTraceSweeps * kTracePoints : 0; }
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
    std::vector<ResTile> tiles;
    std::vector<int> tile_begin(n_layers, 0), tile_count(n_layers, 0);
    const bool relaxed_ok = !(getenv("DFQ_RES_RELAXED") && getenv("DFQ_RES_RELAXED")[0] == '0');   // A/B switch
    int occ = 0;
    if (kResSmemBytes > 48 * 1024 &&
        hipFuncSetAttribute((const void*)le_resident_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kResSmemBytes) != hipSuccess)
        return refuse("dynamic shared memory size refused");
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)le_resident_kernel<false>, kBlock, kResSmemBytes) != hipSuccess || occ < 1)
        return refuse("occupancy query failed");
    // every workgroup must be resident: LDS bounds it (tile + tables), the API's answer is exact for that.  A quarter of
    // the slots stays free: a launch sized to exactly the occupancy limit (le_sweep_kernel, dfq_le.hip) never became fully
    // resident while a second stream kept the chip busy.
    const int cap_tiles = std::min(occ, 3) * cus * 3 / 4;
    // Thread-per-row tiles (depthwise layers) hold one row per thread when the chip has workgroups to spare, else two: the
    // depthwise layer of the longest chain sits on the sweep's critical cycle, and with one row per thread each of its
    // phases takes half as long (MobileNetV2: 571 instead of 559 workgroups).
    const char* spe = getenv("DFQ_RES_SHORT_RPT");
    for (int short_rpt = (spe && atoi(spe) == 2) ? 2 : 1; short_rpt <= 2; ++short_rpt) {
    tiles.clear();
    for (int l = 0; l < n_layers; ++l) {
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
        const Shape sh = pick_shape(R, C, vec, L.khkw, go, i2g, relB >= 0, relA >= 0, short_rpt);
        if (sh.tr < 1) return refuse("a layer does not tile");
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
                T.layout = layout_of(vec, C, T.nc);
                // single-producer statistics (see arrive): a row lives in one tile when the layer has one column block; an input
                // channel (rows of its group x its khkw taps) when no row block cuts a group and no column block cuts the taps
                T.relax_r = (relaxed_ok && n_cb == 1) ? 1 : 0;
                T.relax_c = (relaxed_ok && (n_rb == 1 || sh.tr % go == 0) && (n_cb == 1 || sh.tc % L.khkw == 0)) ? 1 : 0;
                tiles.push_back(T);
            }
        tile_count[l] = n_rb * n_cb;
    }
    if ((int)tiles.size() + 1 <= cap_tiles && (int)tiles.size() <= kResMaxTiles) break;
    }
    if ((int)tiles.size() + 1 > cap_tiles || (int)tiles.size() > kResMaxTiles)       // + 1: the reducer workgroup
        return refuse("the network does not fit the chip's LDS: " + std::to_string(tiles.size()) + " tiles > " +
                      std::to_string(std::min(cap_tiles - 1, kResMaxTiles)) + " resident workgroups");
    for (size_t i = 0; i < tiles.size(); ++i) {
        const ResTile& T = tiles[i];
        int64_t fp;
        if (T.layout == kLayFixed) {
            const int rps = kBlock / pow2_ceil(T.nc / 4);
            fp = (int64_t)ceil_div_i(T.nr, rps) * kBlock * 4;
        } else if (T.layout == kLayShort) {
            fp = (int64_t)ceil_div_i(T.nr, kBlock) * T.row_len * kBlock;
        } else {
            fp = (int64_t)ceil_div_i(T.nr * (T.nc / T.vec), kBlock) * kBlock * T.vec;
        }
        if (fp > kResTileFloats) return refuse("internal: a tile exceeds the LDS tile");
    }
    {   // bit 1 of relax_r: relation A's first layer publishes single-producer row statistics
        std::vector<int> rel_r(n_pl, 0);
        for (const ResTile& T : tiles) rel_r[T.layer] = T.relax_r & 1;
        const char* de = getenv("DFQ_RES_DIRECT");                     // A/B switch: 0 = always wait for the counter first
        const bool direct = !(de && de[0] == '0');
        for (ResTile& T : tiles) if (direct && T.a_layer >= 0 && rel_r[T.a_layer]) T.relax_r |= 2;
        // bit 2 of relax_r: the layer's row counter has no reader -- its rows have one producer each (bit 0: so no tile waits for
        // siblings' rows either), every consumer polls the words directly (bit 1 on ALL tiles of relation B's second layer), and the
        // layer is not a chain start (those pace themselves on their own row counter)
        {
            std::vector<int> all_direct(n_pl, 1);
            for (const ResTile& T : tiles) if (T.a_layer >= 0 && !(T.relax_r & 2)) all_direct[T.a_layer] = 0;
            for (ResTile& T : tiles)
                if (direct && (T.relax_r & 1) && T.relA >= 0 && T.relB >= 0 && all_direct[T.layer]) T.relax_r |= 4;
        }
        // bit 1 of relax_c: relation B's second layer publishes single-producer column statistics
        std::vector<int> rel_c(n_pl, 0);
        for (const ResTile& T : tiles) rel_c[T.layer] = T.relax_c & 1;
        for (ResTile& T : tiles) {
            if (direct && T.b_layer >= 0 && rel_c[T.b_layer]) T.relax_c |= 2;
            if (direct && (T.relax_c & 1)) T.relax_c |= 4;
        }
        // bit 3 of relax_c: closed-form column statistics for the chain ends.  OPT-IN (DFQ_RES_CF=1): bit-exact (tests: the
        // 'resident-cf' engine), and measured no faster on the MI355X -- MobileNetV2 0.585-0.593 ms with, 0.580-0.589 without,
        // DeepLab 0.343-0.348 vs 0.344-0.361 (profiles/r06_experiments.txt): the classifier's 160 tiles are not on the sweep's
        // critical cycle, which runs through the chains' hand-offs
        const char* ce = getenv("DFQ_RES_CF");
        if (relaxed_ok && ce && ce[0] == '1')
            for (ResTile& T : tiles) if (T.relA >= 0 && T.relB < 0) T.relax_c |= 8;
    }
    for (ResTile& T : tiles) {
        // tiles per layer (by paired-layer index)
        for (int l = 0; l < n_layers; ++l) {
            if (pl_of[l] == T.layer) T.nt_self = tile_count[l];
            if (pl_of[l] >= 0 && pl_of[l] == T.a_layer) T.nt_a = tile_count[l];
            if (pl_of[l] >= 0 && pl_of[l] == T.b_layer) T.nt_b = tile_count[l];
        }
    }
    // ---- launch order.  Workgroups are dispatched round-robin over the XCDs and then over the CUs, so workgroups b, b + 256 and
    //      b + 512 share a CU -- and its instruction issue.  The tiles at the END of the longest chain are the ones whose phases
    //      sit on the sweep's critical cycle: they go first (one per CU), the tiles of early layers, which mostly
    //      wait for the verdict, fill the second and third slot.  DFQ_RES_ORDER=0 keeps layer order. ----
    // a tile's share of one entry of the factor log: 1/s_A per (group, input channel) it spans, then s_B per row
    int64_t log_total = 0;
    for (size_t i = 0; i < tiles.size(); ++i) {
        ResTile& T = tiles[i];
        T.slot = (int32_t)i;
        int n_ch = 0;
        if (T.relA >= 0) {
            const int i0 = T.c0 / T.khkw, nci = (T.c0 + T.nc - 1) / T.khkw - i0 + 1;
            const int g_lo = T.r0 / T.go, g_n = (T.r0 + T.nr - 1) / T.go - g_lo + 1;
            n_ch = g_n * nci;
            if (n_ch > kResTab) return refuse("internal: a tile's factor table exceeds the LDS table");
        }
        T.log_off = (int32_t)log_total;
        log_total += (n_ch + (T.relB >= 0 ? T.nr : 0) + 63) / 64 * 64;
    }
    {
        const char* oe = getenv("DFQ_RES_ORDER");
        if (!(oe && oe[0] == '0')) {
            // depth of a paired layer in its chain (relations in list order: a second layer is one deeper than its first)
            std::vector<int> depth(n_pl, 0);
            for (int q = 0; q < n_relations; ++q)
                depth[pl_of[relations[q].second]] = std::max(depth[pl_of[relations[q].second]], depth[pl_of[relations[q].first]] + 1);
            std::stable_sort(tiles.begin(), tiles.end(), [&](const ResTile& x, const ResTile& y) {
                if (depth[x.layer] != depth[y.layer]) return depth[x.layer] > depth[y.layer];
                return x.slot < y.slot;
            });
        }
    }
    LeResident* r = new LeResident();
    r->n_tiles = (int)tiles.size(); r->n_pl = n_pl; r->n_rels = n_relations; r->n_layers = n_layers; r->elements = total;
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
    r->sync_words = (size_t)(16 * n_pl + 11) * kResStride;                  // counters [2][n_pl][8] | progress word [8] | rollback statistics | commit arrivals | tiles that stored
    {   // speculation depth / checkpoint period (A/B switches)
        const char* se = getenv("DFQ_RES_SPEC");
        const char* ce = getenv("DFQ_RES_CKPT");
        if (se && atoi(se) >= 0) r->spec = atoi(se);
        if (ce && atoi(ce) > 0) r->ckpt_every = atoi(ce);
        r->spec = std::min(r->spec, 64);
        r->ckpt_every = std::min(std::max(r->ckpt_every, std::max(1, r->spec)), 64);   // spec <= ckpt_every: two checkpoint buffers suffice
    }
    r->log_total = std::max<int64_t>(log_total, 64);
    bool ok = dfq::dev_malloc((void**)&r->d_tiles, sizeof(ResTile) * tiles.size()) == hipSuccess &&
              dfq::dev_malloc((void**)&r->d_rels, sizeof(ResRel) * n_relations) == hipSuccess &&
              dfq::dev_malloc((void**)&r->d_layer_diff, sizeof(ResLayerDiff) * n_layers) == hipSuccess &&
              dfq::dev_malloc((void**)&r->d_stats, sizeof(u64) * (size_t)r->stat_words) == hipSuccess &&
              dfq::dev_malloc((void**)&r->d_sync, sizeof(u64) * r->sync_words) == hipSuccess &&
              dfq::dev_malloc((void**)&r->d_partials, sizeof(double) * 2 * (kBlock / kWave) * (size_t)(r->spec + 2) * tiles.size()) == hipSuccess &&
              dfq::dev_malloc((void**)&r->d_log, sizeof(float) * (size_t)(r->ckpt_every + r->spec) * (size_t)r->log_total) == hipSuccess &&
              dfq::dev_malloc((void**)&r->d_ckpt, sizeof(float) * 2 * tiles.size() * (size_t)kCkptFloats) == hipSuccess &&
              hipMemcpy(r->d_tiles, tiles.data(), sizeof(ResTile) * tiles.size(), hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(r->d_rels, hr.data(), sizeof(ResRel) * n_relations, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(r->d_layer_diff, ld.data(), sizeof(ResLayerDiff) * n_layers, hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { le_resident_destroy(r); return refuse("device allocation failed"); }
    return r;
}

int le_resident_enqueue(LeResident* r, const dfq_le_config* cfg, LeState* d_state, unsigned long long* d_err, int n_sweeps,
                        hipStream_t st, long long* d_trace, int restart) {
    if (!r || !cfg || !d_state || !d_err) return fail_arg("le_resident_enqueue: bad argument");
    if (n_sweeps <= 0 && !restart) return DFQ_OK;
    // every launch is self-contained: statistics are re-derived from the weights it loads, tags and counters start at zero
    {
        ClearArgs ca;
        void* ps[4] = {r->d_stats, r->d_sync, r->d_partials, nullptr};
        const size_t bs[4] = {sizeof(u64) * (size_t)r->stat_words, sizeof(u64) * r->sync_words,
                              sizeof(double) * 2 * (kBlock / kWave) * (size_t)(r->spec + 2) * (size_t)r->n_tiles, 0};
        long long most = 0;
        for (int k = 0; k < 4; ++k) {
            ca.p[k] = (uint32_t*)ps[k];
            ca.words[k] = (ps[k] && n_sweeps > 0) ? (long long)(bs[k] / 4) : 0;
            most = std::max(most, ca.words[k]);
        }
        const int grid = (int)std::max<long long>(1, std::min<long long>((most + 1023) / 1024, 512));
        hipLaunchKernelGGL(res_prepare_kernel, dim3(grid), dim3(256), 0, st, ca, d_state, restart, cfg->converge_thres,
                           (int)cfg->converge_count, (int)cfg->max_sweeps, d_err);
        DFQ_CHECK_LAUNCH();
    }
    if (n_sweeps <= 0) return DFQ_OK;
    ResArgs a;
    memset(&a, 0, sizeof(a));
    a.tiles = r->d_tiles; a.rels = r->d_rels; a.layer_diff = r->d_layer_diff;
    a.stats = r->d_stats; a.parity_stride = r->parity_stride;
    a.cnt_r = r->d_sync;
    a.cnt_c = r->d_sync + (size_t)8 * r->n_pl * kResStride;
    a.prog = r->d_sync + (size_t)16 * r->n_pl * kResStride;
    a.log = r->d_log; a.ckpt = r->d_ckpt; a.log_total = r->log_total;
    a.spec = r->spec; a.ckpt_every = r->ckpt_every; a.log_ring = r->ckpt_every + r->spec; a.part_ring = r->spec + 2;
    a.err = d_err;
    a.partials = r->d_partials;
    a.state = d_state;
    a.n_tiles = r->n_tiles; a.n_layers = r->n_layers;
    a.n_sweeps = n_sweeps;
    a.max_sweeps = cfg->max_sweeps;
    a.converge_count = cfg->converge_count;
    a.converge_thres = cfg->converge_thres;
    a.trace = d_trace;
    const LeParams q = make_params(cfg);
    SpinGuard guard(st);
    if (d_trace) {
        if (kResSmemBytes > 48 * 1024)
            DFQ_HIP_TRY(hipFuncSetAttribute((const void*)le_resident_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kResSmemBytes));
        DFQ_LAUNCH_RESIDENT_PLAIN(le_resident_kernel<true>, dim3(r->n_tiles + 1), dim3(kBlock), kResSmemBytes, st, a, q);
    } else {
        DFQ_LAUNCH_RESIDENT_PLAIN(le_resident_kernel<false>, dim3(r->n_tiles + 1), dim3(kBlock), kResSmemBytes, st, a, q);
    }
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

}  // namespace dfq
