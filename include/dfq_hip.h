/*
 * dfq_hip.h -- C ABI of libdfq_hip.so, the MI355X (gfx950) engine behind the DFQ calibration
 * hot path of jakc4103/DFQ.
 *
 * This is the drop-in boundary.  The reference is a pure-Python program whose calibration passes
 * are Python loops over channels issuing eager torch-CPU ops; it has no FFI of its own.  Each
 * entry point below replaces one such Python function body (cited as file:line of the reference)
 * and is what a maintainer of the reference would bind from Python with ctypes (see
 * INTEGRATION.md).  The Python package `dfq_amd` is exactly such a binding, keeping the
 * reference's call surface (`cross_layer_equalization`, `bias_correction`,
 * `transform_quant_layer`, `quantize`/`UniformQuantize`, ...).
 *
 * Conventions
 *   - every pointer marked "device" is a HIP device pointer to contiguous float32 (or the stated
 *     type) owned by the caller (in practice: torch-ROCm tensors);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are
 *     asynchronous on that stream unless the comment says "synchronises";
 *   - return value: 0 on success, a negative code on failure; dfq_last_error() gives the
 *     thread-local message.  Nothing aborts, nothing falls back to the CPU;
 *   - weights are OIHW contiguous ([O, I/g, kH, kW]; Linear is [O, I] with kH*kW = 1), float32,
 *     exactly the layout torch gives `nn.Conv2d.weight` / `nn.Linear.weight`.
 *
 * Arithmetic contract (DESIGN.md "Numerics"): all float32 operations are IEEE-754 correctly
 * rounded single operations (no FMA contraction, no fast-math, round-half-even for the
 * quantiser), scalars that the reference computes as Python doubles are computed in float64 on
 * the device.  Integer codes of the fake-quant round trip are bit-exact against the reference.
 */
#ifndef DFQ_HIP_H
#define DFQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFQ_HIP_VERSION 100 /* 0.1.0 */

/* error codes */
#define DFQ_OK 0
#define DFQ_ERR_ARG (-1)     /* invalid argument / unsupported geometry            */
#define DFQ_ERR_HIP (-2)     /* a HIP runtime call failed (message has the details) */
#define DFQ_ERR_STATE (-3)   /* object used in the wrong state                      */
#define DFQ_ERR_ABANDONED (-4) /* a workgroup gave up a bounded in-launch wait (DFQ_SPIN_LIMIT): the tensors the run
                                  rewrites are undefined unless the entry point says otherwise; see *_set_safe_mode */

int dfq_version(void);
const char* dfq_last_error(void);
/* number of visible HIP devices, or a negative error */
int dfq_device_count(void);
/* Destroyed plans park their small device blocks (descriptor tables, statistics arenas: <= DFQ_POOL_MB, default 512 MB per
 * process) in per-device free lists OUTSIDE the caller's allocator so that the next plan costs no hipMalloc.  This returns
 * them to the driver (hipFree) and reports how many bytes that was -- for a host that is about to need the memory itself
 * (torch's caching allocator cannot see these blocks).  No reference counterpart (the reference allocates nothing). */
long long dfq_pool_trim(void);

/* ------------------------------------------------------------------------------------------
 * Layer / relation tables shared by the plans
 * ---------------------------------------------------------------------------------------- */

/* One "targ layer" (Conv2d / Linear family; dfq.py:106, main_cls.py:137-145). */
typedef struct dfq_layer {
    float* weight;         /* device [out_ch, in_per_group, khkw]                       */
    float* bias;           /* device [out_ch] or NULL                                    */
    int32_t out_ch;        /* O                                                          */
    int32_t in_per_group;  /* I / groups                                                 */
    int32_t khkw;          /* kH * kW (1 for Linear)                                     */
    int32_t groups;        /* conv groups (1 for Linear); informational                  */
} dfq_layer;

/* One equalisation pair (utils/relation.py:5-27): layers[first] -> (BN, ReLU...) -> layers[second]. */
typedef struct dfq_relation {
    int32_t first;         /* index into the dfq_layer table                             */
    int32_t second;
    float* bn_weight;      /* device [O1]  BN proxy fake_weight (dfq.py:64-65) or NULL   */
    float* bn_bias;        /* device [O1]  BN proxy fake_bias   (dfq.py:67-68) or NULL   */
    float* scale_cum;      /* device [O1]  cumulative S of Relation.set_scale_vec
                              (relation.py:20-24); must hold 1.0f before the first sweep */
} dfq_relation;

/* ------------------------------------------------------------------------------------------
 * Cross-layer equalisation -- replaces dfq.py:28-75 (_layer_equalization) and the sweep loop
 * dfq.py:78-117 (cross_layer_equalization)
 * ---------------------------------------------------------------------------------------- */
typedef struct dfq_le_plan dfq_le_plan;   /* opaque */

typedef struct dfq_le_config {
    float s_lo, s_hi;          /* s_range (dfq.py:78 default [1e-8, 1e8]) rounded to float32  */
    float inv_lo, inv_hi;      /* float32(1.0 / s_range[k]) computed in double by the caller  */
    int32_t hi_gt_lo;          /* Python `s_range[1] > s_range[0]` (double compare)           */
    float eps;                 /* dfq.py:58 eps                                               */
    int32_t signed_range;      /* 0: max-min (dfq.py:54-55), 1: max|.| (dfq.py:50-51)         */
    double converge_thres;     /* dfq.py:83                                                   */
    int32_t converge_count;    /* dfq.py:83                                                   */
    int32_t max_sweeps;        /* extension: stop after this many sweeps; <0 = reference loop */
} dfq_le_config;

typedef struct dfq_le_result {
    int32_t sweeps;            /* sweeps executed                                             */
    int32_t stall_count;       /* `count` of dfq.py:110-115 at exit                           */
    double diff;               /* `diff` of dfq.py:110-115 at exit                            */
    double last_diff_tmp;      /* diff_tmp of the last sweep                                  */
} dfq_le_result;

/* Builds the device-side work list: dependency levels of the relation list (Gauss-Seidel order of
 * dfq.py:85 is preserved: of two relations sharing a layer the later one's tiles wait, inside the sweep's
 * single launch, for the tiles of the earlier one they depend on), channel tiles, the per-layer
 * convergence-diff bookkeeping.  `layers` / `relations` are host arrays and
 * are copied.  Allocates a few small device buffers (statistics words, partial sums, descriptors);
 * the weights are processed where they are.  Synchronises. */
int dfq_le_plan_create(const dfq_layer* layers, int32_t n_layers,
                       const dfq_relation* relations, int32_t n_relations,
                       dfq_le_plan** out_plan);
/* Batched form: several independent networks in one plan.  `layer_net[l]` in [0, n_nets) is the network
 * of layer l; layers and relations must be listed network by network.  Every launch then covers all
 * networks (relation descriptors come from a table in global memory instead of the kernarg), each
 * network keeps its own loop state, so networks may converge after different sweep counts.  The sweep
 * loop of the batch runs until every network has stopped. */
int dfq_le_plan_create_batch(const dfq_layer* layers, int32_t n_layers, const int32_t* layer_net, int32_t n_nets,
                             const dfq_relation* relations, int32_t n_relations, dfq_le_plan** out_plan);
/* Replicated form: `n_nets` networks of ONE architecture whose tensors lie at the same offsets from a per-network base
 * address (a batch allocated as one arena with a fixed stride per network -- dfq_amd/arena.py).  `layers` / `relations`
 * describe the FIRST network (its own addresses, indices relative to it); network n's tensors are those addresses moved by
 * bases[n] - bases[0] bytes.  The host then builds one network's tables however large the batch is; everything else is as
 * dfq_le_plan_create_batch.  (Replaces, for a batch, the per-network graph walk of dfq.py:78-82.) */
int dfq_le_plan_create_replicated(const dfq_layer* layers, int32_t n_layers, const dfq_relation* relations, int32_t n_relations,
                                  const void* const* bases, int32_t n_nets, dfq_le_plan** out_plan);
void dfq_le_plan_destroy(dfq_le_plan* plan);
int32_t dfq_le_plan_nets(const dfq_le_plan* plan);

/* introspection (tests, bench byte accounting).  levels: equalisation launches per sweep -- 1 (the whole sweep is one
 * launch whose workgroups wait for the tiles they depend on) or, with DFQ_LE_MERGED=0 in the environment at plan
 * creation, one per dependency level; depth: number of dependency levels of the relation list */
/* 1 when the plan's networks are alike and laid out back to back (a batch from one template): the convergence launch then derives
 * a network's descriptor from its workgroup index instead of fetching it. */
int32_t dfq_le_plan_uniform(const dfq_le_plan* plan);
int32_t dfq_le_plan_levels(const dfq_le_plan* plan);
int32_t dfq_le_plan_depth(const dfq_le_plan* plan);
int64_t dfq_le_plan_paired_elements(const dfq_le_plan* plan);   /* sum over relations of n1+n2  */
/* per sweep: elements read AND written (8 B each) / elements only read by the statistics pass over
 * interior layers (4 B each); algorithmic bytes of a sweep = 8 * rw + 4 * ro */
int64_t dfq_le_plan_rw_elements(const dfq_le_plan* plan);
int64_t dfq_le_plan_ro_elements(const dfq_le_plan* plan);
/* Deferred stores of the streaming engine (DFQ_LE_DEFER = depth D at plan creation: 1 = off, 2 or 4; default 4 for batched
 * plans, 1 for a single network): a layer
 * that is only ever scaled one way is read every sweep but written every D-th one, the sweeps in between re-derive its
 * values from the stored ones and the remembered factors -- the same float32 operations, bit-identical results; every
 * enqueue call ends by bringing the weights up to date.  deferred_elements (a part of rw_elements) are those layers'
 * elements: bytes per sweep as executed, averaged over D sweeps = 8 * rw + 4 * ro - 4 * deferred * (D - 1) / D. */
int64_t dfq_le_plan_deferred_elements(const dfq_le_plan* plan);
int32_t dfq_le_plan_defer_depth(const dfq_le_plan* plan);
/* Free-running segments of the streaming engine (dfq_le_cf.hpp; DFQ_LE_CF=0 switches them off, DFQ_LE_CF_GROUP = G in
 * {2, 4, 8}, default 8 for a batched plan, 4 for a single network of >= 6 M paired elements, none for a smaller one: its
 * sweep is two launch latencies, and the lean launches only add to them).  A chain of relations whose every consumed range is closed-form -- a chain's first layer (rows * s,
 * dfq.py:62), depthwise layers in between, its last layer (columns * 1/s, dfq.py:73): max_i fl(w_i * s) == fl(max_i w_i * s)
 * -- has the scale factors of ALL its sweeps follow from a few scalars per channel (dfq.py:39-59 without reading a weight).
 * Its layers are not part of a sweep's launch: at the first sweep of every group of G sweeps ONE lean launch reads them, brings
 * them up to date, stores them and leaves sum |W - W_prev| (dfq.py:105-108) of the group's G sweeps; every enqueue call ends by
 * bringing them up to date like the deferred stores.  free_running_elements are NOT part of rw_elements: bytes per sweep as
 * executed = 8 * rw + 4 * ro - 4 * deferred * (D - 1) / D + 8 * free_running / G.  Bit-identical values and sweep counts. */
int64_t dfq_le_plan_free_running_elements(const dfq_le_plan* plan);
int32_t dfq_le_plan_free_running_group(const dfq_le_plan* plan);
/* 1: the lean launches run in the BACKGROUND (opt-in, DFQ_LE_CF_BG=1: bit-identical, measured no faster): a lean launch
 * looks ahead two groups -- it leaves sum |W - W_prev| of the NEXT group's sweeps, this group's were left by the launch before --
 * so its deadline is a whole group away and it runs on a second, low-priority stream of the plan next to the group's sweep
 * launches, in the launch boundaries of the caller's stream (which waits for it where the sums are read and at every
 * write-back; an enqueue call never returns with one in flight).  Same values, same sums, same sweep counts. */
int32_t dfq_le_plan_lean_background(const dfq_le_plan* plan);
int32_t dfq_le_plan_lean_tiles(const dfq_le_plan* plan);
/* Tuning aid: tile `tile` of the lean launch.  out3 = { kind (0 / 1: rows * s in 16-byte vectors / floats, 2: one thread per
 * row, 3 / 4: columns * 1/s, 5: one thread per row of a chain's last layer), rows, floats per row }. */
int dfq_le_plan_lean_info(const dfq_le_plan* plan, int32_t tile, int64_t* out3);
/* the same two counts for one launch level; returns the number of relations in it */
int32_t dfq_le_plan_level_launches(const dfq_le_plan* plan, int32_t level, int64_t* rw_elems,
                                   int64_t* ro_elems, int32_t* n_workgroups);
/* launch geometry of launch `level` of a sweep: grid_x = its workgroups (all of them work), grid_y = 1 */
int dfq_le_plan_level_grid(const dfq_le_plan* plan, int32_t level, int32_t* grid_x, int32_t* grid_y);

/* Single networks whose paired layers fit the chip's LDS run the WHOLE loop as one persistent, cooperative launch (every
 * workgroup keeps one 32 KB tile of one layer in its LDS for all sweeps; dfq_le_resident.hip): the number of its workgroups,
 * or 0 when the plan uses the streaming one-launch-per-sweep kernel (batched plans, networks too large for the
 * chip's resident workgroups, DFQ_LE_RESIDENT=0) -- then dfq_le_plan_resident_reason says why.  Results are
 * bit-identical either way. */
int32_t dfq_le_plan_resident_tiles(const dfq_le_plan* plan);
const char* dfq_le_plan_resident_reason(const dfq_le_plan* plan);
/* The persistent launch stores ALL OR NOTHING (round 5): its tiles write their result back only once every tile has finished
 * the loop.  If a workgroup abandons an in-launch wait (DFQ_SPIN_LIMIT), no tile stores, the caller's tensors are exactly as
 * they were passed, and dfq_le_run repeats the pass on one launch per level -- which waits for nothing inside a launch --
 * instead of returning DFQ_ERR_STATE; the plan then stays on that engine (dfq_le_plan_resident_tiles becomes 0,
 * dfq_le_plan_resident_reason says why).  Number of runs of this plan that were repeated that way (0 in normal operation).
 * The reference's loop (dfq.py:78-117) has no counterpart: it cannot fail half way. */
int32_t dfq_le_plan_degraded(const dfq_le_plan* plan);
/* Launches of the streaming engine in which workgroups wait for other workgroups (the one-launch sweep): 1 if a run of this
 * plan can end with DFQ_ERR_ABANDONED after having rewritten tensors (the persistent launch stores all or nothing and is not
 * counted).  dfq_le_plan_set_safe_mode: from now on the plan runs one launch per dependency level -- no workgroup waits for
 * another one, nothing can be abandoned (parity-tested since round 1, slower) -- and never the persistent launch.  The Python
 * binding's LEPlan.run() uses the pair to repeat an abandoned pass from a device-side snapshot of the tensors (dfq_amd/dfq.py). */
int32_t dfq_le_plan_has_waits(const dfq_le_plan* plan);
int dfq_le_plan_set_safe_mode(dfq_le_plan* plan);
/* The resident launch applies every sweep to its LDS tiles AT ONCE and learns only later (from a reducer workgroup, off every
 * dependency chain) whether dfq.py:105-115 let that sweep happen: a tile may be up to `spec` sweeps past the stopping point
 * and then restores the newest of its checkpoints and replays the logged per-channel factors (bit-identical: the same two
 * rounded multiplications per element and sweep).  Statistics of the LAST launch (tests, tuning; synchronises `stream`):
 * out5 = {tiles that rolled back, sweeps undone in total, most sweeps undone by one tile, speculation depth (DFQ_RES_SPEC,
 * default 2), sweeps between checkpoints (DFQ_RES_CKPT, default 8)}. */
int dfq_le_resident_stats(dfq_le_plan* plan, void* stream, int64_t* out5);
/* Streaming plans built with DFQ_LE_PERSIST=1 (an experiment, off by default): persistent workgroups of a sweep launch
 * (each walks its share of the sweep's tiles with the next tile's data in flight); 0 when a sweep launches one workgroup
 * per tile or the plan is resident.  DFQ_LE_SWEEP_WGS caps the number. */
int32_t dfq_le_plan_sweep_workgroups(const dfq_le_plan* plan);
/* Tuning aid for the persistent launch: restart, run `n_sweeps` sweeps with per-workgroup phase stamps (100 MHz wall
 * clock): out[(tile * 6 + sweep) * 12 + point] for the first 6 sweeps; points: 0 sweep start, 1 s_A solved, 2 row
 * statistics published, 3 s_B solved, 4 new values + statistics published, 5 ticket taken, 6 decision seen; point 7
 * of sweep 0 = layer << 32 | tile rows << 16 | tile columns.  `capacity` >= dfq_le_resident_trace_words().
 * Modifies the weights like an ordinary run.  Synchronises. */
int64_t dfq_le_resident_trace_words(const dfq_le_plan* plan);
int dfq_le_resident_trace(dfq_le_plan* plan, const dfq_le_config* cfg, int32_t n_sweeps, void* stream, int64_t* out,
                          int64_t capacity);

/* Enqueue exactly `n_sweeps` sweeps plus their convergence bookkeeping on `stream`; never
 * synchronises.  The device-side loop state decides whether a sweep still executes (after the
 * reference's exit condition fires the remaining launches are no-ops).  `restart` != 0 resets the
 * loop state (diff = 10, count = 0) first. */
int dfq_le_enqueue(dfq_le_plan* plan, const dfq_le_config* cfg, int32_t n_sweeps, int32_t restart,
                   void* stream);
/* Copy the loop state back (synchronises `stream`).  dfq_le_query: network 0, *done = every network of the
 * plan has stopped; dfq_le_query_all: `out` has dfq_le_plan_nets() entries. */
int dfq_le_query(dfq_le_plan* plan, void* stream, dfq_le_result* out, int32_t* done);
int dfq_le_query_all(dfq_le_plan* plan, void* stream, dfq_le_result* out, int32_t* all_done);
/* A stopping rule that spans several plans -- the sharded pass (dfq_amd/sharded.py), where dfq.py:105-108's sum runs over the
 * layers of ALL ranks.  dfq_le_set_diff_log (single-network plans; synchronises): from now on the plan also leaves diff_tmp of
 * sweep j (sweeps since the last restart) in log_device[j], j < capacity; capacity 0 switches it off.  The caller runs a chunk of
 * sweeps with the plan's own exit test disabled (converge_thres < 0), all-reduces the chunk's log entries in ONE collective and
 * hands the sums to dfq_le_shared_verdict: the (diff, count) state machine of dfq.py:110-115 over `n` values on the device,
 * ext4_device = { diff, count, sweeps, done } as float64 (initialise to { 10, 0, 0, 0 }: dfq.py:81-82); once `done`, the plan's
 * loop state (plan may be NULL for a rank that owns nothing) is stopped so that sweeps enqueued behind it are no-ops.
 * Asynchronous; nothing is read back. */
int dfq_le_set_diff_log(dfq_le_plan* plan, double* log_device, int32_t capacity);
int dfq_le_shared_verdict(dfq_le_plan* plan, const double* reduced_device, int32_t n, double* ext4_device, double converge_thres,
                          int32_t converge_count, int32_t max_sweeps, void* stream);
/* The whole dfq.py:83-115 loop: enqueue in chunks, poll, stop when the device says so.
 * Synchronises. */
int dfq_le_run(dfq_le_plan* plan, const dfq_le_config* cfg, void* stream, dfq_le_result* out);
/* Measurement aid (bench.py roofline): like dfq_le_enqueue(restart=1) for `n_sweeps` sweeps, but every
 * launch is bracketed by a pair of HIP events recorded on `stream`.  level_ms[l] receives the summed
 * duration of the launches of level l (array of dfq_le_plan_levels() doubles), *control_ms the summed
 * duration of the convergence kernel, *n_level_launches the number of level launches timed,
 * *empty_bracket_ms what a pair of event records measures with nothing between them (to subtract),
 * *lean_ms / *n_lean_launches (either may be NULL) the summed duration and number of the lean launches of the
 * free-running layers (one per group of sweeps).  Synchronises. */
int dfq_le_profile(dfq_le_plan* plan, const dfq_le_config* cfg, int32_t n_sweeps, void* stream,
                   double* level_ms, double* control_ms, int32_t* n_level_launches,
                   double* empty_bracket_ms, double* lean_ms, int32_t* n_lean_launches);

/* Tuning aid: run two sweeps and return the shader-clock stamps (s_memtime) that thread 0 of
 * workgroup `block` (= blockIdx.y * grid_x + blockIdx.x) of launch `launch` took at the phase boundaries of its tile during the second
 * sweep: [0] entry, [1] descriptor+state loaded, [2] data loads issued, [3] scales solved,
 * [4] barrier passed, [5] elements stored, [6] stats published, [7] partial written.  Synchronises;
 * modifies the weights like two ordinary sweeps. */
int dfq_le_trace(dfq_le_plan* plan, const dfq_le_config* cfg, int32_t launch, int32_t block, void* stream,
                 int64_t* stamps16);

/* Tuning aid: what workgroup `block` of launch `launch` does.  out8 = { kind (0/1/2: row tile of 16-byte vectors /
 * scalars / one thread per row; 3/4/5: the same for column tiles), rows, columns, elements read and written,
 * elements only read, 1 if it waits for another relation inside the launch, 1 if it publishes statistics,
 * index of its relation in the level-sorted table }. */
int dfq_le_plan_block_info(const dfq_le_plan* plan, int32_t launch, int32_t block, int64_t* out8);

/* Tuning aid: run three sweeps and return, for EVERY workgroup b of launch `launch` during the third one,
 * out[3b] = entry and out[3b+1] = exit time (100 MHz wall clock; 0 for a workgroup that had no tile) and
 * out[3b+2] = XCC_ID << 32 | HW_ID of the compute unit it ran on.  `capacity_blocks` >= grid_x * grid_y of
 * dfq_le_plan_level_grid.  Synchronises; modifies the weights like three ordinary sweeps. */
int dfq_le_trace_blocks(dfq_le_plan* plan, const dfq_le_config* cfg, int32_t launch, void* stream, int64_t* out,
                        int64_t capacity_blocks);

/* ------------------------------------------------------------------------------------------
 * Tensor primitives -- utils/quantize.py:23-76 (UniformQuantize.forward), :102-119 (QuantMeasure)
 * ---------------------------------------------------------------------------------------- */

/* out2[0] = min(x), out2[1] = max(x) as float32 (device).  `scratch2` is a device uint32[2] that
 * the call zeroes and uses for the order-preserving atomic reduction. */
int dfq_tensor_minmax(const float* x, int64_t n, float* out2, uint32_t* scratch2, void* stream);

/* Fake-quant round trip y = rint(clip((x + (-min)) / scale, qmin, qmax)) * scale + min
 * (quantize.py:70-74), five separately rounded float32 operations.
 *   range_mode 0: min/max are host doubles (the reference's `float(...)` call sites); the scale
 *                 recipe of quantize.py:49-66 runs in float64.
 *   range_mode 1: min/max are read from the device float32 pair `minmax_dev` and the recipe runs
 *                 in float64 on the device (same values as mode 0 without a host round trip).
 *   range_mode 2: min/max from `minmax_dev`, recipe entirely in float32 -- the `min_value=None`
 *                 tensor path of quantize.py:24-35 (bias fake-quant call sites :198,:226,:311,:335).
 * `codes` (device int32[n]) receives the integer codes when not NULL.  x == y (in place) is allowed. */
int dfq_fake_quant(const float* x, float* y, int64_t n, int32_t num_bits, int32_t symmetric,
                   int32_t range_mode, double min_value, double max_value,
                   const float* minmax_dev, int32_t* codes, void* stream);

/* QuantMeasure statistics (quantize.py:103-107): out2[0] = mean_n(min over chw of x[n]),
 * out2[1] = mean_n(max over chw of x[n]).  If `running2` (device float[2] = {running_min,
 * running_max}) is not NULL it is updated in place: running_min = min(running_min, out2[0]),
 * running_max = max(running_max, out2[1]).  `scratch` is a device uint32[2*n_samples]. */
int dfq_sample_minmax_mean(const float* x, int32_t n_samples, int64_t sample_len, float* out2,
                           float* running2, uint32_t* scratch, void* stream);
/* QuantMeasure.forward with update_stat (utils/quantize.py:102-119; improve_dfq.py:280-297 runs it for every activation of
 * every distilled batch) in TWO launches: the per-sample extrema, then one kernel in which every workgroup forms their mean
 * (as dfq_sample_minmax_mean does), folds it into running2 = (running_min, running_max) (quantize.py:106-107) and quantises
 * x -> y with the folded range (float64 recipe, asymmetric).  scratch: 4 * n_samples uint32 owned by the caller, ZERO before the
 * first call; parity alternates 0, 1, 0, ... from call to call on the same scratch (each call clears the half the next one
 * accumulates into).  Same numbers as dfq_sample_minmax_mean(running2) + dfq_fake_quant(range_mode 1). */
int dfq_quant_measure(const float* x, float* y, int32_t n_samples, int64_t sample_len, int32_t num_bits, float* running2,
                      uint32_t* scratch, int32_t parity, void* stream);
/* The same in ONE launch (round 4 experiment, OPT-IN through DFQ_QM_FUSED=1 in the Python layer: measured slower than the two
 * launches on the MI355X -- 189 vs 148 us on a [64, 96, 112, 112] activation, 3.73 vs 1.71 ms of QuantMeasure time per distilled
 * batch of config 5 -- the grid-wide meeting costs more than the launch boundary it replaces): a
 * grid of persistent workgroups -- dfq_quant_measure_fused_grid(n_samples, sample_len) of them, sized to be co-resident --
 * takes the per-sample extrema, waits (bounded) until the whole grid has arrived, then every workgroup forms the mean, folds
 * the running range and quantises its share of x, which up to the size of the chip's caches has not left them.  Bit-identical
 * to dfq_quant_measure.  scratch: 4 * n_samples + 4 uint32, 8-byte aligned, ZERO before the first call (the last four words
 * hold a monotonic arrival counter and an error word); parity alternates as above; `arrivals_before` = the sum of
 * dfq_quant_measure_fused_grid over all earlier calls on this scratch.  dfq_quant_measure_fused_status synchronises and
 * returns DFQ_ERR_STATE if a workgroup of an earlier call gave up waiting (DFQ_SPIN_LIMIT). */
int32_t dfq_quant_measure_fused_grid(int32_t n_samples, int64_t sample_len);
int dfq_quant_measure_fused(const float* x, float* y, int32_t n_samples, int64_t sample_len, int32_t num_bits, float* running2,
                            uint32_t* scratch, int32_t parity, int64_t arrivals_before, void* stream);
int dfq_quant_measure_fused_status(const uint32_t* scratch, int32_t n_samples, void* stream);

/* ------------------------------------------------------------------------------------------
 * Multi-layer weight quantisation -- utils/layer_transform.py:279-296 (quantize_targ_layer)
 * ---------------------------------------------------------------------------------------- */
typedef struct dfq_quant_plan dfq_quant_plan;

/* One plan over a table of tensors ("segments"): per-tensor min/max in one launch, fake-quant of
 * all tensors in a second launch. */
typedef struct dfq_segment {
    float* data;            /* device                                                    */
    int64_t n;
    int32_t num_bits;       /* per segment (weights 8, biases 16 in main_cls.py:181)      */
    int32_t symmetric;
    int32_t* codes;         /* device int32[n] or NULL                                    */
} dfq_segment;

int dfq_quant_plan_create(const dfq_segment* segs, int32_t n_segs, dfq_quant_plan** out_plan);
void dfq_quant_plan_destroy(dfq_quant_plan* plan);
/* min/max of every segment -> quantise every segment in place (range_mode 1 recipe). */
int dfq_quant_plan_run(dfq_quant_plan* plan, void* stream);
/* min/max of every segment only (no quantisation): what a calibration-table writer needs
 * (convert_ncnn.py:183-190) */
int dfq_quant_plan_measure(dfq_quant_plan* plan, void* stream);
/* device float32[2*n_segs] min/max pairs of the last run / measure (valid after the stream reaches it) */
const float* dfq_quant_plan_minmax(const dfq_quant_plan* plan);

/* ------------------------------------------------------------------------------------------
 * Lazy-scale equalisation (opt-in extension): the sweeps of dfq.py:83-101 with a GIVEN sweep count,
 * computed from the pristine weights and the cumulative scale vectors of utils/relation.py:20-24 -- a sweep only READS
 * (4 B per element of the layers it still has to look at, see the byte accounting below), the weights / biases / BN proxies / scale_cum vectors of the relations are written ONCE at the end
 * (W = diag(S_out) W0 diag(1/S_in); 8 B per weight).  Result: within 1e-5 (relative) of the sequentially rescaled tensors of
 * the reference loop run for the same number of sweeps -- NOT bit-identical to them (the default engines are).  The
 * data-dependent exit test of dfq.py:105-115 is not available in this formulation: the caller passes the count (e.g. the one
 * dfq_le_run reports for the same network).  Same layer / relation tables as dfq_le_plan_create_batch.  Asynchronous run.
 * ---------------------------------------------------------------------------------------- */
typedef struct dfq_le_lazy_plan dfq_le_lazy_plan;   /* opaque */
int dfq_le_lazy_plan_create(const dfq_layer* layers, int32_t n_layers, const int32_t* layer_net, int32_t n_nets,
                            const dfq_relation* relations, int32_t n_relations, dfq_le_lazy_plan** out_plan);
void dfq_le_lazy_plan_destroy(dfq_le_lazy_plan* plan);
/* cfg: s_range / eps / signed_range as for dfq_le_run (the convergence fields are ignored); sweeps_per_net[n] >= 0 */
int dfq_le_lazy_run(dfq_le_lazy_plan* plan, const dfq_le_config* cfg, const int32_t* sweeps_per_net, void* stream);
/* byte accounting: passes whose element factors never change run once (chain starts / ends, depthwise layers: their extrema
 * are kept and rescaled per sweep by the solve launch), so the FIRST sweep reads 4 B x paired_elements, every later sweep
 * 4 B x sweep_elements (the non-depthwise layers in the interior of a chain, once per role), and the final materialisation
 * moves 8 B x weight_elements */
int32_t dfq_le_lazy_plan_levels(const dfq_le_lazy_plan* plan);
int64_t dfq_le_lazy_plan_paired_elements(const dfq_le_lazy_plan* plan);
int64_t dfq_le_lazy_plan_sweep_elements(const dfq_le_lazy_plan* plan);
int64_t dfq_le_lazy_plan_weight_elements(const dfq_le_lazy_plan* plan);

/* ------------------------------------------------------------------------------------------
 * Bias correction -- dfq.py:173-293 (bias_correction), :8-25 (_quantize_error)
 * ---------------------------------------------------------------------------------------- */
typedef struct dfq_bc_plan dfq_bc_plan;

/* One BatchNorm contributing to E[x] of a layer (dfq.py:229-270), in the order the reference
 * merges them (depth-sorted, stable). */
typedef struct dfq_bc_source {
    const float* fake_weight;  /* device [channels]  (gamma~)                               */
    const float* fake_bias;    /* device [channels]  (beta~) -- read at the layer's turn    */
    int32_t channels;
    int32_t relu;              /* 1: E = gamma*pdf(-b/g) + b*(1-cdf(-b/g)), clipped at 0    */
    int32_t concat;            /* 1: torch.cat onto the running expectation, 0: add (first
                                  source: ignored)                                          */
} dfq_bc_source;

typedef struct dfq_bc_step {
    int32_t layer;             /* index into the dfq_layer table; bias must be non-NULL     */
    int32_t source_begin;      /* range in the dfq_bc_source table                          */
    int32_t source_count;
    float* next_bn_bias;       /* device [out_ch]: fake_bias of the next BN in graph order,
                                  receives += (-bias) (dfq.py:204-206,293); NULL if none    */
    int32_t net;               /* network of a batched plan (0 for a single network); steps are
                                  listed network by network, each network in graph order: the
                                  j-th steps of all networks share one launch                 */
    int32_t reserved;
} dfq_bc_step;

int dfq_bc_plan_create(const dfq_layer* layers, int32_t n_layers,
                       const dfq_bc_step* steps, int32_t n_steps,
                       const dfq_bc_source* sources, int32_t n_sources,
                       dfq_bc_plan** out_plan);
/* Replicated form (see dfq_le_plan_create_replicated): the three tables describe the FIRST of `n_nets` networks of one
 * architecture (step.net = 0, indices relative to that network), network n's tensors lie bases[n] - bases[0] bytes further.
 * (Replaces, for a batch, the per-network graph walk of dfq.py:194-270.) */
int dfq_bc_plan_create_replicated(const dfq_layer* layers, int32_t n_layers, const dfq_bc_step* steps, int32_t n_steps,
                                  const dfq_bc_source* sources, int32_t n_sources, const void* const* bases, int32_t n_nets,
                                  dfq_bc_plan** out_plan);
void dfq_bc_plan_destroy(dfq_bc_plan* plan);
/* per-tensor min/max of all step layers (one launch) -> the sequential per-layer chain (one launch); a step forms
 * the quant-error row sums eps[o, i] = sum_k (Q(w) - w) of its rows (8 bit, dfq.py:216-219) in registers, straight from
 * the weights: 8 B per weight for the whole pass.  Asynchronous. */
int dfq_bc_plan_run(dfq_bc_plan* plan, int32_t symmetric, void* stream);
/* device pointers into the plan's scratch (tests): the correction vector bias[O] of a step; eps[O*I/g] only for plans
 * created with DFQ_BC_EPS=1 in the environment (debug: one more launch materialises the row sums the chain computes on the
 * fly, same arithmetic) -- NULL otherwise */
/* Synchronises `stream`; DFQ_ERR_STATE if a workgroup of the last run gave up waiting for the correction step it
 * depends on (the chain of all layers is ONE launch whose workgroups wait for the previous layer's; the wait is
 * bounded).  DFQ_BC_MERGED=0 in the environment at plan creation restores one launch per layer. */
int dfq_bc_plan_status(dfq_bc_plan* plan, void* stream);
const float* dfq_bc_plan_eps(const dfq_bc_plan* plan, int32_t step);
const float* dfq_bc_plan_correction(const dfq_bc_plan* plan, int32_t step);
int64_t dfq_bc_plan_weight_elements(const dfq_bc_plan* plan);
/* 1 when the one-launch chain hands values from step to step as tagged 64-bit slots (the default), 0 when it uses per-step
 * counters (DFQ_BC_TAGGED=0, or a graph the tagged scheme does not cover) or one launch per chain position. */
int32_t dfq_bc_plan_tagged(const dfq_bc_plan* plan);
/* 1 if the correction chain runs as ONE launch whose workgroups wait for each other (dfq_bc_plan_status can then report
 * DFQ_ERR_ABANDONED after biases / BN proxies have been rewritten).  dfq_bc_plan_set_safe_mode: one launch per chain position
 * from now on -- nothing waits, nothing can be abandoned (the parity-tested DFQ_BC_MERGED=0 path).  BCPlan.run() of the Python
 * binding uses the pair to repeat an abandoned pass from a device-side snapshot. */
int32_t dfq_bc_plan_has_waits(const dfq_bc_plan* plan);
int dfq_bc_plan_set_safe_mode(dfq_bc_plan* plan);
/* 1 when the LATEST run of the plan used the tagged slots, 0 when it used counters (a run recorded into a graph) or has not run.
 * A run on the NULL stream is an ordinary run (until round 5 it was mistaken for a recording: counters, no guard). */
int32_t dfq_bc_plan_last_run_tagged(const dfq_bc_plan* plan);
/* 1 when a tagged run of the plan is ONE launch: the per-tensor min/max blocks are workgroups of the chain launch, woven in a few chain
 * positions in front of the steps that need them (DFQ_BC_ONE_LAUNCH=0, a kept eps matrix or rows too long for the registers:
 * min/max launch + chain launch).  A never-rewritten BN read through a ReLU does NOT prevent it: the cached moments of such BNs are
 * refreshed by the launch's first blocks and their readers wait for those blocks (cache_arrive). */
int32_t dfq_bc_plan_one_launch(const dfq_bc_plan* plan);
/* Diagnostics.  A library built with -DDFQ_BC_TRACE=1 (tools/bc_trace.py) records five timestamps per workgroup of the one-launch
 * chain; this copies up to `words` 64-bit words of them to `out` and returns the number copied -- 0 from the shipped library. */
int64_t dfq_bc_debug_trace(long long* out, int64_t words);
int64_t dfq_bc_plan_eps_elements(const dfq_bc_plan* plan);
/* Depthwise steps folded into the per-row tail of the step in front of them, and the dependent positions the chain is left
 * with (MobileNetV2: 17 of 52 steps folded -> 35 positions).  A layer with one input channel per group and as many groups as
 * outputs corrects channel o with eps[o] * E[o] alone (dfq.py:281-287 with I/g = 1), and E[o] is what the thread owning row o
 * of the previous step has just produced (dfq.py:204-206, 238-242): that thread performs the depthwise layer's update too --
 * the same operations in the same order, bit-identical -- and one hand-over through the memory system per depthwise layer
 * disappears.  DFQ_BC_FOLD=0 in the environment at plan creation keeps every step. */
int32_t dfq_bc_plan_folded(const dfq_bc_plan* plan);
int32_t dfq_bc_plan_chain_steps(const dfq_bc_plan* plan);

/* _quantize_error (dfq.py:8-25) on one tensor: q(x) - x with per-tensor min/max, 5 reductions:
 *   reduction 0: none   -> out[n]
 *   reduction 1: 'sum'  -> out[0] = sum |eps|
 *   reduction 2: 'mean' -> out[0] = mean eps
 *   reduction 3: 'channel' -> out[0] = sum_o | sum_rest eps |      (rows = shape[0])
 *   reduction 4: 'spatial' -> out[0] = sum_{o,i} | sum_khkw eps |  (rows = shape[0]*shape[1])
 * `rows` is the number of leading-dimension groups for reductions 3/4 (ignored otherwise).
 * `scratch` is device memory of dfq_quant_error_scratch_bytes(n, rows) bytes. */
size_t dfq_quant_error_scratch_bytes(int64_t n, int64_t rows);
int dfq_quant_error(const float* x, int64_t n, int64_t rows, int32_t num_bits, int32_t symmetric,
                    int32_t reduction, float* out, void* scratch, void* stream);

/* ------------------------------------------------------------------------------------------
 * Row / column rescale helpers -- utils/layer_transform.py:231-276 (merge_batchnorm),
 * utils/quantize.py:145-174,:269-289 (merge_scale*), dfq.py:121-170 (bias_absorption, clip_weight)
 * ---------------------------------------------------------------------------------------- */

/* w[o, :] = w[o, :] * s[o]   (op 0)   or   w[o, :] / s[o]   (op 1);  row_len = I/g * khkw */
int dfq_scale_rows(float* w, int32_t rows, int64_t row_len, const float* s, int32_t op, void* stream);
/* Input-channel rescale of a grouped conv weight [O, I/g, khkw]: input channel of element
 * (o, i, k) is  (o / (O/groups)) * I/g + i;  w = w * s[ch] (op 0) or w / s[ch] (op 1). */
int dfq_scale_cols(float* w, int32_t out_ch, int32_t in_per_group, int32_t khkw, int32_t groups,
                   const float* s, int32_t op, void* stream);
/* vector helpers on [n]: y = y * s (op 0), y / s (op 1), y + s (op 2), y - s (op 3) */
int dfq_vec_op(float* y, const float* s, int64_t n, int32_t op, void* stream);

/* Batched rebuild from pristine tensors and cumulative scale vectors (utils/relation.py:20-24; the replicated write of
 * the sharded equalisation):  dst[o,i,k] = fl(fl(src[o,i,k] * s_out[o]) / s_in[ch(o,i)])  with
 * ch = (o / (rows/groups)) * cols + i  -- two separately rounded float32 operations in this order; a NULL vector
 * skips its step (both NULL: a copy).  Vectors (bias, BN proxies) are items with cols = khkw = groups = 1.  src may equal
 * dst.  ONE launch covers all items of a plan (8 B per element); identical inputs give bit-identical outputs on every
 * rank.  `items` is a host array and is copied.  Synchronises (create only). */
typedef struct dfq_rebuild_item {
    const float* src;      /* device [rows, cols, khkw]                                   */
    float* dst;            /* device, same shape                                          */
    const float* s_out;    /* device [rows] or NULL                                       */
    const float* s_in;     /* device [groups * cols] or NULL                              */
    int32_t rows;          /* O                                                           */
    int32_t cols;          /* I / groups                                                  */
    int32_t khkw;
    int32_t groups;
    int32_t in_reciprocal; /* 0: divide by s_in[ch]; 1: s_in holds reciprocals: multiply  */
    int32_t reserved;
} dfq_rebuild_item;
typedef struct dfq_rebuild_plan dfq_rebuild_plan;   /* opaque */
int dfq_rebuild_plan_create(const dfq_rebuild_item* items, int32_t n_items, dfq_rebuild_plan** out_plan);
void dfq_rebuild_plan_destroy(dfq_rebuild_plan* plan);
int64_t dfq_rebuild_plan_elements(const dfq_rebuild_plan* plan);
int dfq_rebuild_plan_run(dfq_rebuild_plan* plan, void* stream);
/* x = clamp(x, lo, hi) (dfq.py:170) */
int dfq_clamp(float* x, int64_t n, float lo, float hi, void* stream);

/* BatchNorm folding (layer_transform.py:246-272) for one (conv, bn) pair.  In place:
 *   k = gamma / sqrt(var + bn_eps);  W[o,:] *= k[o];  b = b*k + (beta - gamma*mean/sqrt(var+bn_eps));
 *   fake_weight = |gamma|; fake_bias = beta;  then gamma=1, var=1, beta=0, mean=0. */
int dfq_fold_batchnorm(float* w, float* b, int32_t out_ch, int64_t row_len, float* gamma,
                       float* beta, float* mean, float* var, float bn_eps, float* fake_weight,
                       float* fake_bias, void* stream);

/* Bias absorption for one relation (dfq.py:138-164):  c = max(0, beta~ - N*gamma~);
 * wc[g] = (sum_khkw W2)[g] . c[g];  b1 -= c;  beta~ -= c;  b2 += wc. */
int dfq_bias_absorb(const float* w2, int32_t o2, int32_t in_per_group, int32_t khkw, int32_t o1,
                    float* b1, float* b2, const float* bn_weight, float* bn_bias, float n_sigma,
                    void* stream);

/* ------------------------------------------------------------------------------------------
 * Stand-alone steps of one equalisation pair / one correction layer: what the
 * plans above fuse, exported one by one.  Composing them reproduces dfq.py:28-75 for one pair bit for bit
 * with the plans (same device arithmetic); they cost two passes over the data and four launches per
 * pair, so the plans are the fast path.  All pointers are device pointers, all calls asynchronous.
 * ---------------------------------------------------------------------------------------- */
/* out[r] = range(W[r, :]): max-min (signed_range 0, dfq.py:54-55) or max|.| (1, dfq.py:50-51) */
int dfq_row_range(const float* w, int64_t rows, int64_t row_len, int32_t signed_range, float* out, void* stream);
/* out[g*I/g + ii] = range over the O/groups rows of pairing group g and the khkw taps of input channel ii
 * of W2 [O, I/g, khkw] (the view of dfq.py:41-46; `groups` = O1 / (I2/g), 1 for an ordinary pair) */
int dfq_col_range(const float* w2, int32_t out_ch, int32_t in_per_group, int32_t khkw, int32_t groups,
                  int32_t signed_range, float* out, void* stream);
/* S[c] = clamp((1/(r1+eps)) * sqrt(r1*r2+eps)) with the Python max/min semantics of dfq.py:58-59 (NaN ->
 * s_hi); Sinv[c] = the float32 value dfq.py:73 multiplies by (1/S, or float32(1/s_range[k]) where the clamp
 * replaced S by a Python float).  Sinv may be NULL. */
int dfq_le_solve(const float* r1, const float* r2, int64_t n, float eps, double s_lo, double s_hi, float* S,
                 float* Sinv, void* stream);
/* dfq.py:62-73: W1[c,:] *= S[c]; b1, bn_weight, bn_bias (each may be NULL) *= S; W2[:, c] *= Sinv[c] */
int dfq_le_apply(float* w1, int32_t o1, int64_t row_len1, float* w2, int32_t o2, int32_t in_per_group2,
                 int32_t khkw2, float* b1, float* bn_weight, float* bn_bias, const float* S, const float* Sinv,
                 void* stream);
/* the four steps above for one pair = _layer_equalization (dfq.py:28-75).  `S` [O1] receives the scale
 * vector, `workspace` is 3*O1 floats of device scratch. */
int dfq_le_pair(float* w1, int32_t o1, int64_t row_len1, float* w2, int32_t o2, int32_t in_per_group2,
                int32_t khkw2, float* b1, float* bn_weight, float* bn_bias, double s_lo, double s_hi,
                int32_t signed_range, float eps, float* S, float* workspace, void* stream);
/* out[0] = float(torch.mean(torch.abs(W - W_prev))) (dfq.py:108): float64 sum in a fixed order, divided by
 * n, rounded once to float32.  `scratch` is dfq_absdiff_mean_scratch_bytes(n) bytes of device memory. */
size_t dfq_absdiff_mean_scratch_bytes(int64_t n);
int dfq_absdiff_mean(const float* w, const float* prev, int64_t n, float* out, void* scratch, void* stream);
/* Per-row (per-output-channel) form of dfq_fake_quant: row r is quantised with its own range -- (mins[r],
 * maxs[r]) if given, its own min/max otherwise (written to minmax_out[2r], [2r+1] if non-NULL).  Same
 * recipe per row as utils/quantize.py:49-74; this is the per-channel weight quantiser of the ncnn table
 * (convert_ncnn.py:178-201: one scale per output channel) and of ZeroQ (quant_utils.py:39-135). */
int dfq_fake_quant_rows(const float* x, float* y, int64_t rows, int64_t row_len, const float* mins,
                        const float* maxs, int32_t num_bits, int32_t symmetric, float* codes, float* minmax_out,
                        void* stream);
/* ZeroQ's per-output-channel asymmetric weight quantiser (ZeroQ/utils/quantization_utils/quant_utils.py:85-135 as
 * used by quant_modules.py:161-171): scale = (1/clamp(max-min, 1e-8)) * (2^k-1) (two roundings, as torch evaluates `n / tensor`), zero point = round(scale*min) + 2^(k-1),
 * q = clamp(round(scale*x - zp), -2^(k-1), 2^(k-1)-1), y = (q + zp)/scale, all in float32.  Row r uses (mins[r],
 * maxs[r]) if given, its own min/max otherwise; `codes` (float, signed integers) and `minmax_out` may be NULL. */
int dfq_zeroq_quant_rows(const float* x, float* y, int64_t rows, int64_t row_len, const float* mins, const float* maxs,
                         int32_t num_bits, float* codes, float* minmax_out, void* stream);
/* out[o] = eps[o, :] . expect[(o / (O/groups)) * I/g ...] (dfq.py:281-287), float64 accumulation */
int dfq_grouped_matvec(const float* eps, const float* expect, int32_t out_ch, int32_t in_per_group,
                       int32_t groups, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Analytic activation ranges -- utils/layer_transform.py:347-609 (set_quant_minmax): the channel arithmetic
 * and reductions behind QuantMeasure.running_min / running_max when no calibration data is used.  The graph
 * walk (find_prev_bn :299-344, branch grouping) stays on the host; see dfq_amd/utils/layer_transform.py.
 * ---------------------------------------------------------------------------------------- */
typedef struct dfq_bn_range_req {
    const float* fake_weight;  /* device [channels]  gamma~ of the BN that feeds the quantiser      */
    const float* fake_bias;    /* device [channels]  beta~                                           */
    int32_t channels;
    int32_t relu_mode;         /* 0: none, 1: ReLU attached, 2: ReLU6 attached                       */
} dfq_bn_range_req;
/* Every "one BN -> one quantiser" case of a network in one launch: out[2i] = min_c(beta~ - N*gamma~)
 * (clamped at 0 if a ReLU/ReLU6 follows), out[2i+1] = max_c(beta~ + N*gamma~) (clamped at 6 for ReLU6)
 * (:403-404, :468-469).  `out` is device [n_reqs][2]; `scratch` is dfq_bn_ranges_scratch_bytes(n_reqs) bytes
 * of device memory.  Copies the request table to the device (synchronises the stream once). */
size_t dfq_bn_ranges_scratch_bytes(int32_t n_reqs);
int dfq_bn_ranges(const dfq_bn_range_req* reqs, int32_t n_reqs, float n_sigma, float* out, void* scratch, void* stream);
/* mean / variance of N(beta~, gamma~^2) after nothing (mode 0: beta~, gamma~^2), ReLU (1, :407-410) or ReLU6
 * (2, :411-418), per channel; accumulate != 0 adds into mean/var (residual adds, :521-531). */
int dfq_relu_moments(const float* weight, const float* bias, int64_t n, int32_t relu_mode, float* mean, float* var,
                     int32_t accumulate, void* stream);
/* an add node followed by ReLU (1) / ReLU6 (2): (mean, var) <- moments(sqrt(var + eps), mean) (:533-540) */
int dfq_moments_after_add(float* mean, float* var, int64_t n, int32_t relu_mode, float eps, void* stream);
/* out2 = (min_c(mean - N*sqrt(var+eps)), max_c(mean + N*sqrt(var+eps))) (:571-573) */
int dfq_moment_range(const float* mean, const float* var, int64_t n, float eps, float n_sigma, float* out2, void* stream);
/* case (d), :455-463: a BN proxy vector pushed through a conv / linear layer without batch norm:
 * v_out[o] = sum_i (sum_k W[o,i,k]) * v_in[group(o)*I/g + i] + bias[o] (bias may be NULL) */
int dfq_bn_through_layer(const float* weight, int32_t out_ch, int32_t in_per_group, int32_t khkw, int32_t groups,
                         const float* bias, const float* v_in, float* v_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm-statistics loss of ZeroQ's data distillation -- ZeroQ/distill_data.py:170-196 (own_loss :40-45):
 * for a BN input x [N, C, H, W] (rows = N*C rows of hw = H*W floats, channel of row r = r % C),
 *   loss2[0] = sum_{n,c} (bn_mean[c] - mean_hw x)^2 / denom,   loss2[1] = sum_{n,c} (bn_std[c] - std_hw(x + eps))^2 / denom
 * (unbiased std; denom = A.size(0) of own_loss: C for the BN terms, N for the input-batch term of :192-196), one read of x; row_mean / row_std [rows] are kept for the backward pass, which writes (or adds
 * into) grad_x = grad_mean_loss * d loss2[0]/dx + grad_std_loss * d loss2[1]/dx with one read of x.
 * `scratch`: dfq_bn_stat_loss_scratch_bytes(rows) bytes.  H*W == 1 follows distill_data.py:181-182: the mean term per
 * (n, c) value, the std term over the contiguous [N, C] block REINTERPRETED as C rows of N values (row r against
 * bn_std[r]); row_std then holds C entries.  The `_dev` backward reads the two upstream gradients from device memory
 * (grad_pair[0] = d/d loss2[0], grad_pair[1] = d/d loss2[1]): no host synchronisation inside an autograd backward.
 * ---------------------------------------------------------------------------------------- */
size_t dfq_bn_stat_loss_scratch_bytes(int64_t rows);
int dfq_bn_stat_loss_forward(const float* x, int64_t rows, int64_t hw, int32_t channels, const float* bn_mean,
                             const float* bn_std, float eps, float denom, float* row_mean, float* row_std, float* loss2,
                             void* scratch, void* stream);
int dfq_bn_stat_loss_backward(const float* x, int64_t rows, int64_t hw, int32_t channels, const float* bn_mean,
                              const float* bn_std, float eps, float denom, const float* row_mean, const float* row_std,
                              float grad_mean_loss, float grad_std_loss, float* grad_x, int32_t accumulate, void* stream);
int dfq_bn_stat_loss_backward_dev(const float* x, int64_t rows, int64_t hw, int32_t channels, const float* bn_mean,
                                  const float* bn_std, float eps, float denom, const float* row_mean, const float* row_std,
                                  const float* grad_pair, float* grad_x, int32_t accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DFQ_HIP_H */
