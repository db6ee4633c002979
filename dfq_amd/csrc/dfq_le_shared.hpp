// Pieces shared by the two equalisation engines: the streaming one-launch-per-sweep kernel (dfq_le.hip) and the
// register-resident whole-loop kernel (dfq_le_resident.hip).
#pragma once

#include <cstdlib>

#include "dfq_common.hpp"

namespace dfq {

struct LeParams {
    float s_lo, s_hi, inv_lo, inv_hi, eps;
    int32_t hi_gt_lo, signed_range;
    int32_t poll_naps;      // s_sleep(8) units between two polls of a dependency counter
    int32_t spin_limit;     // polls after which a workgroup abandons an in-launch wait (DFQ_SPIN_LIMIT; tests force an abandon with 1)
    int32_t defer;          // streaming engine: one-way-scaled layers are stored every `defer`-th sweep (1, 2 or 4; dfq_le.hip)
};

struct LeState {
    double diff;
    double last_diff_tmp;
    int32_t count;
    int32_t sweeps;
    int32_t done;
    int32_t log_cap;        // entries of `log` (0: none)
    int32_t happen;         // streaming engine: the latest sweep KNOWN to happen (its predecessor's verdict was "go on"; 0 after a restart,
                            // -1 when the loop is over before it began).  Only grows during a run: what a background lean launch asks
                            // instead of `done`, which a later convergence launch may raise while it is still running (dfq_le_cf.hpp)
    int32_t pad_;
    // optional: diff_tmp of sweep j (sweeps since the last restart) is also left in log[j] -- a sharded pass (dfq_amd/sharded.py)
    // runs chunks of sweeps without the reference's exit test and all-reduces a chunk's values in ONE collective
    // (dfq_le_set_diff_log; untouched by restarts)
    double* log;
};

// Every in-launch wait is bounded: DFQ_SPIN_LIMIT polls (default: seconds), then the workgroup gives up, raises the plan's
// error word (every other waiter follows within 256 polls) and the next query / status call returns DFQ_ERR_STATE.
inline int32_t spin_limit_from_env(int32_t dflt) {
    const char* e = getenv("DFQ_SPIN_LIMIT");
    return (e && atoi(e) > 0) ? atoi(e) : dflt;
}

// a / b for 0 <= a < 2^20, b >= 1 in four instructions: (a + 0.5) / b is at least 0.5/b away from every
// integer, while v_rcp_f32 (1 ulp) plus the multiply are off by < 2e-7 * a/b < 0.5/b, so truncating
// is exact.  (A 32-bit integer division expands to ~40 dependent instructions and a tile needs a
// dozen of them: they were a visible share of its latency and of the kernel's code size.)
__device__ __forceinline__ int small_div(int a, int b) {
    return (int)(((float)a + 0.5f) * __builtin_amdgcn_rcpf((float)b));
}

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

__device__ __forceinline__ float range_of(float mn, float mx, int signed_range) {
    // max(|mn|, |mx|) == max(mx, -mn) whenever mn <= mx.  (Written without fabsf(): the abs source
    // modifier folded into the following select trips an instruction-selection bug of this compiler.)
    if (signed_range) return fmaxf(mx, -mn);
    return mx - mn;
}

}  // namespace dfq

static inline dfq::LeParams make_params(const dfq_le_config* c) {
    dfq::LeParams q;
    q.s_lo = c->s_lo; q.s_hi = c->s_hi; q.inv_lo = c->inv_lo; q.inv_hi = c->inv_hi; q.eps = c->eps;
    q.hi_gt_lo = c->hi_gt_lo; q.signed_range = c->signed_range;
    const char* pe = getenv("DFQ_LE_POLL_NAPS");
    q.poll_naps = (pe && atoi(pe) > 0) ? atoi(pe) : 2;
    q.spin_limit = dfq::spin_limit_from_env(4000000);   // x (sleep + load): several seconds
    q.defer = 1;                                        // set by the streaming plan (le_defer_depth)
    return q;
}

