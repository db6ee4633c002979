// Stand-alone primitives of the equalisation / bias-correction path: the single
// steps that the plans of dfq_le.hip / dfq_bc.hip fuse, exported one by one so that a caller can drive
// dfq.py:28-75 (_layer_equalization) pair by pair, or rebuild the loop differently.  They use the same
// device functions (le_solve semantics, order-preserving min/max, double-scale quantiser) as the fused
// kernels, so composing them reproduces the plans' results bit for bit; they are NOT the fast path
// (four launches and two passes over the data per pair instead of one pass per level).
#include <algorithm>

#include "dfq_common.hpp"

namespace dfq {

__device__ __forceinline__ float prim_range(float mn, float mx, int signed_range) {
    // max |.| == max(mx, -mn) for mn <= mx (dfq.py:50-51); `+ 0.0f` turns the -0 of an all-zero row into the
    // +0 that abs() returns (the fused kernels skip this: the sign of a zero range cannot reach S)
    if (signed_range) return fmaxf(mx, -mn) + 0.0f;
    return mx - mn;                                     // dfq.py:54-55
}

// one wave per row: out[r] = range(W[r, :])   (dfq.py:50-55, first layer)
__global__ __launch_bounds__(kBlock) void row_range_kernel(const float* __restrict__ w, int64_t rows, int64_t row_len,
                                                           int signed_range, float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (r >= rows) return;
    const int lane = threadIdx.x % kWave;
    const float* row = w + r * row_len;
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t i = lane; i < row_len; i += kWave) {
        const float v = row[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (lane == 0) out[r] = prim_range(mn, mx, signed_range);
}

// one thread per paired channel c = g * I2g + ii: range over the go rows of group g and the khkw taps
// (dfq.py:41-46 view of the second layer).  Neighbouring threads read neighbouring input channels.
__global__ __launch_bounds__(kBlock) void col_range_kernel(const float* __restrict__ w2, int32_t out_ch,
                                                           int32_t in_per_group, int32_t khkw, int32_t groups,
                                                           int signed_range, float* __restrict__ out) {
    const int64_t c = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (c >= (int64_t)groups * in_per_group) return;
    const int g = (int)(c / in_per_group);
    const int ii = (int)(c - (int64_t)g * in_per_group);
    const int go = out_ch / groups;
    float mn = INFINITY, mx = -INFINITY;
    for (int j = 0; j < go; ++j) {
        const float* p = w2 + (((int64_t)g * go + j) * in_per_group + ii) * khkw;
        for (int k = 0; k < khkw; ++k) {
            const float v = p[k];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    }
    out[c] = prim_range(mn, mx, signed_range);
}

// dfq.py:58-59 with Python's max/min semantics on a 0-dim float32 tensor (same as dfq_le.hip le_solve)
__global__ __launch_bounds__(kBlock) void le_solve_kernel(const float* __restrict__ r1, const float* __restrict__ r2,
                                                          int64_t n, float eps, float s_lo, float s_hi, float inv_lo,
                                                          float inv_hi, int hi_gt_lo, float* __restrict__ S,
                                                          float* __restrict__ Sinv) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float a = r1[i] + eps;
    const float recip = 1.0f / a;
    const float prod = r1[i] * r2[i];
    const float rad = prod + eps;
    const float root = sqrtf(rad);
    const float s = recip * root;
    const bool keep_hi = s < s_hi;                 // False for NaN -> hi
    const float t = keep_hi ? s : s_hi;
    const bool keep_lo = keep_hi ? (t > s_lo) : (hi_gt_lo != 0);
    const float so = keep_lo ? t : s_lo;
    S[i] = so;
    if (Sinv) Sinv[i] = keep_lo ? (keep_hi ? (1.0f / so) : inv_hi) : inv_lo;
}

// W2[(g*go + j), ii, k] *= f[g * I2g + ii]   (dfq.py:73 with the precomputed 1/s)
__global__ __launch_bounds__(kBlock) void col_mul_kernel(float* __restrict__ w2, int64_t n, int32_t out_ch,
                                                         int32_t in_per_group, int32_t khkw, int32_t groups,
                                                         const float* __restrict__ f) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n) return;
    const int64_t row = e / ((int64_t)in_per_group * khkw);
    const int64_t rem = e - row * (int64_t)in_per_group * khkw;
    const int ii = (int)(rem / khkw);
    const int g = (int)(row / (out_ch / groups));
    w2[e] = w2[e] * f[(int64_t)g * in_per_group + ii];
}

__global__ __launch_bounds__(kBlock) void row_mul_kernel(float* __restrict__ w, int64_t n, int64_t row_len,
                                                         const float* __restrict__ s) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= n) return;
    w[e] = w[e] * s[e / row_len];
}

__global__ __launch_bounds__(kBlock) void vec_mul3_kernel(float* __restrict__ a, float* __restrict__ b,
                                                          float* __restrict__ c, int64_t n, const float* __restrict__ s) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float v = s[i];
    if (a) a[i] = a[i] * v;
    if (b) b[i] = b[i] * v;
    if (c) c[i] = c[i] * v;
}

// sum |w - prev| in float64, fixed order: per-wave partials, then one workgroup adds them in index order
constexpr int kDiffChunk = kBlock * 16;
__global__ __launch_bounds__(kBlock) void absdiff_partial_kernel(const float* __restrict__ w, const float* __restrict__ prev,
                                                                 int64_t n, double* __restrict__ partials) {
    const int64_t b = (int64_t)blockIdx.x * kDiffChunk;
    const int64_t e = std::min<int64_t>(b + kDiffChunk, n);
    double acc = 0.0;
    for (int64_t i = b + threadIdx.x; i < e; i += kBlock) {
        const float d = w[i] - prev[i];
        acc += (double)__uint_as_float(__float_as_uint(d) & 0x7fffffffu);
    }
    acc = wave_sum(acc);
    if (threadIdx.x % kWave == 0) partials[(int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave] = acc;
}

__global__ __launch_bounds__(kBlock) void absdiff_final_kernel(const double* __restrict__ partials, int64_t n_partials,
                                                               double n, float* __restrict__ out) {
    __shared__ double sh[kBlock / kWave];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n_partials; i += kBlock) acc += partials[i];
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) out[0] = (float)(acc / n);     // float(torch.mean(torch.abs(W - W_prev))) (dfq.py:108)
}

// per-row (= per output channel) fake-quant: row r uses its own (min, max); ranges either given or taken
// from the row itself.  One wave per row.
__global__ __launch_bounds__(kBlock) void fake_quant_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                 int64_t rows, int64_t row_len,
                                                                 const float* __restrict__ mins,
                                                                 const float* __restrict__ maxs, int num_bits,
                                                                 int symmetric, float* __restrict__ codes,
                                                                 float* __restrict__ minmax_out) {
    const int64_t r = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (r >= rows) return;
    const int lane = threadIdx.x % kWave;
    const float* row = x + r * row_len;
    float mn, mx;
    if (mins && maxs) {
        mn = mins[r];
        mx = maxs[r];
    } else {
        mn = INFINITY; mx = -INFINITY;
        for (int64_t i = lane; i < row_len; i += kWave) {
            const float v = row[i];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
        mn = wave_min(mn);
        mx = wave_max(mx);
    }
    if (minmax_out && lane == 0) { minmax_out[2 * r] = mn; minmax_out[2 * r + 1] = mx; }
    const QParams p = qparams_double((double)mn, (double)mx, num_bits, symmetric);
    for (int64_t i = lane; i < row_len; i += kWave) {
        float code;
        const float q = fake_quant_one(row[i], p, &code);
        y[r * row_len + i] = q;
        if (codes) codes[r * row_len + i] = code;
    }
}

// ZeroQ's per-output-channel asymmetric weight quantiser (ZeroQ/utils/quantization_utils/quant_utils.py:85-135
// through quant_modules.py:161-171): float32 tensor arithmetic throughout --
//   scale = (1 / clamp(max - min, 1e-8)) * (2^k - 1);  zp = round(scale * min) + 2^(k-1);
//   q = clamp(round(scale * x - zp), -2^(k-1), 2^(k-1) - 1);  y = (q + zp) / scale.
// One wave per row; the row's own min / max unless given.
__global__ __launch_bounds__(kBlock) void zeroq_quant_rows_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  int64_t rows, int64_t row_len,
                                                                  const float* __restrict__ mins,
                                                                  const float* __restrict__ maxs, int num_bits,
                                                                  float* __restrict__ codes,
                                                                  float* __restrict__ minmax_out) {
    const int64_t r = (int64_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (r >= rows) return;
    const int lane = threadIdx.x % kWave;
    const float* row = x + r * row_len;
    float mn, mx;
    if (mins && maxs) {
        mn = mins[r];
        mx = maxs[r];
    } else {
        mn = INFINITY; mx = -INFINITY;
        for (int64_t i = lane; i < row_len; i += kWave) {
            const float v = row[i];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
        mn = wave_min(mn);
        mx = wave_max(mx);
    }
    if (minmax_out && lane == 0) { minmax_out[2 * r] = mn; minmax_out[2 * r + 1] = mx; }
    const float n = (float)((1ll << num_bits) - 1);
    const float half = (float)(1ll << (num_bits - 1));
    float span = mx - mn;
    span = (span < 1e-8f) ? 1e-8f : span;            // torch.clamp(min=1e-8): NaN stays NaN
    const float scale = (1.0f / span) * n;           // `n / tensor` in torch is tensor.reciprocal() * n: two roundings
    const float zp = rintf(scale * mn) + half;       // .round() is half-to-even
    for (int64_t i = lane; i < row_len; i += kWave) {
        float q = rintf(scale * row[i] - zp);
        q = (q < -half) ? -half : q;                 // torch.clamp propagates NaN
        q = (q > half - 1.0f) ? half - 1.0f : q;
        if (codes) codes[r * row_len + i] = q;
        y[r * row_len + i] = (q + zp) / scale;
    }
}

// bias[o] = eps[o, :] . expect[group(o) * I/g : ...]   (dfq.py:281-287), float64 accumulation, one wave per row
__global__ __launch_bounds__(kBlock) void grouped_matvec_kernel(const float* __restrict__ eps,
                                                                const float* __restrict__ expect, int32_t out_ch,
                                                                int32_t in_per_group, int32_t groups,
                                                                float* __restrict__ out) {
    const int o = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (o >= out_ch) return;
    const int lane = threadIdx.x % kWave;
    const int g = o / (out_ch / groups);
    const float* er = eps + (int64_t)o * in_per_group;
    const float* ex = expect + (int64_t)g * in_per_group;
    double acc = 0.0;
    for (int i = lane; i < in_per_group; i += kWave) acc += (double)er[i] * (double)ex[i];
    acc = wave_sum(acc);
    if (lane == 0) out[o] = (float)acc;
}

}  // namespace dfq

using namespace dfq;

static inline dim3 grid_for(int64_t n, int per_block) { return dim3((unsigned)std::max<int64_t>(1, (n + per_block - 1) / per_block)); }

extern "C" {

int dfq_row_range(const float* w, int64_t rows, int64_t row_len, int32_t signed_range, float* out, void* stream) {
    if (!w || !out || rows <= 0 || row_len <= 0) return fail_arg("dfq_row_range: bad argument");
    hipLaunchKernelGGL(row_range_kernel, grid_for(rows, kBlock / kWave), dim3(kBlock), 0, as_stream(stream), w, rows, row_len,
                       (int)signed_range, out);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_col_range(const float* w2, int32_t out_ch, int32_t in_per_group, int32_t khkw, int32_t groups,
                  int32_t signed_range, float* out, void* stream) {
    if (!w2 || !out || out_ch <= 0 || in_per_group <= 0 || khkw <= 0 || groups <= 0 || out_ch % groups != 0)
        return fail_arg("dfq_col_range: bad argument");
    hipLaunchKernelGGL(col_range_kernel, grid_for((int64_t)groups * in_per_group, kBlock), dim3(kBlock), 0, as_stream(stream),
                       w2, out_ch, in_per_group, khkw, groups, (int)signed_range, out);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_le_solve(const float* r1, const float* r2, int64_t n, float eps, double s_lo, double s_hi, float* S, float* Sinv,
                 void* stream) {
    if (!r1 || !r2 || !S || n <= 0) return fail_arg("dfq_le_solve: bad argument");
    hipLaunchKernelGGL(le_solve_kernel, grid_for(n, kBlock), dim3(kBlock), 0, as_stream(stream), r1, r2, n, eps, (float)s_lo,
                       (float)s_hi, (float)(1.0 / s_lo), (float)(1.0 / s_hi), (int)(s_hi > s_lo), S, Sinv);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_le_apply(float* w1, int32_t o1, int64_t row_len1, float* w2, int32_t o2, int32_t in_per_group2, int32_t khkw2,
                 float* b1, float* bn_weight, float* bn_bias, const float* S, const float* Sinv, void* stream) {
    if (!w1 || !w2 || !S || !Sinv || o1 <= 0 || row_len1 <= 0 || o2 <= 0 || in_per_group2 <= 0 || khkw2 <= 0)
        return fail_arg("dfq_le_apply: bad argument");
    const int groups = (o1 != in_per_group2) ? (o1 / in_per_group2) : 1;      // dfq.py:29-35
    if (groups < 1 || o1 != groups * in_per_group2 || o2 % groups != 0)
        return fail_arg("dfq_le_apply: unsupported pairing O1=%d, I2/g=%d, O2=%d", o1, in_per_group2, o2);
    hipStream_t st = as_stream(stream);
    const int64_t n1 = (int64_t)o1 * row_len1, n2 = (int64_t)o2 * in_per_group2 * khkw2;
    hipLaunchKernelGGL(row_mul_kernel, grid_for(n1, kBlock), dim3(kBlock), 0, st, w1, n1, row_len1, S);            // dfq.py:62
    DFQ_CHECK_LAUNCH();
    if (b1 || bn_weight || bn_bias) {
        hipLaunchKernelGGL(vec_mul3_kernel, grid_for(o1, kBlock), dim3(kBlock), 0, st, bn_weight, bn_bias, b1, (int64_t)o1, S);  // :64-71
        DFQ_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(col_mul_kernel, grid_for(n2, kBlock), dim3(kBlock), 0, st, w2, n2, o2, in_per_group2, khkw2, groups, Sinv);  // :73
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_le_pair(float* w1, int32_t o1, int64_t row_len1, float* w2, int32_t o2, int32_t in_per_group2, int32_t khkw2,
                float* b1, float* bn_weight, float* bn_bias, double s_lo, double s_hi, int32_t signed_range, float eps,
                float* S, float* workspace, void* stream) {
    if (!S || !workspace) return fail_arg("dfq_le_pair: S and a workspace of 3*O1 floats are required");
    if (o1 <= 0 || in_per_group2 <= 0) return fail_arg("dfq_le_pair: bad argument");
    const int groups = (o1 != in_per_group2) ? (o1 / in_per_group2) : 1;
    if (groups < 1 || o1 != groups * in_per_group2 || o2 % groups != 0)
        return fail_arg("dfq_le_pair: unsupported pairing O1=%d, I2/g=%d, O2=%d", o1, in_per_group2, o2);
    float* r1 = workspace;
    float* r2 = workspace + o1;
    float* inv = workspace + 2 * (int64_t)o1;
    int rc;
    if ((rc = dfq_row_range(w1, o1, row_len1, signed_range, r1, stream))) return rc;
    if ((rc = dfq_col_range(w2, o2, in_per_group2, khkw2, groups, signed_range, r2, stream))) return rc;
    if ((rc = dfq_le_solve(r1, r2, o1, eps, s_lo, s_hi, S, inv, stream))) return rc;
    return dfq_le_apply(w1, o1, row_len1, w2, o2, in_per_group2, khkw2, b1, bn_weight, bn_bias, S, inv, stream);
}

size_t dfq_absdiff_mean_scratch_bytes(int64_t n) {
    const int64_t blocks = std::max<int64_t>(1, (n + kDiffChunk - 1) / kDiffChunk);
    return (size_t)blocks * (kBlock / kWave) * sizeof(double);
}

int dfq_absdiff_mean(const float* w, const float* prev, int64_t n, float* out, void* scratch, void* stream) {
    if (!w || !prev || !out || !scratch || n <= 0) return fail_arg("dfq_absdiff_mean: bad argument");
    const int64_t blocks = (n + kDiffChunk - 1) / kDiffChunk;
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(absdiff_partial_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, w, prev, n, (double*)scratch);
    DFQ_CHECK_LAUNCH();
    hipLaunchKernelGGL(absdiff_final_kernel, dim3(1), dim3(kBlock), 0, st, (const double*)scratch, blocks * (kBlock / kWave),
                       (double)n, out);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_fake_quant_rows(const float* x, float* y, int64_t rows, int64_t row_len, const float* mins, const float* maxs,
                        int32_t num_bits, int32_t symmetric, float* codes, float* minmax_out, void* stream) {
    if (!x || !y || rows <= 0 || row_len <= 0 || num_bits < 1 || num_bits > 24 || ((mins == nullptr) != (maxs == nullptr)))
        return fail_arg("dfq_fake_quant_rows: bad argument");
    hipLaunchKernelGGL(fake_quant_rows_kernel, grid_for(rows, kBlock / kWave), dim3(kBlock), 0, as_stream(stream), x, y, rows,
                       row_len, mins, maxs, (int)num_bits, (int)symmetric, codes, minmax_out);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_zeroq_quant_rows(const float* x, float* y, int64_t rows, int64_t row_len, const float* mins, const float* maxs,
                         int32_t num_bits, float* codes, float* minmax_out, void* stream) {
    if (!x || !y || rows <= 0 || row_len <= 0 || num_bits < 2 || num_bits > 24 || ((mins == nullptr) != (maxs == nullptr)))
        return fail_arg("dfq_zeroq_quant_rows: bad argument");
    hipLaunchKernelGGL(zeroq_quant_rows_kernel, grid_for(rows, kBlock / kWave), dim3(kBlock), 0, as_stream(stream), x, y, rows,
                       row_len, mins, maxs, (int)num_bits, codes, minmax_out);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_grouped_matvec(const float* eps, const float* expect, int32_t out_ch, int32_t in_per_group, int32_t groups,
                       float* out, void* stream) {
    if (!eps || !expect || !out || out_ch <= 0 || in_per_group <= 0 || groups <= 0 || out_ch % groups != 0)
        return fail_arg("dfq_grouped_matvec: bad argument");
    hipLaunchKernelGGL(grouped_matvec_kernel, grid_for(out_ch, kBlock / kWave), dim3(kBlock), 0, as_stream(stream), eps, expect,
                       out_ch, in_per_group, groups, out);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

}  // extern "C"
