// Row / column rescale helpers: BatchNorm folding (utils/layer_transform.py:231-276), scale merging
// (utils/quantize.py:145-174, :269-289), bias absorption (dfq.py:121-164), weight clipping
// (dfq.py:167-170).  One-shot streaming kernels; each element is read once and written once.
#include "dfq_common.hpp"

namespace dfq {

__device__ __forceinline__ float apply_op(float a, float b, int op) {
    switch (op) {
        case 0: return a * b;
        case 1: return a / b;
        case 2: return a + b;
        default: return a - b;
    }
}

// one workgroup per row chunk: grid (chunks_per_row, rows)
// 1-D grid of rows x chunks workgroups (a 2-D grid would cap the row count at 65535: a large classifier or
// embedding Linear has more output channels than that)
__global__ __launch_bounds__(kBlock) void scale_rows_kernel(float* __restrict__ w, int64_t row_len, int chunks,
                                                            const float* __restrict__ s, int op) {
    const int64_t row = blockIdx.x / (unsigned)chunks;
    const int chunk = (int)(blockIdx.x - row * chunks);
    const float sv = s[row];
    float* p = w + row * row_len;
    for (int64_t i = (int64_t)chunk * kBlock + threadIdx.x; i < row_len; i += (int64_t)chunks * kBlock)
        p[i] = apply_op(p[i], sv, op);
}

// element (o, i, k) of [O, I/g, khkw] -> input channel (o / (O/groups)) * I/g + i
__global__ __launch_bounds__(kBlock) void scale_cols_kernel(float* __restrict__ w, int64_t row_len, int chunks, int in_per_group,
                                                            int khkw, int out_per_group,
                                                            const float* __restrict__ s, int op) {
    const int64_t o = blockIdx.x / (unsigned)chunks;
    const int chunk = (int)(blockIdx.x - o * chunks);
    const int g = (int)(o / out_per_group);
    const float* sg = s + (int64_t)g * in_per_group;
    float* p = w + o * row_len;
    for (int64_t e = (int64_t)chunk * kBlock + threadIdx.x; e < row_len; e += (int64_t)chunks * kBlock) {
        const int i = (int)(e / khkw);
        p[e] = apply_op(p[e], sg[i], op);
    }
}

__global__ __launch_bounds__(kBlock) void vec_op_kernel(float* __restrict__ y, const float* __restrict__ s, int64_t n, int op) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        y[i] = apply_op(y[i], s[i], op);
}

__global__ __launch_bounds__(kBlock) void clamp_kernel(float* __restrict__ x, int64_t n, float lo, float hi) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        float v = x[i];
        v = (v < lo) ? lo : v;
        v = (v > hi) ? hi : v;
        x[i] = v;
    }
}

// layer_transform.py:246-272, per-channel part.  `var` receives k = gamma/sqrt(var+eps) as a
// temporary for the row-scale launch that follows; bn_reset_kernel restores the identity BN.
__global__ __launch_bounds__(kBlock) void bn_fold_vec_kernel(float* __restrict__ b, int n, float* __restrict__ gamma,
                                                             float* __restrict__ beta, float* __restrict__ mean,
                                                             float* __restrict__ var, float bn_eps,
                                                             float* __restrict__ fake_weight,
                                                             float* __restrict__ fake_bias) {
    const int o = blockIdx.x * kBlock + threadIdx.x;
    if (o >= n) return;
    const float g = gamma[o], bt = beta[o], mu = mean[o], vr = var[o];
    const float sd = sqrtf(vr + bn_eps);
    const float k = g / sd;
    const float gm = g * mu;
    const float shift = bt - gm / sd;
    const float bk = b[o] * k;
    b[o] = bk + shift;
    fake_weight[o] = fabsf(g);
    fake_bias[o] = bt;
    var[o] = k;
    gamma[o] = 1.0f;
    beta[o] = 0.0f;
    mean[o] = 0.0f;
}

__global__ void fill_kernel(float* __restrict__ x, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}

// dfq.py:148-157:  c = clamp(beta~ - N*gamma~, 0);  wc[o] = sum_i (sum_k W2[o,i,k]) * c[g*step_i + i]
// one wave per output row; float64 accumulation of the matvec, float32 sequential sum over k.
__global__ __launch_bounds__(kBlock) void absorb_matvec_kernel(const float* __restrict__ w2, int o2, int in_per_group,
                                                               int khkw, int step_o, const float* __restrict__ bn_weight,
                                                               const float* __restrict__ bn_bias, float n_sigma,
                                                               float* __restrict__ b2) {
    const int lane = threadIdx.x % kWave;
    const int o = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (o >= o2) return;
    const int g = o / step_o;
    const float* row = w2 + (int64_t)o * in_per_group * khkw;
    double acc = 0.0;
    for (int i = lane; i < in_per_group; i += kWave) {
        float ws = 0.0f;
        for (int k = 0; k < khkw; ++k) ws = ws + row[(int64_t)i * khkw + k];
        const int ch = g * in_per_group + i;
        const float nw = n_sigma * bn_weight[ch];
        float c = bn_bias[ch] - nw;
        c = (c < 0.0f) ? 0.0f : c;
        acc += (double)ws * (double)c;
    }
    acc = wave_sum(acc);
    if (lane == 0) b2[o] = b2[o] + (float)acc;
}

__global__ __launch_bounds__(kBlock) void absorb_shift_kernel(float* __restrict__ b1, const float* __restrict__ bn_weight,
                                                              float* __restrict__ bn_bias, int n, float n_sigma) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float nw = n_sigma * bn_weight[i];
    float c = bn_bias[i] - nw;
    c = (c < 0.0f) ? 0.0f : c;
    const float neg = -c;
    b1[i] = b1[i] + neg;
    bn_bias[i] = bn_bias[i] + neg;
}

static int grid1(int64_t n, int cap) {
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace dfq

using namespace dfq;

extern "C" {

int dfq_scale_rows(float* w, int32_t rows, int64_t row_len, const float* s, int32_t op, void* stream) {
    if (!w || !s || rows <= 0 || row_len <= 0 || op < 0 || op > 3) return fail_arg("dfq_scale_rows: bad argument");
    const int chunks = grid1(row_len, 64);
    if ((int64_t)rows * chunks > 0x7fffffff) return fail_arg("dfq_scale_rows: rows=%d too many", rows);
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((int64_t)rows * chunks)), dim3(kBlock), 0, as_stream(stream), w, row_len,
                       chunks, s, (int)op);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_scale_cols(float* w, int32_t out_ch, int32_t in_per_group, int32_t khkw, int32_t groups,
                   const float* s, int32_t op, void* stream) {
    if (!w || !s || out_ch <= 0 || in_per_group <= 0 || khkw <= 0 || groups <= 0 || out_ch % groups != 0 || op < 0 || op > 3)
        return fail_arg("dfq_scale_cols: bad argument");
    const int64_t row_len = (int64_t)in_per_group * khkw;
    const int chunks = grid1(row_len, 64);
    if ((int64_t)out_ch * chunks > 0x7fffffff) return fail_arg("dfq_scale_cols: out_ch=%d too many", out_ch);
    hipLaunchKernelGGL(scale_cols_kernel, dim3((unsigned)((int64_t)out_ch * chunks)), dim3(kBlock), 0, as_stream(stream), w, row_len,
                       chunks, (int)in_per_group, (int)khkw, (int)(out_ch / groups), s, (int)op);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_vec_op(float* y, const float* s, int64_t n, int32_t op, void* stream) {
    if (!y || !s || n <= 0 || op < 0 || op > 3) return fail_arg("dfq_vec_op: bad argument");
    hipLaunchKernelGGL(vec_op_kernel, dim3(grid1(n, 2048)), dim3(kBlock), 0, as_stream(stream), y, s, n, (int)op);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_clamp(float* x, int64_t n, float lo, float hi, void* stream) {
    if (!x || n <= 0) return fail_arg("dfq_clamp: bad argument");
    hipLaunchKernelGGL(clamp_kernel, dim3(grid1(n, 2048)), dim3(kBlock), 0, as_stream(stream), x, n, lo, hi);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_fold_batchnorm(float* w, float* b, int32_t out_ch, int64_t row_len, float* gamma, float* beta,
                       float* mean, float* var, float bn_eps, float* fake_weight, float* fake_bias, void* stream) {
    if (!w || !b || !gamma || !beta || !mean || !var || !fake_weight || !fake_bias || out_ch <= 0 || row_len <= 0)
        return fail_arg("dfq_fold_batchnorm: bad argument");
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(bn_fold_vec_kernel, dim3(grid1(out_ch, 65535)), dim3(kBlock), 0, st, b, (int)out_ch, gamma, beta,
                       mean, var, bn_eps, fake_weight, fake_bias);
    DFQ_CHECK_LAUNCH();
    int rc = dfq_scale_rows(w, out_ch, row_len, var, 0, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(fill_kernel, dim3(grid1(out_ch, 65535)), dim3(kBlock), 0, st, var, (int)out_ch, 1.0f);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_bias_absorb(const float* w2, int32_t o2, int32_t in_per_group, int32_t khkw, int32_t o1, float* b1,
                    float* b2, const float* bn_weight, float* bn_bias, float n_sigma, void* stream) {
    if (!w2 || !b1 || !b2 || !bn_weight || !bn_bias || o2 <= 0 || in_per_group <= 0 || khkw <= 0 || o1 <= 0)
        return fail_arg("dfq_bias_absorb: bad argument");
    const int num_group = o1 / in_per_group;     // dfq.py:144
    if (num_group < 1 || num_group * in_per_group != o1 || o2 % num_group != 0)
        return fail_arg("dfq_bias_absorb: unsupported geometry O1=%d I2/g=%d O2=%d", o1, in_per_group, o2);
    hipStream_t st = as_stream(stream);
    const int wpb = kBlock / kWave;
    hipLaunchKernelGGL(absorb_matvec_kernel, dim3((o2 + wpb - 1) / wpb), dim3(kBlock), 0, st, w2, (int)o2, (int)in_per_group,
                       (int)khkw, (int)(o2 / num_group), bn_weight, (const float*)bn_bias, n_sigma, b2);
    DFQ_CHECK_LAUNCH();
    hipLaunchKernelGGL(absorb_shift_kernel, dim3(grid1(o1, 65535)), dim3(kBlock), 0, st, b1, bn_weight, bn_bias, (int)o1, n_sigma);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

}  // extern "C"
