// Analytic activation ranges from the BatchNorm proxies (utils/layer_transform.py:347-609,
// set_quant_minmax) for gfx950: the per-channel arithmetic and the reductions to (min, max).  The
// graph walk that decides WHICH proxies feed a quantiser (find_prev_bn, branch grouping) is host
// logic, as in the reference; what reaches the device are channel vectors of at most a few thousand
// floats, so the design goal is few launches and one read-back, not bandwidth:
//   * dfq_bn_ranges       every "one BN -> one quantiser" case of a network in ONE launch (one
//                         workgroup per request), results in one [n][2] array;
//   * dfq_relu_moments    mean / variance of N(beta, gamma^2) pushed through ReLU or ReLU6
//                         (layer_transform.py:407-418), optionally accumulated (residual adds);
//   * dfq_moments_after_add   the same transform applied to an accumulated (mean, var) pair when the
//                         add node itself is followed by a ReLU / ReLU6 (:533-540);
//   * dfq_moment_range    min(mean - N*sd), max(mean + N*sd) with sd = sqrt(var + eps) (:571-573);
//   * dfq_bn_through_layer    a BN proxy vector pushed through a conv / linear layer that has no BN
//                         of its own (case d, :451-466).
// Float32 arithmetic in the reference's operation order; pdf / cdf in float64 rounded to float32.
#include <algorithm>
#include <vector>

#include "dfq_common.hpp"

namespace dfq {

struct BnRangeReq {
    const float* fw;
    const float* fb;
    int32_t channels;
    int32_t relu_mode;     // 0 none, 1 ReLU, 2 ReLU6
};

// torch.min / torch.max propagate NaN
__device__ __forceinline__ float nan_min(float a, float b) { return (a < b || a != a) ? a : b; }
__device__ __forceinline__ float nan_max(float a, float b) { return (a > b || a != a) ? a : b; }

__device__ __forceinline__ void block_minmax(float& mn, float& mx, float* sh) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        mn = nan_min(mn, __shfl_xor(mn, m));
        mx = nan_max(mx, __shfl_xor(mx, m));
    }
    const int wave = threadIdx.x / kWave;
    if (threadIdx.x % kWave == 0) { sh[2 * wave] = mn; sh[2 * wave + 1] = mx; }
    __syncthreads();
    mn = sh[0]; mx = sh[1];
#pragma unroll
    for (int w = 1; w < kBlock / kWave; ++w) { mn = nan_min(mn, sh[2 * w]); mx = nan_max(mx, sh[2 * w + 1]); }
}

// get_min_value / get_max_value (layer_transform.py:403-404) + the ReLU clamps of :468-469, one request
// per workgroup
__global__ __launch_bounds__(kBlock) void bn_ranges_kernel(const BnRangeReq* __restrict__ reqs, float n_sigma,
                                                           float* __restrict__ out) {
    __shared__ float sh[2 * (kBlock / kWave)];
    const BnRangeReq r = reqs[blockIdx.x];
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < r.channels; i += kBlock) {
        const float nw = n_sigma * r.fw[i];
        mn = nan_min(r.fb[i] - nw, mn);
        mx = nan_max(r.fb[i] + nw, mx);
    }
    block_minmax(mn, mx, sh);
    if (threadIdx.x == 0) {
        if (r.relu_mode >= 1) mn = (mn > 0.0f) ? mn : 0.0f;     // Python max(0., v): NaN -> 0
        if (r.relu_mode == 2) mx = (mx < 6.0f) ? mx : 6.0f;     // Python min(6., v): NaN -> 6
        out[2 * blockIdx.x + 0] = mn;
        out[2 * blockIdx.x + 1] = mx;
    }
}

// calculate_mean / calculate_var (:407-410)
__device__ __forceinline__ void moments_relu(float w, float b, float& mean, float& var) {
    const float t = (-b) / w;
    float pdf, cdf;
    normal_pdf_cdf(t, pdf, cdf);
    const float one_m = 1.0f - cdf;
    mean = w * pdf + b * one_m;
    const float poly = ((b * b + w * w) + mean * mean) - (2.0f * mean) * b;
    const float t1 = one_m * poly;
    const float t2 = (w * (b - 2.0f * mean)) * pdf;
    const float t3 = (mean * mean) * cdf;
    var = (t1 + t2) + t3;
}

// calculate_mean_6 / calculate_var_6 (:411-418)
__device__ __forceinline__ void moments_relu6(float w, float b, float& mean, float& var) {
    const float lo = (-b) / w;
    const float hi = (6.0f - b) / w;
    float pdf_lo, cdf_lo, pdf_hi, cdf_hi;
    normal_pdf_cdf(lo, pdf_lo, cdf_lo);
    normal_pdf_cdf(hi, pdf_hi, cdf_hi);
    const float dp = pdf_lo - pdf_hi;
    const float dc = cdf_hi - cdf_lo;
    const float top = 1.0f - cdf_hi;
    mean = (w * dp + b * dc) + 6.0f * top;
    const float poly = ((b * b + w * w) + mean * mean) - (2.0f * mean) * b;
    const float t1 = dc * poly;
    const float t2 = (w * -6.0f) * pdf_hi;
    const float t3 = (w * (b - 2.0f * mean)) * dp;
    const float t4 = (mean * mean) * cdf_lo;
    const float d6 = 6.0f - mean;
    const float t5 = (d6 * d6) * top;
    var = (((t1 + t2) + t3) + t4) + t5;
}

__device__ __forceinline__ void moments_of(int mode, float w, float b, float& mean, float& var) {
    if (mode == 1) moments_relu(w, b, mean, var);
    else if (mode == 2) moments_relu6(w, b, mean, var);
    else { mean = b; var = w * w; }                     // :505-507
}

__global__ __launch_bounds__(kBlock) void relu_moments_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                                              int64_t n, int mode, float* __restrict__ mean,
                                                              float* __restrict__ var, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float m, v;
    moments_of(mode, w[i], b[i], m, v);
    if (accumulate) { m = mean[i] + m; v = var[i] + v; }    // mean += mean_tmp; var += var_tmp (:521-531)
    mean[i] = m;
    var[i] = v;
}

__global__ __launch_bounds__(kBlock) void moments_after_add_kernel(float* __restrict__ mean, float* __restrict__ var,
                                                                   int64_t n, int mode, float eps) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float sd = sqrtf(var[i] + eps);
    float m, v;
    moments_of(mode, sd, mean[i], m, v);                     // :533-540
    mean[i] = m;
    var[i] = v;
}

__global__ __launch_bounds__(kBlock) void moment_range_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                                              int64_t n, float eps, float n_sigma, float* __restrict__ out) {
    __shared__ float sh[2 * (kBlock / kWave)];
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t i = threadIdx.x; i < n; i += kBlock) {
        const float nw = n_sigma * sqrtf(var[i] + eps);
        mn = nan_min(mean[i] - nw, mn);
        mx = nan_max(mean[i] + nw, mx);
    }
    block_minmax(mn, mx, sh);
    if (threadIdx.x == 0) { out[0] = mn; out[1] = mx; }
}

// v_out[o] = sum_i (sum_k W[o, i, k]) * v_in[group(o) * I/g + i] + bias[o]: F.conv2d of a [1, C, 1, 1] vector with
// the kernel-summed weight, or F.linear (:455-463).  One wave per output row, float64 accumulation.
__global__ __launch_bounds__(kBlock) void bn_through_layer_kernel(const float* __restrict__ w, int32_t out_ch,
                                                                  int32_t in_per_group, int32_t khkw, int32_t groups,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ v_in,
                                                                  float* __restrict__ v_out) {
    const int o = blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave;
    if (o >= out_ch) return;
    const int lane = threadIdx.x % kWave;
    const int g = o / (out_ch / groups);
    const float* row = w + (int64_t)o * in_per_group * khkw;
    const float* vin = v_in + (int64_t)g * in_per_group;
    double acc = 0.0;
    for (int i = lane; i < in_per_group; i += kWave) {
        float ws = 0.0f;
        for (int k = 0; k < khkw; ++k) ws = ws + row[(int64_t)i * khkw + k];     // layer_weight.view(O, I, -1).sum(-1)
        acc += (double)ws * (double)vin[i];
    }
    acc = wave_sum(acc);
    if (lane == 0) v_out[o] = (float)acc + (bias ? bias[o] : 0.0f);
}

}  // namespace dfq

using namespace dfq;

extern "C" {

int dfq_bn_ranges(const dfq_bn_range_req* reqs, int32_t n_reqs, float n_sigma, float* out, void* scratch, void* stream) {
    if (!reqs || n_reqs <= 0 || !out || !scratch) return fail_arg("dfq_bn_ranges: bad argument");
    std::vector<BnRangeReq> h(n_reqs);
    for (int i = 0; i < n_reqs; ++i) {
        if (!reqs[i].fake_weight || !reqs[i].fake_bias || reqs[i].channels <= 0 || reqs[i].relu_mode < 0 || reqs[i].relu_mode > 2)
            return fail_arg("dfq_bn_ranges: request %d is malformed", i);
        h[i].fw = reqs[i].fake_weight; h[i].fb = reqs[i].fake_bias; h[i].channels = reqs[i].channels; h[i].relu_mode = reqs[i].relu_mode;
    }
    hipStream_t st = as_stream(stream);
    // the request table travels through `scratch` (n_reqs * 24 bytes of device memory); the copy is from a
    // temporary, so it is completed before returning
    DFQ_HIP_TRY(hipMemcpyAsync(scratch, h.data(), sizeof(BnRangeReq) * n_reqs, hipMemcpyHostToDevice, st));
    DFQ_HIP_TRY(hipStreamSynchronize(st));
    hipLaunchKernelGGL(bn_ranges_kernel, dim3(n_reqs), dim3(kBlock), 0, st, (const BnRangeReq*)scratch, n_sigma, out);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

size_t dfq_bn_ranges_scratch_bytes(int32_t n_reqs) { return sizeof(BnRangeReq) * (size_t)std::max(1, n_reqs); }

int dfq_relu_moments(const float* weight, const float* bias, int64_t n, int32_t relu_mode, float* mean, float* var,
                     int32_t accumulate, void* stream) {
    if (!weight || !bias || !mean || !var || n <= 0 || relu_mode < 0 || relu_mode > 2) return fail_arg("dfq_relu_moments: bad argument");
    hipLaunchKernelGGL(relu_moments_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, as_stream(stream), weight,
                       bias, n, (int)relu_mode, mean, var, (int)accumulate);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_moments_after_add(float* mean, float* var, int64_t n, int32_t relu_mode, float eps, void* stream) {
    if (!mean || !var || n <= 0 || relu_mode < 1 || relu_mode > 2) return fail_arg("dfq_moments_after_add: bad argument");
    hipLaunchKernelGGL(moments_after_add_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, as_stream(stream),
                       mean, var, n, (int)relu_mode, eps);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_moment_range(const float* mean, const float* var, int64_t n, float eps, float n_sigma, float* out2, void* stream) {
    if (!mean || !var || !out2 || n <= 0) return fail_arg("dfq_moment_range: bad argument");
    hipLaunchKernelGGL(moment_range_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), mean, var, n, eps, n_sigma, out2);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_bn_through_layer(const float* weight, int32_t out_ch, int32_t in_per_group, int32_t khkw, int32_t groups,
                         const float* bias, const float* v_in, float* v_out, void* stream) {
    if (!weight || !v_in || !v_out || out_ch <= 0 || in_per_group <= 0 || khkw <= 0 || groups <= 0 || out_ch % groups != 0)
        return fail_arg("dfq_bn_through_layer: bad argument");
    hipLaunchKernelGGL(bn_through_layer_kernel, dim3((unsigned)((out_ch + kBlock / kWave - 1) / (kBlock / kWave))), dim3(kBlock), 0,
                       as_stream(stream), weight, out_ch, in_per_group, khkw, groups, bias, v_in, v_out);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

}  // extern "C"
