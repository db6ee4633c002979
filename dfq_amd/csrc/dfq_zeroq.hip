// BatchNorm-statistics loss of ZeroQ's data distillation (ZeroQ/distill_data.py:170-196) for gfx950.
//
// For every BN layer the reference takes the BN's input x [N, C, H, W] and compares, per sample and channel,
// the spatial mean and the (unbiased) spatial standard deviation of x + eps with the BN's running statistics:
//     mean_loss = || bn_mean - mean_hw(x) ||^2 / denom,  std_loss = || bn_std - std_hw(x + eps) ||^2 / denom
// (own_loss, distill_data.py:40-45: denom = A.size(0), i.e. C for the BN terms where the [C] statistics come first
// and broadcast over the N samples, N for the input-batch term of :192-196), back-propagated into the
// synthetic input batch.  In eager PyTorch that is ~10 passes over every activation of the network per
// iteration; here it is ONE read of x for the forward (row sums in float64: mean, std and both losses) and one
// read + one write for the backward (the gradient is an affine function of x per (n, c) row).  HBM-bound:
// 4 B per element forward, 8 B backward.  The convolutions around it stay in MIOpen.
#include <algorithm>

#include "dfq_common.hpp"

namespace dfq {

// rows of H*W floats; one wave per row for short rows, one workgroup per row for long ones
template <int kRowsPerBlock>
__global__ __launch_bounds__(kBlock) void bn_stat_rows_kernel(const float* __restrict__ x, int64_t rows, int64_t hw,
                                                              int32_t channels, const float* __restrict__ bn_mean,
                                                              const float* __restrict__ bn_std, float eps,
                                                              float* __restrict__ row_mean, float* __restrict__ row_std,
                                                              double* __restrict__ row_terms, int terms) {
    __shared__ double sh1[kBlock / kWave], sh2[kBlock / kWave];
    const int sub = (kRowsPerBlock == 1) ? 0 : threadIdx.x / kWave;
    const int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + sub;
    const bool live = r < rows;
    const int lanes = (kRowsPerBlock == 1) ? kBlock : kWave;
    const int t = (kRowsPerBlock == 1) ? threadIdx.x : threadIdx.x % kWave;
    const float* row = x + (live ? r : 0) * hw;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int64_t i = t; i < hw; i += lanes) {
        const float v = row[i];
        const float e = v + eps;                       // torch.std(x + eps): the shift is rounded in float32
        s0 += (double)v;
        s1 += (double)e;
        s2 += (double)e * (double)e;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (kRowsPerBlock == 1) {
        // three block-wide sums in a fixed order
        const int wave = threadIdx.x / kWave;
        double tot[3];
        double part[3] = {s0, s1, s2};
        for (int q = 0; q < 3; ++q) {
            __syncthreads();
            if (threadIdx.x % kWave == 0) sh1[wave] = part[q];
            __syncthreads();
            double a = 0.0;
            for (int w = 0; w < kBlock / kWave; ++w) a += sh1[w];
            tot[q] = a;
        }
        s0 = tot[0]; s1 = tot[1]; s2 = tot[2];
        (void)sh2;
    }
    if (live && t == 0) {
        // `terms`: bit 0 = mean term, bit 1 = std term (the H*W == 1 branch takes them from two different row views)
        const double n = (double)hw;
        const int c = (int)(r % channels);
        if (terms & 1) {
            const float mean = (float)(s0 / n);
            row_mean[r] = mean;
            const float dm = bn_mean[c] - mean;
            row_terms[2 * r + 0] = (double)dm * (double)dm;
        }
        if (terms & 2) {
            const double me = s1 / n;
            const float sd = (float)sqrt((s2 - s1 * me) / (n - 1.0));      // unbiased, like torch.std
            row_std[r] = sd;
            const float ds = bn_std[c] - sd;
            row_terms[2 * r + 1] = (double)ds * (double)ds;
        }
    }
}

// loss2[0] = sum_r terms[2r] / denom, loss2[1] = sum_r terms[2r+1] / denom, fixed summation order
__global__ __launch_bounds__(kBlock) void bn_stat_reduce_kernel(const double* __restrict__ row_terms, int64_t rows,
                                                                int64_t std_rows, double denom, float* __restrict__ loss2) {
    __shared__ double sh[kBlock / kWave];
    double a = 0.0, b = 0.0;
    for (int64_t r = threadIdx.x; r < rows; r += kBlock) a += row_terms[2 * r];
    for (int64_t r = threadIdx.x; r < std_rows; r += kBlock) b += row_terms[2 * r + 1];
    a = block_sum(a, sh);
    b = block_sum(b, sh);
    if (threadIdx.x == 0) { loss2[0] = (float)(a / denom); loss2[1] = (float)(b / denom); }
}

// d(mean_loss)/dx = 2 (mean - bn_mean) / (C * HW);   d(std_loss)/dx = 2 (std - bn_std) / C * (x + eps - mean_e) / ((HW-1) std)
__global__ __launch_bounds__(kBlock) void bn_stat_backward_kernel(const float* __restrict__ x, int64_t rows, int64_t hw,
                                                                  int32_t channels, const float* __restrict__ bn_mean,
                                                                  const float* __restrict__ bn_std, float eps,
                                                                  const float* __restrict__ row_mean,
                                                                  const float* __restrict__ row_std, float denom, float g_mean,
                                                                  float g_std, const float* __restrict__ g_pair,
                                                                  float* __restrict__ grad, int accumulate, int terms) {
    const int64_t r = blockIdx.x;                          // one workgroup per (sample, channel) row
    const int c = (int)(r % channels);
    if (g_pair) { g_mean = g_pair[0]; g_std = g_pair[1]; }   // upstream gradients left on the device by autograd
    // the spatial mean of the row: stored by the forward when it took the mean term over this row view, else recomputed
    float mean;
    if (terms & 1) {
        mean = row_mean[r];
    } else {
        __shared__ double shm[kBlock / kWave];
        double s0 = 0.0;
        for (int64_t i = threadIdx.x; i < hw; i += kBlock) s0 += (double)x[r * hw + i];
        mean = (float)(block_sum(s0, shm) / (double)hw);
    }
    const float sd = (terms & 2) ? row_std[r] : 1.0f;
    const float a = (terms & 1) ? g_mean * 2.0f * (mean - bn_mean[c]) / (denom * (float)hw) : 0.0f;
    const float b = (terms & 2) ? g_std * 2.0f * (sd - bn_std[c]) / (denom * ((float)hw - 1.0f) * sd) : 0.0f;
    const float mean_e = mean + eps;
    for (int64_t i = threadIdx.x; i < hw; i += kBlock) {
        const float g = a + b * ((x[r * hw + i] + eps) - mean_e);
        grad[r * hw + i] = accumulate ? grad[r * hw + i] + g : g;
    }
}

}  // namespace dfq

using namespace dfq;

extern "C" {

size_t dfq_bn_stat_loss_scratch_bytes(int64_t rows) { return sizeof(double) * 2 * (size_t)std::max<int64_t>(1, rows); }

int dfq_bn_stat_loss_forward(const float* x, int64_t rows, int64_t hw, int32_t channels, const float* bn_mean,
                             const float* bn_std, float eps, float denom, float* row_mean, float* row_std, float* loss2,
                             void* scratch, void* stream) {
    if (!x || !bn_mean || !bn_std || !row_mean || !row_std || !loss2 || !scratch || rows <= 0 || channels <= 0 || rows % channels != 0)
        return fail_arg("dfq_bn_stat_loss_forward: bad argument");
    if (hw < 1) return fail_arg("dfq_bn_stat_loss_forward: H*W must be >= 1");
    hipStream_t st = as_stream(stream);
    auto launch_rows = [&](int64_t n_rows, int64_t len, int terms) {
        if (len >= 1024)
            hipLaunchKernelGGL(bn_stat_rows_kernel<1>, dim3((unsigned)n_rows), dim3(kBlock), 0, st, x, n_rows, len, channels, bn_mean, bn_std,
                               eps, row_mean, row_std, (double*)scratch, terms);
        else
            hipLaunchKernelGGL(bn_stat_rows_kernel<kBlock / kWave>, dim3((unsigned)((n_rows + kBlock / kWave - 1) / (kBlock / kWave))),
                               dim3(kBlock), 0, st, x, n_rows, len, channels, bn_mean, bn_std, eps, row_mean, row_std, (double*)scratch, terms);
    };
    int64_t std_rows = rows;
    if (hw == 1) {
        // distill_data.py:181-182: one value per (sample, channel).  The mean term compares every x[n, c] with bn_mean[c];
        // for the std term the reference reinterprets the contiguous [N, C] block as C rows of N values
        // (`tmp_output.view(C, -1)`) and compares the unbiased std of row r with bn_std[r].
        const int64_t n_samples = rows / channels;
        if (n_samples < 2) return fail_arg("dfq_bn_stat_loss_forward: H*W == 1 needs a batch of >= 2 (std over the batch axis)");
        launch_rows(rows, 1, 1);
        DFQ_CHECK_LAUNCH();
        launch_rows(channels, n_samples, 2);
        std_rows = channels;
    } else {
        launch_rows(rows, hw, 3);
    }
    DFQ_CHECK_LAUNCH();
    hipLaunchKernelGGL(bn_stat_reduce_kernel, dim3(1), dim3(kBlock), 0, st, (const double*)scratch, rows, std_rows, (double)denom, loss2);
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

static int bn_stat_backward_impl(const float* x, int64_t rows, int64_t hw, int32_t channels, const float* bn_mean,
                                 const float* bn_std, float eps, float denom, const float* row_mean, const float* row_std,
                                 float g_mean, float g_std, const float* g_pair, float* grad_x, int32_t accumulate, void* stream) {
    if (!x || !bn_mean || !bn_std || !row_mean || !row_std || !grad_x || rows <= 0 || hw < 1 || channels <= 0 || rows > 0x7fffffff ||
        rows % channels != 0)
        return fail_arg("dfq_bn_stat_loss_backward: bad argument");
    hipStream_t st = as_stream(stream);
    if (hw == 1) {
        const int64_t n_samples = rows / channels;
        if (n_samples < 2) return fail_arg("dfq_bn_stat_loss_backward: H*W == 1 needs a batch of >= 2");
        hipLaunchKernelGGL(bn_stat_backward_kernel, dim3((unsigned)rows), dim3(kBlock), 0, st, x, rows, (int64_t)1, channels, bn_mean,
                           bn_std, eps, row_mean, row_std, denom, g_mean, g_std, g_pair, grad_x, (int)accumulate, 1);
        DFQ_CHECK_LAUNCH();
        hipLaunchKernelGGL(bn_stat_backward_kernel, dim3((unsigned)channels), dim3(kBlock), 0, st, x, (int64_t)channels, n_samples,
                           channels, bn_mean, bn_std, eps, row_mean, row_std, denom, g_mean, g_std, g_pair, grad_x, 1, 2);
    } else {
        hipLaunchKernelGGL(bn_stat_backward_kernel, dim3((unsigned)rows), dim3(kBlock), 0, st, x, rows, hw, channels, bn_mean, bn_std,
                           eps, row_mean, row_std, denom, g_mean, g_std, g_pair, grad_x, (int)accumulate, 3);
    }
    DFQ_CHECK_LAUNCH();
    return DFQ_OK;
}

int dfq_bn_stat_loss_backward(const float* x, int64_t rows, int64_t hw, int32_t channels, const float* bn_mean,
                              const float* bn_std, float eps, float denom, const float* row_mean, const float* row_std,
                              float grad_mean_loss, float grad_std_loss, float* grad_x, int32_t accumulate, void* stream) {
    return bn_stat_backward_impl(x, rows, hw, channels, bn_mean, bn_std, eps, denom, row_mean, row_std, grad_mean_loss,
                                 grad_std_loss, nullptr, grad_x, accumulate, stream);
}

int dfq_bn_stat_loss_backward_dev(const float* x, int64_t rows, int64_t hw, int32_t channels, const float* bn_mean,
                                  const float* bn_std, float eps, float denom, const float* row_mean, const float* row_std,
                                  const float* grad_pair, float* grad_x, int32_t accumulate, void* stream) {
    if (!grad_pair) return fail_arg("dfq_bn_stat_loss_backward_dev: null gradient pair");
    return bn_stat_backward_impl(x, rows, hw, channels, bn_mean, bn_std, eps, denom, row_mean, row_std, 0.0f, 0.0f, grad_pair,
                                 grad_x, accumulate, stream);
}

}  // extern "C"
