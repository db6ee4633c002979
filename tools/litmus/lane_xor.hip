// Hardware check of xor_lane_minmax<M> (dfq_amd/csrc/dfq_common.hpp): the register-file butterfly steps (DPP quad permutes,
// DPP row shifts with bank masks, v_permlane16_swap / v_permlane32_swap) against __shfl_xor (ds_bpermute) on random data, and
// the complete 64-lane reductions.  Prints "N mismatches".  Built by __graft_entry__.build(), run by tests/test_litmus.py.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../dfq_amd/csrc/dfq_common.hpp"

using namespace dfq;

template <int M>
__device__ void check_step(float a, float b, unsigned* errors) {
    float mn = a, mx = b;
    xor_lane_minmax<M>(mn, mx);
    const float want_mn = fminf(a, __shfl_xor(a, M)), want_mx = fmaxf(b, __shfl_xor(b, M));
    if (__float_as_uint(mn) != __float_as_uint(want_mn) || __float_as_uint(mx) != __float_as_uint(want_mx)) atomicAdd(errors, 1u);
}

__global__ void lane_xor_kernel(const float* x, const float* y, unsigned* errors) {
    const float a = x[blockIdx.x * blockDim.x + threadIdx.x], b = y[blockIdx.x * blockDim.x + threadIdx.x];
    check_step<1>(a, b, errors); check_step<2>(a, b, errors); check_step<4>(a, b, errors);
    check_step<8>(a, b, errors); check_step<16>(a, b, errors); check_step<32>(a, b, errors);
    float mn = a, mx = b;
    wave_minmax(mn, mx);
    float wmn = a, wmx = b;
    for (int m = 32; m >= 1; m >>= 1) { wmn = fminf(wmn, __shfl_xor(wmn, m)); wmx = fmaxf(wmx, __shfl_xor(wmx, m)); }
    if (__float_as_uint(mn) != __float_as_uint(wmn) || __float_as_uint(mx) != __float_as_uint(wmx)) atomicAdd(errors, 1u);
    // float64 sums: every step and the whole butterfly (same order, so bit-identical)
    const double d = (double)a * 1.25 + (double)b;
    double s1 = d; xor_lane_add<1>(s1);   if (s1 != d + __shfl_xor(d, 1)) atomicAdd(errors, 1u);
    double s2 = d; xor_lane_add<2>(s2);   if (s2 != d + __shfl_xor(d, 2)) atomicAdd(errors, 1u);
    double s4 = d; xor_lane_add<4>(s4);   if (s4 != d + __shfl_xor(d, 4)) atomicAdd(errors, 1u);
    double s8 = d; xor_lane_add<8>(s8);   if (s8 != d + __shfl_xor(d, 8)) atomicAdd(errors, 1u);
    double s16 = d; xor_lane_add<16>(s16); if (s16 != d + __shfl_xor(d, 16)) atomicAdd(errors, 1u);
    double s32 = d; xor_lane_add<32>(s32); if (s32 != d + __shfl_xor(d, 32)) atomicAdd(errors, 1u);
    double ws = d;
    for (int m = 32; m >= 1; m >>= 1) ws += __shfl_xor(ws, m);
    if (wave_sum(d) != ws) atomicAdd(errors, 1u);
}

int main() {
    const int n = 256 * 64;
    std::vector<float> hx(n), hy(n);
    srand(7);
    for (int i = 0; i < n; ++i) { hx[i] = (float)((double)rand() / (double)RAND_MAX) * 20.f - 10.f; hy[i] = (float)((double)rand() / (double)RAND_MAX) * 20.f - 10.f; }
    float *x, *y; unsigned* e; unsigned he = 0;
    if (hipMalloc(&x, n * 4) != hipSuccess || hipMalloc(&y, n * 4) != hipSuccess || hipMalloc(&e, 4) != hipSuccess) { printf("no device memory\n"); return 2; }
    hipMemcpy(x, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(y, hy.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(e, 0, 4);
    hipLaunchKernelGGL(lane_xor_kernel, dim3(n / 256), dim3(256), 0, 0, x, y, e);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(&he, e, 4, hipMemcpyDeviceToHost);
    printf("%u mismatches\n", he);
    return he ? 1 : 0;
}
