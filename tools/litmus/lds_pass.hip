// What does ONE pass over a workgroup's 32 KB LDS tile cost?  The resident equalisation kernel (dfq_le_resident.hip) makes two per
// sweep -- the row statistics of t = w / s_A, then new = t * s_B with |dW| in float64 and the column statistics -- and the
// in-kernel measurement (one more pass per sweep: tools/gpu_r05_ab.sh with -DDFQ_RES_ABLATE=256 / 512) says ~3 us and ~1.4 us for
// ~320 and ~180 wave instructions: twenty cycles per instruction.  This probe runs the same loops alone -- 256 threads, eight
// float4 slots per thread, 1 ... 3 workgroups per CU -- with pieces switched off, to see which piece it is.
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/litmus/lds_pass.hip -o tools/litmus/lds_pass && tools/litmus/lds_pass
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float fvec4 __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256, kSlots = 8, kTile = kBlock * kSlots * 4;

__device__ __forceinline__ float vmin_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// mode bits: 1 float64 |dW| chain, 2 column min/max, 4 write back, 8 per-row factor from LDS, 16: two slots' reads in flight
template <int kMode>
__global__ __launch_bounds__(kBlock, 3) void pass_kernel(const float* src, float* dst, int passes, long long* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* tile = (float*)smem;
    float* sh_s = tile + kTile;
    const int tid = threadIdx.x;
    for (int u = 0; u < kSlots; ++u) *(fvec4*)(tile + (u * kBlock + tid) * 4) = *(const fvec4*)(src + (size_t)blockIdx.x % 64 * kTile + (u * kBlock + tid) * 4);
    if (tid < 64) sh_s[tid] = 1.0f + 1e-7f * tid;
    __syncthreads();
    float iv[4] = {1.0000001f, 0.9999999f, 1.0000002f, 0.9999998f};
    float cmn[4] = {1e30f, 1e30f, 1e30f, 1e30f}, cmx[4] = {-1e30f, -1e30f, -1e30f, -1e30f};
    double acc = 0.0;
    const long long t0 = wall_clock64();
    for (int p = 0; p < passes; ++p) {
        constexpr int kPipe = (kMode & 16) ? 2 : 1;
        for (int u0 = 0; u0 < kSlots; u0 += kPipe) {
            fvec4 xv[kPipe];
            float sr[kPipe];
#pragma unroll
            for (int j = 0; j < kPipe; ++j) {
                xv[j] = *(const fvec4*)(tile + ((u0 + j) * kBlock + tid) * 4);
                sr[j] = (kMode & 8) ? sh_s[(u0 + j) & 63] : 1.0000003f;
            }
#pragma unroll
            for (int j = 0; j < kPipe; ++j) {
                fvec4 nw;
                double part = 0.0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float nv = (xv[j][k] * iv[k]) * sr[j];
                    nw[k] = nv;
                    if (kMode & 1) part += (double)__builtin_fabsf(nv - xv[j][k]);
                    if (kMode & 2) { cmn[k] = vmin_raw(cmn[k], nv); cmx[k] = vmax_raw(cmx[k], nv); }
                }
                if (kMode & 4) *(fvec4*)(tile + ((u0 + j) * kBlock + tid) * 4) = nw;
                else if (nw[0] == 12345.0f) acc += 1.0;
                acc += part;
            }
        }
        __syncthreads();
    }
    const long long t1 = wall_clock64();
    if (tid == 0) { out[blockIdx.x * 2] = t1 - t0; }
    float r = (float)acc;
    for (int k = 0; k < 4; ++k) r += cmn[k] + cmx[k];
    dst[(size_t)blockIdx.x * kBlock + tid] = r + tile[tid];
}

template <int kMode>
static void run(const char* what, int grid, const float* src, float* dst, long long* out) {
    const int passes = 2000;
    const size_t smem = sizeof(float) * (kTile + 64) + 20 * 1024;       // + 20 KB: the tables of the real kernel (three workgroups per CU fit, four do not)
    hipFuncSetAttribute((const void*)pass_kernel<kMode>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(pass_kernel<kMode>, dim3(grid), dim3(kBlock), smem, 0, src, dst, passes, out);
    hipDeviceSynchronize();
    std::vector<long long> h(2 * grid);
    hipMemcpy(h.data(), out, sizeof(long long) * 2 * grid, hipMemcpyDeviceToHost);
    double us = 0;
    for (int b = 0; b < grid; ++b) us += (double)h[2 * b] / 100.0 / passes;
    printf("%-64s grid %4d: %6.3f us per pass\n", what, grid, us / grid);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *src, *dst;
    long long* out;
    hipMalloc(&src, sizeof(float) * 64 * kTile);
    hipMalloc(&dst, sizeof(float) * 1024 * kBlock);
    hipMalloc(&out, sizeof(long long) * 2 * 1024);
    std::vector<float> h(64 * kTile);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * (float)((i * 2654435761u) % 2000) - 1.0f;
    hipMemcpy(src, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice);
    for (int grid : {256, 512, 768}) {
        run<0>("read, two multiplies per element", grid, src, dst, out);
        run<4>("... + write back", grid, src, dst, out);
        run<4 | 8>("... + per-row factor from LDS", grid, src, dst, out);
        run<4 | 8 | 2>("... + column min / max", grid, src, dst, out);
        run<4 | 8 | 2 | 1>("... + float64 |dW| chain  (the kernel's phase-3 pass)", grid, src, dst, out);
        run<4 | 8 | 2 | 1 | 16>("... two slots' reads in flight", grid, src, dst, out);
        run<8 | 2>("row-statistics pass (read, multiplies, min / max)", grid, src, dst, out);
    }
    return 0;
}
