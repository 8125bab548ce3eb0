// Micro-benchmark: sustained fp32 MFMA rate on gfx950 as a function of waves per SIMD and
// interleaved LDS reads.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int NACC, int LDSREADS>
__global__ __launch_bounds__(512) void k32(float* out, int iters, float a0, float b0) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = a0 * i;
    __syncthreads();
    f16v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    const f4v* l4 = reinterpret_cast<const f4v*>(lds);
    for (int it = 0; it < iters; ++it) {
        if (LDSREADS) {
            #pragma unroll
            for (int r = 0; r < LDSREADS; ++r) {
                f4v v = l4[(threadIdx.x + r * 64 + it) & 2047];
                a += v[0]; b += v[3];
            }
        }
        #pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters, float a0, float b0) {
    f4v acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static void run(const char* name, F launch, double flops) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms * 1e-9);
}

int main() {
    float* out; hipMalloc(&out, 4096 * 512 * 4);
    const int iters = 20000;
    for (int threads : {256, 512, 1024}) {
        for (int wgs : {256, 512}) {
            if (threads == 1024) continue;
            char nm[128];
            double f32 = (double)wgs * (threads / 64) * iters * 4 * 4096.0;
            snprintf(nm, 128, "32x32x2 acc4 lds0 thr%d wgs%d", threads, wgs);
            run(nm, [&] { hipLaunchKernelGGL((k32<4, 0>), dim3(wgs), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, f32);
            snprintf(nm, 128, "32x32x2 acc4 lds1(b128) thr%d wgs%d", threads, wgs);
            run(nm, [&] { hipLaunchKernelGGL((k32<4, 1>), dim3(wgs), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, f32);
            snprintf(nm, 128, "32x32x2 acc4 lds2(b128) thr%d wgs%d", threads, wgs);
            run(nm, [&] { hipLaunchKernelGGL((k32<4, 2>), dim3(wgs), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, f32);
            snprintf(nm, 128, "32x32x2 acc4 lds4(b128) thr%d wgs%d", threads, wgs);
            run(nm, [&] { hipLaunchKernelGGL((k32<4, 4>), dim3(wgs), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, f32);
            double f16 = (double)wgs * (threads / 64) * iters * 4 * 2048.0;
            snprintf(nm, 128, "16x16x4 acc4 thr%d wgs%d", threads, wgs);
            run(nm, [&] { hipLaunchKernelGGL((k16<4>), dim3(wgs), dim3(threads), 0, 0, out, iters, 1.f, 2.f); }, f16);
        }
    }
    return 0;
}
