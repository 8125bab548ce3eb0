// Micro-benchmark: one MFMA wave per SIMD fed from LDS the way the weight-gradient kernel does it
// (PER independent 32x32 accumulators, operands fetched as ds_read_b32 or ds_read_b128, ping-pong registers).
// hipcc --offload-arch=gfx950 -O3 -w mfma_lds.hip -o mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int PER, int MODE>   // MODE 0: no LDS; 1: b32 per operand per step; 2: b128 per operand per 4 steps
__global__ __launch_bounds__(256) void k(float* out, int nsteps, int reps, const int* offs) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1e-3f * i;
    __syncthreads();
    f16v acc[PER];
    for (int i = 0; i < PER; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    int aoff[PER], boff[PER];
    for (int j = 0; j < PER; ++j) { aoff[j] = __builtin_amdgcn_readfirstlane(offs[j]); boff[j] = __builtin_amdgcn_readfirstlane(offs[8 + j]); }
    const int lane = threadIdx.x & 63;
    for (int rep = 0; rep < reps; ++rep) {
        if (MODE == 0) {
            float a = lane + rep, b = 2.f;
            for (int st = 0; st < nsteps; ++st)
#pragma unroll
                for (int j = 0; j < PER; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        } else if (MODE == 1) {
            const float* xb = lds + lane + (rep & 7);
            const float* gb = lds + 8192 + lane;
            float a0[PER], b0[PER], a1[PER], b1[PER];
            auto ld = [&](int st, float (&a)[PER], float (&b)[PER]) {
                const float* xr = xb + st * 96;
                const float* gr = gb + st * 96;
#pragma unroll
                for (int j = 0; j < PER; ++j) { a[j] = xr[aoff[j]]; b[j] = gr[boff[j]]; }
            };
            ld(0, a0, b0);
            for (int st = 0; st < nsteps; st += 2) {
                ld(st + 1, a1, b1);
#pragma unroll
                for (int j = 0; j < PER; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[j], 0, 0, 0);
                if (st + 2 < nsteps) ld(st + 2, a0, b0);
#pragma unroll
                for (int j = 0; j < PER; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[j], 0, 0, 0);
            }
        } else {
            const f4v* xb = reinterpret_cast<const f4v*>(lds) + lane + (rep & 7);
            const f4v* gb = reinterpret_cast<const f4v*>(lds + 8192) + lane;
            f4v a0[PER], b0[PER], a1[PER], b1[PER];
            auto ld = [&](int st, f4v (&a)[PER], f4v (&b)[PER]) {
                const f4v* xr = xb + st * 24;
                const f4v* gr = gb + st * 24;
#pragma unroll
                for (int j = 0; j < PER; ++j) { a[j] = xr[aoff[j]]; b[j] = gr[boff[j]]; }
            };
            ld(0, a0, b0);
            for (int st = 0; st < nsteps / 4; st += 2) {
                ld(st + 1, a1, b1);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < PER; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j][q], b0[j][q], acc[j], 0, 0, 0);
                if (st + 2 < nsteps / 4) ld(st + 2, a0, b0);
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int j = 0; j < PER; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j][q], b1[j][q], acc[j], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < PER; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int PER, int MODE>
static void run(const char* name, float* out, const int* offs, int threads) {
    const int nsteps = 64, reps = 400, wgs = 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<PER, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto launch = [&] { hipLaunchKernelGGL((k<PER, MODE>), dim3(wgs), dim3(threads), 100 * 1024, 0, out, nsteps, reps, offs); };
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * (threads / 64) * reps * nsteps * PER * 4096.0;
    printf("%-40s thr%-4d %8.3f ms  %7.1f TFLOP/s\n", name, threads, ms, flops / ms * 1e-9);
}

int main() {
    float* out; (void)hipMalloc(&out, 4096 * 512 * 4);
    int h[16]; for (int j = 0; j < 8; ++j) { h[j] = (j % 3) * 34 + (j / 3); h[8 + j] = (j % 3) * 8; }
    int* offs; (void)hipMalloc(&offs, 64); (void)hipMemcpy(offs, h, 64, hipMemcpyHostToDevice);
    for (int thr : {256}) {
        run<7, 0>("PER7 no-LDS", out, offs, thr);
        run<7, 1>("PER7 b32 (14 reads / 7 MFMA)", out, offs, thr);
        run<7, 2>("PER7 b128 (14 reads / 28 MFMA)", out, offs, thr);
        run<4, 0>("PER4 no-LDS", out, offs, thr);
        run<4, 1>("PER4 b32", out, offs, thr);
        run<4, 2>("PER4 b128", out, offs, thr);
    }
    return 0;
}
