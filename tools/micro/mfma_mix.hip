// Micro-benchmark: cost of co-issued instructions per fp32 MFMA (v_mfma_f32_32x32x2_f32), one or two waves
// per SIMD, exact instruction streams via inline asm.  Each loop iteration = 4 MFMAs on 4 accumulators with
// N extra instructions after each MFMA; LDS results are never consumed (s_waitcnt once per iteration, for the
// loads issued one iteration earlier is not expressible, so lgkmcnt is drained at iteration start).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

enum { PLAIN, VALU, DS32, DS64, DS128, DSW128, GL128, DMA128, SALU };
typedef __attribute__((address_space(3))) void* lds_void_ptr;

template <int KIND, int N, int EVERY>   // N instructions after every EVERY-th MFMA
__global__ __launch_bounds__(512) void k(float* out, const float* in, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1e-3f * i;
    __syncthreads();
    f16v acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const int lane = threadIdx.x & 63;
    float a = lane, b = 2.f;
    int v0 = lane, v1 = 1;
    unsigned addr = (threadIdx.x & 255) * 16;   // LDS byte address, 16 B per lane
    float r32; f2v r64; f4v r128; f4v w128 = {1.f, 2.f, 3.f, 4.f};
    const float* gp = in + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in) + blockIdx.x * 65536, 0, 0x7fffffff, 0x00020000);
    int sacc = 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            if (i % EVERY == 0) {
#pragma unroll
                for (int r = 0; r < N; ++r) {
                    if (KIND == VALU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v0) : "v"(v1));
                    if (KIND == DS32) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r32) : "v"(addr), "n"(r * 4096));
                    if (KIND == DS64) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r64) : "v"(addr), "n"(r * 4096));
                    if (KIND == DS128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r128) : "v"(addr), "n"(r * 4096));
                    if (KIND == DSW128) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(addr), "v"(w128), "n"(16384 + r * 4096) : "memory");
                    if (KIND == GL128) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r128) : "v"(gp), "n"(r * 256));
                    if (KIND == DMA128) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_ptr)(lds + 4096 + (threadIdx.x >> 6) * 256), 16, lane * 16, (it & 63) * 1024, 0, 0);
                    if (KIND == SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc) : : "scc");
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
    float s = v0 + sacc;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (KIND == DS32) s += r32;
    if (KIND == DS64) s += r64[0];
    if (KIND == DS128 || KIND == GL128) s += r128[0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int N, int EVERY>
static void run(const char* name, float* out, const float* in, int threads) {
    const int iters = 8000, wgs = 256;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND, N, EVERY>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto launch = [&] { hipLaunchKernelGGL((k<KIND, N, EVERY>), dim3(wgs), dim3(threads), 100 * 1024, 0, out, in, iters); };
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * (threads / 64) * iters * 4 * 4096.0;
    printf("%-34s thr%-4d %8.3f ms  %7.1f TFLOP/s\n", name, threads, ms, flops / ms * 1e-9);
}

int main() {
    float *out, *in; (void)hipMalloc(&out, 4096 * 512 * 4); (void)hipMalloc(&in, 64 << 20); (void)hipMemset(in, 0, 64 << 20);
    for (int thr : {256, 512}) {
        run<PLAIN, 0, 1>("plain", out, in, thr);
        run<VALU, 1, 1>("v_add x1 /mfma", out, in, thr);
        run<VALU, 2, 1>("v_add x2 /mfma", out, in, thr);
        run<VALU, 4, 1>("v_add x4 /mfma", out, in, thr);
        run<VALU, 8, 1>("v_add x8 /mfma", out, in, thr);
        run<VALU, 12, 1>("v_add x12 /mfma", out, in, thr);
        run<DS32, 1, 1>("ds_read_b32 x1 /mfma", out, in, thr);
        run<DS32, 2, 1>("ds_read_b32 x2 /mfma", out, in, thr);
        run<DS32, 4, 1>("ds_read_b32 x4 /mfma", out, in, thr);
        run<DS64, 1, 1>("ds_read_b64 x1 /mfma", out, in, thr);
        run<DS64, 2, 1>("ds_read_b64 x2 /mfma", out, in, thr);
        run<DS128, 1, 2>("ds_read_b128 x0.5 /mfma", out, in, thr);
        run<DS128, 1, 1>("ds_read_b128 x1 /mfma", out, in, thr);
        run<DS128, 2, 1>("ds_read_b128 x2 /mfma", out, in, thr);
        run<DSW128, 1, 4>("ds_write_b128 x0.25 /mfma", out, in, thr);
        run<DSW128, 1, 2>("ds_write_b128 x0.5 /mfma", out, in, thr);
        run<DSW128, 1, 1>("ds_write_b128 x1 /mfma", out, in, thr);
        run<GL128, 1, 4>("global_load_x4 x0.25 /mfma", out, in, thr);
        run<GL128, 1, 2>("global_load_x4 x0.5 /mfma", out, in, thr);
        run<GL128, 1, 1>("global_load_x4 x1 /mfma", out, in, thr);
        run<DMA128, 1, 4>("lds-dma 1KiB x0.25 /mfma", out, in, thr);
        run<DMA128, 1, 2>("lds-dma 1KiB x0.5 /mfma", out, in, thr);
        run<DMA128, 1, 1>("lds-dma 1KiB x1 /mfma", out, in, thr);
        run<SALU, 4, 1>("s_add x4 /mfma", out, in, thr);
        run<SALU, 12, 1>("s_add x12 /mfma", out, in, thr);
    }
    return 0;
}
