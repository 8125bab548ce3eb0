// LDS-DMA semantics probe (gfx950): destinations above 64 KB and partially masked waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_void_ptr;
__global__ __launch_bounds__(64) void k(const float* in, float* out, int ldsoff_floats, int masklo, int maskhi) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 40960; i += 64) lds[i] = -1.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0x7fffffff, 0x00020000);
    const int lane = threadIdx.x;
    float* dst = lds + ldsoff_floats;
    if (lane >= masklo && lane < maskhi) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)dst, 16, lane * 16, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 40960; i += 64) out[i] = lds[i];
}
int main() {
    float *in, *out; (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, 40960 * 4);
    std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = i;
    (void)hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<float> o(40960);
    struct { int off, lo, hi; } cases[] = {{0, 0, 64}, {20000, 0, 64}, {39000, 0, 64}, {1024, 8, 16}, {30000, 0, 8}};
    for (auto c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, in, out, c.off, c.lo, c.hi);
        (void)hipMemcpy(o.data(), out, 40960 * 4, hipMemcpyDeviceToHost);
        int first = -1, last = -1, n = 0;
        for (int i = 0; i < 40960; ++i) if (o[i] != -1.f) { if (first < 0) first = i; last = i; ++n; }
        printf("dst float off %d lanes [%d,%d): written %d floats, first idx %d (val %.0f) last idx %d (val %.0f)\n", c.off, c.lo, c.hi, n,
               first, first >= 0 ? o[first] : 0.f, last, last >= 0 ? o[last] : 0.f);
    }
    return 0;
}
