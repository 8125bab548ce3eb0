// LDS-DMA semantics probe (gfx950): what lands in LDS for a lane whose buffer offset is OUT OF RANGE (bit 31 of the vector offset
// set against num_records = 0x7fffffff, the marker the Winograd epilogues already use for dropped stores)?  If the hardware writes
// zeros, the out-of-image cells of a raw halo need no table of flags and no zeroing pass after landing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_void_ptr;
__global__ __launch_bounds__(64) void k(const float* in, float* out, int soff, unsigned pattern) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = -1.f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0x7fffffff, 0x00020000);
    const int lane = threadIdx.x;
    unsigned vo = (unsigned)lane * 16u;
    if ((pattern >> (lane & 31)) & 1u) vo |= 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)(lds + 256), 16, (int)vo, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    float *in, *out;
    (void)hipMalloc(&in, 1 << 20);
    (void)hipMalloc(&out, 4096);
    std::vector<float> h(1 << 18);
    for (int i = 0; i < (1 << 18); ++i) h[i] = 1.f + i;
    (void)hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<float> o(1024);
    struct { int soff; unsigned pat; } cases[] = {{0, 0u}, {0, 0xaaaaaaaau}, {4096, 0xaaaaaaaau}, {4096, 0xffff0000u}, {65536, 0x0000000fu}, {0, 0xffffffffu}};
    for (auto c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, in, out, c.soff, c.pat);
        (void)hipMemcpy(o.data(), out, 4096, hipMemcpyDeviceToHost);
        int good = 0, zero = 0, untouched = 0, other = 0, outside = 0;
        for (int l = 0; l < 64; ++l) {
            const bool oob = (c.pat >> (l & 31)) & 1u;
            for (int e = 0; e < 4; ++e) {
                const float v = o[256 + 4 * l + e], want = 1.f + c.soff / 4 + 4 * l + e;
                if (!oob) { if (v == want) ++good; else ++other; }
                else if (v == 0.f) ++zero;
                else if (v == -1.f) ++untouched;
                else ++other;
            }
        }
        for (int i = 0; i < 1024; ++i) if ((i < 256 || i >= 512) && o[i] != -1.f) ++outside;
        printf("soffset %6d pattern %08x: in-range floats ok %d | out-of-range lanes: zero %d untouched %d | wrong %d | written outside the piece %d\n",
               c.soff, c.pat, good, zero, untouched, other, outside);
    }
    return 0;
}
