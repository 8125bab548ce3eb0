// What does issuing an LDS-DMA piece (buffer_load_dwordx4 ... lds, 64 lanes x 16 B) cost the issuing wave?  gfx950.
// 256 workgroups x 4 waves (one per SIMD, like the MFMA kernels); every wave issues NP pieces back to back, s_memtime
// around the issue loop (not around the completion), then waits.  Patterns per piece:
//   seg  = contiguous bytes per group of lanes (1024: one 1 KB run; 128: 8 runs of 128 B; 64: 16 runs of 64 B)
//   gap  = byte distance between the runs of a piece
// footprint: every wave walks its own region of `span` bytes (small: stays in L2; large: streams from HBM).
// Also the plain-load alternative: global_load_dwordx4 into registers (same addresses).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NP, bool DMA, bool GLOBAL = false>
__global__ __launch_bounds__(256, 1) void k(const float* in, unsigned long long* ticks, float* sink, int seg, int gap,
                                            long long span, int iters, int nw = 4) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wid = (long long)blockIdx.x * 4 + wave;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0x7fffffff, 0x00020000);
    const int lanes_per_seg = seg / 16;
    const int voff = (lane / lanes_per_seg) * gap + (lane % lanes_per_seg) * 16;   // byte offset of this lane inside a piece
    const int piece_bytes = (64 / lanes_per_seg) * gap;                             // footprint of one piece
    unsigned long long total = 0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    long long pos = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned base = (unsigned)((wid * span + pos) & 0x7fffffff);
        pos = (pos + (long long)NP * piece_bytes) % span;
        __builtin_amdgcn_s_barrier();
        const unsigned long long t0 = __builtin_readcyclecounter();
        if (wave >= nw) continue;      // round 3: only `nw` of the CU's four waves issue (is the cost per wave or per CU?)
        if constexpr (DMA && GLOBAL) {      // round 3: the global_load_lds_dwordx4 form (per-lane 64-bit address) instead of buffer_load ... lds
            typedef __attribute__((address_space(1))) void* gptr;
            const char* gb = reinterpret_cast<const char*>(in) + base + voff;
#pragma unroll
            for (int u = 0; u < NP; ++u)
                __builtin_amdgcn_global_load_lds((gptr)(gb + (size_t)u * piece_bytes), (lds_void_ptr)(lds + (wave * NP + u) * 256), 16, 0, 0);
        } else if constexpr (DMA) {
#pragma unroll
            for (int u = 0; u < NP; ++u)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_ptr)(lds + (wave * NP + u) * 256), 16, voff,
                                                         base + u * piece_bytes, 0, 0);
        } else {
            f32x4 v[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u)
                v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, base + u * piece_bytes, 0));
            const unsigned long long t1x = __builtin_readcyclecounter();
            total += t1x - t0;
#pragma unroll
            for (int u = 0; u < NP; ++u) acc += v[u];
        }
        if constexpr (DMA) {
            const unsigned long long t1 = __builtin_readcyclecounter();
            total += t1 - t0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lane == 0) ticks[wid] = total;
    if (!DMA) sink[threadIdx.x + blockIdx.x * 256] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int NP, bool DMA, bool GLOBAL = false>
static void run(const char* name, const float* in, unsigned long long* ticks, float* sink, int seg, int gap, long long span, int nw = 4) {
    const int iters = 64;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NP, DMA, GLOBAL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NP, DMA, GLOBAL>), dim3(256), dim3(256), 96 * 1024, 0, in, ticks, sink, seg, gap, span, 4, nw);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NP, DMA, GLOBAL>), dim3(256), dim3(256), 96 * 1024, 0, in, ticks, sink, seg, gap, span, iters, nw);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(1024);
    (void)hipMemcpy(h.data(), ticks, 1024 * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 1024; ++i) if ((i & 3) < nw) s += (double)h[i];
    const double per_piece = s / (256.0 * nw) / iters / NP;
    const double gb = 256.0 * nw * iters * NP * 1024.0 / 1e9;
    printf("%-44s issue %7.1f ticks per piece per wave   kernel %.3f ms = %.2f TB/s\n", name, per_piece, ms, gb / ms);
}

int main() {
    const long long bytes = 2ll << 30;   // the buffer descriptor spans 2 GiB
    float* in;
    unsigned long long* ticks;
    float* sink;
    (void)hipMalloc(&in, bytes);
    (void)hipMemset(in, 0, bytes);
    (void)hipMalloc(&ticks, 1024 * 8);
    (void)hipMalloc(&sink, 256 * 256 * 4);
    const long long small = 64 * 1024, large = 2 * 1024 * 1024;   // per-wave footprint: 64 MB / 2 GB over the chip
    printf("ticks = s_memtime (about 2 per ns); 256 workgroups x 4 waves, 10 pieces per burst\n");
    run<10, true>("DMA  1 x 1024 B             L2-resident", in, ticks, sink, 1024, 1024, small);
    run<10, true>("DMA  8 x 128 B, gap 768     L2-resident", in, ticks, sink, 128, 768, small);
    run<10, true>("DMA 16 x 64 B, gap 384      L2-resident", in, ticks, sink, 64, 384, small);
    run<10, true>("DMA  1 x 1024 B             streaming", in, ticks, sink, 1024, 1024, large);
    run<10, true>("DMA  8 x 128 B, gap 768     streaming", in, ticks, sink, 128, 768, large);
    run<10, true>("DMA 16 x 64 B, gap 384      streaming", in, ticks, sink, 64, 384, large);
    run<10, false>("load 1 x 1024 B             L2-resident", in, ticks, sink, 1024, 1024, small);
    run<10, false>("load 8 x 128 B, gap 768     L2-resident", in, ticks, sink, 128, 768, small);
    run<10, false>("load 16 x 64 B, gap 384     L2-resident", in, ticks, sink, 64, 384, small);
    run<10, false>("load 8 x 128 B, gap 768     streaming", in, ticks, sink, 128, 768, large);
    run<20, true>("DMA  8 x 128 B, gap 768, 20 per burst, streaming", in, ticks, sink, 128, 768, large);
    printf("-- round 3: how many of the CU's four waves issue (10 pieces per burst, 8 x 128 B)\n");
    for (int nw = 1; nw <= 4; ++nw) {
        char nm[96];
        snprintf(nm, sizeof nm, "DMA  8 x 128 B  L2-resident, %d issuing wave(s)", nw);
        run<10, true>(nm, in, ticks, sink, 128, 768, small, nw);
        snprintf(nm, sizeof nm, "load 8 x 128 B  L2-resident, %d issuing wave(s)", nw);
        run<10, false>(nm, in, ticks, sink, 128, 768, small, nw);
    }
    run<10, true, true>("global_load_lds_dwordx4 8 x 128 B  L2-resident, 4 waves", in, ticks, sink, 128, 768, small, 4);
    run<10, true, true>("global_load_lds_dwordx4 8 x 128 B  L2-resident, 1 wave", in, ticks, sink, 128, 768, small, 1);
    run<10, true, true>("global_load_lds_dwordx4 1 x 1024 B streaming, 4 waves", in, ticks, sink, 1024, 1024, large, 4);
    for (int nw = 1; nw <= 4; nw += 3) {
        char nm[96];
        snprintf(nm, sizeof nm, "DMA  8 x 128 B  streaming, %d issuing wave(s)", nw);
        run<10, true>(nm, in, ticks, sink, 128, 768, large, nw);
        snprintf(nm, sizeof nm, "DMA  8 x 128 B  L2, 1 piece per burst, %d wave(s)", nw);
        run<1, true>(nm, in, ticks, sink, 128, 768, small, nw);
        snprintf(nm, sizeof nm, "DMA  8 x 128 B  L2, 3 pieces per burst, %d wave(s)", nw);
        run<3, true>(nm, in, ticks, sink, 128, 768, small, nw);
    }
    return 0;
}
