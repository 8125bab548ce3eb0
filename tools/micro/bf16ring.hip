// Micro-benchmark (round 4): can the WEIGHT STREAM of a bf16 x 3 conv_wino43 contraction keep up?
// r03's bf16split.hip showed the contraction of one chunk 2.5x faster with the operands in registers, but only 1.6x with
// compiler-scheduled weight loads from L2.  The bf16 x 3 form reads 81 KB of U per wave and chunk (three 2-byte planes,
// 16 channels x 96 output channels x 9 frequencies) in 162 MFMAs x 32 cycles: 62 B/clk per CU, the L2's aggregate peak.
// This is the product kernel's contraction with the product kernel's weight protocol:
//   group = (frequency, 32-channel tile): 3 global_load_dwordx4 (the planes), 6 MFMAs (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi);
//   a ring of DEPTH + 1 groups of B registers, loads issued DEPTH groups ahead, hand-counted s_waitcnt vmcnt;
//   A = three ds_read_b128 per frequency (the planes of V), one frequency ahead;
//   U laid out [cog][chunk][36 f][3 tiles][3 planes][64 lanes][16 B]: a wave's 27 groups of a chunk are 81 KB contiguous.
// Cases: Cin = Cout = 96 (U = 1.9 MB: L2 resident), 192 (7.6 MB), 384 (30 MB: beyond the 4 MB L2, the 32 CUs of an XCD
// sweep it roughly in step).  Reports cycles per chunk (median workgroup), wall ms, fp32-equivalent TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/bf16ring.hip -o tools/micro/bf16ring && tools/micro/bf16ring
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

template <int IDX>
__device__ __forceinline__ void mfmabf(f16v& c, const u4v& a, const u4v& b) {
    if constexpr (IDX < 16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void load_b(u4v (&b)[3], unsigned voff, const char* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[0]) : "v"(voff), "s"(sbase) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(b[1]) : "v"(voff), "s"(sbase) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(b[2]) : "v"(voff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_b(u4v (&b)[3]) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) : "n"(N) : "memory");
}
template <int T>
__device__ __forceinline__ void six(f16v& c, const u4v (&a)[3], const u4v (&b)[3]) {
    mfmabf<T>(c, a[2], b[0]); mfmabf<T>(c, a[0], b[2]); mfmabf<T>(c, a[1], b[1]);
    mfmabf<T>(c, a[1], b[0]); mfmabf<T>(c, a[0], b[1]); mfmabf<T>(c, a[0], b[0]);
}
// groups G .. 26 of one chunk; ring slot of group G = G % (DEPTH + 1)
template <int DEPTH, int NT, int G>
__device__ __forceinline__ void groups(f16v (&acc)[9 * NT], u4v (&av)[2][3], u4v (&bv)[DEPTH + 1][3], const u4v* Aw, unsigned voff,
                                       const char* bchunk, const char* bnext) {
    if constexpr (G < 9 * NT) {
        constexpr int GA = G + DEPTH;                        // the group whose weights are requested now
        if constexpr (GA < 9 * NT) load_b(bv[GA % (DEPTH + 1)], voff, bchunk + GA * 3072);
        else load_b(bv[GA % (DEPTH + 1)], voff, bnext + (GA - 9 * NT) * 3072);
        if constexpr (G % NT == 0 && G / NT + 1 < 9) {         // next frequency's V planes
#pragma unroll
            for (int p = 0; p < 3; ++p) av[(G / NT + 1) & 1][p] = Aw[((G / NT + 1) * 3 + p) * 64];
        }
        wait_b<3 * DEPTH>(bv[G % (DEPTH + 1)]);
        six<(NT == 3 ? G : G + 8)>(acc[G], av[(G / NT) & 1], bv[G % (DEPTH + 1)]);
        groups<DEPTH, NT, G + 1>(acc, av, bv, Aw, voff, bchunk, bnext);
    }
}
template <int DEPTH, int NT>
__global__ __launch_bounds__(256, 1) void contract_ring(float* out, const char* __restrict__ U, unsigned long long* cyc, int nchunks,
                                                       int ncog, int reps) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4 * 9 * 3 * 64 * 4; i += 256) lds[i] = 1e-3f * (float)(i % 977);
    __syncthreads();
    f16v acc[9 * NT];
    for (int t = 0; t < 9 * NT; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const u4v* Aw = reinterpret_cast<const u4v*>(lds) + wave * (9 * 3 * 64) + lane;
    const int cog = (blockIdx.x >> 3) % ncog;
    const size_t chunk_bytes = (size_t)36 * 3 * NT * 1024;
    const char* Uw = U + (size_t)cog * nchunks * chunk_bytes + (size_t)wave * 27 * NT * 1024;
    const unsigned voff = lane * 16;
    u4v av[2][3], bv[DEPTH + 1][3];
#pragma unroll
    for (int g = 0; g < DEPTH; ++g) load_b(bv[g], voff, Uw + g * 3072);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
#pragma unroll 1
        for (int c = 0; c < nchunks; ++c) {
            const char* bchunk = Uw + (size_t)c * chunk_bytes;
            const char* bnext = Uw + (size_t)(c + 1 < nchunks ? c + 1 : 0) * chunk_bytes;
#pragma unroll
            for (int p = 0; p < 3; ++p) av[0][p] = Aw[p * 64];
            groups<DEPTH, NT, 0>(acc, av, bv, Aw, voff, bchunk, bnext);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int g = 0; g < DEPTH; ++g) wait_b<0>(bv[g]);
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0.f;
    for (int t = 0; t < 9 * NT; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + tid] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int DEPTH, int NT>
static int run(const char* name, int C, float* out, const char* U, unsigned long long* cyc) {
    const int nchunks = C / 16, ncog = C / (32 * NT), wgs = 256 * 4, reps = 24 * 16 / nchunks / 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&contract_ring<DEPTH, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int ldsb = 4 * 9 * 3 * 64 * 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((contract_ring<DEPTH, NT>), dim3(wgs), dim3(256), ldsb, 0, out, U, cyc, nchunks, ncog, reps); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((contract_ring<DEPTH, NT>), dim3(wgs), dim3(256), ldsb, 0, out, U, cyc, nchunks, ncog, reps);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hc(wgs); CK(hipMemcpy(hc.data(), cyc, wgs * 8, hipMemcpyDeviceToHost));
    std::sort(hc.begin(), hc.end());
    const double per_chunk = (double)hc[wgs / 2] / (nchunks * reps);
    const double flop = 2.0 * 9 * NT * 32 * 32 * 16 * 4.0 * wgs * nchunks * reps;
    printf("%-34s C=%3d (U %5.1f MB) depth %d NT %d  %7.0f cycles / chunk (median WG; MFMA floor %d)  %7.3f ms  %6.1f TFLOP/s fp32-eq  U stream %5.1f B/clk/CU\n",
           name, C, (double)ncog * nchunks * 36 * 3 * NT / 1024.0, DEPTH, NT, per_chunk, 54 * NT * 32, ms, flop / ms * 1e-9, 4.0 * 27 * NT * 1024 / per_chunk);
    return 0;
}

int main() {
    float* out; char* U; unsigned long long* cyc;
    const size_t ubytes = (size_t)4 * 24 * 36 * 9 * 1024 + 4096;
    CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&cyc, 1024 * 8)); CK(hipMalloc(&U, ubytes));
    { std::vector<unsigned short> hb(ubytes / 2); for (size_t i = 0; i < hb.size(); ++i) hb[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 25); CK(hipMemcpy(U, hb.data(), ubytes, hipMemcpyHostToDevice)); }
    printf("== bf16 x 3 contraction of conv_wino43 chunks, weights streamed from L2 with a hand-counted ring (4 waves x 27 tiles, 1024 workgroups)\n");
    for (int C : {96, 192, 384}) {
        if (run<2, 3>("27 tiles, 2 groups ahead", C, out, U, cyc)) return 1;
        if (run<2, 2>("18 tiles, 2 groups ahead", C, out, U, cyc)) return 1;
        if (run<3, 2>("18 tiles, 3 groups ahead", C, out, U, cyc)) return 1;
        if (run<5, 2>("18 tiles, 5 groups ahead", C, out, U, cyc)) return 1;
        if (run<6, 2>("18 tiles, 6 groups ahead", C, out, U, cyc)) return 1;
    }
    return 0;
}
