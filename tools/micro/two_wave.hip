// Micro-benchmark: does an 8-wave workgroup with TWO waves per SIMD, one contracting while the other runs the
// input transform, beat the product structure of conv_wino43_kernel (4 waves, one per SIMD, transform and contraction in
// lock-step phases)?  One 8-channel slice of a 32-tile region per iteration:
//   MODE 0  product shape: 4 waves x 27 accumulator tiles (9 frequencies x 3 n-tiles, N = 96).  Per slice and wave: transform
//           pass 1 (9 ds_read_b128 + 36 v_pk_fma + 9 ds_write_b128), barrier, pass 2 (same), barrier, 108 MFMAs (one
//           ds_read_b128 of A per frequency, three dwordx4 of B from L2 fetched one frequency ahead), barrier.
//   MODE 1  8 waves x 9 accumulator tiles (9 frequencies x 1 n-tile, N = 64), wave w and w + 4 share a SIMD.  Phase p: the waves with
//           ((p + w / 4) & 1) == 0 contract (36 MFMAs), the others run ONE transform pass (9 + 36 + 9 as above); barrier.  Two phases
//           per slice: every wave contracts every slice, pass 1 and pass 2 of the next slice run under the two contractions.
//   MODE 3  as MODE 1 but only the transform passes are ordered (an LDS counter: pass 2 of a slice follows pass 1); the two waves of a
//           SIMD contract concurrently, offset by one pass; one barrier per slice.  Needs V two slices ahead (three V buffers).
//   MODE 2  the same 8 waves in lock step (all transform, barrier, all contract): separates "two waves per SIMD" from "phase offset".
// Reports cycles per slice and per 32 output channels of it (median workgroup, s_memtime) and the wall-clock MFMA rate.
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/two_wave.hip -o tools/micro/two_wave && tools/micro/two_wave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

template <int IDX, int NA = 16>
__device__ __forceinline__ void mfma32(f16v& c, float a, float b) {
    if constexpr (IDX < NA) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pkfma(f2v& d, f2v a, f2v b) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b)); }

// one transform pass of one wave: 9 b128 reads, 36 v_pk_fma, 9 b128 writes, in three column-like groups
__device__ __forceinline__ void transform_pass(float* T, int lane, f2v k0, f2v k1) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        f4v x[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) x[i] = *reinterpret_cast<const f4v*>(T + ((g * 3 + i) * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                f2v lo = {x[i][0], x[i][1]}, hi = {x[i][2], x[i][3]};
                const f4v& y = x[(i + 1) % 3];
                pkfma(lo, k0, f2v{y[0], y[1]}); pkfma(hi, k1, f2v{y[2], y[3]});
                x[i] = f4v{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<f4v*>(T + (((g * 3 + i) * 64 + lane) * 4)) = x[i];
    }
}

template <int IDX, int NA>
__device__ __forceinline__ void four(f16v& c, const f4v& a, const f4v& b) {
    mfma32<IDX, NA>(c, a[0], b[0]); mfma32<IDX, NA>(c, a[1], b[1]); mfma32<IDX, NA>(c, a[2], b[2]); mfma32<IDX, NA>(c, a[3], b[3]);
}
template <int NT, int F>
__device__ __forceinline__ void one_freq(f16v (&acc)[9 * NT], const float* Aw, const float* Bc, int lane, int l31, f4v (&b)[2][NT]) {
    const f4v a = *reinterpret_cast<const f4v*>(Aw + (F * 32 + l31) * 4 + (lane >> 5) * 9 * 32 * 4);
    if constexpr (F < 8) {
#pragma unroll
        for (int n = 0; n < NT; ++n) b[(F + 1) & 1][n] = *reinterpret_cast<const f4v*>(Bc + (((F + 1) * NT + n) * 64 + lane) * 4);
    }
    constexpr int NA = NT == 3 ? 16 : 7;    // accumulator tiles pinned to AGPRs (two waves per SIMD: 128 AGPRs at most)
    four<F * NT + 0, NA>(acc[F * NT + 0], a, b[F & 1][0]);
    if constexpr (NT > 1) four<F * NT + 1, NA>(acc[F * NT + 1], a, b[F & 1][1]);
    if constexpr (NT > 2) four<F * NT + 2, NA>(acc[F * NT + 2], a, b[F & 1][2]);
}
template <int NT>
__device__ __forceinline__ void contract_slice(f16v (&acc)[9 * NT], const float* Aw, const float* Bc, const float* Bnext, int lane, int l31,
                                               f4v (&b0)[NT]) {
    f4v b[2][NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) b[0][n] = b0[n];          // fetched before the previous barrier
    one_freq<NT, 0>(acc, Aw, Bc, lane, l31, b); one_freq<NT, 1>(acc, Aw, Bc, lane, l31, b); one_freq<NT, 2>(acc, Aw, Bc, lane, l31, b);
    one_freq<NT, 3>(acc, Aw, Bc, lane, l31, b); one_freq<NT, 4>(acc, Aw, Bc, lane, l31, b); one_freq<NT, 5>(acc, Aw, Bc, lane, l31, b);
    one_freq<NT, 6>(acc, Aw, Bc, lane, l31, b); one_freq<NT, 7>(acc, Aw, Bc, lane, l31, b); one_freq<NT, 8>(acc, Aw, Bc, lane, l31, b);
#pragma unroll
    for (int n = 0; n < NT; ++n) b0[n] = *reinterpret_cast<const f4v*>(Bnext + (n * 64 + lane) * 4);
}

__device__ __forceinline__ void flag_signal(unsigned* flag, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this wave's LDS writes are done
    if (lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void flag_wait(const unsigned* flag, unsigned target) {
    for (;;) {
        const unsigned v = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile unsigned*>(flag));
        if ((int)(v - target) >= 0) break;
        __builtin_amdgcn_s_sleep(1);
    }
}

template <int MODE>
__global__ __launch_bounds__(MODE == 0 ? 256 : 512, 1) void phases(float* out, const float* __restrict__ Bg, unsigned long long* cyc, int slices, int tx) {
    constexpr int NT = MODE == 0 ? 3 : 1, NW = MODE == 0 ? 4 : 8;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31;
    for (int i = tid; i < 4 * 9 * 32 * 8 + NW * 9 * 64 * 4; i += NW * 64) lds[i] = 1e-3f * (float)(i % 977);
    __syncthreads();
    f16v acc[9 * NT];
    for (int t = 0; t < 9 * NT; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float* Aw = lds + (wave & 3) * (9 * 32 * 8);                   // V[frequency group of the SIMD][k half][9][32][4]
    float* T = lds + 4 * 9 * 32 * 8 + wave * (9 * 64 * 4);
    const f2v k0 = {0.75f, -1.25f}, k1 = {0.5625f, 1.5625f};
    const int role = (MODE == 1 || MODE == 3) ? wave >> 2 : 0;
    unsigned* const flag = reinterpret_cast<unsigned*>(lds + 4 * 9 * 32 * 8 + NW * 9 * 64 * 4);
    if (tid == 0) *flag = 0;
    __syncthreads();
    f4v b0[NT];
    for (int n = 0; n < NT; ++n) b0[n] = *reinterpret_cast<const f4v*>(Bg + (size_t)wave * 9 * 3 * 64 * 4 + (n * 64 + lane) * 4);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < slices; ++s) {
        const float* Bc = Bg + (size_t)((s & 7) * 8 + wave) * 9 * 3 * 64 * 4;      // L2-resident weight slab, rotated
        const float* Bn = Bg + (size_t)(((s + 1) & 7) * 8 + wave) * 9 * 3 * 64 * 4;
        if constexpr (MODE == 0) {
            { if (tx) transform_pass(T, lane, k0, k1); }; __syncthreads();
            { if (tx) transform_pass(T, lane, k0, k1); }; __syncthreads();
            contract_slice<NT>(acc, Aw, Bc, Bn, lane, l31, b0); __syncthreads();
        } else if constexpr (MODE == 2) {
            { if (tx) transform_pass(T, lane, k0, k1); }; __syncthreads();
            contract_slice<NT>(acc, Aw, Bc, Bn, lane, l31, b0); __syncthreads();
        } else if constexpr (MODE == 3) {
            // A: contract, then pass 2 once the four B waves have finished pass 1 of this iteration; B: pass 1, signal, contract
            if (role == 1) { { if (tx) transform_pass(T, lane, k0, k1); }; flag_signal(flag, lane); }
            contract_slice<NT>(acc, Aw, Bc, Bn, lane, l31, b0);
            if (role == 0) { flag_wait(flag, 4u * (unsigned)(s + 1)); { if (tx) transform_pass(T, lane, k0, k1); }; }
            __syncthreads();
        } else {
            if (role == 0) contract_slice<NT>(acc, Aw, Bc, Bn, lane, l31, b0); else { if (tx) transform_pass(T, lane, k0, k1); };
            __syncthreads();
            if (role == 1) contract_slice<NT>(acc, Aw, Bc, Bn, lane, l31, b0); else { if (tx) transform_pass(T, lane, k0, k1); };
            __syncthreads();
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    float sum = 0.f;
    for (int t = 0; t < 9 * NT; ++t) for (int r = 0; r < 16; ++r) sum += acc[t][r];
    out[blockIdx.x * 512 + tid] = sum + T[lane];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
static int run(const char* name, float* out, const float* B, unsigned long long* cyc, int tx = 1) {
    constexpr int NW = MODE == 0 ? 4 : 8, NT = MODE == 0 ? 3 : 1;
    const int wgs = 256, slices = 800;
    const int ldsb = (4 * 9 * 32 * 8 + NW * 9 * 64 * 4) * 4 + 64;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&phases<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((phases<MODE>), dim3(wgs), dim3(NW * 64), ldsb, 0, out, B, cyc, slices, tx); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((phases<MODE>), dim3(wgs), dim3(NW * 64), ldsb, 0, out, B, cyc, slices, tx);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hc(wgs); CK(hipMemcpy(hc.data(), cyc, wgs * 8, hipMemcpyDeviceToHost));
    std::sort(hc.begin(), hc.end());
    const double per_slice = (double)hc[wgs / 2] / slices;                          // s_memtime ticks (100 MHz) -> report wall too
    const int ntiles = MODE == 0 ? 3 : 2;                                          // 32-channel tiles of output per workgroup
    const double flop = 2.0 * 36 * ntiles * 32 * 32 * 8 * (double)wgs * slices;
    printf("%-58s %8.3f ms  %7.1f TFLOP/s   %7.3f us / slice   %7.3f us per 32 output channels\n", name, ms, flop / ms * 1e-9,
           ms * 1e3 / slices, ms * 1e3 / slices / ntiles);
    (void)per_slice; (void)NT;
    return 0;
}

int main() {
    float *out, *B; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8)); CK(hipMalloc(&B, 64 * 9 * 3 * 64 * 4 * 4));
    { std::vector<float> hb(64 * 9 * 3 * 64 * 4); for (size_t i = 0; i < hb.size(); ++i) hb[i] = 1e-3f * (float)(i % 811); CK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); }
    printf("== one 8-channel slice of a 32-tile F(4x4,3x3) region per iteration, 256 workgroups, 800 slices\n");
    if (run<0>("4 waves x 27 tiles, lock-step phases (product structure)", out, B, cyc)) return 1;
    if (run<2>("8 waves x  9 tiles, lock-step phases", out, B, cyc)) return 1;
    if (run<1>("8 waves x  9 tiles, contraction || transform pass", out, B, cyc)) return 1;
    if (run<3>("8 waves x  9 tiles, both contract, LDS-flag pass order", out, B, cyc)) return 1;
    printf("-- the same without the transform passes (contraction, barriers and flags only)\n");
    if (run<0>("4 waves x 27 tiles, lock-step phases (product structure)", out, B, cyc, 0)) return 1;
    if (run<2>("8 waves x  9 tiles, lock-step phases", out, B, cyc, 0)) return 1;
    if (run<1>("8 waves x  9 tiles, contraction || transform pass", out, B, cyc, 0)) return 1;
    if (run<3>("8 waves x  9 tiles, both contract, LDS-flag pass order", out, B, cyc, 0)) return 1;
    return 0;
}
