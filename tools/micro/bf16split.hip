// Micro-benchmark (SURVEY 7 hard-part 1): the contraction phase of one conv_wino43_kernel chunk
// (wave = 9 frequencies x 3 output-channel tiles x 32 tiles x 16 input channels) on
//   (a) v_mfma_f32_32x32x2_f32   -- the product path: 216 MFMAs per chunk and wave, A from LDS (one ds_read_b128 per 4 MFMAs),
//                                   B from L2 (one global_load_dwordx4 per 4 MFMAs);
//   (b) v_mfma_f32_32x32x16_bf16 -- both operands split into three bf16 planes (x = hi + mid + lo, 8 + 8 + 8 significant bits) and
//                                   the six significant cross terms hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid accumulated in
//                                   fp32: 162 MFMAs per chunk and wave, three ds_read_b128 of A per frequency, three
//                                   global_load_dwordx4 of B per (frequency, channel tile).
// Reports cycles per chunk (s_memtime, median over workgroups), wall TFLOP/s-equivalent, LDS bytes of V per chunk, and the
// error of both against an fp64 reference for K = 96 / 192 / 384 (random operands, |x| <= 1).
//   hipcc --offload-arch=gfx950 -O3 -w tools/micro/bf16split.hip -o tools/micro/bf16split && tools/micro/bf16split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __bf16 to_bf16(float x) { return (__bf16)x; }   // v_cvt_pk_bf16_f32: round to nearest even

// ------------------------------------------------------------------------------------------------ timing kernels
// LDS image: fp32 V[9][k quad][32 tiles][4] per wave (18 KB) or three bf16 planes of the same (27 KB); 4 waves per workgroup,
// one per SIMD, 27 accumulator tiles pinned to AGPRs (16) / VGPRs (11) through the asm constraint as in the product kernels
// (left alone hipcc shuttles accumulators beyond 256 registers between the classes around every MFMA).
template <int IDX>
__device__ __forceinline__ void mfma32(f16v& c, float a, float b) {
    if constexpr (IDX < 16) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int IDX>
__device__ __forceinline__ void mfmabf(f16v& c, const bf8v& a, const bf8v& b) {
    if constexpr (IDX < 16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int MODE, int BL2, int F>   // frequency F of 9 (compile-time accumulator indices)
__device__ __forceinline__ void one_freq(f16v (&acc)[27], const float* Aw, const float* Bc, int lane, int l31, int h, const f4v (&breg)[3]) {
    if constexpr (MODE == 0) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const f4v a = *reinterpret_cast<const f4v*>(Aw + ((F * 4 + 2 * g + h) * 32 + l31) * 4);   // [f][k quad][row][4]: conflict-free
            f4v b[3];
#pragma unroll
            for (int n = 0; n < 3; ++n) b[n] = BL2 ? *reinterpret_cast<const f4v*>(Bc + (((F * 2 + g) * 3 + n) * 64 + lane) * 4) : breg[n];
            // lane half h of MFMA j consumes k = 8 g + 4 h + j: one b128 read feeds 4 MFMAs
            mfma32<F * 3 + 0>(acc[F * 3 + 0], a[0], b[0][0]); mfma32<F * 3 + 0>(acc[F * 3 + 0], a[1], b[0][1]);
            mfma32<F * 3 + 0>(acc[F * 3 + 0], a[2], b[0][2]); mfma32<F * 3 + 0>(acc[F * 3 + 0], a[3], b[0][3]);
            mfma32<F * 3 + 1>(acc[F * 3 + 1], a[0], b[1][0]); mfma32<F * 3 + 1>(acc[F * 3 + 1], a[1], b[1][1]);
            mfma32<F * 3 + 1>(acc[F * 3 + 1], a[2], b[1][2]); mfma32<F * 3 + 1>(acc[F * 3 + 1], a[3], b[1][3]);
            mfma32<F * 3 + 2>(acc[F * 3 + 2], a[0], b[2][0]); mfma32<F * 3 + 2>(acc[F * 3 + 2], a[1], b[2][1]);
            mfma32<F * 3 + 2>(acc[F * 3 + 2], a[2], b[2][2]); mfma32<F * 3 + 2>(acc[F * 3 + 2], a[3], b[2][3]);
        }
    } else {
        const u4v* A16 = reinterpret_cast<const u4v*>(Aw);
        const u4v* B16 = reinterpret_cast<const u4v*>(Bc);
        bf8v a[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) a[p] = __builtin_bit_cast(bf8v, A16[((p * 9 + F) * 2 + h) * 32 + l31]);   // [plane][f][k half][row]
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            bf8v b[3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
                b[p] = BL2 ? __builtin_bit_cast(bf8v, B16[((p * 9 + F) * 3 + n) * 64 + lane]) : __builtin_bit_cast(bf8v, breg[p]);
            if (n == 0) {
                mfmabf<F * 3 + 0>(acc[F * 3 + 0], a[2], b[0]); mfmabf<F * 3 + 0>(acc[F * 3 + 0], a[0], b[2]); mfmabf<F * 3 + 0>(acc[F * 3 + 0], a[1], b[1]);
                mfmabf<F * 3 + 0>(acc[F * 3 + 0], a[1], b[0]); mfmabf<F * 3 + 0>(acc[F * 3 + 0], a[0], b[1]); mfmabf<F * 3 + 0>(acc[F * 3 + 0], a[0], b[0]);
            } else if (n == 1) {
                mfmabf<F * 3 + 1>(acc[F * 3 + 1], a[2], b[0]); mfmabf<F * 3 + 1>(acc[F * 3 + 1], a[0], b[2]); mfmabf<F * 3 + 1>(acc[F * 3 + 1], a[1], b[1]);
                mfmabf<F * 3 + 1>(acc[F * 3 + 1], a[1], b[0]); mfmabf<F * 3 + 1>(acc[F * 3 + 1], a[0], b[1]); mfmabf<F * 3 + 1>(acc[F * 3 + 1], a[0], b[0]);
            } else {
                mfmabf<F * 3 + 2>(acc[F * 3 + 2], a[2], b[0]); mfmabf<F * 3 + 2>(acc[F * 3 + 2], a[0], b[2]); mfmabf<F * 3 + 2>(acc[F * 3 + 2], a[1], b[1]);
                mfmabf<F * 3 + 2>(acc[F * 3 + 2], a[1], b[0]); mfmabf<F * 3 + 2>(acc[F * 3 + 2], a[0], b[1]); mfmabf<F * 3 + 2>(acc[F * 3 + 2], a[0], b[0]);
            }
        }
    }
}
template <int MODE, int BL2>   // MODE 0 = fp32 MFMA, 1 = bf16 x 3 split; BL2 1 = B operands from L2 every group, 0 = from registers
__global__ __launch_bounds__(256, 1) void contract(float* out, const float* __restrict__ Bg, unsigned long long* cyc, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
    for (int i = tid; i < 4 * 9 * 32 * 16 * (MODE ? 3 : 2) / 2; i += 256) lds[i] = 1e-3f * (float)(i % 977);
    __syncthreads();
    f16v acc[27];
    for (int t = 0; t < 27; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float* Aw = lds + wave * (9 * 32 * 16) * (MODE ? 3 : 2) / 2;
    f4v breg[3];
    for (int n = 0; n < 3; ++n) breg[n] = *reinterpret_cast<const f4v*>(Bg + (n * 64 + lane) * 4);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int c = 0; c < chunks; ++c) {
        const float* Bc = Bg + (size_t)(c & 7) * 9 * 3 * 16 * 32 * 4;      // L2-resident weight slab, rotated
        one_freq<MODE, BL2, 0>(acc, Aw, Bc, lane, l31, h, breg); one_freq<MODE, BL2, 1>(acc, Aw, Bc, lane, l31, h, breg);
        one_freq<MODE, BL2, 2>(acc, Aw, Bc, lane, l31, h, breg); one_freq<MODE, BL2, 3>(acc, Aw, Bc, lane, l31, h, breg);
        one_freq<MODE, BL2, 4>(acc, Aw, Bc, lane, l31, h, breg); one_freq<MODE, BL2, 5>(acc, Aw, Bc, lane, l31, h, breg);
        one_freq<MODE, BL2, 6>(acc, Aw, Bc, lane, l31, h, breg); one_freq<MODE, BL2, 7>(acc, Aw, Bc, lane, l31, h, breg);
        one_freq<MODE, BL2, 8>(acc, Aw, Bc, lane, l31, h, breg);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0.f;
    for (int t = 0; t < 27; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + tid] = s;
}

// ------------------------------------------------------------------------------------------------ accuracy kernel
// one wave: C[32][32] = A[32][K] * B[K][32] by both methods
__global__ void accuracy(const float* A, const float* B, int K, float* C32, float* Cbf) {
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    f16v c32, cb;
    for (int r = 0; r < 16; ++r) c32[r] = cb[r] = 0.f;
    for (int k = 0; k < K; k += 2) c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + h], B[(k + h) * 32 + l31], c32, 0, 0, 0);
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf8v a[3], b[3];
        for (int i = 0; i < 8; ++i) {
            const float x = A[l31 * K + k0 + 8 * h + i], y = B[(k0 + 8 * h + i) * 32 + l31];
            const __bf16 xh = to_bf16(x), xm = to_bf16(x - (float)xh), xl = to_bf16(x - (float)xh - (float)xm);
            const __bf16 yh = to_bf16(y), ym = to_bf16(y - (float)yh), yl = to_bf16(y - (float)yh - (float)ym);
            a[0][i] = xh; a[1][i] = xm; a[2][i] = xl;
            b[0][i] = yh; b[1][i] = ym; b[2][i] = yl;
        }
        cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], cb, 0, 0, 0);
        cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], cb, 0, 0, 0);
        cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], cb, 0, 0, 0);
        cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], cb, 0, 0, 0);
        cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], cb, 0, 0, 0);
        cb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], cb, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        C32[row * 32 + l31] = c32[r];
        Cbf[row * 32 + l31] = cb[r];
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE, int BL2>
static int run_timing(const char* name, float* out, const float* B, unsigned long long* cyc) {
    const int wgs = 256, chunks = 400;
    const int ldsb = 4 * 9 * 32 * 16 * (MODE ? 6 : 4);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&contract<MODE, BL2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((contract<MODE, BL2>), dim3(wgs), dim3(256), ldsb, 0, out, B, cyc, chunks); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((contract<MODE, BL2>), dim3(wgs), dim3(256), ldsb, 0, out, B, cyc, chunks);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> hc(wgs); CK(hipMemcpy(hc.data(), cyc, wgs * 8, hipMemcpyDeviceToHost));
    std::sort(hc.begin(), hc.end());
    const double per_chunk = (double)hc[wgs / 2] / chunks;
    const double flop = 2.0 * 9 * 3 * 32 * 32 * 16 * 4.0 * wgs * chunks;    // fp32-equivalent FLOP of the contraction
    printf("%-44s %9.0f cycles / chunk (median WG)  %7.3f ms  %7.1f TFLOP/s fp32-equivalent   V in LDS: %5.1f KB / chunk\n", name, per_chunk, ms,
           flop / ms * 1e-9, ldsb / 1024.0);
    return 0;
}

int main() {
    float *out, *B; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&cyc, 256 * 8)); CK(hipMalloc(&B, 8 * 9 * 3 * 16 * 32 * 4 * 4 * 2));
    { std::vector<float> hb(8 * 9 * 3 * 16 * 32 * 4 * 2); for (size_t i = 0; i < hb.size(); ++i) hb[i] = 1e-3f * (float)(i % 811); CK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); }
    printf("== contraction of one conv_wino43_kernel chunk (9 frequencies x 3 n-tiles x 32 tiles x 16 channels per wave), 256 workgroups x 4 waves\n");
    if (run_timing<0, 0>("fp32   32x32x2_f32   216 MFMAs, B in registers", out, B, cyc)) return 1;
    if (run_timing<1, 0>("bf16x3 32x32x16_bf16 162 MFMAs, B in registers", out, B, cyc)) return 1;
    if (run_timing<0, 1>("fp32   32x32x2_f32   216 MFMAs, B from L2", out, B, cyc)) return 1;
    if (run_timing<1, 1>("bf16x3 32x32x16_bf16 162 MFMAs, B from L2", out, B, cyc)) return 1;
    printf("== accuracy vs fp64, C[32][32] = A[32][K] B[K][32], uniform(-1,1) operands: max |err| / max |C|, rms err / max |C|\n");
    for (int K : {96, 192, 384}) {
        std::vector<float> hA(32 * K), hB(K * 32), c32(1024), cbf(1024);
        srand(K);
        for (auto& v : hA) v = 2.f * rand() / RAND_MAX - 1.f;
        for (auto& v : hB) v = 2.f * rand() / RAND_MAX - 1.f;
        float *dA, *dB, *d32, *dbf;
        CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&d32, 4096)); CK(hipMalloc(&dbf, 4096));
        CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(accuracy, dim3(1), dim3(64), 0, 0, dA, dB, K, d32, dbf); CK(hipDeviceSynchronize());
        CK(hipMemcpy(c32.data(), d32, 4096, hipMemcpyDeviceToHost)); CK(hipMemcpy(cbf.data(), dbf, 4096, hipMemcpyDeviceToHost));
        double cmax = 0, e32 = 0, ebf = 0, r32 = 0, rbf = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double ref = 0; for (int k = 0; k < K; ++k) ref += (double)hA[i * K + k] * (double)hB[k * 32 + j];
            cmax = std::max(cmax, std::fabs(ref));
            const double a = c32[i * 32 + j] - ref, b = cbf[i * 32 + j] - ref;
            e32 = std::max(e32, std::fabs(a)); ebf = std::max(ebf, std::fabs(b)); r32 += a * a; rbf += b * b;
        }
        printf("K = %3d   fp32 MFMA: max %.2e rms %.2e    bf16 x 3 (6 terms): max %.2e rms %.2e    ratio (rms) %.2f\n", K, e32 / cmax,
               std::sqrt(r32 / 1024) / cmax, ebf / cmax, std::sqrt(rbf / 1024) / cmax, std::sqrt(rbf / r32));
        (void)hipFree(dA); (void)hipFree(dB); (void)hipFree(d32); (void)hipFree(dbf);
    }
    return 0;
}
