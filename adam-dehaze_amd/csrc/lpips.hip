// LPIPS (AlexNet) specific kernels, gfx950: the fused scaling + space-to-depth input stage that turns
// AlexNet's 11x11 stride-4 convolution into a 3x3 stride-1 one for the MFMA kernel, and the per-tap
// "unit-normalise, squared difference, 1x1 lin, spatial mean" reduction with its gradient.
// Restates lpips.LPIPS(net='alex') as used by /root/reference training/loss.py:86-108 (lpips >= 0.1.4).
#include "common.h"

__global__ __launch_bounds__(256) void lpips_s2d_kernel(const float* __restrict__ img, int H, int W, float a0, float a1,
                                                        float a2, float b0, float b1, float b2, int OHp, int OWp,
                                                        float* __restrict__ out) {
    const int n = blockIdx.y;
    const int64_t total = (int64_t)OHp * OWp * 16;   // one thread per (cell, in-cell pixel): writes 3 channels
    const float* im = img + (size_t)n * 3 * H * W;
    float* o = out + (size_t)n * OHp * OWp * 48;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int sub = (int)(t & 15);
        const int64_t cell = t >> 4;
        const int r = (int)(cell / OWp), q = (int)(cell - (int64_t)r * OWp);
        const int by = sub >> 2, bx = sub & 3;
        const int iy = 4 * r + by - 2, ix = 4 * q + bx - 2;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const size_t p = (size_t)iy * W + ix;
            v0 = im[p] * a0 + b0;
            v1 = im[(size_t)H * W + p] * a1 + b1;
            v2 = im[2 * (size_t)H * W + p] * a2 + b2;
        }
        float* dst = o + cell * 48 + sub * 3;
        dst[0] = v0;
        dst[1] = v1;
        dst[2] = v2;
    }
}

extern "C" int adh_lpips_s2d(void* stream, const float* img, int N, int H, int W, const float* a3, const float* b3, int OHp,
                             int OWp, float* out) {
    if (!img || !a3 || !b3 || !out || N < 1 || H < 11 || W < 11 || OHp < 3 || OWp < 3) return ADH_E_ARG;
    const int64_t total = (int64_t)OHp * OWp * 16;
    hipLaunchKernelGGL(lpips_s2d_kernel, dim3(adh_min_i(adh_ceil_div(total, 256), 4096), N), dim3(256), 0, (hipStream_t)stream,
                       img, H, W, a3[0], a3[1], a3[2], b3[0], b3[1], b3[2], OHp, OWp, out);
    return adh_check_launch();
}

__global__ __launch_bounds__(256) void lpips_s2d_bwd_kernel(const float* __restrict__ g, int H, int W, float a0, float a1,
                                                            float a2, int OHp, int OWp, float* __restrict__ gimg) {
    const int n = blockIdx.y;
    const int64_t HW = (int64_t)H * W;
    const float* gn = g + (size_t)n * OHp * OWp * 48;
    float* gi = gimg + (size_t)n * 3 * HW;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < HW; p += (int64_t)gridDim.x * blockDim.x) {
        const int iy = (int)(p / W), ix = (int)(p - (int64_t)iy * W);
        const int r = (iy + 2) >> 2, by = (iy + 2) & 3, q = (ix + 2) >> 2, bx = (ix + 2) & 3;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (r < OHp && q < OWp) {
            const float* src = gn + ((size_t)r * OWp + q) * 48 + (by * 4 + bx) * 3;
            v0 = src[0] * a0;
            v1 = src[1] * a1;
            v2 = src[2] * a2;
        }
        gi[p] = v0;
        gi[HW + p] = v1;
        gi[2 * HW + p] = v2;
    }
}

extern "C" int adh_lpips_s2d_bwd(void* stream, const float* g, int N, int H, int W, const float* a3, int OHp, int OWp,
                                 float* g_img) {
    if (!g || !a3 || !g_img || N < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(lpips_s2d_bwd_kernel, dim3(adh_min_i(adh_ceil_div((int64_t)H * W, 256), 4096), N), dim3(256), 0,
                       (hipStream_t)stream, g, H, W, a3[0], a3[1], a3[2], OHp, OWp, g_img);
    return adh_check_launch();
}

// ---------------------------------------------------------------------------------------------
// per-tap distance: 8 lanes per pixel, shuffles for the channel reductions
// ---------------------------------------------------------------------------------------------
#define LP_EPS 1e-10f
#define LP_PPB 1024

extern "C" int adh_lpips_layer_num_blocks(int HW) { return adh_ceil_div(HW, LP_PPB); }

__device__ __forceinline__ float sub8_sum(float v) {
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void lpips_layer_kernel(const float* __restrict__ fa, const float* __restrict__ fb,
                                                          const float* __restrict__ w, int HW, int C,
                                                          float* __restrict__ partial, int nblk) {
    __shared__ float red[4];
    const int n = blockIdx.y, blk = blockIdx.x;
    const int sub = threadIdx.x & 7;
    const int CQ = C / 4;
    const float* an = fa + (size_t)n * HW * C;
    const float* bn = fb + (size_t)n * HW * C;
    const int p1 = adh_min_i((blk + 1) * LP_PPB, HW);
    float acc = 0.f;
    for (int p = blk * LP_PPB + (threadIdx.x >> 3); p < p1; p += 32) {
        float sa = 0.f, sb = 0.f;
        for (int q = sub; q < CQ; q += 8) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(an + (size_t)p * C + q * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(bn + (size_t)p * C + q * 4);
            sa += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
            sb += (b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]);
        }
        sa = sub8_sum(sa);
        sb = sub8_sum(sb);
        const float ia = 1.f / (sqrtf(sa) + LP_EPS), ib = 1.f / (sqrtf(sb) + LP_EPS);
        float d = 0.f;
        for (int q = sub; q < CQ; q += 8) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(an + (size_t)p * C + q * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(bn + (size_t)p * C + q * 4);
            const f32x4 ww = *reinterpret_cast<const f32x4*>(w + q * 4);
            const f32x4 df = a * ia - b * ib;
            const f32x4 t = ww * df * df;
            d += (t[0] + t[1]) + (t[2] + t[3]);
        }
        acc += d;   // every lane of the 8-lane group holds a partial; summed by the block reduction below
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)n * nblk + blk] = (red[0] + red[1]) + (red[2] + red[3]);
}

extern "C" int adh_lpips_layer(void* stream, const float* fa, const float* fb, const float* w, int N, int HW, int C,
                               float* partial, int nblk) {
    if (!fa || !fb || !w || !partial || N < 1 || HW < 1 || C < 4 || (C & 3)) return ADH_E_ARG;
    if (nblk != adh_lpips_layer_num_blocks(HW)) return ADH_E_ARG;
    hipLaunchKernelGGL(lpips_layer_kernel, dim3(nblk, N), dim3(256), 0, (hipStream_t)stream, fa, fb, w, HW, C, partial, nblk);
    return adh_check_launch();
}

__global__ __launch_bounds__(256) void lpips_layer_bwd_kernel(const float* __restrict__ fa, const float* __restrict__ fb,
                                                              const float* __restrict__ w, const float* __restrict__ g_val,
                                                              int HW, int C, float* __restrict__ g_fa) {
    const int n = blockIdx.y;
    const int sub = threadIdx.x & 7;
    const int CQ = C / 4;
    const float* an = fa + (size_t)n * HW * C;
    const float* bn = fb + (size_t)n * HW * C;
    float* gn = g_fa + (size_t)n * HW * C;
    const float gk = g_val[n] / (float)HW;
    for (int p = blockIdx.x * 32 + (threadIdx.x >> 3); p < HW; p += gridDim.x * 32) {
        float sa = 0.f, sb = 0.f;
        for (int q = sub; q < CQ; q += 8) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(an + (size_t)p * C + q * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(bn + (size_t)p * C + q * 4);
            sa += (a[0] * a[0] + a[1] * a[1]) + (a[2] * a[2] + a[3] * a[3]);
            sb += (b[0] * b[0] + b[1] * b[1]) + (b[2] * b[2] + b[3] * b[3]);
        }
        sa = sub8_sum(sa);
        sb = sub8_sum(sb);
        const float ra = sqrtf(sa);
        const float s = ra + LP_EPS;
        const float ia = 1.f / s, ib = 1.f / (sqrtf(sb) + LP_EPS);
        // delta_c = 2 w_c (na_c - nb_c) * gk ;  g_a_k = delta_k / s - a_k / (ra s^2) * sum_c delta_c a_c
        float dot = 0.f;
        for (int q = sub; q < CQ; q += 8) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(an + (size_t)p * C + q * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(bn + (size_t)p * C + q * 4);
            const f32x4 ww = *reinterpret_cast<const f32x4*>(w + q * 4);
            const f32x4 dl = ww * (a * ia - b * ib) * (2.f * gk);
            const f32x4 t = dl * a;
            dot += (t[0] + t[1]) + (t[2] + t[3]);
        }
        dot = sub8_sum(dot);
        const float k2 = ra > 0.f ? dot / (ra * s * s) : 0.f;
        for (int q = sub; q < CQ; q += 8) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(an + (size_t)p * C + q * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(bn + (size_t)p * C + q * 4);
            const f32x4 ww = *reinterpret_cast<const f32x4*>(w + q * 4);
            const f32x4 dl = ww * (a * ia - b * ib) * (2.f * gk);
            *reinterpret_cast<f32x4*>(gn + (size_t)p * C + q * 4) = dl * ia - a * k2;
        }
    }
}

extern "C" int adh_lpips_layer_bwd(void* stream, const float* fa, const float* fb, const float* w, const float* g_val,
                                   int N, int HW, int C, float* g_fa) {
    if (!fa || !fb || !w || !g_val || !g_fa || N < 1 || HW < 1 || C < 4 || (C & 3)) return ADH_E_ARG;
    hipLaunchKernelGGL(lpips_layer_bwd_kernel, dim3(adh_min_i(adh_ceil_div(HW, 32), 2048), N), dim3(256), 0,
                       (hipStream_t)stream, fa, fb, w, g_val, HW, C, g_fa);
    return adh_check_launch();
}

__global__ void rows_sum_kernel(const float* __restrict__ partial, int N, int nblk, float scale, float* __restrict__ out,
                                int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)partial[(size_t)n * nblk + b];
    const float v = (float)(s * (double)scale);
    out[n] = accumulate ? out[n] + v : v;
}

extern "C" int adh_rows_sum(void* stream, const float* partial, int N, int nblk, float scale, float* out, int accumulate) {
    if (!partial || !out || N < 1 || nblk < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(rows_sum_kernel, dim3(adh_ceil_div(N, 64)), dim3(64), 0, (hipStream_t)stream, partial, N, nblk, scale,
                       out, accumulate);
    return adh_check_launch();
}
