// Boundary layout, branch heads, routing, losses, optimiser and small spatial ops, gfx950.
// All HBM-bound streaming kernels (16 B per lane where the layout allows, block partials reduced
// in a fixed order).  Reference lines are cited per kernel (paths under /root/reference).
#include "common.h"

#define RED_ELEMS_PER_BLOCK (256 * 16)

extern "C" int adh_version(void) { return 100; }

// ------------------------------------------------------------------------------------------------
// layout
// ------------------------------------------------------------------------------------------------
__global__ void image_to_nhwc8_kernel(const float* __restrict__ img, int64_t HW, int64_t total, float* __restrict__ out) {
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = idx / HW, p = idx - n * HW;
        const float* b = img + n * 3 * HW + p;
        f32x4 lo = {b[0], b[HW], b[2 * HW], 0.f};
        f32x4 hi = {0.f, 0.f, 0.f, 0.f};
        f32x4* o = reinterpret_cast<f32x4*>(out + idx * 8);
        o[0] = lo;
        o[1] = hi;
    }
}

extern "C" int adh_image_to_nhwc8(void* stream, const float* img, int N, int H, int W, float* out) {
    if (!img || !out || N < 1 || H < 1 || W < 1) return ADH_E_ARG;
    const int64_t HW = (int64_t)H * W, total = HW * N;
    hipLaunchKernelGGL(image_to_nhwc8_kernel, dim3(adh_min_i(adh_ceil_div(total, 256), 8192)), dim3(256), 0,
                       (hipStream_t)stream, img, HW, total, out);
    return adh_check_launch();
}

__global__ void image_normalize_kernel(const float* __restrict__ img, int64_t HW, int64_t total, float m0, float m1, float m2,
                                       float is0, float is1, float is2, float* __restrict__ out) {
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = idx / HW, p = idx - n * HW;
        const float* b = img + n * 3 * HW + p;
        f32x4 lo = {(b[0] - m0) * is0, (b[HW] - m1) * is1, (b[2 * HW] - m2) * is2, 0.f};
        f32x4 hi = {0.f, 0.f, 0.f, 0.f};
        f32x4* o = reinterpret_cast<f32x4*>(out + idx * 8);
        o[0] = lo;
        o[1] = hi;
    }
}

extern "C" int adh_image_normalize_to_nhwc8(void* stream, const float* img, int N, int H, int W, float m0, float m1, float m2,
                                            float is0, float is1, float is2, float* out) {
    if (!img || !out || N < 1 || H < 1 || W < 1) return ADH_E_ARG;
    const int64_t HW = (int64_t)H * W, total = HW * N;
    hipLaunchKernelGGL(image_normalize_kernel, dim3(adh_min_i(adh_ceil_div(total, 256), 8192)), dim3(256), 0,
                       (hipStream_t)stream, img, HW, total, m0, m1, m2, is0, is1, is2, out);
    return adh_check_launch();
}

__global__ void image_normalize_bwd_kernel(const float* __restrict__ g, int g_cs, int64_t HW, int64_t total, float is0,
                                           float is1, float is2, float* __restrict__ gimg) {
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = idx / HW, p = idx - n * HW;
        const float* gp = g + idx * g_cs;
        float* b = gimg + n * 3 * HW + p;
        b[0] = gp[0] * is0;
        b[HW] = gp[1] * is1;
        b[2 * HW] = gp[2] * is2;
    }
}

extern "C" int adh_image_normalize_bwd(void* stream, const float* g, int g_cs, int N, int H, int W, float is0, float is1,
                                       float is2, float* g_img) {
    if (!g || !g_img || N < 1 || g_cs < 3) return ADH_E_ARG;
    const int64_t HW = (int64_t)H * W, total = HW * N;
    hipLaunchKernelGGL(image_normalize_bwd_kernel, dim3(adh_min_i(adh_ceil_div(total, 256), 8192)), dim3(256), 0,
                       (hipStream_t)stream, g, g_cs, HW, total, is0, is1, is2, g_img);
    return adh_check_launch();
}

// generic transposes through a 32x33 LDS tile: src [n][C][HW] <-> dst [n][HW][cs]
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, int C, int64_t HW, float* __restrict__ dst, int dst_cs) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int64_t p0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r;
        const int64_t p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? src[((int64_t)n * C + c) * HW + p] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t p = p0 + r;
        const int c = c0 + tx;
        if (p < HW && c < C) dst[((int64_t)n * HW + p) * dst_cs + c] = tile[tx][r];
    }
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, int src_cs, int C, int64_t HW, float* __restrict__ dst) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int64_t p0 = (int64_t)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int64_t p = p0 + r;
        const int c = c0 + tx;
        tile[r][tx] = (p < HW && c < C) ? src[((int64_t)n * HW + p) * src_cs + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r;
        const int64_t p = p0 + tx;
        if (c < C && p < HW) dst[((int64_t)n * C + c) * HW + p] = tile[tx][r];
    }
}

extern "C" int adh_nchw_to_nhwc(void* stream, const float* src, int N, int C, int H, int W, float* dst, int dst_cs) {
    if (!src || !dst || N < 1 || C < 1 || dst_cs < C) return ADH_E_ARG;
    const int64_t HW = (int64_t)H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(adh_ceil_div(HW, 32), adh_ceil_div(C, 32), N), dim3(256), 0,
                       (hipStream_t)stream, src, C, HW, dst, dst_cs);
    return adh_check_launch();
}

extern "C" int adh_nhwc_to_nchw(void* stream, const float* src, int src_cs, int N, int C, int H, int W, float* dst) {
    if (!src || !dst || N < 1 || C < 1 || src_cs < C) return ADH_E_ARG;
    const int64_t HW = (int64_t)H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(adh_ceil_div(HW, 32), adh_ceil_div(C, 32), N), dim3(256), 0,
                       (hipStream_t)stream, src, src_cs, C, HW, dst);
    return adh_check_launch();
}

// ------------------------------------------------------------------------------------------------
// branch heads: final blend (low_intensity.py:41-45,116; medium_intensity.py:117;
// high_intensity.py:135-138,214)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }
__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }

__global__ __launch_bounds__(256) void head_blend_kernel(int mode, const float* __restrict__ x, const float* __restrict__ r,
                                                         int r_cs, const float* __restrict__ gd, int gd_cs,
                                                         const float* __restrict__ alpha, int64_t HW, int64_t total,
                                                         float* __restrict__ out) {
    const float a = (mode == 0) ? alpha[0] : 0.f;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = idx / HW, p = idx - n * HW;
        const float* rp = r + idx * r_cs;
        float gate = 0.f;
        if (mode == 2) gate = sigm(gd[idx * gd_cs]);
        if (mode == 4) gate = 1.f - sigm(gd[idx * gd_cs]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t o = (n * 3 + c) * HW + p;
            const float xv = x[o], rv = rp[c];
            float v;
            if (mode == 0) v = (1.f - a) * xv + a * sigm(rv);
            else if (mode == 1) v = clamp01(xv + tanhf(rv));
            else if (mode == 2 || mode == 4) v = clamp01(xv + tanhf(rv) * gate);
            else v = clamp01(xv + (sigm(rv) - 0.5f) * 2.f);
            out[o] = v;
        }
    }
}

extern "C" int adh_head_blend(void* stream, int mode, const float* x_nchw, const float* r, int r_cs, const float* gd,
                              int gd_cs, const float* alpha, int N, int H, int W, float* out_nchw) {
    if (!x_nchw || !r || !out_nchw || mode < 0 || mode > 4 || r_cs < 3) return ADH_E_ARG;
    if ((mode == 2 || mode == 4) && (!gd || gd_cs < 1)) return ADH_E_ARG;
    if (mode == 0 && !alpha) return ADH_E_ARG;
    const int64_t HW = (int64_t)H * W, total = HW * N;
    hipLaunchKernelGGL(head_blend_kernel, dim3(adh_min_i(adh_ceil_div(total, 256), 8192)), dim3(256), 0,
                       (hipStream_t)stream, mode, x_nchw, r, r_cs, gd, gd_cs, alpha, HW, total, out_nchw);
    return adh_check_launch();
}

extern "C" int adh_head_blend_bwd_num_blocks(int N, int H, int W) {
    return adh_min_i(adh_ceil_div((int64_t)N * H * W, 256), 4096);
}

__global__ __launch_bounds__(256) void head_blend_bwd_kernel(int mode, const float* __restrict__ g,
                                                             const float* __restrict__ x, const float* __restrict__ r,
                                                             int r_cs, const float* __restrict__ gd, int gd_cs,
                                                             const float* __restrict__ alpha, int64_t HW, int64_t total,
                                                             float* __restrict__ g_r, float* __restrict__ g_gd,
                                                             float* __restrict__ galpha_partial) {
    __shared__ float red[4];
    const float a = (mode == 0) ? alpha[0] : 0.f;
    float ga = 0.f;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = idx / HW, p = idx - n * HW;
        const float* rp = r + idx * r_cs;
        float s = 0.f, gate = 0.f;
        if (mode == 2 || mode == 4) {
            s = sigm(gd[idx * gd_cs]);
            gate = (mode == 2) ? s : 1.f - s;
        }
        float ggd = 0.f;
        float gr[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int64_t o = (n * 3 + c) * HW + p;
            const float xv = x[o], rv = rp[c], gv = g[o];
            if (mode == 0) {
                const float sr = sigm(rv);
                gr[c] = gv * a * sr * (1.f - sr);
                ga += gv * (sr - xv);
            } else if (mode == 3) {
                const float sr = sigm(rv);
                const float v = xv + (sr - 0.5f) * 2.f;
                const float m = (v >= 0.f && v <= 1.f) ? gv : 0.f;
                gr[c] = m * 2.f * sr * (1.f - sr);
            } else {
                const float t = tanhf(rv);
                const float k = (mode == 1) ? 1.f : gate;
                const float v = xv + t * k;
                const float m = (v >= 0.f && v <= 1.f) ? gv : 0.f;
                gr[c] = m * k * (1.f - t * t);
                ggd += m * t;
            }
        }
        float* grp = g_r + idx * r_cs;
        for (int c = 0; c < r_cs; ++c) grp[c] = c < 3 ? gr[c] : 0.f;
        if (mode == 2 || mode == 4) {
            float* gg = g_gd + idx * gd_cs;
            const float ds = s * (1.f - s);
            gg[0] = (mode == 2) ? ggd * ds : -ggd * ds;
            for (int c = 1; c < gd_cs; ++c) gg[c] = 0.f;
        }
    }
    if (mode == 0) {
        ga = wave_sum(ga);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ga;
        __syncthreads();
        if (threadIdx.x == 0) galpha_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

extern "C" int adh_head_blend_bwd(void* stream, int mode, const float* g_out_nchw, const float* x_nchw, const float* r,
                                  int r_cs, const float* gd, int gd_cs, const float* alpha, int N, int H, int W, float* g_r,
                                  float* g_gd, float* galpha_partial, int nblk) {
    if (!g_out_nchw || !x_nchw || !r || !g_r || mode < 0 || mode > 4 || r_cs < 3) return ADH_E_ARG;
    if ((mode == 2 || mode == 4) && (!gd || !g_gd || gd_cs < 1)) return ADH_E_ARG;
    if (mode == 0 && (!alpha || !galpha_partial)) return ADH_E_ARG;
    if (nblk != adh_head_blend_bwd_num_blocks(N, H, W)) return ADH_E_ARG;
    const int64_t HW = (int64_t)H * W, total = HW * N;
    hipLaunchKernelGGL(head_blend_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, mode, g_out_nchw, x_nchw, r,
                       r_cs, gd, gd_cs, alpha, HW, total, g_r, g_gd, galpha_partial);
    return adh_check_launch();
}

// ------------------------------------------------------------------------------------------------
// routing (models/routing.py:41-61,110-127)
// ------------------------------------------------------------------------------------------------
__global__ void softmax3_kernel(const float* __restrict__ logits, float invT, int N, float* __restrict__ w) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float a = logits[n * 3] * invT, b = logits[n * 3 + 1] * invT, c = logits[n * 3 + 2] * invT;
    const float m = fmaxf(a, fmaxf(b, c));
    const float ea = expf(a - m), eb = expf(b - m), ec = expf(c - m);
    const float inv = 1.f / (ea + eb + ec);
    w[n * 3] = ea * inv;
    w[n * 3 + 1] = eb * inv;
    w[n * 3 + 2] = ec * inv;
}

extern "C" int adh_softmax3(void* stream, const float* logits, float temperature, int N, float* weights) {
    if (!logits || !weights || N < 1 || temperature == 0.f) return ADH_E_ARG;
    hipLaunchKernelGGL(softmax3_kernel, dim3(adh_ceil_div(N, 64)), dim3(64), 0, (hipStream_t)stream, logits,
                       1.0f / temperature, N, weights);
    return adh_check_launch();
}

__global__ void softmax3_bwd_kernel(const float* __restrict__ w, const float* __restrict__ gw_partial, int nblk, float invT,
                                    int N, float* __restrict__ gl) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double g[3] = {0.0, 0.0, 0.0};
    for (int b = 0; b < nblk; ++b)
        for (int i = 0; i < 3; ++i) g[i] += (double)gw_partial[((size_t)n * nblk + b) * 3 + i];
    const float w0 = w[n * 3], w1 = w[n * 3 + 1], w2 = w[n * 3 + 2];
    const float dot = (float)g[0] * w0 + (float)g[1] * w1 + (float)g[2] * w2;
    gl[n * 3] = w0 * ((float)g[0] - dot) * invT;
    gl[n * 3 + 1] = w1 * ((float)g[1] - dot) * invT;
    gl[n * 3 + 2] = w2 * ((float)g[2] - dot) * invT;
}

extern "C" int adh_softmax3_bwd(void* stream, const float* weights, const float* gw_partial, int nblk, float temperature,
                                int N, float* g_logits) {
    if (!weights || !gw_partial || !g_logits || N < 1 || nblk < 1 || temperature == 0.f) return ADH_E_ARG;
    hipLaunchKernelGGL(softmax3_bwd_kernel, dim3(adh_ceil_div(N, 64)), dim3(64), 0, (hipStream_t)stream, weights,
                       gw_partial, nblk, 1.0f / temperature, N, g_logits);
    return adh_check_launch();
}

__global__ __launch_bounds__(256) void soft_blend_kernel(const float* __restrict__ w, const float* __restrict__ o0,
                                                         const float* __restrict__ o1, const float* __restrict__ o2,
                                                         int64_t per4, float* __restrict__ out) {
    const int n = blockIdx.y;
    const float w0 = w[n * 3], w1 = w[n * 3 + 1], w2 = w[n * 3 + 2];
    const f32x4* a = reinterpret_cast<const f32x4*>(o0) + (int64_t)n * per4;
    const f32x4* b = reinterpret_cast<const f32x4*>(o1) + (int64_t)n * per4;
    const f32x4* c = reinterpret_cast<const f32x4*>(o2) + (int64_t)n * per4;
    f32x4* o = reinterpret_cast<f32x4*>(out) + (int64_t)n * per4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per4; i += (int64_t)gridDim.x * blockDim.x) {
        // same association as the reference's in-place accumulation: ((0 + w0*a) + w1*b) + w2*c
        o[i] = (a[i] * w0 + b[i] * w1) + c[i] * w2;
    }
}

extern "C" int adh_soft_blend(void* stream, const float* weights, const float* o0, const float* o1, const float* o2, int N,
                              int64_t per, float* out) {
    if (!weights || !o0 || !o1 || !o2 || !out || N < 1 || per < 4 || (per & 3)) return ADH_E_ARG;
    const int64_t per4 = per / 4;
    hipLaunchKernelGGL(soft_blend_kernel, dim3(adh_min_i(adh_ceil_div(per4, 256), 2048), N), dim3(256), 0,
                       (hipStream_t)stream, weights, o0, o1, o2, per4, out);
    return adh_check_launch();
}

__global__ __launch_bounds__(256) void soft_blend_bwd_kernel(const float* __restrict__ w, const float* __restrict__ g,
                                                             const float* __restrict__ o0, const float* __restrict__ o1,
                                                             const float* __restrict__ o2, int64_t per4,
                                                             float* __restrict__ g0, float* __restrict__ g1,
                                                             float* __restrict__ g2, float* __restrict__ gw_partial) {
    __shared__ float red[3][4];
    const int n = blockIdx.y;
    const float w0 = w[n * 3], w1 = w[n * 3 + 1], w2 = w[n * 3 + 2];
    const int64_t base = (int64_t)n * per4;
    const f32x4* gg = reinterpret_cast<const f32x4*>(g) + base;
    const f32x4* a = reinterpret_cast<const f32x4*>(o0) + base;
    const f32x4* b = reinterpret_cast<const f32x4*>(o1) + base;
    const f32x4* c = reinterpret_cast<const f32x4*>(o2) + base;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 gv = gg[i];
        if (g0) reinterpret_cast<f32x4*>(g0)[base + i] = gv * w0;
        if (g1) reinterpret_cast<f32x4*>(g1)[base + i] = gv * w1;
        if (g2) reinterpret_cast<f32x4*>(g2)[base + i] = gv * w2;
        const f32x4 pa = gv * a[i], pb = gv * b[i], pc = gv * c[i];
        s0 += (pa[0] + pa[1]) + (pa[2] + pa[3]);
        s1 += (pb[0] + pb[1]) + (pb[2] + pb[3]);
        s2 += (pc[0] + pc[1]) + (pc[2] + pc[3]);
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s0;
        red[1][threadIdx.x >> 6] = s1;
        red[2][threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        gw_partial[((size_t)n * gridDim.x + blockIdx.x) * 3 + threadIdx.x] =
            (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

extern "C" int adh_soft_blend_bwd(void* stream, const float* weights, const float* g, const float* o0, const float* o1,
                                  const float* o2, int N, int64_t per, float* g0, float* g1, float* g2, float* gw_partial,
                                  int nblk) {
    if (!weights || !g || !o0 || !o1 || !o2 || !gw_partial || N < 1 || per < 4 || (per & 3) || nblk < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(soft_blend_bwd_kernel, dim3(nblk, N), dim3(256), 0, (hipStream_t)stream, weights, g, o0, o1, o2,
                       per / 4, g0, g1, g2, gw_partial);
    return adh_check_launch();
}

__global__ void argmax3_kernel(const float* __restrict__ logits, int N, int64_t* __restrict__ idx) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float a = logits[n * 3], b = logits[n * 3 + 1], c = logits[n * 3 + 2];
    int i = 0;
    float m = a;
    if (b > m) { m = b; i = 1; }
    if (c > m) { m = c; i = 2; }
    idx[n] = i;
}

extern "C" int adh_argmax3(void* stream, const float* logits, int N, int64_t* idx) {
    if (!logits || !idx || N < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(argmax3_kernel, dim3(adh_ceil_div(N, 64)), dim3(64), 0, (hipStream_t)stream, logits, N, idx);
    return adh_check_launch();
}

// stable compaction of batch indices per class; single block (N is a batch size)
__global__ void route_compact_kernel(const int64_t* __restrict__ idx, int N, int32_t* __restrict__ sel,
                                     int32_t* __restrict__ counts) {
    const int cls = threadIdx.x;
    if (cls >= 3) return;
    int k = 0;
    for (int n = 0; n < N; ++n)
        if (idx[n] == cls) sel[cls * N + k++] = n;
    counts[cls] = k;
}

extern "C" int adh_route_compact(void* stream, const int64_t* idx, int N, int32_t* sel, int32_t* counts) {
    if (!idx || !sel || !counts || N < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(route_compact_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, idx, N, sel, counts);
    return adh_check_launch();
}

__global__ __launch_bounds__(256) void gather_images_kernel(const float* __restrict__ src, const int32_t* __restrict__ sel,
                                                            int64_t per4, float* __restrict__ dst, int scatter) {
    const int j = blockIdx.y;
    const int n = sel[j];
    const f32x4* s = reinterpret_cast<const f32x4*>(src) + (int64_t)(scatter ? j : n) * per4;
    f32x4* d = reinterpret_cast<f32x4*>(dst) + (int64_t)(scatter ? n : j) * per4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per4; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
}

extern "C" int adh_gather_images(void* stream, const float* src, const int32_t* sel, int count, int64_t per, float* dst) {
    if (!src || !sel || !dst || count < 1 || per < 4 || (per & 3)) return ADH_E_ARG;
    hipLaunchKernelGGL(gather_images_kernel, dim3(adh_min_i(adh_ceil_div(per / 4, 256), 1024), count), dim3(256), 0,
                       (hipStream_t)stream, src, sel, per / 4, dst, 0);
    return adh_check_launch();
}

extern "C" int adh_scatter_images(void* stream, const float* src, const int32_t* sel, int count, int64_t per, float* dst) {
    if (!src || !sel || !dst || count < 1 || per < 4 || (per & 3)) return ADH_E_ARG;
    hipLaunchKernelGGL(gather_images_kernel, dim3(adh_min_i(adh_ceil_div(per / 4, 256), 1024), count), dim3(256), 0,
                       (hipStream_t)stream, src, sel, per / 4, dst, 1);
    return adh_check_launch();
}

// ------------------------------------------------------------------------------------------------
// losses (training/loss.py:138 L1Loss, :81 mse_loss, :200 CrossEntropyLoss)
// ------------------------------------------------------------------------------------------------
extern "C" int adh_reduce_num_blocks(int64_t n) { return adh_max_i(1, adh_min_i(adh_ceil_div(n, RED_ELEMS_PER_BLOCK), 2048)); }

template <int SQ>
__global__ __launch_bounds__(256) void diff_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           int64_t n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const int64_t n4 = n / 4;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 dv = a4[i] - b4[i];
        if (SQ) s += (dv[0] * dv[0] + dv[1] * dv[1]) + (dv[2] * dv[2] + dv[3] * dv[3]);
        else s += (fabsf(dv[0]) + fabsf(dv[1])) + (fabsf(dv[2]) + fabsf(dv[3]));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = n4 * 4; i < n; ++i) {
            const float dv = a[i] - b[i];
            s += SQ ? dv * dv : fabsf(dv);
        }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

extern "C" int adh_l1_partial(void* stream, const float* a, const float* b, int64_t n, float* partial) {
    if (!a || !b || !partial || n < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(diff_partial_kernel<0>, dim3(adh_reduce_num_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, n,
                       partial);
    return adh_check_launch();
}

extern "C" int adh_mse_partial(void* stream, const float* a, const float* b, int64_t n, float* partial) {
    if (!a || !b || !partial || n < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(diff_partial_kernel<1>, dim3(adh_reduce_num_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, n,
                       partial);
    return adh_check_launch();
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int nblk, double scale,
                                                           float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += (double)partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * scale);
}

extern "C" int adh_sum_partials(void* stream, const float* partial, int nblk, double scale, float* out) {
    if (!partial || !out || nblk < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, nblk, scale, out);
    return adh_check_launch();
}

template <int SQ>
__global__ __launch_bounds__(256) void diff_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                       float gscale, const float* __restrict__ upstream,
                                                       float* __restrict__ g) {
    const float k = gscale * (upstream ? upstream[0] : 1.f);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float dv = a[i] - b[i];
        // torch: d|x|/dx = sign(x) with sign(0) = 0
        g[i] = SQ ? 2.f * dv * k : (dv > 0.f ? k : (dv < 0.f ? -k : 0.f));
    }
}

extern "C" int adh_l1_bwd(void* stream, const float* a, const float* b, int64_t n, float gscale, const float* upstream,
                          float* g_a) {
    if (!a || !b || !g_a || n < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(diff_bwd_kernel<0>, dim3(adh_min_i(adh_ceil_div(n, 256), 8192)), dim3(256), 0, (hipStream_t)stream,
                       a, b, n, gscale, upstream, g_a);
    return adh_check_launch();
}

extern "C" int adh_mse_bwd(void* stream, const float* a, const float* b, int64_t n, float gscale, const float* upstream,
                           float* g_a) {
    if (!a || !b || !g_a || n < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(diff_bwd_kernel<1>, dim3(adh_min_i(adh_ceil_div(n, 256), 8192)), dim3(256), 0, (hipStream_t)stream,
                       a, b, n, gscale, upstream, g_a);
    return adh_check_launch();
}

// single block: N is a batch size
__global__ __launch_bounds__(256) void cross_entropy3_kernel(const float* __restrict__ logits,
                                                             const int64_t* __restrict__ labels, int N,
                                                             float* __restrict__ loss, float* __restrict__ dlogits) {
    __shared__ double red[256];
    double s = 0.0;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float a = logits[n * 3], b = logits[n * 3 + 1], c = logits[n * 3 + 2];
        const float m = fmaxf(a, fmaxf(b, c));
        const float ea = expf(a - m), eb = expf(b - m), ec = expf(c - m);
        const float se = ea + eb + ec;
        const int y = (int)labels[n];
        const float ly = (y == 0 ? a : (y == 1 ? b : c));
        s += (double)(logf(se) + m - ly);
        if (dlogits) {
            const float inv = 1.f / (se * (float)N);
            dlogits[n * 3] = ea * inv - (y == 0 ? 1.f / N : 0.f);
            dlogits[n * 3 + 1] = eb * inv - (y == 1 ? 1.f / N : 0.f);
            dlogits[n * 3 + 2] = ec * inv - (y == 2 ? 1.f / N : 0.f);
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(red[0] / (double)N);
}

extern "C" int adh_cross_entropy3(void* stream, const float* logits, const int64_t* labels, int N, float* loss,
                                  float* dlogits) {
    if (!logits || !labels || !loss || N < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(cross_entropy3_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, N, loss, dlogits);
    return adh_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Adam (training/train_joint.py:86-90; torch.optim.Adam semantics, L2 weight decay added to grad)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, int step, float lr, float beta1,
                                                   float beta2, float eps, float wd, int repeats) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pv = p[i], mv = m[i], vv = v[i];
        const float gv0 = g[i];
        int st = step;
        for (int r = 0; r < repeats; ++r) {
            ++st;
            const float gv = gv0 + wd * pv;
            mv = beta1 * mv + (1.f - beta1) * gv;
            vv = beta2 * vv + (1.f - beta2) * gv * gv;
            const float bc1 = 1.f - powf(beta1, (float)st);
            const float bc2 = 1.f - powf(beta2, (float)st);
            const float denom = sqrtf(vv) / sqrtf(bc2) + eps;
            pv -= (lr / bc1) * (mv / denom);
        }
        p[i] = pv;
        m[i] = mv;
        v[i] = vv;
    }
}

extern "C" int adh_adam_step(void* stream, float* p, const float* g, float* m, float* v, int64_t n, int step, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int repeats) {
    if (!p || !g || !m || !v || n < 1 || repeats < 1 || step < 0) return ADH_E_ARG;
    hipLaunchKernelGGL(adam_kernel, dim3(adh_min_i(adh_ceil_div(n, 256), 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m,
                       v, n, step, lr, beta1, beta2, eps, weight_decay, repeats);
    return adh_check_launch();
}

// ------------------------------------------------------------------------------------------------
// misc elementwise
// ------------------------------------------------------------------------------------------------
__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] += src[i];
}

extern "C" int adh_add_inplace(void* stream, float* dst, const float* src, int64_t n) {
    if (!dst || !src || n < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(adh_min_i(adh_ceil_div(n, 256), 8192)), dim3(256), 0, (hipStream_t)stream,
                       dst, src, n);
    return adh_check_launch();
}

__global__ __launch_bounds__(256) void axpby_strided_kernel(float* __restrict__ dst, int dst_cs,
                                                            const float* __restrict__ src, int src_cs, int64_t P, int CQ,
                                                            float a, float b) {
    const int64_t total = P * CQ;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = idx / CQ;
        const int c = (int)(idx - p * CQ) * 4;
        f32x4* d = reinterpret_cast<f32x4*>(dst + p * dst_cs + c);
        const f32x4 s = *reinterpret_cast<const f32x4*>(src + p * src_cs + c);
        *d = (a == 0.f) ? s * b : (*d) * a + s * b;
    }
}

extern "C" int adh_axpby_strided(void* stream, float* dst, int dst_cs, const float* src, int src_cs, int64_t P, int C,
                                 float a, float b) {
    if (!dst || !src || P < 1 || C < 4 || (C & 3) || (dst_cs & 3) || (src_cs & 3)) return ADH_E_ARG;
    hipLaunchKernelGGL(axpby_strided_kernel, dim3(adh_min_i(adh_ceil_div(P * (C / 4), 256), 8192)), dim3(256), 0,
                       (hipStream_t)stream, dst, dst_cs, src, src_cs, P, C / 4, a, b);
    return adh_check_launch();
}

// MaxPool2d(k, stride, pad) (medium_intensity.py:145,150; high_intensity.py:162,165; torchvision resnet/densenet
// stem pool k3 s2 p1); idx = winning input pixel, first on ties (scan order = ATen's)
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, int x_cs, int H, int W, int CQ, int k,
                                                      int stride, int pad, int OH, int OW, float* __restrict__ out,
                                                      int out_cs, int32_t* __restrict__ idx) {
    const int n = blockIdx.y;
    const int64_t total = (int64_t)OH * OW * CQ;
    const float* xn = x + (size_t)n * H * W * x_cs;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t op = t / CQ;
        const int c = (int)(t - op * CQ) * 4;
        const int oy = (int)(op / OW), ox = (int)(op - (int64_t)oy * OW);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int mi[4] = {-1, -1, -1, -1};
        for (int dy = 0; dy < k; ++dy) {
            const int iy = oy * stride - pad + dy;
            if (iy < 0 || iy >= H) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int ix = ox * stride - pad + dx;
                if (ix < 0 || ix >= W) continue;
                const int ip = iy * W + ix;
                const f32x4 v = *reinterpret_cast<const f32x4*>(xn + (size_t)ip * x_cs + c);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (v[j] > m[j] || mi[j] < 0) {
                        m[j] = v[j];
                        mi[j] = ip;
                    }
            }
        }
        const size_t o = (size_t)n * OH * OW + op;
        *reinterpret_cast<f32x4*>(out + o * out_cs + c) = m;
        if (idx) {
#pragma unroll
            for (int j = 0; j < 4; ++j) idx[o * (CQ * 4) + c + j] = mi[j];
        }
    }
}

extern "C" int adh_maxpool(void* stream, const float* x, int x_cs, int N, int H, int W, int C, int k, int stride, int pad,
                           float* out, int out_cs, int32_t* idx) {
    if (!x || !out || k < 1 || stride < 1 || pad < 0 || pad > k / 2 || C < 4 || (C & 3)) return ADH_E_ARG;
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    if (OH < 1 || OW < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(maxpool_kernel, dim3(adh_min_i(adh_ceil_div((int64_t)OH * OW * (C / 4), 256), 4096), N), dim3(256), 0,
                       (hipStream_t)stream, x, x_cs, H, W, C / 4, k, stride, pad, OH, OW, out, out_cs, idx);
    return adh_check_launch();
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ g, int g_cs,
                                                          const int32_t* __restrict__ idx, int OH, int OW, int C, int k,
                                                          int stride, int pad, int H, int W, float* __restrict__ gx,
                                                          int gx_cs) {
    // gather form: every input pixel sums the windows that cover it and elected it (deterministic)
    const int n = blockIdx.y;
    const int64_t total = (int64_t)H * W * C;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ip = t / C;
        const int c = (int)(t - ip * C);
        const int iy = (int)(ip / W), ix = (int)(ip - (int64_t)iy * W);
        float v = 0.f;
        // windows oy with oy*stride - pad <= iy <= oy*stride - pad + k - 1
        int oy_lo = (iy + pad - k + 1 + stride - 1) / stride;
        if (iy + pad - k + 1 < 0) oy_lo = 0;
        const int oy_hi = adh_min_i((iy + pad) / stride, OH - 1);
        int ox_lo = (ix + pad - k + 1 + stride - 1) / stride;
        if (ix + pad - k + 1 < 0) ox_lo = 0;
        const int ox_hi = adh_min_i((ix + pad) / stride, OW - 1);
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const size_t o = (size_t)n * OH * OW + (size_t)oy * OW + ox;
                if (idx[o * C + c] == (int)ip) v += g[o * g_cs + c];
            }
        gx[((size_t)n * H * W + ip) * gx_cs + c] = v;
    }
}

extern "C" int adh_maxpool_bwd(void* stream, const float* g, int g_cs, const int32_t* idx, int N, int OH, int OW, int C,
                               int k, int stride, int pad, int H, int W, float* gx, int gx_cs) {
    if (!g || !idx || !gx || k < 1 || stride < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(adh_min_i(adh_ceil_div((int64_t)H * W * C, 256), 4096), N), dim3(256), 0,
                       (hipStream_t)stream, g, g_cs, idx, OH, OW, C, k, stride, pad, H, W, gx, gx_cs);
    return adh_check_launch();
}

__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, int x_cs, int H, int W, int CQ, int k,
                                                      int OH, int OW, float* __restrict__ out, int out_cs) {
    const int n = blockIdx.y;
    const int64_t total = (int64_t)OH * OW * CQ;
    const float* xn = x + (size_t)n * H * W * x_cs;
    const float inv = 1.0f / (float)(k * k);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t op = t / CQ;
        const int c = (int)(t - op * CQ) * 4;
        const int oy = (int)(op / OW), ox = (int)(op - (int64_t)oy * OW);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx)
                s += *reinterpret_cast<const f32x4*>(xn + ((size_t)(oy * k + dy) * W + ox * k + dx) * x_cs + c);
        *reinterpret_cast<f32x4*>(out + ((size_t)n * OH * OW + op) * out_cs + c) = s * inv;
    }
}

extern "C" int adh_avgpool(void* stream, const float* x, int x_cs, int N, int H, int W, int C, int k, float* out, int out_cs) {
    if (!x || !out || k < 1 || C < 4 || (C & 3) || H < k || W < k) return ADH_E_ARG;
    const int OH = H / k, OW = W / k;
    hipLaunchKernelGGL(avgpool_kernel, dim3(adh_min_i(adh_ceil_div((int64_t)OH * OW * (C / 4), 256), 4096), N), dim3(256), 0,
                       (hipStream_t)stream, x, x_cs, H, W, C / 4, k, OH, OW, out, out_cs);
    return adh_check_launch();
}

__global__ __launch_bounds__(256) void global_avgpool_bwd_kernel(const float* __restrict__ g, int HW, int C,
                                                                 float* __restrict__ gx, int gx_cs) {
    const int n = blockIdx.y;
    const int64_t total = (int64_t)HW * C;
    const float inv = 1.0f / (float)HW;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = t / C;
        const int c = (int)(t - p * C);
        gx[((size_t)n * HW + p) * gx_cs + c] = g[(size_t)n * C + c] * inv;
    }
}

extern "C" int adh_global_avgpool_bwd(void* stream, const float* g, int N, int HW, int C, float* gx, int gx_cs) {
    if (!g || !gx || N < 1 || HW < 1 || C < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(global_avgpool_bwd_kernel, dim3(adh_min_i(adh_ceil_div((int64_t)HW * C, 256), 2048), N), dim3(256), 0,
                       (hipStream_t)stream, g, HW, C, gx, gx_cs);
    return adh_check_launch();
}

__global__ void mul_kernel(float* __restrict__ dst, const float* __restrict__ a, const float* __restrict__ b, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = a[i] * b[i];
}

extern "C" int adh_mul(void* stream, float* dst, const float* a, const float* b, int64_t n) {
    if (!dst || !a || !b || n < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(mul_kernel, dim3(adh_min_i(adh_ceil_div(n, 256), 8192)), dim3(256), 0, (hipStream_t)stream, dst, a, b, n);
    return adh_check_launch();
}

// bilinear resize (F.interpolate align_corners False: medium_intensity.py:93-99; nn.UpsamplingBilinear2d
// = align_corners True: medium_intensity.py:147,152).  Source index computation follows ATen's
// area_pixel_compute_source_index.
__device__ __forceinline__ void bilin_src(int o, int in, int out, int align, int& i0, int& i1, float& l1) {
    float src;
    if (align) {
        const float sc = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
        src = sc * (float)o;
    } else {
        const float sc = (float)in / (float)out;
        src = sc * ((float)o + 0.5f) - 0.5f;
        if (src < 0.f) src = 0.f;
    }
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ x, int x_cs, int H, int W, int CQ, int OH,
                                                       int OW, int align, float* __restrict__ out, int out_cs) {
    const int n = blockIdx.y;
    const int64_t total = (int64_t)OH * OW * CQ;
    const float* xn = x + (size_t)n * H * W * x_cs;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t op = t / CQ;
        const int c = (int)(t - op * CQ) * 4;
        const int oy = (int)(op / OW), ox = (int)(op - (int64_t)oy * OW);
        int y0, y1, x0, x1;
        float ly, lx;
        bilin_src(oy, H, OH, align, y0, y1, ly);
        bilin_src(ox, W, OW, align, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const f32x4 v00 = *reinterpret_cast<const f32x4*>(xn + ((size_t)y0 * W + x0) * x_cs + c);
        const f32x4 v01 = *reinterpret_cast<const f32x4*>(xn + ((size_t)y0 * W + x1) * x_cs + c);
        const f32x4 v10 = *reinterpret_cast<const f32x4*>(xn + ((size_t)y1 * W + x0) * x_cs + c);
        const f32x4 v11 = *reinterpret_cast<const f32x4*>(xn + ((size_t)y1 * W + x1) * x_cs + c);
        const f32x4 r = (v00 * hx + v01 * lx) * hy + (v10 * hx + v11 * lx) * ly;
        *reinterpret_cast<f32x4*>(out + ((size_t)n * OH * OW + op) * out_cs + c) = r;
    }
}

extern "C" int adh_bilinear(void* stream, const float* x, int x_cs, int N, int H, int W, int C, int OH, int OW,
                            int align_corners, float* out, int out_cs) {
    if (!x || !out || C < 4 || (C & 3) || H < 1 || W < 1 || OH < 1 || OW < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(bilinear_kernel, dim3(adh_min_i(adh_ceil_div((int64_t)OH * OW * (C / 4), 256), 4096), N), dim3(256), 0,
                       (hipStream_t)stream, x, x_cs, H, W, C / 4, OH, OW, align_corners, out, out_cs);
    return adh_check_launch();
}

// adjoint (scatter with fp32 atomics; gx must be zero-initialised by the caller)
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ g, int g_cs, int H, int W, int C, int OH,
                                                           int OW, int align, float* __restrict__ gx, int gx_cs) {
    const int n = blockIdx.y;
    const int64_t total = (int64_t)OH * OW * C;
    float* gxn = gx + (size_t)n * H * W * gx_cs;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t op = t / C;
        const int c = (int)(t - op * C);
        const int oy = (int)(op / OW), ox = (int)(op - (int64_t)oy * OW);
        int y0, y1, x0, x1;
        float ly, lx;
        bilin_src(oy, H, OH, align, y0, y1, ly);
        bilin_src(ox, W, OW, align, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float gv = g[((size_t)n * OH * OW + op) * g_cs + c];
        atomicAdd(gxn + ((size_t)y0 * W + x0) * gx_cs + c, gv * hy * hx);
        atomicAdd(gxn + ((size_t)y0 * W + x1) * gx_cs + c, gv * hy * lx);
        atomicAdd(gxn + ((size_t)y1 * W + x0) * gx_cs + c, gv * ly * hx);
        atomicAdd(gxn + ((size_t)y1 * W + x1) * gx_cs + c, gv * ly * lx);
    }
}

extern "C" int adh_bilinear_bwd(void* stream, const float* g, int g_cs, int N, int H, int W, int C, int OH, int OW,
                                int align_corners, float* gx, int gx_cs) {
    if (!g || !gx || C < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(adh_min_i(adh_ceil_div((int64_t)OH * OW * C, 256), 4096), N), dim3(256), 0,
                       (hipStream_t)stream, g, g_cs, H, W, C, OH, OW, align_corners, gx, gx_cs);
    return adh_check_launch();
}
