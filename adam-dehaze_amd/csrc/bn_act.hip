// BatchNorm2d (+ residual + ReLU) forward/backward passes around the conv kernels, gfx950.
// HBM-bound streaming kernels: 16 B per lane, channel-quad per thread, per-block partial sums
// reduced in a fixed order (deterministic).  Follows nn.BatchNorm2d as used by
// /root/reference models/dehazing/base_model.py:15-19,36-41 (eps 1e-5, momentum 0.1, biased batch
// variance for normalisation, unbiased for the running estimate).
#include "common.h"

// ---------------------------------------------------------------------------------------------
// finalize: partials[nblk][2][NcP] -> scale/shift/mean/invstd (+ running stats)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partials, int nblk, int NcP, int C,
                                                           double count, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float momentum,
                                                           float* running_mean, float* running_var, float* scale,
                                                           float* shift, float* save_mean, float* save_invstd) {
    // 32 channels x 32 row slices per block; each thread keeps 4 independent fp64 chains in flight
    __shared__ double red[2][32][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    double s0 = 0.0, s1 = 0.0, q0 = 0.0, q1 = 0.0;
    if (c < C) {
        int b = ry;
        for (; b + 32 < nblk; b += 64) {
            s0 += (double)partials[((size_t)b * 2 + 0) * NcP + c];
            q0 += (double)partials[((size_t)b * 2 + 1) * NcP + c];
            s1 += (double)partials[((size_t)(b + 32) * 2 + 0) * NcP + c];
            q1 += (double)partials[((size_t)(b + 32) * 2 + 1) * NcP + c];
        }
        for (; b < nblk; b += 32) {
            s0 += (double)partials[((size_t)b * 2 + 0) * NcP + c];
            q0 += (double)partials[((size_t)b * 2 + 1) * NcP + c];
        }
    }
    red[0][ry][cx] = s0 + s1;
    red[1][ry][cx] = q0 + q1;
    __syncthreads();
    if (ry == 0 && c < C) {
        double S = 0.0, Q = 0.0;
        for (int r = 0; r < 32; ++r) {
            S += red[0][r][cx];
            Q += red[1][r][cx];
        }
        const double mean = S / count;
        double var = Q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)eps);
        const float gm = gamma ? gamma[c] : 1.f;
        const float bt = beta ? beta[c] : 0.f;
        const float sc = (float)(gm * invstd);
        scale[c] = sc;
        shift[c] = (float)(bt - mean * gm * invstd);
        if (save_mean) save_mean[c] = (float)mean;
        if (save_invstd) save_invstd[c] = (float)invstd;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        if (running_var) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

extern "C" int adh_bn_finalize(void* stream, const float* partials, int nblk, int NcP, int C, double count,
                               const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* scale, float* shift, float* save_mean, float* save_invstd) {
    if (!partials || !scale || !shift || nblk < 1 || C < 1 || NcP < C || count <= 0) return ADH_E_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(adh_ceil_div(C, 32)), dim3(1024), 0, (hipStream_t)stream, partials, nblk,
                       NcP, C, count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean,
                       save_invstd);
    return adh_check_launch();
}

__global__ void bn_fold_eval_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv,
                                    float eps, const float* conv_bias, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.0f / sqrtf(rv[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.f) * invstd;
    float sh = (beta ? beta[c] : 0.f) - rm[c] * sc;
    if (conv_bias) sh += conv_bias[c] * sc;
    scale[c] = sc;
    shift[c] = sh;
}

extern "C" int adh_bn_fold_eval(void* stream, int C, const float* gamma, const float* beta, const float* running_mean,
                                const float* running_var, float eps, const float* conv_bias, float* scale,
                                float* shift) {
    if (C < 1 || !running_mean || !running_var || !scale || !shift) return ADH_E_ARG;
    hipLaunchKernelGGL(bn_fold_eval_kernel, dim3(adh_ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, C, gamma,
                       beta, running_mean, running_var, eps, conv_bias, scale, shift);
    return adh_check_launch();
}

// ---------------------------------------------------------------------------------------------
// apply: out = act(y*scale + shift (+ residual))
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y, int y_cs,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ residual, int res_cs, int act,
                                                       float* __restrict__ out, int out_cs, int64_t P, int CQ) {
    const int64_t total = P * CQ;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = idx / CQ;
        const int c = (int)(idx - p * CQ) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(y + p * y_cs + c);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c);
        v = v * sc + sh;
        if (residual) v += *reinterpret_cast<const f32x4*>(residual + p * res_cs + c);
        if (act == ADH_ACT_RELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        *reinterpret_cast<f32x4*>(out + p * out_cs + c) = v;
    }
}

extern "C" int adh_bn_apply(void* stream, const float* y, int y_cs, const float* scale, const float* shift,
                            const float* residual, int res_cs, int act, float* out, int out_cs, int64_t P, int C) {
    if (!y || !scale || !shift || !out || P < 1 || C < 4 || (C & 3) || (y_cs & 3) || (out_cs & 3) || (res_cs & 3))
        return ADH_E_ARG;
    const int CQ = C / 4;
    const int blocks = adh_min_i(adh_ceil_div(P * CQ, 256), 256 * 16);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, y, y_cs, scale, shift, residual,
                       res_cs, act, out, out_cs, P, CQ);
    return adh_check_launch();
}

// ---------------------------------------------------------------------------------------------
// backward pass 1: per-block sums of g and g*xhat
// ---------------------------------------------------------------------------------------------
#define BNB_PPB 2048  // pixels per block

extern "C" int adh_bn_bwd_num_blocks(int64_t P, int C) {
    (void)C;
    return adh_ceil_div(P, BNB_PPB);
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ g_out, int g_cs,
                                                            const float* __restrict__ out, int out_cs, int act,
                                                            const float* __restrict__ y, int y_cs,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* partials, int64_t P,
                                                            int C) {
    __shared__ f32x4 red[2][256];
    const int CQ = C / 4;
    const int R = 256 / CQ;  // pixel rows handled concurrently
    const int cq = threadIdx.x % CQ;
    const int prow = threadIdx.x / CQ;
    const bool active = prow < R;
    const int c = cq * 4;
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
        const int64_t p0 = (int64_t)blockIdx.x * BNB_PPB;
        const int64_t p1 = p0 + BNB_PPB < P ? p0 + BNB_PPB : P;
        for (int64_t p = p0 + prow; p < p1; p += R) {
            f32x4 g = *reinterpret_cast<const f32x4*>(g_out + p * g_cs + c);
            if (act == ADH_ACT_RELU) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(out + p * out_cs + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
            }
            const f32x4 yy = *reinterpret_cast<const f32x4*>(y + p * y_cs + c);
            sg += g;
            sgx += g * ((yy - mu) * is);
        }
    }
    red[0][threadIdx.x] = sg;
    red[1][threadIdx.x] = sgx;
    __syncthreads();
    if (threadIdx.x < CQ) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < R; ++r) {
            a += red[0][r * CQ + threadIdx.x];
            b += red[1][r * CQ + threadIdx.x];
        }
        *reinterpret_cast<f32x4*>(partials + ((size_t)blockIdx.x * 2 + 0) * C + c) = a;
        *reinterpret_cast<f32x4*>(partials + ((size_t)blockIdx.x * 2 + 1) * C + c) = b;
    }
}

extern "C" int adh_bn_bwd_reduce(void* stream, const float* g_out, int g_cs, const float* out, int out_cs, int act,
                                 const float* y, int y_cs, const float* mean, const float* invstd, float* partials,
                                 int64_t P, int C) {
    if (!g_out || !y || !mean || !invstd || !partials || P < 1 || C < 4 || (C & 3) || C > 1024) return ADH_E_ARG;
    if (act == ADH_ACT_RELU && !out) return ADH_E_ARG;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(adh_bn_bwd_num_blocks(P, C)), dim3(256), 0, (hipStream_t)stream, g_out,
                       g_cs, out, out_cs, act, y, y_cs, mean, invstd, partials, P, C);
    return adh_check_launch();
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nblk, int C,
                                                              double count, const float* gamma, const float* invstd,
                                                              float* dgamma, float* dbeta, int accumulate, float* coef) {
    __shared__ double red[2][8][32];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    double s = 0.0, q = 0.0;
    if (c < C) {
        for (int b = ry; b < nblk; b += 8) {
            s += (double)partials[((size_t)b * 2 + 0) * C + c];
            q += (double)partials[((size_t)b * 2 + 1) * C + c];
        }
    }
    red[0][ry][cx] = s;
    red[1][ry][cx] = q;
    __syncthreads();
    if (ry == 0 && c < C) {
        double S = 0.0, Q = 0.0;
        for (int r = 0; r < 8; ++r) {
            S += red[0][r][cx];
            Q += red[1][r][cx];
        }
        if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)Q : (float)Q;
        if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)S : (float)S;
        coef[0 * C + c] = (gamma ? gamma[c] : 1.f) * invstd[c];
        coef[1 * C + c] = (float)(S / count);
        coef[2 * C + c] = (float)(Q / count);
    }
}

extern "C" int adh_bn_bwd_finalize(void* stream, const float* partials, int nblk, int C, double count,
                                   const float* gamma, const float* invstd, float* dgamma, float* dbeta, int accumulate,
                                   float* coef) {
    if (!partials || !invstd || !coef || nblk < 1 || C < 1 || count <= 0) return ADH_E_ARG;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(adh_ceil_div(C, 32)), dim3(256), 0, (hipStream_t)stream, partials,
                       nblk, C, count, gamma, invstd, dgamma, dbeta, accumulate, coef);
    return adh_check_launch();
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g_out, int g_cs,
                                                           const float* __restrict__ out, int out_cs, int act,
                                                           const float* __restrict__ y, int y_cs,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, int training,
                                                           float* __restrict__ g_y, int gy_cs, float* __restrict__ g_res,
                                                           int gres_cs, int64_t P, int C) {
    const int CQ = C / 4;
    const int64_t total = P * CQ;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = idx / CQ;
        const int c = (int)(idx - p * CQ) * 4;
        f32x4 g = *reinterpret_cast<const f32x4*>(g_out + p * g_cs + c);
        if (act == ADH_ACT_RELU) {
            const f32x4 o = *reinterpret_cast<const f32x4*>(out + p * out_cs + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
        }
        if (g_res) *reinterpret_cast<f32x4*>(g_res + p * gres_cs + c) = g;
        const f32x4 k0 = *reinterpret_cast<const f32x4*>(coef + c);
        f32x4 r;
        if (training) {
            const f32x4 mg = *reinterpret_cast<const f32x4*>(coef + C + c);
            const f32x4 mgx = *reinterpret_cast<const f32x4*>(coef + 2 * C + c);
            const f32x4 yy = *reinterpret_cast<const f32x4*>(y + p * y_cs + c);
            const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
            const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
            r = k0 * (g - mg - (yy - mu) * is * mgx);
        } else {
            r = k0 * g;
        }
        *reinterpret_cast<f32x4*>(g_y + p * gy_cs + c) = r;
    }
}

extern "C" int adh_bn_bwd_apply(void* stream, const float* g_out, int g_cs, const float* out, int out_cs, int act,
                                const float* y, int y_cs, const float* mean, const float* invstd, const float* coef,
                                int training, float* g_y, int gy_cs, float* g_res, int gres_cs, int64_t P, int C) {
    if (!g_out || !coef || !g_y || P < 1 || C < 4 || (C & 3)) return ADH_E_ARG;
    if (training && (!y || !mean || !invstd)) return ADH_E_ARG;
    if (act == ADH_ACT_RELU && !out) return ADH_E_ARG;
    const int blocks = adh_min_i(adh_ceil_div(P * (C / 4), 256), 256 * 16);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g_out, g_cs, out, out_cs, act,
                       y, y_cs, mean, invstd, coef, training, g_y, gy_cs, g_res, gres_cs, P, C);
    return adh_check_launch();
}
