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
                                                           float* shift, float* save_mean, float* save_invstd, int64_t* num_batches_tracked) {
    // 32 channels x 32 row slices per block; each thread keeps 4 independent fp64 chains in flight
    __shared__ double red[2][32][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    if (c < C) {
        int b = ry;
        for (; b + 96 < nblk; b += 128) {   // eight independent loads in flight: the loop is latency-bound
            s0 += (double)partials[((size_t)b * 2 + 0) * NcP + c];
            q0 += (double)partials[((size_t)b * 2 + 1) * NcP + c];
            s1 += (double)partials[((size_t)(b + 32) * 2 + 0) * NcP + c];
            q1 += (double)partials[((size_t)(b + 32) * 2 + 1) * NcP + c];
            s2 += (double)partials[((size_t)(b + 64) * 2 + 0) * NcP + c];
            q2 += (double)partials[((size_t)(b + 64) * 2 + 1) * NcP + c];
            s3 += (double)partials[((size_t)(b + 96) * 2 + 0) * NcP + c];
            q3 += (double)partials[((size_t)(b + 96) * 2 + 1) * NcP + c];
        }
        for (; b < nblk; b += 32) {
            s0 += (double)partials[((size_t)b * 2 + 0) * NcP + c];
            q0 += (double)partials[((size_t)b * 2 + 1) * NcP + c];
        }
    }
    s0 += s2;
    s1 += s3;
    q0 += q2;
    q1 += q3;
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
        if (num_batches_tracked && c == 0) *num_batches_tracked += 1;     // BatchNorm2d's counter (one writer)
    }
}

extern "C" int adh_bn_finalize(void* stream, const float* partials, int nblk, int NcP, int C, double count,
                               const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* scale, float* shift, float* save_mean, float* save_invstd,
                               int64_t* num_batches_tracked) {
    if (!partials || !scale || !shift || nblk < 1 || C < 1 || NcP < C || count <= 0) return ADH_E_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(adh_ceil_div(C, 32)), dim3(1024), 0, (hipStream_t)stream, partials, nblk,
                       NcP, C, count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean,
                       save_invstd, num_batches_tracked);
    return adh_check_launch();
}

// ---------------------------------------------------------------------------------------------
// Synchronised BatchNorm (data parallel, SURVEY 8e mode ii): the per-block partials are summed to sums[2][C] (+ the
// element count in sums[2 C]) in fp64 on the device, the host all-reduces that small vector over the ranks (RCCL), and the
// *_sums forms of the finalize kernels work from the global sums -- the statistics of the single-process reference's
// BatchNorm2d over the whole batch (/root/reference models/dehazing/base_model.py:15-16).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bn_partial_sums_kernel(const float* __restrict__ partials, int nblk, int pitch, int C,
                                                               double count, double* __restrict__ sums) {
    __shared__ double red[2][32][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    double s0 = 0.0, s1 = 0.0, q0 = 0.0, q1 = 0.0;
    if (c < C) {
        int b = ry;
        for (; b + 32 < nblk; b += 64) {
            s0 += (double)partials[((size_t)b * 2 + 0) * pitch + c];
            q0 += (double)partials[((size_t)b * 2 + 1) * pitch + c];
            s1 += (double)partials[((size_t)(b + 32) * 2 + 0) * pitch + c];
            q1 += (double)partials[((size_t)(b + 32) * 2 + 1) * pitch + c];
        }
        for (; b < nblk; b += 32) {
            s0 += (double)partials[((size_t)b * 2 + 0) * pitch + c];
            q0 += (double)partials[((size_t)b * 2 + 1) * pitch + c];
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
        sums[c] = S;
        sums[C + c] = Q;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) sums[2 * C] = count;
}

extern "C" int adh_bn_partial_sums(void* stream, const float* partials, int nblk, int pitch, int C, double count, double* sums) {
    if (!partials || !sums || nblk < 1 || C < 1 || pitch < C || count <= 0) return ADH_E_ARG;
    hipLaunchKernelGGL(bn_partial_sums_kernel, dim3(adh_ceil_div(C, 32)), dim3(1024), 0, (hipStream_t)stream, partials, nblk,
                       pitch, C, count, sums);
    return adh_check_launch();
}

__global__ void bn_finalize_sums_kernel(const double* __restrict__ sums, int C, const float* __restrict__ gamma,
                                        const float* __restrict__ beta, float eps, float momentum, float* running_mean,
                                        float* running_var, float* scale, float* shift, float* save_mean, float* save_invstd,
                                        int64_t* num_batches_tracked) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double count = sums[2 * C];
    const double mean = sums[c] / count;
    double var = sums[C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float gm = gamma ? gamma[c] : 1.f;
    const float bt = beta ? beta[c] : 0.f;
    scale[c] = (float)(gm * invstd);
    shift[c] = (float)(bt - mean * gm * invstd);
    if (save_mean) save_mean[c] = (float)mean;
    if (save_invstd) save_invstd[c] = (float)invstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
    if (num_batches_tracked && c == 0) *num_batches_tracked += 1;
}

extern "C" int adh_bn_finalize_sums(void* stream, const double* sums, int C, const float* gamma, const float* beta, float eps,
                                    float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                                    float* save_mean, float* save_invstd, int64_t* num_batches_tracked) {
    if (!sums || !scale || !shift || C < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(bn_finalize_sums_kernel, dim3(adh_ceil_div(C, 128)), dim3(128), 0, (hipStream_t)stream, sums, C, gamma,
                       beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd, num_batches_tracked);
    return adh_check_launch();
}

// backward: d-gamma / d-beta stay the LOCAL sums (the gradient all-reduce averages them like every other parameter
// gradient); the two means inside the input gradient come from the GLOBAL sums (torch.nn.SyncBatchNorm's backward)
__global__ void bn_bwd_finalize_sums_kernel(const double* __restrict__ local_sums, const double* __restrict__ global_sums, int C,
                                            const float* gamma, const float* invstd, float* dgamma, float* dbeta,
                                            int accumulate, float* coef) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double count = global_sums[2 * C];
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)local_sums[C + c] : (float)local_sums[C + c];
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)local_sums[c] : (float)local_sums[c];
    coef[0 * C + c] = (gamma ? gamma[c] : 1.f) * invstd[c];
    coef[1 * C + c] = (float)(global_sums[c] / count);
    coef[2 * C + c] = (float)(global_sums[C + c] / count);
}

extern "C" int adh_bn_bwd_finalize_sums(void* stream, const double* local_sums, const double* global_sums, int C,
                                        const float* gamma, const float* invstd, float* dgamma, float* dbeta, int accumulate,
                                        float* coef) {
    if (!local_sums || !global_sums || !invstd || !coef || C < 1) return ADH_E_ARG;
    hipLaunchKernelGGL(bn_bwd_finalize_sums_kernel, dim3(adh_ceil_div(C, 128)), dim3(128), 0, (hipStream_t)stream, local_sums,
                       global_sums, C, gamma, invstd, dgamma, dbeta, accumulate, coef);
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

// eval-mode (frozen statistics) BatchNorm backward helpers: with out_pre = gamma*xhat + beta the normalised input is
// xhat = (out_pre - beta) / gamma, so the train-mode reduction kernels give d-gamma / d-beta when fed y := out_pre,
// mean := beta, invstd := 1/gamma (0 where gamma == 0: xhat cannot be recovered there and d-gamma reads 0)
__global__ void bn_eval_bwd_vectors_kernel(int C, int C4, const float* gamma, const float* beta, float* mean, float* invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C4) return;
    const float g = c < C ? gamma[c] : 0.f;
    mean[c] = c < C ? beta[c] : 0.f;
    invstd[c] = g != 0.f ? 1.0f / g : 0.f;
}

extern "C" int adh_bn_eval_bwd_vectors(void* stream, int C, int C4, const float* gamma, const float* beta, float* mean,
                                       float* invstd) {
    if (C < 1 || C4 < C || !gamma || !beta || !mean || !invstd) return ADH_E_ARG;
    hipLaunchKernelGGL(bn_eval_bwd_vectors_kernel, dim3(adh_ceil_div(C4, 256)), dim3(256), 0, (hipStream_t)stream, C, C4, gamma,
                       beta, mean, invstd);
    return adh_check_launch();
}

// ---------------------------------------------------------------------------------------------
// apply: out = act(y*scale + shift (+ residual))
// ---------------------------------------------------------------------------------------------
// Streaming layout shared by the element-wise kernels below: the launch has T = gridDim.x * 256 threads with T a
// multiple of CQ (channel quads per pixel), so a thread keeps one channel quad for the whole sweep -- its per-channel
// constants are loaded once and no index division happens inside the loop -- and walks pixels p0, p0 + T/CQ, ...
// four at a time (4-12 independent 16-byte loads in flight per lane: these kernels are HBM-bound).  The tensors are
// far larger than L2 + Infinity Cache and each element is touched once per pass, so all of their loads / stores carry the
// non-temporal hint (measured on 8 x 512 x 1024 x 96: bn_apply 0.632 -> 0.603 ms, bn_bwd_apply 1.30 -> 1.16-1.22 ms).
#ifndef EW_UNROLL
#define EW_UNROLL 8
#endif
// Grid caps (blocks of 256 threads; a block walks EW_UNROLL pixels per loop trip at a stride of the whole grid).  Swept on
// 8 x 512 x 1024 x 96 at the end of round 4 (profiles/r04b_sweep_bn_grid.txt; caps of 8 / 32 / 64 / 128 / 512 x 256 blocks): the
// two-stream kernel (bn_apply: y -> out) is fastest with one trip per thread (0.608 -> 0.541 ms), the three-stream one
// (bn_bwd_apply: g, y -> g_y) at 32 x 256 (1.220 -> 1.135 ms) and gets slower again beyond -- the optimum is not monotonic in
// the grid size (which pages of which HBM channels the concurrent blocks touch), so these are measured constants, not a rule.
#ifndef EW_MAXBLK_APPLY
#define EW_MAXBLK_APPLY (256 * 512)
#endif
#ifndef EW_MAXBLK_BWD
#define EW_MAXBLK_BWD (256 * 32)
#endif
static int ew_blocks(int64_t P, int CQ, int EW_MAXBLK) {
    int g = CQ, r = 256;   // gcd(CQ, 256)
    while (r) { const int t = g % r; g = r; r = t; }
    const int mult = CQ / g;                                  // blocks must be a multiple of this
    int64_t want = (P * CQ + 256 * EW_UNROLL - 1) / (256 * EW_UNROLL);
    if (want > EW_MAXBLK) want = EW_MAXBLK;
    int64_t blocks = (want + mult - 1) / mult * mult;
    if (blocks < mult) blocks = mult;
    return (int)blocks;
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ y, int y_cs,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ residual, int res_cs, int act,
                                                       float* __restrict__ out, int out_cs, int64_t P, int CQ,
                                                       uint8_t* __restrict__ mask_bits) {
    const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
    const int64_t pstep = (int64_t)gridDim.x * 256 / CQ;
    int64_t p = t / CQ;
    const int c = (int)(t - p * CQ) * 4;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c);
    for (; p < P; p += EW_UNROLL * pstep) {
        f32x4 v[EW_UNROLL], r[EW_UNROLL];
#pragma unroll
        for (int u = 0; u < EW_UNROLL; ++u) {
            const int64_t q = p + u * pstep;
            const bool ok = q < P;
            v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y + (ok ? q : p) * y_cs + c));
            if (residual) r[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(residual + (ok ? q : p) * res_cs + c));
        }
#pragma unroll
        for (int u = 0; u < EW_UNROLL; ++u) {
            const int64_t q = p + u * pstep;
            f32x4 w;
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = fmaf(v[u][j], sc[j], sh[j]);   // (the backward mask below repeats exactly this)
            if (residual) w += r[u];
            if (mask_bits) {
                // ReLU mask of this quad as a nibble; quads 2m, 2m+1 of a pixel sit in adjacent lanes (CQ is even) and share
                // byte (q*CQ + cq) / 2: the backward passes read 1 bit per element instead of `out` (4 bytes)
                const int nib = (w[0] > 0.f ? 1 : 0) | (w[1] > 0.f ? 2 : 0) | (w[2] > 0.f ? 4 : 0) | (w[3] > 0.f ? 8 : 0);
                const int hi = __shfl_xor(nib, 1, 64);
                if (q < P && !(threadIdx.x & 1)) mask_bits[(q * CQ + (c >> 2)) >> 1] = (uint8_t)(nib | (hi << 4));
            }
            if (act == ADH_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = fmaxf(w[j], 0.f);
            }
            if (q < P) __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(out + q * out_cs + c));
        }
    }
}

extern "C" int adh_bn_apply(void* stream, const float* y, int y_cs, const float* scale, const float* shift,
                            const float* residual, int res_cs, int act, float* out, int out_cs, int64_t P, int C,
                            uint8_t* mask_bits) {
    if (!y || !scale || !shift || !out || P < 1 || C < 4 || (C & 3) || (y_cs & 3) || (out_cs & 3) || (res_cs & 3))
        return ADH_E_ARG;
    const int CQ = C / 4;
    if (mask_bits && ((CQ & 1) || act != ADH_ACT_RELU)) return ADH_E_ARG;   // nibble pairs: an even number of quads per pixel
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks(P, CQ, EW_MAXBLK_APPLY)), dim3(256), 0, (hipStream_t)stream, y, y_cs, scale, shift,
                       residual, res_cs, act, out, out_cs, P, CQ, mask_bits);
    return adh_check_launch();
}

// ---------------------------------------------------------------------------------------------
// backward pass 1: per-block sums of g and g*xhat
// ---------------------------------------------------------------------------------------------
#ifndef BNB_PPB
#define BNB_PPB 512   // pixels per block (swept at the end of round 4 over 256 .. 8192, profiles/r04b_sweep_bn_grid.txt: 2048 left the half- and
                      // quarter-resolution layers with 512 / 128 blocks on 256 CUs; bn_bwd_reduce 5.36 -> 4.65 ms per step)
#endif
#ifndef BNB_UNROLL
#define BNB_UNROLL 4  // pixels per trip and thread (x 2 streams of 16-byte loads in flight)
#endif

extern "C" int adh_bn_bwd_num_blocks(int64_t P, int C) {
    (void)C;
    return adh_ceil_div(P, BNB_PPB);
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ g_out, int g_cs,
                                                            const float* __restrict__ out, int out_cs, int act,
                                                            const float* __restrict__ y, int y_cs,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, float* partials, int64_t P,
                                                            int C, const float* __restrict__ mask_ss,
                                                            const uint8_t* __restrict__ mask_bits) {
    __shared__ f32x4 red[2][256];
    // channels in groups of at most 1024 (256 quads, one per thread): one group for every layer of the dehazing branches; the
    // resnet50 HDEN's 2048-channel BatchNorms (round 4) take two
  for (int cbase = 0; cbase < C; cbase += 1024) {
    const int CQ = (C - cbase < 1024 ? C - cbase : 1024) / 4;
    const int R = 256 / CQ;  // pixel rows handled concurrently
    const int cq = threadIdx.x % CQ;
    const int prow = threadIdx.x / CQ;
    const bool active = prow < R;
    const int c = cbase + cq * 4;
    f32x4 sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
        const f32x4 is = *reinterpret_cast<const f32x4*>(invstd + c);
        // mask_ss = {scale[C], shift[C]} of the forward pass: the ReLU mask is recomputed from y (out is not read)
        f32x4 msc = {0.f, 0.f, 0.f, 0.f}, msh = msc;
        if (mask_ss) {
            msc = *reinterpret_cast<const f32x4*>(mask_ss + c);
            msh = *reinterpret_cast<const f32x4*>(mask_ss + C + c);
        }
        const int64_t p0 = (int64_t)blockIdx.x * BNB_PPB;
        const int64_t p1 = p0 + BNB_PPB < P ? p0 + BNB_PPB : P;
        for (int64_t p = p0 + prow; p < p1; p += BNB_UNROLL * R) {
            f32x4 g[BNB_UNROLL], o[BNB_UNROLL], yy[BNB_UNROLL];
            int mb[BNB_UNROLL];
#pragma unroll
            for (int u = 0; u < BNB_UNROLL; ++u) {
                const int64_t q = p + u * R < p1 ? p + u * R : p;
                g[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g_out + q * g_cs + c));
                if (mask_bits) mb[u] = mask_bits[(q * (C / 4) + (c >> 2)) >> 1] >> (4 * ((c >> 2) & 1));
                else if (act == ADH_ACT_RELU && !mask_ss) o[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(out + q * out_cs + c));
                yy[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y + q * y_cs + c));
            }
#pragma unroll
            for (int u = 0; u < BNB_UNROLL; ++u) {
                if (p + u * R < p1) {
                    f32x4 gg = g[u];
                    if (mask_bits) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) gg[j] = ((mb[u] >> j) & 1) ? gg[j] : 0.f;
                    } else if (act == ADH_ACT_RELU) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float ov = mask_ss ? fmaf(yy[u][j], msc[j], msh[j]) : o[u][j];
                            gg[j] = ov > 0.f ? gg[j] : 0.f;
                        }
                    }
                    sg += gg;
                    sgx += gg * ((yy[u] - mu) * is);
                }
            }
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
    __syncthreads();   // (the next channel group reuses `red`)
  }
}

extern "C" int adh_bn_bwd_reduce(void* stream, const float* g_out, int g_cs, const float* out, int out_cs, int act,
                                 const float* y, int y_cs, const float* mean, const float* invstd, float* partials,
                                 int64_t P, int C, const float* mask_ss, const uint8_t* mask_bits) {
    if (!g_out || !y || !mean || !invstd || !partials || P < 1 || C < 4 || (C & 3) || C > 4096) return ADH_E_ARG;
    if (act == ADH_ACT_RELU && !out && !mask_ss && !mask_bits) return ADH_E_ARG;
    if (mask_bits && (((C / 4) & 1) || act != ADH_ACT_RELU)) return ADH_E_ARG;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(adh_bn_bwd_num_blocks(P, C)), dim3(256), 0, (hipStream_t)stream, g_out,
                       g_cs, out, out_cs, act, y, y_cs, mean, invstd, partials, P, C, mask_ss, mask_bits);
    return adh_check_launch();
}

__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partials, int nblk, int pitch, int C,
                                                               double count, const float* gamma, const float* invstd,
                                                               float* dgamma, float* dbeta, int accumulate, float* coef,
                                                               int centered) {
    // centered: row 1 holds sum g m (y - mean) (the fused data-gradient epilogue of conv_wino43.hip): times invstd = sum g m xhat
    // 32 channels x 32 row slices per block, four independent fp64 chains per thread (latency-bound loop)
    __shared__ double red[2][32][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0, q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
    if (c < C) {
        int b = ry;
        for (; b + 96 < nblk; b += 128) {
            s0 += (double)partials[((size_t)b * 2 + 0) * pitch + c];
            q0 += (double)partials[((size_t)b * 2 + 1) * pitch + c];
            s1 += (double)partials[((size_t)(b + 32) * 2 + 0) * pitch + c];
            q1 += (double)partials[((size_t)(b + 32) * 2 + 1) * pitch + c];
            s2 += (double)partials[((size_t)(b + 64) * 2 + 0) * pitch + c];
            q2 += (double)partials[((size_t)(b + 64) * 2 + 1) * pitch + c];
            s3 += (double)partials[((size_t)(b + 96) * 2 + 0) * pitch + c];
            q3 += (double)partials[((size_t)(b + 96) * 2 + 1) * pitch + c];
        }
        for (; b < nblk; b += 32) {
            s0 += (double)partials[((size_t)b * 2 + 0) * pitch + c];
            q0 += (double)partials[((size_t)b * 2 + 1) * pitch + c];
        }
    }
    red[0][ry][cx] = (s0 + s1) + (s2 + s3);
    red[1][ry][cx] = (q0 + q1) + (q2 + q3);
    __syncthreads();
    if (ry == 0 && c < C) {
        double S = 0.0, Q = 0.0;
        for (int r = 0; r < 32; ++r) {
            S += red[0][r][cx];
            Q += red[1][r][cx];
        }
        if (centered) Q *= (double)invstd[c];
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
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(adh_ceil_div(C, 32)), dim3(1024), 0, (hipStream_t)stream, partials,
                       nblk, C, C, count, gamma, invstd, dgamma, dbeta, accumulate, coef, 0);
    return adh_check_launch();
}

// the same from the rows adh_conv_wino43_dgrad_bnred wrote: partials[nblk][2][pitch] = (sum g m, sum g m (y - mean))
extern "C" int adh_bn_bwd_finalize_centered(void* stream, const float* partials, int nblk, int pitch, int C, double count,
                                            const float* gamma, const float* invstd, float* dgamma, float* dbeta,
                                            int accumulate, float* coef) {
    if (!partials || !invstd || !coef || nblk < 1 || C < 1 || pitch < C || count <= 0) return ADH_E_ARG;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(adh_ceil_div(C, 32)), dim3(1024), 0, (hipStream_t)stream, partials,
                       nblk, pitch, C, count, gamma, invstd, dgamma, dbeta, accumulate, coef, 1);
    return adh_check_launch();
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ g_out, int g_cs,
                                                           const float* __restrict__ out, int out_cs, int act,
                                                           const float* __restrict__ y, int y_cs,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, int training,
                                                           float* __restrict__ g_y, int gy_cs, float* __restrict__ g_res,
                                                           int gres_cs, int64_t P, int C, const float* __restrict__ mask_ss,
                                                           const uint8_t* __restrict__ mask_bits) {
    const int CQ = C / 4;
    const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
    const int64_t pstep = (int64_t)gridDim.x * 256 / CQ;
    int64_t p = t / CQ;
    const int c = (int)(t - p * CQ) * 4;
    const f32x4 k0 = *reinterpret_cast<const f32x4*>(coef + c);
    f32x4 mg = {0.f, 0.f, 0.f, 0.f}, kx = mg, mu = mg, msc = mg, msh = mg;
    if (mask_ss) {
        msc = *reinterpret_cast<const f32x4*>(mask_ss + c);
        msh = *reinterpret_cast<const f32x4*>(mask_ss + C + c);
    }
    if (training) {
        mg = *reinterpret_cast<const f32x4*>(coef + C + c);
        const f32x4 mgx = *reinterpret_cast<const f32x4*>(coef + 2 * C + c);
        mu = *reinterpret_cast<const f32x4*>(mean + c);
        kx = *reinterpret_cast<const f32x4*>(invstd + c) * mgx;
    }
    for (; p < P; p += EW_UNROLL * pstep) {
        f32x4 g[EW_UNROLL], o[EW_UNROLL], yy[EW_UNROLL];
        int mb[EW_UNROLL];
#pragma unroll
        for (int u = 0; u < EW_UNROLL; ++u) {
            const int64_t q = p + u * pstep < P ? p + u * pstep : p;
            g[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g_out + q * g_cs + c));
            if (mask_bits) mb[u] = mask_bits[(q * CQ + (c >> 2)) >> 1] >> (4 * ((c >> 2) & 1));
            else if (act == ADH_ACT_RELU && !mask_ss) o[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(out + q * out_cs + c));
            if (training) yy[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y + q * y_cs + c));
        }
#pragma unroll
        for (int u = 0; u < EW_UNROLL; ++u) {
            const int64_t q = p + u * pstep;
            if (q < P) {
                f32x4 gg = g[u];
                if (mask_bits) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) gg[j] = ((mb[u] >> j) & 1) ? gg[j] : 0.f;
                } else if (act == ADH_ACT_RELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float ov = mask_ss ? fmaf(yy[u][j], msc[j], msh[j]) : o[u][j];
                        gg[j] = ov > 0.f ? gg[j] : 0.f;
                    }
                }
                if (g_res) __builtin_nontemporal_store(gg, reinterpret_cast<f32x4*>(g_res + q * gres_cs + c));
                f32x4 r;
                if (training) r = k0 * (gg - mg - (yy[u] - mu) * kx);
                else r = k0 * gg;
                __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(g_y + q * gy_cs + c));
            }
        }
    }
}

extern "C" int adh_bn_bwd_apply(void* stream, const float* g_out, int g_cs, const float* out, int out_cs, int act,
                                const float* y, int y_cs, const float* mean, const float* invstd, const float* coef,
                                int training, float* g_y, int gy_cs, float* g_res, int gres_cs, int64_t P, int C,
                                const float* mask_ss, const uint8_t* mask_bits) {
    if (!g_out || !coef || !g_y || P < 1 || C < 4 || (C & 3)) return ADH_E_ARG;
    if (training && (!y || !mean || !invstd)) return ADH_E_ARG;
    if (mask_ss && !(training && act == ADH_ACT_RELU)) return ADH_E_ARG;
    if (act == ADH_ACT_RELU && !out && !mask_ss && !mask_bits) return ADH_E_ARG;
    if (mask_bits && (((C / 4) & 1) || act != ADH_ACT_RELU)) return ADH_E_ARG;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(P, C / 4, EW_MAXBLK_BWD)), dim3(256), 0, (hipStream_t)stream, g_out, g_cs, out, out_cs, act,
                       y, y_cs, mean, invstd, coef, training, g_y, gy_cs, g_res, gres_cs, P, C, mask_ss, mask_bits);
    return adh_check_launch();
}
