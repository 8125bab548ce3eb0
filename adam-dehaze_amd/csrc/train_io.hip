// Training-loop and evaluation-harness kernels around the hot path, gfx950: multi-tensor Adam (one launch per step),
// the reference's synthetic fog model on device, and per-image PSNR / SSIM without per-image D2H copies.
// All HBM-bound streaming / stencil kernels; reductions are two-stage with a fixed order (deterministic, no atomics).
// Reference lines cited per kernel (paths under /root/reference).
#include "common.h"

// ------------------------------------------------------------------------------------------------
// multi-tensor Adam (training/train_joint.py:86-90, training/train_dehazing.py:52-57)
// ------------------------------------------------------------------------------------------------
#define ADAM_CHUNK 16384   // floats per workgroup

// one update of `repeats` listed copies of a tensor; dup_mode 0: the single-tensor loop of torch.optim.Adam (CPU default,
// every torch < 2.0): `repeats` consecutive full updates; dup_mode 1: torch >= 2.0's foreach form on CUDA, where the
// duplicate entries of a parameter list alias each other inside the _foreach_ calls (weight decay from the original p,
// m lerped `repeats` times, v scaled by beta2^repeats then incremented `repeats` times, one bias correction at
// step + repeats, `repeats` identical parameter updates).
__device__ __forceinline__ void adam_elem(float& pv, float gv0, float& mv, float& vv, int step, int repeats, int dup_mode,
                                          float lr, float beta1, float beta2, float eps, float wd, const float* bc1s,
                                          const float* bc2s) {
    if (dup_mode == 0 || repeats == 1) {
        for (int r = 0; r < repeats; ++r) {
            const float gv = gv0 + wd * pv;
            mv = beta1 * mv + (1.f - beta1) * gv;
            vv = beta2 * vv + (1.f - beta2) * gv * gv;
            const float denom = sqrtf(vv) / bc2s[r] + eps;
            pv -= (lr / bc1s[r]) * (mv / denom);
        }
    } else {
        const float gv = gv0 + wd * pv;
        for (int r = 0; r < repeats; ++r) mv = mv + (gv - mv) * (1.f - beta1);
        for (int r = 0; r < repeats; ++r) vv *= beta2;
        for (int r = 0; r < repeats; ++r) vv += (1.f - beta2) * gv * gv;
        const float denom = sqrtf(vv) / bc2s[repeats - 1] + eps;
        const float upd = (lr / bc1s[repeats - 1]) * (mv / denom);
        for (int r = 0; r < repeats; ++r) pv -= upd;
    }
}

#define ADAM_MAX_REPEATS 4

__global__ __launch_bounds__(256) void adam_multi_kernel(const adh_adam_tensor* __restrict__ table,
                                                         const int32_t* __restrict__ chunks, float lr, float beta1,
                                                         float beta2, float eps, float wd, float gscale, int dup_mode,
                                                         int calls_since_upload) {
    const int ti = chunks[2 * blockIdx.x], ci = chunks[2 * blockIdx.x + 1];
    adh_adam_tensor t = table[ti];
    t.step += calls_since_upload * t.repeats;   // the table stays resident: every call advances a tensor by `repeats`
    const int64_t base = (int64_t)ci * ADAM_CHUNK;
    const int n = (int)((t.n - base) < ADAM_CHUNK ? (t.n - base) : ADAM_CHUNK);
    float* __restrict__ p = t.p + base;
    const float* __restrict__ g = t.g + base;
    float* __restrict__ m = t.m + base;
    float* __restrict__ v = t.v + base;
    // bias corrections of the steps this launch takes (uniform per workgroup)
    float bc1s[ADAM_MAX_REPEATS], bc2s[ADAM_MAX_REPEATS];
#pragma unroll
    for (int r = 0; r < ADAM_MAX_REPEATS; ++r) {
        const float st = (float)(t.step + r + 1);
        bc1s[r] = 1.f - powf(beta1, st);
        bc2s[r] = sqrtf(1.f - powf(beta2, st));
    }
    const bool vec = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0;
    const int n4 = vec ? (n >> 2) : 0;
    for (int i = threadIdx.x; i < n4; i += 256) {
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i], mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i] * gscale;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a = pv[j], b = mv[j], c = vv[j];
            adam_elem(a, gv[j], b, c, t.step, t.repeats, dup_mode, lr, beta1, beta2, eps, wd, bc1s, bc2s);
            pv[j] = a; mv[j] = b; vv[j] = c;
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    for (int i = n4 * 4 + threadIdx.x; i < n; i += 256) {
        float a = p[i], b = m[i], c = v[i];
        adam_elem(a, g[i] * gscale, b, c, t.step, t.repeats, dup_mode, lr, beta1, beta2, eps, wd, bc1s, bc2s);
        p[i] = a; m[i] = b; v[i] = c;
    }
}

extern "C" int adh_adam_chunk_elems(void) { return ADAM_CHUNK; }

extern "C" int adh_adam_multi(void* stream, const adh_adam_tensor* table_dev, const int32_t* chunks_dev, int nchunks, float lr,
                              float beta1, float beta2, float eps, float weight_decay, float grad_scale, int dup_mode,
                              int max_repeats, int calls_since_upload) {
    if (!table_dev || !chunks_dev || nchunks < 1 || max_repeats < 1 || max_repeats > ADAM_MAX_REPEATS ||
        (dup_mode != 0 && dup_mode != 1) || calls_since_upload < 0)
        return ADH_E_ARG;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table_dev, chunks_dev, lr, beta1,
                       beta2, eps, weight_decay, grad_scale, dup_mode, calls_since_upload);
    return adh_check_launch();
}

// ------------------------------------------------------------------------------------------------
// synthetic fog  I = J*t + A*(1-t),  t = exp(-beta * depth)   (utils/helpers.py:241-258)
// depth = 0.3 + 0.7*sqrt((x-0.5)^2 + (y-0.2)^2) on np.linspace(0,1,W) x np.linspace(0,1,H); the reference evaluates the
// transmission in float64 (numpy) and stores float32: so does this kernel (one exp per pixel, shared by the 3 channels)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fog_kernel(const float* __restrict__ clear, const float* __restrict__ beta,
                                                  const float* __restrict__ airlight, int H, int W, float* __restrict__ hazy) {
    const int n = blockIdx.y;
    const int64_t HW = (int64_t)H * W;
    const double b = (double)beta[n], A = (double)airlight[n];
    const double sx = W > 1 ? 1.0 / (double)(W - 1) : 0.0, sy = H > 1 ? 1.0 / (double)(H - 1) : 0.0;
    const float* src = clear + (int64_t)n * 3 * HW;
    float* dst = hazy + (int64_t)n * 3 * HW;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
        const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
        const double dx = x * sx - 0.5, dy = y * sy - 0.2;
        const double t = exp(-b * (0.3 + 0.7 * sqrt(dx * dx + dy * dy)));
        const double a = A * (1.0 - t);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float h = (float)((double)src[c * HW + i] * t + a);
            dst[c * HW + i] = fminf(fmaxf(h, 0.f), 1.f);
        }
    }
}

extern "C" int adh_apply_fog(void* stream, const float* clear_nchw, const float* beta, const float* airlight, int N, int H,
                             int W, float* hazy_nchw) {
    if (!clear_nchw || !beta || !airlight || !hazy_nchw || N < 1 || H < 1 || W < 1 || N > 65535) return ADH_E_ARG;
    const int64_t HW = (int64_t)H * W;
    hipLaunchKernelGGL(fog_kernel, dim3(adh_min_i(adh_ceil_div(HW, 256), 2048), N), dim3(256), 0, (hipStream_t)stream,
                       clear_nchw, beta, airlight, H, W, hazy_nchw);
    return adh_check_launch();
}

// ------------------------------------------------------------------------------------------------
// PSNR per image (evaluation/metrics.py:27, training/train_joint.py:217: skimage peak_signal_noise_ratio(target, pred,
// data_range=1.0) = 10 log10(1 / mean((t-p)^2)), the mean taken in float64 over all 3*H*W elements)
// ------------------------------------------------------------------------------------------------
#define PSNR_ELEMS_PER_BLOCK (256 * 32)

extern "C" int adh_psnr_num_blocks(int64_t per_image) {
    return adh_max_i(1, adh_min_i(adh_ceil_div(per_image, PSNR_ELEMS_PER_BLOCK), 1024));
}

__global__ __launch_bounds__(256) void sqerr_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            int64_t per, double* __restrict__ partial) {
    const int n = blockIdx.y;
    const float* pa = a + (int64_t)n * per;
    const float* pb = b + (int64_t)n * per;
    double acc = 0.0;
    const bool vec = ((per & 3) == 0) && (((((uintptr_t)pa) | ((uintptr_t)pb)) & 15) == 0);
    if (vec) {
        const int64_t n4 = per >> 2;
        for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
            const f32x4 d = reinterpret_cast<const f32x4*>(pa)[i] - reinterpret_cast<const f32x4*>(pb)[i];
            acc += (double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2] + (double)d[3] * d[3];
        }
    } else {
        for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256) {
            const float d = pa[i] - pb[i];
            acc += (double)d * d;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)n * gridDim.x + blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ void psnr_finalize_kernel(const double* __restrict__ partial, int nblk, double inv_count, float data_range,
                                     float* __restrict__ mse, float* __restrict__ psnr) {
    const int n = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) acc += partial[(int64_t)n * nblk + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (threadIdx.x == 0) {
        const double e = acc * inv_count;
        if (mse) mse[n] = (float)e;
        // skimage returns +inf for identical images (division by zero warning); so do we
        psnr[n] = (float)(10.0 * log10((double)data_range * data_range / e));
    }
}

extern "C" int adh_psnr(void* stream, const float* pred, const float* target, int N, int64_t per_image, float data_range,
                        double* partial, int nblk, float* mse, float* psnr) {
    if (!pred || !target || !partial || !psnr || N < 1 || N > 65535 || per_image < 1 || nblk != adh_psnr_num_blocks(per_image))
        return ADH_E_ARG;
    hipLaunchKernelGGL(sqerr_partial_kernel, dim3(nblk, N), dim3(256), 0, (hipStream_t)stream, pred, target, per_image, partial);
    hipLaunchKernelGGL(psnr_finalize_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, partial, nblk, 1.0 / (double)per_image,
                       data_range, mse, psnr);
    return adh_check_launch();
}

// ------------------------------------------------------------------------------------------------
// SSIM per image (evaluation/metrics.py:29-32, training/train_joint.py:221-224): skimage structural_similarity on the
// channel-mean grayscale images with its defaults -- 7x7 uniform window, K1 0.01, K2 0.03, sample covariance
// (NP/(NP-1)), data_range 1 -- and the mean of S over the image cropped by (win-1)/2 = 3 pixels on every side.  The
// cropped windows never touch the border, so the filter's boundary mode does not enter.
// One workgroup = 32x32 outputs; the 38x38 grayscale halos of both images are staged in LDS, every thread sums its four
// 7x7 windows in float64 (MI355X fp64 vector rate is not the limit here; the kernel reads 24 B per pixel once).
// ------------------------------------------------------------------------------------------------
#define SSIM_T 32
#define SSIM_WIN 7
#define SSIM_HALO (SSIM_T + SSIM_WIN - 1)

extern "C" int adh_ssim_num_blocks(int H, int W) {
    if (H < SSIM_WIN || W < SSIM_WIN) return 0;
    return adh_ceil_div(H - SSIM_WIN + 1, SSIM_T) * adh_ceil_div(W - SSIM_WIN + 1, SSIM_T);
}

__global__ __launch_bounds__(256) void ssim_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                           int H, int W, float data_range, double* __restrict__ partial) {
    __shared__ float gp[SSIM_HALO][SSIM_HALO + 1];
    __shared__ float gt[SSIM_HALO][SSIM_HALO + 1];
    const int n = blockIdx.y;
    const int OW = W - SSIM_WIN + 1, OH = H - SSIM_WIN + 1;
    const int tiles_x = (OW + SSIM_T - 1) / SSIM_T;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * SSIM_T, x0 = tx * SSIM_T;   // top-left of the halo in image coordinates
    const int64_t HW = (int64_t)H * W;
    const float* pp = pred + (int64_t)n * 3 * HW;
    const float* pt = target + (int64_t)n * 3 * HW;
    for (int i = threadIdx.x; i < SSIM_HALO * SSIM_HALO; i += 256) {
        const int r = i / SSIM_HALO, c = i - r * SSIM_HALO;
        const int y = y0 + r, x = x0 + c;
        float a = 0.f, b = 0.f;
        if (y < H && x < W) {
            const int64_t o = (int64_t)y * W + x;
            // np.mean(img, axis=2) on float32: (c0 + c1 + c2) / 3 in float32
            a = ((pp[o] + pp[HW + o]) + pp[2 * HW + o]) / 3.0f;
            b = ((pt[o] + pt[HW + o]) + pt[2 * HW + o]) / 3.0f;
        }
        gp[r][c] = a;
        gt[r][c] = b;
    }
    __syncthreads();
    const double C1 = (0.01 * data_range) * (0.01 * data_range), C2 = (0.03 * data_range) * (0.03 * data_range);
    const double NP = SSIM_WIN * SSIM_WIN, cov_norm = NP / (NP - 1.0), inv = 1.0 / NP;
    double acc = 0.0;
    const int lx = threadIdx.x & 31, ly0 = threadIdx.x >> 5;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        const int ly = ly0 + 8 * k;
        if (y0 + ly >= OH || x0 + lx >= OW) continue;
        double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
        for (int dy = 0; dy < SSIM_WIN; ++dy)
#pragma unroll
            for (int dx = 0; dx < SSIM_WIN; ++dx) {
                const double a = gt[ly + dy][lx + dx], b = gp[ly + dy][lx + dx];   // im1 = target, im2 = pred
                sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
            }
        const double ux = sx * inv, uy = sy * inv;
        const double vx = cov_norm * (sxx * inv - ux * ux), vy = cov_norm * (syy * inv - uy * uy);
        const double vxy = cov_norm * (sxy * inv - ux * uy);
        const double A1 = 2 * ux * uy + C1, A2 = 2 * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
        acc += (A1 * A2) / (B1 * B2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)n * gridDim.x + blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ void ssim_finalize_kernel(const double* __restrict__ partial, int nblk, double inv_count, float* __restrict__ ssim) {
    const int n = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 64) acc += partial[(int64_t)n * nblk + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (threadIdx.x == 0) ssim[n] = (float)(acc * inv_count);
}

extern "C" int adh_ssim_gray(void* stream, const float* pred_nchw, const float* target_nchw, int N, int H, int W,
                             float data_range, double* partial, int nblk, float* ssim) {
    if (!pred_nchw || !target_nchw || !partial || !ssim || N < 1 || N > 65535) return ADH_E_ARG;
    if (H < SSIM_WIN || W < SSIM_WIN) return ADH_E_UNSUPPORTED;   // skimage raises: win_size exceeds image extent
    if (nblk != adh_ssim_num_blocks(H, W)) return ADH_E_ARG;
    hipLaunchKernelGGL(ssim_partial_kernel, dim3(nblk, N), dim3(256), 0, (hipStream_t)stream, pred_nchw, target_nchw, H, W,
                       data_range, partial);
    hipLaunchKernelGGL(ssim_finalize_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, partial, nblk,
                       1.0 / ((double)(H - SSIM_WIN + 1) * (double)(W - SSIM_WIN + 1)), ssim);
    return adh_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Paired augmentation (/root/reference data/dataset.py:59-64,100-116): RandomHorizontalFlip, RandomVerticalFlip and
// ColorJitter(brightness 0.1, contrast 0.1) applied to hazy / clear / dehazed with ONE seed per sample, so the three images
// get the same flips and the same jitter factors.  torchvision semantics for float tensors: brightness = clamp(b * x, 0, 1),
// contrast = clamp(c * x + (1 - c) * mean(gray(x)), 0, 1) with gray = 0.2989 R + 0.587 G + 0.114 B over the whole image,
// in the order the sample's permutation puts them.  params[n] = {flip_h, flip_v, brightness_first, b, c} (host draws).
// Two launches: the grayscale mean of the image the contrast step sees (fp64 two-stage sum), then one gather pass.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void aug_gray_partial_kernel(const float* __restrict__ x, const float* __restrict__ params, int64_t HW,
                                                               double* __restrict__ partial) {
    const int n = blockIdx.y;
    const float* p = params + n * 5;
    const float b = p[2] != 0.f ? p[3] : 1.f;            // brightness comes first: the contrast step sees clamp(b * x)
    const float* px = x + (int64_t)n * 3 * HW;
    double acc = 0.0;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
        const float r = fminf(fmaxf(b * px[i], 0.f), 1.f), g = fminf(fmaxf(b * px[HW + i], 0.f), 1.f),
                    bl = fminf(fmaxf(b * px[2 * HW + i], 0.f), 1.f);
        acc += (double)(0.2989f * r + 0.587f * g + 0.114f * bl);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)n * gridDim.x + blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

__global__ __launch_bounds__(256) void aug_apply_kernel(const float* __restrict__ x, const float* __restrict__ params,
                                                        const double* __restrict__ partial, int nblk, int H, int W,
                                                        float* __restrict__ out) {
    const int n = blockIdx.y;
    const int64_t HW = (int64_t)H * W;
    const float* p = params + n * 5;
    const bool fh = p[0] != 0.f, fv = p[1] != 0.f, bfirst = p[2] != 0.f;
    const float b = p[3], c = p[4];
    double m = 0.0;
    for (int i = 0; i < nblk; ++i) m += partial[(int64_t)n * nblk + i];     // same order in every thread: deterministic
    const float mean = (float)(m / (double)HW);
    const float* src = x + (int64_t)n * 3 * HW;
    float* dst = out + (int64_t)n * 3 * HW;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
        const int y = (int)(i / W), xx = (int)(i - (int64_t)y * W);
        const int64_t j = (int64_t)(fv ? H - 1 - y : y) * W + (fh ? W - 1 - xx : xx);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float v = src[ch * HW + j];
            if (bfirst) {
                v = fminf(fmaxf(b * v, 0.f), 1.f);
                v = fminf(fmaxf(c * v + (1.f - c) * mean, 0.f), 1.f);
            } else {
                v = fminf(fmaxf(c * v + (1.f - c) * mean, 0.f), 1.f);
                v = fminf(fmaxf(b * v, 0.f), 1.f);
            }
            dst[ch * HW + i] = v;
        }
    }
}

extern "C" int adh_augment_num_blocks(int64_t HW) { return adh_max_i(1, adh_min_i(adh_ceil_div(HW, 256 * 16), 256)); }

extern "C" int adh_paired_augment(void* stream, const float* x_nchw, const float* params, int N, int H, int W, double* partial,
                                  int nblk, float* out_nchw) {
    if (!x_nchw || !params || !partial || !out_nchw || N < 1 || N > 65535 || H < 1 || W < 1) return ADH_E_ARG;
    const int64_t HW = (int64_t)H * W;
    if (nblk != adh_augment_num_blocks(HW) || x_nchw == out_nchw) return ADH_E_ARG;
    hipLaunchKernelGGL(aug_gray_partial_kernel, dim3(nblk, N), dim3(256), 0, (hipStream_t)stream, x_nchw, params, HW, partial);
    hipLaunchKernelGGL(aug_apply_kernel, dim3(adh_min_i(adh_ceil_div(HW, 256), 2048), N), dim3(256), 0, (hipStream_t)stream, x_nchw,
                       params, partial, nblk, H, W, out_nchw);
    return adh_check_launch();
}
