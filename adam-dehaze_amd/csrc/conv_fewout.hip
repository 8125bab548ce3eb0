// 3x3 stride-1 pad-1 convolution with at most FOUR output channels (the reconstruction head of every branch: Conv2d(48 -> 3),
// /root/reference models/dehazing/high_intensity.py:78, medium_intensity.py, low_intensity.py), gfx950.
//
// The MFMA kernels pad the output channels to a 32-wide tile: 3 real channels of 32 is a tenth of the work they execute
// (Conv2d 48 -> 3 at 8 x 512 x 1024: 1.25 ms on conv_wino_kernel<1>, 1.18 ms direct, for 10.9 GFLOP and 0.87 GB of traffic).
// With four output channels per pixel the op is a per-pixel dot product: thread = one output pixel, its four accumulators in
// registers, the 10 x 34-pixel halo of an 8 x 32 tile in LDS (pixel pitch Cin + 4 floats: the 16 lanes of a ds_read_b128 group
// then hit 16 different bank quads), the weights [tap][channel quad][ci][co 4] read through the scalar cache (the address is
// wave-uniform: s_load_dwordx16 per tap and channel quad, the products are v_pk_fma_f32 with an SGPR-pair weight operand).
// Two workgroups per CU (71 KB of LDS each at 48 channels): one stages while the other computes.
// Epilogue: scale / shift (bias), ReLU; no residual, no statistics (the caller falls back to the general kernels for those).
#include "common.h"
#include <cstdlib>

#define FO_TH 8
#define FO_TW 32
#define FO_HH (FO_TH + 2)
#define FO_HW (FO_TW + 2)
#ifndef FO_CQ
#define FO_CQ 6      // channel quads per LDS chunk
#endif

__global__ __launch_bounds__(256, 4) void conv_fewout_fwd_kernel(const adh_conv_desc d, const f32x4* __restrict__ wp, int tiles_x,
                                                                 int tiles_y, int CQ) {
    extern __shared__ __attribute__((aligned(16))) float halo[];   // [FO_HH][FO_HW][channels of the chunk + 4]
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = ty * FO_TH, ox0 = tx * FO_TW;
    // the halo goes through LDS in chunks of at most FO_CQ channel quads (24 channels: 35 KB, four workgroups per CU)
    const int py = tid >> 5, px = tid & 31;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* in_n = d.in + (size_t)n * d.IH * d.IW * d.in_cstride;
    for (int c0 = 0; c0 < CQ; c0 += FO_CQ) {
        const int cqn = adh_min_i(FO_CQ, CQ - c0);
        const int pitch = cqn * 4 + 4;
        if (c0) __syncthreads();                                  // the previous chunk is consumed
        // ---- stage: one float4 (pixel, channel quad) per thread and trip, zeros outside the image
        const int nq = FO_HH * FO_HW * cqn;
        for (int i = tid; i < nq; i += 256) {
            const int cq = i % cqn, p = i / cqn;
            const int hy = p / FO_HW, hx = p - hy * FO_HW;
            const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW)
                v = *reinterpret_cast<const f32x4*>(in_n + ((size_t)iy * d.IW + ix) * d.in_cstride + (c0 + cq) * 4);
            *reinterpret_cast<f32x4*>(halo + p * pitch + cq * 4) = v;
        }
        __syncthreads();
        // ---- thread = output pixel (row tid / 32, column tid % 32)
        const float* hbase = halo + (py * FO_HW + px) * pitch;
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            const int dy = t / 3, dx = t - dy * 3;
            const float* hp = hbase + (dy * FO_HW + dx) * pitch;
            const f32x4* wt = wp + ((size_t)t * CQ + c0) * 4;     // wave-uniform: scalar loads
#pragma unroll 2
            for (int cq = 0; cq < cqn; ++cq) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(hp + cq * 4);
                acc += x[0] * wt[cq * 4 + 0];
                acc += x[1] * wt[cq * 4 + 1];
                acc += x[2] * wt[cq * 4 + 2];
                acc += x[3] * wt[cq * 4 + 3];
            }
        }
    }
    const int oy = oy0 + py, ox = ox0 + px;
    if (oy >= d.OH || ox >= d.OW) return;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (c < d.Cout) {
            if (d.scale) sc[c] = d.scale[c];
            if (d.shift) sh[c] = d.shift[c];
        }
    }
    f32x4 v = acc * sc + sh;
    if (d.act == ADH_ACT_RELU) v = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
    float* o = d.out + ((size_t)n * d.OH * d.OW + (size_t)oy * d.OW + ox) * d.out_cstride;
    if (d.Cout == 4 && (d.out_cstride & 3) == 0) {
        *reinterpret_cast<f32x4*>(o) = v;
    } else {   // (a narrower output may be a channel slice of a wider buffer: only its own channels are written)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < d.Cout) o[c] = v[c];
    }
}

static int fewout_plan(const adh_conv_desc* d) {
    static const bool enabled = !(getenv("ADH_FEWOUT") && getenv("ADH_FEWOUT")[0] == '0');   // A/B switch
    if (!enabled || !d) return 0;
    if (d->KH != 3 || d->KW != 3 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1) return 0;
    if (d->out_oy != 0 || d->out_ox != 0 || d->dy0 != -1 || d->dx0 != -1 || d->dstep_y != 1 || d->dstep_x != 1) return 0;
    if (d->Cout < 1 || d->Cout > 4 || d->Cin < 4 || d->Cin % 4 != 0 || d->Cin > 96 || d->in_cstride % 4 != 0) return 0;
    if (d->VH != d->OH || d->VW != d->OW || d->IH != d->OH || d->IW != d->OW) return 0;
    if (d->residual || d->stats) return 0;
    if (d->out_cstride < d->Cout) return 0;
    if (d->Cout == 4 && (d->out_cstride & 3) == 0 && ((uintptr_t)d->out & 15)) return 0;
    return 1;
}

extern "C" int adh_conv_fewout_supported(const adh_conv_desc* d) { return fewout_plan(d); }

extern "C" int adh_conv_fewout_forward(void* stream, const adh_conv_desc* d) {
    if (!fewout_plan(d)) return ADH_E_UNSUPPORTED;
    if (!d->in || !d->out || !d->wp || ((uintptr_t)d->in & 15) || ((uintptr_t)d->wp & 15)) return ADH_E_ARG;
    const int CQ = d->Cin / 4;
    const int tiles_x = adh_ceil_div(d->OW, FO_TW), tiles_y = adh_ceil_div(d->OH, FO_TH);
    const int lds = FO_HH * FO_HW * (adh_min_i(CQ, FO_CQ) * 4 + 4) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fewout_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(conv_fewout_fwd_kernel, dim3(tiles_x * tiles_y * d->N), dim3(256), lds, (hipStream_t)stream, *d,
                       reinterpret_cast<const f32x4*>(d->wp), tiles_x, tiles_y, CQ);
    return adh_check_launch();
}

// wp[tap][k / 4][k % 4][co 4] = W(co, k, tap) through the layout L (zeros for co >= L.Nc, k >= L.K); K8 = round_up(L.K, 8) channels,
// which is the d->Cin the forward call must carry (the engine's padded contraction length)
__global__ void pack_weights_fewout_kernel(const float* __restrict__ src, const adh_wlayout L, int K4, float* __restrict__ wp) {
    const int total = 9 * K4 * 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i & 3, k = (i >> 2) % K4, t = (i >> 2) / K4;
        const int ty = t / 3, tx = t - ty * 3;
        float v = 0.f;
        if (co < L.Nc && k < L.K)
            v = src[(int64_t)L.tap_off0 + ty * L.tap_off_sy + tx * L.tap_off_sx + (int64_t)k * L.stride_k + (int64_t)co * L.stride_n];
        wp[i] = v;
    }
}

extern "C" int adh_pack_weights_fewout(void* stream, const float* src, const adh_wlayout* L, float* wp) {
    if (!src || !L || !wp || L->K < 1 || L->Nc < 1 || L->Nc > 4 || L->KHt != 3 || L->KWt != 3) return ADH_E_ARG;
    const int K4 = adh_round_up(L->K, 8);
    hipLaunchKernelGGL(pack_weights_fewout_kernel, dim3(adh_min_i(adh_ceil_div(9 * K4 * 4, 256), 64)), dim3(256), 0, (hipStream_t)stream, src,
                       *L, K4, wp);
    return adh_check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// The mirror image: 3x3 s1 p1 with at most FOUR input channels (one channel quad) and 4 * COQ output channels -- the data
// gradient of the reconstruction head (3 -> 48) and the first layer of the guidance branch in eval mode (3 -> 16).  On the
// MFMA kernels K = 9 x 3 is padded to 9 x 8 and the tile to 32 output channels (Conv 3 -> 48 data gradient at full size:
// 0.71 ms, write-bound floor 0.16).  Thread = one output pixel, 4 * COQ accumulators, the 10 x 34 x 4-channel halo in LDS
// (5.4 KB), weights [tap][ci][co] through the scalar cache.  Epilogue: scale / shift / ReLU, BatchNorm partial statistics for
// Cout <= 16 (the train-mode first layer of the guidance branch); no residual.
// ---------------------------------------------------------------------------------------------------------------------
template <int COQ>
__global__ __launch_bounds__(256) void conv_fewin_fwd_kernel(const adh_conv_desc d, const f32x4* __restrict__ wp, int tiles_x,
                                                             int tiles_y) {
    __shared__ __attribute__((aligned(16))) float halo[FO_HH * FO_HW * 4];
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int oy0 = ty * FO_TH, ox0 = tx * FO_TW;
    const float* in_n = d.in + (size_t)n * d.IH * d.IW * d.in_cstride;
    for (int p = tid; p < FO_HH * FO_HW; p += 256) {
        const int hy = p / FO_HW, hx = p - hy * FO_HW;
        const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW) v = *reinterpret_cast<const f32x4*>(in_n + ((size_t)iy * d.IW + ix) * d.in_cstride);
        *reinterpret_cast<f32x4*>(halo + p * 4) = v;
    }
    __syncthreads();
    const int py = tid >> 5, px = tid & 31;
    f32x4 acc[COQ];
#pragma unroll
    for (int q = 0; q < COQ; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int t = 0; t < 9; ++t) {
        const int dy = t / 3, dx = t - dy * 3;
        const f32x4 x = *reinterpret_cast<const f32x4*>(halo + ((py + dy) * FO_HW + px + dx) * 4);
        const f32x4* wt = wp + (size_t)t * 4 * COQ;          // [ci 4][COQ] float4, wave-uniform
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
            for (int q = 0; q < COQ; ++q) acc[q] += x[ci] * wt[ci * COQ + q];
    }
    const int oy = oy0 + py, ox = ox0 + px;
    const bool inside = oy < d.OH && ox < d.OW;
    float* o = d.out + ((size_t)n * d.OH * d.OW + (size_t)(inside ? oy : 0) * d.OW + (inside ? ox : 0)) * d.out_cstride;
    // BatchNorm partial statistics (train-mode ConvBlock): row blockIdx.x of d.stats = [2][NcP] sums over this tile's pixels
    float* red = halo;                       // [4 waves][2][4 * COQ] after the halo is consumed
    if (d.stats) __syncthreads();
#pragma unroll
    for (int q = 0; q < COQ; ++q) {
        if (q * 4 < d.Cout) {               // Cout % 4 == 0 (fewin_plan)
            f32x4 v = acc[q];
            if (d.scale) v = v * *reinterpret_cast<const f32x4*>(d.scale + q * 4);
            if (d.shift) v = v + *reinterpret_cast<const f32x4*>(d.shift + q * 4);
            if (d.stats) {                  // statistics of the pre-activation output, as the other conv epilogues
                f32x4 s1 = inside ? v : f32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1 * s1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s1[e] = wave_sum(s1[e]);
                    s2[e] = wave_sum(s2[e]);
                }
                if ((tid & 63) == 0) {
                    *reinterpret_cast<f32x4*>(red + ((tid >> 6) * 2 + 0) * 4 * COQ + q * 4) = s1;
                    *reinterpret_cast<f32x4*>(red + ((tid >> 6) * 2 + 1) * 4 * COQ + q * 4) = s2;
                }
            }
            if (d.act == ADH_ACT_RELU) v = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
            if (inside) *reinterpret_cast<f32x4*>(o + q * 4) = v;
        }
    }
    if (d.stats) {
        __syncthreads();
        if (tid < 2 * 4 * COQ) {
            const int which = tid / (4 * COQ), c = tid - which * 4 * COQ;
            if (c < d.Cout) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) v += red[(w * 2 + which) * 4 * COQ + c];
                d.stats[((size_t)blockIdx.x * 2 + which) * d.NcP + c] = v;
            }
        }
    }
}

static int fewin_plan(const adh_conv_desc* d) {
    static const bool enabled = !(getenv("ADH_FEWOUT") && getenv("ADH_FEWOUT")[0] == '0');
    if (!enabled || !d) return 0;
    if (d->KH != 3 || d->KW != 3 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1) return 0;
    if (d->out_oy != 0 || d->out_ox != 0 || d->dy0 != -1 || d->dx0 != -1 || d->dstep_y != 1 || d->dstep_x != 1) return 0;
    if (d->Cin != 8 || d->in_cstride % 4 != 0 || d->in_cstride < 4) return 0;   // K <= 4 real channels in the first quad (the caller checks K)
    if (d->Cout < 4 || d->Cout > 64 || d->Cout % 4 != 0 || d->out_cstride % 4 != 0 || d->out_cstride < d->Cout) return 0;
    if (d->VH != d->OH || d->VW != d->OW || d->IH != d->OH || d->IW != d->OW) return 0;
    if (d->residual) return 0;
    if (d->stats && d->Cout > 16) return 0;     // (statistics: the 16-channel instantiation only -- wave reductions per channel)
    if (((uintptr_t)d->out & 15) || (d->scale && ((uintptr_t)d->scale & 15)) || (d->shift && ((uintptr_t)d->shift & 15))) return 0;
    return 1;
}

extern "C" int adh_conv_fewin_supported(const adh_conv_desc* d) { return fewin_plan(d); }
// rows of d->stats the launch writes (one per 8 x 32 tile)
extern "C" int adh_conv_fewin_num_blocks(const adh_conv_desc* d) {
    if (!fewin_plan(d)) return ADH_E_UNSUPPORTED;
    return adh_ceil_div(d->OW, FO_TW) * adh_ceil_div(d->OH, FO_TH) * d->N;
}

// number of output-channel quads the packed weights / the kernel instantiation carry for `cout` channels: 4, 12 or 16
static int fewin_coq(int cout) { return cout <= 16 ? 4 : (cout <= 48 ? 12 : 16); }

extern "C" int adh_conv_fewin_forward(void* stream, const adh_conv_desc* d) {
    if (!fewin_plan(d)) return ADH_E_UNSUPPORTED;
    if (!d->in || !d->out || !d->wp || ((uintptr_t)d->in & 15) || ((uintptr_t)d->wp & 15)) return ADH_E_ARG;
    const int tiles_x = adh_ceil_div(d->OW, FO_TW), tiles_y = adh_ceil_div(d->OH, FO_TH);
    const dim3 grid(tiles_x * tiles_y * d->N), block(256);
    const f32x4* wp = reinterpret_cast<const f32x4*>(d->wp);
    hipStream_t s = (hipStream_t)stream;
    switch (fewin_coq(d->Cout)) {
        case 4: hipLaunchKernelGGL(conv_fewin_fwd_kernel<4>, grid, block, 0, s, *d, wp, tiles_x, tiles_y); break;
        case 12: hipLaunchKernelGGL(conv_fewin_fwd_kernel<12>, grid, block, 0, s, *d, wp, tiles_x, tiles_y); break;
        default: hipLaunchKernelGGL(conv_fewin_fwd_kernel<16>, grid, block, 0, s, *d, wp, tiles_x, tiles_y); break;
    }
    return adh_check_launch();
}

// wp[tap][ci 4][COQ * 4] = W(co, ci, tap) through the layout L (zeros for ci >= L.K, co >= L.Nc), COQ = fewin_coq(L.Nc)
__global__ void pack_weights_fewin_kernel(const float* __restrict__ src, const adh_wlayout L, int CO, float* __restrict__ wp) {
    const int total = 9 * 4 * CO;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = i % CO, ci = (i / CO) & 3, t = i / (4 * CO);
        const int ty = t / 3, tx = t - ty * 3;
        float v = 0.f;
        if (co < L.Nc && ci < L.K)
            v = src[(int64_t)L.tap_off0 + ty * L.tap_off_sy + tx * L.tap_off_sx + (int64_t)ci * L.stride_k + (int64_t)co * L.stride_n];
        wp[i] = v;
    }
}

extern "C" int adh_pack_weights_fewin(void* stream, const float* src, const adh_wlayout* L, float* wp) {
    if (!src || !L || !wp || L->K < 1 || L->K > 4 || L->Nc < 1 || L->Nc > 64 || L->KHt != 3 || L->KWt != 3) return ADH_E_ARG;
    const int CO = fewin_coq(L->Nc) * 4;
    hipLaunchKernelGGL(pack_weights_fewin_kernel, dim3(adh_min_i(adh_ceil_div(9 * 4 * CO, 256), 64)), dim3(256), 0, (hipStream_t)stream, src, *L,
                       CO, wp);
    return adh_check_launch();
}
