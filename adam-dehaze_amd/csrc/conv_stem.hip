// The 7x7 stride-1 pad-3 stem convolution of the Medium / Complex models (3 -> 64 / 96 channels on the NHWC8 image) and its
// weight gradient, fp32 MFMA 16x16x4, gfx950.  Replaces ATen's conv2d / conv2d-backward-weight for init_conv
// (/root/reference models/dehazing/medium_intensity.py:20, high_intensity.py:24 -- ConvBlock(3, base, 7, 1, 3)).
//
// With 3 real channels in an 8-channel pixel the general kernels spend 5/8 of every MFMA on zero padding (and 32-row
// tiles another factor).  Here the MFMA index that runs over the filter is packed: (kx, c) with c < 3, so one filter
// row is 21 entries -- the per-lane LDS offset (kx * 8 + c) does the gather.
//
//   weight gradient   D[(kx,c) 16 rows][co 16] += A[(kx,c)][4 px] B[4 px][co]      rows: kx 0..4 (15 rows) | kx 5..6
//                     14 row tiles (7 ky x 2) x one 16-channel column tile per wave; 6 (4) waves = 96 (64) channels
//                     A from the LDS image halo, B (the output gradient) straight from global memory -- every element
//                     is used by exactly one wave -- prefetched four pixel groups ahead
//   forward           D[px 16][co 16] += A[px][(kx,c) 4 of 24] B[(kx,c)][co]       6 k-steps per ky, 42 per 16 px x 16 co
//                     A from the LDS halo (per-lane offsets of the 6 k-steps), B = packed weights [ky][24][co] in LDS
// Output of the weight gradient: one slab [49][8][NcP] per workgroup; adh_wgrad_reduce_small adds them.
#include "common.h"
#include <cstdlib>

typedef float f32x4t __attribute__((ext_vector_type(4)));

#define ST_HC 76          // halo columns kept per row: 70 needed (x0-3 .. x0+66) + over-read of the padded (kx, c) rows

// ------------------------------------------------------------------------------------------------ weight gradient
template <int NW>   // waves = Cout / 16
__global__ __launch_bounds__(NW * 64) void conv_wgrad_stem_kernel(const adh_conv_desc d, int tiles_x, int tiles_y, int ntiles,
                                                                  float* __restrict__ slab, int NcP) {
    __shared__ __attribute__((aligned(16))) float xs[10 * ST_HC * 8 + 128];   // image halo [10][ST_HC][8] (+ over-read pad)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i16 = lane & 15, kk = lane >> 4;
    const int rowoff = (i16 / 3) * 8 + i16 % 3;      // (kx, c) = (i16 / 3, i16 % 3): LDS offset inside a pixel row
    const int gcs = d.out_cstride;

    f32x4t acc[14];
#pragma unroll
    for (int t = 0; t < 14; ++t) acc[t] = f32x4t{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int r = tile;
        const int tx = r % tiles_x;
        r /= tiles_x;
        const int ty = r % tiles_y;
        const int n = r / tiles_y;
        const int y0 = ty * 4, x0 = tx * 64;
        const float* xin = d.in + (int64_t)n * d.IH * d.IW * 8;
        const float* gin = d.out + (int64_t)n * d.VH * d.VW * gcs + wave * 16 + i16;
        __syncthreads();
        for (int i = tid; i < 10 * ST_HC * 2; i += NW * 64) {
            const int px = i >> 1, q = i & 1;
            const int row = px / ST_HC, col = px - row * ST_HC;
            const int iy = y0 - 3 + row, ix = x0 - 3 + col;
            f32x4t v = {0.f, 0.f, 0.f, 0.f};
            if (iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW) v = *reinterpret_cast<const f32x4t*>(xin + ((int64_t)iy * d.IW + ix) * 8 + q * 4);
            *reinterpret_cast<f32x4t*>(xs + px * 8 + q * 4) = v;
        }
        if (tid < 128) xs[10 * ST_HC * 8 + tid] = 0.f;
        __syncthreads();
        // 64 groups of 4 pixels (4 rows x 16); B = g[(y0 + row)][x0 + 4 * grp + kk][16 * wave + i16], prefetched 4 ahead
        auto gload = [&](int grp) {
            const int row = grp >> 4, col = (grp & 15) * 4 + kk;
            const int gy = y0 + row, gx = x0 + col;
            return (gy < d.VH && gx < d.VW) ? gin[((int64_t)gy * d.VW + gx) * gcs] : 0.f;
        };
        float bq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bq[u] = gload(u);
#pragma unroll 1
        for (int g4 = 0; g4 < 64; g4 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int grp = g4 + u;
                const float b = bq[u];
                bq[u] = grp + 4 < 64 ? gload(grp + 4) : 0.f;
                const int row = grp >> 4, col = (grp & 15) * 4 + kk;
                const float* xl = xs + (row * ST_HC + col) * 8 + rowoff;
#pragma unroll
                for (int ky = 0; ky < 7; ++ky) {
                    acc[ky * 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(xl[ky * ST_HC * 8], b, acc[ky * 2], 0, 0, 0);
                    acc[ky * 2 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(xl[ky * ST_HC * 8 + 40], b, acc[ky * 2 + 1], 0, 0, 0);
                }
            }
        }
    }
    // slab[blockIdx][tap = ky * 7 + kx][c < 8][co]: rows (kx, c) of the two tiles of each ky
    float* const sl = slab + (int64_t)blockIdx.x * 49 * 8 * NcP;
    const int co = wave * 16 + i16;
#pragma unroll
    for (int t = 0; t < 14; ++t) {
        const int ky = t >> 1, tl = t & 1;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int row = 4 * kk + rr;                   // D layout: row = 4 * (lane / 16) + reg
            const int kx = tl * 5 + row / 3, c = row % 3;
            if (row < 15 && kx < 7) sl[((int64_t)(ky * 7 + kx) * 8 + c) * NcP + co] = acc[t][rr];
        }
    }
}

static int stem_ok(const adh_conv_desc* d) {
    if (!d) return 0;
    if (d->KH != 7 || d->KW != 7 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1) return 0;
    if (d->out_oy != 0 || d->out_ox != 0 || d->dy0 != -3 || d->dx0 != -3 || d->dstep_y != 1 || d->dstep_x != 1) return 0;
    if (d->VH != d->IH || d->VW != d->IW || d->in_cstride != 8 || d->Cin > 8) return 0;   // (callers: at most 3 real channels)
    if (d->Cout != 96 && d->Cout != 64) return 0;
    if (d->out_cstride < d->Cout) return 0;
    return 1;
}

// number of slabs [49][8][NcP] adh_conv_wgrad_stem writes for `d` (d->in = NHWC8 image, d->out = dL/dy); 0 = not the stem
extern "C" int adh_conv_wgrad_stem_slabs(const adh_conv_desc* d) {
    if (!stem_ok(d)) return 0;
    const int64_t ntiles = (int64_t)d->N * adh_ceil_div(d->VH, 4) * adh_ceil_div(d->VW, 64);
    return (int)(ntiles < 512 ? ntiles : 512);
}

extern "C" int adh_conv_wgrad_stem(void* stream, const adh_conv_desc* d, float* slab, int NcP) {
    if (!stem_ok(d)) return ADH_E_UNSUPPORTED;
    if (!slab || !d->in || !d->out || NcP < d->Cout) return ADH_E_ARG;
    if ((uintptr_t)d->in & 15) return ADH_E_ARG;
    const int tiles_x = adh_ceil_div(d->VW, 64), tiles_y = adh_ceil_div(d->VH, 4);
    const int ntiles = d->N * tiles_x * tiles_y;
    const int nwg = adh_conv_wgrad_stem_slabs(d);
    hipStream_t s = (hipStream_t)stream;
    if (d->Cout == 96)
        hipLaunchKernelGGL((conv_wgrad_stem_kernel<6>), dim3(nwg), dim3(384), 0, s, *d, tiles_x, tiles_y, ntiles, slab, NcP);
    else
        hipLaunchKernelGGL((conv_wgrad_stem_kernel<4>), dim3(nwg), dim3(256), 0, s, *d, tiles_x, tiles_y, ntiles, slab, NcP);
    return adh_check_launch();
}

// ------------------------------------------------------------------------------------------------ forward
// weights packed [ky 7][e 24 = (kx, c) row, 21 real][NcP]  (adh_pack_weights_stem)
#define ST_WROWS (7 * 24)

template <int NT>   // 16-channel tiles: Cout = 16 * NT
__global__ __launch_bounds__(256) void conv_stem_fwd_kernel(const adh_conv_desc d, int tiles_x, int tiles_y, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // xs [10][ST_HC][8] + pad | ws [7 * 24][16 * NT] | red
    constexpr int XS_F = 10 * ST_HC * 8 + 128;
    constexpr int CO = 16 * NT;
    float* const xs = lds;
    float* const ws = lds + XS_F;
    float* const red = ws + ST_WROWS * CO;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i16 = lane & 15, kk = lane >> 4;

    for (int i = tid; i < ST_WROWS * CO / 4; i += 256) {
        const int row = i / (CO / 4), q = i - row * (CO / 4);
        *reinterpret_cast<f32x4t*>(ws + row * CO + q * 4) = *reinterpret_cast<const f32x4t*>(d.wp + (int64_t)row * d.NcP + q * 4);
    }
    // B gather offsets of the six k-steps: entry e = 4 * step + kk -> pixel kx = e / 3, channel c = e % 3
    int eoff[6];
#pragma unroll
    for (int st = 0; st < 6; ++st) {
        const int e = 4 * st + kk;
        eoff[st] = (e / 3) * 8 + e % 3;
    }
    // per-lane epilogue constants: channels cot * 16 + 4 * kk .. + 3
    f32x4t sc[NT], sh[NT], ssum[NT], ssq[NT];
#pragma unroll
    for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = ct * 16 + 4 * kk + e;
            sc[ct][e] = (d.scale && co < d.Cout) ? d.scale[co] : 1.f;
            sh[ct][e] = (d.shift && co < d.Cout) ? d.shift[co] : 0.f;
        }
        ssum[ct] = f32x4t{0.f, 0.f, 0.f, 0.f};
        ssq[ct] = f32x4t{0.f, 0.f, 0.f, 0.f};
    }
    const bool vec = (d.out_cstride & 3) == 0 && ((uintptr_t)d.out & 15) == 0;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int r = tile;
        const int tx = r % tiles_x;
        r /= tiles_x;
        const int ty = r % tiles_y;
        const int n = r / tiles_y;
        const int y0 = ty * 4, x0 = tx * 64;
        const float* xin = d.in + (int64_t)n * d.IH * d.IW * 8;
        float* out_n = d.out + (int64_t)n * d.OH * d.OW * d.out_cstride;
        __syncthreads();
        for (int i = tid; i < 10 * ST_HC * 2; i += 256) {
            const int px = i >> 1, q = i & 1;
            const int row = px / ST_HC, col = px - row * ST_HC;
            const int iy = y0 - 3 + row, ix = x0 - 3 + col;
            f32x4t v = {0.f, 0.f, 0.f, 0.f};
            if (iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW) v = *reinterpret_cast<const f32x4t*>(xin + ((int64_t)iy * d.IW + ix) * 8 + q * 4);
            *reinterpret_cast<f32x4t*>(xs + px * 8 + q * 4) = v;
        }
        if (tid < 128) xs[10 * ST_HC * 8 + tid] = 0.f;
        __syncthreads();
        // wave w: output row y0 + w, two halves of 32 pixels (two 16-pixel column tiles share every weight read)
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            f32x4t acc[NT][2];
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) acc[ct][0] = acc[ct][1] = f32x4t{0.f, 0.f, 0.f, 0.f};
            const float* xl = xs + (wave * ST_HC + half * 32 + i16) * 8;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                __builtin_amdgcn_sched_barrier(0);   // one filter row of operands in flight, not seven (registers)
#pragma unroll
                for (int st = 0; st < 6; ++st) {
                    const float b0 = xl[ky * ST_HC * 8 + eoff[st]], b1 = xl[ky * ST_HC * 8 + 16 * 8 + eoff[st]];
                    const float* wl = ws + (ky * 24 + 4 * st + kk) * CO + i16;
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) {
                        const float a = wl[ct * 16];
                        acc[ct][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b0, acc[ct][0], 0, 0, 0);
                        acc[ct][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b1, acc[ct][1], 0, 0, 0);
                    }
                }
            }
            // D[co = 16 ct + 4 kk + reg][px = i16]: four consecutive channels of one pixel per lane
            const int oy = y0 + wave;
#pragma unroll
            for (int pg = 0; pg < 2; ++pg) {
                const int ox = x0 + half * 32 + pg * 16 + i16;
                if (oy < d.OH && ox < d.OW) {
                    float* po = out_n + ((int64_t)oy * d.OW + ox) * d.out_cstride + 4 * kk;
#pragma unroll
                    for (int ct = 0; ct < NT; ++ct) {
                        f32x4t v = acc[ct][pg] * sc[ct] + sh[ct];
                        ssum[ct] += v;
                        ssq[ct] += v * v;
                        if (d.act == ADH_ACT_RELU) v = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                        if (vec && ct * 16 + 4 * kk + 3 < d.Cout) {
                            *reinterpret_cast<f32x4t*>(po + ct * 16) = v;
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (ct * 16 + 4 * kk + e < d.Cout) po[ct * 16 + e] = v[e];
                        }
                    }
                }
            }
        }
    }
    if (d.stats) {
        // per channel: sum over the 16 pixel lanes (i16), then over the 4 waves through LDS; one row pair per workgroup
#pragma unroll
        for (int ct = 0; ct < NT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    ssum[ct][e] += __shfl_xor(ssum[ct][e], o, 64);
                    ssq[ct][e] += __shfl_xor(ssq[ct][e], o, 64);
                }
        __syncthreads();
        if (i16 == 0) {
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) {
                *reinterpret_cast<f32x4t*>(red + (0 * 4 + wave) * CO + ct * 16 + 4 * kk) = ssum[ct];
                *reinterpret_cast<f32x4t*>(red + (1 * 4 + wave) * CO + ct * 16 + 4 * kk) = ssq[ct];
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * CO; i += 256) {
            const int which = i / CO, co = i - which * CO;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[(which * 4 + w) * CO + co];
            d.stats[((size_t)blockIdx.x * 2 + which) * d.NcP + co] = v;
        }
    }
}

static int stem_fwd_ok(const adh_conv_desc* d) {
    static const bool enabled = !(getenv("ADH_STEM_FWD") && getenv("ADH_STEM_FWD")[0] == '0');   // A/B switch
    if (!enabled || !stem_ok(d)) return 0;
    if (d->residual || d->NcP < d->Cout || (d->NcP & 3)) return 0;
    return 1;
}

// workgroups (= BatchNorm partial-statistics rows) of adh_conv_stem_forward; 0 = not the stem
extern "C" int adh_conv_stem_num_blocks(const adh_conv_desc* d) {
    if (!stem_fwd_ok(d)) return 0;
    const int64_t ntiles = (int64_t)d->N * adh_ceil_div(d->OH, 4) * adh_ceil_div(d->OW, 64);
    return (int)(ntiles < 256 ? ntiles : 256);
}

extern "C" int adh_conv_stem_forward(void* stream, const adh_conv_desc* d) {
    if (!stem_fwd_ok(d)) return ADH_E_UNSUPPORTED;
    if (!d->in || !d->out || !d->wp || ((uintptr_t)d->in & 15) || ((uintptr_t)d->wp & 15)) return ADH_E_ARG;
    const int tiles_x = adh_ceil_div(d->OW, 64), tiles_y = adh_ceil_div(d->OH, 4);
    const int ntiles = d->N * tiles_x * tiles_y;
    const int nwg = adh_conv_stem_num_blocks(d);
    hipStream_t s = (hipStream_t)stream;
#define STEM_LAUNCH(nt_) \
    { \
        const int ldsb = (10 * ST_HC * 8 + 128 + ST_WROWS * 16 * nt_ + 8 * 16 * nt_) * 4; \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stem_fwd_kernel<nt_>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, ldsb); \
        hipLaunchKernelGGL((conv_stem_fwd_kernel<nt_>), dim3(nwg), dim3(256), ldsb, s, *d, tiles_x, tiles_y, ntiles); \
    }
    if (d->Cout == 96) STEM_LAUNCH(6)
    else STEM_LAUNCH(4)
#undef STEM_LAUNCH
    return adh_check_launch();
}

// wp[ky][e = kx * 3 + c (21 real of 24)][NcP] <- src(layout L, 7x7)
__global__ void pack_weights_stem_kernel(const float* __restrict__ src, const adh_wlayout L, int NcP, float* __restrict__ wp) {
    const int total = ST_WROWS * NcP;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int n = idx % NcP, row = idx / NcP;
        const int ky = row / 24, e = row - ky * 24;
        const int kx = e / 3, c = e - kx * 3;
        float v = 0.f;
        if (e < 21 && n < L.Nc && c < L.K)
            v = src[(int64_t)L.tap_off0 + ky * L.tap_off_sy + kx * L.tap_off_sx + (int64_t)c * L.stride_k + (int64_t)n * L.stride_n];
        wp[idx] = v;
    }
}

extern "C" int adh_pack_weights_stem(void* stream, const float* src, const adh_wlayout* L, float* wp) {
    if (!src || !L || !wp || L->KHt != 7 || L->KWt != 7 || L->K > 3) return ADH_E_ARG;
    const int NcP = adh_round_up(L->Nc, 32);
    hipLaunchKernelGGL(pack_weights_stem_kernel, dim3(adh_ceil_div(ST_WROWS * NcP, 256)), dim3(256), 0, (hipStream_t)stream, src,
                       *L, NcP, wp);
    return adh_check_launch();
}
