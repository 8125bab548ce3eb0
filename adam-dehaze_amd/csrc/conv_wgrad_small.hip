// Weight gradient of the 3x3 stride-1 pad-1 convolutions with few channels (the guidance branch 3 -> 16 -> 16 and the
// output convolution 48 -> 3 of the Complex model, all at full resolution), fp32 MFMA 16x16x4, gfx950.
// Replaces the weight half of ATen's conv2d backward for those layers
// (/root/reference models/dehazing/high_intensity.py:74-90 -- output_conv / detail_branch).
//
// These layers are HBM-bound (19 GFLOP against 0.5 GB of operands), but in 32x32 MFMA tiles three quarters of every
// tile are padding and the general kernel runs them at 1.4 ms.  Here one 16-row MFMA tile is exactly one filter tap x
// 16 input channels, and the 16 columns are the (up to 16) output channels:
//     dW[tap][ci][co] = sum over pixels  x[pixel + tap][ci] * g[pixel][co]      ->  D(16 ci x 16 co) += A(16 x 4 px) B(4 px x 16)
// A workgroup stages a 4 x 64-pixel tile of g and its 6 x 66 halo of x in LDS (plain loads, zero fill at the borders;
// several workgroups per CU hide the latency); wave w contracts pixel row w: per group of 4 pixels one ds_read_b32 of g
// and one per (tap, channel block) of x feed 9 * CIB MFMAs.  Channel pitches smaller than 16 (the NHWC8 image, the
// 3-channel output gradient) simply let the tile read into the next pixel: those rows / columns are never copied out.
// Output: one slab [9][KP][NcP] per workgroup (waves summed through LDS); adh_wgrad_reduce adds the slabs.
#include "common.h"

typedef float f32x4s __attribute__((ext_vector_type(4)));

template <int XC, int GC, int CIB>
__global__ __launch_bounds__(256) void conv_wgrad_small_kernel(const adh_conv_desc d, int tiles_x, int tiles_y, int ntiles,
                                                               float* __restrict__ slab, int KP, int NcP) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int XS_F = 6 * 66 * XC + 16;        // x halo [6][66][XC] (+ over-read pad)
    // g tile [4][64][GC] (+ pad) follows
    float* const xs = lds;
    float* const gs = lds + XS_F;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i16 = lane & 15, kk = lane >> 4;

    f32x4s acc[9 * CIB];
#pragma unroll
    for (int t = 0; t < 9 * CIB; ++t) acc[t] = f32x4s{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int r = tile;
        const int tx = r % tiles_x;
        r /= tiles_x;
        const int ty = r % tiles_y;
        const int n = r / tiles_y;
        const int y0 = ty * 4, x0 = tx * 64;
        const float* xin = d.in + (int64_t)n * d.IH * d.IW * XC;
        const float* gin = d.out + (int64_t)n * d.VH * d.VW * GC;
        __syncthreads();   // previous tile fully consumed
        // x halo: rows y0-1 .. y0+4, columns x0-1 .. x0+64
        constexpr int XQ = XC / 4;
        for (int i = tid; i < 6 * 66 * XQ; i += 256) {
            const int px = i / XQ, q = i - px * XQ;
            const int row = px / 66, col = px - row * 66;
            const int iy = y0 - 1 + row, ix = x0 - 1 + col;
            f32x4s v = {0.f, 0.f, 0.f, 0.f};
            if (iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW)
                v = *reinterpret_cast<const f32x4s*>(xin + ((int64_t)iy * d.IW + ix) * XC + q * 4);
            *reinterpret_cast<f32x4s*>(xs + px * XC + q * 4) = v;
        }
        constexpr int GQ = GC / 4;
        for (int i = tid; i < 4 * 64 * GQ; i += 256) {
            const int px = i / GQ, q = i - px * GQ;
            const int row = px >> 6, col = px & 63;
            const int gy = y0 + row, gx = x0 + col;
            f32x4s v = {0.f, 0.f, 0.f, 0.f};
            if (gy < d.VH && gx < d.VW) v = *reinterpret_cast<const f32x4s*>(gin + ((int64_t)gy * d.VW + gx) * GC + q * 4);
            *reinterpret_cast<f32x4s*>(gs + px * GC + q * 4) = v;
        }
        if (tid < 16) {   // over-read pads: finite values
            xs[6 * 66 * XC + tid] = 0.f;
            gs[4 * 64 * GC + tid] = 0.f;
        }
        __syncthreads();
        // wave w: output row w of the tile, 16 groups of 4 pixels
        const float* gl = gs + (wave * 64 + kk) * GC + i16;
        const float* xl = xs + (wave * 66 + kk) * XC + i16;
#pragma unroll 2
        for (int kg = 0; kg < 16; ++kg) {
            const float b = gl[kg * 4 * GC];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int cb = 0; cb < CIB; ++cb) {
                        const float a = xl[(dy * 66 + kg * 4 + dx) * XC + cb * 16];
                        acc[(dy * 3 + dx) * CIB + cb] =
                            __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[(dy * 3 + dx) * CIB + cb], 0, 0, 0);
                    }
        }
    }

    // waves 1..3 hand their tiles to wave 0 through LDS one after the other (one wave's worth of LDS), then wave 0 writes
    // the workgroup's slab: slab[blockIdx][tap][ci][co]
    float* red = lds;   // [9 * CIB tiles][64 lanes][4]
#pragma unroll 1
    for (int w = 1; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 9 * CIB; ++t) *reinterpret_cast<f32x4s*>(red + (t * 64 + lane) * 4) = acc[t];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < 9 * CIB; ++t) acc[t] += *reinterpret_cast<const f32x4s*>(red + (t * 64 + lane) * 4);
        }
    }
    if (wave == 0) {
        float* const sl = slab + (int64_t)blockIdx.x * 9 * KP * NcP;
        const int co = lane & 15;
#pragma unroll
        for (int t = 0; t < 9 * CIB; ++t) {
            const int tap = t / CIB, cb = t - tap * CIB;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int ci = cb * 16 + 4 * (lane >> 4) + rr;   // D layout of 16x16x4: row = 4 * (lane / 16) + reg, column = lane % 16
                if (ci < KP && co < NcP) sl[((int64_t)tap * KP + ci) * NcP + co] = acc[t][rr];
            }
        }
    }
}

// Few OUTPUT channels (the 48 -> 3 output convolution): the roles are exchanged -- the tile rows are (tap, co) pairs
// packed with pitch COUT = 4 (36 rows = 3 tiles instead of 9 taps x 3 channel blocks = 27 tiles), the columns are 16
// input channels:  D[(tap, co)][ci] += A[(tap, co)][4 px] B[4 px][ci],  A = g[pixel - tap + 1][co] from the LDS halo of
// the output gradient (per-lane offset), B = x[pixel][ci] from the LDS tile of x (no halo).
template <int XC, int GC, int COUT>
__global__ __launch_bounds__(256) void conv_wgrad_fewout_kernel(const adh_conv_desc d, int tiles_x, int tiles_y, int ntiles,
                                                                float* __restrict__ slab, int KP, int NcP) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int CJB = XC / 16;                   // column tiles (input channels)
    constexpr int RT = (9 * COUT + 15) / 16;       // row tiles
    constexpr int GS_F = 6 * 66 * GC + 16;
    float* const gs = lds;                         // g halo [6][66][GC]: rows y0-1 .. y0+4, columns x0-1 .. x0+64
    float* const xs = lds + GS_F;                  // x tile [4][64][XC]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int i16 = lane & 15, kk = lane >> 4;
    int aoff[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int rI = adh_min_i(t * 16 + i16, 9 * COUT - 1);
        const int tap = rI / COUT, co = rI - tap * COUT;
        const int dy = tap / 3, dx = tap - dy * 3;
        aoff[t] = ((2 - dy) * 66 + (2 - dx)) * GC + co;
    }
    f32x4s acc[RT * CJB];
#pragma unroll
    for (int t = 0; t < RT * CJB; ++t) acc[t] = f32x4s{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int r = tile;
        const int tx = r % tiles_x;
        r /= tiles_x;
        const int ty = r % tiles_y;
        const int n = r / tiles_y;
        const int y0 = ty * 4, x0 = tx * 64;
        const float* xin = d.in + (int64_t)n * d.IH * d.IW * XC;
        const float* gin = d.out + (int64_t)n * d.VH * d.VW * GC;
        __syncthreads();
        constexpr int GQ = GC / 4;
        for (int i = tid; i < 6 * 66 * GQ; i += 256) {
            const int px = i / GQ, q = i - px * GQ;
            const int row = px / 66, col = px - row * 66;
            const int gy = y0 - 1 + row, gx = x0 - 1 + col;
            f32x4s v = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < d.VH && gx >= 0 && gx < d.VW) v = *reinterpret_cast<const f32x4s*>(gin + ((int64_t)gy * d.VW + gx) * GC + q * 4);
            *reinterpret_cast<f32x4s*>(gs + px * GC + q * 4) = v;
        }
        constexpr int XQ = XC / 4;
        for (int i = tid; i < 4 * 64 * XQ; i += 256) {
            const int px = i / XQ, q = i - px * XQ;
            const int row = px >> 6, col = px & 63;
            const int iy = y0 + row, ix = x0 + col;
            f32x4s v = {0.f, 0.f, 0.f, 0.f};
            if (iy < d.IH && ix < d.IW) v = *reinterpret_cast<const f32x4s*>(xin + ((int64_t)iy * d.IW + ix) * XC + q * 4);
            *reinterpret_cast<f32x4s*>(xs + px * XC + q * 4) = v;
        }
        if (tid < 16) gs[6 * 66 * GC + tid] = 0.f;
        __syncthreads();
        const float* gl = gs + (wave * 66 + kk) * GC;
        const float* xl = xs + (wave * 64 + kk) * XC + i16;
#pragma unroll 4
        for (int kg = 0; kg < 16; ++kg) {
            float a[RT], b[CJB];
#pragma unroll
            for (int t = 0; t < RT; ++t) a[t] = gl[kg * 4 * GC + aoff[t]];
#pragma unroll
            for (int c = 0; c < CJB; ++c) b[c] = xl[kg * 4 * XC + c * 16];
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int c = 0; c < CJB; ++c)
                    acc[t * CJB + c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[c], acc[t * CJB + c], 0, 0, 0);
        }
    }
    float* red = lds;   // [RT * CJB tiles][64 lanes][4]
#pragma unroll 1
    for (int w = 1; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < RT * CJB; ++t) *reinterpret_cast<f32x4s*>(red + (t * 64 + lane) * 4) = acc[t];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int t = 0; t < RT * CJB; ++t) acc[t] += *reinterpret_cast<const f32x4s*>(red + (t * 64 + lane) * 4);
        }
    }
    if (wave == 0) {
        float* const sl = slab + (int64_t)blockIdx.x * 9 * KP * NcP;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int c = 0; c < CJB; ++c)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int rI = t * 16 + 4 * kk + rr;          // D layout: row = 4 * (lane / 16) + reg, column = lane % 16
                    const int tap = rI / COUT, co = rI - tap * COUT;
                    const int ci = c * 16 + i16;
                    if (rI < 9 * COUT && ci < KP) sl[((int64_t)tap * KP + ci) * NcP + co] = acc[t * CJB + c][rr];
                }
    }
}

static int small_combo(const adh_conv_desc* d) {
    if (!d) return 0;
    if (d->KH != 3 || d->KW != 3 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1) return 0;
    if (d->out_oy != 0 || d->out_ox != 0 || d->dy0 != -1 || d->dx0 != -1 || d->dstep_y != 1 || d->dstep_x != 1) return 0;
    if (d->VH != d->IH || d->VW != d->IW || d->Cin > d->in_cstride || d->Cout > 16 || d->Cout > d->out_cstride) return 0;
    if (d->in_cstride == 8 && d->out_cstride == 16 && d->Cin <= 8) return 1;
    if (d->in_cstride == 16 && d->out_cstride == 16) return 2;
    if (d->in_cstride == 48 && d->out_cstride == 8) return d->Cout <= 4 ? 4 : 3;
    return 0;
}

// number of workgroups (= slabs [9][KP][NcP]) adh_conv_wgrad_small writes for `d`; 0 = not one of its shapes
extern "C" int adh_conv_wgrad_small_slabs(const adh_conv_desc* d) {
    if (!small_combo(d)) return 0;
    const int64_t ntiles = (int64_t)d->N * adh_ceil_div(d->VH, 4) * adh_ceil_div(d->VW, 64);
    return (int)(ntiles < 768 ? ntiles : 768);
}

extern "C" int adh_conv_wgrad_small(void* stream, const adh_conv_desc* d, float* slab, int KP, int NcP) {
    const int combo = small_combo(d);
    if (!combo) return ADH_E_UNSUPPORTED;
    if (!slab || !d->in || !d->out || KP < d->in_cstride || KP < 16 || NcP < 16) return ADH_E_ARG;
    if (((uintptr_t)d->in & 15) || ((uintptr_t)d->out & 15)) return ADH_E_ARG;
    const int tiles_x = adh_ceil_div(d->VW, 64), tiles_y = adh_ceil_div(d->VH, 4);
    const int ntiles = d->N * tiles_x * tiles_y;
    const int nwg = adh_conv_wgrad_small_slabs(d);
    hipStream_t s = (hipStream_t)stream;
#define SMALL_LAUNCH(xc_, gc_, cib_) \
    { \
        const int ldsb = (6 * 66 * xc_ + 16 + 4 * 64 * gc_ + 16) * 4;   /* (>= one wave's 9 * cib tiles) */ \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_small_kernel<xc_, gc_, cib_>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, ldsb); \
        hipLaunchKernelGGL((conv_wgrad_small_kernel<xc_, gc_, cib_>), dim3(nwg), dim3(256), ldsb, s, *d, tiles_x, tiles_y, ntiles, \
                           slab, KP, NcP); \
    }
    if (combo == 1) SMALL_LAUNCH(8, 16, 1)
    else if (combo == 2) SMALL_LAUNCH(16, 16, 1)
    else if (combo == 3) SMALL_LAUNCH(48, 8, 3)
    else {
        const int ldsb = (6 * 66 * 8 + 16 + 4 * 64 * 48) * 4;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_fewout_kernel<48, 8, 4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
        hipLaunchKernelGGL((conv_wgrad_fewout_kernel<48, 8, 4>), dim3(nwg), dim3(256), ldsb, s, *d, tiles_x, tiles_y, ntiles, slab, KP,
                           NcP);
    }
#undef SMALL_LAUNCH
    return adh_check_launch();
}

// dst(layout L) (+)= sum over slabs[s][tap][k][n]: one wave per output element (few outputs, hundreds of slabs: the
// thread-per-output reduce of conv_wgrad.hip would run one long dependent chain per thread)
__global__ __launch_bounds__(256) void wgrad_reduce_wave_kernel(const float* __restrict__ slab, int nslabs, int KP, int NcP,
                                                                const adh_wlayout L, float* __restrict__ dst, int accumulate) {
    const int T = L.KHt * L.KWt;
    const int64_t total = (int64_t)T * L.K * L.Nc;
    const int64_t split_stride = (int64_t)T * KP * NcP;
    const int lane = threadIdx.x & 63;
    for (int64_t idx = blockIdx.x * 4 + (threadIdx.x >> 6); idx < total; idx += (int64_t)gridDim.x * 4) {
        const int n = (int)(idx % L.Nc);
        int64_t r = idx / L.Nc;
        const int k = (int)(r % L.K);
        const int tap = (int)(r / L.K);
        const float* p = slab + ((int64_t)tap * KP + k) * NcP + n;
        float s = 0.f;
        for (int i = lane; i < nslabs; i += 64) s += p[(int64_t)i * split_stride];
        s = wave_sum(s);
        if (lane == 0) {
            const int tyy = tap / L.KWt, txx = tap - tyy * L.KWt;
            const int64_t off = (int64_t)L.tap_off0 + tyy * L.tap_off_sy + txx * L.tap_off_sx + (int64_t)k * L.stride_k +
                                (int64_t)n * L.stride_n;
            dst[off] = accumulate ? dst[off] + s : s;
        }
    }
}

extern "C" int adh_wgrad_reduce_small(void* stream, const float* slab, int nslabs, int KP, int NcP, const adh_wlayout* L,
                                      float* dst, int accumulate) {
    if (!slab || !L || !dst || nslabs < 1) return ADH_E_ARG;
    const int64_t total = (int64_t)L->KHt * L->KWt * L->K * L->Nc;
    hipLaunchKernelGGL(wgrad_reduce_wave_kernel, dim3(adh_min_i(adh_ceil_div(total, 4), 4096)), dim3(256), 0, (hipStream_t)stream,
                       slab, nslabs, KP, NcP, *L, dst, accumulate);
    return adh_check_launch();
}
