// Gather-form implicit-GEMM convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// One kernel serves Conv2d k1/k3/k7 s1, Conv2d k4 s2, every output-parity class of
// ConvTranspose2d k4 s2 p1 and the data gradients of all of them (adam_dehaze_hip.h, adh_conv_desc).
// Replaces the ATen conv2d / conv_transpose2d calls behind ConvBlock and the decoder stages
// (/root/reference models/dehazing/base_model.py:11-13, medium_intensity.py:53,63).
//
// This is the general kernel: adh_conv_forward first offers the launch to conv_rows_kernel (conv_rows.hip: 2x2- and
// 3x3-tap forms on 16x32-aligned grids), and the engine routes 3x3 s1 and k4 s2 / transposed layers with Cin % 16 == 0
// to the Winograd kernels (conv_wino.hip); what arrives here are the stems (7x7, Cin 3), the small heads, 1x1
// convolutions and ragged grids.
//
// Work decomposition (per 256-thread workgroup = 4 waves, 2 workgroups per CU):
//   tile = 8 rows x 32 columns of virtual output pixels x (32*TN) output channels;
//   wave w owns rows 2w,2w+1 (two 32-pixel MFMA row blocks) x TN column blocks -> 2*TN accumulators.
//   K loop: input channels in chunks of KC (8..32); for each chunk the input halo of the tile is
//   staged ONCE into LDS (channel-quad major: [KC/4][pixel][4], 16-B pixel pitch so every tap is a
//   plain address offset and ds_read_b128 is conflict free), then all taps x channel-quads are
//   contracted from LDS.  Weights are pre-packed [tap][k/4][n][4] and read straight from global/L2
//   into VGPRs (every wave of every block reads the same slab; 16 B per lane, coalesced).
//   One ds_read_b128 / global_load_dwordx4 feeds 4 MFMAs: lane half h of MFMA j consumes
//   k = 8g + 4h + j (the k order inside the sum is free as long as both operands agree).
//   Next chunk's halo is prefetched into registers while the current one is contracted.
// Epilogue (fused): per-channel scale/shift (bias or folded BN), residual add, ReLU, and the
// per-block sum / sum-of-squares partials train-mode BatchNorm needs.
#include "common.h"

#define CONV_TW 32
#define CONV_TH 8
#define CONV_MAXIT 12   // max float4 staged per thread per chunk (6 for the TN=4 variant: register budget)

__host__ static int conv_geometry(const adh_conv_desc* d, ConvGeom* g, int TH, int maxit) {
    if (!d || d->KH < 1 || d->KW < 1 || d->in_sy < 1 || d->in_sx < 1) return ADH_E_ARG;
    if (d->Cin % 8 != 0 || d->in_cstride % 4 != 0) return ADH_E_ARG;
    g->TH = TH;
    g->dmin_y = adh_tap_min(d->dy0, d->dstep_y, d->KH);
    g->dmin_x = adh_tap_min(d->dx0, d->dstep_x, d->KW);
    int dmax_y = adh_tap_max(d->dy0, d->dstep_y, d->KH);
    int dmax_x = adh_tap_max(d->dx0, d->dstep_x, d->KW);
    g->halo_h = (TH - 1) * d->in_sy + (dmax_y - g->dmin_y) + 1;
    g->halo_w = (CONV_TW - 1) * d->in_sx + (dmax_x - g->dmin_x) + 1;
    g->npx = g->halo_h * g->halo_w;
    g->npxp = g->npx | 1;
    g->tiles_x = adh_ceil_div(d->VW, CONV_TW);
    g->tiles_y = adh_ceil_div(d->VH, TH);
    // largest chunk (32,16,8) that divides Cin, fits the staging registers and ~64 KB of LDS
    int KC = 32;
    while (KC > 8 && (d->Cin % KC != 0 || (int64_t)g->npx * (KC / 4) > (int64_t)maxit * 256 ||
                      (int64_t)g->npxp * KC * 4 > 66000))
        KC >>= 1;
    if (d->Cin % KC != 0 || (int64_t)g->npx * (KC / 4) > (int64_t)maxit * 256) return ADH_E_UNSUPPORTED;
    g->KC = KC;
    g->KQ_log2 = (KC == 32) ? 3 : (KC == 16) ? 2 : 1;
    g->KQtot = d->Cin / 4;
    return ADH_OK;
}

template <int TN, int MAXIT>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(const adh_conv_desc d, const ConvGeom g) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];  // [KC/4][npxp]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int h = lane >> 5;

    int tile = blockIdx.x;
    const int tx = tile % g.tiles_x;
    tile /= g.tiles_x;
    const int ty = tile % g.tiles_y;
    const int n = tile / g.tiles_y;
    const int vy0 = ty * CONV_TH, vx0 = tx * CONV_TW;
    const int co0 = blockIdx.y * (32 * TN);

    const int iy0 = vy0 * d.in_sy + g.dmin_y;
    const int ix0 = vx0 * d.in_sx + g.dmin_x;
    const float* in_n = d.in + (size_t)n * d.IH * d.IW * d.in_cstride;

    const int KQ = 1 << g.KQ_log2;
    const int items = g.npx << g.KQ_log2;

    // per-thread staging plan: global element offset (or -1) of each float4 this thread stages
    int goff[MAXIT];
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int item = tid + it * 256;
        int off = -1;
        if (item < items) {
            const int pix = item >> g.KQ_log2;
            const int cq = item & (KQ - 1);
            const int hy = pix / g.halo_w;
            const int hx = pix - hy * g.halo_w;
            const int iy = iy0 + hy, ix = ix0 + hx;
            if (iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW) off = (iy * d.IW + ix) * d.in_cstride + cq * 4;
        }
        goff[it] = off;
    }

    f32x4 stage[MAXIT];
    auto load_chunk = [&](int c) {
        const float* base = in_n + c * g.KC;
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (goff[it] >= 0) v = *reinterpret_cast<const f32x4*>(base + goff[it]);
            stage[it] = v;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int it = 0; it < MAXIT; ++it) {
            const int item = tid + it * 256;
            if (item < items) {
                const int pix = item >> g.KQ_log2;
                const int cq = item & (KQ - 1);
                lds[cq * g.npxp + pix] = stage[it];
            }
        }
    };

    f32x16 acc[2][TN];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const f32x4* wp4 = reinterpret_cast<const f32x4*>(d.wp);
    const int G = g.KC >> 3;                 // 8-channel groups per chunk
    const int ntaps = d.KH * d.KW;
    const int iters = ntaps * G;
    const int nchunks = d.Cin / g.KC;
    // pixel index (inside the halo) of this lane for row block tm, tap (0,0) relative to dmin
    const int arow0 = (2 * wave) * d.in_sy * g.halo_w + l31 * d.in_sx;
    const int arow1 = arow0 + d.in_sy * g.halo_w;

    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();  // previous chunk fully consumed
        store_chunk();
        __syncthreads();
        if (c + 1 < nchunks) load_chunk(c + 1);  // in flight while this chunk is contracted

        // flattened (tap, g) loop with one-iteration-ahead operand prefetch
        int tty = 0, ttx = 0, gg = 0;
        f32x4 a_cur[2], b_cur[TN], a_nxt[2], b_nxt[TN];
        auto fetch = [&](int fty, int ftx, int fg, f32x4 (&a)[2], f32x4 (&b)[TN]) {
            const int dy = d.dy0 + fty * d.dstep_y - g.dmin_y;
            const int dx = d.dx0 + ftx * d.dstep_x - g.dmin_x;
            const int cq = 2 * fg + h;
            const int poff = dy * g.halo_w + dx + cq * g.npxp;
            a[0] = lds[arow0 + poff];
            a[1] = lds[arow1 + poff];
            const int tap = fty * d.KW + ftx;
            const f32x4* wrow = wp4 + (size_t)(tap * g.KQtot + (c << g.KQ_log2) + cq) * d.NcP + co0 + l31;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[tn] = wrow[32 * tn];
        };
        fetch(0, 0, 0, a_cur, b_cur);
        a_nxt[0] = a_cur[0];
        a_nxt[1] = a_cur[1];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b_nxt[tn] = b_cur[tn];
        for (int it = 0; it < iters; ++it) {
            int ng = gg + 1, ntx = ttx, nty = tty;
            if (ng == G) {
                ng = 0;
                ++ntx;
                if (ntx == d.KW) {
                    ntx = 0;
                    ++nty;
                }
            }
            if (it + 1 < iters) fetch(nty, ntx, ng, a_nxt, b_nxt);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[tm][j], b_cur[tn][j], acc[tm][tn], 0, 0, 0);
            a_cur[0] = a_nxt[0];
            a_cur[1] = a_nxt[1];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b_cur[tn] = b_nxt[tn];
            gg = ng;
            ttx = ntx;
            tty = nty;
        }
    }

    // ------------------------------- epilogue -------------------------------------------------
    float ssum[TN], ssq[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) ssum[tn] = ssq[tn] = 0.f;

    float* out_n = d.out + (size_t)n * d.OH * d.OW * d.out_cstride;
    const float* res_n = d.residual ? d.residual + (size_t)n * d.OH * d.OW * d.res_cstride : nullptr;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int co = co0 + 32 * tn + l31;
        const bool cvalid = co < d.Cout;
        const float sc = (d.scale && cvalid) ? d.scale[co] : 1.f;
        const float sh = (d.shift && cvalid) ? d.shift[co] : 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm) {
            const int vy = vy0 + 2 * wave + tm;
            const int oy = vy * d.out_sy + d.out_oy;
            const bool yvalid = cvalid && vy < d.VH;
            // residual values are fetched for the whole 16-row fragment first so the loads overlap
            float resv[16];
            if (res_n) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const int vx = vx0 + i;
                    const bool valid = yvalid && vx < d.VW;
                    const size_t pix = valid ? (size_t)oy * d.OW + (vx * d.out_sx + d.out_ox) : 0;
                    resv[r] = res_n[valid ? pix * d.res_cstride + co : 0];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int vx = vx0 + i;
                const bool valid = yvalid && vx < d.VW;
                float v = acc[tm][tn][r] * sc + sh;
                if (valid) {
                    ssum[tn] += v;
                    ssq[tn] += v * v;
                    const size_t pix = (size_t)oy * d.OW + (vx * d.out_sx + d.out_ox);
                    if (res_n) v += resv[r];
                    if (d.act == ADH_ACT_RELU) v = fmaxf(v, 0.f);
                    out_n[pix * d.out_cstride + co] = v;
                }
            }
        }
    }

    if (d.stats) {
        // combine the two lane halves (same channel), then the 4 waves through LDS (fixed order)
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);  // [4][2][32*TN]
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            float s = ssum[tn] + __shfl_xor(ssum[tn], 32, 64);
            float q = ssq[tn] + __shfl_xor(ssq[tn], 32, 64);
            if (h == 0) {
                red[(wave * 2 + 0) * (32 * TN) + 32 * tn + l31] = s;
                red[(wave * 2 + 1) * (32 * TN) + 32 * tn + l31] = q;
            }
        }
        __syncthreads();
        if (tid < 2 * 32 * TN) {
            const int which = tid / (32 * TN);
            const int cl = tid - which * (32 * TN);
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[(w * 2 + which) * (32 * TN) + cl];
            d.stats[((size_t)blockIdx.x * 2 + which) * d.NcP + co0 + cl] = v;
        }
    }
}

static int pick_tn(int NcP) {
    const int t = NcP / 32;
    if (t % 4 == 0) return 4;
    if (t % 3 == 0) return 3;
    if (t % 2 == 0) return 2;
    return 1;
}

static int conv_plan(const adh_conv_desc* d, ConvGeom* g, int* TN) {
    if (!d || d->NcP % 32 != 0 || d->NcP < 32) return ADH_E_ARG;
    *TN = pick_tn(d->NcP);
    int rc = conv_geometry(d, g, CONV_TH, *TN == 4 ? 6 : CONV_MAXIT);
    if (rc == ADH_E_UNSUPPORTED && *TN == 4) {  // halo too large for the 6-deep staging: use two 64-wide groups
        *TN = 2;
        rc = conv_geometry(d, g, CONV_TH, CONV_MAXIT);
    }
    return rc;
}

extern "C" int adh_conv_lds_bytes(const adh_conv_desc* d) {
    ConvGeom g;
    int TN;
    int rc = conv_plan(d, &g, &TN);
    if (rc) return rc;
    return adh_max_i(g.npxp * g.KC * 4, 4 * 2 * 32 * TN * 4);
}

extern "C" int adh_conv_num_blocks(const adh_conv_desc* d) {
    const int nb_rows = adh_rows_fwd_num_blocks(d);
    if (nb_rows > 0) return nb_rows;
    ConvGeom g;
    int TN;
    int rc = conv_plan(d, &g, &TN);
    if (rc) return rc;
    return g.tiles_x * g.tiles_y * d->N;
}

extern "C" int adh_conv_forward(void* stream, const adh_conv_desc* d) {
    ConvGeom g;
    int TN;
    int rc = conv_plan(d, &g, &TN);
    if (rc) return rc;
    if (!d->in || !d->out || !d->wp) return ADH_E_ARG;
    if (d->NcP < d->Cout) return ADH_E_ARG;
    if (d->out_cstride < d->Cout || (d->residual && d->res_cstride < d->Cout)) return ADH_E_ARG;
    if (d->VH < 1 || d->VW < 1 || d->N < 1) return ADH_E_ARG;
    // every virtual pixel must land inside the output tensor
    if ((d->VH - 1) * d->out_sy + d->out_oy >= d->OH || (d->VW - 1) * d->out_sx + d->out_ox >= d->OW ||
        d->out_oy < 0 || d->out_ox < 0)
        return ADH_E_ARG;
    if (((uintptr_t)d->in & 15) || ((uintptr_t)d->wp & 15)) return ADH_E_ARG;
    if ((int64_t)d->IH * d->IW * d->in_cstride >= (1ll << 31) || (int64_t)d->OH * d->OW * d->out_cstride >= (1ll << 31))
        return ADH_E_UNSUPPORTED;
    rc = adh_rows_fwd_launch(stream, d);
    if (rc != ADH_E_UNSUPPORTED) return rc;
    const int lds = adh_max_i(g.npxp * g.KC * 4, 4 * 2 * 32 * TN * 4);
    dim3 grid(g.tiles_x * g.tiles_y * d->N, d->NcP / (32 * TN));
    hipStream_t s = (hipStream_t)stream;
    switch (TN) {
        case 4: hipLaunchKernelGGL((conv_igemm_kernel<4, 6>), grid, dim3(256), lds, s, *d, g); break;
        case 3: hipLaunchKernelGGL((conv_igemm_kernel<3, CONV_MAXIT>), grid, dim3(256), lds, s, *d, g); break;
        case 2: hipLaunchKernelGGL((conv_igemm_kernel<2, CONV_MAXIT>), grid, dim3(256), lds, s, *d, g); break;
        default: hipLaunchKernelGGL((conv_igemm_kernel<1, CONV_MAXIT>), grid, dim3(256), lds, s, *d, g); break;
    }
    return adh_check_launch();
}

// ------------------------------------ weight packing -----------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ src, const adh_wlayout L, int KQ, int NcP,
                                    f32x4* __restrict__ wp) {
    const int64_t total = (int64_t)L.KHt * L.KWt * KQ * NcP;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % NcP);
        int64_t r = idx / NcP;
        const int kq = (int)(r % KQ);
        const int tap = (int)(r / KQ);
        const int tyy = tap / L.KWt, txx = tap - tyy * L.KWt;
        const int toff = L.tap_off0 + tyy * L.tap_off_sy + txx * L.tap_off_sx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < L.Nc) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kq * 4 + j;
                if (k < L.K) v[j] = src[(int64_t)toff + (int64_t)k * L.stride_k + (int64_t)n * L.stride_n];
            }
        }
        wp[idx] = v;
    }
}

extern "C" int adh_pack_weights(void* stream, const float* src, const adh_wlayout* L, float* wp) {
    if (!src || !L || !wp || L->K < 1 || L->Nc < 1) return ADH_E_ARG;
    const int KQ = adh_round_up(L->K, 8) / 4;
    const int NcP = adh_round_up(L->Nc, 32);
    const int64_t total = (int64_t)L->KHt * L->KWt * KQ * NcP;
    const int blocks = adh_min_i(adh_ceil_div(total, 256), 4096);
    hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, *L, KQ, NcP,
                       reinterpret_cast<f32x4*>(wp));
    return adh_check_launch();
}
