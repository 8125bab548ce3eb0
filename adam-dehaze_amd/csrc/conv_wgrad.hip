// Weight gradient of the gather-form convolution on fp32 MFMA, gfx950.
//
// dW[tap][k][n] = sum over virtual pixels m of X_tap[m][k] * G[m][n], where X_tap[m] is the input
// pixel tap `tap` reads for virtual pixel m and G is the gradient wrt the conv output (adh_conv_desc,
// with d->out read as G).  This is the autograd backward ATen runs for Conv2d / ConvTranspose2d in
// the reference's training step (/root/reference training/train_joint.py:153, train_dehazing.py:96).
//
// Decomposition: a workgroup (4 waves) owns one 32-wide input-channel tile x (32*TN) output channels x
// T taps and sweeps pixel tiles (TH rows x 32 columns), keeping dW in MFMA accumulators; the T*TN
// 32x32 output tiles are dealt to the 4 waves in compile-time contiguous shares (WgShare).  Per tile the input halo [pixels][32 ch] and the G tile [pixels][32*TN] are
// staged once in LDS (pixel-major, so the MFMA operands -- A[i=channel][k=pixel], B[k=pixel][j=channel]
// -- are conflict-free ds_read_b32 with consecutive lanes on consecutive channels, and every tap is an
// address offset).  The pixel dimension is split over gridDim.x workgroups; each writes its partial to
// slab[split] and adh_wgrad_reduce sums the splits in a fixed order (deterministic, no atomics).
#include "common.h"

#define WG_TW 32

// T = taps handled per block (consecutive linear taps from blockIdx.z*T).
// 512 threads: waves 0-3 contract the current pixel tile from LDS buffer `cur` on the MFMA pipe while
// waves 4-7 stage the next tile into the other buffer (loader / consumer split, one barrier per tile).
template <int T, int TN>
__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(const adh_conv_desc d, const ConvGeom g, float* slab,
                                                            int ntiles, int KP, int nco_groups) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int BN = 32 * TN;
    constexpr int PER = (T * TN + 3) / 4;
    const int XP = g.xp;
    const int buf_floats = g.npx * XP + 32 + g.TH * WG_TW * BN;   // +32: packed mode reads up to 3 pixels past a row

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int l31 = lane & 31;
    const int h = lane >> 5;

    const int ci_tile = blockIdx.x / nco_groups;
    const int co_grp = blockIdx.x - ci_tile * nco_groups;
    const int ci0 = ci_tile * 32;
    const int co0 = co_grp * BN;
    const int tap_base = blockIdx.z * T;

    const int XQ = XP / 4;
    const int xitems = g.npx * XQ;
    const int gitems = g.TH * WG_TW * (BN / 4);

    if (loader) {
        // ------------------------------------------------ loader waves: global -> LDS, one tile ahead
        const int lt = tid & 255;
        auto stage = [&](int tile, float* xs, float* gs) {
            int tt = tile;
            const int tx = tt % g.tiles_x;
            tt /= g.tiles_x;
            const int ty = tt % g.tiles_y;
            const int n = tt / g.tiles_y;
            const int vy0 = ty * g.TH, vx0 = tx * WG_TW;
            const int iy0 = vy0 * d.in_sy + g.dmin_y;
            const int ix0 = vx0 * d.in_sx + g.dmin_x;
            const float* in_n = d.in + (size_t)n * d.IH * d.IW * d.in_cstride;
            const float* g_n = d.out + (size_t)n * d.OH * d.OW * d.out_cstride;
            // loads are issued unconditionally (out-of-range lanes read the tensor base and are zeroed afterwards)
            // in batches of 8 per thread so that 8 x 16 B per lane are in flight before the first LDS write
            for (int base = lt; base < xitems; base += 256 * 8) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int item = base + u * 256;
                    const int pix = item / XQ, cq = item - pix * XQ;
                    const int hy = pix / g.halo_w, hx = pix - hy * g.halo_w;
                    const int iy = iy0 + hy, ix = ix0 + hx;
                    const int ci = ci0 + cq * 4;
                    const bool ok = item < xitems && iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW && ci < d.Cin;
                    const float* src = ok ? in_n + ((size_t)iy * d.IW + ix) * d.in_cstride + ci : in_n;
                    v[u] = *reinterpret_cast<const f32x4*>(src);
                    if (!ok) v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int item = base + u * 256;
                    if (item < xitems) {
                        const int pix = item / XQ, cq = item - pix * XQ;
                        *reinterpret_cast<f32x4*>(xs + pix * XP + cq * 4) = v[u];
                    }
                }
            }
            for (int base = lt; base < gitems; base += 256 * 8) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int item = base + u * 256;
                    const int pix = item / (BN / 4), cq = item - pix * (BN / 4);
                    const int vy = vy0 + (pix >> 5), vx = vx0 + (pix & 31);
                    const int co = co0 + cq * 4;
                    const bool ok = item < gitems && vy < d.VH && vx < d.VW && co < d.Cout;
                    const size_t opix = (size_t)(vy * d.out_sy + d.out_oy) * d.OW + (vx * d.out_sx + d.out_ox);
                    const float* src = ok ? g_n + opix * d.out_cstride + co : g_n;
                    v[u] = *reinterpret_cast<const f32x4*>(src);
                    if (!ok) v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int item = base + u * 256;
                    if (item < gitems) {
                        const int pix = item / (BN / 4), cq = item - pix * (BN / 4);
                        *reinterpret_cast<f32x4*>(gs + pix * BN + cq * 4) = v[u];
                    }
                }
            }
        };
        if ((int)blockIdx.y < ntiles) stage(blockIdx.y, smem, smem + (size_t)g.npx * XP + 32);
        __syncthreads();
        int cur = 0;
        for (int tile = blockIdx.y; tile < ntiles; tile += gridDim.y) {
            const int next = tile + gridDim.y;
            if (next < ntiles) {
                float* xn = smem + (size_t)(cur ^ 1) * buf_floats;
                stage(next, xn, xn + (size_t)g.npx * XP + 32);
            }
            __syncthreads();
            cur ^= 1;
        }
    } else {
        // ------------------------------------------------ compute waves: LDS -> MFMA
        f32x16 acc[PER];
#pragma unroll
        for (int t = 0; t < PER; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        // the T*TN output tiles (t = tap_local*TN + tn) are dealt to the 4 compute waves in contiguous shares
        // of PER; LDS offsets (floats) of each tile's two operands are wave-uniform scalars
        int aoff[PER], boff[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int t = adh_min_i(wave * PER + j, T * TN - 1);
            const int tl = t / TN;
            const int tap = tap_base + tl;
            const int tty = tap / d.KW, ttx = tap - tty * d.KW;
            const int dy = d.dy0 + tty * d.dstep_y - g.dmin_y;
            const int dx = d.dx0 + ttx * d.dstep_x - g.dmin_x;
            aoff[j] = (dy * g.halo_w + dx) * XP;
            boff[j] = 32 * (t - tl * TN);
        }
        __syncthreads();
        int cur = 0;
        for (int tile = blockIdx.y; tile < ntiles; tile += gridDim.y) {
            const float* xs = smem + (size_t)cur * buf_floats;
            const float* gs = xs + (size_t)g.npx * XP + 32;
            // software-pipelined k loop (one k-step = one pixel pair): operands of step s+1 are fetched from
            // LDS before the MFMAs of step s issue, ping-ponging two register sets
            const float* xbase = xs + h * d.in_sx * XP + l31;
            const float* gbase = gs + h * BN + l31;
            const int rowx = d.in_sy * g.halo_w * XP, qx = 2 * d.in_sx * XP;
            const int nsteps = g.TH * (WG_TW / 2);
            float a0[PER], b0[PER], a1[PER], b1[PER];
            auto ld = [&](int st, float (&a)[PER], float (&b)[PER]) {
                const int r = st >> 4, q = st & 15;
                const float* xr = xbase + r * rowx + q * qx;
                const float* gr = gbase + (r * WG_TW + q * 2) * BN;
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    a[j] = xr[aoff[j]];
                    b[j] = gr[boff[j]];
                }
            };
            ld(0, a0, b0);
            for (int st = 0; st < nsteps; st += 2) {
                ld(st + 1, a1, b1);
#pragma unroll
                for (int j = 0; j < PER; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[j], 0, 0, 0);
                if (st + 2 < nsteps) ld(st + 2, a0, b0);
#pragma unroll
                for (int j = 0; j < PER; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[j], 0, 0, 0);
            }
            __syncthreads();
            cur ^= 1;
        }
        // partial result -> slab[split][tap][KP][NcP]
        const int Ttot = d.KH * d.KW;
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int t = wave * PER + j;
            if (t < T * TN) {
                const int tap = tap_base + t / TN;
                const int tn = t - (t / TN) * TN;
                float* base = slab + (((size_t)blockIdx.y * Ttot + tap) * KP + ci0) * d.NcP + co0 + 32 * tn + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                    base[(size_t)i * d.NcP] = acc[j][r];
                }
            }
        }
    }
}

__host__ static int wgrad_geometry(const adh_conv_desc* d, ConvGeom* g, int TN, int xp) {
    if (!d || d->KH < 1 || d->KW < 1) return ADH_E_ARG;
    if (d->Cin % 4 != 0 || d->in_cstride % 4 != 0 || d->Cout % 4 != 0 || d->out_cstride % 4 != 0) return ADH_E_ARG;
    g->dmin_y = adh_tap_min(d->dy0, d->dstep_y, d->KH);
    g->dmin_x = adh_tap_min(d->dx0, d->dstep_x, d->KW);
    const int ey = adh_tap_max(d->dy0, d->dstep_y, d->KH) - g->dmin_y;
    const int ex = adh_tap_max(d->dx0, d->dstep_x, d->KW) - g->dmin_x;
    int TH = 4;
    for (;;) {
        g->halo_h = (TH - 1) * d->in_sy + ey + 1;
        g->halo_w = (WG_TW - 1) * d->in_sx + ex + 1 + (32 / xp - 1);   // packed mode: an i-tile spans 32/xp pixels
        g->npx = g->halo_h * g->halo_w;
        const int64_t bytes = (int64_t)g->npx * xp * 4 + 128 + (int64_t)TH * WG_TW * 32 * TN * 4;
        if (2 * bytes <= 160 * 1024 || TH == 1) break;   // two buffers in the CU's 160 KB
        TH >>= 1;
    }
    g->TH = TH;
    g->npxp = g->npx;
    g->tiles_x = adh_ceil_div(d->VW, WG_TW);
    g->tiles_y = adh_ceil_div(d->VH, TH);
    g->KC = 32;
    g->KQ_log2 = 3;
    g->KQtot = 0;
    g->xp = xp;
    return ADH_OK;
}

template <int T, int TN>
static void launch_wgrad(hipStream_t s, const adh_conv_desc* d, const ConvGeom& g, float* slab, int nsplit, int KP) {
    const int ntiles = g.tiles_x * g.tiles_y * d->N;
    const int nco_groups = d->NcP / (32 * TN);
    const int lds = 2 * (g.npx * g.xp * 4 + 128 + g.TH * WG_TW * 32 * TN * 4);
    // x = (ci tile, co group), y = pixel split: the workgroups that re-read one pixel range run together, so the
    // re-reads of the G / X tiles are served by L2 / Infinity Cache instead of HBM
    dim3 grid((KP / 32) * nco_groups, nsplit, (d->KH * d->KW) / T);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<T, TN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv_wgrad_kernel<T, TN>), grid, dim3(512), lds, s, *d, g, slab, ntiles, KP, nco_groups);
}

static int wgrad_pick_tn(int NcP, int T) {
    const int t = NcP / 32;
    for (int tn = 4; tn >= 1; --tn)
        if (t % tn == 0 && (T * tn + 3) / 4 <= 12) return tn;
    return 1;
}

extern "C" int adh_conv_wgrad(void* stream, const adh_conv_desc* d, float* slab, int nsplit) {
    if (!d || !slab || nsplit < 1 || !d->in || !d->out) return ADH_E_ARG;
    if (d->NcP % 32 != 0 || d->NcP < d->Cout) return ADH_E_ARG;
    const int Ttot = d->KH * d->KW;
    int T;
    if (Ttot == 9 || Ttot == 16 || Ttot == 4 || Ttot == 1) T = Ttot;
    else if (Ttot == 49 || Ttot == 14) T = 7;   // one kernel row (or packed column group) per block (blockIdx.z)
    else return ADH_E_UNSUPPORTED;
    // packed small-Cin mode (7x7 stems): the 32-wide MFMA row tile spans 4 adjacent pixels x 8 channels, so a
    // "tap" is (ky, group of 4 kx) and dstep_x == 4; see adh_wgrad_reduce_packed for the unpacking
    const bool packed = (d->dstep_x == 4 && d->Cin == 8 && d->in_sx == 1);
    const int TN = wgrad_pick_tn(d->NcP, T);
    ConvGeom g;
    int rc = wgrad_geometry(d, &g, TN, packed ? 8 : 32);
    if (rc) return rc;
    if ((d->VH - 1) * d->out_sy + d->out_oy >= d->OH || (d->VW - 1) * d->out_sx + d->out_ox >= d->OW) return ADH_E_ARG;
    const int KP = adh_round_up(d->Cin, 32);
    hipStream_t s = (hipStream_t)stream;
#define WG_CASE(t, tn) \
    if (T == t && TN == tn) { launch_wgrad<t, tn>(s, d, g, slab, nsplit, KP); return adh_check_launch(); }
    WG_CASE(9, 4) WG_CASE(9, 3) WG_CASE(9, 2) WG_CASE(9, 1)
    WG_CASE(16, 3) WG_CASE(16, 2) WG_CASE(16, 1)
    WG_CASE(4, 4) WG_CASE(4, 3) WG_CASE(4, 2) WG_CASE(4, 1)
    WG_CASE(7, 4) WG_CASE(7, 3) WG_CASE(7, 2) WG_CASE(7, 1)
    WG_CASE(1, 4) WG_CASE(1, 3) WG_CASE(1, 2) WG_CASE(1, 1)
#undef WG_CASE
    return ADH_E_UNSUPPORTED;
}

// dst(layout L) (+)= sum over splits, fixed order
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, int nsplit, int KP, int NcP, const adh_wlayout L,
                                    float* __restrict__ dst, int accumulate) {
    const int T = L.KHt * L.KWt;
    const int64_t total = (int64_t)T * L.K * L.Nc;
    const int64_t split_stride = (int64_t)T * KP * NcP;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % L.Nc);
        int64_t r = idx / L.Nc;
        const int k = (int)(r % L.K);
        const int tap = (int)(r / L.K);
        const float* p = slab + ((int64_t)tap * KP + k) * NcP + n;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int s = 0;
        for (; s + 4 <= nsplit; s += 4) {
            s0 += p[(int64_t)(s + 0) * split_stride];
            s1 += p[(int64_t)(s + 1) * split_stride];
            s2 += p[(int64_t)(s + 2) * split_stride];
            s3 += p[(int64_t)(s + 3) * split_stride];
        }
        for (; s < nsplit; ++s) s0 += p[(int64_t)s * split_stride];
        const float sum = (s0 + s1) + (s2 + s3);
        const int tyy = tap / L.KWt, txx = tap - tyy * L.KWt;
        const int64_t off = (int64_t)L.tap_off0 + tyy * L.tap_off_sy + txx * L.tap_off_sx + (int64_t)k * L.stride_k +
                            (int64_t)n * L.stride_n;
        dst[off] = accumulate ? dst[off] + sum : sum;
    }
}

extern "C" int adh_wgrad_reduce(void* stream, const float* slab, int nsplit, int KP, int NcP, const adh_wlayout* L,
                                float* dst, int accumulate) {
    if (!slab || !L || !dst || nsplit < 1) return ADH_E_ARG;
    const int64_t total = (int64_t)L->KHt * L->KWt * L->K * L->Nc;
    const int blocks = adh_min_i(adh_ceil_div(total, 256), 8192);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slab, nsplit, KP, NcP, *L,
                       dst, accumulate);
    return adh_check_launch();
}

// Packed small-Cin slabs: slab[s][tap=(ky*KWg+kxg)][i=(kxl*8+ci)][NcP] -> dst OIHW [Cout][Cin][KH][KW]
__global__ void wgrad_reduce_packed_kernel(const float* __restrict__ slab, int nsplit, int NcP, int Cin, int KH, int KW,
                                           int Cout, float* __restrict__ dst, int accumulate) {
    const int KWg = (KW + 3) / 4;
    const int64_t total = (int64_t)Cout * Cin * KH * KW;
    const int64_t split_stride = (int64_t)KH * KWg * 32 * NcP;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int kx = (int)(idx % KW);
        int64_t r = idx / KW;
        const int ky = (int)(r % KH);
        r /= KH;
        const int ci = (int)(r % Cin);
        const int co = (int)(r / Cin);
        const int tap = ky * KWg + (kx >> 2);
        const int i = (kx & 3) * 8 + ci;
        const float* p = slab + ((int64_t)tap * 32 + i) * NcP + co;
        float s0 = 0.f, s1 = 0.f;
        int s = 0;
        for (; s + 2 <= nsplit; s += 2) {
            s0 += p[(int64_t)s * split_stride];
            s1 += p[(int64_t)(s + 1) * split_stride];
        }
        if (s < nsplit) s0 += p[(int64_t)s * split_stride];
        const float sum = s0 + s1;
        dst[idx] = accumulate ? dst[idx] + sum : sum;
    }
}

extern "C" int adh_wgrad_reduce_packed(void* stream, const float* slab, int nsplit, int NcP, int Cin, int KH, int KW, int Cout,
                                       float* dst, int accumulate) {
    if (!slab || !dst || nsplit < 1 || Cin < 1 || Cin > 8) return ADH_E_ARG;
    const int64_t total = (int64_t)Cout * Cin * KH * KW;
    hipLaunchKernelGGL(wgrad_reduce_packed_kernel, dim3(adh_min_i(adh_ceil_div(total, 256), 4096)), dim3(256), 0,
                       (hipStream_t)stream, slab, nsplit, NcP, Cin, KH, KW, Cout, dst, accumulate);
    return adh_check_launch();
}
