// Direct ("rows") forward convolution on fp32 MFMA for the gather forms whose taps step through the input in
// units of the per-pixel input stride: every output-parity class of ConvTranspose2d k4 s2 (2x2 taps), Conv2d k4 s2
// (four input-parity classes of 2x2 taps accumulated in one launch), their data gradients (the same two forms
// with the roles swapped) and 3x3 unit-stride convolutions when the Winograd path is switched off.  Replaces the
// same ATen calls as conv_igemm.hip (/root/reference models/dehazing/base_model.py:11-13,
// medium_intensity.py:53,63) with the same descriptor and the same fused epilogue.
//
// Same design rules as conv_wgrad.hip / conv_wino.hip (tools/micro/: VALU work does not hide behind fp32 MFMAs,
// LDS reads / LDS-DMA / scalar work do): 4 waves, one per SIMD, 512 registers each; a workgroup owns a region of
// 16 rows x 32 columns of virtual pixels x 32*Q output channels, wave w rows 4w..4w+3 = four 32-pixel MFMA row
// tiles x Q column tiles (up to 24 accumulator tiles, the first 16 pinned to AGPRs).
// K loop over slabs = (16 input channels, input-parity class):
//   * the slab's halo [16+KH-1][48 px][16 ch] arrives by LDS-DMA (16-pixel pieces, double buffered, one barrier
//     per slab); the channel quad a lane fetches is XOR-swizzled with its pixel index, which makes the
//     ds_read_b128 MFMA operand reads below conflict-free although a pixel is only 64 bytes;
//   * per (tap, 8-channel half): four ds_read_b128 (A, one per row tile; tap / row offsets are immediates) and Q
//     buffer_load_dwordx4 (B = packed weights [tap][k/4][n][4], L2 resident, scalar offsets), one group ahead,
//     then 16*Q MFMAs (lane half h of MFMA j consumes k = 8g + 4h + j, see conv_igemm.hip).
// Epilogue straight from the accumulators: scale/shift, residual, ReLU, BN partial sums, buffer stores whose
// addresses are a per-lane constant plus scalar offsets.
#include "common.h"
#include <cstdlib>

#define FR_TH 16
#define FR_HP 48   // halo row pitch in pixels (3 DMA pieces of 16 pixels; 32+KW-1 used)

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct FrCls {
    int ymin, xmin;   // input pixel of halo (0,0) for virtual pixel (0,0)
    int tap0;         // packed-weight tap index of this class's tap (0,0)
};
struct FrArgs {
    FrCls cls[4];
    int ncls, xps;           // input-parity classes; input pixels per halo pixel
    int tap_sy, tap_sx;      // packed-weight tap index strides
    int nchunks, KQ;         // Cin / 16, Cin / 4
    int tiles_x, tiles_y, nregions, ncog;
};

template <int IDX>
__device__ __forceinline__ void fr_mfma(f32x16& c, float a, float b) {
    if constexpr (IDX < 16) asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int Q, int KK, int P, int J>
__device__ __forceinline__ void fr_group(f32x16 (&acc)[4 * Q], const f32x4 (&a)[4], const f32x4 (&b)[Q]) {
    if constexpr (KK < 4) {
        fr_mfma<P * Q + J>(acc[P * Q + J], a[P][KK], b[J][KK]);
        if constexpr (J + 1 < Q) fr_group<Q, KK, P, J + 1>(acc, a, b);
        else if constexpr (P + 1 < 4) fr_group<Q, KK, P + 1, 0>(acc, a, b);
        else fr_group<Q, KK + 1, 0, 0>(acc, a, b);
    }
}

template <int KH, int KW, bool REV, int Q>
__global__ __launch_bounds__(256, 1) void conv_rows_kernel(const adh_conv_desc d, const FrArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HR = FR_TH + KH - 1;
    constexpr int NC = 32 + KW - 1;
    constexpr int XF = HR * FR_HP * 16;      // floats per slab buffer
    constexpr int NP = HR * 3;               // DMA pieces per slab
    constexpr int NKG = KH * KW * 2;         // (tap, channel half) groups per slab

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;

    // XCD-aware decode: the output-channel groups of one region share an XCD (and the input region in its L2)
    const int bid = blockIdx.x;
    const int q_ = bid >> 3;
    const int cg = q_ % g.ncog;
    const int region = (q_ / g.ncog) * 8 + (bid & 7);
    if (region >= g.nregions) return;
    int rr = region;
    const int tx = rr % g.tiles_x;
    rr /= g.tiles_x;
    const int ty = rr % g.tiles_y;
    const int n = rr / g.tiles_y;
    const int vy0 = ty * FR_TH, vx0 = tx * 32;
    const int co0 = cg * 32 * Q;

    // ------------------------------------------------------------------ halo staging (LDS-DMA)
    const int xcs = d.in_cstride * 4 * g.xps;              // halo pixel pitch in bytes
    const int xrs = d.IW * d.in_cstride * 4 * g.xps;       // halo row pitch in bytes
    const float* const in_n = d.in + (int64_t)n * d.IH * d.IW * d.in_cstride;
    const int pxl = lane >> 2;
    const int q_lane = (lane & 3) ^ ((pxl >> 1) & 3);      // channel quad this lane fetches (swizzled slot = lane & 3)
    const int xv = pxl * xcs + q_lane * 16;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto stage = [&](int slab, int b) {
        const int chunk = slab / g.ncls, c = slab - chunk * g.ncls;
        const int ymin = g.cls[c].ymin, xmin = g.cls[c].xmin;
        // descriptor base = input pixel of halo (0,0) of this region (may lie outside the image: never dereferenced there)
        const int iy0 = vy0 * g.xps + ymin, ix0 = vx0 * g.xps + xmin;
        const float* base = in_n + ((int64_t)iy0 * d.IW + ix0) * d.in_cstride + chunk * 16;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
        float* xs = smem + b * XF;
#pragma unroll 1
        for (int j = wave; j < NP; j += 4) {
            const int r = (j * 171) >> 9, cb = j - r * 3;        // j / 3 for j < 171
            const int iy = iy0 + r * g.xps;
            const int col = cb * 16 + pxl;
            const int ix = ix0 + col * g.xps;
            const bool need = col < NC;
            const bool ok = need && iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW;
            float* dst = xs + (r * FR_HP + cb * 16) * 16;
            if (ok) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)dst, 16, xv, r * xrs + cb * 16 * xcs, 0, 0);
            if (need && !ok) *reinterpret_cast<f32x4*>(dst + lane * 4) = zero4;
        }
    };

    // ------------------------------------------------------------------ operand addressing
    // A: slab[(row)*48 + col][slot] with slot = quad ^ ((col >> 1) & 3); lane = pixel l31 of a row tile, quad = 2g' + h
    int alane[KW][2];
#pragma unroll
    for (int ox = 0; ox < KW; ++ox) {
        const int col = l31 + ox;
        const int s = (col >> 1) & 3;
        alane[ox][0] = ((wave * 4) * FR_HP + col) * 16 + ((h ^ s) * 4);
        alane[ox][1] = alane[ox][0] ^ 8;
    }
    // B: packed weights [tap][kq][n] float4; lane = column n = l31 of a tile, kq = 4*chunk + 2g' + h
    const int b_voff = (h * d.NcP + l31) * 16;
    const __amdgpu_buffer_rsrc_t b_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.wp + (int64_t)co0 * 4), 0, 0x7fffffff, 0x00020000);
    auto b_off = [&](int slab, int kg) {   // kg = tap * 2 + g'
        const int chunk = slab / g.ncls, c = slab - chunk * g.ncls;
        const int t = kg >> 1, gg = kg & 1;
        const int tap = g.cls[c].tap0 + (t / KW) * g.tap_sy + (t % KW) * g.tap_sx;
        return ((tap * g.KQ + chunk * 4 + 2 * gg) * d.NcP) * 16;
    };
    auto load_b = [&](f32x4 (&b)[Q], int soff) {
#pragma unroll
        for (int j = 0; j < Q; ++j)
            b[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_voff, soff + j * 512, 0));
    };
    auto load_a = [&](f32x4 (&a)[4], const float* xs, int kg) {
        const int t = kg >> 1, gg = kg & 1;
        const int oy = REV ? KH - 1 - t / KW : t / KW, ox = REV ? KW - 1 - t % KW : t % KW;
        const float* p = xs + alane[ox][gg] + oy * (FR_HP * 16);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) a[pt] = *reinterpret_cast<const f32x4*>(p + pt * (FR_HP * 16));
    };

    f32x16 acc[4 * Q];
#pragma unroll
    for (int t = 0; t < 4 * Q; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int nslabs = g.nchunks * g.ncls;
    f32x4 av[2][4], bv[2][Q];
    stage(0, 0);
    load_b(bv[0], b_off(0, 0));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#pragma unroll 1
    for (int s = 0; s < nslabs; ++s) {
        const float* xs = smem + (s & 1) * XF;
        if (s + 1 < nslabs) stage(s + 1, (s + 1) & 1);
        const int sn = s + 1 < nslabs ? s + 1 : s;
        load_a(av[0], xs, 0);
#pragma unroll
        for (int kg = 0; kg < NKG; ++kg) {
            // operands of the next group (the first group of the next slab for the weights) in flight
            if (kg + 1 < NKG) {
                load_b(bv[(kg + 1) & 1], b_off(s, kg + 1));
                load_a(av[(kg + 1) & 1], xs, kg + 1);
            } else {
                load_b(bv[(kg + 1) & 1], b_off(sn, 0));
            }
            fr_group<Q, 0, 0, 0>(acc, av[kg & 1], bv[kg & 1]);
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch distance at one group (register budget)
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---------------------------------------------------------------------- fused epilogue from the accumulators
    float* out_n = d.out + (int64_t)n * d.OH * d.OW * d.out_cstride;
    const float* res_n = d.residual ? d.residual + (int64_t)n * d.OH * d.OW * d.res_cstride : nullptr;
    const int ocs = d.out_cstride * 4, rcs = d.res_cstride * 4;
    const __amdgpu_buffer_rsrc_t or_ = __builtin_amdgcn_make_buffer_rsrc(out_n, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr_ =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(res_n ? res_n : d.in), 0, 0x7fffffff, 0x00020000);
    // lane = (pixel column 4h + .., channel l31); element r of a tile is pixel column (r&3) + 8*(r>>2) + 4h
    const int ovoff = (4 * h * d.out_sx) * ocs + l31 * 4, rvoff = (4 * h * d.out_sx) * rcs + l31 * 4;
    float ssum[Q], ssq[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        ssum[j] = 0.f;
        ssq[j] = 0.f;
        const int co = co0 + j * 32 + l31;   // always a real channel: the host requires Cout % 32 == 0
        const float sc = d.scale ? d.scale[co] : 1.f;
        const float sh = d.shift ? d.shift[co] : 0.f;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int vy = vy0 + wave * 4 + pt;
            const int opix0 = (vy * d.out_sy + d.out_oy) * d.OW + vx0 * d.out_sx + d.out_ox;
            float rv[16];
            if (res_n) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rr_, rvoff, (opix0 + ((r & 3) + 8 * (r >> 2)) * d.out_sx) * rcs + (co0 + j * 32) * 4, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[pt * Q + j][r] * sc + sh;
                ssum[j] += v;
                ssq[j] += v * v;
                if (res_n) v += rv[r];
                if (d.act == ADH_ACT_RELU) v = fmaxf(v, 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), or_, ovoff,
                                                      (opix0 + ((r & 3) + 8 * (r >> 2)) * d.out_sx) * ocs + (co0 + j * 32) * 4, 0);
            }
        }
    }
    if (d.stats) {
        // combine the two lane halves (same channel), then the 4 waves through LDS in a fixed order
        float* red = smem;   // [4][2][32*Q]; the slab buffers are idle (last barrier passed)
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            const float s = ssum[j] + __shfl_xor(ssum[j], 32, 64);
            const float q2 = ssq[j] + __shfl_xor(ssq[j], 32, 64);
            if (h == 0) {
                red[(wave * 2 + 0) * (32 * Q) + 32 * j + l31] = s;
                red[(wave * 2 + 1) * (32 * Q) + 32 * j + l31] = q2;
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * 32 * Q; i += 256) {
            const int which = i / (32 * Q);
            const int cl = i - which * (32 * Q);
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[(w * 2 + which) * (32 * Q) + cl];
            d.stats[((size_t)region * 2 + which) * d.NcP + co0 + cl] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct FrPlan {
    int KH, KW, rev, Q;
    FrArgs a;
};

// 1 when `d` runs on conv_rows_kernel (and fills the plan), 0 otherwise
static int rows_fwd_plan(const adh_conv_desc* d, FrPlan* p) {
    if (!d || d->NcP % 32 != 0 || d->NcP < 32) return 0;
    if (d->VW % 32 != 0 || d->VH % FR_TH != 0 || d->Cin % 16 != 0 || d->Cout != d->NcP) return 0;
    static const bool enabled = !(getenv("ADH_ROWS_FWD") && atoi(getenv("ADH_ROWS_FWD")) == 0);   // A/B switch
    if (!enabled) return 0;
    if (d->in_cstride % 4 != 0 || d->in_sy != d->in_sx || d->dstep_y != d->dstep_x || d->out_sy != d->out_sx) return 0;
    if ((int64_t)(d->IH + 40) * d->IW * d->in_cstride >= (1ll << 29) || (int64_t)d->OH * d->OW * d->out_cstride >= (1ll << 29))
        return 0;
    if (d->residual && (int64_t)d->OH * d->OW * d->res_cstride >= (1ll << 29)) return 0;
    const int nt = d->NcP / 32;
    int Q;
    if (nt % 6 == 0) Q = 6;
    else if (nt % 4 == 0) Q = 4;
    else if (nt % 3 == 0) Q = 3;
    else if (nt % 2 == 0) Q = 2;
    else Q = 1;
    FrArgs& a = p->a;
    a.nchunks = d->Cin / 16;
    a.KQ = d->Cin / 4;
    a.tiles_x = d->VW / 32;
    a.tiles_y = d->VH / FR_TH;
    a.nregions = a.tiles_x * a.tiles_y * d->N;
    a.ncog = nt / Q;
    p->Q = Q;
    const int s = d->in_sy, ds = d->dstep_y;
    if (((d->KH == 3 && d->KW == 3) || (d->KH == 2 && d->KW == 2)) && (ds == s || ds == -s) && (s == 1 || s == 2)) {
        p->KH = d->KH; p->KW = d->KW; p->rev = ds < 0;
        a.ncls = 1; a.xps = s;
        a.cls[0].ymin = d->dy0 + (ds < 0 ? (d->KH - 1) * ds : 0);
        a.cls[0].xmin = d->dx0 + (ds < 0 ? (d->KW - 1) * ds : 0);
        a.cls[0].tap0 = 0; a.tap_sy = d->KW; a.tap_sx = 1;
        for (int c = 1; c < 4; ++c) a.cls[c] = a.cls[0];
        return 1;
    }
    if (d->KH == 4 && d->KW == 4 && s == 2 && ds == 1) {
        // kernel index ky = 2*ty + py reads input row 2*(vy + ty) + dy0 + py: four 2x2-tap classes on the stride-2 grid
        p->KH = 2; p->KW = 2; p->rev = 0;
        a.ncls = 4; a.xps = 2; a.tap_sy = 8; a.tap_sx = 2;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                FrCls& c = a.cls[py * 2 + px];
                c.ymin = d->dy0 + py; c.xmin = d->dx0 + px; c.tap0 = py * 4 + px;
            }
        return 1;
    }
    return 0;
}

template <int KH, int KW, bool REV, int Q>
static int launch_rows_fwd(hipStream_t s, const adh_conv_desc* d, const FrArgs& a) {
    const int lds = 2 * (FR_TH + KH - 1) * FR_HP * 16 * 4;
    const int nblocks = ((a.nregions + 7) / 8) * a.ncog * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows_kernel<KH, KW, REV, Q>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv_rows_kernel<KH, KW, REV, Q>), dim3(nblocks), dim3(256), lds, s, *d, a);
    return adh_check_launch();
}

int adh_rows_fwd_num_blocks(const adh_conv_desc* d) {
    FrPlan p;
    return rows_fwd_plan(d, &p) ? p.a.nregions : 0;
}

// returns ADH_E_UNSUPPORTED when the descriptor is not a rows-kernel shape (the caller falls back to conv_igemm)
int adh_rows_fwd_launch(void* stream, const adh_conv_desc* d) {
    FrPlan p;
    if (!rows_fwd_plan(d, &p)) return ADH_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
#define FR_CASE(kh_, kw_, rev_, q_) \
    if (p.KH == kh_ && p.KW == kw_ && p.rev == rev_ && p.Q == q_) return launch_rows_fwd<kh_, kw_, rev_ != 0, q_>(s, d, p.a);
#define FR_CASES(kh_, kw_, rev_) \
    FR_CASE(kh_, kw_, rev_, 6) FR_CASE(kh_, kw_, rev_, 4) FR_CASE(kh_, kw_, rev_, 3) FR_CASE(kh_, kw_, rev_, 2) FR_CASE(kh_, kw_, rev_, 1)
    FR_CASES(2, 2, 0)
    FR_CASES(2, 2, 1)
    FR_CASES(3, 3, 0)
    FR_CASES(3, 3, 1)
#undef FR_CASES
#undef FR_CASE
    return ADH_E_UNSUPPORTED;
}
