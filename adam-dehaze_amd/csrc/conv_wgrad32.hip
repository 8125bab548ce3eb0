// Weight gradient of the 2x2-tap gather forms in the Winograd F(3x3, 2x2) domain, fp32 MFMA, gfx950.
//
// The 2x2-tap forms are what Conv2d k4 s2 (four kernel-parity classes on the stride-2 input grid) and every output-parity
// class of ConvTranspose2d k4 s2 decompose into (conv_wgrad.hip, wgrad_rows_plan); their weight gradients are the autograd
// backward ATen runs for those layers in the reference's training step (/root/reference models/dehazing/
// high_intensity.py:100-118, training/train_joint.py:153).  conv_wgrad_rows_kernel<2,2,..> computes them directly: 4 MACs
// per output pixel and (ci, co) pair.  With
//
//   Y(3x3) = G^T [ (A g A^T) .* (B^T d B) ] G          d: 4x4 input patch, g: 2x2 filter   (conv_wino.hip, F(3x3,2x2))
//
// the gradient with respect to U = A g A^T is  dU[f][ci][co] = sum over 3x3 tiles of (B^T d B)[f][tile][ci] * (G dY G^T)[f][tile][co]:
// 16 products per tile of 9 pixels instead of 36 -- 4/9 of the MFMA work -- and dg = A^T dU A is applied once, by the
// reduce kernel.  Same skeleton as the Winograd-domain gradient of the 3x3 layers (conv_wgrad_rows_kernel<3,3,0,TN,true>):
// 4 waves, one per SIMD, wave w owns frequency row a = w (4 frequencies x TN n-tiles = 12 accumulator tiles, all in
// AGPRs); a workgroup owns 32 input channels x 32 TN output channels and sweeps a contiguous range of pixel regions; the
// raw input halo and the raw dY tile arrive by LDS-DMA, double buffered, one barrier per region; both operands are
// transformed in registers, in the lane layout the MFMA wants (lane = channel), straight from the raw LDS images:
//   A side: r[c] = d[rA][c] + sg d[rB][c] (row a of B^T d; rows and sign are per-wave constants),
//           V = r0 - r2, r1 + r2, r2 - r1, r1 - r3                                           (4 packed VALU)
//   B side: t[j] = dY[0][j] + cc dY[1][j] + dY[2][j] for a = 1, 2 (cc = +1 / -1); t[j] = dY[0][j] / dY[2][j] for a = 0 / 3,
//           W = t0, t0 + t1 + t2, t0 - t1 + t2, t2                              (7 / 3 VALU per n-tile)
//   (row a and column b of G carry a factor 1/2 for a, b in {1, 2}: applied by the reduce kernel.)
// Region = 3 rows x 48 columns of the class grid = 16 tiles = 8 MFMA k-steps (lane half h takes tile column 2s + h): the
// largest region whose halo (4 x 49 px x 32 ch) and dY tile (3 x 48 px x 32 TN ch) double-buffer in the CU's 160 KB at
// TN = 3.  Ragged grids (the class grids are 2^k sized, 3 and 48 are not): dY pixels beyond the grid are zero-filled,
// halo pixels beyond the image too.
#include "common.h"
#include <cstring>
#include <cstdlib>
#include <type_traits>

#ifndef G32_DBG
#define G32_DBG 0   // dev builds (timing only, wrong results): 1 = stage only the first region, 2 = skip the contraction
#endif
#define G32_TH 3
#define G32_TW 48
#define G32_HR 4                                  // halo rows
#define G32_HP 49                                 // halo row pitch in pixels (the 7th DMA piece of a row writes pixel 48 only)
#define G32_XF (G32_HR * G32_HP * 32)             // floats of the halo image
#define G32_GROW (G32_TW * 32)                    // floats of one dY row of one n-tile

typedef __attribute__((address_space(3))) void* lds_void_ptr_g32;

struct Wg32Args {
    int ymin, xmin, xps;          // input pixel of halo (0,0) for virtual pixel (0,0); input pixels per halo pixel
    int ntiles, nsplit, ngroups, nco_groups, tiles_x, tiles_y;
    int cls, ncls;                // class of this launch / classes of the layer: slab[split][cls][16][KP][NcP]
};

struct Wg32TileGeom {             // wave-uniform description of one staged region
    __amdgpu_buffer_rsrc_t xr, gr;
    int iy0, ix0, so, sg0, vy0, vx0;
    bool interior, g_full;
};

template <int I, int TN>
__device__ __forceinline__ void g32_mfma_all(f32x16 (&acc)[4 * TN], const float (&V)[4], const float (&M)[TN][4]) {
    if constexpr (I < 4 * TN) {
        // (s_nop 1: two wait states between the VALU instruction that wrote an operand and the MFMA -- see adh_mfma_operand_fence)
        asm("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[I]) : "v"(V[I / TN]), "v"(M[I % TN][I / TN]));
        g32_mfma_all<I + 1, TN>(acc, V, M);
    }
}

template <int TN>
__global__ __launch_bounds__(256, 1) void conv_wgrad32_kernel(const adh_conv_desc d, const Wg32Args g, float* slab) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int GF = TN * G32_TH * G32_GROW;     // floats of the dY image [TN][3][48][32]
    constexpr int BUF = G32_XF + GF;
    constexpr int NGP = 18 * TN;                   // DMA pieces of the dY image (8 pixels x 32 channels each)
    constexpr int GU = (NGP + 3) / 4;              // ... per wave (at most)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;

    // XCD-aware decode (conv_wgrad_rows_kernel): the workgroups that share one pixel range land on one XCD
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int q = bid >> 3;
    const int grp = q % g.ngroups;
    const int split = (q / g.ngroups) * 8 + xcd;
    if (split >= g.nsplit) return;
    const int ci_tile = grp / g.nco_groups;
    const int co_grp = grp - ci_tile * g.nco_groups;
    const int ci0 = ci_tile * 32;
    const int co0 = co_grp * 32 * TN;

    const int per = g.ntiles / g.nsplit, rem = g.ntiles - per * g.nsplit;
    const int t_begin = split * per + adh_min_i(split, rem);
    const int t_end = t_begin + per + (split < rem ? 1 : 0);

    // descriptors are per image; the halo descriptor's base is pixel (ymin, xmin) of the image, so every scalar offset is
    // non-negative; lanes that would fall outside the tensor are never issued
    const int xcs = d.in_cstride * 4 * g.xps, gcs = d.out_cstride * 4 * d.out_sx;
    const int xrs = d.IW * d.in_cstride * 4 * g.xps, grs = d.OW * d.out_cstride * 4 * d.out_sy;
    const float* xbase = d.in + ((int64_t)g.ymin * d.IW + g.xmin) * d.in_cstride + ci0;
    const float* gbase = d.out + ((int64_t)d.out_oy * d.OW + d.out_ox) * d.out_cstride + co0;
    const int64_t ximg = (int64_t)d.IH * d.IW * d.in_cstride, gimg = (int64_t)d.OH * d.OW * d.out_cstride;
    const int xv = (lane >> 3) * xcs + (lane & 7) * 16;    // per-lane byte offsets inside one 8-pixel piece
    const int gv = (lane >> 3) * gcs + (lane & 7) * 16;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const bool g_partial = co0 + 32 * TN > d.Cout;   // this group's last n-tile has channels beyond Cout

    auto tile_geom = [&](int n, int ty, int tx) {
        Wg32TileGeom t;
        t.xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xbase + n * ximg), 0, 0x7fffffff, 0x00020000);
        t.gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gbase + n * gimg), 0, 0x7fffffff, 0x00020000);
        t.vy0 = ty * G32_TH;
        t.vx0 = tx * G32_TW;
        t.iy0 = t.vy0 * g.xps + g.ymin;
        t.ix0 = t.vx0 * g.xps + g.xmin;
        t.so = t.vy0 * xrs + t.vx0 * xcs;
        t.sg0 = t.vy0 * grs + t.vx0 * gcs;
        t.interior = t.iy0 >= 0 && t.iy0 + (G32_HR - 1) * g.xps < d.IH && t.ix0 >= 0 && t.ix0 + (G32_HP - 1) * g.xps < d.IW;
        t.g_full = t.vy0 + G32_TH <= d.VH && t.vx0 + G32_TW <= d.VW;
        return t;
    };
    // stage region t into buffer b: this wave's share of the DMA pieces (28 of the halo, 18 TN of dY)
    auto stage = [&](const Wg32TileGeom& t, int b) {
        float* xs = smem + b * BUF;
        float* gs = xs + G32_XF;
#pragma unroll
        for (int u = 0; u < 7; ++u) {
            const int j = wave + 4 * u;                           // < 28
            const int pr = (j * 37) >> 8, pc = j - 7 * pr;        // j / 7, j % 7
            float* dst = xs + (pr * G32_HP + pc * 8) * 32;
            const int po = t.so + pr * xrs + pc * 8 * xcs;
            const int c = pc * 8 + (lane >> 3);                   // halo column of this lane's pixel
            if (t.interior) {
                if (pc == 6) {
                    if (c < G32_HP) __builtin_amdgcn_raw_ptr_buffer_load_lds(t.xr, (lds_void_ptr_g32)dst, 16, xv, po, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(t.xr, (lds_void_ptr_g32)dst, 16, xv, po, 0, 0);
                }
            } else {
                const int iy = t.iy0 + pr * g.xps, ix = t.ix0 + c * g.xps;
                const bool ok = c < G32_HP && iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW;
                if (ok) __builtin_amdgcn_raw_ptr_buffer_load_lds(t.xr, (lds_void_ptr_g32)dst, 16, xv, po, 0, 0);
                if (!ok && c < G32_HP) *reinterpret_cast<f32x4*>(dst + lane * 4) = zero4;
            }
        }
#pragma unroll
        for (int u = 0; u < GU; ++u) {
            const int i = wave + 4 * u;
            if (i < NGP) {
                const int tn = i / 18, r18 = i - 18 * tn, r = r18 / 6, cb = r18 - 6 * r;
                float* dst = gs + ((tn * G32_TH + r) * G32_TW + cb * 8) * 32;
                const int po = t.sg0 + r * grs + cb * 8 * gcs + tn * 128;
                const bool ch_ok = !g_partial || co0 + tn * 32 + (lane & 7) * 4 < d.Cout;
                if (t.g_full) {
                    if (ch_ok) __builtin_amdgcn_raw_ptr_buffer_load_lds(t.gr, (lds_void_ptr_g32)dst, 16, gv, po, 0, 0);
                } else {
                    const bool ok = t.vy0 + r < d.VH && t.vx0 + cb * 8 + (lane >> 3) < d.VW;
                    if (ok && ch_ok) __builtin_amdgcn_raw_ptr_buffer_load_lds(t.gr, (lds_void_ptr_g32)dst, 16, gv, po, 0, 0);
                    if (!ok) *reinterpret_cast<f32x4*>(dst + lane * 4) = zero4;
                }
            }
        }
    };
    if (g_partial) {   // the masked channel quads of a partial n-tile keep these zeros
        for (int i = tid; i < 2 * BUF / 4; i += 256) reinterpret_cast<f32x4*>(smem)[i] = zero4;
        __syncthreads();
    }

    f32x16 acc[4 * TN];
#pragma unroll
    for (int t = 0; t < 4 * TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    int tx = t_begin % g.tiles_x;
    int ty = (t_begin / g.tiles_x) % g.tiles_y;
    int n = t_begin / (g.tiles_x * g.tiles_y);
    if (t_begin < t_end) stage(tile_geom(n, ty, tx), 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // per-wave constants of the operand transforms (frequency row a = wave)
    const int rA = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int rB = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
    const float sg = wave == 1 ? 1.f : -1.f;
    const float k0 = wave == 3 ? 0.f : 1.f, k1 = wave == 1 ? 1.f : (wave == 2 ? -1.f : 0.f), k2 = wave == 0 ? 0.f : 1.f;   // row a of G (x2 for a = 1, 2)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef const volatile __attribute__((address_space(3))) float* lds_vf;   // one ds_read_b32 per element, 16-bit immediates

    int cur = 0;
    for (int tile = t_begin; tile < t_end; ++tile) {
        int ntx = tx + 1, nty = ty, nn = n;
        if (ntx == g.tiles_x) { ntx = 0; ++nty; }
        if (nty == g.tiles_y) { nty = 0; ++nn; }
        if (!(G32_DBG & 1) && tile + 1 < t_end) stage(tile_geom(nn, nty, ntx), cur ^ 1);

        // lane (h, l31): tile column 2 s + h in k-step s, channel l31; patch / tile column c of step s sits 6 s + c pixels
        // behind the lane's base (3 h pixels into the row)
        lds_vf xa = (lds_vf)(smem + cur * BUF + (rA * G32_HP + 3 * h) * 32 + l31);
        lds_vf xb = (lds_vf)(smem + cur * BUF + (rB * G32_HP + 3 * h) * 32 + l31);
        lds_vf g0 = (lds_vf)(smem + cur * BUF + G32_XF + (3 * h) * 32 + l31);
        f32x2 va[2][2], vb[2][2];
        auto ld_a = [&](int s, f32x2 (&a)[2], f32x2 (&b)[2]) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                a[c] = f32x2{xa[(6 * s + 2 * c) * 32], xa[(6 * s + 2 * c + 1) * 32]};
                b[c] = f32x2{xb[(6 * s + 2 * c) * 32], xb[(6 * s + 2 * c + 1) * 32]};
            }
        };
        auto a_side = [&](const f32x2 (&a)[2], const f32x2 (&b)[2], float (&V)[4]) {
            const f32x2 r01 = sg * b[0] + a[0];
            const f32x2 r23 = sg * b[1] + a[1];
            f32x2 v01, v32;   // (V0, V1) = (r0 - r2, r1 + r2), (V3, -V2) = (r1 - r3, r1 - r2)
            asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(v01) : "v"(r01), "v"(r23));
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(v32) : "v"(r01), "v"(r23));
            V[0] = v01.x;
            V[1] = v01.y;
            V[2] = v32.y;   // -V2: the B side hands over -W2
            V[3] = v32.x;
        };
        // One instruction stream for all four waves: t[j] = k0 dY[0][j] + k1 dY[1][j] + k2 dY[2][j] with the wave's row of G as
        // (k0, k1, k2) = (1,0,0), (1,1,1), (1,-1,1), (0,0,1).  (Waves 0 and 3 could read one row instead of three, but the
        // waves meet at the barrier anyway, and a run-time branch around the MFMA loops makes hipcc move the accumulators
        // between register classes at the join: 200 v_accvgpr_write per region, measured.)
        f32x2 p[2][3][TN];
        float qv[2][3][TN];
        auto ld_b = [&](int s, f32x2 (&pp)[3][TN], float (&qq)[3][TN]) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    pp[r][j] = f32x2{g0[((j * G32_TH + r) * G32_TW + 6 * s) * 32], g0[((j * G32_TH + r) * G32_TW + 6 * s + 1) * 32]};
                    qq[r][j] = g0[((j * G32_TH + r) * G32_TW + 6 * s + 2) * 32];
                }
        };
        ld_a(0, va[0], vb[0]);
        ld_b(0, p[0], qv[0]);
#pragma unroll
        for (int s = 0; s < ((G32_DBG & 2) ? 0 : 8); ++s) {
            if (s + 1 < 8) {
                ld_a(s + 1, va[(s + 1) & 1], vb[(s + 1) & 1]);
                ld_b(s + 1, p[(s + 1) & 1], qv[(s + 1) & 1]);
            }
            float V[4], M[TN][4];
            a_side(va[s & 1], vb[s & 1], V);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x2 t01 = k0 * p[s & 1][0][j];
                t01 = k1 * p[s & 1][1][j] + t01;
                t01 = k2 * p[s & 1][2][j] + t01;
                float t2 = k0 * qv[s & 1][0][j];
                t2 = k1 * qv[s & 1][1][j] + t2;
                t2 = k2 * qv[s & 1][2][j] + t2;
                const float sm = t01.x + t2;
                M[j][0] = t01.x;
                M[j][1] = sm + t01.y;
                M[j][2] = t01.y - sm;   // -W2
                M[j][3] = t2;
            }
            g32_mfma_all<0, TN>(acc, V, M);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
        tx = ntx; ty = nty; n = nn;
    }

    // partial result -> slab[split][cls][f = 4*wave + b][KP][NcP]
    const int KP = d.Cin;
    float* sbase = slab + (((size_t)split * g.ncls + g.cls) * 16 + wave * 4) * KP * d.NcP;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float* base = sbase + ((size_t)b * KP + ci0) * d.NcP + co0 + 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                base[(size_t)i * d.NcP] = acc[b * TN + j][r];
            }
        }
}

// =====================================================================================================================
// Round 3: conv_wgrad32v2_kernel -- the same F(3x3,2x2)-domain gradient with the design of conv_wgrad_wino43_kernel
// (conv_wgrad43.hip): the transformed operands go through LDS ONCE as V[16 frequencies][tile][64 ci | 96 co] and feed the
// MFMAs by plain ds_read_b32, instead of being re-derived from the raw images in registers by every wave (31 VALU per 12
// MFMAs in the MFMA stream, and 82 LDS-DMA pieces per 96 MFMAs and wave: the kernel above is bound by those, 0.41 of the
// MFMA peak).  Workgroup = 64 input x 96 output channels x all 16 frequencies; wave w owns frequency row a = w: 4
// frequencies x 2 x 3 channel tiles = 24 accumulator tiles (16 in AGPRs).  Strip = 8 tiles of 3x3 (3 rows x 24 pixels of the
// class grid, halo 4 x 25): 16 contraction steps of 6 MFMAs; 55 LDS-DMA pieces per strip (13.75 per wave and 96 MFMAs), one
// behind the first MFMA of steps 0..13; one-pass register transforms, thread = (tile, channel quad, frequency row),
// 20 wave-rounds per strip (8 of x, 12 of dY), five per wave.  Conflict-free LDS images without padding: an x pixel is 256 B
// (the 16 lanes of a ds_read_b128 group = its 16 quads), a dY pixel 384 B = 96 banks = 32 mod 64 (adjacent tiles, 3 pixels
// apart, land in the other bank half).  Shapes: Cin % 64 == 0, Cout % 96 == 0 (ConvTranspose 384 -> 96 / 192, Conv2d k4 s2
// 192 -> 384) and, with the roles of the two layouts exchanged (template parameter CI = 96: 96 input x 64 output channels,
// 58 pieces per strip), Cin % 96 == 0, Cout % 64 == 0 (Conv2d k4 s2 96 -> 192); everything else stays on the kernel above / the
// direct kernel.
// =====================================================================================================================
#ifndef H2_DBG
#define H2_DBG 0   // dev builds (timing only): 1 = skip the transform, 4 = skip the contraction, 8 = stage only the first strip
#endif
#define H2_T 8
#define H2_VPLANE (H2_T * 160)                    //  1,280 floats per frequency: [tile][CI input | CO output channels], CI + CO = 160
#define H2_V_F (16 * H2_VPLANE)                   // 20,480
#define H2_TAB_F (4 * 64)
// Channel split of the workgroup: CI = 64 input x 96 output channels (2 x 3 channel tiles) or CI = 96 x 64 (3 x 2: Conv2d k4 s2
// 96 -> 192, whose 96 input channels do not divide by 64).  The operand with 64 channels has 256-byte pixels (DMA piece = 4
// pixels x 16 quads, transform rounds of 4 tiles x 16 quads), the one with 96 channels 384-byte pixels (three piece patterns
// repeating every 8 pixels, rounds of 8 tiles x 8 quads per 32-channel tile).
template <int CI>
struct H2L {
    static constexpr int CO = 160 - CI;
    static constexpr int MI = CI / 32, NJ = CO / 32;
    static constexpr int XROW = 28 * CI;              // floats per raw halo row: 28 pixel slots (25 used)
    static constexpr int RAWX_F = 4 * XROW;
    static constexpr int GROW = 24 * CO;              // floats per raw dY row
    static constexpr int RAWG_F = 3 * GROW;
    static constexpr int NXP = CI == 64 ? 7 : 10;     // DMA pieces per raw x row
    static constexpr int NGP = CO == 64 ? 6 : 9;      // ... per raw dY row
    static constexpr int NSTEP = (4 * NXP + 3 * NGP + 3) / 4;   // contraction steps that carry a piece: 14 / 15 (of 16)
    static constexpr int LDS_BYTES = (RAWX_F + RAWG_F + H2_V_F + H2_TAB_F) * 4;   // 139,264 B / 144,384 B
};
static_assert(H2L<64>::MI * H2L<64>::NJ == 6 && H2L<96>::MI * H2L<96>::NJ == 6, "six channel-tile pairs per frequency");

struct Wg32v2Args {
    int ymin[4], xmin[4], xps;   // halo origin per class
    int S, SX, TY, nsplit, ngroups, ncob;
    int cls, ncls;               // class of this launch (cblocks == 0) ...
    int cblocks;                 // ... or > 0: all ncls classes in one grid, workgroups [c * cblocks, (c + 1) * cblocks) = class c
    int64_t gofs[4];             // float offset of dY's pixel (out_oy, out_ox) per class (the class descriptors of a transposed layer differ in it)
};

template <int IDX>
__device__ __forceinline__ void h2_mfma(f32x16& c, float a, float b) {
    if constexpr (IDX < 16) asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
typedef const volatile __attribute__((address_space(3))) float* h2_lds_vf;   // one ds_read_b32 per operand, 16-bit immediates
// step ST = b * 4 + ks: frequency b of the wave's row, tile pair ks
template <int CI, int ST>
__device__ __forceinline__ void h2_load_ops(h2_lds_vf vlane, float (&a)[H2L<CI>::MI], float (&b)[H2L<CI>::NJ]) {
    h2_lds_vf p = vlane + (ST / 4) * H2_VPLANE + (ST % 4) * 320;
#pragma unroll
    for (int m = 0; m < H2L<CI>::MI; ++m) a[m] = p[32 * m];
#pragma unroll
    for (int j = 0; j < H2L<CI>::NJ; ++j) b[j] = p[CI + 32 * j];
}
template <int CI, int B6, int E>
__device__ __forceinline__ void h2_mfma_rest(f32x16 (&acc)[24], const float (&a)[H2L<CI>::MI], const float (&b)[H2L<CI>::NJ]) {
    if constexpr (E < 6) {
        h2_mfma<B6 + E>(acc[B6 + E], a[E / H2L<CI>::NJ], b[E % H2L<CI>::NJ]);
        h2_mfma_rest<CI, B6, E + 1>(acc, a, b);
    }
}
template <int CI, int ST, typename Hook>
__device__ __forceinline__ void h2_contract(f32x16 (&acc)[24], h2_lds_vf vlane, float (&a)[2][H2L<CI>::MI], float (&b)[2][H2L<CI>::NJ],
                                            Hook&& hook) {
    if constexpr (ST < 16) {
        if constexpr (ST + 1 < 16) h2_load_ops<CI, ST + 1>(vlane, a[(ST + 1) & 1], b[(ST + 1) & 1]);
        constexpr int B6 = (ST / 4) * 6;                          // accumulator (frequency b, ci tile m, co tile j) = b * 6 + m * NJ + j
        h2_mfma<B6 + 0>(acc[B6 + 0], a[ST & 1][0], b[ST & 1][0]);
        hook(std::integral_constant<int, ST>{});
        h2_mfma_rest<CI, B6, 1>(acc, a[ST & 1], b[ST & 1]);
        h2_contract<CI, ST + 1>(acc, vlane, a, b, hook);
    }
}

// x round: frequency row A of B^T d B for one group R of (tile, channel quad) units
//   B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
// 64-channel layout: R = tile half (tiles 4 R .. 4 R + 3 x 16 quads); 96-channel layout: R = 32-channel tile (8 tiles x 8 quads).
// A round is split into begin (its LDS reads) and finish (arithmetic + writes) so that a wave's five rounds can be software
// pipelined: the reads of round i + 1 are in flight while round i computes (h2_five_rounds).  Unpipelined, the transform phase
// was LDS-latency-bound: 1.0 ms of a 4.7 ms launch for ~130 instructions per wave and strip.
template <int CI, int A, int R>
struct H2X {
    f32x4 dA[4], dB[4];
    float* dst;
    __device__ __forceinline__ void begin(const float* rawx, const float* /*rawg*/, float* V, int lane) {
        constexpr int rA = A == 0 ? 0 : (A == 2 ? 2 : 1), rB = A == 0 ? 2 : (A == 1 ? 2 : (A == 2 ? 1 : 3));
        const int tile = CI == 64 ? 4 * R + (lane >> 4) : (lane >> 3);
        const int ch = CI == 64 ? (lane & 15) * 4 : R * 32 + (lane & 7) * 4;
        const float* s = rawx + 3 * tile * CI + ch;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            dA[c] = *reinterpret_cast<const f32x4*>(s + rA * H2L<CI>::XROW + c * CI);
            dB[c] = *reinterpret_cast<const f32x4*>(s + rB * H2L<CI>::XROW + c * CI);
        }
        dst = V + (4 * A) * H2_VPLANE + tile * 160 + ch;
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void finish(float m1) {
        f32x4 r[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = A == 1 ? dA[c] + dB[c] : adh_pksub(dA[c], dB[c], m1);
        *reinterpret_cast<f32x4*>(dst) = adh_pksub(r[0], r[2], m1);
        *reinterpret_cast<f32x4*>(dst + 1 * H2_VPLANE) = r[1] + r[2];
        *reinterpret_cast<f32x4*>(dst + 2 * H2_VPLANE) = adh_pksub(r[2], r[1], m1);
        *reinterpret_cast<f32x4*>(dst + 3 * H2_VPLANE) = adh_pksub(r[1], r[3], m1);
        __builtin_amdgcn_sched_barrier(0);
    }
};
// dY round: frequency row A of G' dY G'^T with G' = [[1,0,0],[1,1,1],[1,-1,1],[0,0,1]] for one group R of units (as above, by the
// layout of the OUTPUT channels); rows / columns 1, 2 of G are these halved: the factor is applied by the reduce kernel
template <int CI, int A, int R>
struct H2G {
    static constexpr int CO = H2L<CI>::CO;
    f32x4 y[(A == 0 || A == 3) ? 1 : 3][3];
    float* dst;
    __device__ __forceinline__ void begin(const float* /*rawx*/, const float* rawg, float* V, int lane) {
        const int tile = CO == 64 ? 4 * R + (lane >> 4) : (lane >> 3);
        const int ch = CO == 64 ? (lane & 15) * 4 : R * 32 + (lane & 7) * 4;
        const float* s = rawg + 3 * tile * CO + ch;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if constexpr (A == 0) y[0][c] = *reinterpret_cast<const f32x4*>(s + c * CO);
            else if constexpr (A == 3) y[0][c] = *reinterpret_cast<const f32x4*>(s + 2 * H2L<CI>::GROW + c * CO);
            else {
                y[0][c] = *reinterpret_cast<const f32x4*>(s + c * CO);
                y[1][c] = *reinterpret_cast<const f32x4*>(s + H2L<CI>::GROW + c * CO);
                y[2][c] = *reinterpret_cast<const f32x4*>(s + 2 * H2L<CI>::GROW + c * CO);
            }
        }
        dst = V + (4 * A) * H2_VPLANE + tile * 160 + CI + ch;
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void finish(float m1) {
        f32x4 t[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if constexpr (A == 0 || A == 3) t[c] = y[0][c];
            else {
                const f32x4 e = y[0][c] + y[2][c];
                t[c] = A == 1 ? e + y[1][c] : adh_pksub(e, y[1][c], m1);
            }
        }
        const f32x4 e = t[0] + t[2];
        *reinterpret_cast<f32x4*>(dst) = t[0];
        *reinterpret_cast<f32x4*>(dst + 1 * H2_VPLANE) = e + t[1];
        *reinterpret_cast<f32x4*>(dst + 2 * H2_VPLANE) = adh_pksub(e, t[1], m1);
        *reinterpret_cast<f32x4*>(dst + 3 * H2_VPLANE) = t[2];
        __builtin_amdgcn_sched_barrier(0);
    }
};
template <typename R0, typename R1, typename R2, typename R3, typename R4>
__device__ __forceinline__ void h2_five_rounds(const float* rawx, const float* rawg, float* V, int lane, float m1) {
    R0 r0; r0.begin(rawx, rawg, V, lane);
    R1 r1; r1.begin(rawx, rawg, V, lane);
    r0.finish(m1);
    R2 r2; r2.begin(rawx, rawg, V, lane);
    r1.finish(m1);
    R3 r3; r3.begin(rawx, rawg, V, lane);
    r2.finish(m1);
    R4 r4; r4.begin(rawx, rawg, V, lane);
    r3.finish(m1);
    r4.finish(m1);
}

template <int CI>
__device__ __forceinline__ void h2_workgroup(const adh_conv_desc& d, const Wg32v2Args& g, float* __restrict__ slab, float* smem) {
    typedef H2L<CI> L;
    constexpr int CO = L::CO, MI = L::MI, NJ = L::NJ;
    // one grid for all classes (Conv2d k4 s2: four kernel-parity classes): 4 x cblocks workgroups pack into whole rounds of the
    // chip where four launches of one (not quite full) round each do not
    const int cls = g.cblocks > 0 ? (int)blockIdx.x / g.cblocks : g.cls;
    const int bid = (int)blockIdx.x - (g.cblocks > 0 ? cls * g.cblocks : 0);
    const int ymin = g.ymin[cls], xmin = g.xmin[cls];
    const int q2 = bid >> 3;
    const int grp = q2 % g.ngroups;
    const int split = (q2 / g.ngroups) * 8 + (bid & 7);
    if (split >= g.nsplit) return;
    float* const rawx = smem;
    float* const rawg = smem + L::RAWX_F;
    float* const V = rawg + L::RAWG_F;
    int* const tab = reinterpret_cast<int*>(V + H2_V_F);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;
    const int cob = grp % g.ncob, cib = grp / g.ncob;
    const int ci0 = cib * CI, co0 = cob * CO;

    const int xcs = d.in_cstride * 4 * g.xps, gcs = d.out_cstride * 4 * d.out_sx;
    const int xrs = d.IW * d.in_cstride * 4 * g.xps, grs = d.OW * d.out_cstride * 4 * d.out_sy;
    const float* xbase = d.in + ((int64_t)ymin * d.IW + xmin) * d.in_cstride + ci0;
    const float* gbase = d.out + g.gofs[cls] + co0;
    const int64_t ximg = (int64_t)d.IH * d.IW * d.in_cstride, gimg = (int64_t)d.OH * d.OW * d.out_cstride;
    // per-lane global offsets of the pieces.  64-channel operand: piece = 4 pixels x 16 quads, one pattern.  96-channel operand: a
    // piece is 1 KB of 384-byte pixels, three patterns; pieces k, k + 3 are 8 pixels apart
    const int cs64 = CI == 64 ? xcs : gcs, cs96 = CI == 64 ? gcs : xcs;
    if (tid < 64) {
        tab[tid] = (tid >> 4) * cs64 + (tid & 15) * 16;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int b = 1024 * k + 16 * tid;
            tab[(1 + k) * 64 + tid] = (b / 384) * cs96 + (b % 384);
        }
    }
    __syncthreads();
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_ptr_g32)smem;
    const float m1 = adh_opaque(-1.f);

    struct Pat { int p64, p96[3]; };
    auto load_patterns = [&]() {
        typedef const volatile __attribute__((address_space(3))) int* lds_vi;
        lds_vi t = (lds_vi)(tab + lane);
        Pat p;
        p.p64 = t[0]; p.p96[0] = t[64]; p.p96[1] = t[128]; p.p96[2] = t[192];
        return p;
    };
    struct Geom { __amdgpu_buffer_rsrc_t xr, gr; int vy0, vx0, iy0, ix0; unsigned xb, gb; bool interior; };
    auto strip_geom = [&](int s) {
        Geom t;
        int r = s;
        const int sx = r % g.SX;
        r /= g.SX;
        const int ty = r % g.TY;
        const int n = r / g.TY;
        t.xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xbase + n * ximg), 0, 0x7fffffff, 0x00020000);
        t.gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gbase + n * gimg), 0, 0x7fffffff, 0x00020000);
        t.vy0 = ty * 3;
        t.vx0 = sx * 24;
        t.iy0 = t.vy0 * g.xps + ymin;
        t.ix0 = t.vx0 * g.xps + xmin;
        t.xb = __builtin_amdgcn_readfirstlane((unsigned)(t.vy0 * xrs + t.vx0 * xcs));
        t.gb = __builtin_amdgcn_readfirstlane((unsigned)(t.vy0 * grs + t.vx0 * gcs));
        t.interior = t.iy0 >= 0 && t.iy0 + 3 * g.xps < d.IH && t.ix0 >= 0 && t.ix0 + 27 * g.xps < d.IW &&
                     t.vy0 + 3 <= d.VH && t.vx0 + 24 <= d.VW;
        return t;
    };
    // piece pc of raw x row `row` / piece k of raw dY row `row` (first pixel of the piece: 4 pc resp. 8 (pc / 3) + pattern)
    auto dma_x = [&](const Geom& t, const int row, const int pc, const int pat) {
        const unsigned ldsa = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(row * L::XROW + pc * 256) * 4);
        const unsigned so = __builtin_amdgcn_readfirstlane(t.xb + row * xrs + (CI == 64 ? pc * 4 : (pc / 3) * 8) * xcs);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(t.xr, (lds_void_ptr_g32)(uintptr_t)ldsa, 16, pat, so, 0, 0);
    };
    auto dma_g = [&](const Geom& t, const int row, const int k, const int pat) {
        const unsigned ldsa = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(L::RAWX_F + row * L::GROW + k * 256) * 4);
        const unsigned so = __builtin_amdgcn_readfirstlane(t.gb + row * grs + (CO == 64 ? k * 4 : (k / 3) * 8) * gcs);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(t.gr, (lds_void_ptr_g32)(uintptr_t)ldsa, 16, pat, so, 0, 0);
    };
    // piece of contraction step ST on this wave; the pattern of a 96-channel piece is ST % 3 on every wave (compile time).
    // CI = 64: waves 0..2: dY row w pieces 0..5, x row w pieces 0..6, then x row 3 piece 5 / 6 (waves 0 / 1); wave 3: dY pieces 6..8 of
    //          rows 0..2, then x row 3 pieces 0..4 (steps 0..13).
    // CI = 96: wave w: x row w pieces 0..9 (steps 0..9), then the 18 dY pieces dealt e = w + 4 j (steps 10 + j, j = 0..4).
    auto stage_piece = [&](auto stc, const Geom& t, const Pat& p) {
        constexpr int ST = decltype(stc)::value;
        const int p96 = p.p96[ST % 3];
        if constexpr (CI == 64) {
            if constexpr (ST < 6) {
                if (wave < 3) dma_g(t, wave, ST, p96);
                else dma_g(t, ST / 3, 6 + ST % 3, p96);
            } else if constexpr (ST < 9) {
                if (wave < 3) dma_x(t, wave, ST - 6, p.p64);
                else dma_g(t, 2, 6 + ST % 3, p96);
            } else if constexpr (ST < 13) {
                if (wave < 3) dma_x(t, wave, ST - 6, p.p64);
                else dma_x(t, 3, ST - 9, p.p64);
            } else if constexpr (ST == 13) {
                if (wave == 0) dma_x(t, 3, 5, p.p64);
                else if (wave == 1) dma_x(t, 3, 6, p.p64);
                else if (wave == 3) dma_x(t, 3, 4, p.p64);
            }
        } else {
            if constexpr (ST < 10) {
                dma_x(t, wave, ST, p96);
            } else if constexpr (ST < 15) {
                const int e = wave + 4 * (ST - 10);
                if (e < 18) dma_g(t, e / 6, e % 6, p.p64);
            }
        }
    };
    auto stage_interior = [&](const Geom& t) {
        const Pat p = load_patterns();
        stage_piece(std::integral_constant<int, 0>{}, t, p);
        stage_piece(std::integral_constant<int, 1>{}, t, p);
        stage_piece(std::integral_constant<int, 2>{}, t, p);
        stage_piece(std::integral_constant<int, 3>{}, t, p);
        stage_piece(std::integral_constant<int, 4>{}, t, p);
        stage_piece(std::integral_constant<int, 5>{}, t, p);
        stage_piece(std::integral_constant<int, 6>{}, t, p);
        stage_piece(std::integral_constant<int, 7>{}, t, p);
        stage_piece(std::integral_constant<int, 8>{}, t, p);
        stage_piece(std::integral_constant<int, 9>{}, t, p);
        stage_piece(std::integral_constant<int, 10>{}, t, p);
        stage_piece(std::integral_constant<int, 11>{}, t, p);
        stage_piece(std::integral_constant<int, 12>{}, t, p);
        stage_piece(std::integral_constant<int, 13>{}, t, p);
        stage_piece(std::integral_constant<int, 14>{}, t, p);
    };
    // border / ragged strip: zero both raw images, then load only what lies inside the image / the class grid
    auto stage_border = [&](const int sb) {
        const Geom t = strip_geom(sb);
        int ln = lane, td = tid;
        asm volatile("" : "+v"(ln), "+v"(td));        // keeps LICM from hoisting this rare path's lane predicates out of the loop
        const Pat p = load_patterns();
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = td; i < (L::RAWX_F + L::RAWG_F) / 4; i += 256) *reinterpret_cast<f32x4*>(smem + i * 4) = z;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            const int row = wave;                         // x row w
            const int iy = t.iy0 + row * g.xps;
#pragma unroll 1
            for (int pc = 0; pc < L::NXP; ++pc) {
                const int c = CI == 64 ? pc * 4 + (ln >> 4) : (1024 * (pc % 3) + 16 * ln) / 384 + 8 * (pc / 3);
                const int ix = t.ix0 + c * g.xps;
                if (c < 25 && iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW)
                    dma_x(t, row, pc, CI == 64 ? p.p64 : (pc % 3 == 0 ? p.p96[0] : (pc % 3 == 1 ? p.p96[1] : p.p96[2])));
            }
        }
        if (wave < 3) {                                   // dY row w
            const int row = wave;
#pragma unroll
            for (int k = 0; k < L::NGP; ++k) {
                const int px = CO == 64 ? k * 4 + (ln >> 4) : (1024 * (k % 3) + 16 * ln) / 384 + 8 * (k / 3);
                if (t.vy0 + row < d.VH && t.vx0 + px < d.VW) dma_g(t, row, k, CO == 64 ? p.p64 : p.p96[k % 3]);
            }
        }
    };

    // ------------------------------------------------------------------ transform: 20 rounds, five per wave
    auto transform = [&]() {
        if constexpr (CI == 64) {
            if (wave == 0) h2_five_rounds<H2X<64, 0, 0>, H2X<64, 0, 1>, H2G<64, 0, 0>, H2G<64, 1, 1>, H2G<64, 2, 2>>(rawx, rawg, V, lane, m1);
            else if (wave == 1) h2_five_rounds<H2X<64, 1, 0>, H2X<64, 1, 1>, H2G<64, 1, 0>, H2G<64, 2, 1>, H2G<64, 3, 2>>(rawx, rawg, V, lane, m1);
            else if (wave == 2) h2_five_rounds<H2X<64, 2, 0>, H2X<64, 2, 1>, H2G<64, 2, 0>, H2G<64, 3, 1>, H2G<64, 0, 2>>(rawx, rawg, V, lane, m1);
            else h2_five_rounds<H2X<64, 3, 0>, H2X<64, 3, 1>, H2G<64, 3, 0>, H2G<64, 0, 1>, H2G<64, 1, 2>>(rawx, rawg, V, lane, m1);
        } else {   // three x rounds (one per 32-channel tile) of the wave's own frequency row, one heavy + one light dY round
            if (wave == 0) h2_five_rounds<H2X<96, 0, 0>, H2X<96, 0, 1>, H2X<96, 0, 2>, H2G<96, 1, 0>, H2G<96, 0, 0>>(rawx, rawg, V, lane, m1);
            else if (wave == 1) h2_five_rounds<H2X<96, 1, 0>, H2X<96, 1, 1>, H2X<96, 1, 2>, H2G<96, 1, 1>, H2G<96, 0, 1>>(rawx, rawg, V, lane, m1);
            else if (wave == 2) h2_five_rounds<H2X<96, 2, 0>, H2X<96, 2, 1>, H2X<96, 2, 2>, H2G<96, 2, 0>, H2G<96, 3, 0>>(rawx, rawg, V, lane, m1);
            else h2_five_rounds<H2X<96, 3, 0>, H2X<96, 3, 1>, H2X<96, 3, 2>, H2G<96, 2, 1>, H2G<96, 3, 1>>(rawx, rawg, V, lane, m1);
        }
    };

    f32x16 acc[24];
#pragma unroll
    for (int t = 0; t < 24; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    h2_lds_vf const vlane = (h2_lds_vf)(V + (4 * wave) * H2_VPLANE + h * 160 + l31);

    const int per = (g.S + g.nsplit - 1) / g.nsplit;
    int s = split * per;
    const int s_end = adh_min_i(s + per, g.S);
    if (s < s_end) {
        const Geom t = strip_geom(s);
        if (t.interior) stage_interior(t);
        else stage_border(s);
    }
#pragma unroll 1
    while (s < s_end) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // strip s landed; every wave is past the previous contraction (V is free)
        if (!(H2_DBG & 1)) transform();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // V complete, raw consumed: the next strip may land
        const int sn = s + 1;
        const bool more = sn < s_end && !(H2_DBG & 8);
        Geom tn = strip_geom(more ? sn : s);
        const bool inner = more && tn.interior;
        if (!(H2_DBG & 4)) {
            float oa[2][MI], ob[2][NJ];
            h2_load_ops<CI, 0>(vlane, oa[0], ob[0]);
            const Pat pt = load_patterns();
            h2_contract<CI, 0>(acc, vlane, oa, ob, [&](auto stc) {
                if constexpr (decltype(stc)::value < L::NSTEP) {
                    if (inner) stage_piece(stc, tn, pt);
                }
            });
        } else if (inner) {
            stage_interior(tn);
        }
        if (more && !inner) stage_border(sn);
        s = sn;
    }

    // partial result -> slab[split][cls][f = 4 * wave + b][KP][NcP]
    const int KP = d.Cin;
    float* sbase = slab + (((size_t)split * g.ncls + cls) * 16 + wave * 4) * KP * d.NcP;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int m = 0; m < MI; ++m)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float* base = sbase + ((size_t)b * KP + ci0 + 32 * m) * d.NcP + co0 + 32 * j + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                    base[(size_t)i * d.NcP] = acc[(b * MI + m) * NJ + j][r];
                }
            }
}

// (two plain kernels instead of one kernel template: hipcc's host pass rejects the template-id of this kernel -- "substitution
// failure" without a reason -- once the body instantiates its generic lambdas)
__global__ __launch_bounds__(256, 1) void conv_wgrad32v2_kernel(const adh_conv_desc d, const Wg32v2Args g, float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) float smem[];   // rawx | rawg | V | pattern table
    h2_workgroup<64>(d, g, slab, smem);
}
__global__ __launch_bounds__(256, 1) void conv_wgrad32v2c96_kernel(const adh_conv_desc d, const Wg32v2Args g, float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    h2_workgroup<96>(d, g, slab, smem);
}

// ------------------------------------------------------------------------------------------------ host side
struct Wg32Class {
    int ymin, xmin, xps;
    int tap0, tap_sy, tap_sx, rev;   // kernel tap (ty, tx) of the class is tap index tap0 + ty tap_sy + tx tap_sx of d's KH x KW taps;
};                                   // rev: taps walk backwards, halo offset (hy, hx) belongs to tap (1 - hy, 1 - hx)
struct Wg32Plan {
    int ncls, TN;
    int v2;                          // conv_wgrad32v2_kernel<CI>: 64 (Cin % 64 == 0, Cout % 96 == 0), 96 (Cin % 96 == 0, Cout % 64 == 0), 0 = no
    int S, SX, TY, ncob;             // its strips (3 x 24 class-grid pixels) and output-channel groups
    Wg32Args base;
    Wg32Class cls[4];
};

// the shapes conv_wgrad.hip's wgrad_rows_plan hands to conv_wgrad_rows_kernel<2,2,..> (same class decomposition), without its
// grid-alignment conditions
static int wgrad32_plan(const adh_conv_desc* d, int nsplit, Wg32Plan* p) {
    // ADH_WINO32_WGRAD: 0 = off, 1 (default) = where it wins, 2 = every eligible shape
    const char* env = getenv("ADH_WINO32_WGRAD");   // (read per call: the tests switch it)
    const int mode = env ? atoi(env) : 1;
    if (!mode || !d) return 0;
    // The kernel is bound by the issue of its LDS-DMA pieces, not by its MFMAs (82 pieces per 96 MFMAs and wave: the same
    // bytes per pixel as the direct kernel stages, for 4/9 of the MFMAs): measured at the headline shapes, ConvTranspose
    // 384 -> 96 10.25 -> 8.4 ms, but Conv2d k4 s2 96 -> 192 5.2 -> 6.2 ms.  The four-class (k4 s2) form therefore stays on
    // the direct kernel unless forced.
    static const bool v2_enabled = !(getenv("ADH_WGRAD32_V2") && getenv("ADH_WGRAD32_V2")[0] == '0');   // A/B switch
    p->v2 = 0;
    if (v2_enabled && d->Cout == d->NcP) {
        if (d->Cin % 64 == 0 && d->NcP % 96 == 0) p->v2 = 64;
        else if (d->Cin % 96 == 0 && d->NcP % 64 == 0) p->v2 = 96;
    }
    if (mode < 2 && d->KH == 4 && !p->v2) return 0;
    if (d->Cin % 32 != 0 || d->Cout % 4 != 0 || d->NcP != adh_round_up(d->Cout, 32)) return 0;
    if (d->in_cstride % 4 != 0 || d->out_cstride % 4 != 0) return 0;
    if (d->in_sy != d->in_sx || d->dstep_y != d->dstep_x || d->out_sy != d->out_sx) return 0;
    if (d->VH < 1 || d->VW < 1) return 0;
    if ((d->VH - 1) * d->out_sy + d->out_oy >= d->OH || (d->VW - 1) * d->out_sx + d->out_ox >= d->OW) return 0;
    if ((int64_t)(d->IH + 8) * d->IW * d->in_cstride * 4 >= (int64_t)1 << 31) return 0;
    if ((int64_t)d->OH * d->OW * d->out_cstride * 4 >= (int64_t)1 << 31) return 0;
    const int t = d->NcP / 32;
    p->TN = t % 3 == 0 ? 3 : (t % 2 == 0 ? 2 : 1);
    Wg32Args& b = p->base;
    b.tiles_x = adh_ceil_div(d->VW, G32_TW);
    b.tiles_y = adh_ceil_div(d->VH, G32_TH);
    b.ntiles = b.tiles_x * b.tiles_y * d->N;
    b.nsplit = nsplit;
    b.nco_groups = d->NcP / (32 * p->TN);
    b.ngroups = (d->Cin / 32) * b.nco_groups;
    if (p->v2) {
        p->SX = adh_ceil_div(d->VW, 3 * H2_T);
        p->TY = b.tiles_y;
        p->S = p->SX * p->TY * d->N;
        p->ncob = d->NcP / (160 - p->v2);
        b.ngroups = (d->Cin / p->v2) * p->ncob;
        b.ntiles = p->S;
    }
    const int s = d->in_sy, ds = d->dstep_y;
    if (d->KH == 2 && d->KW == 2 && (ds == s || ds == -s) && (s == 1 || s == 2)) {
        p->ncls = 1;
        Wg32Class& c = p->cls[0];
        c.xps = s;
        c.rev = ds < 0;
        c.ymin = d->dy0 + (ds < 0 ? ds : 0);
        c.xmin = d->dx0 + (ds < 0 ? ds : 0);
        c.tap0 = 0; c.tap_sy = 2; c.tap_sx = 1;
        return 1;
    }
    if (d->KH == 4 && d->KW == 4 && s == 2 && ds == 1) {
        // Conv2d k4 s2: kernel index ky = 2 ty + py reads input row 2 (vy + ty) + dy0 + py
        p->ncls = 4;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                Wg32Class& c = p->cls[py * 2 + px];
                c.xps = 2;
                c.rev = 0;
                c.ymin = d->dy0 + py;
                c.xmin = d->dx0 + px;
                c.tap0 = py * 4 + px; c.tap_sy = 8; c.tap_sx = 2;
            }
        return 1;
    }
    return 0;
}

extern "C" int adh_conv_wgrad_wino32_groups(const adh_conv_desc* d) {
    if (!d) return ADH_E_ARG;
    Wg32Plan p;
    return wgrad32_plan(d, 1, &p) ? p.base.ngroups : 0;
}

// frequency slabs the caller must provide per pixel split: 16 per class
extern "C" int adh_conv_wgrad_wino32_classes(const adh_conv_desc* d) {
    if (!d) return ADH_E_ARG;
    Wg32Plan p;
    return wgrad32_plan(d, 1, &p) ? p.ncls : 0;
}

// kernel launches one adh_conv_wgrad_wino32 call makes: the classes of a k4 s2 form share one grid on conv_wgrad32v2_kernel
// (the caller's split count then sees classes x groups workgroup groups in that grid)
extern "C" int adh_conv_wgrad_wino32_launches(const adh_conv_desc* d) {
    if (!d) return ADH_E_ARG;
    Wg32Plan p;
    if (!wgrad32_plan(d, 1, &p)) return 0;
    return p.v2 ? 1 : p.ncls;
}

extern "C" int adh_conv_wgrad_wino32_tiles(const adh_conv_desc* d) {
    if (!d) return ADH_E_ARG;
    Wg32Plan p;
    return wgrad32_plan(d, 1, &p) ? p.base.ntiles : 0;
}

extern "C" int adh_conv_wgrad_wino32(void* stream, const adh_conv_desc* d, float* slab, int nsplit) {
    if (!d || !slab || nsplit < 1 || !d->in || !d->out) return ADH_E_ARG;
    Wg32Plan p;
    if (!wgrad32_plan(d, nsplit, &p)) return ADH_E_UNSUPPORTED;
    if (((uintptr_t)d->in & 15) || ((uintptr_t)d->out & 15)) return ADH_E_ARG;
    const int lds = 2 * (G32_XF + p.TN * G32_TH * G32_GROW) * 4;
    const int nblocks = ((nsplit + 7) / 8) * p.base.ngroups * 8;
    hipStream_t s = (hipStream_t)stream;
    if (p.v2) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad32v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad32v2c96_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        Wg32v2Args a;
        for (int c = 0; c < 4; ++c) { a.ymin[c] = p.cls[c < p.ncls ? c : 0].ymin; a.xmin[c] = p.cls[c < p.ncls ? c : 0].xmin; }
        a.xps = p.cls[0].xps;
        a.S = p.S; a.SX = p.SX; a.TY = p.TY; a.nsplit = nsplit; a.ngroups = p.base.ngroups; a.ncob = p.ncob;
        a.cls = 0; a.ncls = p.ncls;
        a.cblocks = p.ncls > 1 ? nblocks : 0;            // all classes of a k4 s2 form in one grid
        for (int c = 0; c < 4; ++c) a.gofs[c] = ((int64_t)d->out_oy * d->OW + d->out_ox) * d->out_cstride;
        const int grid = nblocks * p.ncls;
        if (p.v2 == 64) hipLaunchKernelGGL(conv_wgrad32v2_kernel, dim3(grid), dim3(256), H2L<64>::LDS_BYTES, s, *d, a, slab);
        else hipLaunchKernelGGL(conv_wgrad32v2c96_kernel, dim3(grid), dim3(256), H2L<96>::LDS_BYTES, s, *d, a, slab);
        const int rc = adh_check_launch();
        if (rc) return rc;
        return ADH_OK;
    }
    for (int c = 0; c < p.ncls; ++c) {
        Wg32Args a = p.base;
        a.ymin = p.cls[c].ymin;
        a.xmin = p.cls[c].xmin;
        a.xps = p.cls[c].xps;
        a.cls = c;
        a.ncls = p.ncls;
#define G32_CASE(tn_) \
        if (p.TN == tn_) { \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad32_kernel<tn_>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            hipLaunchKernelGGL((conv_wgrad32_kernel<tn_>), dim3(nblocks), dim3(256), lds, s, *d, a, slab); \
        }
        G32_CASE(3) G32_CASE(2) G32_CASE(1)
#undef G32_CASE
        const int rc = adh_check_launch();
        if (rc) return rc;
    }
    return ADH_OK;
}

struct Wg32Taps {
    int ncls;
    int tap0[4], tap_sy[4], tap_sx[4], rev[4];
};

// dst(layout L) (+)= A^T (scaled sum over splits of slab[s][cls][16][KP][NcP]) A per class, scattered to the class's taps.
// A = [[1,0],[1,1],[1,-1],[0,-1]]; rows / columns 1, 2 of the slab carry the deferred factor 1/2 of G.
__global__ void wgrad_reduce_wino32_kernel(const float* __restrict__ slab, int nsplit, int KP, int NcP, const adh_wlayout L,
                                           const Wg32Taps tp, float* __restrict__ dst, int accumulate) {
    const int64_t total = (int64_t)tp.ncls * L.K * L.Nc;
    const int64_t fstride = (int64_t)KP * NcP, cls_stride = 16 * fstride, split_stride = tp.ncls * cls_stride;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % L.Nc);
        int64_t r = idx / L.Nc;
        const int k = (int)(r % L.K);
        const int c = (int)(r / L.K);
        const float* p = slab + c * cls_stride + (int64_t)k * NcP + n;
        float u[4][4];
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            float s0 = 0.f, s1 = 0.f;
            int sp = 0;
            for (; sp + 2 <= nsplit; sp += 2) {
                s0 += p[(int64_t)sp * split_stride + f * fstride];
                s1 += p[(int64_t)(sp + 1) * split_stride + f * fstride];
            }
            if (sp < nsplit) s0 += p[(int64_t)sp * split_stride + f * fstride];
            const int a = f >> 2, b = f & 3;
            const float sc = ((a == 1 || a == 2) ? 0.5f : 1.f) * ((b == 1 || b == 2) ? 0.5f : 1.f);
            u[a][b] = sc * (s0 + s1);
        }
        float t[2][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            t[0][b] = u[0][b] + u[1][b] + u[2][b];
            t[1][b] = u[1][b] - u[2][b] - u[3][b];
        }
#pragma unroll
        for (int hy = 0; hy < 2; ++hy) {
            const float w2[2] = {t[hy][0] + t[hy][1] + t[hy][2], t[hy][1] - t[hy][2] - t[hy][3]};
#pragma unroll
            for (int hx = 0; hx < 2; ++hx) {
                const int ty = tp.rev[c] ? 1 - hy : hy, tx = tp.rev[c] ? 1 - hx : hx;
                const int tap = tp.tap0[c] + ty * tp.tap_sy[c] + tx * tp.tap_sx[c];
                const int tyy = tap / L.KWt, txx = tap - tyy * L.KWt;
                const int64_t off = (int64_t)L.tap_off0 + tyy * L.tap_off_sy + txx * L.tap_off_sx + (int64_t)k * L.stride_k +
                                    (int64_t)n * L.stride_n;
                dst[off] = accumulate ? dst[off] + w2[hx] : w2[hx];
            }
        }
    }
}


// The class descriptors of ONE transposed layer (2 x 2-tap forms that differ in out_oy / out_ox and dy0 / dx0 only) as one grid of
// conv_wgrad32v2_kernel: slab[split][m][16][Cin][NcP], the splits are summed here (into split 0); the caller then runs
// adh_wgrad_reduce_wino32(slab + m * 16 * Cin * NcP, nsplit = 1, &descs[m], ..) per class.  ADH_E_UNSUPPORTED: launch them one by one.
extern "C" int adh_conv_wgrad_wino32_multi(void* stream, const adh_conv_desc* descs, int n, float* slab, int nsplit) {
    if (!descs || !slab || n < 2 || n > 4 || nsplit < 1) return ADH_E_ARG;
    Wg32Plan p0;
    if (!wgrad32_plan(&descs[0], nsplit, &p0) || !p0.v2 || p0.ncls != 1) return ADH_E_UNSUPPORTED;
    Wg32v2Args a;
    for (int m = 0; m < n; ++m) {
        const adh_conv_desc* d = &descs[m];
        Wg32Plan p;
        if (!d->in || !d->out || ((uintptr_t)d->in & 15) || ((uintptr_t)d->out & 15)) return ADH_E_ARG;
        if (!wgrad32_plan(d, nsplit, &p) || p.v2 != p0.v2 || p.ncls != 1 || p.S != p0.S || p.SX != p0.SX || p.TY != p0.TY ||
            p.cls[0].xps != p0.cls[0].xps)
            return ADH_E_UNSUPPORTED;
        adh_conv_desc x = descs[0], y = *d;
        x.out_oy = y.out_oy = 0; x.out_ox = y.out_ox = 0; x.dy0 = y.dy0 = 0; x.dx0 = y.dx0 = 0;
        x.wp = y.wp = nullptr; x.stats = y.stats = nullptr;
        if (memcmp(&x, &y, sizeof(x)) != 0) return ADH_E_UNSUPPORTED;
        a.ymin[m] = p.cls[0].ymin; a.xmin[m] = p.cls[0].xmin;
        a.gofs[m] = ((int64_t)d->out_oy * d->OW + d->out_ox) * d->out_cstride;
    }
    for (int m = n; m < 4; ++m) { a.ymin[m] = a.ymin[0]; a.xmin[m] = a.xmin[0]; a.gofs[m] = a.gofs[0]; }
    const int nblocks = ((nsplit + 7) / 8) * p0.base.ngroups * 8;
    a.xps = p0.cls[0].xps;
    a.S = p0.S; a.SX = p0.SX; a.TY = p0.TY; a.nsplit = nsplit; a.ngroups = p0.base.ngroups; a.ncob = p0.ncob;
    a.cls = 0; a.ncls = n; a.cblocks = nblocks;
    hipStream_t s = (hipStream_t)stream;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad32v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad32v2c96_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (p0.v2 == 64) hipLaunchKernelGGL(conv_wgrad32v2_kernel, dim3(nblocks * n), dim3(256), H2L<64>::LDS_BYTES, s, descs[0], a, slab);
    else hipLaunchKernelGGL(conv_wgrad32v2c96_kernel, dim3(nblocks * n), dim3(256), H2L<96>::LDS_BYTES, s, descs[0], a, slab);
    const int rc = adh_check_launch();
    if (rc) return rc;
    if (nsplit > 1) adh_wgrad_sum_splits(s, slab, nsplit, (int64_t)n * 16 * descs[0].Cin * descs[0].NcP / 4);
    return adh_check_launch();
}

extern "C" int adh_wgrad_reduce_wino32(void* stream, float* slab, int nsplit, const adh_conv_desc* d, int KP, int NcP,
                                       const adh_wlayout* L, float* dst, int accumulate) {
    if (!slab || !L || !dst || !d || nsplit < 1 || (NcP & 3)) return ADH_E_ARG;
    Wg32Plan p;
    if (!wgrad32_plan(d, nsplit, &p)) return ADH_E_UNSUPPORTED;
    if (L->KHt * L->KWt != d->KH * d->KW) return ADH_E_ARG;
    Wg32Taps tp;
    tp.ncls = p.ncls;
    for (int c = 0; c < 4; ++c) {
        const Wg32Class& k = p.cls[c < p.ncls ? c : 0];
        tp.tap0[c] = k.tap0; tp.tap_sy[c] = k.tap_sy; tp.tap_sx[c] = k.tap_sx; tp.rev[c] = k.rev;
    }
    hipStream_t s = (hipStream_t)stream;
    if (nsplit > 1) {
        const int64_t n4 = (int64_t)p.ncls * 16 * KP * NcP / 4;
        adh_wgrad_sum_splits(s, slab, nsplit, n4);
    }
    const int64_t total = (int64_t)p.ncls * L->K * L->Nc;
    hipLaunchKernelGGL(wgrad_reduce_wino32_kernel, dim3(adh_min_i(adh_ceil_div(total, 64), 16384)), dim3(64), 0, s, slab, 1, KP,
                       NcP, *L, tp, dst, accumulate);
    return adh_check_launch();
}
