// Weight gradient of the 3x3 stride-1 pad-1 convolutions in the Winograd F(4x4,3x3) domain, fp32 MFMA, gfx950.
// Replaces the weight-gradient half of ATen's conv2d backward for those layers
// (/root/reference models/dehazing/base_model.py:11-13,26-41 -- nn.Conv2d inside ConvBlock / ResidualBlock).
//
//   dW(3x3) = sum over 4x4 output tiles t of  corr(d_t (6x6 input patch), g_t (4x4 output-gradient tile))
//           = A'^T [ sum_t (G' g_t G'^T) .* (B^T d_t B) ] A'            (F(3x3, 4x4): 36 products per 144 MACs)
//
// with the same points {0, +-3/4, +-5/4, inf} and the same B^T as conv_wino43.hip; G' = rows [1, p, p^2, p^3] / N_p
// (inf: [0,0,0,1]), A'^T = rows p^i, i < 3 (inf column: [0,0,1]).  The kernel uses the un-normalised rows [1, p, p^2,
// p^3] (dyadic constants, exact in fp32); the 1 / (N_a N_b) factors are applied in double by the reduce kernel.
// 1/4 of the direct algorithm's MFMA work (the F(2x2,3x3)-domain kernel in conv_wgrad.hip: 4/9).
//
// Both MFMA operands are transformed data, so the transform cost per product is what decides the design.  A workgroup
// owns 96 input channels x 96 output channels x ONE THIRD of the frequencies -- the row pairs {0,5}, {1,2}, {3,4} of the
// 6x6 frequency grid, which are exactly the pairs the even/odd structure of the transforms produces together -- i.e.
// 12 frequencies x 3 x 3 channel tiles = 108 accumulator tiles, 27 per wave (16 pinned to AGPRs).  Its first transform
// pass then computes only its two rows (4 FMAs per column instead of 12) and every transformed value feeds three
// MFMAs, which brings the VALU : MFMA ratio to that of the forward kernel.
//   per strip of 6 tiles (4 rows x 24 pixels; K = 6 of the contraction):
//     LDS-DMA   x halo [6][26 px][96 ci] and g [4][24 px][96 co], 2 pixels (48 lanes x 16 B) per instruction;
//     pass 1    thread = (column, channel quad), loop over the 6 tiles: raw -> T[2 rows][6 | 4 columns]        (V planes)
//     pass 2    thread = (row, channel quad, every second tile): T -> V[2][6] in place (x: B^T, g: [1,p,p^2,p^3])
//     contract  wave w: frequencies 3w..3w+2 of the 12, per frequency and tile pair 3 + 3 ds_read_b32 feed 9 MFMAs
//   raw is single buffered (152 KB of LDS in all): the next strip is staged as soon as pass 1 has consumed the current
//   one and lands during pass 2 + contraction.  No global operand loads in the loop: all waits are compiler-visible.
// Strips at the image border (and the ragged last strip of a row) zero the raw buffers and load with per-pixel lane masks.
// Output: slab[split][36][KP][NcP] partial sums; adh_wgrad_reduce_wino43 sums the splits and applies A'^T (.) A'.
//
// Status (round 1): parity-green, NOT the default (engine.USE_WINO43_WGRAD / ADH_WINO43_WGRAD=1 selects it): 3.9 / 3.7 / 3.7
// ms on the 96 / 192 / 384-channel layers against 3.3 / 2.9 / 3.0 for the F(2x2,3x3)-domain kernel.  Compile-time
// ablation (G4_DBG) at 96 channels: contraction 1.21 ms (= the MFMA floor for 174 GFLOP), pass 1 0.59, pass 2 0.66, and
// 1.1 ms of the staging exposed although nothing waits on it before the next strip (staging alone runs at 8.5 TB/s of
// L2 + HBM traffic: every input tile is read by the three frequency groups).  HBM traffic itself is as planned (4.25 GB
// per launch, PMC: two of the three reads hit L2).  Spreading the DMA instructions over pass 2 and the contraction did
// not help.  Next steps: fewer bytes per product (taller strips with a rolling halo; two frequency groups per
// workgroup at 64 x 96 channels), balance the transform passes over all four waves.
#include "common.h"
#include <cstdlib>
#include <type_traits>

#ifndef G4_DBG
#define G4_DBG 0   // dev builds: 1 = skip pass 1, 2 = skip pass 2, 4 = skip the contraction, 8 = stage only the first strip
#endif
#define G4_T 6                                    // tiles per strip
#define G4_XCOLS 26                               // halo columns of a strip
#define G4_GCOLS 24
#define G4_RAWX_F (6 * G4_XCOLS * 96)             // 14,976 floats
#define G4_RAWG_F (4 * G4_GCOLS * 96)             //  9,216 floats
#define G4_VPLANE (G4_T * 192)                    // floats per frequency plane: [tile][96 ci | 96 co]
#define G4_V_F (12 * G4_VPLANE)                   // 13,824 floats
#define G4_LDS_BYTES ((G4_RAWX_F + G4_RAWG_F + G4_V_F) * 4)

#define G4_A 0.75f
#define G4_B 1.25f
#define G4_A2 0.5625f
#define G4_B2 1.5625f
#define G4_A2B2 0.87890625f
#define G4_S2 2.125f

typedef __attribute__((address_space(3))) void* lds_void_ptr5;

struct Wg43Args {
    int TY, SX, S;         // tile rows, strips per tile row, total strips (N * TY * SX)
    int nsplit;
    int ngroups, ncob;     // groups = cib x cob x 3 frequency groups
    int KP;
};

template <int IDX>
__device__ __forceinline__ void g4_mfma(f32x16& c, float a, float b) {
    if constexpr (IDX < 16) asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// local frequency FL (0..2 of this wave), nine MFMAs: 3 input-channel tiles x 3 output-channel tiles
template <int FL, int MI, int NJ>
__device__ __forceinline__ void g4_nine(f32x16 (&acc)[27], const float (&a)[3], const float (&b)[3]) {
    if constexpr (MI < 3) {
        g4_mfma<(FL * 3 + MI) * 3 + NJ>(acc[(FL * 3 + MI) * 3 + NJ], a[MI], b[NJ]);
        if constexpr (NJ + 1 < 3) g4_nine<FL, MI, NJ + 1>(acc, a, b);
        else g4_nine<FL, MI + 1, 0>(acc, a, b);
    }
}
// step ST = FL * 3 + KS of 9: operands of step ST + 1 are read before the nine MFMAs of step ST are issued
template <int ST>
__device__ __forceinline__ void g4_load_ops(const float* vlane, float (&a)[3], float (&b)[3]) {
    const float* p = vlane + (ST / 3) * G4_VPLANE + (ST % 3) * 2 * 192;
#pragma unroll
    for (int m = 0; m < 3; ++m) a[m] = p[m * 32];
#pragma unroll
    for (int j = 0; j < 3; ++j) b[j] = p[96 + j * 32];
}
template <int ST, typename Hook>
__device__ __forceinline__ void g4_contract(f32x16 (&acc)[27], const float* vlane, float (&a)[2][3], float (&b)[2][3],
                                            Hook&& hook) {
    if constexpr (ST < 9) {
        if constexpr (ST + 1 < 9) g4_load_ops<ST + 1>(vlane, a[(ST + 1) & 1], b[(ST + 1) & 1]);
        hook(std::integral_constant<int, ST>{});      // (staging instructions of the next strip ride along)
        g4_nine<ST / 3, 0, 0>(acc, a[ST & 1], b[ST & 1]);
        g4_contract<ST + 1>(acc, vlane, a, b, hook);
    }
}

struct G4Neg { float s2, b2, a2, a, b; };   // negative constants in opaque SGPRs (see conv_wino43.hip, W4Neg)
__device__ __forceinline__ G4Neg g4_neg_constants() {
    G4Neg n = {-G4_S2, -G4_B2, -G4_A2, -G4_A, -G4_B};
    asm volatile("" : "+s"(n.s2), "+s"(n.b2), "+s"(n.a2), "+s"(n.a), "+s"(n.b));
    return n;
}

// One launch for the three frequency groups fg = rows {0,5}, {1,2}, {3,4} of the 6x6 grid.  Block index -> (XCD, fg,
// channel-block pair, split): the three groups of a split sit next to each other on one XCD and read the same strips at
// the same time, so two of the three reads of every input tile hit that XCD's L2.  fg only selects constants and two
// short code paths in the first transform pass (uniform branches).
__global__ __launch_bounds__(256, 1) void conv_wgrad_wino43_kernel(const adh_conv_desc d, const Wg43Args g,
                                                                   float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // rawx | rawg | V
    const int bid = blockIdx.x;
    const int qd = bid >> 3;
    const int fg = qd % 3;
    const int q2 = qd / 3;
    const int grp = q2 % g.ngroups;
    const int split = (q2 / g.ngroups) * 8 + (bid & 7);
    if (split >= g.nsplit) return;
    float* const rawx = lds;
    float* const rawg = lds + G4_RAWX_F;
    float* const V = lds + G4_RAWX_F + G4_RAWG_F;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;
    const int cob = grp % g.ncob, cib = grp / g.ncob;

    const int xcs = d.in_cstride * 4, gcs = d.out_cstride * 4;       // pixel pitches in bytes
    // per-lane DMA offsets: 2 pixels x 24 channel quads per instruction (lanes 48..63 idle)
    const int pj = lane >= 24 ? 1 : 0, pq = lane - 24 * pj;
    const int vox = pj * xcs + pq * 16 + cib * 384, vog = pj * gcs + pq * 16 + cob * 384;
    const bool dma_lane = lane < 48;

    const G4Neg ng = g4_neg_constants();
    // first-pass constants of this frequency group (fg = 1: points +-a, fg = 2: points +-b)
    const float kp = fg == 1 ? G4_A : G4_B, kn = fg == 1 ? ng.a : ng.b;
    const float kxn = fg == 1 ? ng.b2 : ng.a2, kg2 = fg == 1 ? G4_A2 : G4_B2;

    // stage strip s (uniform): raw buffers <- x halo and g tile.  Descriptors cover the whole tensors (x shifted by one row
    // + one pixel so that every scalar offset is non-negative); pixel pairs: wave w takes pairs w, w + 4, w + 8 (+ 12)
    // of every row -- 24 / 18 / 18 / 18 x pairs + 12 g pairs per wave, one LDS-DMA instruction each
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(d.in) - (int64_t)(d.IW + 1) * xcs), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.out), 0, 0xffffffff, 0x00020000);
    const unsigned xrow = (unsigned)d.IW * xcs, grow = (unsigned)d.VW * gcs;
    float* const rawx_w = rawx + wave * 2 * 96;
    float* const rawg_w = rawg + wave * 2 * 96;
    // one staging instruction of an interior strip: slot i < 24 = x pair (row i / 4, k = i % 4), else g pair
    auto dma_slot = [&](const int i, const unsigned xb, const unsigned gb) {
        if (i < 24) {
            const int row = i >> 2, k = i & 3;
            if (k < 3 || wave == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr5)(rawx_w + (row * G4_XCOLS + 8 * k) * 96), 16, vox,
                                                         xb + row * xrow + k * 8 * xcs, 0, 0);
        } else {
            const int row = (i - 24) / 3, k = (i - 24) % 3;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lds_void_ptr5)(rawg_w + (row * G4_GCOLS + 8 * k) * 96), 16, vog,
                                                     gb + row * grow + k * 8 * gcs, 0, 0);
        }
    };
    // strip s -> scalar offsets of this wave's first pair; returns whether the strip is interior
    auto strip_origin = [&](int s, int& y0, int& x0, unsigned& xb, unsigned& gb) {
        int r = s;
        const int sx = r % g.SX;
        r /= g.SX;
        const int ty = r % g.TY;
        const int n = r / g.TY;
        y0 = ty * 4;
        x0 = sx * 24;
        xb = __builtin_amdgcn_readfirstlane((unsigned)((n * d.IH + y0) * d.IW + x0 + 2 * wave) * xcs);
        gb = __builtin_amdgcn_readfirstlane((unsigned)((n * d.VH + y0) * d.VW + x0 + 2 * wave) * gcs);
        return y0 >= 1 && y0 + 5 <= d.IH && x0 >= 1 && x0 + 25 <= d.IW;
    };
    // all 36 slots at once (first strip of a split)
    auto stage_interior = [&](unsigned xb, unsigned gb) {
        if (dma_lane) {
#pragma unroll
            for (int i = 0; i < 36; ++i) dma_slot(i, xb, gb);
        }
    };
    // border strip: zero both raw buffers, then load only the pixels inside the image
    auto stage_border = [&](int y0, int x0, unsigned xb, unsigned gb) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = tid; i < (G4_RAWX_F + G4_RAWG_F) / 4; i += 256) *reinterpret_cast<f32x4*>(lds + i * 4) = z;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll 1
        for (int row = 0; row < 6; ++row) {
            const int iy = y0 - 1 + row;
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                const int ix = x0 - 1 + 2 * (wave + 4 * k) + pj;
                if ((k < 3 || wave == 0) && dma_lane && iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr5)(rawx_w + (row * G4_XCOLS + 8 * k) * 96), 16, vox,
                                                             xb + row * xrow + k * 8 * xcs, 0, 0);
            }
        }
#pragma unroll 1
        for (int row = 0; row < 4; ++row)
#pragma unroll 1
            for (int k = 0; k < 3; ++k) {
                const int gx = x0 + 2 * (wave + 4 * k) + pj;
                if (dma_lane && gx < d.VW)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lds_void_ptr5)(rawg_w + (row * G4_GCOLS + 8 * k) * 96), 16, vog,
                                                             gb + row * grow + k * 8 * gcs, 0, 0);
            }
    };

    // ------------------------------------------------------------------ transforms
    const int slot = (tid * 171) >> 12;             // tid / 24 for tid < 256
    const int q4 = (tid - slot * 24) * 4;           // channel quad offset (floats)
    // pass 1: slot 0..5 = x column, 6..9 = g column, 10 = idle; loop over the six tiles
    auto pass1 = [&]() {
        if (slot < 6) {
            const float* src = rawx + slot * 96 + q4;
            float* dst = V + slot * G4_VPLANE + q4;
#pragma unroll
            for (int t = 0; t < G4_T; ++t) {
                const float* s = src + t * 4 * 96;
                f32x4 o0, o1;
                if (fg == 0) {
                    const f32x4 d0 = *reinterpret_cast<const f32x4*>(s), d1 = *reinterpret_cast<const f32x4*>(s + G4_XCOLS * 96),
                                d2 = *reinterpret_cast<const f32x4*>(s + 2 * G4_XCOLS * 96),
                                d3 = *reinterpret_cast<const f32x4*>(s + 3 * G4_XCOLS * 96),
                                d4 = *reinterpret_cast<const f32x4*>(s + 4 * G4_XCOLS * 96),
                                d5 = *reinterpret_cast<const f32x4*>(s + 5 * G4_XCOLS * 96);
                    o0 = G4_A2B2 * d0 + (ng.s2 * d2 + d4);
                    o1 = G4_A2B2 * d1 + (ng.s2 * d3 + d5);
                } else {
                    const f32x4 d1 = *reinterpret_cast<const f32x4*>(s + G4_XCOLS * 96),
                                d2 = *reinterpret_cast<const f32x4*>(s + 2 * G4_XCOLS * 96),
                                d3 = *reinterpret_cast<const f32x4*>(s + 3 * G4_XCOLS * 96),
                                d4 = *reinterpret_cast<const f32x4*>(s + 4 * G4_XCOLS * 96);
                    const f32x4 p = kxn * d2 + d4, qq = kxn * d1 + d3;      // kxn = -b^2 (rows 1,2) / -a^2 (rows 3,4)
                    o0 = kp * qq + p;                                        // kp = a / b, kn = -kp
                    o1 = kn * qq + p;
                }
                *reinterpret_cast<f32x4*>(dst + t * 192) = o0;
                *reinterpret_cast<f32x4*>(dst + 6 * G4_VPLANE + t * 192) = o1;
                __builtin_amdgcn_sched_barrier(0);   // one tile of loads in flight, not six (registers)
            }
        } else if (slot < 10) {
            const float* src = rawg + (slot - 6) * 96 + q4;
            float* dst = V + (slot - 6) * G4_VPLANE + 96 + q4;
#pragma unroll
            for (int t = 0; t < G4_T; ++t) {
                const float* s = src + t * 4 * 96;
                f32x4 o0, o1;
                if (fg == 0) {
                    o0 = *reinterpret_cast<const f32x4*>(s);
                    o1 = *reinterpret_cast<const f32x4*>(s + 3 * G4_GCOLS * 96);
                } else {
                    const f32x4 g0 = *reinterpret_cast<const f32x4*>(s), g1 = *reinterpret_cast<const f32x4*>(s + G4_GCOLS * 96),
                                g2 = *reinterpret_cast<const f32x4*>(s + 2 * G4_GCOLS * 96),
                                g3 = *reinterpret_cast<const f32x4*>(s + 3 * G4_GCOLS * 96);
                    const f32x4 e = kg2 * g2 + g0, o = kg2 * g3 + g1;        // kg2 = a^2 / b^2
                    o0 = kp * o + e;
                    o1 = kn * o + e;
                }
                *reinterpret_cast<f32x4*>(dst + t * 192) = o0;
                *reinterpret_cast<f32x4*>(dst + 6 * G4_VPLANE + t * 192) = o1;
                __builtin_amdgcn_sched_barrier(0);   // one tile of loads in flight, not six (registers)
            }
        }
    };
    // pass 2: slot 0..7: role = slot & 3 (x row 0, x row 1, g row 0, g row 1), tiles (slot >> 2) + {0, 2, 4}
    auto pass2 = [&](const int k) {
        if (slot < 8) {
            const int role = slot & 3;
            float* pv = V + (role & 1) * 6 * G4_VPLANE + (role >> 1) * 96 + q4 + ((slot >> 2) + 2 * k) * 192;
            if (role < 2) {
                f32x4 t[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) t[c] = *reinterpret_cast<const f32x4*>(pv + c * G4_VPLANE);
                const f32x4 v0 = G4_A2B2 * t[0] + (ng.s2 * t[2] + t[4]);
                const f32x4 v5 = G4_A2B2 * t[1] + (ng.s2 * t[3] + t[5]);
                const f32x4 p = ng.b2 * t[2] + t[4], r = ng.a2 * t[2] + t[4];
                const f32x4 qq = ng.b2 * t[1] + t[3], s = ng.a2 * t[1] + t[3];
                *reinterpret_cast<f32x4*>(pv) = v0;
                *reinterpret_cast<f32x4*>(pv + 1 * G4_VPLANE) = G4_A * qq + p;
                *reinterpret_cast<f32x4*>(pv + 2 * G4_VPLANE) = ng.a * qq + p;
                *reinterpret_cast<f32x4*>(pv + 3 * G4_VPLANE) = G4_B * s + r;
                *reinterpret_cast<f32x4*>(pv + 4 * G4_VPLANE) = ng.b * s + r;
                *reinterpret_cast<f32x4*>(pv + 5 * G4_VPLANE) = v5;
            } else {
                f32x4 t[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) t[c] = *reinterpret_cast<const f32x4*>(pv + c * G4_VPLANE);
                const f32x4 ea = G4_A2 * t[2] + t[0], oa = G4_A2 * t[3] + t[1];
                const f32x4 eb = G4_B2 * t[2] + t[0], ob = G4_B2 * t[3] + t[1];
                *reinterpret_cast<f32x4*>(pv + 1 * G4_VPLANE) = G4_A * oa + ea;     // (plane 0 already holds t[0])
                *reinterpret_cast<f32x4*>(pv + 2 * G4_VPLANE) = ng.a * oa + ea;
                *reinterpret_cast<f32x4*>(pv + 3 * G4_VPLANE) = G4_B * ob + eb;
                *reinterpret_cast<f32x4*>(pv + 4 * G4_VPLANE) = ng.b * ob + eb;
                *reinterpret_cast<f32x4*>(pv + 5 * G4_VPLANE) = t[3];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // ------------------------------------------------------------------ main loop over this split's strips
    f32x16 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // A / B operand lane address: V[3 * wave + fl][2 * ks + h][mi * 32 + l31 | 96 + nj * 32 + l31]
    const float* const vlane = V + (3 * wave) * G4_VPLANE + h * 192 + l31;

    // this split's strips: a contiguous range (the three frequency-group workgroups of a split walk it together)
    const int per = (g.S + g.nsplit - 1) / g.nsplit;
    int s = split * per;
    const int s_end = adh_min_i(s + per, g.S);
    if (s < s_end) {
        int y0, x0;
        unsigned xb, gb;
        if (strip_origin(s, y0, x0, xb, gb)) stage_interior(xb, gb);
        else stage_border(y0, x0, xb, gb);
    }
#pragma unroll 1
    while (s < s_end) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // strip s landed; every wave is past the previous contraction
        if (!(G4_DBG & 1)) pass1();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // raw consumed, T complete
        // stage the next strip: it lands during pass 2 + contraction.  (Spreading the 36 instructions of a wave over the
        // pass-2 steps and the first contraction steps instead of issuing them back to back was measured: 4 % slower.)
        const int sn = s + 1;
        if (sn < s_end && !(G4_DBG & 8)) {
            int y0, x0;
            unsigned xb, gb;
            if (strip_origin(sn, y0, x0, xb, gb)) stage_interior(xb, gb);
            else stage_border(y0, x0, xb, gb);
        }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (!(G4_DBG & 2)) pass2(k);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!(G4_DBG & 4)) {
            float oa[2][3], ob[2][3];
            g4_load_ops<0>(vlane, oa[0], ob[0]);
            g4_contract<0>(acc, vlane, oa, ob, [](auto) {});
        }
        s = sn;
    }

    // ------------------------------------------------------------------ partial sums -> slab[split][f][KP][NcP]
    // local frequency fl = a' * 6 + b  ->  global row {0,5} / {1,2} / {3,4}[a']
    float* const sl = slab + (int64_t)split * 36 * g.KP * d.NcP;
#pragma unroll
    for (int fi = 0; fi < 3; ++fi) {
        const int fl = 3 * wave + fi;
        const int ap = fl >= 6 ? 1 : 0, b = fl - 6 * ap;
        const int ga = fg == 0 ? (ap ? 5 : 0) : 2 * fg - 1 + ap;
        float* const pf = sl + (int64_t)(ga * 6 + b) * g.KP * d.NcP;
#pragma unroll
        for (int mi = 0; mi < 3; ++mi)
#pragma unroll
            for (int nj = 0; nj < 3; ++nj)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ci = cib * 96 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    pf[(int64_t)ci * d.NcP + cob * 96 + nj * 32 + l31] = acc[(fi * 3 + mi) * 3 + nj][r];
                }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int wgrad43_plan(const adh_conv_desc* d, int nsplit, Wg43Args* a) {
    static const bool enabled = !(getenv("ADH_WINO43_WGRAD") && getenv("ADH_WINO43_WGRAD")[0] == '0');   // A/B switch
    if (!enabled || !d) return 0;
    if (d->KH != 3 || d->KW != 3 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1) return 0;
    if (d->out_oy != 0 || d->out_ox != 0 || d->dy0 != -1 || d->dx0 != -1 || d->dstep_y != 1 || d->dstep_x != 1) return 0;
    if (d->Cin % 96 != 0 || d->NcP % 96 != 0 || d->Cout != d->NcP || (d->in_cstride & 3) || (d->out_cstride & 3)) return 0;
    if (d->in_cstride < d->Cin || d->out_cstride < d->Cout) return 0;
    if (d->VH != d->IH || d->VW != d->IW || d->VH != d->OH || d->VW != d->OW || (d->VH & 3) || (d->VW & 3)) return 0;
    if ((int64_t)d->N * (d->IH + 2) * d->IW * d->in_cstride >= (1ll << 30) || (int64_t)d->N * d->VH * d->VW * d->out_cstride >= (1ll << 30))
        return 0;
    a->TY = d->VH / 4;
    a->SX = adh_ceil_div(d->VW / 4, G4_T);
    a->S = d->N * a->TY * a->SX;
    a->nsplit = nsplit;
    a->ncob = d->NcP / 96;
    a->ngroups = (d->Cin / 96) * a->ncob;
    a->KP = d->Cin;
    return 1;
}

// workgroups per pixel split (channel-block pairs x three frequency groups); 0 = not this path
extern "C" int adh_conv_wgrad_wino43_groups(const adh_conv_desc* d) {
    Wg43Args a;
    return wgrad43_plan(d, 1, &a) ? 3 * a.ngroups : 0;
}

extern "C" int adh_conv_wgrad_wino43(void* stream, const adh_conv_desc* d, float* slab, int nsplit) {
    if (!d || !slab || nsplit < 1 || !d->in || !d->out) return ADH_E_ARG;
    Wg43Args a;
    if (!wgrad43_plan(d, nsplit, &a)) return ADH_E_UNSUPPORTED;
    if (((uintptr_t)d->in & 15) || ((uintptr_t)d->out & 15)) return ADH_E_ARG;
    const int nblocks = ((nsplit + 7) / 8) * 3 * a.ngroups * 8;
    hipStream_t s = (hipStream_t)stream;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_wino43_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    hipLaunchKernelGGL(conv_wgrad_wino43_kernel, dim3(nblocks), dim3(256), G4_LDS_BYTES, s, *d, a, slab);
    return adh_check_launch();
}

// dst(layout L, 3x3) (+)= A'^T (slab[0][a*6+b] / (N_a N_b)) A'   (slab[0] = sum over splits)
__global__ void wgrad_reduce_wino43_kernel(const float* __restrict__ slab, int KP, int NcP, const adh_wlayout L,
                                           float* __restrict__ dst, int accumulate) {
    const int64_t total = (int64_t)L.K * L.Nc;
    const int64_t fstride = (int64_t)KP * NcP;
    const double a = G4_A, b = G4_B;
    const double n0 = a * a * b * b, na = 2.0 * a * a * (a * a - b * b), nb = 2.0 * b * b * (b * b - a * a);
    const double inv[6] = {1.0 / n0, 1.0 / na, 1.0 / na, 1.0 / nb, 1.0 / nb, 1.0};
    const double AT[3][6] = {{1, 1, 1, 1, 1, 0}, {0, a, -a, b, -b, 0}, {0, a * a, a * a, b * b, b * b, 1}};
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % L.Nc);
        const int k = (int)(idx / L.Nc);
        const float* p = slab + (int64_t)k * NcP + n;
        float u[36];
#pragma unroll
        for (int f = 0; f < 36; ++f) u[f] = p[f * fstride];
        double t[3][6];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int bb = 0; bb < 6; ++bb) {
                double acc = 0.0;
#pragma unroll
                for (int aa = 0; aa < 6; ++aa) acc += AT[i][aa] * inv[aa] * (double)u[aa * 6 + bb];
                t[i][bb] = acc * inv[bb];
            }
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                double w = 0.0;
#pragma unroll
                for (int bb = 0; bb < 6; ++bb) w += t[i][bb] * AT[j][bb];
                const int64_t off = (int64_t)L.tap_off0 + i * L.tap_off_sy + j * L.tap_off_sx + (int64_t)k * L.stride_k +
                                    (int64_t)n * L.stride_n;
                dst[off] = accumulate ? dst[off] + (float)w : (float)w;
            }
    }
}

extern "C" int adh_wgrad_reduce_wino43(void* stream, float* slab, int nsplit, int KP, int NcP, const adh_wlayout* L,
                                       float* dst, int accumulate) {
    if (!slab || !L || !dst || nsplit < 1 || L->KHt != 3 || L->KWt != 3 || (NcP & 3)) return ADH_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (nsplit > 1) {
        const int64_t n4 = (int64_t)36 * KP * NcP / 4;
        adh_wgrad_sum_splits(s, slab, nsplit, n4);
    }
    const int64_t total = (int64_t)L->K * L->Nc;
    hipLaunchKernelGGL(wgrad_reduce_wino43_kernel, dim3(adh_min_i(adh_ceil_div(total, 64), 16384)), dim3(64), 0, s, slab, KP, NcP,
                       *L, dst, accumulate);
    return adh_check_launch();
}
