// Weight gradient of the gather-form convolution on fp32 MFMA, gfx950.
//
// dW[tap][k][n] = sum over virtual pixels m of X_tap[m][k] * G[m][n], where X_tap[m] is the input
// pixel tap `tap` reads for virtual pixel m and G is the gradient wrt the conv output (adh_conv_desc,
// with d->out read as G).  This is the autograd backward ATen runs for Conv2d / ConvTranspose2d in
// the reference's training step (/root/reference training/train_joint.py:153, train_dehazing.py:96).
//
// Three kernels, chosen by adh_conv_wgrad / adh_conv_wgrad_wino:
//   * conv_wgrad_rows_kernel<KH,KW,REV,TN>          -- "row split": 3x3 s1 and the 2x2-tap forms on 4x32-aligned grids
//                                                      (Cin % 32 == 0): k split over 4 waves, LDS-DMA staging, pinned
//                                                      accumulators; see the block comment above that kernel;
//   * conv_wgrad_rows_kernel<3,3,false,TN,WINO=true> -- the same skeleton accumulating in the Winograd F(2x2,3x3) domain
//                                                      (4/9 of the MFMA work, operands transformed in registers);
//   * conv_wgrad_kernel<T,TN> (first in this file)   -- general fallback: any tap count / ragged grids / Cin % 32 != 0 and
//                                                      the packed 7x7 stem.  A workgroup owns one 32-wide input-channel
//                                                      tile x 32*TN output channels x T taps and sweeps pixel tiles (TH rows
//                                                      x 32 columns); waves 0-3 contract from LDS while waves 4-7 stage the
//                                                      next tile through registers.
// In all of them the input halo [pixels][32 ch] and the G tile [pixels][32*TN] sit in LDS pixel-major, so the MFMA
// operands -- A[i=channel][k=pixel], B[k=pixel][j=channel] -- are conflict-free ds_read_b32 with consecutive lanes on
// consecutive channels and every tap is an address offset; the pixel dimension is split over workgroups that write
// partial slabs, summed in a fixed order by the reduce kernels (deterministic, no atomics).
#include "common.h"
#include <cstdlib>

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

// ---------------------------------------------------------------------------------------------------------
// "Row split" fast path.  Measured on gfx950: every VALU instruction issued on a SIMD takes ~5-8 cycles away
// from its fp32 MFMA pipe (tools/micro/mfma_mix.hip), so this kernel keeps the vector ALU idle in steady state:
//   * 4 waves (one per SIMD, the whole 512-register file each).  A workgroup owns 32 input channels x 32*TN
//     output channels x KH*KW taps and sweeps a contiguous range of pixel tiles (4 rows x 32 columns); wave w
//     contracts pixel row w of every tile into ALL KH*KW*TN accumulator tiles, i.e. the pixel (k) dimension
//     is split over the waves and every wave runs the same instruction stream;
//   * the input halo [4+KH-1][40 px][32 ch] and the G tile [TN][128 px][32 ch] arrive by LDS-DMA
//     (buffer_load_dwordx4 ... lds: per-lane offset constant for the whole kernel, per-instruction scalar
//     offset), double buffered, one barrier per tile, no staging registers and no ds_write;
//   * every MFMA operand is a ds_read_b32 whose tap / step / n-tile offset is an instruction immediate.
// Out-of-image halo rows are zero-filled instead of loaded; out-of-image halo columns are lane-masked and
// zero-filled (buffer range checking does not cover the scalar offset, so nothing is read out of bounds).
// Wave w of pixel split s writes slab[4*s + w].
//
// Shapes: taps step through the input in units of the per-pixel input stride (|dstep| == in_s), so the halo
// is a dense patch of the in_s-subsampled input.  That covers Conv2d 3x3 s1 (3x3 taps), every output-parity
// class of ConvTranspose2d k4 s2 (2x2 taps walked backwards, G strided by 2) and Conv2d k4 s2, which the host
// splits into its four kernel-parity classes (2x2 taps with step 2 on the stride-2 input grid).
#define WR_HP 40   // halo row pitch in pixels (5 DMA pieces of 8 pixels; 32+KW-1 used)
#define WR_TH 4

typedef __attribute__((address_space(3))) void* lds_void_ptr;

#ifdef WR_PROF   // dev build (tools/prof_wino43.sh): per-workgroup s_memtime stamps of conv_wgrad_rows_kernel and the CU it ran on
__device__ unsigned long long wr_prof_buf[16384 * 32];
#define WR_STAMP(i) do { if (tid == 0 && bid < 16384) wr_prof_buf[bid * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
extern "C" int adh_wr_prof_read(void* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(wr_prof_buf), sizeof(wr_prof_buf)) == hipSuccess ? 0 : -1;
}
#else
#define WR_STAMP(i) do {} while (0)
#endif

struct WrArgs {
    int ymin, xmin;            // input pixel of halo (0,0) for virtual pixel (0,0)
    int xps;                   // input pixels per halo pixel (in_s)
    int Ttot;                  // taps of the slab (KH*KW of the full kernel)
    int tap0, tap_sy, tap_sx;  // slab tap index of this launch's tap (ty, tx)
    int ntiles, nsplit, ngroups, nco_groups, tiles_x, tiles_y;
};

// Accumulator tile IDX of a wave: the first 16 tiles live in the 256 AGPRs, the rest in VGPRs.  hipcc would
// otherwise keep >256 accumulator registers in AGPRs and copy every tile to VGPRs and back around each MFMA
// (96 v_accvgpr moves per 18 MFMAs measured), so the register class is pinned through the asm constraint.
template <int IDX>
__device__ __forceinline__ void wr_mfma(f32x16& c, float a, float b) {
    if constexpr (IDX < 16) asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int T, int TN, int I>
__device__ __forceinline__ void wr_mfma_all(f32x16 (&acc)[T * TN], const float (&a)[T], const float (&b)[TN]) {
    if constexpr (I < T * TN) {
        wr_mfma<I>(acc[I], a[I / TN], b[I % TN]);
        wr_mfma_all<T, TN, I + 1>(acc, a, b);
    }
}

// Winograd mode: acc[b*TN + j] += V[b] (x) M[j][b] for the wave's four frequencies b and TN n-tiles
// (operands written by VALU instructions just before: the MFMA carries two wait states of its own -- a VALU -> MFMA
// hazard hipcc pads for MFMA instructions but cannot see inside inline asm; common.h, adh_mfma_operand_fence)
template <int TN, int I>
__device__ __forceinline__ void wr_mfma_wino(f32x16 (&acc)[4 * TN], const float (&V)[4], const float (&M)[TN][4]) {
    if constexpr (I < 4 * TN) {
        static_assert(4 * TN <= 16, "all Winograd-mode accumulators sit in AGPRs");
        asm("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[I]) : "v"(V[I / TN]), "v"(M[I % TN][I / TN]));
        wr_mfma_wino<TN, I + 1>(acc, V, M);
    }
}

// REV: taps walk backwards through the halo (dstep < 0): tap (ty, tx) reads halo offset (KH-1-ty, KW-1-tx)
// WINO (3x3 taps only): accumulate in the Winograd F(2x2,3x3) domain instead -- see the block comment at the
// contraction below; wave w then owns frequency row w and the slab holds 16 frequencies instead of 9 taps.
template <int KH, int KW, bool REV, int TN, bool WINO = false>
__global__ __launch_bounds__(256, 1) void conv_wgrad_rows_kernel(const adh_conv_desc d, const WrArgs g, float* slab) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    static_assert(!WINO || (KH == 3 && KW == 3 && !REV), "the Winograd weight gradient is a 3x3 form");
    constexpr int T = WINO ? 4 : KH * KW;          // accumulator tiles per n-tile (frequencies of this wave / taps)
    constexpr int HR = WR_TH + KH - 1;             // halo rows
    constexpr int NC = 32 + KW - 1;                // halo columns in use
    constexpr int NP = HR * 5;                     // DMA pieces of the halo
    constexpr int PU = (NP + 3) / 4;               // pieces per wave (at most)
    static_assert(PU <= 8, "halo piece tables hold 8 entries");
    constexpr int XF = HR * WR_HP * 32;            // floats of the halo image
    constexpr int GF = TN * WR_TH * 32 * 32;       // floats of the G image
    constexpr int BUF = XF + GF;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;

    // XCD-aware decode: the `ngroups` workgroups that share one pixel range (and so its G tiles) get ids that
    // are congruent mod 8, i.e. land on one XCD and share its L2
    const int bid = blockIdx.x;
    WR_STAMP(0);
#ifdef WR_PROF
    if (tid == 0 && bid < 16384) {
        wr_prof_buf[bid * 32 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_ID
        wr_prof_buf[bid * 32 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // XCC_ID
    }
#endif
    const int xcd = bid & 7;
    const int q = bid >> 3;
    const int grp = q % g.ngroups;
    const int split = (q / g.ngroups) * 8 + xcd;
    if (split >= g.nsplit) return;
    const int ci_tile = grp / g.nco_groups;
    const int co_grp = grp - ci_tile * g.nco_groups;
    const int ci0 = ci_tile * 32;
    const int co0 = co_grp * 32 * TN;

    // contiguous tile range of this split
    const int per = g.ntiles / g.nsplit, rem = g.ntiles - per * g.nsplit;
    const int t_begin = split * per + adh_min_i(split, rem);
    const int t_end = t_begin + per + (split < rem ? 1 : 0);

    // Buffer descriptors.  The halo descriptor's base is the (virtual) pixel (ymin, xmin) of the image, so every
    // scalar offset below is non-negative; lanes / rows that would fall outside the tensor are never issued.
    // Offsets are 32-bit and relative to the current image, whose base goes into the descriptor (rebuilt per
    // tile: four scalar instructions), so only one image has to stay below 2 GiB.
    const int xcs = d.in_cstride * 4 * g.xps, gcs = d.out_cstride * 4 * d.out_sx;   // halo / G pixel pitches in bytes
    const int xrs = d.IW * d.in_cstride * 4 * g.xps, grs = d.OW * d.out_cstride * 4 * d.out_sy;   // row pitches
    const float* xbase = d.in + ((int64_t)g.ymin * d.IW + g.xmin) * d.in_cstride + ci0;
    const float* gbase = d.out + ((int64_t)d.out_oy * d.OW + d.out_ox) * d.out_cstride + co0;
    const int64_t ximg = (int64_t)d.IH * d.IW * d.in_cstride, gimg = (int64_t)d.OH * d.OW * d.out_cstride;
    const int xv = (lane >> 3) * xcs + (lane & 7) * 16;    // per-lane byte offsets inside one 8-pixel piece
    const int gv = (lane >> 3) * gcs + (lane & 7) * 16;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const bool g_partial = co0 + 32 * TN > d.Cout;   // this group's last n-tile has channels beyond Cout

    // this wave's halo pieces: piece j = wave + 4u covers halo row j/5, pixels 8*(j%5) .. +7.  The wave-uniform row /
    // column / offsets of a piece are recomputed with a few scalar instructions where they are used: kept in 32 SGPRs
    // they were spilled to VGPR lanes and came back as ~40-80 v_readlane per tile -- vector instructions inside an
    // MFMA-bound loop, where scalar ones are free
    auto piece_row = [&](int u) { return ((wave + 4 * u) * 205) >> 10; };                    // (wave + 4u) / 5, j < 64
    auto piece_cb = [&](int u) { return (wave + 4 * u) - 5 * piece_row(u); };
    const int npieces = (NP - wave + 3) / 4;
    const bool tail_ok = (lane >> 3) < NC - 32;   // lanes of a right-most piece that hold used pixels

    // stage tile (n, ty, tx) into buffer `b`: this wave's share of the DMA pieces
    auto stage = [&](int n, int ty, int tx, int b) {
        float* xs = smem + b * BUF;
        float* gs = xs + XF;
        const __amdgpu_buffer_rsrc_t xr =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xbase + n * ximg), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t gr =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gbase + n * gimg), 0, 0x7fffffff, 0x00020000);
        // input pixel of halo (0,0) and of its last used row / column
        const int iy0 = ty * WR_TH * g.xps + g.ymin, ix0 = tx * 32 * g.xps + g.xmin;
        const int so = ty * WR_TH * xrs + tx * 32 * xcs;
        const bool interior = iy0 >= 0 && iy0 + (HR - 1) * g.xps < d.IH && ix0 >= 0 && ix0 + (NC - 1) * g.xps < d.IW;
        if (interior) {
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                if (u < npieces) {
                    const int pr = piece_row(u), pc = piece_cb(u);
                    float* dst = xs + (pr * WR_HP + pc * 8) * 32;
                    const int po = so + pr * xrs + pc * 8 * xcs;
                    if (pc == 4) {
                        if (tail_ok) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)dst, 16, xv, po, 0, 0);
                    } else {
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)dst, 16, xv, po, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                if (u < npieces) {
                    const int pr = piece_row(u), pc = piece_cb(u);
                    float* dst = xs + (pr * WR_HP + pc * 8) * 32;
                    const int po = so + pr * xrs + pc * 8 * xcs;
                    const int iy = iy0 + pr * g.xps;
                    const int c = pc * 8 + (lane >> 3);
                    const int ix = ix0 + c * g.xps;
                    const bool lane_ok = iy >= 0 && iy < d.IH && c < NC && ix >= 0 && ix < d.IW;
                    if (lane_ok) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)dst, 16, xv, po, 0, 0);
                    if (!lane_ok && c < NC) *reinterpret_cast<f32x4*>(dst + lane * 4) = zero4;
                }
            }
        }
        // G: wave w stages pixel row w: TN x 4 pieces (pixels always inside the tensor; the channel quads of a partial
        // last n-tile -- Cout % 32 != 0 -- are lane-masked: their LDS cells keep the zeros written at kernel start)
        const int sg = (ty * WR_TH + wave) * grs + tx * 32 * gcs;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                float* dst = gs + (tn * 128 + wave * 32 + cb * 8) * 32;
                if (!g_partial || co0 + tn * 32 + (lane & 7) * 4 < d.Cout)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lds_void_ptr)dst, 16, gv, sg + cb * 8 * gcs + tn * 128, 0, 0);
            }
    };
    if (g_partial) {
        for (int i = tid; i < 2 * BUF / 4; i += 256) reinterpret_cast<f32x4*>(smem)[i] = zero4;
        __syncthreads();
    }

    f32x16 acc[T * TN];
#pragma unroll
    for (int t = 0; t < T * TN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // tile coordinates advance incrementally (scalar)
    int tx = t_begin % g.tiles_x;
    int ty = (t_begin / g.tiles_x) % g.tiles_y;
    int n = t_begin / (g.tiles_x * g.tiles_y);
    if (t_begin < t_end) stage(n, ty, tx, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int cur = 0;
    WR_STAMP(1);
    for (int tile = t_begin; tile < t_end; ++tile) {
        int ntx = tx + 1, nty = ty, nn = n;
        if (ntx == g.tiles_x) { ntx = 0; ++nty; }
        if (nty == g.tiles_y) { nty = 0; ++nn; }
        if (tile == t_begin + 1) WR_STAMP(8);
        if (tile + 1 < t_end) stage(nn, nty, ntx, cur ^ 1);
        if (tile == t_begin + 1) WR_STAMP(9);

        if constexpr (!WINO) {
            const float* xl = smem + cur * BUF + (wave * WR_HP + h) * 32 + l31;
            const float* gl = smem + cur * BUF + XF + (wave * 32 + h) * 32 + l31;
            float a0[T], b0[TN], a1[T], b1[TN];
            auto ld = [&](const float* xp, const float* gp, float (&av)[T], float (&bv)[TN]) {
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int oy = REV ? KH - 1 - t / KW : t / KW, ox = REV ? KW - 1 - t % KW : t % KW;
                    av[t] = xp[(oy * WR_HP + ox) * 32];
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = gp[j * 128 * 32];
            };
            // software pipeline over the 16 pixel pairs of this wave's row: operands of pair s+1 are in flight while
            // the T*TN MFMAs of pair s issue (two register sets, pointers bumped once per two pairs)
            ld(xl, gl, a0, b0);
#pragma unroll 1
            for (int st = 0; st < 16; st += 2) {
                ld(xl + 64, gl + 64, a1, b1);
                wr_mfma_all<T, TN, 0>(acc, a0, b0);
                xl += 128;
                gl += 128;
                if (st + 2 < 16) ld(xl, gl, a0, b0);
                wr_mfma_all<T, TN, 0>(acc, a1, b1);
            }
        } else {
            // Winograd-domain weight gradient.  With Y = A^T[(G w G^T) . (B^T d B)]A per 2x2 output tile,
            //   dL/d(G w G^T)[f][ci][co] = sum over tiles of (B^T d B)[f][tile][ci] * (A gy A^T)[f][tile][co],
            // 16 products of k = tiles instead of 9 of k = pixels: 4/9 of the MFMA work; G^T(.)G is applied once, in
            // adh_wgrad_reduce_wino.  The 4x32-pixel tile is 2x16 Winograd tiles = 16 MFMA k-steps (lane half h takes
            // tile column 2s+h).  Wave w owns frequency row a = w.  Both operands are transformed in registers straight
            // from the raw LDS images in the lane layout the MFMA wants (lane = channel), so nothing is staged twice:
            //   A side: r[c] = d[rA][c] + sg*d[rB][c] (row a of B^T d; rows and sign are per-wave constants),
            //           V[b] = r0-r2, r1+r2, r2-r1, r1-r3                                         (8 VALU)
            //   B side: t[j] = gy[rP][j] + be*gy[1][j] (row a of A gy; rP = 1 and be = 0 for a = 3),
            //           M[b] = t0, t0+t1, t0-t1, t1                                    (4 VALU per n-tile)
            // The two minus signs of A's last row / column (a = 3: -gy1, b = 3: -t1) are left out here and applied by the
            // reduce kernel (dU[a][b] *= s_a s_b, s_3 = -1).
            const int rA = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
            const int rB = wave == 0 ? 2 : (wave == 1 ? 2 : (wave == 2 ? 1 : 3));
            const float sg = wave == 1 ? 1.f : -1.f;
            const int rP = wave == 3 ? 1 : 0;
            const float be = wave == 1 ? 1.f : (wave == 2 ? -1.f : 0.f);
            // volatile: one ds_read_b32 per element with a 16-bit immediate offset from ONE base register per image, instead
            // of ds_read2_b32 pairs whose 8-bit offsets force a base update (v_add_u32) every other step -- 56 VALU per tile
            // that came straight out of the MFMA time; 1.5 LDS reads per MFMA are free
            typedef const volatile __attribute__((address_space(3))) float* lds_vf;
            lds_vf xa = (lds_vf)(smem + cur * BUF + (rA * WR_HP + 2 * h) * 32 + l31);
            lds_vf xb = (lds_vf)(smem + cur * BUF + (rB * WR_HP + 2 * h) * 32 + l31);
            lds_vf gp = (lds_vf)(smem + cur * BUF + XF + (rP * 32 + 2 * h) * 32 + l31);
            lds_vf gq = (lds_vf)(smem + cur * BUF + XF + (32 + 2 * h) * 32 + l31);
            // Packed arithmetic: adjacent tile columns sit in register pairs (ds_read2_b32, second offset + one pixel), so
            // every transform step is one v_pk_* on two columns:
            //   (r0, r1), (r2, r3) = va + sg * vb                                  2 x v_pk_fma
            //   (V0, V1) = (r0 - r2, r1 + r2),  (V3, -V2) = (r1 - r3, r1 - r2)      2 x v_pk_add (lane selects + sign modifiers)
            //   (t0, t1) = gP + be * gQ,  (M1, -M2) = (t0 + t1, t1 - t0)           2 per n-tile
            // (V2 and M2 both carry a minus sign: their product does not.)  10 VALU per 12 MFMAs instead of 20.
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 ra[2][2], rb[2][2], gP[2][TN], gQ[2][TN];
            auto ld = [&](int tr, int sp, f32x2 (&va)[2], f32x2 (&vb)[2], f32x2 (&vp)[TN], f32x2 (&vq)[TN]) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    va[c] = f32x2{xa[((2 * tr) * WR_HP + 4 * sp + 2 * c) * 32], xa[((2 * tr) * WR_HP + 4 * sp + 2 * c + 1) * 32]};
                    vb[c] = f32x2{xb[((2 * tr) * WR_HP + 4 * sp + 2 * c) * 32], xb[((2 * tr) * WR_HP + 4 * sp + 2 * c + 1) * 32]};
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {   // output columns 0, 1 of the tile: rows P and Q of the wave's frequency row
                    vp[j] = f32x2{gp[(j * 128 + (2 * tr) * 32 + 4 * sp) * 32], gp[(j * 128 + (2 * tr) * 32 + 4 * sp + 1) * 32]};
                    vq[j] = f32x2{gq[(j * 128 + (2 * tr) * 32 + 4 * sp) * 32], gq[(j * 128 + (2 * tr) * 32 + 4 * sp + 1) * 32]};
                }
            };
            ld(0, 0, ra[0], rb[0], gP[0], gQ[0]);
#pragma unroll
            for (int st = 0; st < 16; ++st) {
                if (st + 1 < 16)
                    ld((st + 1) >> 3, (st + 1) & 7, ra[(st + 1) & 1], rb[(st + 1) & 1], gP[(st + 1) & 1], gQ[(st + 1) & 1]);
                const f32x2 r01 = sg * rb[st & 1][0] + ra[st & 1][0];
                const f32x2 r23 = sg * rb[st & 1][1] + ra[st & 1][1];
                f32x2 v01, v32;   // (V0, V1), (V3, -V2)
                asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(v01) : "v"(r01), "v"(r23));
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(v32) : "v"(r01), "v"(r23));
                float V[4] = {v01.x, v01.y, v32.y, v32.x};
                float M[TN][4];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f32x2 t = be * gQ[st & 1][j] + gP[st & 1][j];
                    f32x2 m12;    // (M1, -M2) = (t0 + t1, t1 - t0)
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[1,0]" : "=v"(m12) : "v"(t), "v"(t));
                    M[j][0] = t.x;
                    M[j][1] = m12.x;
                    M[j][2] = m12.y;
                    M[j][3] = t.y;
                }
                wr_mfma_wino<TN, 0>(acc, V, M);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (tile == t_begin + 1) WR_STAMP(10);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (tile == t_begin + 1) WR_STAMP(11);
        __builtin_amdgcn_s_barrier();
        if (tile == t_begin + 1) WR_STAMP(12);
        cur ^= 1;
        tx = ntx; ty = nty; n = nn;
    }
    WR_STAMP(2);

    const int KP = d.Cin;
    if constexpr (WINO) {
        // partial result -> slab[split][f = 4*wave + b][KP][NcP]
        float* sbase = slab + ((size_t)split * 16 + wave * 4) * KP * d.NcP;
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
    } else {
        // partial result -> slab[4*split + wave][tap][KP][NcP]
        float* sbase = slab + ((size_t)(split * 4 + wave) * g.Ttot) * KP * d.NcP;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int tap = g.tap0 + (t / KW) * g.tap_sy + (t % KW) * g.tap_sx;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float* base = sbase + ((size_t)tap * KP + ci0) * d.NcP + co0 + 32 * j + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
                    base[(size_t)i * d.NcP] = acc[t * TN + j][r];
                }
            }
        }
    }
    WR_STAMP(3);
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

// Row-split fast path: eligibility and launch plan.  Returns the number of sub-launches (0 = not eligible).
struct WrPlan {
    int KH, KW, rev, TN;
    WrArgs a;
};

static int wgrad_rows_plan(const adh_conv_desc* d, int nsplit, WrPlan plan[4]) {
    if (d->VW % 32 != 0 || d->VH % WR_TH != 0) return 0;
    if (d->Cin % 32 != 0 || d->Cout % 4 != 0 || d->NcP != adh_round_up(d->Cout, 32)) return 0;
    if (d->in_cstride % 4 != 0 || d->out_cstride % 4 != 0) return 0;
    if (d->in_sy != d->in_sx || d->dstep_y != d->dstep_x || d->out_sy != d->out_sx) return 0;
    if ((d->VH - 1) * d->out_sy + d->out_oy >= d->OH || (d->VW - 1) * d->out_sx + d->out_ox >= d->OW) return 0;
    // 32-bit byte offsets inside one image (the descriptors are per image)
    if ((int64_t)(d->IH + 8) * d->IW * d->in_cstride * 4 >= (int64_t)1 << 31) return 0;
    if ((int64_t)d->OH * d->OW * d->out_cstride * 4 >= (int64_t)1 << 31) return 0;
    const int t = d->NcP / 32;
    const int TN = t % 3 == 0 ? 3 : (t % 2 == 0 ? 2 : 1);
    WrArgs base;
    base.tiles_x = d->VW / 32;
    base.tiles_y = d->VH / WR_TH;
    base.ntiles = base.tiles_x * base.tiles_y * d->N;
    base.nsplit = nsplit;
    base.nco_groups = d->NcP / (32 * TN);
    base.ngroups = (d->Cin / 32) * base.nco_groups;
    base.Ttot = d->KH * d->KW;
    const int s = d->in_sy, ds = d->dstep_y;
    if (((d->KH == 3 && d->KW == 3) || (d->KH == 2 && d->KW == 2)) && (ds == s || ds == -s) && (s == 1 || s == 2)) {
        // dense patch of the s-subsampled input; reversed taps start (K-1) steps earlier
        WrPlan& p = plan[0];
        p.KH = d->KH; p.KW = d->KW; p.rev = ds < 0; p.TN = TN;
        p.a = base;
        p.a.xps = s;
        p.a.ymin = d->dy0 + (ds < 0 ? (d->KH - 1) * ds : 0);
        p.a.xmin = d->dx0 + (ds < 0 ? (d->KW - 1) * ds : 0);
        p.a.tap0 = 0; p.a.tap_sy = d->KW; p.a.tap_sx = 1;
        return 1;
    }
    if (d->KH == 4 && d->KW == 4 && s == 2 && ds == 1) {
        // Conv2d k4 s2: kernel index ky = 2*ty + py reads input row 2*(vy + ty) + dy0 + py -- for each kernel parity
        // (py, px) a 2x2-tap problem on the stride-2 input grid
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                WrPlan& p = plan[py * 2 + px];
                p.KH = 2; p.KW = 2; p.rev = 0; p.TN = TN;
                p.a = base;
                p.a.xps = 2;
                p.a.ymin = d->dy0 + py;
                p.a.xmin = d->dx0 + px;
                p.a.tap0 = py * 4 + px; p.a.tap_sy = 8; p.a.tap_sx = 2;
            }
        return 4;
    }
    return 0;
}

template <int KH, int KW, bool REV, int TN>
static int launch_wgrad_rows(hipStream_t s, const adh_conv_desc* d, const WrArgs& a, float* slab) {
    const int lds = 2 * (((WR_TH + KH - 1) * WR_HP + TN * WR_TH * 32) * 32 * 4);
    const int nblocks = ((a.nsplit + 7) / 8) * a.ngroups * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_rows_kernel<KH, KW, REV, TN>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv_wgrad_rows_kernel<KH, KW, REV, TN>), dim3(nblocks), dim3(256), lds, s, *d, a, slab);
    return adh_check_launch();
}

static int dispatch_wgrad_rows(hipStream_t s, const adh_conv_desc* d, const WrPlan& p, float* slab) {
#define WR_CASE(kh_, kw_, rev_, tn_) \
    if (p.KH == kh_ && p.KW == kw_ && p.rev == rev_ && p.TN == tn_) \
        return launch_wgrad_rows<kh_, kw_, rev_ != 0, tn_>(s, d, p.a, slab);
    WR_CASE(3, 3, 0, 3) WR_CASE(3, 3, 0, 2) WR_CASE(3, 3, 0, 1)
    WR_CASE(2, 2, 0, 3) WR_CASE(2, 2, 0, 2) WR_CASE(2, 2, 0, 1)
    WR_CASE(2, 2, 1, 3) WR_CASE(2, 2, 1, 2) WR_CASE(2, 2, 1, 1)
#undef WR_CASE
    return ADH_E_UNSUPPORTED;
}

static bool wgrad_rows_instantiated(const WrPlan& p) {
    return (p.KH == 3 && p.KW == 3 && !p.rev) || (p.KH == 2 && p.KW == 2);
}

// ---- Winograd-domain weight gradient of 3x3 unit-stride convolutions (conv_wgrad_rows_kernel<3,3,0,TN,true>) ----
static int wgrad_wino_plan(const adh_conv_desc* d, int nsplit, WrPlan* p) {
    static const bool enabled = !(getenv("ADH_WINO_WGRAD") && atoi(getenv("ADH_WINO_WGRAD")) == 0);   // A/B switch
    if (!enabled) return 0;
    WrPlan plan[4];
    if (wgrad_rows_plan(d, nsplit, plan) != 1) return 0;
    if (plan[0].KH != 3 || plan[0].KW != 3 || plan[0].rev || plan[0].a.xps != 1) return 0;
    if (d->dy0 != -1 || d->dx0 != -1) return 0;
    *p = plan[0];
    return 1;
}

extern "C" int adh_conv_wgrad_wino_groups(const adh_conv_desc* d) {
    if (!d) return ADH_E_ARG;
    WrPlan p;
    return wgrad_wino_plan(d, 1, &p) ? p.a.ngroups : 0;
}

extern "C" int adh_conv_wgrad_wino(void* stream, const adh_conv_desc* d, float* slab, int nsplit) {
    if (!d || !slab || nsplit < 1 || !d->in || !d->out) return ADH_E_ARG;
    WrPlan p;
    if (!wgrad_wino_plan(d, nsplit, &p)) return ADH_E_UNSUPPORTED;
    const int lds = 2 * (((WR_TH + 2) * WR_HP + p.TN * WR_TH * 32) * 32 * 4);
    const int nblocks = ((nsplit + 7) / 8) * p.a.ngroups * 8;
    hipStream_t s = (hipStream_t)stream;
#define WW_CASE(tn_) \
    if (p.TN == tn_) { \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_rows_kernel<3, 3, false, tn_, true>), \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((conv_wgrad_rows_kernel<3, 3, false, tn_, true>), dim3(nblocks), dim3(256), lds, s, *d, p.a, slab); \
        return adh_check_launch(); \
    }
    WW_CASE(3) WW_CASE(2) WW_CASE(1)
#undef WW_CASE
    return ADH_E_UNSUPPORTED;
}

// dst(layout L, 3x3) (+)= G^T (sum over splits of slab[s][16][KP][NcP], with the two deferred signs) G
__global__ void wgrad_reduce_wino_kernel(const float* __restrict__ slab, int nsplit, int KP, int NcP, const adh_wlayout L,
                                         float* __restrict__ dst, int accumulate) {
    const int64_t total = (int64_t)L.K * L.Nc;
    const int64_t fstride = (int64_t)KP * NcP, split_stride = 16 * fstride;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % L.Nc);
        const int k = (int)(idx / L.Nc);
        const float* p = slab + (int64_t)k * NcP + n;
        // 16 frequencies x nsplit partial sums: keep 32-64 independent loads in flight (the inner loops are over
        // frequencies, the splits advance two at a time) -- with few (k, n) pairs this kernel is latency-bound
        float ua[16], ub[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) ua[f] = ub[f] = 0.f;
        int sp = 0;
        for (; sp + 2 <= nsplit; sp += 2) {
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                ua[f] += p[(int64_t)sp * split_stride + f * fstride];
                ub[f] += p[(int64_t)(sp + 1) * split_stride + f * fstride];
            }
        }
        if (sp < nsplit) {
#pragma unroll
            for (int f = 0; f < 16; ++f) ua[f] += p[(int64_t)sp * split_stride + f * fstride];
        }
        float u[4][4];
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            const float sign = ((f >> 2) == 3) != ((f & 3) == 3) ? -1.f : 1.f;
            u[f >> 2][f & 3] = sign * (ua[f] + ub[f]);
        }
        // G^T u G with G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
        float t[3][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            t[0][b] = u[0][b] + 0.5f * (u[1][b] + u[2][b]);
            t[1][b] = 0.5f * (u[1][b] - u[2][b]);
            t[2][b] = 0.5f * (u[1][b] + u[2][b]) + u[3][b];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float w[3];
            w[0] = t[i][0] + 0.5f * (t[i][1] + t[i][2]);
            w[1] = 0.5f * (t[i][1] - t[i][2]);
            w[2] = 0.5f * (t[i][1] + t[i][2]) + t[i][3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int64_t off = (int64_t)L.tap_off0 + i * L.tap_off_sy + j * L.tap_off_sx + (int64_t)k * L.stride_k +
                                    (int64_t)n * L.stride_n;
                dst[off] = accumulate ? dst[off] + w[j] : w[j];
            }
        }
    }
}

// slab[0] = sum over splits, in a fixed order (deterministic).  Block = 64 elements (16 bytes each) x 4 split groups: group y
// adds splits y, y + 4, .. with eight independent loads in flight, the four partial sums meet in LDS.  (One thread per
// element walking all the splits left the 96-channel layers -- 36,864 elements, 144 blocks -- latency-bound: 42 us.)
__global__ __launch_bounds__(256) void wgrad_sum_splits_kernel(float* __restrict__ slab, int nsplit, int64_t n4) {
    __shared__ f32x4 part[3][64];
    f32x4* s4 = reinterpret_cast<f32x4*>(slab);
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    for (int64_t i0 = blockIdx.x * (int64_t)64; i0 < n4; i0 += (int64_t)gridDim.x * 64) {
        const int64_t i = i0 + x;
        f32x4 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < n4) {
            int sp = y;
            for (; sp + 28 < nsplit; sp += 32) {
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] += s4[(int64_t)(sp + 4 * u) * n4 + i];
            }
            for (; sp < nsplit; sp += 4) a[0] += s4[(int64_t)sp * n4 + i];
        }
        const f32x4 t = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        if (y) part[y - 1][x] = t;
        __syncthreads();
        if (y == 0 && i < n4) s4[i] = ((t + part[0][x]) + (part[1][x] + part[2][x]));
        __syncthreads();
    }
}

// shared by the other Winograd-domain reduce entry points (conv_wgrad32.hip, conv_wgrad43.hip)
void adh_wgrad_sum_splits(hipStream_t s, float* slab, int nsplit, int64_t n4) {
    hipLaunchKernelGGL(wgrad_sum_splits_kernel, dim3(adh_min_i(adh_ceil_div(n4, 64), 4096)), dim3(256), 0, s, slab, nsplit, n4);
}

extern "C" int adh_wgrad_reduce_wino(void* stream, float* slab, int nsplit, int KP, int NcP, const adh_wlayout* L,
                                     float* dst, int accumulate) {
    if (!slab || !L || !dst || nsplit < 1 || L->KHt != 3 || L->KWt != 3 || (NcP & 3)) return ADH_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (nsplit > 1) {
        // many splits x few (k, n) pairs would leave the transform kernel latency-bound: stream-sum the splits first
        const int64_t n4 = (int64_t)16 * KP * NcP / 4;
        adh_wgrad_sum_splits(s, slab, nsplit, n4);
    }
    const int64_t total = (int64_t)L->K * L->Nc;
    hipLaunchKernelGGL(wgrad_reduce_wino_kernel, dim3(adh_min_i(adh_ceil_div(total, 64), 16384)), dim3(64), 0, s, slab, 1,
                       KP, NcP, *L, dst, accumulate);
    return adh_check_launch();
}

extern "C" int adh_conv_wgrad_groups(const adh_conv_desc* d) {
    if (!d) return ADH_E_ARG;
    WrPlan plan[4];
    const int n = wgrad_rows_plan(d, 1, plan);
    return n && wgrad_rows_instantiated(plan[0]) ? plan[0].a.ngroups : 0;
}

extern "C" int adh_conv_wgrad_slabs(const adh_conv_desc* d, int nsplit) {
    if (!d || nsplit < 1) return ADH_E_ARG;
    return adh_conv_wgrad_groups(d) > 0 ? 4 * nsplit : nsplit;
}

extern "C" int adh_conv_wgrad(void* stream, const adh_conv_desc* d, float* slab, int nsplit) {
    if (!d || !slab || nsplit < 1 || !d->in || !d->out) return ADH_E_ARG;
    if (d->NcP % 32 != 0 || d->NcP < d->Cout) return ADH_E_ARG;
    {
        WrPlan plan[4];
        const int np = wgrad_rows_plan(d, nsplit, plan);
        if (np && wgrad_rows_instantiated(plan[0])) {
            for (int i = 0; i < np; ++i) {
                const int rc = dispatch_wgrad_rows((hipStream_t)stream, d, plan[i], slab);
                if (rc) return rc;
            }
            return ADH_OK;
        }
    }
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
