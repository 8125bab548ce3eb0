// Winograd F(2x2, 3x3) convolution on fp32 MFMA, gfx950: 3x3 stride-1 pad-1 convolutions (and their data
// gradients, which are 3x3 stride-1 convolutions with flipped / transposed weights) at 2.25x fewer MFMA
// FLOPs than the direct gather kernel.  Same descriptor, same fused epilogue (scale/shift, residual, ReLU,
// BatchNorm partial statistics) as conv_igemm.hip; replaces the same ATen conv2d calls
// (/root/reference models/dehazing/base_model.py:11-13,26-41).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// Workgroup = 512 threads: output region 8 rows x 32 cols = 4 x 16 Winograd tiles (two 32-tile MFMA row blocks)
// x 32 output channels.
//   waves 4-7 (producers): stage the region's raw 10x34-pixel halo of the next 16-channel chunk in LDS (coalesced,
//       each input element fetched once), then one (tile, channel quad) per thread: read the 4x4x4 patch from the
//       raw tile, apply B^T d B in registers, write the 16 frequency planes to LDS
//       V[buf][freq][cquad][tile][4]  (16-byte tile pitch: conflict-free ds_read_b128 for the MFMA operand);
//   waves 0-3 (consumers): 4 of the 16 frequencies each; per chunk of 16 input channels 64 MFMAs against the
//       transformed weights U[freq][k/4][n][4] read straight from global/L2.  Two barriers per chunk (raw tile
//       written | patches transformed); V is double buffered so chunk c+1 is produced while chunk c is contracted.
//   epilogue: accumulators -> LDS M[freq][tile][co], all 512 threads apply A^T M A and the fused epilogue.
#include "common.h"

#define WN_KC 16                 // input channels per chunk
#define WN_TILES 64              // Winograd tiles per workgroup
#define WN_VBUF (16 * (WN_KC / 4) * WN_TILES)   // float4 per V buffer (65536 B)
#define WN_RAW_W 34
#define WN_RAW_PX (10 * WN_RAW_W)               // raw halo pixels of one 8x32 region
#define WN_RAW_IT 6                             // ceil(340 * 4 / 256) float4 per producer thread

// Weight prefetch outside hipcc's waitcnt bookkeeping (cdna_hip_programming.md 5.7): hipcc waits for a plain
// prefetch at the first MFMA of the SAME chunk (a conservative vmcnt on the loop-carried register set), exposing the
// L2 latency every chunk.  The asm load is invisible to that pass; wait_b() is the hand-placed wait, naming every
// destination so no consumer is scheduled above it.  The consumer waves issue no other vector-memory operation
// inside the chunk loop, so vmcnt(0) there waits exactly for the set issued one chunk earlier.
__device__ __forceinline__ void gload_b128(f32x4& dst, const f32x4* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void wait_b(f32x4 (&b)[4][2]) {
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[3][0]),
                   "+v"(b[3][1])
                 :
                 : "memory");
}

struct WinoGeom {
    int tiles_x, tiles_y;        // 32-col x 8-row regions
    int nchunks;
    int KQtot;                   // Cin / 4
};

__global__ __launch_bounds__(512, 2) void conv_wino_kernel(const adh_conv_desc d, const WinoGeom g) {
    extern __shared__ __attribute__((aligned(16))) f32x4 lds[];   // V[2][16][KC/4][64] ; reused as M[16][64][32] floats

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;

    int reg = blockIdx.x;
    const int tx = reg % g.tiles_x;
    reg /= g.tiles_x;
    const int ty = reg % g.tiles_y;
    const int n = reg / g.tiles_y;
    const int oy0 = ty * 8, ox0 = tx * 32;
    const int co0 = blockIdx.y * 32;

    f32x4* raw = lds + 2 * WN_VBUF;   // [KC/4][340 pixels] float4
    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int pt = tid - 256;
        // one producer wave per channel quad, lanes = the 64 tiles: the V writes of a wave are 64 consecutive
        // 16-B slots (conflict free); the raw tile is quad-major so the patch reads are at worst 2-way
        const int tile = pt & 63, cq = pt >> 6;
        const int trow = tile >> 4, tcol = tile & 15;
        const float* in_n = d.in + (size_t)n * d.IH * d.IW * d.in_cstride;
        // raw staging plan: float4 #item = pixel*4 + cquad of the 10x34 halo (origin oy0-1, ox0-1)
        int goff[WN_RAW_IT];
#pragma unroll
        for (int it = 0; it < WN_RAW_IT; ++it) {
            const int item = pt + it * 256;
            int o = -1;
            if (item < WN_RAW_PX * 4) {
                const int pix = item >> 2, q = item & 3;
                const int hy = pix / WN_RAW_W, hx = pix - hy * WN_RAW_W;
                const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
                if (iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW) o = (iy * d.IW + ix) * d.in_cstride + q * 4;
            }
            goff[it] = o;
        }
        f32x4 rr[WN_RAW_IT];
        auto load_raw = [&](int c) {
            const float* base = in_n + c * WN_KC;
#pragma unroll
            for (int it = 0; it < WN_RAW_IT; ++it) {
                // unconditional load (out-of-image lanes read the tensor base and are zeroed when stored), so the six
                // loads stay in flight across the next barrier instead of being waited for here
                rr[it] = *reinterpret_cast<const f32x4*>(goff[it] >= 0 ? base + goff[it] : in_n);
            }
        };
        auto store_raw = [&]() {
#pragma unroll
            for (int it = 0; it < WN_RAW_IT; ++it) {
                const int item = pt + it * 256;
                if (item < WN_RAW_PX * 4)
                    raw[(item & 3) * WN_RAW_PX + (item >> 2)] = goff[it] >= 0 ? rr[it] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        };
        const f32x4* patch = raw + cq * WN_RAW_PX + (2 * trow) * WN_RAW_W + 2 * tcol;
        auto transform = [&](int c) {
            f32x4 t[16];
#pragma unroll
            for (int b = 0; b < 4; ++b) {      // rows:  B^T d
                const f32x4 d0 = patch[0 * WN_RAW_W + b], d1 = patch[1 * WN_RAW_W + b];
                const f32x4 d2 = patch[2 * WN_RAW_W + b], d3 = patch[3 * WN_RAW_W + b];
                t[0 * 4 + b] = d0 - d2;
                t[1 * 4 + b] = d1 + d2;
                t[2 * 4 + b] = d2 - d1;
                t[3 * 4 + b] = d1 - d3;
            }
            f32x4* Vb = lds + (c & 1) * WN_VBUF + cq * WN_TILES + tile;
#pragma unroll
            for (int a = 0; a < 4; ++a) {      // columns: (B^T d) B
                Vb[(a * 4 + 0) * (WN_KC / 4) * WN_TILES] = t[a * 4 + 0] - t[a * 4 + 2];
                Vb[(a * 4 + 1) * (WN_KC / 4) * WN_TILES] = t[a * 4 + 1] + t[a * 4 + 2];
                Vb[(a * 4 + 2) * (WN_KC / 4) * WN_TILES] = t[a * 4 + 2] - t[a * 4 + 1];
                Vb[(a * 4 + 3) * (WN_KC / 4) * WN_TILES] = t[a * 4 + 1] - t[a * 4 + 3];
            }
        };
        // prologue: chunk 0 through the raw tile, chunk 1 already in flight
        load_raw(0);
        store_raw();
        if (g.nchunks > 1) load_raw(1);
        __syncthreads();       // P: raw(0) complete
        transform(0);
        __syncthreads();       // b_0: V[0] visible
        for (int c = 0; c < g.nchunks; ++c) {
            // consumers contract chunk c; meanwhile produce chunk c+1
            if (c + 1 < g.nchunks) store_raw();              // raw(c+1) from the registers loaded one interval ago
            if (c + 2 < g.nchunks) load_raw(c + 2);
            __syncthreads();   // m_c: raw(c+1) complete
            if (c + 1 < g.nchunks) transform(c + 1);
            __syncthreads();   // b_{c+1}: V[(c+1)&1] visible (after the last chunk: E1, everyone done with V)
        }
        __syncthreads();       // E2: accumulators are in LDS
    } else {
        // ------------------------------------------------------------------ consumers
        f32x16 acc[4][2];
#pragma unroll
        for (int fl = 0; fl < 4; ++fl)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[fl][mb][r] = 0.f;
        const f32x4* U4 = reinterpret_cast<const f32x4*>(d.wp);
        f32x4 bA[4][2], bB[4][2];   // two weight register sets, ping-ponged over an unrolled-by-two chunk loop
        auto fetch_b = [&](int c, f32x4 (&b)[4][2]) {
#pragma unroll
            for (int fl = 0; fl < 4; ++fl)
#pragma unroll
                for (int gg = 0; gg < 2; ++gg)
                    gload_b128(b[fl][gg],
                               U4 + (size_t)((wave * 4 + fl) * g.KQtot + c * (WN_KC / 4) + 2 * gg + h) * d.NcP + co0 + l31);
        };
        auto contract = [&](int c, const f32x4 (&bw)[4][2]) {
            const f32x4* Vb = lds + (c & 1) * WN_VBUF + h * WN_TILES + l31;
            // 4 double steps: two frequencies x one channel group each, i.e. 16 MFMAs on FOUR accumulators issued
            // round-robin (dependent MFMAs are 4 issue slots apart); the LDS operands of double step s+1 are read
            // before the MFMAs of double step s issue
            auto rd = [&](int ds, f32x4 (&a)[4]) {
                const int p = ds >> 1, gg = ds & 1;
                const f32x4* v0 = Vb + ((wave * 4 + 2 * p) * (WN_KC / 4) + 2 * gg) * WN_TILES;
                const f32x4* v1 = v0 + (WN_KC / 4) * WN_TILES;
                a[0] = v0[0];
                a[1] = v0[32];
                a[2] = v1[0];
                a[3] = v1[32];
            };
            f32x4 ac[4], an[4];
            rd(0, ac);
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const int p = ds >> 1, gg = ds & 1;
                if (ds == 1) __syncthreads();   // m_c
                if (ds < 3) rd(ds + 1, an);
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 b0 = bw[2 * p][gg], b1 = bw[2 * p + 1][gg];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[2 * p][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[0][j], b0[j], acc[2 * p][0], 0, 0, 0);
                    acc[2 * p][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[1][j], b0[j], acc[2 * p][1], 0, 0, 0);
                    acc[2 * p + 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[2][j], b1[j], acc[2 * p + 1][0], 0, 0, 0);
                    acc[2 * p + 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[3][j], b1[j], acc[2 * p + 1][1], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) ac[i] = an[i];
            }
            __syncthreads();   // b_{c+1} (E1 after the last chunk)
        };
        fetch_b(0, bA);
        __syncthreads();       // P
        __syncthreads();       // b_0
        for (int c = 0; c < g.nchunks; c += 2) {
            wait_b(bA);                                   // issued one chunk ago: already landed
            if (c + 1 < g.nchunks) fetch_b(c + 1, bB);   // next chunk's weights stay in flight during this contraction
            __builtin_amdgcn_sched_barrier(0);
            contract(c, bA);
            if (c + 1 < g.nchunks) {
                wait_b(bB);
                if (c + 2 < g.nchunks) fetch_b(c + 2, bA);
                __builtin_amdgcn_sched_barrier(0);
                contract(c + 1, bB);
            }
        }
        float* M = reinterpret_cast<float*>(lds);   // [16][64][32]
#pragma unroll
        for (int fl = 0; fl < 4; ++fl)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = (r & 3) + 8 * (r >> 2) + 4 * h + 32 * mb;
                    M[((wave * 4 + fl) * WN_TILES + t) * 32 + l31] = acc[fl][mb][r];
                }
        __syncthreads();       // E2
    }

    // ---------------------------------------------------------------------- output transform + fused epilogue
    const float* M = reinterpret_cast<const float*>(lds);
    const int cl = tid & 31;
    const int co = co0 + cl;
    const bool cvalid = co < d.Cout;
    const float sc = (d.scale && cvalid) ? d.scale[co] : 1.f;
    const float sh = (d.shift && cvalid) ? d.shift[co] : 0.f;
    float* out_n = d.out + (size_t)n * d.OH * d.OW * d.out_cstride;
    const float* res_n = d.residual ? d.residual + (size_t)n * d.OH * d.OW * d.res_cstride : nullptr;
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int t = (tid >> 5) + 16 * p;
        const int trow = t >> 4, tcol = t & 15;
        float m[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) m[f] = M[(f * WN_TILES + t) * 32 + cl];
        float s0[4], s1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            s0[b] = m[0 * 4 + b] + m[1 * 4 + b] + m[2 * 4 + b];
            s1[b] = m[1 * 4 + b] - m[2 * 4 + b] - m[3 * 4 + b];
        }
        float y[2][2];
        y[0][0] = s0[0] + s0[1] + s0[2];
        y[0][1] = s0[1] - s0[2] - s0[3];
        y[1][0] = s1[0] + s1[1] + s1[2];
        y[1][1] = s1[1] - s1[2] - s1[3];
        float rv[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        if (res_n) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int oy = oy0 + 2 * trow + a, ox = ox0 + 2 * tcol + b;
                    const bool ok = cvalid && oy < d.OH && ox < d.OW;
                    rv[a][b] = res_n[ok ? ((size_t)oy * d.OW + ox) * d.res_cstride + co : 0];
                }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int oy = oy0 + 2 * trow + a, ox = ox0 + 2 * tcol + b;
                if (cvalid && oy < d.OH && ox < d.OW) {
                    float v = y[a][b] * sc + sh;
                    ssum += v;
                    ssq += v * v;
                    const size_t pix = (size_t)oy * d.OW + ox;
                    if (res_n) v += rv[a][b];
                    if (d.act == ADH_ACT_RELU) v = fmaxf(v, 0.f);
                    out_n[pix * d.out_cstride + co] = v;
                }
            }
    }
    if (d.stats) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);   // [2][16][32]
        red[(0 * 16 + (tid >> 5)) * 32 + cl] = ssum;
        red[(1 * 16 + (tid >> 5)) * 32 + cl] = ssq;
        __syncthreads();
        if (tid < 64) {
            const int which = tid >> 5;
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) v += red[(which * 16 + r) * 32 + cl];
            d.stats[((size_t)blockIdx.x * 2 + which) * d.NcP + co0 + cl] = v;
        }
    }
}

extern "C" int adh_conv_wino_supported(const adh_conv_desc* d) {
    if (!d) return 0;
    if (d->KH != 3 || d->KW != 3 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1) return 0;
    if (d->out_oy != 0 || d->out_ox != 0 || d->dy0 != -1 || d->dx0 != -1 || d->dstep_y != 1 || d->dstep_x != 1) return 0;
    if (d->Cin % WN_KC != 0 || d->in_cstride % 4 != 0) return 0;
    if (d->VH != d->OH || d->VW != d->OW || d->IH != d->OH || d->IW != d->OW) return 0;
    return 1;
}

extern "C" int adh_conv_wino_forward(void* stream, const adh_conv_desc* d) {
    if (!adh_conv_wino_supported(d)) return ADH_E_UNSUPPORTED;
    if (!d->in || !d->out || !d->wp || d->NcP % 32 != 0 || d->NcP < d->Cout) return ADH_E_ARG;
    if (d->out_cstride < d->Cout || (d->residual && d->res_cstride < d->Cout)) return ADH_E_ARG;
    if (((uintptr_t)d->in & 15) || ((uintptr_t)d->wp & 15)) return ADH_E_ARG;
    if ((int64_t)d->IH * d->IW * d->in_cstride >= (1ll << 31) || (int64_t)d->OH * d->OW * d->out_cstride >= (1ll << 31))
        return ADH_E_UNSUPPORTED;
    WinoGeom g;
    g.tiles_x = adh_ceil_div(d->OW, 32);
    g.tiles_y = adh_ceil_div(d->OH, 8);
    g.nchunks = d->Cin / WN_KC;
    g.KQtot = d->Cin / 4;
    const int lds = 2 * WN_VBUF * 16 + WN_RAW_PX * (WN_KC / 4) * 16;   // V x2 (reused for the accumulator exchange) + raw tile
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    dim3 grid(g.tiles_x * g.tiles_y * d->N, d->NcP / 32);
    hipLaunchKernelGGL(conv_wino_kernel, grid, dim3(512), lds, (hipStream_t)stream, *d, g);
    return adh_check_launch();
}

// U[f = a*4+b][k/4][n][4] = (G g G^T)[a][b] for every (k, n); g taken through the same adh_wlayout as the direct
// pack (so the flipped / transposed dgrad filters come for free)
__global__ void pack_weights_wino_kernel(const float* __restrict__ src, const adh_wlayout L, int KQ, int NcP,
                                         f32x4* __restrict__ wp) {
    const int64_t total = (int64_t)KQ * NcP;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % NcP);
        const int kq = (int)(idx / NcP);
        f32x4 u[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) u[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (n < L.Nc) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kq * 4 + j;
                if (k >= L.K) continue;
                float gg[3][3];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        gg[a][b] = src[(int64_t)L.tap_off0 + a * L.tap_off_sy + b * L.tap_off_sx + (int64_t)k * L.stride_k +
                                       (int64_t)n * L.stride_n];
                float tt[4][3];   // G g
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    tt[0][b] = gg[0][b];
                    tt[1][b] = 0.5f * (gg[0][b] + gg[1][b] + gg[2][b]);
                    tt[2][b] = 0.5f * (gg[0][b] - gg[1][b] + gg[2][b]);
                    tt[3][b] = gg[2][b];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {   // (G g) G^T
                    u[a * 4 + 0][j] = tt[a][0];
                    u[a * 4 + 1][j] = 0.5f * (tt[a][0] + tt[a][1] + tt[a][2]);
                    u[a * 4 + 2][j] = 0.5f * (tt[a][0] - tt[a][1] + tt[a][2]);
                    u[a * 4 + 3][j] = tt[a][2];
                }
            }
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) wp[((int64_t)f * KQ + kq) * NcP + n] = u[f];
    }
}

extern "C" int adh_pack_weights_wino(void* stream, const float* src, const adh_wlayout* L, float* wp) {
    if (!src || !L || !wp || L->K < 1 || L->Nc < 1 || L->KHt != 3 || L->KWt != 3) return ADH_E_ARG;
    const int KQ = adh_round_up(L->K, 8) / 4;
    const int NcP = adh_round_up(L->Nc, 32);
    const int64_t total = (int64_t)KQ * NcP;
    hipLaunchKernelGGL(pack_weights_wino_kernel, dim3(adh_min_i(adh_ceil_div(total, 128), 4096)), dim3(128), 0,
                       (hipStream_t)stream, src, *L, KQ, NcP, reinterpret_cast<f32x4*>(wp));
    return adh_check_launch();
}
