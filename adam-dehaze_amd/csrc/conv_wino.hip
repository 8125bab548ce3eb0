// Winograd F(2x2, 3x3) convolution on fp32 MFMA, gfx950: 3x3 stride-1 pad-1 convolutions (and their data
// gradients, which are 3x3 stride-1 convolutions with flipped / transposed weights) at 2.25x fewer MFMA
// FLOPs than the direct gather kernel.  Same descriptor, same fused epilogue (scale/shift, residual, ReLU,
// BatchNorm partial statistics) as conv_igemm.hip; replaces the same ATen conv2d calls
// (/root/reference models/dehazing/base_model.py:11-13,26-41).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// Design rules (measured, tools/micro/): on gfx950 a VALU instruction takes its issue cycles away from the fp32
// MFMA pipe of the same SIMD -- they do not overlap -- while LDS reads, LDS-DMA and a few global loads per MFMA
// are free.  So there are no producer waves: four waves (one per SIMD, 512 registers each) all run the same
// stream, the MFMA loops carry no vector ALU work, and every byte is staged by instructions that need no VALU.
//
// Workgroup = 256 threads: output region 8 rows x 32 cols = 4 x 16 Winograd tiles (two 32-tile MFMA row
// blocks) x 32*NT output channels; wave w owns the four frequencies (w, 0..3) of all of them: 8*NT 32x32
// accumulator tiles, the first 16 pinned to AGPRs and the rest to VGPRs.
// Per chunk of 16 input channels:
//   * raw halo 10 x 34 pixels x 16 ch -> LDS by LDS-DMA, laid out [row][column parity][16 px][16 ch] (+ the two
//     right-most columns in a per-row tail) so that the 4x4 patch reads of the transform are conflict-free;
//   * transform: thread (tile, channel quad) reads its patch (16 ds_read_b128), applies B^T d B in registers and
//     writes the 16 frequency planes V[f][tile][quad slot ^ swizzle(tile)] (conflict-free both for these writes
//     and for the ds_read_b128 MFMA operand reads);
//   * contraction: 8 groups (4 frequencies x two 8-channel halves) of 8*NT MFMAs; per group two ds_read_b128
//     (A) and NT global_load_dwordx4 (B = transformed weights U[f][k/4][n][4], L2 resident), both fetched one
//     group ahead; addresses are immediates / scalar, waits are counted by hand.
// V is double buffered (chunk c+1 is transformed between the two halves of chunk c's contraction); the raw tile
// is single buffered: two barriers per chunk.
// Epilogue: accumulators -> LDS M[freq][tile][co] (one 32-channel tile at a time), all threads apply A^T M A
// and the fused epilogue.
#include "common.h"
#ifndef W_STORE_AUX
#define W_STORE_AUX 0   // cache policy of the epilogue stores (buffer instruction aux bits; 2 = nt)
#endif
#ifndef W3_DBG
#define W3_DBG 0   // dev builds of conv_wino32_kernel: 1 = skip the input transform, 2 = skip the contraction, 4 = skip the epilogue, 8 = no staging inside the loop
#endif
#include <cstdlib>
#include <cstring>
#define B3_NO_SPLIT ((W3_DBG & 16) != 0)
#include "bf16x3.h"

#define W2_KC 16
#define W2_TILES 64
#define W2_VBUF_F (16 * W2_TILES * W2_KC)          // floats per V buffer (64 KB)
#define W2_RAW_PITCH 576                           // floats per raw row: [2][16 px][16 ch] + 64 tail floats
#define W2_RAW_F (10 * W2_RAW_PITCH)
#define W2_LDS_BYTES ((2 * W2_VBUF_F + W2_RAW_F) * 4)

typedef __attribute__((address_space(3))) void* lds_void_ptr;

struct WinoGeom {
    int tiles_x, tiles_y;        // 32-col x 8-row regions
    int nregions;
    int nchunks;
    int KQtot;                   // Cin / 4
    int ncog;                    // output-channel groups of 32*NT
    int dbg;                     // ADH_WINO_DEBUG bit 1 (ablation runs only): skip the epilogue
};

// Register class of accumulator tile (FT = frequency * 2 + tile half, channel tile J) of a wave: 16 tiles fit the 256
// AGPRs, the other 8*NT - 16 (NT = 3: eight) live in VGPRs (see conv_wgrad.hip, wr_mfma) -- those of channel tile 0,
// which the epilogue drains first, so that its output transform does not run next to 128 live accumulator VGPRs.
template <int NT, int FT, int J>
constexpr bool w2_in_agpr() {
    constexpr int nv = 8 * NT - 16;
    if constexpr (nv <= 0) return true;
    else return !(J == 0 && FT >= 8 - nv);
}
template <bool AGPR>
__device__ __forceinline__ void w2_mfma(f32x16& c, float a, float b) {
    if constexpr (AGPR) asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// accumulators of channel tile J -> M[4 wave + F][tile row 8 (R >> 2) + 2 (R & 3) + h + 32 TH][l31] (h, l31 are in `addr`),
// straight from the register class they live in
template <int NT, int J, int FT, int R>
__device__ __forceinline__ void w2_store_m(const f32x16 (&acc)[8 * NT], unsigned addr) {
    if constexpr (FT < 8) {
        constexpr int off = ((FT >> 1) * W2_TILES + 8 * (R >> 2) + ((R & 3) << 1) + 32 * (FT & 1)) * 128;
        if constexpr (w2_in_agpr<NT, FT, J>())
            asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "a"(acc[FT * NT + J][R]), "n"(off) : "memory");
        else
            asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(acc[FT * NT + J][R]), "n"(off) : "memory");
        if constexpr (R + 1 < 16) w2_store_m<NT, J, FT, R + 1>(acc, addr);
        else w2_store_m<NT, J, FT + 1, 0>(acc, addr);
    }
}

// one contraction group: local frequency F, both tile halves, NT output-channel tiles, 4 k-steps
template <int NT, int F, int KK, int TH, int J, int KEND = 4>   // k-steps KK .. KEND-1 (a whole group: 0 .. 3)
__device__ __forceinline__ void w2_group(f32x16 (&acc)[8 * NT], const f32x4 (&a)[2], const f32x4 (&b)[NT]) {
    if constexpr (KK < KEND) {
        w2_mfma<w2_in_agpr<NT, F * 2 + TH, J>()>(acc[(F * 2 + TH) * NT + J], a[TH][KK], b[J][KK]);
        if constexpr (J + 1 < NT) w2_group<NT, F, KK, TH, J + 1, KEND>(acc, a, b);
        else if constexpr (TH == 0) w2_group<NT, F, KK, 1, 0, KEND>(acc, a, b);
        else w2_group<NT, F, KK + 1, 0, 0, KEND>(acc, a, b);
    }
}

// Weight fetch outside hipcc's waitcnt bookkeeping; scalar base + per-lane 32-bit offset + immediate: no address
// VALU.  Why asm: hipcc does not count LDS-DMA pieces in vmcnt, so with plain loads every wait it places after the
// raw-tile DMA also drains the DMA (measured +4 % kernel time); the hand-counted waits below let the 8 pieces stay
// in flight for two groups.  Contract (cdna_hip_programming.md 5.7): the destination registers must reach
// w2_wait_b untouched -- keep the issue -> wait span free of control flow that could make the compiler copy them
// (a run-time branch around the waits corrupted NT = 1, 2 in development; tests/test_gpu_parity.py::
// test_winograd_matches_direct_path covers every NT with interior and edge regions).
template <int NT>
__device__ __forceinline__ void w2_load_b(f32x4 (&b)[NT], unsigned voff, const float* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[0]) : "v"(voff), "s"(sbase) : "memory");
    if constexpr (NT > 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:512" : "=v"(b[1]) : "v"(voff), "s"(sbase) : "memory");
    if constexpr (NT > 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(b[2]) : "v"(voff), "s"(sbase) : "memory");
}
// wait until at most N vector-memory operations issued after `b`'s loads are outstanding; names b so that no
// consumer is scheduled above the wait
template <int N, int NT>
__device__ __forceinline__ void w2_wait_b(f32x4 (&b)[NT]) {
    if constexpr (NT == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(b[0]) : "n"(N) : "memory");
    if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
    if constexpr (NT == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) : "n"(N) : "memory");
}

template <int NT>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const adh_conv_desc d, const WinoGeom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // V[2][16][64][16] | raw[10][576]; V reused as M[16][64][32]
    float* const raw = lds + 2 * W2_VBUF_F;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;

    // XCD-aware decode: the workgroups that read one input region (one per output-channel group) land on one XCD
    const int bid = blockIdx.x;
    const int q = bid >> 3;
    const int cg = q % g.ncog;
    const int region = (q / g.ncog) * 8 + (bid & 7);
    if (region >= g.nregions) return;
    int rr = region;
    const int tx = rr % g.tiles_x;
    rr /= g.tiles_x;
    const int ty = rr % g.tiles_y;
    const int n = rr / g.tiles_y;
    const int oy0 = ty * 8, ox0 = tx * 32;
    const int co0 = cg * 32 * NT;

    // ------------------------------------------------------------------ raw halo staging plan (LDS-DMA)
    // 32 pieces per chunk, 8 per wave (piece j = 4u + wave): j < 20 full pieces (row j/2, column parity j&1, 16 pixels x
    // 16 ch), 20 <= j < 30 the tail of row j-20 (columns 32, 33 in lanes 0-7; lanes 8-15 re-read column 0 into the
    // pad so the instruction always issues), j >= 30 repeat pieces 28, 29: every wave issues exactly 8 per chunk,
    // which is what the hand-counted vmcnt waits below rely on.  Out-of-image cells are fetched with an out-of-range offset: the LDS-DMA
    // unit writes zeros for them.
    const int xcs = d.in_cstride * 4;                                    // pixel pitch in bytes
    const float* xbase = d.in + (int64_t)n * d.IH * d.IW * d.in_cstride - d.in_cstride;   // pixel (0, -1) of image n
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xbase), 0, 0x7fffffff, 0x00020000);
    const int cq_l = lane & 3, px_l = lane >> 2;
    const int vfull = 2 * px_l * xcs + cq_l * 16;                          // + parity * xcs (scalar)
    const int vtail = (lane < 8 ? (32 + (px_l & 1)) : 1) * xcs + cq_l * 16;
    const bool interior = oy0 >= 1 && oy0 + 8 < d.IH && ox0 >= 1 && ox0 + 33 < d.IW;
    int p_dst[8], p_so[8];      // LDS float offset in raw; scalar byte offset without the chunk term
    bool p_tail[8], p_par[8], p_rowok[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        int j = 4 * u + wave;
        if (j >= 30) j -= 2;
        p_tail[u] = j >= 20;
        const int row = p_tail[u] ? j - 20 : j >> 1;
        p_par[u] = !p_tail[u] && (j & 1);
        const int iy = oy0 - 1 + row;
        p_rowok[u] = iy >= 0 && iy < d.IH;
        const int iyc = adh_min_i(adh_max_i(iy, 0), d.IH - 1);
        p_dst[u] = row * W2_RAW_PITCH + (p_tail[u] ? 512 : (p_par[u] ? 256 : 0));
        p_so[u] = (iyc * d.IW + ox0 + (p_par[u] ? 1 : 0)) * xcs;
    }
    // per-lane validity of the columns a full (parity 0 / 1) or tail piece covers
    const int ix_f = ox0 - 1 + 2 * px_l;            // + parity
    const bool ok_f0 = ix_f >= 0 && ix_f < d.IW, ok_f1 = ix_f + 1 < d.IW;
    const bool ok_t = lane < 8 ? (ox0 + 31 + (px_l & 1) < d.IW) : (lane < 16);
    auto stage_raw = [&](int c) {
        const int cb = c * (W2_KC * 4);
        if (interior) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (p_tail[u]) {
                    if (lane < 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)(raw + p_dst[u]), 16, vtail, p_so[u] + cb, 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)(raw + p_dst[u]), 16, vfull, p_so[u] + cb, 0, 0);
                }
            }
        } else {
            // edge regions: a cell outside the image gets an out-of-range offset (bit 31 against num_records = 0x7fffffff), for which
            // the LDS-DMA unit writes zeros (tools/micro/dma_oob.hip): the padding needs no pass over the landed tile.  (Lanes 16 .. 63
            // of a tail piece stay masked: their 16 bytes would land in the next row.)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = p_rowok[u] && (p_tail[u] ? (lane >= 8 || ok_t) : (p_par[u] ? ok_f1 : ok_f0));
                const int vo = ok ? (p_tail[u] ? vtail : vfull) : (int)0x80000000;
                if (!p_tail[u] || lane < 16)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)(raw + p_dst[u]), 16, vo, p_so[u] + cb, 0, 0);
            }
        }
    };
    // ------------------------------------------------------------------ input transform: thread = (tile, channel quad)
    const int tcol = lane >> 2;          // tile row = wave
    const int tile_t = wave * 16 + tcol;
    int colb[4];
    colb[0] = tcol * 16;
    colb[1] = 256 + tcol * 16;
    colb[2] = tcol < 15 ? (tcol + 1) * 16 : 512;
    colb[3] = tcol < 15 ? 256 + (tcol + 1) * 16 : 512 + 16;
    const float* patch = raw + (2 * wave) * W2_RAW_PITCH + cq_l * 4;
    const int vslot_t = tile_t * 16 + ((cq_l ^ ((tile_t >> 1) & 3)) * 4);
    auto transform = [&](int buf) {
        f32x4 t[16];
#pragma unroll
        for (int b = 0; b < 4; ++b) {      // rows:  B^T d
            const float* pc = patch + colb[b];
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(pc + 0 * W2_RAW_PITCH);
            const f32x4 d1 = *reinterpret_cast<const f32x4*>(pc + 1 * W2_RAW_PITCH);
            const f32x4 d2 = *reinterpret_cast<const f32x4*>(pc + 2 * W2_RAW_PITCH);
            const f32x4 d3 = *reinterpret_cast<const f32x4*>(pc + 3 * W2_RAW_PITCH);
            t[0 * 4 + b] = d0 - d2;
            t[1 * 4 + b] = d1 + d2;
            t[2 * 4 + b] = d2 - d1;
            t[3 * 4 + b] = d1 - d3;
        }
        float* Vb = lds + buf * W2_VBUF_F + vslot_t;
#pragma unroll
        for (int a = 0; a < 4; ++a) {      // columns: (B^T d) B
            *reinterpret_cast<f32x4*>(Vb + (a * 4 + 0) * 1024) = t[a * 4 + 0] - t[a * 4 + 2];
            *reinterpret_cast<f32x4*>(Vb + (a * 4 + 1) * 1024) = t[a * 4 + 1] + t[a * 4 + 2];
            *reinterpret_cast<f32x4*>(Vb + (a * 4 + 2) * 1024) = t[a * 4 + 2] - t[a * 4 + 1];
            *reinterpret_cast<f32x4*>(Vb + (a * 4 + 3) * 1024) = t[a * 4 + 1] - t[a * 4 + 3];
        }
    };

    // ------------------------------------------------------------------ contraction operands
    // A: V[buf][F = 4*wave + f][tile = 32*th + l31][slot(2g + h)], slot(q) = q ^ ((tile >> 1) & 3); g flips slot bit 1
    const int sw = (l31 >> 1) & 3;
    const int a_lane = (wave * 4) * 1024 + l31 * 16 + ((h ^ sw) * 4);           // floats, g = 0
    // B: U[F][kq = 4c + 2g + h][n = co0 + 32 j + l31] (float4)
    const unsigned b_voff = (unsigned)((h * d.NcP + l31) * 16);
    const float* const b_wave = d.wp + ((int64_t)(wave * 4) * g.KQtot * d.NcP + co0) * 4;
    const int64_t b_fstride = (int64_t)g.KQtot * d.NcP * 4;                        // floats per frequency
    const int b_kqstride = d.NcP * 4;                                              // floats per channel quad

    f32x16 acc[8 * NT];
#pragma unroll
    for (int t = 0; t < 8 * NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    f32x4 av[2][2], bv[2][NT];
    auto load_a = [&](const float* vb, int f, int gg, f32x4 (&a)[2]) {
        const float* p = vb + f * 1024 + (gg ? (a_lane ^ 8) : a_lane);
        a[0] = *reinterpret_cast<const f32x4*>(p);
        a[1] = *reinterpret_cast<const f32x4*>(p + 512);
    };
    auto b_ptr = [&](int c, int f, int gg) { return b_wave + f * b_fstride + (int64_t)(c * 4 + 2 * gg) * b_kqstride; };

    // ------------------------------------------------------------------ prologue
    w2_load_b<NT>(bv[0], b_voff, b_ptr(0, 0, 0));
    stage_raw(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    transform(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

#pragma unroll 1
    for (int c = 0; c < g.nchunks; ++c) {
        const float* vb = lds + (c & 1) * W2_VBUF_F;
        const bool more = c + 1 < g.nchunks;
        const int cn = more ? c + 1 : c;   // the last chunk re-issues its own (harmless) loads: uniform counts
        load_a(vb, 0, 0, av[0]);
        // ---- groups 0..3: channel half g = 0
        w2_load_b<NT>(bv[1], b_voff, b_ptr(c, 1, 0));
        stage_raw(cn);
        load_a(vb, 1, 0, av[1]);
        w2_wait_b<NT + 8, NT>(bv[0]);
        w2_group<NT, 0, 0, 0, 0>(acc, av[0], bv[0]);

        w2_load_b<NT>(bv[0], b_voff, b_ptr(c, 2, 0));
        load_a(vb, 2, 0, av[0]);
        w2_wait_b<NT + 8, NT>(bv[1]);
        w2_group<NT, 1, 0, 0, 0>(acc, av[1], bv[1]);

        w2_load_b<NT>(bv[1], b_voff, b_ptr(c, 3, 0));
        load_a(vb, 3, 0, av[1]);
        w2_wait_b<NT, NT>(bv[0]);            // also retires this wave's 8 raw pieces (in-order return)
        w2_group<NT, 2, 0, 0, 0>(acc, av[0], bv[0]);

        w2_load_b<NT>(bv[0], b_voff, b_ptr(c, 0, 1));
        load_a(vb, 0, 1, av[0]);
        w2_wait_b<NT, NT>(bv[1]);
        w2_group<NT, 3, 0, 0, 0>(acc, av[1], bv[1]);
        // ---- raw(c+1) complete -> transform into the other V buffer
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // (unconditional: after the last chunk it fills the idle V buffer from the re-staged tile -- a branch here
        // would sit between the asm weight fetch above and its wait, see w2_load_b)
        transform((c + 1) & 1);
        // ---- groups 4..7: channel half g = 1
        w2_load_b<NT>(bv[1], b_voff, b_ptr(c, 1, 1));
        load_a(vb, 1, 1, av[1]);
        w2_wait_b<NT, NT>(bv[0]);
        w2_group<NT, 0, 0, 0, 0>(acc, av[0], bv[0]);

        w2_load_b<NT>(bv[0], b_voff, b_ptr(c, 2, 1));
        load_a(vb, 2, 1, av[0]);
        w2_wait_b<NT, NT>(bv[1]);
        w2_group<NT, 1, 0, 0, 0>(acc, av[1], bv[1]);

        w2_load_b<NT>(bv[1], b_voff, b_ptr(c, 3, 1));
        load_a(vb, 3, 1, av[1]);
        w2_wait_b<NT, NT>(bv[0]);
        w2_group<NT, 2, 0, 0, 0>(acc, av[0], bv[0]);

        w2_load_b<NT>(bv[0], b_voff, b_ptr(cn, 0, 0));
        w2_wait_b<NT, NT>(bv[1]);
        w2_group<NT, 3, 0, 0, 0>(acc, av[1], bv[1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    w2_wait_b<0, NT>(bv[0]);   // the trailing (dummy) weight fetch
    if (g.dbg & 1) {
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 8 * NT; ++t) s += acc[t][0];
        if (s == 123.456f) d.out[tid] = s;
        return;
    }

    // ---------------------------------------------------------------------- output transform + fused epilogue
    float* M = lds;   // [16][64][32]
    const int cl = tid & 31;
    float* out_n = d.out + (size_t)n * d.OH * d.OW * d.out_cstride;
    const float* res_n = d.residual ? d.residual + (size_t)n * d.OH * d.OW * d.res_cstride : nullptr;
    float* red = raw;   // [2][8][32] statistics exchange (the raw tile is idle now)
    const bool full = oy0 + 8 <= d.OH && ox0 + 32 <= d.OW && co0 + 32 * NT <= d.Cout;
    const int ocs = d.out_cstride * 4, rcs = d.res_cstride * 4;
    const __amdgpu_buffer_rsrc_t or_ = __builtin_amdgcn_make_buffer_rsrc(out_n, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr_ =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(res_n ? res_n : d.in), 0, 0x7fffffff, 0x00020000);
    const int ovoff = 2 * (tid >> 5) * ocs + cl * 4, rvoff = 2 * (tid >> 5) * rcs + cl * 4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if (j) __builtin_amdgcn_s_barrier();   // previous tile's M fully consumed
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int th = 0; th < 2; ++th)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = (r & 3) + 8 * (r >> 2) + 4 * h + 32 * th;
                    M[((wave * 4 + f) * W2_TILES + t) * 32 + l31] = acc[(f * 2 + th) * NT + j][r];
                }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int co = co0 + j * 32 + cl;
        const bool cvalid = co < d.Cout;
        const float sc = (d.scale && cvalid) ? d.scale[co] : 1.f;
        const float sh = (d.shift && cvalid) ? d.shift[co] : 0.f;
        float ssum = 0.f, ssq = 0.f;
        if (full) {
            // whole region inside the image, all 32 channels real: no predicates, and every global address is the
            // per-thread constant voffset plus a scalar offset (buffer instructions: no address VALU)
            const int cbytes = (co0 + j * 32) * 4;
            // residual values are fetched one tile ahead of their use
            float rv[2][2][2] = {};
            auto load_res = [&](int p, float (&r)[2][2]) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        r[a][b] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                            rr_, rvoff, ((oy0 + 2 * (p >> 1) + a) * d.OW + ox0 + 16 * (p & 1) + b) * rcs + cbytes, 0));
            };
            if (res_n) load_res(0, rv[0]);
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int t = (tid >> 5) + 8 * p;
                float m[16];
#pragma unroll
                for (int f = 0; f < 16; ++f) m[f] = M[(f * W2_TILES + t) * 32 + cl];
                if (res_n && p + 1 < 8) load_res(p + 1, rv[(p + 1) & 1]);
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
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        float v = y[a][b] * sc + sh;
                        ssum += v;
                        ssq += v * v;
                        if (res_n) v += rv[p & 1][a][b];
                        if (d.act == ADH_ACT_RELU) v = fmaxf(v, 0.f);
                        __builtin_amdgcn_raw_buffer_store_b32(
                            __float_as_uint(v), or_, ovoff, ((oy0 + 2 * (p >> 1) + a) * d.OW + ox0 + 16 * (p & 1) + b) * ocs + cbytes, 0);
                    }
            }
        } else {
#pragma unroll 1
            for (int p = 0; p < 8; ++p) {
                const int t = (tid >> 5) + 8 * p;
                const int trow = t >> 4, tc = t & 15;
                float m[16];
#pragma unroll
                for (int f = 0; f < 16; ++f) m[f] = M[(f * W2_TILES + t) * 32 + cl];
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
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int oy = oy0 + 2 * trow + a, ox = ox0 + 2 * tc + b;
                        if (cvalid && oy < d.OH && ox < d.OW) {
                            float v = y[a][b] * sc + sh;
                            ssum += v;
                            ssq += v * v;
                            const size_t pix = (size_t)oy * d.OW + ox;
                            if (res_n) v += res_n[pix * d.res_cstride + co];
                            if (d.act == ADH_ACT_RELU) v = fmaxf(v, 0.f);
                            out_n[pix * d.out_cstride + co] = v;
                        }
                    }
            }
        }
        if (d.stats) {
            if (j) __builtin_amdgcn_s_barrier();   // previous tile's partial sums fully read
            red[(0 * 8 + (tid >> 5)) * 32 + cl] = ssum;
            red[(1 * 8 + (tid >> 5)) * 32 + cl] = ssq;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid < 64) {
                const int which = tid >> 5;
                float v = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) v += red[(which * 8 + r) * 32 + cl];
                d.stats[((size_t)region * 2 + which) * d.NcP + co0 + j * 32 + cl] = v;
            }
        }
    }
}

extern "C" int adh_conv_wino_supported(const adh_conv_desc* d) {
    if (!d) return 0;
    if (d->KH != 3 || d->KW != 3 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1) return 0;
    if (d->out_oy != 0 || d->out_ox != 0 || d->dy0 != -1 || d->dx0 != -1 || d->dstep_y != 1 || d->dstep_x != 1) return 0;
    if (d->Cin % W2_KC != 0 || d->in_cstride % 4 != 0) return 0;
    if (d->VH != d->OH || d->VW != d->OW || d->IH != d->OH || d->IW != d->OW) return 0;
    return 1;
}

extern "C" int adh_conv_wino_num_blocks(const adh_conv_desc* d) {
    if (!adh_conv_wino_supported(d)) return ADH_E_UNSUPPORTED;
    return adh_ceil_div(d->OW, 32) * adh_ceil_div(d->OH, 8) * d->N;
}

template <int NT>
static int launch_wino(hipStream_t s, const adh_conv_desc* d, WinoGeom g) {
    g.ncog = d->NcP / (32 * NT);
    const int nblocks = ((g.nregions + 7) / 8) * g.ncog * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<NT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    hipLaunchKernelGGL((conv_wino_kernel<NT>), dim3(nblocks), dim3(256), W2_LDS_BYTES, s, *d, g);
    return adh_check_launch();
}

extern "C" int adh_conv_wino_forward(void* stream, const adh_conv_desc* d) {
    if (!adh_conv_wino_supported(d)) return ADH_E_UNSUPPORTED;
    if (!d->in || !d->out || !d->wp || d->NcP % 32 != 0 || d->NcP < d->Cout) return ADH_E_ARG;
    if (d->out_cstride < d->Cout || (d->residual && d->res_cstride < d->Cout)) return ADH_E_ARG;
    if (((uintptr_t)d->in & 15) || ((uintptr_t)d->wp & 15)) return ADH_E_ARG;
    if ((int64_t)(d->IH + 2) * d->IW * d->in_cstride >= (1ll << 29) || (int64_t)d->OH * d->OW * d->out_cstride >= (1ll << 31))
        return ADH_E_UNSUPPORTED;
    WinoGeom g;
    g.tiles_x = adh_ceil_div(d->OW, 32);
    g.tiles_y = adh_ceil_div(d->OH, 8);
    g.nregions = g.tiles_x * g.tiles_y * d->N;
    g.nchunks = d->Cin / W2_KC;
    g.KQtot = d->Cin / 4;
    const char* dbg = getenv("ADH_WINO_DEBUG");
    g.dbg = dbg ? atoi(dbg) : 0;
    const int nt = d->NcP / 32;
    hipStream_t s = (hipStream_t)stream;
    if (nt % 3 == 0) return launch_wino<3>(s, d, g);
    if (nt % 2 == 0) return launch_wino<2>(s, d, g);
    return launch_wino<1>(s, d, g);
}

// =====================================================================================================================
// Winograd F(3x3, 2x2): the 2x2-tap gather forms (each output-parity class of ConvTranspose2d k4 s2, Conv2d k4 s2 as four
// input-parity classes, and their data gradients) at 4/9 of the direct MFMA work.
//
//   Y(3x3) = G^T [ (A g A^T) .* (B^T d B) ] G        d: 4x4 input patch, g: 2x2 filter
//
// is the F(2x2,3x3) algorithm with the roles of filter and output exchanged (same trilinear form): the input transform
// B^T d B is the one above, the weights are packed as U = A g A^T (adh_pack_weights_wino32) and the output transform is
// G^T M G.  Same kernel skeleton as conv_wino_kernel (4 waves, wave = frequency row, AGPR-pinned accumulators, LDS-DMA
// raw staging, asm weight prefetch with counted waits); what differs:
//   * region = 4 x 16 tiles of 3x3 = 12 x 48 virtual pixels, tiles step by 3: the raw halo 13 x 49 pixels is laid out
//     [row][column mod 3][16 px][16 ch] (+ column 48 of every row in a tail), 40 DMA pieces per slab, 10 per wave; edge
//     regions load per-lane coordinates (every instruction issues with all lanes), out-of-image cells with an out-of-range
//     offset for which the LDS-DMA unit writes zeros;
//   * the raw tile is double buffered and V single buffered (the larger halo does not leave room for two V buffers):
//     a slab is contracted, then the next one is transformed -- VALU and MFMA work do not overlap anyway;
//   * the K loop runs over slabs = (16 input channels, input-parity class); a class only changes scalar offsets;
//   * outputs are written with the descriptor's output stride / offset (parity classes of a transposed convolution).
#define W3_RAW_PITCH 768                             // floats per raw row: [3 planes][16 px][16 ch]
#define W3_RAW_ROWS 13
#define W3_RAW_F (W3_RAW_ROWS * W3_RAW_PITCH + 256)  // + tail [16 slots][16 ch]: column 48 of rows 0..12
#define W3_LDS_BYTES ((W2_VBUF_F + 2 * W3_RAW_F + 2 * 8 * 32) * 4)

#ifdef W3_PROF   // dev build (tools/prof_wino43.sh): per-workgroup s_memtime stamps and the CU each workgroup ran on
__device__ unsigned long long w3_prof_buf[16384 * 32];
#define W3_STAMP(i) do { if (tid == 0 && bid < 16384) w3_prof_buf[bid * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
extern "C" int adh_w3_prof_read(void* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(w3_prof_buf), sizeof(w3_prof_buf)) == hipSuccess ? 0 : -1;
}
#else
#define W3_STAMP(i) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------ bf16 x 3 contraction of
// conv_wino32_kernel (round 4, opt-in: adh_conv_wino32_forward_bf16x3; the scheme of conv_wino43.hip / bf16x3.h).  This kernel
// suits it better than the F(4x4,3x3) one: a wave holds TWO 32-tile row blocks per frequency (8 NT accumulator tiles), so every
// weight register feeds twice the MFMAs (0.25 weight loads per MFMA), 8 NT tiles leave 128 registers for the planes in the making,
// and the contraction is 84 % of a slab.  Per slab and wave: frequencies fi = 0 .. 3, groups G = fi NT + j of 12 MFMAs (both
// row blocks x six plane pairs) on 3 weight registers requested two groups ahead; the planes of frequency fi + 1 (two sets of
// eight values) are split in the gaps of frequency fi.  The staging pieces of the next slab (10 per wave) ride behind the first
// MFMA of the groups from 1 on.
template <int NT>
constexpr int w3b_pieces(int G) {   // pieces issued in group G
    if (W3_DBG & 8) return 0;
    if (G < 1) return 0;
    constexpr int ng = 4 * NT - 1;                       // groups 1 .. 4 NT - 1 may carry pieces
    constexpr int per = (10 + ng - 2) / (ng - 1);        // leave the last group free
    const int first = (G - 1) * per;
    return first >= 10 ? 0 : (10 - first < per ? 10 - first : per);
}
template <int NT>
constexpr int w3b_piece0(int G) { return (G - 1) * ((10 + 4 * NT - 3) / (4 * NT - 2)); }
template <int NM, int M, int S = 0>
constexpr int w3b_first_step() {   // first of the 40 steps (two sets of 20) whose gap is >= M; steps run from gap 3 to gap NM - 1
    if constexpr (S >= 40) return 40;
    else if constexpr (3 + S * (NM - 3) / 40 >= M) return S;
    else return w3b_first_step<NM, M, S + 1>();
}
template <int S0, int S1>
__device__ __forceinline__ void w3b_steps(W4BNext (&n)[2], float m1) {   // step s < 20: set 0, else set 1
    if constexpr (S0 < S1) {
        w4b_step<S0 % 20>(n[S0 / 20], m1);
        w3b_steps<S0 + 1, S1>(n, m1);
    }
}
__device__ __forceinline__ void w3b_load_s(W4BNext (&n)[2], const float* p0, const float* p1) {
    // p0 / p1: the lane's channel quads h and 2 + h of tile l31; + 512 floats = tile l31 + 32 (the second row block)
#pragma unroll
    for (int th = 0; th < 2; ++th) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p0 + th * 512), b = *reinterpret_cast<const f32x4*>(p1 + th * 512);
        n[th].s[0] = a[0]; n[th].s[1] = a[1]; n[th].s[2] = a[2]; n[th].s[3] = a[3];
        n[th].s[4] = b[0]; n[th].s[5] = b[1]; n[th].s[6] = b[2]; n[th].s[7] = b[3];
    }
}
template <int NT, int FI, int M>
__device__ __forceinline__ void w3b_gap(W4BNext (&n)[2], const float* v0, const float* v1, float m1) {
    constexpr int NM = 12 * NT;
    if constexpr (FI + 1 < 4) {
        w3b_steps<w3b_first_step<NM, M>(), w3b_first_step<NM, M + 1>()>(n, m1);
        if constexpr (M == NM - 1 && FI + 2 < 4) w3b_load_s(n, v0 + (FI + 2) * 1024, v1 + (FI + 2) * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
}

struct W3Slab { int iy0, ix0, cb, so0; bool interior; };   // wave-uniform geometry of one slab's halo
struct W3Stage {            // what issuing staging pieces needs, as plain data (a callable cannot be passed into a function template
    __amdgpu_buffer_rsrc_t xr;   // from a __global__ template: hipcc's host pass then drops the kernel's stub)
    float* rawbase;
    int vfull, vtail, wave, px_l, cq_l, xps, IH, IW, in_cstride, xcs, xrs;
};
// pieces u0 .. u1 - 1 of the ten this wave stages per slab into raw buffer `buf` (the body of conv_wino32_kernel's stage_raw)
__device__ __forceinline__ void w3b_stage(const W3Stage& st, const W3Slab& sg, int buf, int u0, int u1) {
    float* raw = st.rawbase + buf * W3_RAW_F;
    int wv = st.wave;
    asm volatile("" : "+s"(wv));
    if (sg.interior) {
        for (int u = u0; u < u1; ++u) {
            const int j = 4 * u + wv;
            if (j < 39) {
                const int r = (j * 171) >> 9, p = j - r * 3;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(st.xr, (lds_void_ptr)(raw + r * W3_RAW_PITCH + p * 256), 16, st.vfull,
                                                         __builtin_amdgcn_readfirstlane(sg.so0 + r * st.xrs + p * st.xcs), 0, 0);
            } else {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(st.xr, (lds_void_ptr)(raw + W3_RAW_ROWS * W3_RAW_PITCH), 16, st.vtail,
                                                         __builtin_amdgcn_readfirstlane(sg.so0), 0, 0);
            }
        }
    } else {
        for (int u = u0; u < u1; ++u) {
            const int j = 4 * u + wv;
            const bool tail = j >= 39;
            const int rs = (j * 171) >> 9, p = tail ? 0 : j - rs * 3;
            const int r = tail ? (st.px_l < W3_RAW_ROWS ? st.px_l : 0) : rs;
            const int col = tail ? 48 : 3 * st.px_l + p;
            // a cell outside the image: bit 31 of the offset = out of range against num_records = 0x7fffffff whatever the scalar
            // offset, and the LDS-DMA unit writes zeros for such a lane (tools/micro/dma_oob.hip): the padding needs no pass over
            // the landed halo
            const int iy = sg.iy0 + r * st.xps, ix = sg.ix0 + col * st.xps;
            const bool ok = iy >= 0 && iy < st.IH && ix >= 0 && ix < st.IW;
            const int vo = ok ? (iy * st.IW + ix) * st.in_cstride * 4 + st.cq_l * 16 : (int)0x80000000;
            const int doff = __builtin_amdgcn_readfirstlane(tail ? W3_RAW_ROWS * W3_RAW_PITCH : rs * W3_RAW_PITCH + p * 256);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(st.xr, (lds_void_ptr)(raw + doff), 16, vo, __builtin_amdgcn_readfirstlane(sg.cb), 0, 0);
        }
    }
}
// groups G .. 4 NT - 1 of one slab (see above); a[th][plane] = the planes of the current frequency for the two row blocks
template <int NT, int G>
__device__ __forceinline__ void w3b_groups(f32x16 (&acc)[8 * NT], u32x4 (&a)[2][3], W4BNext (&n)[2], u32x4 (&bv)[3][3], const float* v0,
                                           const float* v1, unsigned b_voff, const char* b_slab, float m1, const W3Stage& st,
                                           const W3Slab& sgn, int nbuf) {
    if constexpr (G < 4 * NT) {
        constexpr int FI = G / NT, J = G % NT, G2 = G + 2;
        if constexpr (G2 < 4 * NT) w4b_load_b(bv[G2 % 3], b_voff, b_slab + G2 * 3072);
        constexpr int newer = 3 * (4 * NT - 1 - G < 2 ? 4 * NT - 1 - G : 2) + w3b_pieces<NT>(G - 2) + w3b_pieces<NT>(G - 1);
        u32x4(&b)[3] = bv[G % 3];
        w4b_wait_b<newer>(b);
        constexpr bool ag0 = w2_in_agpr<NT, FI * 2 + 0, J>(), ag1 = w2_in_agpr<NT, FI * 2 + 1, J>();
        f32x16& c0 = acc[(FI * 2 + 0) * NT + J];
        f32x16& c1 = acc[(FI * 2 + 1) * NT + J];
        w4b_mfma<ag0>(c0, a[0][0], b[0]);
        if constexpr (w3b_pieces<NT>(G) > 0) {
            __builtin_amdgcn_sched_barrier(0);
            w3b_stage(st, sgn, nbuf, w3b_piece0<NT>(G), w3b_piece0<NT>(G) + w3b_pieces<NT>(G));
        }
        w3b_gap<NT, FI, 12 * J + 1>(n, v0, v1, m1);
        w4b_mfma<ag0>(c0, a[0][0], b[1]);
        w3b_gap<NT, FI, 12 * J + 2>(n, v0, v1, m1);
        w4b_mfma<ag0>(c0, a[0][0], b[2]);
        w3b_gap<NT, FI, 12 * J + 3>(n, v0, v1, m1);
        w4b_mfma<ag0>(c0, a[0][1], b[0]);
        w3b_gap<NT, FI, 12 * J + 4>(n, v0, v1, m1);
        w4b_mfma<ag0>(c0, a[0][1], b[1]);
        w3b_gap<NT, FI, 12 * J + 5>(n, v0, v1, m1);
        w4b_mfma<ag0>(c0, a[0][2], b[0]);
        w3b_gap<NT, FI, 12 * J + 6>(n, v0, v1, m1);
        w4b_mfma<ag1>(c1, a[1][0], b[0]);
        w3b_gap<NT, FI, 12 * J + 7>(n, v0, v1, m1);
        w4b_mfma<ag1>(c1, a[1][0], b[1]);
        w3b_gap<NT, FI, 12 * J + 8>(n, v0, v1, m1);
        w4b_mfma<ag1>(c1, a[1][0], b[2]);
        w3b_gap<NT, FI, 12 * J + 9>(n, v0, v1, m1);
        w4b_mfma<ag1>(c1, a[1][1], b[0]);
        w3b_gap<NT, FI, 12 * J + 10>(n, v0, v1, m1);
        w4b_mfma<ag1>(c1, a[1][1], b[1]);
        w3b_gap<NT, FI, 12 * J + 11>(n, v0, v1, m1);
        w4b_mfma<ag1>(c1, a[1][2], b[0]);
        if constexpr (J == NT - 1 && FI + 1 < 4) {
            // hand-over to frequency FI + 1 (planes written by VALU instructions at least one MFMA ago; the s_nop covers a register
            // copy landing here: hipcc's hazard recogniser does not look into the MFMA statements)
#pragma unroll
            for (int th = 0; th < 2; ++th) {
                a[th][0] = u32x4{n[th].h[0], n[th].h[1], n[th].h[2], n[th].h[3]};
                a[th][1] = u32x4{n[th].m[0], n[th].m[1], n[th].m[2], n[th].m[3]};
                a[th][2] = u32x4{n[th].l[0], n[th].l[1], n[th].l[2], n[th].l[3]};
            }
            asm volatile("s_nop 1" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]));
        } else {
            w3b_gap<NT, FI, 12 * J + 12>(n, v0, v1, m1);
        }
        w3b_groups<NT, G + 1>(acc, a, n, bv, v0, v1, b_voff, b_slab, m1, st, sgn, nbuf);
    }
}

struct Wino32Geom {
    int tiles_x, tiles_y;        // 48-col x 12-row regions of virtual pixels
    int nregions;
    int nchunks;                 // Cin / 16
    int KQtot;                   // Cin / 4
    int ncog;
    int ncls, xps;               // input-parity classes (1 or 4), input pixels per virtual pixel
    int ymin[4], xmin[4];        // input pixel of halo (0,0) for virtual pixel (0,0), per class
    int wcls;                    // floats between the packed weights of two classes
    // merged launch (adh_conv_wino32_forward_multi): `nmerge` single-class descriptors that differ only in their weights, output
    // parity offset, halo origin (ymin / xmin [m]) and statistics rows run as ONE grid of nmerge * mblocks workgroups, region-major
    int nmerge, mblocks;
    int region0;                 // first region of this launch (a layer may be split into a main and a tail launch, wino32_tail_split)
    int m_out_oy[4], m_out_ox[4];
    const float* m_wp[4];
    float* m_stats[4];
};

template <int NT, bool BF3 = false>
__global__ __launch_bounds__(256, 1) void conv_wino32_kernel(const adh_conv_desc d, const Wino32Geom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // V[16][64][16] | raw[2][13][768]+tail | red
    float* const rawbase = lds + W2_VBUF_F;
    float* const red = rawbase + 2 * W3_RAW_F;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;

    // Eight consecutive workgroups (one per XCD) take eight consecutive regions; the (class, channel group) pairs of a region
    // follow each other on the same XCD, so the classes of a merged launch (descriptor mc; workgroup-uniform) read the region's
    // halo -- the same input pixels for all four parity classes -- close together in time on one L2 (measured against the
    // class-major order: FETCH_SIZE -10 %, ConvTranspose 384 -> 96 forward 5.86 -> 5.73 ms; most of the 4x re-read remains)
    const int bid = (int)blockIdx.x;
    const int gq = bid >> 3, per_region = g.ncog * g.nmerge;
    const int r8 = gq / per_region, rem = gq - r8 * per_region;
    const int mc = rem / g.ncog;                      // 0 for a single-descriptor launch
    const float* const wp_m = g.nmerge > 1 ? g.m_wp[mc] : d.wp;
    float* const stats_m = g.nmerge > 1 ? g.m_stats[mc] : d.stats;
    const int out_oy_m = g.nmerge > 1 ? g.m_out_oy[mc] : d.out_oy, out_ox_m = g.nmerge > 1 ? g.m_out_ox[mc] : d.out_ox;
    W3_STAMP(0);
#ifdef W3_PROF
    if (tid == 0 && bid < 16384) {
        w3_prof_buf[bid * 32 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_ID
        w3_prof_buf[bid * 32 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // XCC_ID
    }
#endif
    const int cg = rem - mc * g.ncog;
    const int region = g.region0 + r8 * 8 + (bid & 7);
    if (region >= g.nregions) return;
    int rr = region;
    const int tx = rr % g.tiles_x;
    rr /= g.tiles_x;
    const int ty = rr % g.tiles_y;
    const int n = rr / g.tiles_y;
    const int vy0 = ty * 12, vx0 = tx * 48;
    const int co0 = cg * 32 * NT;

    // ------------------------------------------------------------------ raw halo staging (LDS-DMA), 10 pieces per wave
    const int xcs = d.in_cstride * 4 * g.xps;                  // halo pixel pitch in bytes
    const int xrs = d.IW * d.in_cstride * 4 * g.xps;           // halo row pitch in bytes
    const float* const in_n = d.in + (int64_t)n * d.IH * d.IW * d.in_cstride;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in_n), 0, 0x7fffffff, 0x00020000);
    const int cq_l = lane & 3, px_l = lane >> 2;
    // all classes' halo columns are inside the image: no clamping / fixing needed (rows are checked per piece)
    // pieces u0 .. u1-1 of the 10 this wave stages per slab (the whole tile in the prologue, two per contraction group later)
    const int vfull = 3 * px_l * xcs + cq_l * 16;
    const int vtail = (px_l < W3_RAW_ROWS ? px_l : 0) * xrs + 48 * xcs + cq_l * 16;
    typedef W3Slab SlabGeom;   // wave-uniform, computed once per slab
    auto slab_geom = [&](int slab) {
        SlabGeom sg;
        const int chunk = slab / g.ncls, c = slab - chunk * g.ncls;
        sg.iy0 = vy0 * g.xps + g.ymin[c + mc];
        sg.ix0 = vx0 * g.xps + g.xmin[c + mc];
        sg.interior = sg.iy0 >= 0 && sg.iy0 + 12 * g.xps < d.IH && sg.ix0 >= 0 && sg.ix0 + 48 * g.xps < d.IW;
        sg.cb = chunk * 64;
        sg.so0 = (sg.iy0 * d.IW + sg.ix0) * d.in_cstride * 4 + sg.cb;
        return sg;
    };
    auto stage_raw = [&](const SlabGeom& sg, int buf, int u0, int u1) {
        const int iy0 = sg.iy0, ix0 = sg.ix0, cb = sg.cb, so0 = sg.so0;
        float* raw = rawbase + buf * W3_RAW_F;
        int wv = wave;
        asm volatile("" : "+s"(wv));   // the piece geometry is a few SALU per piece; hoisted out of the slab loop it lands in spilled SGPRs, and every reload is a v_readlane in the MFMA stream
        if (sg.interior) {
#pragma unroll
            for (int u = u0; u < u1; ++u) {
                const int j = 4 * u + wv;
                if (j < 39) {
                    const int r = (j * 171) >> 9, p = j - r * 3;       // j / 3
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)(raw + r * W3_RAW_PITCH + p * 256), 16, vfull,
                                                             __builtin_amdgcn_readfirstlane(so0 + r * xrs + p * xcs), 0, 0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)(raw + W3_RAW_ROWS * W3_RAW_PITCH), 16, vtail,
                                                             __builtin_amdgcn_readfirstlane(so0), 0, 0);
                }
            }
        } else {
            // per-lane coordinates: every instruction issues with all lanes; a cell outside the image gets an out-of-range
            // offset (bit 31), for which the LDS-DMA unit writes zeros
#pragma unroll
            for (int u = u0; u < u1; ++u) {
                const int j = 4 * u + wv;
                const bool tail = j >= 39;
                const int rs = (j * 171) >> 9, p = tail ? 0 : j - rs * 3;      // full pieces: row, plane (wave-uniform)
                const int r = tail ? (px_l < W3_RAW_ROWS ? px_l : 0) : rs;
                const int col = tail ? 48 : 3 * px_l + p;
                const int iy = iy0 + r * g.xps, ix = ix0 + col * g.xps;
                const bool ok = iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW;
                const int vo = ok ? (iy * d.IW + ix) * d.in_cstride * 4 + cq_l * 16 : (int)0x80000000;
                const int doff = __builtin_amdgcn_readfirstlane(tail ? W3_RAW_ROWS * W3_RAW_PITCH : rs * W3_RAW_PITCH + p * 256);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr)(raw + doff), 16, vo, __builtin_amdgcn_readfirstlane(cb), 0, 0);
            }
        }
    };
    // ------------------------------------------------------------------ input transform: thread = (tile, channel quad)
    const int tcol = lane >> 2;          // tile row = wave
    const int tile_t = wave * 16 + tcol;
    int colb[3], colb3[4];               // float offsets of patch columns 0..2 (row pitch immediate) and of column 3 per row
#pragma unroll
    for (int b = 0; b < 3; ++b) colb[b] = (3 * wave) * W3_RAW_PITCH + b * 256 + tcol * 16 + cq_l * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        colb3[i] = (tcol < 15 ? (3 * wave + i) * W3_RAW_PITCH + (tcol + 1) * 16 : W3_RAW_ROWS * W3_RAW_PITCH + (3 * wave + i) * 16) + cq_l * 4;
    const int vslot_t = tile_t * 16 + ((cq_l ^ ((tile_t >> 1) & 3)) * 4);
    const float m1 = adh_opaque(-1.f);
    auto transform = [&](int buf) {
        // frequency row by frequency row (row a of B^T d needs two patch rows): 8 patch reads per row instead of 16 in
        // total, but only ~50 live registers -- this kernel has none to spare next to 8*NT accumulator tiles
        const float* raw = rawbase + buf * W3_RAW_F;
        float* Vb = lds + vslot_t;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int rA = a == 0 ? 0 : (a == 2 ? 2 : 1), rB = a == 0 ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 3));
            f32x4 t[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f32x4 dA = *reinterpret_cast<const f32x4*>(b < 3 ? raw + colb[b] + rA * W3_RAW_PITCH : raw + colb3[rA]);
                const f32x4 dB = *reinterpret_cast<const f32x4*>(b < 3 ? raw + colb[b] + rB * W3_RAW_PITCH : raw + colb3[rB]);
                t[b] = a == 1 ? dA + dB : adh_pksub(dA, dB, m1);
            }
            *reinterpret_cast<f32x4*>(Vb + (a * 4 + 0) * 1024) = adh_pksub(t[0], t[2], m1);
            *reinterpret_cast<f32x4*>(Vb + (a * 4 + 1) * 1024) = t[1] + t[2];
            *reinterpret_cast<f32x4*>(Vb + (a * 4 + 2) * 1024) = adh_pksub(t[2], t[1], m1);
            *reinterpret_cast<f32x4*>(Vb + (a * 4 + 3) * 1024) = adh_pksub(t[1], t[3], m1);
        }
    };

    // ------------------------------------------------------------------ contraction operands (as conv_wino_kernel)
    const int sw = (l31 >> 1) & 3;
    const int a_lane = (wave * 4) * 1024 + l31 * 16 + ((h ^ sw) * 4);
    const unsigned b_voff = (unsigned)((h * d.NcP + l31) * 16);
    const float* const b_wave = wp_m + ((int64_t)(wave * 4) * g.KQtot * d.NcP + co0) * 4;
    const int64_t b_fstride = (int64_t)g.KQtot * d.NcP * 4;
    const int b_kqstride = d.NcP * 4;

    f32x16 acc[8 * NT];
#pragma unroll
    for (int t = 0; t < 8 * NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // bf16 x 3 form: A = the same reads, split in registers; B = adh_pack_weights_wino32_bf16x3's
    // [class][channel group][chunk][16 f][NT][3 planes][64 lanes][16 B]: the 4 NT groups of a wave and slab are 12 NT KB contiguous
    const float* const v0 = lds + a_lane;
    const float* const v1 = lds + (a_lane ^ 8);
    const unsigned b3_voff = (unsigned)lane * 16u;
    const size_t b3_chunk = (size_t)16 * NT * 3072, b3_cls = (size_t)g.ncog * g.nchunks * b3_chunk;
    const char* const b3_wave = reinterpret_cast<const char*>(wp_m) + ((size_t)cg * g.nchunks * 16 + (size_t)wave * 4) * (NT * 3072);
    auto b3_ptr = [&](int slab) {
        const int chunk = slab / g.ncls, c = slab - chunk * g.ncls;
        return b3_wave + (size_t)c * b3_cls + (size_t)chunk * b3_chunk;
    };
    W3Stage st3;
    st3.xr = xr; st3.rawbase = rawbase; st3.vfull = vfull; st3.vtail = vtail; st3.wave = wave; st3.px_l = px_l; st3.cq_l = cq_l;
    st3.xps = g.xps; st3.IH = d.IH; st3.IW = d.IW; st3.in_cstride = d.in_cstride; st3.xcs = xcs; st3.xrs = xrs;
    u32x4 av3[2][3], bv3[3][3];
    W4BNext nx[2];
    const float m1s = adh_opaque(-1.f);

    f32x4 av[2][2], bv[2][NT];
    auto load_a = [&](int f, int gg, f32x4 (&a)[2]) {
        const float* p = lds + f * 1024 + (gg ? (a_lane ^ 8) : a_lane);
        a[0] = *reinterpret_cast<const f32x4*>(p);
        a[1] = *reinterpret_cast<const f32x4*>(p + 512);
    };
    auto b_ptr = [&](int slab, int f, int gg) {
        const int chunk = slab / g.ncls, c = slab - chunk * g.ncls;
        return b_wave + (int64_t)c * g.wcls + f * b_fstride + (int64_t)(chunk * 4 + 2 * gg) * b_kqstride;
    };

    // ------------------------------------------------------------------ prologue
    const int nslabs = g.nchunks * g.ncls;
    if constexpr (!BF3) w2_load_b<NT>(bv[0], b_voff, b_ptr(0, 0, 0));
    stage_raw(slab_geom(0), 0, 0, 10);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!(W3_DBG & 1)) transform(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    W3_STAMP(1);
#pragma unroll 1
    for (int s = 0; s < nslabs; ++s) {
        if (s == 1) W3_STAMP(8);
        // raw(s+1) is staged into buffer (s+1)&1 (whose last reader was the transform of slab s-1) during this slab's
        // contraction: two pieces behind the first k-step of groups 1..5 -- issued back to back the ten pieces block the
        // wave's instruction stream for ~1700 cycles (tools/prof_wino43.py), spread out they hide behind the MFMAs.  vmcnt
        // returns in order: the weight wait of group i allows the next weight set and the pieces of group i-1 to be
        // outstanding, and the wait of group 7 retires the last pieces.
        const int sn = s + 1 < nslabs ? s + 1 : s;     // the last slab re-stages itself: uniform counts
        constexpr int P = (W3_DBG & 8) ? 0 : 2;
        const SlabGeom sgn = slab_geom(sn);
        if constexpr (BF3) {
            if (!(W3_DBG & 2)) {
                // the weights of groups 0 and 1 are requested HERE, not behind the previous slab's transform: a register an asm load is
                // still filling must not be carried over the loop's back edge (hipcc resolves the carry with register copies that run
                // before the load has landed: conv_wino43.hip, w4b_groups); the planes of the wave's first frequency are split while
                // they are on their way; then the values of the second frequency are requested
                w4b_load_b(bv3[0], b3_voff, b3_ptr(s));
                w4b_load_b(bv3[1], b3_voff, b3_ptr(s) + 3072);
                w3b_load_s(nx, v0, v1);
                w3b_steps<0, 40>(nx, m1s);
#pragma unroll
                for (int th = 0; th < 2; ++th) {
                    av3[th][0] = u32x4{nx[th].h[0], nx[th].h[1], nx[th].h[2], nx[th].h[3]};
                    av3[th][1] = u32x4{nx[th].m[0], nx[th].m[1], nx[th].m[2], nx[th].m[3]};
                    av3[th][2] = u32x4{nx[th].l[0], nx[th].l[1], nx[th].l[2], nx[th].l[3]};
                }
                asm volatile("s_nop 1" : "+v"(av3[0][0]), "+v"(av3[0][1]), "+v"(av3[0][2]), "+v"(av3[1][0]), "+v"(av3[1][1]), "+v"(av3[1][2]));
                w3b_load_s(nx, v0 + 1024, v1 + 1024);
                __builtin_amdgcn_sched_barrier(0);
                w3b_groups<NT, 0>(acc, av3, nx, bv3, v0, v1, b3_voff, b3_ptr(s), m1s, st3, sgn, (s + 1) & 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last pieces (issued at least two groups ago) have landed
            }
        } else if (!(W3_DBG & 2)) {
        load_a(0, 0, av[0]);
        w2_load_b<NT>(bv[1], b_voff, b_ptr(s, 1, 0));
        load_a(1, 0, av[1]);
        w2_wait_b<NT, NT>(bv[0]);
        w2_group<NT, 0, 0, 0, 0>(acc, av[0], bv[0]);

#define W3_GROUP_WITH_PIECES(F_, CUR, U0)                                             \
        w2_group<NT, F_, 0, 0, 0, 1>(acc, av[CUR], bv[CUR]);                           \
        __builtin_amdgcn_sched_barrier(0);                                            \
        if (P) stage_raw(sgn, (s + 1) & 1, U0, U0 + 2);                               \
        __builtin_amdgcn_sched_barrier(0);                                            \
        w2_group<NT, F_, 1, 0, 0>(acc, av[CUR], bv[CUR]);

        w2_load_b<NT>(bv[0], b_voff, b_ptr(s, 2, 0));
        load_a(2, 0, av[0]);
        w2_wait_b<NT, NT>(bv[1]);
        W3_GROUP_WITH_PIECES(1, 1, 0)

        w2_load_b<NT>(bv[1], b_voff, b_ptr(s, 3, 0));
        load_a(3, 0, av[1]);
        w2_wait_b<NT + P, NT>(bv[0]);
        W3_GROUP_WITH_PIECES(2, 0, 2)

        w2_load_b<NT>(bv[0], b_voff, b_ptr(s, 0, 1));
        load_a(0, 1, av[0]);
        w2_wait_b<NT + P, NT>(bv[1]);
        W3_GROUP_WITH_PIECES(3, 1, 4)

        w2_load_b<NT>(bv[1], b_voff, b_ptr(s, 1, 1));
        load_a(1, 1, av[1]);
        w2_wait_b<NT + P, NT>(bv[0]);
        W3_GROUP_WITH_PIECES(0, 0, 6)

        w2_load_b<NT>(bv[0], b_voff, b_ptr(s, 2, 1));
        load_a(2, 1, av[0]);
        w2_wait_b<NT + P, NT>(bv[1]);
        W3_GROUP_WITH_PIECES(1, 1, 8)
#undef W3_GROUP_WITH_PIECES

        w2_load_b<NT>(bv[1], b_voff, b_ptr(s, 3, 1));
        load_a(3, 1, av[1]);
        w2_wait_b<NT + P, NT>(bv[0]);
        w2_group<NT, 2, 0, 0, 0>(acc, av[0], bv[0]);

        w2_load_b<NT>(bv[0], b_voff, b_ptr(sn, 0, 0));
        w2_wait_b<NT, NT>(bv[1]);               // everything older than bv[0], the pieces included, has landed
        w2_group<NT, 3, 0, 0, 0>(acc, av[1], bv[1]);
        }
        // ---- V is free once every wave is here; raw(s+1) landed during this slab (last wait above)
        if (s == 1) W3_STAMP(9);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s == 1) W3_STAMP(10);
        if (!(W3_DBG & 1)) transform((s + 1) & 1);              // (after the last slab: a harmless re-transform, keeps the span branch-free)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (s == 1) W3_STAMP(11);
        __builtin_amdgcn_s_barrier();
        if (s == 1) W3_STAMP(12);
    }
    if constexpr (!BF3) w2_wait_b<0, NT>(bv[0]);
    W3_STAMP(2);

    if (W3_DBG & 4) return;
    // ---------------------------------------------------------------------- output transform G^T M G + fused epilogue
    // One 32-channel tile at a time: accumulators -> M[16][64 tile rows][32 co] in LDS (tile t sits in row t' = t with its
    // low three bits rotated, so the two lane halves -- tiles t and t + 4 -- hit different banks), then thread = (tile,
    // channel quad), two tiles each: float4 reads of the 4x4 frequency patch, G^T m G, fused epilogue, 16-byte stores.
    float* M = lds;   // spans V and the raw buffers
    const int eq = tid & 7;
    // One code path for full and ragged regions, without per-pixel address arithmetic or exec-mask branches (as in
    // conv_wino43.hip): raw buffer stores / residual loads whose address is a per-thread byte offset plus a
    // workgroup-uniform scalar offset per pixel; bit 31 of the vector offset marks a pixel outside the virtual grid or a
    // channel quad beyond Cout -- out of range against num_records = 0x7fffffff whatever the scalar offset, so the store is
    // dropped and the load returns zeros.
    float* out_n = d.out + (size_t)n * d.OH * d.OW * d.out_cstride;
    const float* res_n = d.residual ? d.residual + (size_t)n * d.OH * d.OW * d.res_cstride : nullptr;
    const int o_px = d.out_cstride * 4 * d.out_sx, o_row = d.OW * d.out_cstride * 4 * d.out_sy;   // byte pitches per virtual pixel
    const int r_px = d.res_cstride * 4 * d.out_sx, r_row = d.OW * d.res_cstride * 4 * d.out_sy;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out_n, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(res_n ? res_n : out_n), 0, 0x7fffffff, 0x00020000);
    const bool ragged = vy0 + 12 > d.VH || vx0 + 48 > d.VW;     // workgroup-uniform
    const float act_lo = d.act == ADH_ACT_RELU ? 0.f : -INFINITY;   // ReLU as max(v, 0), identity as max(v, -inf)
    const unsigned m_wbase = (unsigned)(((wave * 4) * W2_TILES + h) * 32 + l31) * 4u;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if (j == 1) W3_STAMP(16);
        __builtin_amdgcn_s_barrier();   // V / raw (first tile) or the previous tile's M fully consumed
        if (j == 1) W3_STAMP(17);
        if (j == 0) w2_store_m<NT, 0, 0, 0>(acc, m_wbase);
        if (j == 1) w2_store_m<NT, (NT > 1 ? 1 : 0), 0, 0>(acc, m_wbase);
        if (j == 2) w2_store_m<NT, (NT > 2 ? 2 : 0), 0, 0>(acc, m_wbase);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (j == 1) W3_STAMP(18);
        __builtin_amdgcn_s_barrier();
        if (j == 1) W3_STAMP(19);
        const int cq0 = co0 + j * 32 + eq * 4;
        const bool quad_ok = cq0 + 3 < d.Cout;             // Cout % 4 == 0 (wino32_plan): a quad is real or padding
        f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
        if (d.scale && quad_ok) sc4 = *reinterpret_cast<const f32x4*>(d.scale + cq0);
        if (d.shift && quad_ok) sh4 = *reinterpret_cast<const f32x4*>(d.shift + cq0);
        const unsigned chanpen = quad_ok ? 0u : 0x80000000u;
        f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int t = (tid >> 3) + 32 * p;                       // tile: row t >> 4, column t & 15
            const int tp = (t & ~7) | ((t & 3) << 1) | ((t >> 2) & 1);
            const float* mp = M + tp * 32 + eq * 4;
            const int vyb = vy0 + 3 * (t >> 4), vxb = vx0 + 3 * (t & 15);
            unsigned rowpen[3], colpen[3];
            float rowf[3], colf[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                rowpen[r] = vyb + r < d.VH ? 0u : 0x80000000u;
                colpen[r] = vxb + r < d.VW ? 0u : 0x80000000u;
                rowf[r] = vyb + r < d.VH ? 1.f : 0.f;
                colf[r] = vxb + r < d.VW ? 1.f : 0.f;
            }
            const int pix0 = (vyb * d.out_sy + out_oy_m) * d.OW + vxb * d.out_sx + out_ox_m;
            const unsigned o_vj = (unsigned)(pix0 * d.out_cstride * 4 + cq0 * 4) | chanpen;
            const unsigned r_vj = (unsigned)(pix0 * d.res_cstride * 4 + cq0 * 4) | chanpen;
            f32x4 rres[9];
            if (res_n) {   // workgroup-uniform; the asm keeps it a branch
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj)
                        rres[i * 3 + jj] = __builtin_bit_cast(
                            f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, r_vj | rowpen[i] | colpen[jj], i * r_row + jj * r_px, 0));
                asm volatile("" ::: "memory");
            } else {
#pragma unroll
                for (int q2 = 0; q2 < 9; ++q2) rres[q2] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // u[i][b] = sum_a G^T[i][a] m[a][b],  G^T = [[1, 1/2, 1/2, 0], [0, 1/2, -1/2, 0], [0, 1/2, 1/2, 1]]
            f32x4 u[3][4];
#pragma unroll
            for (int b2 = 0; b2 < 4; ++b2) {
                f32x4 m[4];
#pragma unroll
                for (int a2 = 0; a2 < 4; ++a2) m[a2] = *reinterpret_cast<const f32x4*>(mp + (a2 * 4 + b2) * (W2_TILES * 32));
                const f32x4 hs = 0.5f * (m[1] + m[2]), hd = 0.5f * adh_pksub(m[1], m[2], m1);
                u[0][b2] = m[0] + hs;
                u[1][b2] = hd;
                u[2][b2] = hs + m[3];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const f32x4 hs = 0.5f * (u[i][1] + u[i][2]), hd = 0.5f * adh_pksub(u[i][1], u[i][2], m1);
                const f32x4 y[3] = {u[i][0] + hs, hd, hs + u[i][3]};
#pragma unroll
                for (int jj = 0; jj < 3; ++jj) {
                    f32x4 v = y[jj] * sc4 + sh4;
                    if (stats_m) {
                        f32x4 vs = v;
                        if (ragged) {   // pixels outside the virtual grid do not count (a real, workgroup-uniform branch)
                            vs = v * (rowf[i] * colf[jj]);
                            asm volatile("" : "+v"(vs));
                        }
                        ssum += vs;
                        ssq += vs * vs;
                    }
                    v += rres[i * 3 + jj];
                    v = {fmaxf(v[0], act_lo), fmaxf(v[1], act_lo), fmaxf(v[2], act_lo), fmaxf(v[3], act_lo)};
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, o_vj | rowpen[i] | colpen[jj],
                                                           i * o_row + jj * o_px, W_STORE_AUX);
                    // gfx950 128-bit buffer-store data hazard (conv_wino43.hip, W4_STORE_NOPS): the asm reads the data
                    // registers, so they stay untouched until the wait states behind the store have passed
                    asm volatile("s_nop 1" : "+v"(v)::"memory");
                }
            }
        }
        if (j == 1) W3_STAMP(21);
        if (stats_m) {
            // sum over the 8 tiles of this wave (lane bits 3..5), then over the 4 waves through LDS
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) {
                    ssum[e] += __shfl_xor(ssum[e], o, 64);
                    ssq[e] += __shfl_xor(ssq[e], o, 64);
                }
            }
            if (j) __builtin_amdgcn_s_barrier();
            if (lane < 8) {
                *reinterpret_cast<f32x4*>(red + (0 * 4 + wave) * 32 + eq * 4) = ssum;
                *reinterpret_cast<f32x4*>(red + (1 * 4 + wave) * 32 + eq * 4) = ssq;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid < 64) {
                const int which = tid >> 5, cl = tid & 31;
                float v = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) v += red[(which * 4 + r) * 32 + cl];
                stats_m[((size_t)region * 2 + which) * d.NcP + co0 + j * 32 + cl] = v;
            }
        }
    }
    W3_STAMP(3);
}

// eligibility + geometry of the F(3x3,2x2) path: forward-walking 2x2 taps on the in_s-subsampled input, or the 4x4 s2
// convolution as four input-parity classes
static int wino32_plan(const adh_conv_desc* d, Wino32Geom* g) {
    static const bool enabled = !(getenv("ADH_WINO32") && atoi(getenv("ADH_WINO32")) == 0);   // A/B switch
    if (!enabled || !d) return 0;
    if (d->Cin % W2_KC != 0 || d->in_cstride % 4 != 0 || d->NcP % 32 != 0) return 0;
    if (d->in_sy != d->in_sx || d->dstep_y != d->dstep_x || d->out_sy != d->out_sx) return 0;
    if ((int64_t)(d->IH + 2) * d->IW * d->in_cstride >= (1ll << 29)) return 0;
    // the epilogue stores 16-byte channel quads through a buffer descriptor spanning one image (conv_rows.hip takes the rest)
    if (d->Cout % 4 != 0 || d->out_cstride % 4 != 0 || ((uintptr_t)d->out & 15) || (int64_t)d->OH * d->OW * d->out_cstride >= (1ll << 29))
        return 0;
    if (d->residual && (d->res_cstride % 4 != 0 || ((uintptr_t)d->residual & 15) ||
                        (int64_t)d->OH * d->OW * d->res_cstride >= (1ll << 29)))
        return 0;
    if ((d->scale && ((uintptr_t)d->scale & 15)) || (d->shift && ((uintptr_t)d->shift & 15))) return 0;
    if (d->KH == 2 && d->KW == 2 && d->dstep_y == d->in_sy && (d->in_sy == 1 || d->in_sy == 2)) {
        g->ncls = 1; g->xps = d->in_sy;
        for (int c = 0; c < 4; ++c) { g->ymin[c] = d->dy0; g->xmin[c] = d->dx0; }
    } else if (d->KH == 4 && d->KW == 4 && d->in_sy == 2 && d->dstep_y == 1) {
        g->ncls = 4; g->xps = 2;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) { g->ymin[py * 2 + px] = d->dy0 + py; g->xmin[py * 2 + px] = d->dx0 + px; }
    } else {
        return 0;
    }
    g->tiles_x = adh_ceil_div(d->VW, 48);
    g->tiles_y = adh_ceil_div(d->VH, 12);
    g->nregions = g->tiles_x * g->tiles_y * d->N;
    g->nchunks = d->Cin / W2_KC;
    g->KQtot = d->Cin / 4;
    g->wcls = 16 * g->KQtot * d->NcP * 4;
    g->nmerge = 1;
    g->region0 = 0;
    g->mblocks = 0;
    return 1;
}

extern "C" int adh_conv_wino32_supported(const adh_conv_desc* d) {
    Wino32Geom g;
    return wino32_plan(d, &g);
}

extern "C" int adh_conv_wino32_num_blocks(const adh_conv_desc* d) {
    Wino32Geom g;
    if (!wino32_plan(d, &g)) return ADH_E_UNSUPPORTED;
    return g.nregions;
}

// Grid tails.  A launch of B blocks runs ceil(B / CUs) rounds of one workgroup per CU; Conv2d k4 s2 96 -> 192 at the headline size is
// 3,872 blocks = 15.125 rounds: the last round keeps 32 of 256 CUs busy for a full region time (5.5 % of the launch; 8.25 rounds for
// 192 -> 384 and the merged ConvTranspose 384 -> 96, 4.5 for 384 -> 192).  The regions of that last partial round run as a SECOND
// launch with one 32-channel tile per workgroup (NT = 1): three times the blocks at ~0.42 of the time each, i.e. the tail costs
// 0.42 (or 0.84) of a round instead of one.  Returns the number of regions of the main launch (a multiple of 8; == nregions: no split).
// fp32 form only: the bf16 x 3 weights are laid out per NT.
static int wino32_tail_split(int nregions, int group_blocks /* blocks per 8 regions */, int NT) {
    static int ncu = -1;
    if (ncu < 0) {
        int dev = 0;
        hipDeviceProp_t p;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 256;
        const char* e = getenv("ADH_WINO32_TAIL");      // A/B switch: 0 = one launch per layer
        if (e && atoi(e) == 0) ncu = 0;
    }
    if (ncu <= 0 || NT < 2) return nregions;
    const int ngroups = (nregions + 7) / 8, nblocks = ngroups * group_blocks;
    const int full = nblocks / ncu * ncu;                        // blocks of the full rounds
    if (full == 0 || full == nblocks) return nregions;
    const int main_groups = full / group_blocks;                 // (whole groups of 8 regions only)
    const int tail_blocks = nblocks - main_groups * group_blocks;
    const int tail_rounds = (NT * tail_blocks + ncu - 1) / ncu;  // rounds of the NT = 1 tail launch
    const double cost = tail_rounds * (NT == 3 ? 0.42 : 0.56), now = (double)((tail_blocks + ncu - 1) / ncu);
    if (main_groups == 0 || cost > 0.6 * now) return nregions;   // (0.84 of a round for a tail of more than a third: measured slower)
    return main_groups * 8;
}

template <int NT, bool BF3 = false>
static int launch_wino32_part(hipStream_t s, const adh_conv_desc* d, Wino32Geom g, int region0, int region_end) {
    g.ncog = d->NcP / (32 * NT);
    g.region0 = region0;
    g.nregions = region_end;
    const int nblocks = ((region_end - region0 + 7) / 8) * g.ncog * 8;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino32_kernel<NT, BF3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv_wino32_kernel<NT, BF3>), dim3(nblocks), dim3(256), W3_LDS_BYTES, s, *d, g);
    return adh_check_launch();
}
template <int NT, bool BF3 = false>
static int launch_wino32(hipStream_t s, const adh_conv_desc* d, Wino32Geom g) {
    const int nreg = g.nregions;
    const int main_regions = BF3 ? nreg : wino32_tail_split(nreg, 8 * (d->NcP / (32 * NT)), NT);
    const int rc = launch_wino32_part<NT, BF3>(s, d, g, 0, main_regions);
    if (rc || main_regions == nreg) return rc;
    return launch_wino32_part<1, BF3>(s, d, g, main_regions, nreg);
}

static int wino32_forward_impl(void* stream, const adh_conv_desc* d, bool bf3) {
    Wino32Geom g;
    if (!wino32_plan(d, &g)) return ADH_E_UNSUPPORTED;
    if (!d->in || !d->out || !d->wp || d->NcP < d->Cout) return ADH_E_ARG;
    if (d->out_cstride < d->Cout || (d->residual && d->res_cstride < d->Cout)) return ADH_E_ARG;
    if ((d->VH - 1) * d->out_sy + d->out_oy >= d->OH || (d->VW - 1) * d->out_sx + d->out_ox >= d->OW || d->out_oy < 0 ||
        d->out_ox < 0)
        return ADH_E_ARG;
    if (((uintptr_t)d->in & 15) || ((uintptr_t)d->wp & 15)) return ADH_E_ARG;
    const int nt = d->NcP / 32;
    hipStream_t s = (hipStream_t)stream;
    if (bf3) {
        if (nt % 3 == 0) return launch_wino32<3, true>(s, d, g);
        if (nt % 2 == 0) return launch_wino32<2, true>(s, d, g);
        return launch_wino32<1, true>(s, d, g);
    }
    if (nt % 3 == 0) return launch_wino32<3>(s, d, g);
    if (nt % 2 == 0) return launch_wino32<2>(s, d, g);
    return launch_wino32<1>(s, d, g);
}

extern "C" int adh_conv_wino32_forward(void* stream, const adh_conv_desc* d) { return wino32_forward_impl(stream, d, false); }
// The same launch (and the merged form below) with the contraction on v_mfma_f32_32x32x16_bf16 over exact three-plane bf16 splits
// of both operands (opt-in, ADH_CONTRACT=bf16x3; d->wp from adh_pack_weights_wino32_bf16x3): see the comment at w3b_pieces.
extern "C" int adh_conv_wino32_forward_bf16x3(void* stream, const adh_conv_desc* d) { return wino32_forward_impl(stream, d, true); }

// Several single-class launches of ONE layer as one grid: the output-parity classes of a transposed convolution / of the data
// gradient of a k4 s2 convolution (2 x 2-tap forms).  Their grids are not multiples of the CU count (ConvTranspose 384 -> 192:
// 1056 workgroups = 4.125 rounds of one workgroup per CU), so four launches waste up to four partial rounds; one grid wastes one.
// The descriptors may differ in wp, out_oy / out_ox, dy0 / dx0 and stats only (checked); n = 2 .. 4.
static int wino32_forward_multi_impl(void* stream, const adh_conv_desc* descs, int n, bool bf3) {
    if (!descs || n < 1 || n > 4) return ADH_E_ARG;
    if (n == 1) return wino32_forward_impl(stream, descs, bf3);
    Wino32Geom g0;
    if (!wino32_plan(&descs[0], &g0) || g0.ncls != 1) return ADH_E_UNSUPPORTED;
    for (int m = 0; m < n; ++m) {
        const adh_conv_desc* d = &descs[m];
        Wino32Geom g;
        if (!wino32_plan(d, &g) || g.ncls != 1) return ADH_E_UNSUPPORTED;
        if (!d->in || !d->out || !d->wp || d->NcP < d->Cout) return ADH_E_ARG;
        if (d->out_cstride < d->Cout || (d->residual && d->res_cstride < d->Cout)) return ADH_E_ARG;
        if ((d->VH - 1) * d->out_sy + d->out_oy >= d->OH || (d->VW - 1) * d->out_sx + d->out_ox >= d->OW || d->out_oy < 0 ||
            d->out_ox < 0)
            return ADH_E_ARG;
        if (((uintptr_t)d->in & 15) || ((uintptr_t)d->wp & 15)) return ADH_E_ARG;
        adh_conv_desc a = descs[0], b = *d;           // everything but the per-class fields must agree
        a.wp = b.wp = nullptr; a.stats = b.stats = nullptr;
        a.out_oy = b.out_oy = 0; a.out_ox = b.out_ox = 0; a.dy0 = b.dy0 = 0; a.dx0 = b.dx0 = 0;
        if (memcmp(&a, &b, sizeof(a)) != 0 || (!descs[0].stats) != (!d->stats)) return ADH_E_UNSUPPORTED;
        g0.ymin[m] = g.ymin[0]; g0.xmin[m] = g.xmin[0];
        g0.m_out_oy[m] = d->out_oy; g0.m_out_ox[m] = d->out_ox;
        g0.m_wp[m] = d->wp; g0.m_stats[m] = d->stats;
    }
    for (int m = n; m < 4; ++m) {
        g0.ymin[m] = g0.ymin[0]; g0.xmin[m] = g0.xmin[0];
        g0.m_out_oy[m] = 0; g0.m_out_ox[m] = 0; g0.m_wp[m] = nullptr; g0.m_stats[m] = nullptr;
    }
    const int nt = descs[0].NcP / 32;
    const int NT = nt % 3 == 0 ? 3 : (nt % 2 == 0 ? 2 : 1);
    hipStream_t s = (hipStream_t)stream;
    const int nreg = g0.nregions;
    // (the regions of the last partial round as a second launch of NT = 1 workgroups: wino32_tail_split)
    const int main_regions = bf3 ? nreg : wino32_tail_split(nreg, 8 * n * (descs[0].NcP / (32 * NT)), NT);
    for (int part = 0; part < 2; ++part) {
        const int r0 = part ? main_regions : 0, r1 = part ? nreg : main_regions, nt_part = part ? 1 : NT;
        if (r1 <= r0) break;
        g0.ncog = descs[0].NcP / (32 * nt_part);
        g0.nmerge = n;
        g0.region0 = r0;
        g0.nregions = r1;
        g0.mblocks = ((r1 - r0 + 7) / 8) * g0.ncog * 8;
        const int nblocks = n * g0.mblocks;
#define W3_LAUNCH_MULTI(NT_, BF_)                                                                                                       \
        do {                                                                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino32_kernel<NT_, BF_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      160 * 1024);                                                                                       \
            hipLaunchKernelGGL((conv_wino32_kernel<NT_, BF_>), dim3(nblocks), dim3(256), W3_LDS_BYTES, s, descs[0], g0);                 \
        } while (0)
        if (bf3) {
            if (nt_part == 3) W3_LAUNCH_MULTI(3, true);
            else if (nt_part == 2) W3_LAUNCH_MULTI(2, true);
            else W3_LAUNCH_MULTI(1, true);
        } else {
            if (nt_part == 3) W3_LAUNCH_MULTI(3, false);
            else if (nt_part == 2) W3_LAUNCH_MULTI(2, false);
            else W3_LAUNCH_MULTI(1, false);
        }
#undef W3_LAUNCH_MULTI
        const int rc = adh_check_launch();
        if (rc) return rc;
    }
    return ADH_OK;
}

extern "C" int adh_conv_wino32_forward_multi(void* stream, const adh_conv_desc* descs, int n) {
    return wino32_forward_multi_impl(stream, descs, n, false);
}
extern "C" int adh_conv_wino32_forward_multi_bf16x3(void* stream, const adh_conv_desc* descs, int n) {
    return wino32_forward_multi_impl(stream, descs, n, true);
}

// U[cls][f = a*4+b][k/4][n][4] = (A g_cls A^T)[a][b], A = [[1,0],[1,1],[1,-1],[0,-1]]; g_cls[ty][tx] is tap
// (ty*cstep + cy, tx*cstep + cx) of the layout L: cstep = 1, one class for a 2x2 layout; cstep = 2, four classes
// (cy, cx) for a 4x4 layout
__global__ void pack_weights_wino32_kernel(const float* __restrict__ src, const adh_wlayout L, int KQ, int NcP, int ncls,
                                           f32x4* __restrict__ wp) {
    const int64_t total = (int64_t)ncls * KQ * NcP;
    const int cstep = ncls == 4 ? 2 : 1;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % NcP);
        int64_t r = idx / NcP;
        const int kq = (int)(r % KQ);
        const int cls = (int)(r / KQ);
        const int cy = cls >> 1, cx = cls & 1;
        f32x4 u[16];
#pragma unroll
        for (int f = 0; f < 16; ++f) u[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (n < L.Nc) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kq * 4 + j;
                if (k >= L.K) continue;
                float gg[2][2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        gg[a][b] = src[(int64_t)L.tap_off0 + (a * cstep + cy) * L.tap_off_sy + (b * cstep + cx) * L.tap_off_sx +
                                       (int64_t)k * L.stride_k + (int64_t)n * L.stride_n];
                float tt[4][2];   // A g
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    tt[0][b] = gg[0][b];
                    tt[1][b] = gg[0][b] + gg[1][b];
                    tt[2][b] = gg[0][b] - gg[1][b];
                    tt[3][b] = -gg[1][b];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a) {   // (A g) A^T
                    u[a * 4 + 0][j] = tt[a][0];
                    u[a * 4 + 1][j] = tt[a][0] + tt[a][1];
                    u[a * 4 + 2][j] = tt[a][0] - tt[a][1];
                    u[a * 4 + 3][j] = -tt[a][1];
                }
            }
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) wp[(((int64_t)cls * 16 + f) * KQ + kq) * NcP + n] = u[f];
    }
}

extern "C" int adh_pack_weights_wino32(void* stream, const float* src, const adh_wlayout* L, float* wp) {
    if (!src || !L || !wp || L->K < 1 || L->Nc < 1) return ADH_E_ARG;
    if (!((L->KHt == 2 && L->KWt == 2) || (L->KHt == 4 && L->KWt == 4))) return ADH_E_ARG;
    const int ncls = L->KHt == 4 ? 4 : 1;
    const int KQ = adh_round_up(L->K, 8) / 4;
    const int NcP = adh_round_up(L->Nc, 32);
    const int64_t total = (int64_t)ncls * KQ * NcP;
    hipLaunchKernelGGL(pack_weights_wino32_kernel, dim3(adh_min_i(adh_ceil_div(total, 128), 4096)), dim3(128), 0,
                       (hipStream_t)stream, src, *L, KQ, NcP, ncls, reinterpret_cast<f32x4*>(wp));
    return adh_check_launch();
}

// The same U (the same fp32 sums and differences), split exactly into three bf16 planes and laid out for
// conv_wino32_kernel<NT, true>: [class][channel group of 32 NT][chunk of 16 k][16 f][NT tiles][3 planes][half h][32 n][8 k] bf16
// (the eight k of half h: channels 4h .. 4h+3 and 8+4h .. 8+4h+3 of the chunk), NT as the launch picks it from NcP.
// wp: ncls * 16 * Kp * NcP * 6 bytes (Kp = K rounded up to 16, NcP = Nc rounded up to 32).
__global__ void pack_weights_wino32_bf16x3_kernel(const float* __restrict__ src, const adh_wlayout L, int KO, int NcP, int ncls, int NT,
                                                  u32x4* __restrict__ wp) {
    const int64_t total = (int64_t)ncls * KO * NcP;
    const int cstep = ncls == 4 ? 2 : 1;
    const int nchunks = KO / 2, ncog = NcP / (32 * NT);
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx % NcP);
        int64_t r = idx / NcP;
        const int ko = (int)(r % KO);
        const int cls = (int)(r / KO);
        const int cy = cls >> 1, cx = cls & 1;
        const int cog = n / (32 * NT), j = (n >> 5) % NT, l31 = n & 31, chunk = ko >> 1, hh = ko & 1;
        float u[16][8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = chunk * 16 + (i < 4 ? 4 * hh + i : 8 + 4 * hh + (i - 4));
            float gg[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    gg[a][b] = (n < L.Nc && k < L.K)
                                   ? src[(int64_t)L.tap_off0 + (a * cstep + cy) * L.tap_off_sy + (b * cstep + cx) * L.tap_off_sx +
                                         (int64_t)k * L.stride_k + (int64_t)n * L.stride_n]
                                   : 0.f;
            float tt[4][2];   // A g
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                tt[0][b] = gg[0][b];
                tt[1][b] = gg[0][b] + gg[1][b];
                tt[2][b] = gg[0][b] - gg[1][b];
                tt[3][b] = -gg[1][b];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {   // (A g) A^T
                u[a * 4 + 0][i] = tt[a][0];
                u[a * 4 + 1][i] = tt[a][0] + tt[a][1];
                u[a * 4 + 2][i] = tt[a][0] - tt[a][1];
                u[a * 4 + 3][i] = -tt[a][1];
            }
        }
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            unsigned pl[3][4];
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) {
                const float x0 = u[f][2 * i2], x1 = u[f][2 * i2 + 1];
                const unsigned hi = w4b_cvt_pk(x0, x1);
                const float r0 = x0 - __builtin_bit_cast(float, hi << 16), r1 = x1 - __builtin_bit_cast(float, hi & 0xffff0000u);
                const unsigned mid = w4b_cvt_pk(r0, r1);
                const float s0 = r0 - __builtin_bit_cast(float, mid << 16), s1 = r1 - __builtin_bit_cast(float, mid & 0xffff0000u);
                pl[0][i2] = hi;
                pl[1][i2] = mid;
                pl[2][i2] = w4b_cvt_pk(s0, s1);
            }
            const int64_t grp = ((((int64_t)cls * ncog + cog) * nchunks + chunk) * 16 + f) * NT + j;
#pragma unroll
            for (int p2 = 0; p2 < 3; ++p2) wp[(grp * 3 + p2) * 64 + hh * 32 + l31] = u32x4{pl[p2][0], pl[p2][1], pl[p2][2], pl[p2][3]};
        }
    }
}

extern "C" int adh_pack_weights_wino32_bf16x3(void* stream, const float* src, const adh_wlayout* L, void* wp) {
    if (!src || !L || !wp || L->K < 1 || L->Nc < 1) return ADH_E_ARG;
    if (!((L->KHt == 2 && L->KWt == 2) || (L->KHt == 4 && L->KWt == 4))) return ADH_E_ARG;
    const int ncls = L->KHt == 4 ? 4 : 1;
    const int KO = adh_round_up(L->K, 16) / 8;
    const int NcP = adh_round_up(L->Nc, 32);
    const int nt = NcP / 32, NT = nt % 3 == 0 ? 3 : (nt % 2 == 0 ? 2 : 1);
    const int64_t total = (int64_t)ncls * KO * NcP;
    hipLaunchKernelGGL(pack_weights_wino32_bf16x3_kernel, dim3(adh_min_i(adh_ceil_div(total, 128), 4096)), dim3(128), 0,
                       (hipStream_t)stream, src, *L, KO, NcP, ncls, NT, reinterpret_cast<u32x4*>(wp));
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
