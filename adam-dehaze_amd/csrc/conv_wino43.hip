// Winograd F(4x4, 3x3) convolution on fp32 MFMA, gfx950: the 3x3 stride-1 pad-1 convolutions (and their data
// gradients) at 36/144 = 1/4 of the direct algorithm's MFMA FLOPs (F(2x2,3x3) in conv_wino.hip: 4/9).  Same
// descriptor, same fused epilogue; replaces the same ATen conv2d calls
// (/root/reference models/dehazing/base_model.py:11-13,26-41).
//
//   Y(4x4) = A^T [ (G g G^T) .* (B^T d B) ] A        d: 6x6 input patch, g: 3x3 filter   (Lavin & Gray, 2015)
//
// Interpolation points {0, +-3/4, +-5/4, inf} instead of the textbook {0, +-1, +-2, inf}: same operation count (the
// +- pairs keep the even/odd split), every constant still a dyadic rational, and 3-4x less rounding error -- simulated
// in fp32 with sequential accumulation over 96..384 channels: 1.5e-6..3.4e-6 of the output scale against 5e-6..1e-5 for
// the textbook points and 1.0e-6..1.5e-6 for the direct algorithm (point choice after Barabasz et al., "Error analysis
// and improving the accuracy of Winograd convolution for deep neural networks", 2018).  fp32 arithmetic throughout.
//
// Workgroup = 4 waves x 512 registers, PERSISTENT since round 4: one per CU, walking virtual blocks
// v = blockIdx.x + k gridDim.x = (output-channel group, region); output region 16 rows x 32 cols = 4 x 8 tiles of 4x4 = one
// 32-tile MFMA row block, x 32*NT output channels; the 36 frequencies are dealt 9 per wave: 9*NT accumulator tiles, the first
// 16 pinned to AGPRs.  Per chunk of 16 input channels:
//   * the 18 x 34 raw halo arrives by LDS-DMA in the layout [row][column mod 4][index][16 ch] (34 pixel slots per row)
//     so that the patch columns of adjacent tiles are adjacent in LDS; every lane's source offset comes from a 612-entry
//     slot table (one region-independent table of offsets relative to the halo's first pixel; border regions get an absolute
//     one): all lanes always load, a slot outside the image has an OUT-OF-RANGE offset, for which the LDS-DMA unit writes zeros
//     (tools/micro/dma_oob.hip) -- the padding costs no pass over the landed halo; 40 pieces per chunk, 10 per wave, staged
//     during the previous chunk's contraction -- the last chunk of a region stages the first chunk of the workgroup's NEXT region;
//   * the input transform runs as two 1-D passes through LDS (column pass raw -> T, row pass T -> V in place): ~12
//     registers live instead of the 72 a 6x6 tile would need next to 27 accumulator tiles;
//   * contraction: per wave 18 groups (9 frequencies x two 8-channel halves) of 4*NT MFMAs; A = one ds_read_b128 of
//     V[f][tile][quad ^ swizzle(tile)], B = NT global_load_dwordx4 of U[f][k/4][n][4] (adh_pack_weights_wino43), fetched
//     two groups ahead with hand-counted waits (conv_wino.hip, w2_load_b: same contract).
// V is single buffered: a chunk is transformed, then contracted (VALU and fp32 MFMA work do not overlap anyway -- built and
// measured in round 4: a form with the transform of the next eight channels pinned inside the MFMA stream of the current eight,
// V and raw double-buffered at eight channels, was parity-green and exactly as fast; every v_pk_fma_f32 of the transform takes
// ~16 cycles of the fp32 FMA datapath the MFMA runs on wherever it is issued.  profiles/r04_ab_wino43_pipelined_transform.txt,
// commit 32ab26b).
// Epilogue: six rounds of (32-channel tile, half region): accumulators -> LDS M[36][16 tiles][32 co] -- exactly V's 72 KB, so the
// staged halo of the next region survives --, A^T M A and the fused epilogue, the next round's M write dealt over the row pass.
#include "common.h"
#ifndef W_STORE_AUX
#define W_STORE_AUX 0   // cache policy of the epilogue stores (buffer instruction aux bits; 2 = nt)
#endif
#include <cstdlib>
#include <type_traits>

#ifndef W4_DBG
#define W4_DBG 0   // dev builds (-DW4_DBG=n): 1 = skip the input transform, 2 = skip the contraction, 4 = skip the epilogue, 8 = stage only the first chunk
#endif
#ifndef W4_STORE_NOPS
#define W4_STORE_NOPS 1   // wait states - 1 after each epilogue store (see the comment at the store)
#endif
#define W4_KC 16
#define W4_ZERO_C false   // the FIRST / Z template parameters below (first MFMA on a tile with C = 0 instead of zeroed accumulators) are the
                          // switch of a measured experiment: they need the contraction instantiated twice behind `if (c == 0)`, and hipcc then
                          // spills ~700 registers; the product zeroes the accumulators with MFMAs at the top of a region
#define W4_TILES 32
#define W4_VF (36 * W4_TILES * W4_KC)              // floats of V (73,728 B)
#define W4_SLOTS 612                               // pixel slots of the raw halo: 18 rows x 34 columns
#define W4_ROWSLOTS 34
#define W4_RAWF (40 * 256)                         // floats of the raw buffer: 40 DMA pieces of 16 slots (640 >= 612 slots)
#define W4_TAB (W4_VF + W4_RAWF)                   // float offset of the lane offset tables [3][640 slots][4 quads]: relative | absolute x 2
#define W4_TABF (4 * 640)                          // floats of one table
#define W4_MF (36 * 16 * 32)                       // floats of the epilogue's M: 36 frequencies x 16 tiles x 32 co = exactly V (73,728 B)
#define W4_RED (W4_TAB + 3 * W4_TABF)              // statistics scratch (behind the tables)
#define W4_LDS_BYTES ((W4_RED + 2 * 8 * 32) * 4)
static_assert(W4_MF <= W4_VF, "the epilogue's M must not reach the raw halo: the next region's first chunk lands there meanwhile");
#ifndef W4_P2
#define W4_P2 0                                    // DMA pieces (of 10 per wave and chunk) issued during transform pass 2 ...
#endif
#define W4_PIECE_G0 2                              // ... the others one per contraction group from group W4_PIECE_G0 on

// ---- bf16 x 3 contraction (round 4, opt-in: adh_conv_wino43_forward_bf16x3).  Same regions, same transform, same epilogue;
// the contraction runs on v_mfma_f32_32x32x16_bf16 with both operands split EXACTLY into three bf16 planes (x = hi + mid + lo:
// 8 + 8 + 8 significant bits, each plane the round-to-nearest of what the previous ones left) and the six significant cross
// terms hi*hi, hi*mid, hi*lo, mid*hi, mid*mid, lo*hi accumulated in fp32: 162 MFMAs of 32 cycles per chunk and wave instead of
// 216 of 64 (tools/micro/bf16split.hip: 2.5x at 0.8x the fp32 MFMA's error against fp64; bf16ring.hip: the weight stream keeps
// up at 1.12 - 1.38x the MFMA floor).  U is split at pack time.  V stays fp32 in LDS, exactly as the fp32 form writes it; every
// lane splits the eight values of its own A operand in registers, in the issue slots the bf16 MFMAs leave free (one wave per
// SIMD: a VALU instruction behind an MFMA costs its issue cycles only while fewer than ~5 ride in a 32-cycle gap) -- a first
// version that split in transform pass 2 and kept three planes in LDS paid 1.5 us per chunk for it (profiles/r04_bf16x3_*).
// With V at 72 KB the raw halo is DOUBLE buffered: the pieces of chunk c + 1 are issued during pass 1 of chunk c and have landed
// before the first weight wait of chunk c's contraction (vmcnt retires in order: a piece issued between two weight waits makes
// the second one wait for HBM).  LDS map (bytes): [0, 73728) V | [73728, 155648) raw halo x 2 | [155648, 160768) slot tables,
// one entry per slot (the channel quad is added per lane).
#define W4B_RAW_B 73728
#define W4B_RAWBUF_B 40960
#define W4B_TAB_B 155648                           // three tables of 640 offsets: relative | absolute x 2
#define W4B_LDS_BYTES (W4B_TAB_B + 3 * 2560)       // (the statistics scratch sits in the raw buffer the last chunk has consumed)
static_assert(W4B_LDS_BYTES <= 160 * 1024, "LDS map of the bf16 x 3 variant");
#define B3_NO_SPLIT ((W4_DBG & 16) != 0)
#include "bf16x3.h"

typedef __attribute__((address_space(3))) void* lds_void_ptr4;

#ifdef W4_PROF   // dev build (tools/prof_wino43.sh): per-workgroup s_memtime stamps and the CU each workgroup ran on
__device__ unsigned long long w4_prof_buf[16384 * 32];
#define W4_STAMP(i) do { if (tid == 0 && v < 16384) w4_prof_buf[v * 32 + (i)] = __builtin_readcyclecounter(); } while (0)   // v = the virtual block id (one record per region)
extern "C" int adh_w4_prof_read(void* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(w4_prof_buf), sizeof(w4_prof_buf)) == hipSuccess ? 0 : -1;
}
#else
#define W4_STAMP(i) do {} while (0)
#endif

struct Wino43Geom {
    int tiles_x, tiles_y;        // 32-col x 16-row regions
    int nregions;
    int nchunks;
    int KQtot;
    int ncog;
    int nvblocks;                // virtual blocks = 8-aligned regions x output-channel groups, dealt to the persistent workgroups
    unsigned m_ncog, m_tx, m_ty; // ceil(2^32 / divisor) of ncog, tiles_x, tiles_y (0: divide): n / d = mulhi(n, m) for n d < 2^32
    const float* bn_mean;        // BNRED launches: the producing layer's batch mean (centres the second sum)
};

// Register class of accumulator tile (frequency FI, channel tile J): 16 tiles fit the 256 AGPRs, the other 9*NT - 16 live
// in VGPRs.  The VGPR-resident ones are those of the channel tiles the epilogue drains FIRST (all of J = 0, then the
// last frequencies of J = 1), so that its output transform does not run next to 128 live accumulator VGPRs.
template <int NT, int FI, int J>
constexpr bool w4_in_agpr() {
    constexpr int nv = 9 * NT - 16;                      // tiles that must live in VGPRs (<= 0: none)
    if constexpr (nv <= 0) return true;
    else if constexpr (nv <= 9) return !(J == 0 && FI >= 9 - nv);
    else return !(J == 0 || (J == 1 && FI >= 9 - (nv - 9)));
}
// Z: the first MFMA on an accumulator tile in a region's first chunk takes C = 0 instead of the tile (432 registers per lane are
// not zeroed once per region: ~1 us of v_mov / v_accvgpr_write with nothing to hide behind in a persistent workgroup).  The
// constraint stays "+": the compiler sees a read-modify-write either way, the hardware ignores the old contents.
template <bool AGPR, bool Z = false>
__device__ __forceinline__ void w4_mfma(f32x16& c, float a, float b) {
    if constexpr (Z) {
        if constexpr (AGPR) asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "+a"(c) : "v"(a), "v"(b));
        else asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "+v"(c) : "v"(a), "v"(b));
    } else {
        if constexpr (AGPR) asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
        else asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
}
// one group: local frequency FI, NT output-channel tiles, k-steps KK .. KEND-1 (a whole group: 0 .. 3)
#ifndef W4_MFMA_ZERO
#define W4_MFMA_ZERO 1   // 1: accumulators zeroed by one v_mfma_f32_32x32x16_bf16 with zero operands and C = 0 per tile (16 registers in 32 cycles
#endif                   // of the matrix pipe) instead of 16 v_mov / v_accvgpr_write each
template <int NT, int T = 0>
__device__ __forceinline__ void w4_acc_zero_(f32x16 (&acc)[9 * NT], const u32x4& z) {
    if constexpr (T < 9 * NT) {
        if constexpr (w4_in_agpr<NT, T / NT, T % NT>()) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=a"(acc[T]) : "v"(z));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=&v"(acc[T]) : "v"(z));
        w4_acc_zero_<NT, T + 1>(acc, z);
    }
}
template <int NT>
__device__ __forceinline__ void w4_acc_zero(f32x16 (&acc)[9 * NT]) {
    u32x4 z = {0u, 0u, 0u, 0u};
    asm volatile("s_nop 1" : "+v"(z));   // (an MFMA reads its operands two wait states after a VALU write at the earliest: common.h)
    w4_acc_zero_<NT>(acc, z);
}
// (Z: this group is the first one on its accumulator tiles in the region: k-step 0 starts them from zero)
template <int NT, int FI, int KK, int J, int KEND = 4, bool Z = false>
__device__ __forceinline__ void w4_group(f32x16 (&acc)[9 * NT], const f32x4& a, const f32x4 (&b)[NT]) {
    if constexpr (KK < KEND) {
        w4_mfma<w4_in_agpr<NT, FI, J>(), (Z && KK == 0)>(acc[FI * NT + J], a[KK], b[J][KK]);
        if constexpr (J + 1 < NT) w4_group<NT, FI, KK, J + 1, KEND, Z>(acc, a, b);
        else w4_group<NT, FI, KK + 1, 0, KEND, Z>(acc, a, b);
    }
}
// does contraction group G carry a staging piece?  (dev build W4_DBG & 8: no staging inside the loop)
constexpr int w4_piece(int G) { return (!(W4_DBG & 8) && G >= W4_PIECE_G0 && G < W4_PIECE_G0 + 10 - W4_P2) ? 1 : 0; }
constexpr int w4_p2_pieces() { return (W4_DBG & 8) ? 0 : W4_P2; }

struct W4Stage {                 // what a contraction group needs to issue one LDS-DMA piece of the NEXT chunk's raw halo
    __amdgpu_buffer_rsrc_t rsrc; // the image
    const int* tab_lane;         // this lane's entry of the offset table; piece u at [256 u]
    float* lds;
    int lds_wave;                // byte offset of this wave's piece 0 in LDS (wave-uniform)
    int cb;                      // byte offset of the chunk's first channel (wave-uniform)
};
template <int NT>
__device__ __forceinline__ void w4_load_b(f32x4 (&b)[NT], unsigned voff, const float* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[0]) : "v"(voff), "s"(sbase) : "memory");
    if constexpr (NT > 1) asm volatile("global_load_dwordx4 %0, %1, %2 offset:512" : "=v"(b[1]) : "v"(voff), "s"(sbase) : "memory");
    if constexpr (NT > 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(b[2]) : "v"(voff), "s"(sbase) : "memory");
}
template <int N, int NT>
__device__ __forceinline__ void w4_wait_b(f32x4 (&b)[NT]) {
    if constexpr (NT == 1) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(b[0]) : "n"(N) : "memory");
    if constexpr (NT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(b[0]), "+v"(b[1]) : "n"(N) : "memory");
    if constexpr (NT == 3) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) : "n"(N) : "memory");
}

// Points +-a, +-b:  a = 3/4, b = 5/4
#define W4_A 0.75f
#define W4_B 1.25f
#define W4_A2 0.5625f        // a^2
#define W4_B2 1.5625f        // b^2
#define W4_A3 0.421875f      // a^3
#define W4_B3 1.953125f      // b^3
#define W4_A2B2 0.87890625f  // a^2 b^2
#define W4_S2 2.125f         // a^2 + b^2

// 1-D input transform B^T (6 points), written so that at most eight float4 values are live: the results go to
// LDS (dst + a * stride floats) as soon as they exist.  Rows of B^T = coefficients of prod_{q != p} (x - q):
//   p = 0:    a^2 b^2 d0 - (a^2 + b^2) d2 + d4          p = +-a:  (d4 - b^2 d2) +- a (d3 - b^2 d1)
//   p = inf:  a^2 b^2 d1 - (a^2 + b^2) d3 + d5          p = +-b:  (d4 - a^2 d2) +- b (d3 - a^2 d1)
struct W4Neg { float s2, b2, a2, a, b; };   // -(a^2+b^2), -b^2, -a^2, -a, -b held in SGPRs the compiler cannot see through:
// it would otherwise rewrite `(-c) * x + y` as fma(-x, c, y) and spend a v_xor per register on the sign
__device__ __forceinline__ W4Neg w4_neg_constants() {
    W4Neg n = {-W4_S2, -W4_B2, -W4_A2, -W4_A, -W4_B};
    asm volatile("" : "+s"(n.s2), "+s"(n.b2), "+s"(n.a2), "+s"(n.a), "+s"(n.b));
    return n;
}
__device__ __forceinline__ void w4_bt_store(f32x4 (&d)[6], float* dst, int stride, const W4Neg& n) {
    d[0] = W4_A2B2 * d[0] + (n.s2 * d[2] + d[4]);
    *reinterpret_cast<f32x4*>(dst) = d[0];
    d[5] = W4_A2B2 * d[1] + (n.s2 * d[3] + d[5]);
    *reinterpret_cast<f32x4*>(dst + 5 * stride) = d[5];
    const f32x4 p = n.b2 * d[2] + d[4], r = n.a2 * d[2] + d[4];
    const f32x4 q = n.b2 * d[1] + d[3], s = n.a2 * d[1] + d[3];
    *reinterpret_cast<f32x4*>(dst + 1 * stride) = W4_A * q + p;
    *reinterpret_cast<f32x4*>(dst + 2 * stride) = n.a * q + p;
    *reinterpret_cast<f32x4*>(dst + 3 * stride) = W4_B * s + r;
    *reinterpret_cast<f32x4*>(dst + 4 * stride) = n.b * s + r;
}

// accumulators of channel tile J, tile half TH (accumulator registers 8 TH .. 8 TH + 7 = MFMA rows 16 TH .. 16 TH + 15) ->
// M[9 wave + FI][tile row 8 ((R >> 2) & 1) + 2 (R & 3) + h][l31] (h, l31 are in `addr`), straight from the register class they live
// in (DS instructions take AGPR data operands on gfx950).  M holds 16 tiles: a frequency is 16 rows x 128 B = 2 KB.
template <int NT, int J, int TH, int FI, int R>
__device__ __forceinline__ void w4_store_mh_(const f32x16 (&acc)[9 * NT], unsigned addr) {
    if constexpr (FI < 9) {
        // registers RR and RR + 1 sit two tile rows = 256 B apart: one ds_write2st64_b32 (offsets in units of 256 B) moves both
        constexpr int RR = 8 * TH + R;
        constexpr int off = FI * 8 + ((R >> 2) & 1) * 4 + (R & 3);
        if constexpr (w4_in_agpr<NT, FI, J>())
            asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(addr), "a"(acc[FI * NT + J][RR]),
                         "a"(acc[FI * NT + J][RR + 1]), "n"(off), "n"(off + 1)
                         : "memory");
        else
            asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(addr), "v"(acc[FI * NT + J][RR]),
                         "v"(acc[FI * NT + J][RR + 1]), "n"(off), "n"(off + 1)
                         : "memory");
        if constexpr (R + 2 < 8) w4_store_mh_<NT, J, TH, FI, R + 2>(acc, addr);
        else w4_store_mh_<NT, J, TH, FI + 1, 0>(acc, addr);
    }
}
template <int NT, int J>
__device__ __forceinline__ void w4_store_mh(const f32x16 (&acc)[9 * NT], unsigned addr, int th) {   // th: a constant after unrolling
    if (th == 0) w4_store_mh_<NT, J, 0, 0, 0>(acc, addr);
    else w4_store_mh_<NT, J, 1, 0, 0>(acc, addr);
}
// one frequency (four instructions) of the same: the epilogue deals the next round's M write over the pixel steps of the current
// round's row pass (36 ds_write2st64_b32 in a row stall the wave on the LDS queue for ~0.45 us; between VALU work and stores they are free)
template <int NT, int J, int TH, int FI>
__device__ __forceinline__ void w4_store_mh_one(const f32x16 (&acc)[9 * NT], unsigned addr) {
#pragma unroll
    for (int R = 0; R < 8; R += 2) {
        const int RR = 8 * TH + R;
        const int off = FI * 8 + ((R >> 2) & 1) * 4 + (R & 3);
        if constexpr (w4_in_agpr<NT, FI, J>())
            asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(addr), "a"(acc[FI * NT + J][RR]),
                         "a"(acc[FI * NT + J][RR + 1]), "n"(off), "n"(off + 1)
                         : "memory");
        else
            asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(addr), "v"(acc[FI * NT + J][RR]),
                         "v"(acc[FI * NT + J][RR + 1]), "n"(off), "n"(off + 1)
                         : "memory");
    }
}
template <int NT, int J, int TH>
__device__ __forceinline__ void w4_store_mh_fi(const f32x16 (&acc)[9 * NT], unsigned addr, int fi) {   // fi: a constant after unrolling
    if (fi == 0) w4_store_mh_one<NT, J, TH, 0>(acc, addr);
    if (fi == 1) w4_store_mh_one<NT, J, TH, 1>(acc, addr);
    if (fi == 2) w4_store_mh_one<NT, J, TH, 2>(acc, addr);
    if (fi == 3) w4_store_mh_one<NT, J, TH, 3>(acc, addr);
    if (fi == 4) w4_store_mh_one<NT, J, TH, 4>(acc, addr);
    if (fi == 5) w4_store_mh_one<NT, J, TH, 5>(acc, addr);
    if (fi == 6) w4_store_mh_one<NT, J, TH, 6>(acc, addr);
    if (fi == 7) w4_store_mh_one<NT, J, TH, 7>(acc, addr);
    if (fi == 8) w4_store_mh_one<NT, J, TH, 8>(acc, addr);
}
// frequency fi of round (j, th) (constants after unrolling; nothing for j >= NT)
template <int NT>
__device__ __forceinline__ void w4_store_mh_round(const f32x16 (&acc)[9 * NT], unsigned addr, int j, int th, int fi) {
    if (j == 0 && th == 0) w4_store_mh_fi<NT, 0, 0>(acc, addr, fi);
    if (j == 0 && th == 1) w4_store_mh_fi<NT, 0, 1>(acc, addr, fi);
    if constexpr (NT > 1) {
        if (j == 1 && th == 0) w4_store_mh_fi<NT, 1, 0>(acc, addr, fi);
        if (j == 1 && th == 1) w4_store_mh_fi<NT, 1, 1>(acc, addr, fi);
    }
    if constexpr (NT > 2) {
        if (j == 2 && th == 0) w4_store_mh_fi<NT, 2, 0>(acc, addr, fi);
        if (j == 2 && th == 1) w4_store_mh_fi<NT, 2, 1>(acc, addr, fi);
    }
}

// runs the 18 groups of one chunk; GI = group index (frequency GI % 9, channel half GI / 9)
// (FIRST: the region's first chunk -- groups 0 .. 8 start their accumulator tiles from zero)
template <int NT, int GI, bool FIRST = false>
__device__ __forceinline__ void w4_chunk(f32x16 (&acc)[9 * NT], f32x4 (&av)[2], f32x4 (&bv)[3][NT], const float* vlane,
                                         const float* vlane1, unsigned b_voff, const float* b_chunk, const float* b_next, int64_t b_fstride,
                                         int b_kq2, const W4Stage& st) {
    if constexpr (GI < 18) {
        // operands two groups ahead (weights) / one group ahead (V)
        constexpr int G2 = GI + 2;
        if constexpr (G2 < 18) w4_load_b<NT>(bv[G2 % 3], b_voff, b_chunk + (G2 % 9) * b_fstride + (G2 / 9) * b_kq2);
        else w4_load_b<NT>(bv[G2 % 3], b_voff, b_next + (G2 - 18) * b_fstride);
        if constexpr (GI + 1 < 18) {
            constexpr int G1 = GI + 1;
            av[G1 & 1] = *reinterpret_cast<const f32x4*>((G1 / 9 ? vlane1 : vlane) + (G1 % 9) * (W4_TILES * W4_KC));
        }
        // newer than this group's weights: the weights of the next two groups and the pieces of the previous two
        if constexpr (w4_piece(GI)) {
            constexpr int u = W4_P2 + GI - W4_PIECE_G0;
            const int vo = st.tab_lane[256 * u];
            w4_wait_b<2 * NT + w4_piece(GI - 2) + w4_piece(GI - 1), NT>(bv[GI % 3]);
            w4_group<NT, GI % 9, 0, 0, 1, (FIRST && GI < 9)>(acc, av[GI & 1], bv[GI % 3]);
            __builtin_amdgcn_sched_barrier(0);
            // one piece of the next chunk's halo, behind the first k-step: the MFMA pipe is busy for NT * 64 cycles and the
            // raw buffer has no reader between the transform of this chunk and the end of its contraction
            __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, (lds_void_ptr4)(reinterpret_cast<char*>(st.lds) + st.lds_wave + u * 4096),
                                                     16, vo, st.cb, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            w4_group<NT, GI % 9, 1, 0>(acc, av[GI & 1], bv[GI % 3]);
        } else {
            w4_wait_b<2 * NT + w4_piece(GI - 2) + w4_piece(GI - 1), NT>(bv[GI % 3]);
            w4_group<NT, GI % 9, 0, 0, 4, (FIRST && GI < 9)>(acc, av[GI & 1], bv[GI % 3]);
        }
        w4_chunk<NT, GI + 1, FIRST>(acc, av, bv, vlane, vlane1, b_voff, b_chunk, b_next, b_fstride, b_kq2, st);
    }
}

// the whole contraction of one chunk (fp32 form): groups 0 and 1 by hand (their weights were requested before the transform), then w4_chunk
template <int NT, bool FIRST>
__device__ __forceinline__ void w4_contract(f32x16 (&acc)[9 * NT], f32x4 (&av)[2], f32x4 (&bv)[3][NT], const float* vlane, const float* vlane1,
                                            unsigned b_voff, const float* b_chunk, const float* b_next, int64_t b_fstride, int b_kq,
                                            const W4Stage& st) {
    av[0] = *reinterpret_cast<const f32x4*>(vlane);
    // group 0
    w4_load_b<NT>(bv[2], b_voff, b_chunk + 2 * b_fstride);
    av[1] = *reinterpret_cast<const f32x4*>(vlane + 1 * (W4_TILES * W4_KC));
    w4_wait_b<2 * NT + w4_p2_pieces(), NT>(bv[0]);
    w4_group<NT, 0, 0, 0, 4, FIRST>(acc, av[0], bv[0]);
    // group 1
    w4_load_b<NT>(bv[0], b_voff, b_chunk + 3 * b_fstride);
    av[0] = *reinterpret_cast<const f32x4*>(vlane + 2 * (W4_TILES * W4_KC));
    w4_wait_b<2 * NT + w4_p2_pieces(), NT>(bv[1]);
    w4_group<NT, 1, 0, 0, 4, FIRST>(acc, av[1], bv[1]);
    // groups 2..17
    w4_chunk<NT, 2, FIRST>(acc, av, bv, vlane, vlane1, b_voff, b_chunk, b_next, b_fstride, 2 * b_kq, st);
}

// ------------------------------------------------------------------------------------------------ bf16 x 3 contraction (helpers: bf16x3.h)
// the gap behind MFMA M (1 .. NM = 6 NT) of a frequency in which step S runs: from gap 3 on (the values requested in gap NM - 1 of
// the frequency before have landed by then), the last one in gap NM - 1
template <int NM>
constexpr int w4b_gap_of(int S) { return 3 + S * (NM - 3) / 20; }
template <int NM, int M, int S = 0>
constexpr int w4b_first_step() {   // first step whose gap is >= M
    if constexpr (S >= 20) return 20;
    else if constexpr (w4b_gap_of<NM>(S) >= M) return S;
    else return w4b_first_step<NM, M, S + 1>();
}
__device__ __forceinline__ void w4b_load_s(W4BNext& n, const float* v0, const float* v1) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(v0), b = *reinterpret_cast<const f32x4*>(v1);
    n.s[0] = a[0]; n.s[1] = a[1]; n.s[2] = a[2]; n.s[3] = a[3];
    n.s[4] = b[0]; n.s[5] = b[1]; n.s[6] = b[2]; n.s[7] = b[3];
}
// what rides behind MFMA M of frequency FI: the steps of its gap towards frequency FI + 1's planes and, behind the last step, the
// request for frequency FI + 2's values; pinned there (the scheduler would otherwise bunch the VALU work of a frequency together,
// and the MFMA pipe idles behind a bunch)
template <int NT, int FI, int M>
__device__ __forceinline__ void w4b_gap(W4BNext& n, const float* vlane, const float* vlane1, float m1) {
    constexpr int NM = 6 * NT;
    if constexpr (FI + 1 < 9) {
        w4b_steps<w4b_first_step<NM, M>(), w4b_first_step<NM, M + 1>()>(n, m1);
        if constexpr (M == NM - 1 && FI + 2 < 9) w4b_load_s(n, vlane + (FI + 2) * (W4_TILES * W4_KC), vlane1 + (FI + 2) * (W4_TILES * W4_KC));
    }
    __builtin_amdgcn_sched_barrier(0);
}
// groups G .. 9 NT - 1 of one chunk: group = (frequency G / NT, channel tile G % NT) = three weight loads two groups ahead
// (ring of three) and six MFMAs on one accumulator tile (back to back on one accumulator is full rate for this instruction), in
// the order hi*hi, hi*mid, hi*lo | mid*hi, mid*mid | lo*hi.  NOTHING is requested across the chunk boundary.  (Tried, as the fp32
// form does it: the last two groups requesting the next chunk's groups 0 and 1.  hipcc resolved the loop-carried ring with register
// COPIES -- v_mov of a register whose load was still in flight, placed right behind the load, in front of any wait the source can
// express -- and interior regions, where nothing slow sits between the last group and the back edge, computed garbage.  It bought
// nothing either: 1.881 / 1.428 / 1.374 ms against 1.879 / 1.443 / 1.380, the staging pieces gate group 2 instead of group 0.
// tests/test_host_cpu.py now walks both forms' ISA for reads of a register between its asm load and the wait that covers it.)
#ifndef W4B_P1
#define W4B_P1 5      // dev: staging pieces issued in pass 1 (the rest in pass 2)
#endif
template <int NT, int G, bool FIRST = false>   // FIRST: the region's first chunk -- every group starts its accumulator tile from zero
__device__ __forceinline__ void w4b_groups(f32x16 (&acc)[9 * NT], u32x4 (&a)[3], W4BNext& n, u32x4 (&bv)[3][3], const float* vlane,
                                           const float* vlane1, unsigned b_voff, const char* b_chunk, float m1) {
    if constexpr (G < 9 * NT) {
        constexpr int FI = G / NT, J = G % NT, G2 = G + 2;
        if constexpr (G2 < 9 * NT) w4b_load_b(bv[G2 % 3], b_voff, b_chunk + G2 * 3072);
        constexpr int newer = 3 * (9 * NT - 1 - G < 2 ? 9 * NT - 1 - G : 2);
        constexpr bool agpr = w4_in_agpr<NT, FI, J>();
        u32x4(&b)[3] = bv[G % 3];
        w4b_wait_b<newer>(b);
        w4b_mfma<agpr, FIRST>(acc[G], a[0], b[0]);
        w4b_gap<NT, FI, 6 * J + 1>(n, vlane, vlane1, m1);
        w4b_mfma<agpr>(acc[G], a[0], b[1]);
        w4b_gap<NT, FI, 6 * J + 2>(n, vlane, vlane1, m1);
        w4b_mfma<agpr>(acc[G], a[0], b[2]);
        w4b_gap<NT, FI, 6 * J + 3>(n, vlane, vlane1, m1);
        w4b_mfma<agpr>(acc[G], a[1], b[0]);
        w4b_gap<NT, FI, 6 * J + 4>(n, vlane, vlane1, m1);
        w4b_mfma<agpr>(acc[G], a[1], b[1]);
        w4b_gap<NT, FI, 6 * J + 5>(n, vlane, vlane1, m1);
        w4b_mfma<agpr>(acc[G], a[2], b[0]);
        if constexpr (J == NT - 1 && FI + 1 < 9) {
            // hand-over: the planes of frequency FI + 1 (written by VALU instructions at least one MFMA ago; the asm holds the two wait
            // states an MFMA operand needs behind a VALU write in case a register copy lands here: hipcc's hazard recogniser does not
            // look into the MFMA statements)
            a[0] = u32x4{n.h[0], n.h[1], n.h[2], n.h[3]};
            a[1] = u32x4{n.m[0], n.m[1], n.m[2], n.m[3]};
            a[2] = u32x4{n.l[0], n.l[1], n.l[2], n.l[3]};
            asm volatile("s_nop 1" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]));
        } else {
            w4b_gap<NT, FI, 6 * J + 6>(n, vlane, vlane1, m1);
        }
        w4b_groups<NT, G + 1, FIRST>(acc, a, n, bv, vlane, vlane1, b_voff, b_chunk, m1);
    }
}
// 1-D input transform as w4_bt_store, with staging piece pb + i of the next chunk riding behind the i-th store (i < np), one at
// a time: a burst backs the address unit up.  (No callable parameter: a __global__ template whose body passes a lambda into a
// function template loses its host stub in hipcc.)
__device__ __forceinline__ void w4b_bt_store_staging(f32x4 (&d)[6], float* dst, int stride, const W4Neg& n, const W4Stage& st,
                                                     const int (&vo)[4], int lds_buf, int pb, int np) {
    int nput = 0;
    auto put = [&](int a, const f32x4& v) {
        *reinterpret_cast<f32x4*>(dst + a * stride) = v;
        if (nput < np) {
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, (lds_void_ptr4)(reinterpret_cast<char*>(st.lds) + lds_buf + (pb + nput) * 4096), 16,
                                                     vo[nput], st.cb, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        ++nput;
    };
    d[0] = W4_A2B2 * d[0] + (n.s2 * d[2] + d[4]);
    put(0, d[0]);
    d[5] = W4_A2B2 * d[1] + (n.s2 * d[3] + d[5]);
    put(5, d[5]);
    const f32x4 p = n.b2 * d[2] + d[4], r = n.a2 * d[2] + d[4];
    const f32x4 q = n.b2 * d[1] + d[3], s = n.a2 * d[1] + d[3];
    put(1, W4_A * q + p);
    put(2, n.a * q + p);
    put(3, W4_B * s + r);
    put(4, n.b * s + r);
}

// where a region sits: image, output origin, byte offset of the halo's first pixel (0 for a border region), halo not inside the image
// (namespace scope: a local type inside the __global__ template costs the kernel its host stub in hipcc)
struct W4Where { int n, oy0, ox0, base; bool border; };

template <int NT, bool BNRED = false, bool BF3 = false>
__global__ __launch_bounds__(256, 1) void conv_wino43_kernel(const adh_conv_desc d, const Wino43Geom g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // V | raw | two slot tables | red ; the epilogue's M aliases V only
    constexpr int RAW0 = BF3 ? W4B_RAW_B / 4 : W4_VF;                   // float offset of the raw halo
    // Slot tables: source byte offset of every pixel slot of the 18 x 34 halo.  fp32 form: [640][4] (slot s, channel quad q); bf16 x 3
    // form: one entry per slot (the lane adds its quad) and two raw buffers.  A slot outside the image (and the padding slots >= 612)
    // has BIT 31 of its offset set: out of range against num_records = 0x7fffffff for every scalar offset, and the LDS-DMA unit
    // writes ZEROS for an out-of-range lane (tools/micro/dma_oob.hip) -- the padding of the convolution costs no flags and no pass
    // over the landed halo.  Table 0 holds offsets RELATIVE to the halo's first pixel and serves every region whose halo lies inside
    // the image (the region's origin goes into the scalar offset of the pieces): built once per workgroup.  A border region gets an
    // absolute table (clamped coordinates, bit 31) in table 1 or 2 -- two, because the persistent workgroup stages the first chunk
    // of its NEXT region during the last contraction of the current one.
    constexpr int TABN = BF3 ? 640 : W4_TABF;                            // ints per table
    int* const tab_base = reinterpret_cast<int*>(lds + (BF3 ? W4B_TAB_B / 4 : W4_TAB));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int h = lane >> 5;

    // ------------------------------------------------------------------ persistent workgroup: virtual block ids v = blockIdx.x + k gridDim.x
    // (gridDim.x is a multiple of 8: v & 7 = the XCD, as for a plain launch).  v -> (output-channel group, region) as before.
    int v = blockIdx.x;
    // n / dv for a workgroup-uniform n: one s_mul_hi_u32 with the host's reciprocal (the decode of a region is six divisions by
    // launch constants; as runtime divisions they were a third of the 1.8 us a persistent workgroup spent between two regions)
    // (readfirstlane: the multiply is a VALU instruction, and a scalar offset or resource derived from a VGPR turns every LDS-DMA
    // piece into a waterfall loop)
    auto udiv = [&](int n, int dv, unsigned m) { return __builtin_amdgcn_readfirstlane(m ? (int)__umulhi((unsigned)n, m) : n / dv); };
    auto region_of = [&](int vb, int& cgo) {
        const int q = vb >> 3, qd = udiv(q, g.ncog, g.m_ncog);
        cgo = q - qd * g.ncog;
        return qd * 8 + (vb & 7);
    };
    typedef W4Where Where;
    const int xcs = d.in_cstride * 4;
    auto where_of = [&](int reg) {
        Where w;
        const int r1 = udiv(reg, g.tiles_x, g.m_tx), tx = reg - r1 * g.tiles_x;
        w.n = udiv(r1, g.tiles_y, g.m_ty);
        const int ty = r1 - w.n * g.tiles_y;
        w.oy0 = ty * 16;
        w.ox0 = tx * 32;
        w.border = !(w.oy0 >= 1 && w.oy0 + 17 <= d.IH && w.ox0 >= 1 && w.ox0 + 33 <= d.IW);
        w.base = w.border ? 0 : ((w.oy0 - 1) * d.IW + w.ox0 - 1) * xcs;
        return w;
    };
    // slot s = row*34 + {9,9,8,8 per plane}.  REL: offsets relative to the halo's first pixel, no clamping
    auto build_table = [&](int* tab, int oy0, int ox0, bool rel) {
        for (int s = tid; s < 640; s += 256) {
            const int sc = s < W4_SLOTS ? s : W4_SLOTS - 1;
            const int row = sc / W4_ROWSLOTS, rem = sc - row * W4_ROWSLOTS;
            const int plane = rem < 9 ? 0 : (rem < 18 ? 1 : (rem < 26 ? 2 : 3));
            const int idx = rem - (plane == 0 ? 0 : (plane == 1 ? 9 : (plane == 2 ? 18 : 26)));
            const int iy = rel ? row : oy0 - 1 + row, ix = rel ? 4 * idx + plane : ox0 - 1 + 4 * idx + plane;
            const int iyc = rel ? iy : adh_min_i(adh_max_i(iy, 0), d.IH - 1), ixc = rel ? ix : adh_min_i(adh_max_i(ix, 0), d.IW - 1);
            const bool ok = s < W4_SLOTS && (rel || (iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW));
            const int off = ((iyc * d.IW + ixc) * xcs) | (ok ? 0 : (int)0x80000000);
            typedef int i32x4 __attribute__((ext_vector_type(4)));
            if constexpr (BF3) tab[s] = off;
            else *reinterpret_cast<i32x4*>(tab + 4 * s) = i32x4{off, off + 16, off + 32, off + 48};
        }
    };
    // (image as a uniform POINTER, made a resource where it is used: hipcc selects between two buffer resources in VGPRs, and
    // every LDS-DMA piece that takes the result becomes a waterfall loop)
    auto image_ptr = [&](int n) { return d.in + (int64_t)n * d.IH * d.IW * d.in_cstride; };
    auto image_rsrc = [&](const float* in_n) {
        const uint64_t a = reinterpret_cast<uint64_t>(in_n);
        const uint64_t u = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)a);
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(u), 0, 0x7fffffff, 0x00020000);
    };
    const int cq_l = lane & 3;
    // 40 pieces per chunk, 10 per wave: piece k = 4u + wave covers slots 16k..16k+15; lane l of piece k loads (slot 16k + l / 4,
    // quad l % 4): its table entry is tab[64 k + l], i.e. one contiguous read per piece
    // (bf16 x 3 form: piece k = 4u + wave, lane l stages slot 16k + l / 4: entry tab[64 u + 16 wave + l / 4], + 16 (l % 4) bytes)
    const int tab_lane_o = BF3 ? 16 * wave + (lane >> 2) : 64 * wave + lane;   // this lane's entry of piece 0 inside a table
    W4Stage st;
    st.lds = lds;
    st.lds_wave = __builtin_amdgcn_readfirstlane((RAW0 + wave * 256) * 4);
    st.cb = 0;
    constexpr int TABU = BF3 ? 64 : 256;          // table entries between two pieces of a wave
    const int quad16_o = (lane & 3) * 16;
    // bf16 x 3 form: next to 27 accumulator tiles, the weight ring and the planes in the making, the contraction has fewer than ten
    // registers to spare: nothing a thread needs only OUTSIDE the contraction may stay live across it.  Such per-thread constants are
    // rebuilt from v_mbcnt where they are used (the asm keeps hipcc from hoisting the rebuild back out of the chunk loop).
    auto lane_rebuilt = [&]() {
        int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(l));
        return l;
    };

    // ------------------------------------------------------------------ input transform (two 1-D passes through LDS)
    // thread = (tile, channel quad); pass 1 item k: patch column c = 2k + (tid >> 7); pass 2 item k: frequency row a = ...
    const int tile_t = (tid >> 2) & 31, trow = tile_t >> 3, tcol = tile_t & 7;
    const int csel = wave >> 1;                                                 // 0 / 1 (wave-uniform)
    // swizzled quad slot inside V[f][tile]
    const int vslot_o = tile_t * 16 + ((cq_l ^ ((tile_t >> 1) & 3)) * 4);
    // patch pixel (row i, column c) of this thread's tile sits in raw row 4*trow + i, plane c & 3, index tcol + (c >> 2);
    // slot offset of column c = 2k + csel inside a row: {0, 18, 1} (csel 0: planes 0, 2, 0) / {9, 26, 10} (csel 1)
    // -> two lane pointers (k = 0 and, one slot further, k = 2; k = 1), the patch row is an instruction immediate
    const int rbase_t = ((4 * trow) * W4_ROWSLOTS + tcol) * 16 + cq_l * 4;
    const int raw_k0_o = (RAW0 + rbase_t) / 4 + (csel ? 9 : 0) * 4;      // float4 indices into lds[] (an opaque *pointer* would lose the LDS
    const int raw_k1_o = (RAW0 + rbase_t) / 4 + (csel ? 26 : 18) * 4;    // address space: flat loads; an opaque float index the alignment: b32 reads)
    const W4Neg negc = w4_neg_constants();
    constexpr int p2n = BF3 ? 0 : w4_p2_pieces();   // pieces of the next chunk's halo issued during pass 2 (fp32 form, dev option)
    // cbuf (bf16 x 3 form) = the raw buffer this chunk reads; the other one is staged from table `stab` (st.rsrc / st.cb: image and
    // channel offset of what is staged -- the next chunk of this region or the first chunk of the next one)
    auto transform = [&](int cbuf, const int* stab) {
        int vslot_t = vslot_o, raw_k0 = raw_k0_o, raw_k1 = raw_k1_o, quad16 = quad16_o;
        const int* tabl = stab + tab_lane_o;
        if constexpr (BF3) {   // (see lane_rebuilt)
            const int ln = lane_rebuilt();
            const int tl = ((wave & 1) << 4) | (ln >> 2), cq = ln & 3;
            vslot_t = tl * 16 + ((cq ^ ((tl >> 1) & 3)) * 4);
            const int rb = ((4 * (tl >> 3)) * W4_ROWSLOTS + (tl & 7)) * 16 + cq * 4;
            raw_k0 = (RAW0 + rb) / 4 + (csel ? 9 : 0) * 4;
            raw_k1 = (RAW0 + rb) / 4 + (csel ? 26 : 18) * 4;
            quad16 = cq * 16;
            tabl = stab + 16 * wave + (ln >> 2);
        }
        // pass 1: T[a][c] = sum_i B^T[a][i] d[i][c] for this thread's three columns c = csel, 2 + csel, 4 + csel
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int c = 2 * k + csel;
            f32x4 dd[6];
            int rp = k == 1 ? raw_k1 : raw_k0 + (k == 2 ? 4 : 0);
            if constexpr (BF3) rp += cbuf * (W4B_RAWBUF_B / 16);
            asm volatile("" : "+v"(rp));   // (keeps the six row addresses from being hoisted out of the chunk loop into six registers)
            // bf16 x 3 form: the ten pieces of the NEXT chunk's halo ride behind the stores of the two passes, 2 + 2 + 1 each, into the
            // other raw buffer (all ten in this pass queue up in the address unit, which serves the four waves' pieces one by one)
            int vo[4];
            const int pb = W4B_P1 == 5 ? 2 * k : (k == 0 ? 0 : (k == 1 ? 4 : 7)),
                      pn = (W4_DBG & 8) ? 0 : (W4B_P1 == 5 ? (k == 2 ? 1 : 2) : (k == 0 ? 4 : 3));
            if constexpr (BF3) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < pn) vo[e] = tabl[TABU * (pb + e)] + quad16;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) dd[i] = reinterpret_cast<const f32x4*>(lds)[rp + i * (W4_ROWSLOTS * 4)];
            if constexpr (BF3)
                w4b_bt_store_staging(dd, lds + c * (W4_TILES * W4_KC) + vslot_t, 6 * W4_TILES * W4_KC, negc, st, vo,
                                     st.lds_wave + (cbuf ^ 1) * W4B_RAWBUF_B, pb, pn);
            else
                w4_bt_store(dd, lds + c * (W4_TILES * W4_KC) + vslot_t, 6 * W4_TILES * W4_KC, negc);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // pass 2 (in place): V[a][b] = sum_c T[a][c] B[c][b] for this thread's three rows a = csel, 2 + csel, 4 + csel
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int a = 2 * k + csel;
            f32x4 tt[6];
            // (fp32 form, dev option W4_P2) pieces k, k + 3, .. of the next chunk's halo (the raw buffer's last reader was pass 1): table entries first
            int vo[4];
            const int pb = 5 + 2 * k, pn = (W4_DBG & 8) || W4B_P1 != 5 ? 0 : (k == 2 ? 1 : 2);   // bf16 x 3 form: pieces 5 .. 9
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (BF3) {
                    if (e < pn) vo[e] = tabl[TABU * (pb + e)] + quad16;
                } else {
                    if (k + 3 * e < p2n) vo[e] = tabl[TABU * (k + 3 * e)];
                }
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) tt[c] = *reinterpret_cast<const f32x4*>(lds + (a * 6 + c) * (W4_TILES * W4_KC) + vslot_t);
            if constexpr (BF3)
                w4b_bt_store_staging(tt, lds + (a * 6) * (W4_TILES * W4_KC) + vslot_t, W4_TILES * W4_KC, negc, st, vo,
                                     st.lds_wave + (cbuf ^ 1) * W4B_RAWBUF_B, pb, pn);
            else
                w4_bt_store(tt, lds + (a * 6) * (W4_TILES * W4_KC) + vslot_t, W4_TILES * W4_KC, negc);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k + 3 * e < p2n)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, (lds_void_ptr4)(reinterpret_cast<char*>(lds) + st.lds_wave + (k + 3 * e) * 4096),
                                                             16, vo[e], st.cb, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ------------------------------------------------------------------ contraction operands
    // A: V[f = 9*wave + fi][tile = l31][slot(2g + h)], slot(q) = q ^ ((tile >> 1) & 3); g flips slot bit 1 (byte 32)
    const int sw = (l31 >> 1) & 3;
    const float* const vlane = lds + (wave * 9) * (W4_TILES * W4_KC) + l31 * 16 + ((h ^ sw) * 4);          // g = 0
    const float* const vlane1 = lds + (wave * 9) * (W4_TILES * W4_KC) + l31 * 16 + (((2 + h) ^ sw) * 4);   // g = 1
    // B: U[f][kq = 4c + 2g + h][n = co0 + 32 j + l31] (float4)
    const unsigned b_voff = (unsigned)((h * d.NcP + l31) * 16);
    const int64_t b_fstride = (int64_t)g.KQtot * d.NcP * 4;   // floats per frequency
    const int b_kq = d.NcP * 4;                                // floats per channel quad
    // bf16 x 3 form.  A: the same two 16-byte reads per frequency (the lane's tile, channel quads h and 2 + h), split in registers.
    // B: adh_pack_weights_wino43_bf16x3's [cog][chunk][36 f][NT][3 planes][64 lanes][16 B]: the 9 NT groups of a wave and chunk
    // are 27 NT KB contiguous, lane offset 16 * lane
    const unsigned b3_voff = (unsigned)lane * 16u;
    const size_t b3_cstride = (size_t)36 * NT * 3072;                                   // bytes of U per chunk and channel group

    f32x16 acc[9 * NT];
    f32x4 av[2], bv[3][NT];
    u32x4 av3[3], bv3[3][3];
    W4BNext nx;
    const float m1s = adh_opaque(-1.f);

    // ------------------------------------------------------------------ prologue of the workgroup: table and first chunk of its first region
    int cg, region = region_of(v, cg);
    if (region >= g.nregions) return;
    // (Persistent workgroups start together and do identical work, so all CUs stage, transform and contract in phase.  A one-off
    // start-up delay of 1 .. 14 us by workgroup index made no difference in either form: measured, removed.)
    Where wc = where_of(region);
    build_table(tab_base, 0, 0, true);
    if (wc.border) build_table(tab_base + TABN, wc.oy0, wc.ox0, false);
    st.rsrc = image_rsrc(image_ptr(wc.n));
    W4_STAMP(0);
    __syncthreads();   // slot tables
    {   // chunk 0 of the first region; the later chunks (and regions) arrive piece by piece inside the main loop
        const int* const tab0 = tab_base + (wc.border ? TABN : 0);
        int vo[10];   // all table reads first: one LDS round trip instead of ten
#pragma unroll
        for (int u = 0; u < 10; ++u) vo[u] = tab0[tab_lane_o + TABU * u] + (BF3 ? quad16_o : 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 10; ++u)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(st.rsrc, (lds_void_ptr4)(reinterpret_cast<char*>(lds) + st.lds_wave + u * 4096), 16,
                                                     vo[u], wc.base, 0, 0);
    }
    int cpar = 0;      // bf16 x 3 form: the raw buffer the next transform reads (toggles per chunk, across regions)
    int tsel = 0;      // which absolute table is this region's (if it is a border region)

#pragma unroll 1
    for (int it = 0;; ++it) {
#ifdef W4_PROF
        if (it) W4_STAMP(0);
        if (tid == 0 && v < 16384) {
            w4_prof_buf[v * 32 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_ID
            w4_prof_buf[v * 32 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));   // XCC_ID
        }
#endif
        const int n = wc.n, oy0 = wc.oy0, ox0 = wc.ox0;
        const int co0 = cg * 32 * NT;
        const float* const in_cur = image_ptr(n);
        int* const tab_cur = wc.border ? tab_base + (1 + tsel) * TABN : tab_base;
        // the next region of this workgroup (regions grow with v: the first invalid one ends the list)
        const int vn = v + (int)gridDim.x;
        int cgn = 0, regn = 0;
        bool has_next = vn < g.nvblocks;
        if (has_next) {
            regn = region_of(vn, cgn);
            has_next = regn < g.nregions;
        }
        Where wn = wc;
        const float* in_next = in_cur;
        int* tab_nxt = tab_cur;
        if (has_next) {   // its image and, for a border region, its table (that buffer's last reader was the previous region's main loop)
            wn = where_of(regn);
            in_next = image_ptr(wn.n);
            tab_nxt = tab_base;
            if (wn.border) {
                tab_nxt = tab_base + (1 + (tsel ^ 1)) * TABN;
                build_table(tab_nxt, wn.oy0, wn.ox0, false);
            }
        }
        W4_STAMP(4);
        const float* const b_wave = d.wp + ((int64_t)(wave * 9) * g.KQtot * d.NcP + co0) * 4;
        const char* const b3_wave = reinterpret_cast<const char*>(d.wp) + ((size_t)cg * g.nchunks * 36 + (size_t)wave * 9) * (NT * 3072);
        if constexpr (!BF3) {   // the weights of groups 0 and 1: on their way during the first transform
            w4_load_b<NT>(bv[0], b_voff, b_wave);
            w4_load_b<NT>(bv[1], b_voff, b_wave + b_fstride);
        }
        W4_STAMP(5);
#if W4_MFMA_ZERO
        w4_acc_zero<NT>(acc);
#else
#pragma unroll
        for (int t = 0; t < 9 * NT; ++t)   // (~0.9 us per region.  Tried: the first chunk's first MFMA on every tile with C = 0 instead -- a second
#pragma unroll                             // instantiation of the contraction behind `if (c == 0)`: hipcc spills ~700 registers.)
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#endif
        // first region: its halo was requested above; later regions: it landed during the previous region's last contraction and
        // was waited for (vmcnt(0)) in front of that region's epilogue barriers
        if (it == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // halo and the next region's table visible

        W4_STAMP(1);
#pragma unroll 1
        for (int c = 0; c < g.nchunks; ++c) {
            const bool more = c + 1 < g.nchunks;
            // what this chunk stages: the next chunk of this region, or -- last chunk -- the first chunk of the next region (from its
            // table and image); the very last chunk of the workgroup re-stages itself (uniform counts)
            const bool cross = !more && has_next;
            st.cb = __builtin_amdgcn_readfirstlane((more ? (c + 1) * (W4_KC * 4) : (cross ? 0 : c * (W4_KC * 4))) + (cross ? wn.base : wc.base));
            st.rsrc = image_rsrc(cross ? in_next : in_cur);
            const int* const stab = cross ? tab_nxt : tab_cur;
            st.tab_lane = stab + tab_lane_o;
            // ---- transform raw(c) -> V (every wave is past the previous chunk's contraction: barrier at the loop end)
            if (c == 1) W4_STAMP(8);
            if (c == 1) W4_STAMP(9);
            if (!(W4_DBG & 1)) transform(cpar, stab);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (c == 1) W4_STAMP(10);
            if constexpr (BF3) {   // the weights of groups 0 and 1: on their way during the barrier and the first frequency's split
                if (!(W4_DBG & 2)) {
                    w4b_load_b(bv3[0], b3_voff, b3_wave + c * b3_cstride);
                    w4b_load_b(bv3[1], b3_voff, b3_wave + c * b3_cstride + 3072);
                }
            }
            __builtin_amdgcn_s_barrier();
            if (c == 1) W4_STAMP(11);
            // ---- contraction.  In flight: the weights of groups 0 and 1 and, newer, the pieces pass 2 issued.  The raw buffer is
            // free from pass 2 to the end of the contraction (its only reader is pass 1), so the next chunk's halo lands in it
            // meanwhile; the weight waits of the later groups retire the pieces in order
            if constexpr (BF3) {
                if (!(W4_DBG & 2)) {
                    // the planes of the wave's first frequency, split while the first weights are on their way; then the values of the second
                    w4b_load_s(nx, vlane, vlane1);
                    w4b_steps<0, 20>(nx, m1s);
                    av3[0] = u32x4{nx.h[0], nx.h[1], nx.h[2], nx.h[3]};
                    av3[1] = u32x4{nx.m[0], nx.m[1], nx.m[2], nx.m[3]};
                    av3[2] = u32x4{nx.l[0], nx.l[1], nx.l[2], nx.l[3]};
                    asm volatile("s_nop 1" : "+v"(av3[0]), "+v"(av3[1]), "+v"(av3[2]));
                    w4b_load_s(nx, vlane + W4_TILES * W4_KC, vlane1 + W4_TILES * W4_KC);
                    __builtin_amdgcn_sched_barrier(0);
                    w4b_groups<NT, 0, W4_ZERO_C>(acc, av3, nx, bv3, vlane, vlane1, b3_voff, b3_wave + c * b3_cstride, m1s);
                }
            } else if (!(W4_DBG & 2)) {
                const int cn = more ? c + 1 : c;   // (the last chunk requests its own first weights again: drained below)
                const float* b_chunk = b_wave + (int64_t)(c * 4) * b_kq;
                const float* b_next = b_wave + (int64_t)(cn * 4) * b_kq;
                w4_contract<NT, W4_ZERO_C>(acc, av, bv, vlane, vlane1, b_voff, b_chunk, b_next, b_fstride, b_kq, st);
            }
            if (c == 1) W4_STAMP(12);
            cpar ^= 1;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (c == 1) W4_STAMP(13);
        }
        // everything this wave requested has landed: the weights re-requested by the last chunk and, above all, its pieces of the
        // next region's first chunk -- the epilogue's barriers make them visible to the other waves
        if constexpr (!BF3) {
            w4_wait_b<0, NT>(bv[0]);
            w4_wait_b<0, NT>(bv[1]);
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        W4_STAMP(2);

        // ---------------------------------------------------------------------- output transform A^T M A + fused epilogue
        // Six rounds: one output-channel tile (32 co) and one half of the region (16 tiles = two tile rows = accumulator registers
        // 8 th .. 8 th + 7) at a time: accumulators -> M[36][16 tiles][32 co] in LDS -- exactly V's 72 KB, so the raw halo of the next
        // region and the tables stay intact -- (tile t sits in row t' = t with its low three bits rotated so that the two lane halves,
        // tiles t and t + 4, hit different banks), then thread = (tile row t', channel quad, output-row half rh = wave >> 1): float4
        // reads of its 6 x 6 frequency patch, the two rows 2 rh, 2 rh + 1 of A^T m A, 16-byte stores.  The M write of round r + 1 is
        // issued between the column pass of round r (the last reader of M) and its row pass: the LDS drains it under the row pass's
        // VALU work and stores, and the accumulator registers it frees are dead before the row pass needs its own.
        if (!(W4_DBG & 4)) {
            float* M = lds;
            const int etp = (tid >> 3) & 15, eq = tid & 7;
            const int rh = wave >> 1;                                         // wave-uniform: rows 2 rh, 2 rh + 1 of every tile
            const int et = (etp & 8) | ((etp & 1) << 2) | ((etp >> 1) & 3);   // the tile (of its half) in row etp
            const int ey0 = oy0 + 4 * (et >> 3) + 2 * rh, ex0 = ox0 + 4 * (et & 7);    // region half 0; half 1 sits 8 rows below
            // One code path for full and ragged regions, without per-pixel address arithmetic or exec-mask branches: raw buffer
            // stores / residual loads whose address is a per-thread byte offset (constant for the region) plus a workgroup-uniform
            // scalar offset per pixel.  A pixel outside the image or a channel quad beyond Cout gets bit 31 of its vector offset set.  The hardware's range check is
            // `vector offset >= num_records - scalar offset` (unsigned), so with num_records = 0x7fffffff such an access is out
            // of range for every scalar offset (the store is dropped, the load returns zeros), while a real access, whose byte
            // offset inside the image is below 2^31 (wino43_plan), never is.  (An exact num_records would be wrong here: with a
            // scalar offset above it the subtraction wraps and nothing is checked.)
            float* out_n = d.out + (size_t)n * d.OH * d.OW * d.out_cstride;
            const float* res_n = d.residual ? d.residual + (size_t)n * d.OH * d.OW * d.res_cstride : nullptr;
            const int o_px = d.out_cstride * 4, o_row = d.OW * o_px;       // byte pitches (workgroup-uniform)
            const int r_px = d.res_cstride * 4, r_row = d.OW * r_px;
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(out_n, 0, 0x7fffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t rrsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(res_n ? res_n : out_n), 0, 0x7fffffff, 0x00020000);
            const bool ragged = oy0 + 16 > d.OH || ox0 + 32 > d.OW;     // workgroup-uniform
            const float act_lo = d.act == ADH_ACT_RELU ? 0.f : -INFINITY;   // ReLU as max(v, 0), identity as max(v, -inf)
            unsigned colpen[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) colpen[r] = ex0 + r < d.OW ? 0u : 0x80000000u;
            const unsigned o_vbase = (unsigned)((ey0 * d.OW + ex0) * o_px + (co0 + eq * 4) * 4);
            const unsigned r_vbase = (unsigned)((ey0 * d.OW + ex0) * r_px + (co0 + eq * 4) * 4);
            const unsigned m_wbase = (unsigned)((wave * 9) * 2048 + h * 128 + l31 * 4);   // byte address of M[9 wave][h][l31]
            const float* const mp = M + etp * 32 + eq * 4;
            const float m1 = adh_opaque(-1.f);
            // statistics scratch: behind the tables; bf16 x 3 form (no room left): in the raw buffer the last chunk has consumed
            // (cpar is the one the next region's first chunk has landed in)
            float* const red = BF3 ? lds + RAW0 + (cpar ^ 1) * (W4B_RAWBUF_B / 4) : lds + W4_RED;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            // (V fully consumed: barrier at the end of the last chunk)
            w4_store_mh<NT, 0>(acc, m_wbase, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int cq0 = co0 + j * 32 + eq * 4;
                const bool quad_ok = cq0 + 3 < d.Cout;             // Cout % 4 == 0 (wino43_plan): a quad is real or padding
                f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
                if (d.scale && quad_ok) sc4 = *reinterpret_cast<const f32x4*>(d.scale + cq0);
                if (d.shift && quad_ok) sh4 = *reinterpret_cast<const f32x4*>(d.shift + cq0);
                f32x4 mean4 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (BNRED) {
                    if (quad_ok) mean4 = *reinterpret_cast<const f32x4*>(g.bn_mean + cq0);
                }
                const unsigned chanpen = quad_ok ? 0u : 0x80000000u;
                const unsigned o_vj = (o_vbase + j * 128) | chanpen, r_vj = (r_vbase + j * 128) | chanpen;
                f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int th = 0; th < 2; ++th) {
                    if (j == 1 && th == 0) W4_STAMP(16);
                    // ---- column pass: rows 2 rh, 2 rh + 1 of A^T m for the six frequency columns
                    // rows of A^T: [1 1 1 1 1 0], [0 a -a b -b 0], [0 a^2 a^2 b^2 b^2 0], [0 a^3 -a^3 b^3 -b^3 1]
                    f32x4 u[2][6];
#pragma unroll
                    for (int b = 0; b < 6; ++b) {
                        f32x4 m[6];
#pragma unroll
                        for (int a = 0; a < 6; ++a) m[a] = *reinterpret_cast<const f32x4*>(mp + (a * 6 + b) * (16 * 32));
                        const f32x4 s12 = m[1] + m[2], d12 = adh_pksub(m[1], m[2], m1), s34 = m[3] + m[4], d34 = adh_pksub(m[3], m[4], m1);
                        if (rh == 0) {
                            u[0][b] = m[0] + s12 + s34;
                            u[1][b] = W4_A * d12 + W4_B * d34;
                        } else {
                            u[0][b] = W4_A2 * s12 + W4_B2 * s34;
                            u[1][b] = W4_A3 * d12 + W4_B3 * d34 + m[5];
                        }
                    }
                    if (j == 1 && th == 0) W4_STAMP(17);
                    __builtin_amdgcn_s_barrier();   // M fully consumed
                    if (j == 1 && th == 0) W4_STAMP(18);
                    // ---- the next round's accumulators go into M during the row pass: frequency 0 now, one more behind every pixel
                    const int jn = th == 0 ? j : j + 1, thn = th ^ 1;   // the next round (jn == NT: none)
                    w4_store_mh_round<NT>(acc, m_wbase, jn, thn, 0);
                    if (j == 1 && th == 0) W4_STAMP(19);
                    // ---- row pass, fused epilogue, stores
                    unsigned rowpen[2];
#pragma unroll
                    for (int r = 0; r < 2; ++r) rowpen[r] = ey0 + 8 * th + r < d.OH ? 0u : 0x80000000u;
                    f32x4 rres[8];   // the residual of this thread's 2 x 4 pixels
                    if (res_n) {   // workgroup-uniform; the asm keeps it a branch (the zero-initialised alternative is free)
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj)
                                rres[ii * 4 + jj] = __builtin_bit_cast(
                                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrsrc, r_vj | rowpen[ii] | colpen[jj],
                                                                                 (8 * th + ii) * r_row + jj * r_px, 0));
                        asm volatile("" ::: "memory");
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q) rres[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int ii = 0; ii < 2; ++ii) {
                        const f32x4 s12 = u[ii][1] + u[ii][2], d12 = adh_pksub(u[ii][1], u[ii][2], m1), s34 = u[ii][3] + u[ii][4],
                                    d34 = adh_pksub(u[ii][3], u[ii][4], m1);
                        const f32x4 y[4] = {u[ii][0] + s12 + s34, W4_A * d12 + W4_B * d34, W4_A2 * s12 + W4_B2 * s34,
                                            W4_A3 * d12 + W4_B3 * d34 + u[ii][5]};
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const unsigned oaddr = o_vj | rowpen[ii] | colpen[jj];
                            if constexpr (BNRED) {
                                // data-gradient launch that also takes the PRODUCING layer's BatchNorm-backward sums:
                                // the output is that layer's output gradient g; `residual` is its raw convolution output y, scale /
                                // shift its forward BN scale / shift, so m = [fma(y, scale, shift) > 0] is its ReLU mask (the forward
                                // expression of bn_apply_kernel, bit for bit); statistics rows = sum g m, sum g m (y - mean).  g is stored as is.
                                const f32x4 yv = rres[ii * 4 + jj];
                                // a pixel outside the image / a padding channel quad has bit 31 of its store offset set: it does not count
                                // (no per-pixel float masks: eight more live registers spill 80 here)
                                const bool inside = (int)oaddr >= 0;
                                f32x4 gm;
#pragma unroll
                                for (int e = 0; e < 4; ++e) gm[e] = (inside && __builtin_fmaf(yv[e], sc4[e], sh4[e]) > 0.f) ? y[jj][e] : 0.f;
                                ssum += gm;
                                ssq += gm * adh_pksub(yv, mean4, m1);
                                f32x4 v4 = y[jj];
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), orsrc, oaddr,
                                                                       (8 * th + ii) * o_row + jj * o_px, W_STORE_AUX);
                                asm volatile("s_nop %1" : "+v"(v4) : "n"(W4_STORE_NOPS) : "memory");
                                w4_store_mh_round<NT>(acc, m_wbase, jn, thn, 1 + ii * 4 + jj);
                                continue;
                            }
                            f32x4 v4 = y[jj] * sc4 + sh4;
                            if (d.stats) {
                                f32x4 vs = v4;
                                if (ragged) {   // pixels outside the image do not count (a real, workgroup-uniform branch: the asm keeps
                                    // the compiler from turning it into selects); bit 31 of the store offset marks them
                                    const bool inside = (int)oaddr >= 0;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) vs[e] = inside ? v4[e] : 0.f;
                                    asm volatile("" : "+v"(vs));
                                }
                                ssum += vs;
                                ssq += vs * vs;
                            }
                            v4 += rres[ii * 4 + jj];      // zeros without a residual
                            v4 = {fmaxf(v4[0], act_lo), fmaxf(v4[1], act_lo), fmaxf(v4[2], act_lo), fmaxf(v4[3], act_lo)};
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), orsrc, oaddr, (8 * th + ii) * o_row + jj * o_px,
                                                                   W_STORE_AUX);
                            // gfx950: a 128-bit buffer store whose data VGPRs are overwritten by the very next instructions stores
                            // the NEW values in some lanes (lanes 12-15 of every 16) even when its soffset is an SGPR (measured:
                            // tools/dev_w43_probe.py, tools/dev_w32_probe.py; LLVM pads only the immediate-soffset form,
                            // GCNHazardRecognizer "12-dword store hazard", and schedules such an overwrite right behind the store).
                            // The asm reads the data registers, so they stay untouched until its wait states have passed.
                            asm volatile("s_nop %1" : "+v"(v4) : "n"(W4_STORE_NOPS) : "memory");
                            w4_store_mh_round<NT>(acc, m_wbase, jn, thn, 1 + ii * 4 + jj);
                        }
                    }
                    if (j == 1 && th == 0) W4_STAMP(20);
                    if (!(j == NT - 1 && th == 1)) {   // the next round's M is complete
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                    }
                    if (j == 1 && th == 0) W4_STAMP(21);
                }
                if (d.stats) {
                    // sum over the 8 tile rows of this wave (lane bits 3..5; both region halves are already in), then over the 4 waves
                    // (two tile groups x two output-row halves) through LDS
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int o = 8; o < 64; o <<= 1) {
                            ssum[e] += __shfl_xor(ssum[e], o, 64);
                            ssq[e] += __shfl_xor(ssq[e], o, 64);
                        }
                    }
                    if (j) __builtin_amdgcn_s_barrier();   // the previous tile's rows have been read
                    if (lane < 8) {
                        *reinterpret_cast<f32x4*>(red + (0 * 4 + wave) * 32 + eq * 4) = ssum;
                        *reinterpret_cast<f32x4*>(red + (1 * 4 + wave) * 32 + eq * 4) = ssq;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (tid < 64) {
                        const int which = tid >> 5, cl = tid & 31;
                        float sv = 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) sv += red[(which * 4 + r) * 32 + cl];
                        d.stats[((size_t)region * 2 + which) * d.NcP + co0 + j * 32 + cl] = sv;
                    }
                }
            }
        }
        W4_STAMP(3);
        if (!has_next) break;
        // (the last round's M is V, which the next region's first transform writes: behind the barrier at the top of the region)
        v = vn;
        region = regn;
        cg = cgn;
        wc = wn;
        tsel ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int wino43_plan(const adh_conv_desc* d, Wino43Geom* g) {
    static const bool enabled = !(getenv("ADH_WINO43") && getenv("ADH_WINO43")[0] == '0');   // A/B switch
    if (!enabled || !d) return 0;
    if (d->KH != 3 || d->KW != 3 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1) return 0;
    if (d->out_oy != 0 || d->out_ox != 0 || d->dy0 != -1 || d->dx0 != -1 || d->dstep_y != 1 || d->dstep_x != 1) return 0;
    if (d->Cin % W4_KC != 0 || d->in_cstride % 4 != 0 || d->NcP % 32 != 0) return 0;
    if (d->VH != d->OH || d->VW != d->OW || d->IH != d->OH || d->IW != d->OW) return 0;
    if ((int64_t)(d->IH + 2) * d->IW * d->in_cstride >= (1ll << 29)) return 0;
    // the epilogue stores 16-byte channel quads through a buffer descriptor spanning one image (conv_wino.hip takes the rest)
    if (d->Cout % 4 != 0 || d->out_cstride % 4 != 0 || ((uintptr_t)d->out & 15) || (int64_t)d->OH * d->OW * d->out_cstride >= (1ll << 29))
        return 0;
    if (d->residual && (d->res_cstride % 4 != 0 || ((uintptr_t)d->residual & 15) ||
                        (int64_t)d->OH * d->OW * d->res_cstride >= (1ll << 29)))
        return 0;
    if ((d->scale && ((uintptr_t)d->scale & 15)) || (d->shift && ((uintptr_t)d->shift & 15))) return 0;
    g->tiles_x = adh_ceil_div(d->OW, 32);
    g->tiles_y = adh_ceil_div(d->OH, 16);
    g->nregions = g->tiles_x * g->tiles_y * d->N;
    g->nchunks = d->Cin / W4_KC;
    g->KQtot = d->Cin / 4;
    g->bn_mean = nullptr;
    return 1;
}

extern "C" int adh_conv_wino43_supported(const adh_conv_desc* d) {
    Wino43Geom g;
    return wino43_plan(d, &g);
}

extern "C" int adh_conv_wino43_num_blocks(const adh_conv_desc* d) {
    Wino43Geom g;
    if (!wino43_plan(d, &g)) return ADH_E_UNSUPPORTED;
    return g.nregions;
}

// Persistent launch: one workgroup per CU (512 registers per lane and ~140 KB of LDS allow no second one anyway), a multiple of
// 8 so that v & 7 stays the XCD of every virtual block a workgroup runs.  The shares are STATIC (workgroup w runs blocks w, w + G, ..):
// right when the launch has the chip to itself, wrong when another stream's kernel holds CUs for long -- a workgroup that starts
// late still owns 1 / G of the work and the whole launch waits for it.  The data-parallel step overlaps its gradient all-reduces
// with the backward pass, so parallel.GradientSynchronizer switches to one workgroup per virtual block
// (adh_conv_wino43_set_persistent(0)); so does ADH_WINO43_GRID=0.  (A dynamic hand-out -- per-XCD atomic counters, the block after
// next requested at the top of the epilogue -- was built and measured: hipcc's atomicAdd waits for its result on the spot, ~0.7 us
// per region = +1 % at 96 channels; an asm atomic left in flight across the epilogue is exactly the register-copy hazard of
// DESIGN 4.15.  Not adopted.)
static int w4_persistent = 1;
extern "C" int adh_conv_wino43_set_persistent(int on) {
    const int old = w4_persistent;
    w4_persistent = on ? 1 : 0;
    return old;
}
static int wino43_grid(int nvblocks) {
    if (!w4_persistent) return nvblocks;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount >= 8)
            ncu = p.multiProcessorCount / 8 * 8;
        else
            ncu = 256;
        const char* e = getenv("ADH_WINO43_GRID");   // dev: 0 = one workgroup per virtual block (no persistence)
        if (e && atoi(e) >= 0) ncu = atoi(e) / 8 * 8;
    }
    return (ncu && nvblocks > ncu) ? ncu : nvblocks;
}

template <int NT, bool BNRED = false, bool BF3 = false>
static int launch_wino43(hipStream_t s, const adh_conv_desc* d, Wino43Geom g) {
    g.ncog = d->NcP / (32 * NT);
    g.nvblocks = ((g.nregions + 7) / 8) * g.ncog * 8;
    // reciprocals for the region decode: n / d == mulhi(n, ceil(2^32 / d)) whenever n d < 2^32 (n <= nvblocks here); d = 1 and
    // oversized problems divide
    auto recip = [&](int dv) {
        return (dv > 1 && (int64_t)g.nvblocks * dv < (1ll << 32)) ? (unsigned)(((1ull << 32) + (unsigned)dv - 1) / (unsigned)dv) : 0u;
    };
    g.m_ncog = recip(g.ncog);
    g.m_tx = recip(g.tiles_x);
    g.m_ty = recip(g.tiles_y);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino43_kernel<NT, BNRED, BF3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv_wino43_kernel<NT, BNRED, BF3>), dim3(wino43_grid(g.nvblocks)), dim3(256), BF3 ? W4B_LDS_BYTES : W4_LDS_BYTES, s,
                       *d, g);
    return adh_check_launch();
}

static int wino43_check_args(const adh_conv_desc* d) {
    if (!d->in || !d->out || !d->wp || d->NcP < d->Cout) return ADH_E_ARG;
    if (d->out_cstride < d->Cout || (d->residual && d->res_cstride < d->Cout)) return ADH_E_ARG;
    if (((uintptr_t)d->in & 15) || ((uintptr_t)d->wp & 15)) return ADH_E_ARG;
    return ADH_OK;
}

extern "C" int adh_conv_wino43_forward(void* stream, const adh_conv_desc* d) {
    Wino43Geom g;
    if (!wino43_plan(d, &g)) return ADH_E_UNSUPPORTED;
    const int rc = wino43_check_args(d);
    if (rc) return rc;
    const int nt = d->NcP / 32;
    hipStream_t s = (hipStream_t)stream;
    if (nt % 3 == 0) return launch_wino43<3>(s, d, g);
    if (nt % 2 == 0) return launch_wino43<2>(s, d, g);
    return launch_wino43<1>(s, d, g);
}

// Data gradient + the producing layer's BatchNorm-backward sums in one launch (see the epilogue): d describes the data-gradient
// convolution; d->residual / res_cstride = the producer's raw convolution output y, d->scale / d->shift = its forward BN
// scale / shift, bn_mean its batch mean, d->stats = [adh_conv_wino43_num_blocks(d)][2][NcP] rows of (sum g m, sum g m (y - mean));
// d->act must be NONE.  adh_bn_bwd_finalize_centered turns the rows into d-gamma / d-beta / the coefficients of adh_bn_bwd_apply.
extern "C" int adh_conv_wino43_dgrad_bnred(void* stream, const adh_conv_desc* d, const float* bn_mean) {
    Wino43Geom g;
    if (!wino43_plan(d, &g)) return ADH_E_UNSUPPORTED;
    const int rc = wino43_check_args(d);
    if (rc) return rc;
    if (!d->residual || !d->scale || !d->shift || !d->stats || !bn_mean || d->act != ADH_ACT_NONE) return ADH_E_ARG;
    if ((uintptr_t)bn_mean & 15) return ADH_E_ARG;
    g.bn_mean = bn_mean;
    const int nt = d->NcP / 32;
    hipStream_t s = (hipStream_t)stream;
    if (nt % 3 == 0) return launch_wino43<3, true>(s, d, g);
    if (nt % 2 == 0) return launch_wino43<2, true>(s, d, g);
    return launch_wino43<1, true>(s, d, g);
}

// The same two launches with the contraction on v_mfma_f32_32x32x16_bf16 over exact three-plane bf16 splits of both operands
// (opt-in, ADH_CONTRACT=bf16x3 on the host side; d->wp from adh_pack_weights_wino43_bf16x3).  Same arguments, same results to
// fp32 rounding: the split is exact and the three dropped cross terms are at 2^-24 of a product.
extern "C" int adh_conv_wino43_forward_bf16x3(void* stream, const adh_conv_desc* d) {
    Wino43Geom g;
    if (!wino43_plan(d, &g)) return ADH_E_UNSUPPORTED;
    const int rc = wino43_check_args(d);
    if (rc) return rc;
    const int nt = d->NcP / 32;
    hipStream_t s = (hipStream_t)stream;
    if (nt % 3 == 0) return launch_wino43<3, false, true>(s, d, g);
    if (nt % 2 == 0) return launch_wino43<2, false, true>(s, d, g);
    return launch_wino43<1, false, true>(s, d, g);
}

extern "C" int adh_conv_wino43_dgrad_bnred_bf16x3(void* stream, const adh_conv_desc* d, const float* bn_mean) {
    Wino43Geom g;
    if (!wino43_plan(d, &g)) return ADH_E_UNSUPPORTED;
    const int rc = wino43_check_args(d);
    if (rc) return rc;
    if (!d->residual || !d->scale || !d->shift || !d->stats || !bn_mean || d->act != ADH_ACT_NONE) return ADH_E_ARG;
    if ((uintptr_t)bn_mean & 15) return ADH_E_ARG;
    g.bn_mean = bn_mean;
    const int nt = d->NcP / 32;
    hipStream_t s = (hipStream_t)stream;
    if (nt % 3 == 0) return launch_wino43<3, true, true>(s, d, g);
    if (nt % 2 == 0) return launch_wino43<2, true, true>(s, d, g);
    return launch_wino43<1, true, true>(s, d, g);
}

// ------------------------------------------------------------------------------------------------ weight packs
// U[f = a*6+b][k/4][n][4] = (G g G^T)[a][b];  row of G for point p: [1, p, p^2] / prod_{q != p} (p - q), for inf: [0, 0, 1]
// (computed in double: the entries of G are not dyadic).
// Both pack kernels: workgroup = 16 k x 32 n of the filter bank, staged through LDS with coalesced loads (one thread per (k, n) reading
// its nine taps straight from the OIHW / IOHW tensor touched 64 cache lines per load instruction: 105 us per 384 x 384 layer in the
// bf16 x 3 form, 4.3 ms per training step over the 41 packs; round 4), thread = (n, frequency row fa, ...): six frequencies each.
#define W4_PACK_K 16
#define W4_PACK_N 32
struct W4PackG { double G[6][3]; };
__device__ __forceinline__ W4PackG w4_pack_g() {
    const double a = W4_A, b = W4_B;
    const double n0 = a * a * b * b, na = 2.0 * a * a * (a * a - b * b), nb = 2.0 * b * b * (b * b - a * a);
    return W4PackG{{{1.0 / n0, 0.0, 0.0},      {1.0 / na, a / na, a * a / na}, {1.0 / na, -a / na, a * a / na},
                    {1.0 / nb, b / nb, b * b / nb}, {1.0 / nb, -b / nb, b * b / nb}, {0.0, 0.0, 1.0}}};
}
// g[k][n][m] = the nine taps of filter (k0 + k, n0 + n) in MEMORY order (m = 0 is the lowest address); tap (p, q) sits at
// m = m0 + p * sy + q * sx.  The taps of one (k, n) pair are nine consecutive floats whenever |sy| = 3 and |sx| = 1 (every 3x3 layer
// here, forward and flipped); the (k, n) pairs are walked along whichever of the two strides is 9, so that a wave reads long runs.
template <int TK>
__device__ __forceinline__ int w4_pack_load_tile(const float* __restrict__ src, const adh_wlayout& L, int k0, int n0,
                                                 float (*g)[W4_PACK_N][9]) {
    const int sy = L.tap_off_sy, sx = L.tap_off_sx;
    const int tapmin = L.tap_off0 + (sy < 0 ? 2 * sy : 0) + (sx < 0 ? 2 * sx : 0);
    const bool k_inner = L.stride_k == 9;
    for (int e = threadIdx.x; e < TK * W4_PACK_N * 9; e += blockDim.x) {
        const int t = e % 9, pr = e / 9;
        const int k = k_inner ? pr % TK : pr / W4_PACK_N, n = k_inner ? pr / TK : pr % W4_PACK_N;
        float v = 0.f;
        if (k0 + k < L.K && n0 + n < L.Nc) v = src[(int64_t)tapmin + t + (int64_t)(k0 + k) * L.stride_k + (int64_t)(n0 + n) * L.stride_n];
        g[k][n][t] = v;
    }
    __syncthreads();
    return L.tap_off0 - tapmin;
}
__device__ __forceinline__ void w4_pack_row(double (&ga)[3], const W4PackG& G, int fa) {   // row fa of G without a dynamically indexed private array
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        ga[p] = G.G[0][p];
#pragma unroll
        for (int r = 1; r < 6; ++r) ga[p] = fa == r ? G.G[r][p] : ga[p];
    }
}

// grid = (KQ / 2, NcP / 32), 384 threads = 32 n x 6 frequency rows x 2 channel quads (KQ is even: K rounded up to 8)
__global__ __launch_bounds__(384) void pack_weights_wino43_kernel(const float* __restrict__ src, const adh_wlayout L, int KQ, int NcP,
                                                                  f32x4* __restrict__ wp) {
    __shared__ float g[8][W4_PACK_N][9];
    const int k0 = blockIdx.x * 8, n0 = blockIdx.y * W4_PACK_N;
    const int m0 = w4_pack_load_tile<8>(src, L, k0, n0, g);
    const int nl = threadIdx.x & 31, rest = threadIdx.x >> 5, fa = rest % 6, kql = rest / 6;
    const W4PackG G = w4_pack_g();
    double ga[3];
    w4_pack_row(ga, G, fa);
    const int n = n0 + nl, kq = blockIdx.x * 2 + kql;
#pragma unroll
    for (int fb = 0; fb < 6; ++fb) {
        f32x4 u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double acc = 0.0;
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q2 = 0; q2 < 3; ++q2)
                    acc += ga[p] * G.G[fb][q2] * (double)g[kql * 4 + j][nl][m0 + p * L.tap_off_sy + q2 * L.tap_off_sx];
            u[j] = (float)acc;
        }
        wp[((int64_t)(fa * 6 + fb) * KQ + kq) * NcP + n] = u;
    }
}

static bool w4_pack_layout_ok(const adh_wlayout* L) {
    return (L->tap_off_sy == 3 || L->tap_off_sy == -3) && (L->tap_off_sx == 1 || L->tap_off_sx == -1);
}

extern "C" int adh_pack_weights_wino43(void* stream, const float* src, const adh_wlayout* L, float* wp) {
    if (!src || !L || !wp || L->K < 1 || L->Nc < 1 || L->KHt != 3 || L->KWt != 3) return ADH_E_ARG;
    if (!w4_pack_layout_ok(L)) return ADH_E_UNSUPPORTED;   // taps of one filter must be nine consecutive floats (any 3x3 OIHW / IOHW tensor, flipped or not)
    const int KQ = adh_round_up(L->K, 8) / 4;
    const int NcP = adh_round_up(L->Nc, 32);
    hipLaunchKernelGGL(pack_weights_wino43_kernel, dim3(KQ / 2, NcP / 32), dim3(384), 0, (hipStream_t)stream, src, *L, KQ, NcP,
                       reinterpret_cast<f32x4*>(wp));
    return adh_check_launch();
}

// The same U = G g G^T (computed in double, rounded to fp32 exactly as above), split into three bf16 planes and laid out for
// conv_wino43_kernel<NT, *, true>: [channel group of 32 NT][chunk of 16 k][36 f][NT tiles][3 planes][half h][32 n][8 k] bf16 (the
// eight k of half h: channels 4h .. 4h+3 and 8+4h .. 8+4h+3 of the chunk), NT as the launch picks it from NcP.  wp: 36 * Kp * NcP * 6 bytes (Kp = K rounded up to 16, NcP = Nc rounded up to 32).
// grid = (chunks, NcP / 32), 384 threads = 32 n x 6 frequency rows x 2 lane halves
__global__ __launch_bounds__(384) void pack_weights_wino43_bf16x3_kernel(const float* __restrict__ src, const adh_wlayout L, int nchunks, int NcP,
                                                                         int NT, u32x4* __restrict__ wp) {
    __shared__ float g[W4_PACK_K][W4_PACK_N][9];
    const int chunk = blockIdx.x, n0 = blockIdx.y * W4_PACK_N;
    const int m0 = w4_pack_load_tile<W4_PACK_K>(src, L, chunk * 16, n0, g);
    const int l31 = threadIdx.x & 31, rest = threadIdx.x >> 5, fa = rest % 6, hh = rest / 6;
    const W4PackG G = w4_pack_g();
    double ga[3];
    w4_pack_row(ga, G, fa);
    const int nt = n0 >> 5, cog = nt / NT, j = nt % NT;
#pragma unroll
    for (int fb = 0; fb < 6; ++fb) {
        unsigned pl[3][4];
#pragma unroll
        for (int i2 = 0; i2 < 4; ++i2) {
            float u2[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                // k-slot i of lane half hh = the kernel's A operand: channel quads hh and 2 + hh of the chunk
                const int i = 2 * i2 + e, kl = i < 4 ? 4 * hh + i : 8 + 4 * hh + (i - 4);
                double acc = 0.0;
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int q2 = 0; q2 < 3; ++q2) acc += ga[p] * G.G[fb][q2] * (double)g[kl][l31][m0 + p * L.tap_off_sy + q2 * L.tap_off_sx];
                u2[e] = (float)acc;
            }
            const unsigned hi = w4b_cvt_pk(u2[0], u2[1]);
            const float r0 = u2[0] - __builtin_bit_cast(float, hi << 16), r1 = u2[1] - __builtin_bit_cast(float, hi & 0xffff0000u);
            const unsigned mid = w4b_cvt_pk(r0, r1);
            const float s0 = r0 - __builtin_bit_cast(float, mid << 16), s1 = r1 - __builtin_bit_cast(float, mid & 0xffff0000u);
            pl[0][i2] = hi;
            pl[1][i2] = mid;
            pl[2][i2] = w4b_cvt_pk(s0, s1);
        }
        const int64_t grp = (((int64_t)cog * nchunks + chunk) * 36 + fa * 6 + fb) * NT + j;
#pragma unroll
        for (int p = 0; p < 3; ++p) wp[(grp * 3 + p) * 64 + hh * 32 + l31] = u32x4{pl[p][0], pl[p][1], pl[p][2], pl[p][3]};
    }
}

extern "C" int adh_pack_weights_wino43_bf16x3(void* stream, const float* src, const adh_wlayout* L, void* wp) {
    if (!src || !L || !wp || L->K < 1 || L->Nc < 1 || L->KHt != 3 || L->KWt != 3) return ADH_E_ARG;
    if (!w4_pack_layout_ok(L)) return ADH_E_UNSUPPORTED;
    const int nchunks = adh_round_up(L->K, 16) / 16;
    const int NcP = adh_round_up(L->Nc, 32);
    const int nt = NcP / 32, NT = nt % 3 == 0 ? 3 : (nt % 2 == 0 ? 2 : 1);
    hipLaunchKernelGGL(pack_weights_wino43_bf16x3_kernel, dim3(nchunks, NcP / 32), dim3(384), 0, (hipStream_t)stream, src, *L, nchunks, NcP, NT,
                       reinterpret_cast<u32x4*>(wp));
    return adh_check_launch();
}
