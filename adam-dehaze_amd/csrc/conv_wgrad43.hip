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
// Round 3 redesign.  The round-1 kernel gave a workgroup 96 x 96 channels x ONE THIRD of the frequencies: every tile
// crossed L2 -> LDS three times (6.9 MAC per staged byte: 41 GB/s per CU at the MFMA rate, more than the path delivers),
// its raw tiles were single buffered and its two-pass transform wrote every value to LDS twice at ~80 B/clk.  Now:
//   * workgroup = 32 input channels x 96 output channels x ALL 36 frequencies: 11.7 MAC per staged byte (24 GB/s per CU);
//     wave w owns frequencies 9w .. 9w+8 x 3 output-channel tiles = 27 accumulator tiles (16 pinned to AGPRs) -- the
//     geometry of conv_wino43_kernel<3>'s contraction;
//   * strips of 4 tiles (4 rows x 16 pixels of dL/dy, 6 x 18 halo of x; K = 4 tiles = two MFMA k-steps);
//   * ONE transform pass, in registers: thread = (tile, channel quad, frequency row alpha): it applies the row transform
//     for its alpha while it reads the raw columns (3-4 ds_read_b128 per column), then the column transform, and writes the
//     six V values of its row: every transformed value is written to LDS once (ds_write_b128 runs at ~80 B/clk per CU
//     -- the LDS write path, not the VALU, bounded the old two-pass form).  Lanes pair up rows that share their code:
//     {1,2} and {3,4} differ in one constant's sign, {0,5} in the raw rows they read;
//   * raw tiles arrive by LDS-DMA issued INSIDE the contraction (one piece behind the first MFMA of steps 0..10, as in
//     conv_wino43_kernel): the raw buffers are free from the moment the transform has finished, the pieces land under
//     the remaining contraction steps.  No global operand loads in the loop: all waits are compiler-visible;
//   * conflict-free LDS images: x pixel slots of 128 B with a gap slot after every 4th pixel (tile pitch 5 slots = 160
//     banks = 32 mod 64), dL/dy pixels of 384 B with the channel quads of odd tiles rotated by 8 (again 32 banks), so
//     that the 16 lanes of a ds_read_b128 group -- two tiles x 8 quads -- hit 64 different banks.
// Strips at the image border (and the ragged last strip of a row) zero the raw buffers and load with per-lane masks.
// Output: slab[split][36][KP][NcP] partial sums; adh_wgrad_reduce_wino43 sums the splits and applies A'^T (.) A'.
#include "common.h"
#include <cstdlib>
#include <type_traits>

#ifndef G4_DBG
#define G4_DBG 0   // dev builds: 1 = skip the transform, 4 = skip the contraction, 8 = stage only the first strip,
                   // 16 = do not wait for the pieces (wrong results), 32 = pieces in one burst before the contraction.
                   // (The round-3 issue-cost probes -- 64 / 128 / 256 / 512: every strip staging the same pixels, degenerate per-lane
                   // address patterns, one active lane per piece -- are gone from the product source: they rewrote addresses that the
                   // border path takes relative to a clamped origin, and one of them faulted on the GPU box when applied there.
                   // What they measured is in profiles/r03_ablate_wgrad43_dma.txt.)
#endif
static_assert((G4_DBG & ~(1 | 4 | 8 | 16 | 32)) == 0, "unknown G4_DBG bit: the address-rewriting probes were removed (see above)");
#ifndef G4_STAGE_REGS
#define G4_STAGE_REGS 0   // 1: raw tiles through registers (buffer_load_dwordx4 in steps 0..10, ds_write_b128 seven steps later) instead of LDS-DMA
#endif
#ifndef G4_GLOBAL_DMA
#define G4_GLOBAL_DMA 0   // 1: global_load_lds_dwordx4 pieces: 52 vs 107 cycles of issue per piece in isolation (tools/micro/dma_cost.hip), no difference in this kernel (2.81 vs 2.81 ms); 0: buffer_load ... lds (range-checked)
#endif
#ifndef G4_OPS_AFTER_HOOK
#define G4_OPS_AFTER_HOOK 0
#endif
#define G4_T 4                                    // tiles per strip
#define G4_XROW 768                               // floats per raw x row: 24 slots x 32 channels (slot = col + col / 4)
#define G4_RAWX_F (6 * G4_XROW)                   //  4,608 floats
#define G4_GROW (16 * 96)                         // floats per raw dL/dy row
#define G4_RAWG_F (4 * G4_GROW)                   //  6,144 floats
#define G4_VPLANE (G4_T * 128)                    // floats per frequency plane: [tile][32 ci | 96 co]
#define G4_V_F (36 * G4_VPLANE)                   // 18,432 floats
#define G4_TAB_F (6 * 64)                         // per-lane global offsets of the six LDS-DMA piece patterns
#define G4_LDS_BYTES ((G4_RAWX_F + G4_RAWG_F + G4_V_F + G4_TAB_F) * 4)     // 118,272 B

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
    int ngroups, ncob;     // groups = (Cin / 32) x (Cout / 96)
    int KP;
};

template <int IDX>
__device__ __forceinline__ void g4_mfma(f32x16& c, float a, float b) {
    if constexpr (IDX < 16) asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// step ST = fi * 2 + ks (local frequency fi of 9, tile pair ks of 2): operands of step ST + 1 are read before the three MFMAs
// of step ST are issued; hook(ST) (one LDS-DMA piece of the next strip) sits behind the first MFMA
// volatile: one ds_read_b32 per operand with a 16-bit immediate offset from ONE base register (hipcc otherwise pairs them
// into ds_read2_b32, whose 8-bit offsets need a v_add_u32 on the base every step: VALU that comes out of the MFMA time)
typedef const volatile __attribute__((address_space(3))) float* g4_lds_vf;
template <int ST>
__device__ __forceinline__ void g4_load_ops(g4_lds_vf vlane, float& a, float (&b)[3]) {
    g4_lds_vf p = vlane + (ST / 2) * G4_VPLANE + (ST % 2) * 256;
    a = p[0];
#pragma unroll
    for (int j = 0; j < 3; ++j) b[j] = p[32 + j * 32];
}
template <int ST, typename Hook>
__device__ __forceinline__ void g4_contract(f32x16 (&acc)[27], g4_lds_vf vlane, float (&a)[2], float (&b)[2][3], Hook&& hook) {
    if constexpr (ST < 18) {
#if G4_OPS_AFTER_HOOK
        g4_mfma<(ST / 2) * 3 + 0>(acc[(ST / 2) * 3 + 0], a[ST & 1], b[ST & 1][0]);
        hook(std::integral_constant<int, ST>{});
        if constexpr (ST + 1 < 18) g4_load_ops<ST + 1>(vlane, a[(ST + 1) & 1], b[(ST + 1) & 1]);
#else
        if constexpr (ST + 1 < 18) g4_load_ops<ST + 1>(vlane, a[(ST + 1) & 1], b[(ST + 1) & 1]);
        g4_mfma<(ST / 2) * 3 + 0>(acc[(ST / 2) * 3 + 0], a[ST & 1], b[ST & 1][0]);
        hook(std::integral_constant<int, ST>{});
#endif
        g4_mfma<(ST / 2) * 3 + 1>(acc[(ST / 2) * 3 + 1], a[ST & 1], b[ST & 1][1]);
        g4_mfma<(ST / 2) * 3 + 2>(acc[(ST / 2) * 3 + 2], a[ST & 1], b[ST & 1][2]);
        g4_contract<ST + 1>(acc, vlane, a, b, hook);
    }
}

struct G4Neg { float s2, b2, a2, a, b; };   // negative constants in opaque SGPRs (see conv_wino43.hip, W4Neg)
__device__ __forceinline__ G4Neg g4_neg_constants() {
    G4Neg n = {-G4_S2, -G4_B2, -G4_A2, -G4_A, -G4_B};
    asm volatile("" : "+s"(n.s2), "+s"(n.b2), "+s"(n.a2), "+s"(n.a), "+s"(n.b));
    return n;
}

// ---------------------------------------------------------------------------------------------------- transform rounds
// A round = 64 units (8 channel quads x 2 rows of a frequency-row pair x 4 tiles) of one operand: `begin` issues the loads
// of raw column 0, `rows` applies the row transform column by column with the next column's loads in flight, `finish`
// applies the column transform and writes the six V values.  The caller interleaves the rounds of a wave so that the
// first loads of round i + 1 are in flight during finish(i) (one wave per SIMD: nothing else hides the LDS latency).
template <int PAIR>
struct G4X {
    static constexpr int NL = PAIR == 0 ? 3 : 4;
    f32x4 L[2][4];
    f32x4 T[6];
    const float* src;
    float* dst;
    float kp;
    __device__ __forceinline__ void load(int c, f32x4 (&l)[4]) {
        const float* s = src + (c + c / 4) * 32;
        if constexpr (PAIR == 0) {
            l[0] = *reinterpret_cast<const f32x4*>(s);
            l[1] = *reinterpret_cast<const f32x4*>(s + 2 * G4_XROW);
            l[2] = *reinterpret_cast<const f32x4*>(s + 4 * G4_XROW);
        } else {
            l[0] = *reinterpret_cast<const f32x4*>(s + G4_XROW);
            l[1] = *reinterpret_cast<const f32x4*>(s + 2 * G4_XROW);
            l[2] = *reinterpret_cast<const f32x4*>(s + 3 * G4_XROW);
            l[3] = *reinterpret_cast<const f32x4*>(s + 4 * G4_XROW);
        }
    }
    __device__ __forceinline__ void begin(const float* rawx, float* V, int t, int q8, int pb, const G4Neg ng) {
        if constexpr (PAIR == 0) {
            src = rawx + t * 160 + q8 * 4 + pb * G4_XROW;          // alpha 0: rows 0,2,4; alpha 5: rows 1,3,5
            dst = V + pb * 5 * 6 * G4_VPLANE + t * 128 + q8 * 4;
            kp = 0.f;
        } else {
            src = rawx + t * 160 + q8 * 4;
            dst = V + (2 * PAIR - 1 + pb) * 6 * G4_VPLANE + t * 128 + q8 * 4;
            kp = PAIR == 1 ? (pb ? ng.a : G4_A) : (pb ? ng.b : G4_B);
        }
        load(0, L[0]);
    }
    __device__ __forceinline__ void rows(const G4Neg ng) {
        const float kxn = PAIR == 1 ? ng.b2 : ng.a2;                // rows 1,2: -b^2; rows 3,4: -a^2
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            if (c + 1 < 6) load(c + 1, L[(c + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4(&l)[4] = L[c & 1];
            if constexpr (PAIR == 0) T[c] = G4_A2B2 * l[0] + (ng.s2 * l[1] + l[2]);
            else {
                const f32x4 p = kxn * l[1] + l[3], qq = kxn * l[0] + l[2];
                T[c] = kp * qq + p;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __device__ __forceinline__ void finish(const G4Neg ng) {
        *reinterpret_cast<f32x4*>(dst) = G4_A2B2 * T[0] + (ng.s2 * T[2] + T[4]);
        *reinterpret_cast<f32x4*>(dst + 5 * G4_VPLANE) = G4_A2B2 * T[1] + (ng.s2 * T[3] + T[5]);
        {
            const f32x4 p = ng.b2 * T[2] + T[4], qq = ng.b2 * T[1] + T[3];
            *reinterpret_cast<f32x4*>(dst + 1 * G4_VPLANE) = G4_A * qq + p;
            *reinterpret_cast<f32x4*>(dst + 2 * G4_VPLANE) = ng.a * qq + p;
        }
        {
            const f32x4 r = ng.a2 * T[2] + T[4], s = ng.a2 * T[1] + T[3];
            *reinterpret_cast<f32x4*>(dst + 3 * G4_VPLANE) = G4_B * s + r;
            *reinterpret_cast<f32x4*>(dst + 4 * G4_VPLANE) = ng.b * s + r;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
};
// dL/dy rows of the pair PAIR for output-channel tile J: G' g G'^T with the un-normalised rows [1, p, p^2, p^3]
template <int PAIR, int J>
struct G4G {
    f32x4 L[2][4];
    f32x4 T[4];
    const float* src;
    float* dst;
    float kp;
    __device__ __forceinline__ void load(int c, f32x4 (&l)[4]) {
        const float* s = src + c * 96;
        l[0] = *reinterpret_cast<const f32x4*>(s);
        if constexpr (PAIR != 0) {
            l[1] = *reinterpret_cast<const f32x4*>(s + G4_GROW);
            l[2] = *reinterpret_cast<const f32x4*>(s + 2 * G4_GROW);
            l[3] = *reinterpret_cast<const f32x4*>(s + 3 * G4_GROW);
        }
    }
    __device__ __forceinline__ void begin(const float* rawg, float* V, int t, int q8, int pb, const G4Neg ng) {
        // the channel quads of odd tiles sit rotated by 8 in the raw image (bank spreading): tile J lives at (J + (t & 1)) % 3
        const int jq = ((t & 1) ? ((J + 1) % 3) : J) * 32 + q8 * 4;
        if constexpr (PAIR == 0) {
            src = rawg + t * 384 + jq + pb * 3 * G4_GROW;           // alpha 0: row 0; alpha 5: row 3
            dst = V + pb * 5 * 6 * G4_VPLANE + t * 128 + 32 + J * 32 + q8 * 4;
            kp = 0.f;
        } else {
            src = rawg + t * 384 + jq;
            dst = V + (2 * PAIR - 1 + pb) * 6 * G4_VPLANE + t * 128 + 32 + J * 32 + q8 * 4;
            kp = PAIR == 1 ? (pb ? ng.a : G4_A) : (pb ? ng.b : G4_B);
        }
        load(0, L[0]);
    }
    __device__ __forceinline__ void rows(const G4Neg ng) {
        const float kg2 = PAIR == 1 ? G4_A2 : G4_B2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c + 1 < 4) load(c + 1, L[(c + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const f32x4(&l)[4] = L[c & 1];
            if constexpr (PAIR == 0) T[c] = l[0];
            else {
                const f32x4 e = kg2 * l[2] + l[0], o = kg2 * l[3] + l[1];
                T[c] = kp * o + e;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __device__ __forceinline__ void finish(const G4Neg ng) {
        *reinterpret_cast<f32x4*>(dst) = T[0];
        *reinterpret_cast<f32x4*>(dst + 5 * G4_VPLANE) = T[3];
        {
            const f32x4 ea = G4_A2 * T[2] + T[0], oa = G4_A2 * T[3] + T[1];
            *reinterpret_cast<f32x4*>(dst + 1 * G4_VPLANE) = G4_A * oa + ea;
            *reinterpret_cast<f32x4*>(dst + 2 * G4_VPLANE) = ng.a * oa + ea;
        }
        {
            const f32x4 eb = G4_B2 * T[2] + T[0], ob = G4_B2 * T[3] + T[1];
            *reinterpret_cast<f32x4*>(dst + 3 * G4_VPLANE) = G4_B * ob + eb;
            *reinterpret_cast<f32x4*>(dst + 4 * G4_VPLANE) = ng.b * ob + eb;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
};
// three rounds of one wave, software pipelined: begin(i + 1) before finish(i)
template <typename R0, typename R1, typename R2>
__device__ __forceinline__ void g4_three_rounds(const float* raw0, const float* raw1, const float* raw2, float* V, int t, int q8,
                                                int pb, const G4Neg ng) {
    R0 r0;
    r0.begin(raw0, V, t, q8, pb, ng);
    r0.rows(ng);
    R1 r1;
    r1.begin(raw1, V, t, q8, pb, ng);
    r0.finish(ng);
    r1.rows(ng);
    R2 r2;
    r2.begin(raw2, V, t, q8, pb, ng);
    r1.finish(ng);
    r2.rows(ng);
    r2.finish(ng);
}

// Block index -> (XCD, channel-block pair, split): the workgroups of a split sit next to each other on one XCD and walk
// the same strips at the same time, so the dL/dy tile that the Cin / 32 input-channel groups share (and the x slices
// the Cout / 96 output groups share) are served by that XCD's L2 after the first read.
__global__ __launch_bounds__(256, 1) void conv_wgrad_wino43_kernel(const adh_conv_desc d, const Wg43Args g,
                                                                   float* __restrict__ slab) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // rawx | rawg | V
    const int bid = blockIdx.x;
    const int q2 = bid >> 3;
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
    // per-lane global offsets of the LDS-DMA pieces (a piece = 1 KB of the LDS image = 64 lanes x 16 B):
    //   x row = 3 pieces: lane -> slot u = 8 k + lane / 8 -> halo column u - u / 5 (gap and spare slots load a duplicate)
    //   dL/dy row = 6 pieces, pieces k and k + 3 are 8 pixels apart: lane -> (pixel, rotated channel quad)
    auto x_pattern = [&](const int k) {
        const int u = 8 * k + (lane >> 3);
        return adh_min_i(u - u / 5, 17) * xcs + (lane & 7) * 16 + cib * 128;
    };
    auto g_pattern = [&](const int k) {
        const int b = 1024 * k + 16 * lane;
        const int px = b / 384, qpos = (b % 384) / 16;
        return px * gcs + ((qpos + 24 - 8 * ((px >> 2) & 1)) % 24) * 16 + cob * 384;
    };
    // (kept in LDS, one ds_read_b32 per piece: six more live VGPRs across the transform phase made hipcc spill them and
    // reload them from scratch inside the contraction, where every scratch wait also drains the LDS-DMA pieces in flight)
    int* const tab = reinterpret_cast<int*>(lds + G4_RAWX_F + G4_RAWG_F + G4_V_F);
    if (tid < 64) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            tab[k * 64 + tid] = x_pattern(k);
            tab[(3 + k) * 64 + tid] = g_pattern(k);
        }
    }
    __syncthreads();

    const G4Neg ng = g4_neg_constants();

    // Descriptors cover the whole tensors (x shifted by one row + one pixel so that every scalar offset is non-negative)
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(d.in) - (int64_t)(d.IW + 1) * xcs), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.out), 0, 0xffffffff, 0x00020000);
    const unsigned xrow = (unsigned)d.IW * xcs, grow = (unsigned)d.VW * gcs;
    typedef __attribute__((address_space(1))) void* g4_gptr;
    const char* const xbase_c = reinterpret_cast<const char*>(d.in) - (int64_t)(d.IW + 1) * xcs;
    const char* const gbase_c = reinterpret_cast<const char*>(d.out);
    const unsigned lds0 = (unsigned)(uintptr_t)(lds_void_ptr5)lds;     // 32-bit LDS address of the dynamic segment
    // LDS destination (M0) and scalar offset go through readfirstlane at the point of use: hipcc otherwise carries some
    // of these wave-uniform values in VGPRs across the loop and issues the piece from a waterfall loop
    auto dma_x = [&](const int row, auto kc, const unsigned xb, const int pat) {
        constexpr int k = decltype(kc)::value;
        const unsigned ldsa = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(row * G4_XROW + k * 256) * 4);
        const unsigned so = __builtin_amdgcn_readfirstlane(xb + row * xrow);
        if constexpr (G4_GLOBAL_DMA)    // global_load_lds_dwordx4 (scalar base + per-lane offset): half the issue cost of the buffer form
            __builtin_amdgcn_global_load_lds((g4_gptr)(xbase_c + so + (unsigned)pat), (lds_void_ptr5)(uintptr_t)ldsa, 16, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void_ptr5)(uintptr_t)ldsa, 16, pat, so, 0, 0);
    };
    auto dma_g = [&](const int row, auto kc, const unsigned gb, const int pat) {
        constexpr int k = decltype(kc)::value;
        const unsigned ldsa = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(G4_RAWX_F + row * G4_GROW + k * 256) * 4);
        const unsigned so = __builtin_amdgcn_readfirstlane(gb + row * grow + (k / 3) * 8 * gcs);
        if constexpr (G4_GLOBAL_DMA)
            __builtin_amdgcn_global_load_lds((g4_gptr)(gbase_c + so + (unsigned)pat), (lds_void_ptr5)(uintptr_t)ldsa, 16, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (lds_void_ptr5)(uintptr_t)ldsa, 16, pat, so, 0, 0);
    };
    // piece schedule of an interior strip: wave w stages dL/dy row w (steps 0..5), x row w (6..8) and its share of x rows
    // 4 and 5 (9, 10): 11 / 11 / 10 / 10 pieces per wave
    // the six per-lane patterns, read from the table at the start of every staging sequence (volatile: a hoisted read
    // would be six more VGPRs live across the transform phase)
    struct Pat { int x0, x1, x2, g0, g1, g2; };
    auto load_patterns = [&]() {
        typedef const volatile __attribute__((address_space(3))) int* lds_vi;
        lds_vi t = (lds_vi)(tab + lane);
        Pat p;
        p.g0 = t[3 * 64]; p.g1 = t[4 * 64]; p.g2 = t[5 * 64];
        p.x0 = t[0]; p.x1 = t[64]; p.x2 = t[128];
        return p;
    };
    auto stage_piece = [&](auto stc, const unsigned xb, const unsigned gb, const Pat& p) {
        constexpr int ST = decltype(stc)::value;
        if constexpr (ST < 6) dma_g(wave, std::integral_constant<int, ST>{}, gb, ST % 3 == 0 ? p.g0 : (ST % 3 == 1 ? p.g1 : p.g2));
        else if constexpr (ST < 9) dma_x(wave, std::integral_constant<int, ST - 6>{}, xb, ST == 6 ? p.x0 : (ST == 7 ? p.x1 : p.x2));
        else if constexpr (ST == 9) {
            if (wave == 0) dma_x(4, std::integral_constant<int, 0>{}, xb, p.x0);
            else if (wave == 1) dma_x(4, std::integral_constant<int, 2>{}, xb, p.x2);
            else if (wave == 2) dma_x(5, std::integral_constant<int, 1>{}, xb, p.x1);
            else dma_x(5, std::integral_constant<int, 2>{}, xb, p.x2);
        } else if constexpr (ST == 10) {
            if (wave == 0) dma_x(4, std::integral_constant<int, 1>{}, xb, p.x1);
            else if (wave == 1) dma_x(5, std::integral_constant<int, 0>{}, xb, p.x0);
        }
    };
    auto stage_interior = [&](const unsigned xb, const unsigned gb) {
        const Pat p = load_patterns();
        stage_piece(std::integral_constant<int, 0>{}, xb, gb, p);
        stage_piece(std::integral_constant<int, 1>{}, xb, gb, p);
        stage_piece(std::integral_constant<int, 2>{}, xb, gb, p);
        stage_piece(std::integral_constant<int, 3>{}, xb, gb, p);
        stage_piece(std::integral_constant<int, 4>{}, xb, gb, p);
        stage_piece(std::integral_constant<int, 5>{}, xb, gb, p);
        stage_piece(std::integral_constant<int, 6>{}, xb, gb, p);
        stage_piece(std::integral_constant<int, 7>{}, xb, gb, p);
        stage_piece(std::integral_constant<int, 8>{}, xb, gb, p);
        stage_piece(std::integral_constant<int, 9>{}, xb, gb, p);
        stage_piece(std::integral_constant<int, 10>{}, xb, gb, p);
    };
    // register-staged form of the same pieces: (LDS byte address, scalar offset, pattern, resource) of piece ST of this wave
    auto piece_of = [&](auto stc, const unsigned xb, const unsigned gb, const Pat& p, unsigned& ldsa, unsigned& so, int& pat, bool& isg) {
        constexpr int ST = decltype(stc)::value;
        if constexpr (ST < 6) {
            isg = true;
            ldsa = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(G4_RAWX_F + wave * G4_GROW + ST * 256) * 4);
            so = __builtin_amdgcn_readfirstlane(gb + wave * grow + (ST / 3) * 8 * gcs);
            pat = ST % 3 == 0 ? p.g0 : (ST % 3 == 1 ? p.g1 : p.g2);
            return true;
        } else {
            isg = false;
            int row = wave, k = ST - 6;
            bool valid = true;
            if constexpr (ST == 9) { row = wave < 2 ? 4 : 5; k = wave == 0 ? 0 : (wave == 2 ? 1 : 2); }
            if constexpr (ST == 10) { row = 4 + wave; k = 1 - wave; valid = wave < 2; }
            ldsa = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(row * G4_XROW + k * 256) * 4);
            so = __builtin_amdgcn_readfirstlane(xb + row * xrow);
            pat = k == 0 ? p.x0 : (k == 1 ? p.x1 : p.x2);
            return valid;
        }
    };
    // strip s -> scalar offsets of its first halo pixel / first dL/dy pixel; returns whether the strip is interior
    auto strip_origin = [&](int s, int& y0, int& x0, unsigned& xb, unsigned& gb) {
        int r = s;
        const int sx = r % g.SX;
        r /= g.SX;
        const int ty = r % g.TY;
        const int n = r / g.TY;
        y0 = ty * 4;
        x0 = sx * 16;
        xb = __builtin_amdgcn_readfirstlane((unsigned)((n * d.IH + y0) * d.IW + x0) * xcs);
        gb = __builtin_amdgcn_readfirstlane((unsigned)((n * d.VH + y0) * d.VW + x0) * gcs);
        return y0 >= 1 && y0 + 5 <= d.IH && x0 >= 1 && x0 + 17 <= d.IW;
    };
    // border strip: zero both raw buffers, then load only the pixels inside the image (x rows w, w + 4; dL/dy row w)
    auto stage_border = [&](const int sb) {
        int y0, x0;
        unsigned xb, gb;
        (void)strip_origin(sb, y0, x0, xb, gb);   // recomputed here: nothing of it stays live across the contraction
        int ln = lane, td = tid;                  // opaque copies: keeps LICM from hoisting this rare path's lane predicates
        asm volatile("" : "+v"(ln), "+v"(td));   // (nine 64-bit masks + their operands) out of the strip loop
        const Pat bp = load_patterns();
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int i = td; i < (G4_RAWX_F + G4_RAWG_F) / 4; i += 256) *reinterpret_cast<f32x4*>(lds + i * 4) = z;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll 1
        for (int row = wave; row < 6; row += 4) {
            const int iy = y0 - 1 + row;
            const bool rowok = iy >= 0 && iy < d.IH;
            auto one = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const int u = 8 * k + (ln >> 3);
                const int ix = x0 - 1 + u - u / 5;
                if (rowok && (u % 5) != 4 && u < 22 && ix >= 0 && ix < d.IW) dma_x(row, kc, xb, k == 0 ? bp.x0 : (k == 1 ? bp.x1 : bp.x2));
            };
            one(std::integral_constant<int, 0>{});
            one(std::integral_constant<int, 1>{});
            one(std::integral_constant<int, 2>{});
        }
        {
            auto one = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const int px = (1024 * (k % 3) + 16 * ln) / 384 + 8 * (k / 3);
                if (x0 + px < d.VW) dma_g(wave, kc, gb, k % 3 == 0 ? bp.g0 : (k % 3 == 1 ? bp.g1 : bp.g2));
            };
            one(std::integral_constant<int, 0>{});
            one(std::integral_constant<int, 1>{});
            one(std::integral_constant<int, 2>{});
            one(std::integral_constant<int, 3>{});
            one(std::integral_constant<int, 4>{});
            one(std::integral_constant<int, 5>{});
        }
    };

    // ------------------------------------------------------------------ transform roles (58 / 58 / 52 / 60 float4 FMAs)
    const int q8 = lane & 7, pb = (lane >> 3) & 1, tt = lane >> 4;
    auto transform = [&]() {
        if (wave == 0) g4_three_rounds<G4X<1>, G4G<1, 0>, G4G<0, 0>>(rawx, rawg, rawg, V, tt, q8, pb, ng);
        else if (wave == 1) g4_three_rounds<G4X<2>, G4G<2, 0>, G4G<0, 1>>(rawx, rawg, rawg, V, tt, q8, pb, ng);
        else if (wave == 2) g4_three_rounds<G4X<0>, G4G<1, 1>, G4G<0, 2>>(rawx, rawg, rawg, V, tt, q8, pb, ng);
        else g4_three_rounds<G4G<1, 2>, G4G<2, 1>, G4G<2, 2>>(rawg, rawg, rawg, V, tt, q8, pb, ng);
    };

    // ------------------------------------------------------------------ main loop over this split's strips
    f32x16 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // A / B operand lane address: V[9 * wave + fi][2 * ks + h][l31 | 32 + nj * 32 + l31]
    g4_lds_vf const vlane = (g4_lds_vf)(V + (9 * wave) * G4_VPLANE + h * 128 + l31);

    // this split's strips: a contiguous range (the workgroups of a split walk it together)
    const int per = (g.S + g.nsplit - 1) / g.nsplit;
    int s = split * per;
    const int s_end = adh_min_i(s + per, g.S);
    if (s < s_end) {
        int y0, x0;
        unsigned xb, gb;
        if (strip_origin(s, y0, x0, xb, gb)) stage_interior(xb, gb);
        else stage_border(s);
    }
#pragma unroll 1
    while (s < s_end) {
        if (!(G4_DBG & 16)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // strip s landed; every wave is past the previous contraction (V is free)
        if (!(G4_DBG & 1)) transform();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // V complete, raw consumed: the next strip may land
        const int sn = s + 1;
        int y0 = 0, x0 = 0;
        unsigned xb = 0, gb = 0;
        const bool more = sn < s_end && !(G4_DBG & 8);
        const bool inner = more && strip_origin(sn, y0, x0, xb, gb);
        if (!(G4_DBG & 4)) {
            float oa[2], ob[2][3];
            g4_load_ops<0>(vlane, oa[0], ob[0]);
            if ((G4_DBG & 32) && inner) stage_interior(xb, gb);
            Pat pt = load_patterns();
            if constexpr (G4_STAGE_REGS) {
                constexpr int LAG = 5;                       // steps between a load and its LDS write (5 x 192 MFMA cycles)
                f32x4 stg[11];
                // LDS write addresses: three per-lane bases defined HERE (opaque: hoisted out of the strip loop they would be live
                // across the transform phase) + immediates
                unsigned bx = lane * 16 + wave * (G4_XROW * 4), bg = lane * 16 + (G4_RAWX_F + wave * G4_GROW) * 4, b0 = lane * 16;
                asm volatile("" : "+v"(bx), "+v"(bg), "+v"(b0));
                // a strip without an interior successor still issues the loads (from the tensors' first pixels) and skips the writes:
                // one instruction stream, no accumulator phis
                const unsigned xb1 = inner ? xb : (unsigned)(d.IW + 1) * xcs, gb1 = inner ? gb : 0u;
                g4_contract<0>(acc, vlane, oa, ob, [&](auto stc) {
                    constexpr int ST = decltype(stc)::value;
                    if constexpr (ST < 11) {
                        unsigned ldsa, so;
                        int pat;
                        bool isg;
                        (void)piece_of(stc, xb1, gb1, pt, ldsa, so, pat, isg);
                        stg[ST] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(isg ? gr : xr, pat, so, 0));
                    }
                    if constexpr (ST >= LAG && ST - LAG < 11) {
                        constexpr int P = ST - LAG;
                        unsigned ldsa = 0, so;
                        int pat;
                        bool isg;
                        typedef __attribute__((address_space(3))) f32x4* lds_f4;
                        if constexpr (P < 6) {
                            if (inner) *(lds_f4)(uintptr_t)(bg + P * 1024) = stg[P];
                        } else if constexpr (P < 9) {
                            if (inner) *(lds_f4)(uintptr_t)(bx + (P - 6) * 1024) = stg[P];
                        } else {
                            const bool valid = piece_of(std::integral_constant<int, P>{}, xb1, gb1, pt, ldsa, so, pat, isg) && inner;
                            if (valid) *(lds_f4)(uintptr_t)(b0 + (ldsa - lds0)) = stg[P];
                        }
                    }
                });
            } else {
                // The pieces of this wave need five scalar offsets per strip (dL/dy row w: two halves; x row w; x rows 4, 5): computed
                // once here and pinned to SGPRs, so that a piece in the MFMA stream is s_add m0 + buffer_load and nothing else (left to
                // hipcc the offsets lived in VGPRs under SGPR pressure: a v_add + v_readfirstlane per piece between two MFMAs)
                unsigned sg0 = __builtin_amdgcn_readfirstlane(gb + wave * grow), sg1 = __builtin_amdgcn_readfirstlane(gb + wave * grow + 8 * gcs);
                unsigned sx = __builtin_amdgcn_readfirstlane(xb + wave * xrow), sx4 = __builtin_amdgcn_readfirstlane(xb + 4 * xrow),
                         sx5 = __builtin_amdgcn_readfirstlane(xb + 5 * xrow);
                unsigned mg = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(G4_RAWX_F + wave * G4_GROW) * 4);
                unsigned mx = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(wave * G4_XROW) * 4);
                asm volatile("" : "+s"(sg0), "+s"(sg1), "+s"(sx), "+s"(sx4), "+s"(sx5), "+s"(mg), "+s"(mx));
                auto lean = [&](const __amdgpu_buffer_rsrc_t& rs, const unsigned m0v, const int pat, const unsigned so) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_ptr5)(uintptr_t)m0v, 16, pat, so, 0, 0);
                };
                g4_contract<0>(acc, vlane, oa, ob, [&](auto stc) {
                    constexpr int ST = decltype(stc)::value;
                    if constexpr (ST < 11 && !(G4_DBG & 32)) {
                        if (inner) {
                            if constexpr (ST < 6) lean(gr, mg + ST * 1024, ST % 3 == 0 ? pt.g0 : (ST % 3 == 1 ? pt.g1 : pt.g2), ST < 3 ? sg0 : sg1);
                            else if constexpr (ST < 9) lean(xr, mx + (ST - 6) * 1024, ST == 6 ? pt.x0 : (ST == 7 ? pt.x1 : pt.x2), sx);
                            else if constexpr (ST == 9) {
                                if (wave == 0) lean(xr, lds0 + (4 * G4_XROW + 0 * 256) * 4, pt.x0, sx4);
                                else if (wave == 1) lean(xr, lds0 + (4 * G4_XROW + 2 * 256) * 4, pt.x2, sx4);
                                else if (wave == 2) lean(xr, lds0 + (5 * G4_XROW + 1 * 256) * 4, pt.x1, sx5);
                                else lean(xr, lds0 + (5 * G4_XROW + 2 * 256) * 4, pt.x2, sx5);
                            } else {
                                if (wave == 0) lean(xr, lds0 + (4 * G4_XROW + 1 * 256) * 4, pt.x1, sx4);
                                else if (wave == 1) lean(xr, lds0 + (5 * G4_XROW + 0 * 256) * 4, pt.x0, sx5);
                            }
                        }
                    }
                });
            }
        } else if (inner) {
            stage_interior(xb, gb);
        }
        if (more && !inner) stage_border(sn);
        s = sn;
    }

    // ------------------------------------------------------------------ partial sums -> slab[split][f][KP][NcP]
    float* const sl = slab + (int64_t)split * 36 * g.KP * d.NcP;
#pragma unroll
    for (int fi = 0; fi < 9; ++fi) {
        float* const pf = sl + (int64_t)(9 * wave + fi) * g.KP * d.NcP;
#pragma unroll
        for (int nj = 0; nj < 3; ++nj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = cib * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                pf[(int64_t)ci * d.NcP + cob * 96 + nj * 32 + l31] = acc[fi * 3 + nj][r];
            }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int wgrad43_plan(const adh_conv_desc* d, int nsplit, Wg43Args* a) {
    static const bool enabled = !(getenv("ADH_WINO43_WGRAD") && getenv("ADH_WINO43_WGRAD")[0] == '0');   // A/B switch
    if (!enabled || !d) return 0;
    if (d->KH != 3 || d->KW != 3 || d->in_sy != 1 || d->in_sx != 1 || d->out_sy != 1 || d->out_sx != 1) return 0;
    if (d->out_oy != 0 || d->out_ox != 0 || d->dy0 != -1 || d->dx0 != -1 || d->dstep_y != 1 || d->dstep_x != 1) return 0;
    if (d->Cin % 32 != 0 || d->NcP % 96 != 0 || d->Cout != d->NcP || (d->in_cstride & 3) || (d->out_cstride & 3)) return 0;
    if (d->in_cstride < d->Cin || d->out_cstride < d->Cout) return 0;
    if (d->VH != d->IH || d->VW != d->IW || d->VH != d->OH || d->VW != d->OW || (d->VH & 3) || (d->VW & 3)) return 0;
    if ((int64_t)d->N * (d->IH + 2) * d->IW * d->in_cstride >= (1ll << 30) || (int64_t)d->N * d->VH * d->VW * d->out_cstride >= (1ll << 30))
        return 0;
    a->TY = d->VH / 4;
    a->SX = adh_ceil_div(d->VW / 4, G4_T);
    a->S = d->N * a->TY * a->SX;
    a->nsplit = nsplit;
    a->ncob = d->NcP / 96;
    a->ngroups = (d->Cin / 32) * a->ncob;
    a->KP = d->Cin;
    return 1;
}

// workgroups per pixel split (channel-block pairs); 0 = not this path
extern "C" int adh_conv_wgrad_wino43_groups(const adh_conv_desc* d) {
    Wg43Args a;
    return wgrad43_plan(d, 1, &a) ? a.ngroups : 0;
}

// strips (4 x 16 output pixels) the pixel splits divide among themselves
extern "C" int adh_conv_wgrad_wino43_strips(const adh_conv_desc* d) {
    Wg43Args a;
    return wgrad43_plan(d, 1, &a) ? a.S : 0;
}

extern "C" int adh_conv_wgrad_wino43(void* stream, const adh_conv_desc* d, float* slab, int nsplit) {
    if (!d || !slab || nsplit < 1 || !d->in || !d->out) return ADH_E_ARG;
    Wg43Args a;
    if (!wgrad43_plan(d, nsplit, &a)) return ADH_E_UNSUPPORTED;
    if (((uintptr_t)d->in & 15) || ((uintptr_t)d->out & 15)) return ADH_E_ARG;
    const int nblocks = ((nsplit + 7) / 8) * a.ngroups * 8;
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
