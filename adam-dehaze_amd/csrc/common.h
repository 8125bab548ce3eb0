// Shared device/host helpers for libadamdehaze_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/adam_dehaze_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define ADH_WAVE 64

static inline int adh_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADH_OK : ADH_E_LAUNCH;
}

__host__ __device__ static inline int adh_min_i(int a, int b) { return a < b ? a : b; }
__host__ __device__ static inline int adh_max_i(int a, int b) { return a > b ? a : b; }
static inline int adh_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int adh_round_up(int a, int b) { return (a + b - 1) / b * b; }

// Tap-extent helpers (host + device)
__host__ __device__ static inline int adh_tap_min(int d0, int step, int k) {
    int e = d0 + (k - 1) * step;
    return d0 < e ? d0 : e;
}
__host__ __device__ static inline int adh_tap_max(int d0, int step, int k) {
    int e = d0 + (k - 1) * step;
    return d0 > e ? d0 : e;
}

// Geometry of one conv launch, derived on the host and passed by value.
struct ConvGeom {
    int dmin_y, dmin_x;     // smallest tap offsets
    int halo_h, halo_w;     // input footprint of one TH x 32 tile
    int npx, npxp;          // halo pixels, padded (odd) pitch
    int tiles_x, tiles_y;
    int KC, KQ_log2;        // channels per LDS chunk, log2(KC/4)
    int KQtot;              // padded Cin / 4
    int TH;                 // tile rows (8 for forward)
    int xp;                 // wgrad: floats per staged input pixel (32, or Cin alloc (8) in packed small-Cin mode)
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

// a - b on float4 as two v_pk_fma_f32 (b * (-1) + a, exact: one rounding) instead of the four v_sub_f32 hipcc emits for a
// vector subtraction; `m1` is -1.0f held in an SGPR the compiler cannot see through (adh_opaque(-1.f)), otherwise it folds
// the product back into a subtraction.  Matters where VALU instructions come out of fp32 MFMA time.
__device__ __forceinline__ float adh_opaque(float v) {
    asm volatile("" : "+s"(v));
    return v;
}
__device__ __forceinline__ f32x4 adh_pksub(const f32x4& a, const f32x4& b, float m1) { return m1 * b + a; }

// gfx950: an MFMA that reads a VGPR a VALU instruction wrote one or two instructions earlier gets the OLD value (measured
// with tools/dev_wgrad32_probe.py on conv_wgrad32_kernel: the product of the frequency whose MFMA hipcc scheduled right behind the v_fmac that
// finishes its B operand lost exactly that term).  hipcc pads VALU -> MFMA hazards for MFMA *instructions*; ours are inline
// asm, which its hazard recogniser does not look into.  The operands therefore pass through an asm that holds two wait
// states behind the last VALU write before any MFMA may issue.
template <int TN>
__device__ __forceinline__ void adh_mfma_operand_fence(float (&V)[4], float (&M)[TN][4]) {
    if constexpr (TN == 1)
        asm volatile("s_nop 1" : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(M[0][0]), "+v"(M[0][1]), "+v"(M[0][2]), "+v"(M[0][3]));
    if constexpr (TN == 2)
        asm volatile("s_nop 1" : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(M[0][0]), "+v"(M[0][1]), "+v"(M[0][2]), "+v"(M[0][3]),
                     "+v"(M[1][0]), "+v"(M[1][1]), "+v"(M[1][2]), "+v"(M[1][3]));
    if constexpr (TN == 3)
        asm volatile("s_nop 1" : "+v"(V[0]), "+v"(V[1]), "+v"(V[2]), "+v"(V[3]), "+v"(M[0][0]), "+v"(M[0][1]), "+v"(M[0][2]), "+v"(M[0][3]),
                     "+v"(M[1][0]), "+v"(M[1][1]), "+v"(M[1][2]), "+v"(M[1][3]), "+v"(M[2][0]), "+v"(M[2][1]), "+v"(M[2][2]), "+v"(M[2][3]));
}

// conv_wgrad.hip: slab[0] = sum over `nsplit` partial slabs of n4 float4 each (in place, fixed order)
void adh_wgrad_sum_splits(hipStream_t s, float* slab, int nsplit, int64_t n4);

// conv_rows.hip: direct forward kernel for the 2x2 / 3x3-tap gather forms (0 blocks / ADH_E_UNSUPPORTED when `d`
// is not one of its shapes; conv_igemm.hip then takes the launch)
int adh_rows_fwd_num_blocks(const adh_conv_desc* d);
int adh_rows_fwd_launch(void* stream, const adh_conv_desc* d);
