// bf16 x 3 contraction helpers shared by conv_wino43.hip and conv_wino.hip (gfx950): the exact three-plane split of fp32 values
// into bf16 (x = hi + mid + lo, 8 + 8 + 8 significant bits), the MFMA / weight-load / counted-wait statements, and the twenty
// "macro-steps" that turn a lane's eight fp32 A-operand values into its three plane registers a few VALU instructions at a
// time (they ride in the issue slots v_mfma_f32_32x32x16_bf16 leaves free).  DESIGN 4.15.
#pragma once
#include "common.h"
#ifndef B3_NO_SPLIT
#define B3_NO_SPLIT 0   // dev builds: 1 = skip the split steps (wrong results): what the fillers cost the contraction
#endif
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// v_cvt_pk_bf16_f32: two floats -> two bf16 (round to nearest even), a in the low half
__device__ __forceinline__ unsigned w4b_cvt_pk(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// (Z: C = 0 instead of the accumulator's old contents -- the first MFMA on a tile in a region, conv_wino43.hip)
template <bool AGPR, bool Z = false>
__device__ __forceinline__ void w4b_mfma(f32x16& c, const u32x4& a, const u32x4& b) {
    if constexpr (Z) {
        if constexpr (AGPR) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "+a"(c) : "v"(a), "v"(b));
        else asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "+v"(c) : "v"(a), "v"(b));
    } else {
        if constexpr (AGPR) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
        else asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
}
// the three planes of one (frequency, 32-channel tile) of U: [plane][64 lanes][16 B] = 3 KB contiguous (adh_pack_weights_wino43_bf16x3)
__device__ __forceinline__ void w4b_load_b(u32x4 (&b)[3], unsigned voff, const char* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(b[0]) : "v"(voff), "s"(sbase) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(b[1]) : "v"(voff), "s"(sbase) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(b[2]) : "v"(voff), "s"(sbase) : "memory");
}
template <int N>
__device__ __forceinline__ void w4b_wait_b(u32x4 (&b)[3]) {
    asm volatile("s_waitcnt vmcnt(%3)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]) : "n"(N) : "memory");
}
// The A operand of the NEXT frequency in the making: s = the lane's eight fp32 values of V (channels 4h .. 4h+3 and 8+4h .. 8+4h+3
// of its tile), turned into the residuals in place; h / m / l = the planes, two bf16 per register.  x = hi + mid + lo exactly
// (24 significant bits = 3 x 8: every subtraction is exact).  Twenty steps: per pair p of values  A hi = cvt(s) | B s -= hi |
// C mid = cvt(s) | D s -= mid | E lo = cvt(s), ordered A0 B0 .. A3 B3 C0 D0 .. C3 D3 E0 .. E3 (one to four VALU instructions each).
struct W4BNext {
    float s[8];
    unsigned h[4], m[4], l[4];
};
// x -= the bf16 in the low / high half of pk.  One as a subtraction, the other as fma(., -1, .) with an opaque -1 (exact either
// way): two subtractions side by side hipcc packs into one v_pk_add_f32, which costs far more than two scalar instructions in
// the shadow of an MFMA (MI355X guide: +13 cycles each).
#ifndef W4B_PKSUB
#define W4B_PKSUB 1   // 1: two plain subtractions (hipcc packs them: v_pk_add_f32); 0: the sub + fma form (register pressure: see DESIGN 4.15)
#endif
__device__ __forceinline__ void w4b_sub_halves(float& x0, float& x1, unsigned pk, float m1) {
    x0 -= __builtin_bit_cast(float, pk << 16);
#if W4B_PKSUB
    x1 -= __builtin_bit_cast(float, pk & 0xffff0000u);
#else
    x1 = __builtin_fmaf(__builtin_bit_cast(float, pk & 0xffff0000u), m1, x1);
#endif
}
template <int STEP>
__device__ __forceinline__ void w4b_step(W4BNext& n, float m1) {
    if (B3_NO_SPLIT) return;   // dev build: no split (wrong results): what the fillers cost the contraction
    constexpr int p = STEP < 16 ? (STEP & 7) >> 1 : STEP - 16;
    constexpr int kind = STEP < 16 ? (STEP & 1) + 2 * (STEP >> 3) : 4;
    if constexpr (kind == 0) n.h[p] = w4b_cvt_pk(n.s[2 * p], n.s[2 * p + 1]);
    if constexpr (kind == 1) w4b_sub_halves(n.s[2 * p], n.s[2 * p + 1], n.h[p], m1);
    if constexpr (kind == 2) n.m[p] = w4b_cvt_pk(n.s[2 * p], n.s[2 * p + 1]);
    if constexpr (kind == 3) w4b_sub_halves(n.s[2 * p], n.s[2 * p + 1], n.m[p], m1);
    if constexpr (kind == 4) n.l[p] = w4b_cvt_pk(n.s[2 * p], n.s[2 * p + 1]);
}
template <int S0, int S1>
__device__ __forceinline__ void w4b_steps(W4BNext& n, float m1) {
    if constexpr (S0 < S1) {
        w4b_step<S0>(n, m1);
        w4b_steps<S0 + 1, S1>(n, m1);
    }
}
