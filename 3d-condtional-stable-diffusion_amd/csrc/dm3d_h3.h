// dm3d_h3.h — device helpers of the split-float16 ("H3") arithmetic shared by the conv, GEMM and elementwise kernels.
//
// x (float32) = hi + lo with hi = fp16(x), lo = fp16(x - hi): |x - hi - lo| <= 2^-22 |x| while lo stays a normal float16.
// A product a.b is taken as al.bh + ah.bl + ah.bh (three v_mfma_f32_32x32x16_f16 into one float32 accumulator).
// Storage unit everywhere (LDS records, DM3D_FMT_H2 rows, packed weights): 16 consecutive k as a 64-byte record of four
// 16-byte slots [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15]; in LDS slot s of record v sits at physical slot s ^ swz(v).
#pragma once
#include "dm3d_common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int DM3D_REC = 32;            // halfs per record

__device__ __forceinline__ int swz(int v) { return (v >> 2) & 3; }

// 8 consecutive k -> float16 hi and lo terms; lim0/lim1 are 65504 (clamp to the float16 range) or 0 (padding / out-of-range
// position -> exact zero), so one v_med3_f32 clamps and masks.  Per pair: v_cvt_pk_f16_f32 (hi, RNE), 2 x v_fma_mix_f32
// (x - hi with hi read as f16 straight from the packed register), v_cvt_pk_f16_f32 (lo) — 3 VALU per element where hipcc's
// own lowering of the same arithmetic takes 7.  VALU issue slots are what the H3 kernels run out of first.
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, float lim0, float lim1, h8& hi, h8& lo) {
    u32x4 ph, pl;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const float lim = e < 4 ? lim0 : lim1;
        const float x0 = __builtin_amdgcn_fmed3f(e < 4 ? v0[e] : v1[e - 4], -lim, lim);
        const float x1 = __builtin_amdgcn_fmed3f(e < 4 ? v0[e + 1] : v1[e - 3], -lim, lim);
        unsigned int a, r;
        float r0, r1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(a) : "v"(x0), "v"(x1));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(a), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(a), "v"(x1));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(r0), "v"(r1));
        ph[e >> 1] = a;
        pl[e >> 1] = r;
    }
    hi = __builtin_bit_cast(h8, ph);
    lo = __builtin_bit_cast(h8, pl);
}

// one value -> (hi | lo << 16) as raw float16 bit patterns
__device__ __forceinline__ unsigned int split1_bits(float x) {
    x = __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
    const _Float16 a = (_Float16)x;
    const _Float16 r = (_Float16)(x - (float)a);
    return (unsigned int)__builtin_bit_cast(unsigned short, a) | ((unsigned int)__builtin_bit_cast(unsigned short, r) << 16);
}


// ---- "H3F8" arithmetic: the two cross terms of the split on float8 (e4m3) operands -------------------------------------------------
// a.b = ah.bh + al.bh + ah.bl with ah.bh on v_mfma_f32_16x16x32_f16 as before and BOTH cross terms on ONE stream of
// v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales; 2.25x the float16 MFMA rate per FLOP, tools/micro/mfma_f8_scaled.hip): the A side
// interleaves [ah8 x8 | al8 x8] per 8 channels, the B side [bl8 x8 | bh8 x8], so a K slice of 16 bytes contracts ah.bl + al.bh at once.
// Scales keep the small terms inside e4m3's range and cancel in each product: al8 = fp8(al * 2^11), bh8 = fp8(bh * 2^-11);
// ah8 = fp8(ah / 4), bl8 = fp8(bl * 4).  The cross terms are 2^-11 of the product, so their 2^-4 relative float8 rounding costs
// ~2^-15 per product: eps of the whole U-Net 5-9e-5 instead of 5-8e-6 (tests/study_fp8_cross_terms.py), against the 1e-3 contract.
// Activations are clamped to +-448 (al * 2^11 <= |a| must stay finite in e4m3: the conversions return NaN beyond 448, they do not
// saturate — tools/micro/cvt_fp8.hip); the range guard of this mode uses that bound.
// Record (64 bytes per 16 channels): slot p = hi16 c(8p..8p+7); slot 2+p = [ah8 c(8p..) x8 | al8 c(8p..) x8]  (weights: [bl8 | bh8]).
constexpr float DM3D_F8_LIMIT = 448.0f;
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

// one output dword = 4 channels: elements (x[0], x[1]) -> low half, (x[2], x[3]) -> high half
__device__ __forceinline__ void f8_quad(const float (&x)[4], unsigned int& hi01, unsigned int& hi23, unsigned int& ah8, unsigned int& al8) {
    float r[4];
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi01) : "v"(x[0]), "v"(x[1]));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi23) : "v"(x[2]), "v"(x[3]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(hi01), "v"(x[0]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(hi01), "v"(x[1]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[2]) : "v"(hi23), "v"(x[2]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[3]) : "v"(hi23), "v"(x[3]));
    s16x2 oa = {0, 0}, ol = {0, 0};
    oa = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(oa, __builtin_bit_cast(h16x2, hi01), 4.0f, false);              // ah / 4
    oa = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(oa, __builtin_bit_cast(h16x2, hi23), 4.0f, true);
    ol = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(ol, r[0], r[1], 0.00048828125f, false);                         // al * 2^11
    ol = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(ol, r[2], r[3], 0.00048828125f, true);
    ah8 = __builtin_bit_cast(unsigned int, oa);
    al8 = __builtin_bit_cast(unsigned int, ol);
}

// 8 consecutive channels -> hi (8 float16) and x8 = [ah8 x8 | al8 x8]; lim0 / lim1: DM3D_F8_LIMIT, or 0 for padding
__device__ __forceinline__ void split8_f8(const f32x4& v0, const f32x4& v1, float lim0, float lim1, h8& hi, h8& x8) {
    u32x4 ph, px;
    float a[4], b[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = __builtin_amdgcn_fmed3f(v0[e], -lim0, lim0); b[e] = __builtin_amdgcn_fmed3f(v1[e], -lim1, lim1); }
    unsigned int h0, h1, h2, h3, a0, a1, l0, l1;
    f8_quad(a, h0, h1, a0, l0);
    f8_quad(b, h2, h3, a1, l1);
    ph[0] = h0; ph[1] = h1; ph[2] = h2; ph[3] = h3;
    px[0] = a0; px[1] = a1; px[2] = l0; px[3] = l1;          // [ah8 c0-7 | al8 c0-7]
    hi = __builtin_bit_cast(h8, ph);
    x8 = __builtin_bit_cast(h8, px);
}

// the same from an already split float16 pair of 8 channels (a DM3D_FMT_H2 record piece): hi stays, lo16 -> al8, hi16 -> ah8
__device__ __forceinline__ h8 h2_to_x8(const h8& hi, const h8& lo) {
    const u32x4 h = __builtin_bit_cast(u32x4, hi), l = __builtin_bit_cast(u32x4, lo);
    u32x4 px;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        // scalar copies first: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 whatever the index (hipcc 7.2)
        const unsigned int h0 = h[2 * d], h1 = h[2 * d + 1], l0 = l[2 * d], l1 = l[2 * d + 1];
        s16x2 oa = {0, 0}, ol = {0, 0};
        oa = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(oa, __builtin_bit_cast(h16x2, h0), 4.0f, false);
        oa = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(oa, __builtin_bit_cast(h16x2, h1), 4.0f, true);
        ol = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(ol, __builtin_bit_cast(h16x2, l0), 0.00048828125f, false);
        ol = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(ol, __builtin_bit_cast(h16x2, l1), 0.00048828125f, true);
        px[d] = __builtin_bit_cast(unsigned int, oa);
        px[2 + d] = __builtin_bit_cast(unsigned int, ol);
    }
    return __builtin_bit_cast(h8, px);
}
