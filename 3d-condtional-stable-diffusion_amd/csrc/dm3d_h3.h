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
