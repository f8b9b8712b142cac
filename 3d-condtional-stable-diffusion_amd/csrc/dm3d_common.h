// dm3d_common.h — shared host/device helpers for libdm3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include "dm3d.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing (never throws across the ABI) ---------------------------------------------------------------
char* dm3d_err_buf();   // thread-local, 512 bytes
static inline int dm3d_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dm3d_err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}
#define DM3D_REQUIRE(cond, ...) do { if (!(cond)) return dm3d_fail(DM3D_EINVAL, __VA_ARGS__); } while (0)
#define DM3D_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return dm3d_fail(DM3D_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)
static inline int dm3d_launch_check(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return dm3d_fail(DM3D_EHIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return DM3D_OK;
}
static inline bool dm3d_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int64_t dm3d_round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// H3 range guard (include/dm3d.h, range_flag): running max |value| of what an epilogue stores.  -DDM3D_NO_RANGE_GUARD builds the
// A/B variant without it (tools/mk_variant.sh).
#ifdef DM3D_NO_RANGE_GUARD
#define DM3D_AMAX(a, v) ((void)(v))
#else
#define DM3D_AMAX(a, v) ((a) = fmaxf((a), fabsf(v)))
#endif

// ---- device helpers -----------------------------------------------------------------------------------------------
__device__ __forceinline__ float dm3d_silu(float y) {
    // y * sigmoid(y) as mul, v_exp_f32, add, v_rcp_f32, mul (both ~1 ulp, far inside the 1e-3 parity budget).
    // __frcp_rn / 1.0f/x would expand to the ~10-instruction correctly-rounded division sequence.
    return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.4426950408889634f));
}
__device__ __forceinline__ float dm3d_act(float v, int act) {
    if (act == DM3D_ACT_RELU) return fmaxf(v, 0.0f);
    if (act == DM3D_ACT_SILU) return dm3d_silu(v);
    return v;
}

// One (tap, Cin-chunk) step of the implicit GEMM for a wave owning MR x NR tiles of 32x32 outputs.
// v_mfma_f32_32x32x2_f32: lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31]; the two lane halves carry
// two different k, so with CK channels per chunk half h takes channels [h*CK/2, (h+1)*CK/2) as CK/8 float4 reads and
// MFMA (q, j) contracts channels {q*4+j, CK/2+q*4+j}.  A and B use the same assignment, so the order is irrelevant.
//   a_lds[mr]: this lane's voxel row (channel 0 of the chunk) for row-tile mr, already offset by the tap
//   b_lds[nr]: this lane's output-channel row of the weight slice for col-tile nr
template <int MR, int NR, int CK>
__device__ __forceinline__ void dm3d_mma_step(f32x16 (&acc)[MR][NR], const float* (&a_lds)[MR],
                                              const float* (&b_lds)[NR], int half) {
    constexpr int KQ = CK / 8;
    f32x4 a[MR][KQ], b[NR][KQ];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int q = 0; q < KQ; ++q) a[mr][q] = *reinterpret_cast<const f32x4*>(a_lds[mr] + half * (CK / 2) + q * 4);
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
        for (int q = 0; q < KQ; ++q) b[nr][q] = *reinterpret_cast<const f32x4*>(b_lds[nr] + half * (CK / 2) + q * 4);
#pragma unroll
    for (int q = 0; q < KQ; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                    acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mr][q][j], b[nr][q][j], acc[mr][nr], 0, 0, 0);
}

// C/D layout of the 32x32 MFMA: register r of lane l holds row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31.
__device__ __forceinline__ int dm3d_acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Weight of tap2 = (td,th,tw) in {0,1}^3 of the 2x2x2 conv that reproduces, for output parity par = (a,b,c), a 3x3x3 conv on
// the nearest-2x upsampled tensor (reference UpSample, conditional_dm3d.py:288-296): along each axis the three taps of the
// k3 kernel read source voxels {i-1, i, i} (parity 0) or {i, i, i+1} (parity 1), so the taps sharing a source are summed.
__device__ __forceinline__ float dm3d_up_weight(const float* __restrict__ w, int cin, int cout, int par, int tap2, int ci, int co) {
    float acc = 0.f;
    const int pa[3] = {par >> 2, (par >> 1) & 1, par & 1};
    const int tt[3] = {tap2 >> 2, (tap2 >> 1) & 1, tap2 & 1};
    int lo[3], hi[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
        if (pa[ax] == 0) { lo[ax] = tt[ax] == 0 ? 0 : 1; hi[ax] = tt[ax] == 0 ? 0 : 2; }
        else             { lo[ax] = tt[ax] == 0 ? 0 : 2; hi[ax] = tt[ax] == 0 ? 1 : 2; }
    }
    for (int kd = lo[0]; kd <= hi[0]; ++kd)
        for (int kh = lo[1]; kh <= hi[1]; ++kh)
            for (int kw = lo[2]; kw <= hi[2]; ++kw)
                acc += w[((long)((kd * 3 + kh) * 3 + kw) * cin + ci) * cout + co];
    return acc;
}

// Weight of tap2 of the 2x2x2 conv that reproduces, for output parity par, Conv3DTranspose(k=4, strides=2, padding="same")
// (reference vqvae3d_monai.py:373-377).  Forward conv y[i] = sum_k x[2i+k-1] w[k]; its transpose out[j] = sum_{2i+k-1=j} y[i] w[k]:
// j = 2m reads y[m-1] w[3] + y[m] w[1]; j = 2m+1 reads y[m] w[2] + y[m+1] w[0].  Keras layout [kd,kh,kw,Cout,Cin].
__device__ __forceinline__ float dm3d_convt_weight(const float* __restrict__ w, int cin, int cout, int par, int tap2, int ci, int co) {
    const int pa[3] = {par >> 2, (par >> 1) & 1, par & 1};
    const int tt[3] = {tap2 >> 2, (tap2 >> 1) & 1, tap2 & 1};
    int k[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) k[ax] = pa[ax] == 0 ? (tt[ax] == 0 ? 3 : 1) : (tt[ax] == 0 ? 2 : 0);
    return w[((long)((k[0] * 4 + k[1]) * 4 + k[2]) * cout + co) * cin + ci];
}
