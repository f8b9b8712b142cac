// dm3d_conv_h3.hip — the same implicit-GEMM Conv3d as dm3d_conv.hip, on the 16-bit matrix pipe with float32-grade results.
//
// gfx950 has no TF32/xf32 path and its exact-fp32 MFMA runs at 1/16 of the 16-bit rate.  Here every float32 operand is
// split into two float16 terms, x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (|x - hi - lo| <= 2^-22 |x|), and the
// product a.b is evaluated as al.bh + ah.bl + ah.bh by three v_mfma_f32_32x32x16_f16 passes into one float32
// accumulator; the dropped al.bl term is <= 2^-22 |a.b|.  That is float32-grade arithmetic at up to 1/3 of the 16-bit
// MFMA peak = 5.3x the fp32-MFMA peak.  The split costs the same 4 bytes per element in LDS as float32.
//   * activations: split while the halo tile is staged into LDS (after the fused silu(x*scale+shift) and zero padding),
//     clamped to +-65504 first;
//   * weights: split once by dm3d_pack_weights_h3, pre-multiplied by 2^w_exp so that the lo terms are normal float16
//     numbers; the epilogue multiplies the accumulator by 2^-w_exp (exact).
//
// Tile structure (see dm3d_conv.hip for the rationale): workgroup = 256 threads, TD x TH x TW output voxels x 64 output
// channels; per 16-channel chunk the input halo lives in LDS as 80-byte records [16 hi | 16 lo | pad]; the k^3 taps walk it
// by address only; weight slices arrive in groups of 3 taps (one (kd,kh) row), double-buffered, one barrier per group.
// One k3 tap = 12 MFMAs (384 cycles) per wave against 8 ds_read_b128.
#include "dm3d_conv_args.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int REC = 40;            // halfs per LDS record: 16 hi, 16 lo, 8 pad (80 B keeps ds_read_b128 banks spread)

__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, h8& hi, h8& lo) {
    // 8 consecutive channels -> float16 hi and lo terms, x = hi + lo up to 2^-22 |x|
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = fminf(fmaxf(e < 4 ? v0[e] : v1[e - 4], -65504.0f), 65504.0f);
        const _Float16 a = (_Float16)x;
        hi[e] = a;
        lo[e] = (_Float16)(x - (float)a);
    }
}

template <int TD, int TH, int TW, int S, int KS, int WM, int WN, int MINW>
__global__ __launch_bounds__(256, MINW) void conv3d_igemm_h3(const ConvArgs p) {
    constexpr int CK = 16, NT = 64;
    constexpr int TM = TD * TH * TW;
    constexpr int HD = (TD - 1) * S + KS, HH = (TH - 1) * S + KS, HW = (TW - 1) * S + KS;
    constexpr int HVOX = HD * HH * HW;
    constexpr int TAPS = KS * KS * KS;
    constexpr int G = KS;                       // taps per weight group (one kw row), TAPS / G groups per chunk
    constexpr int NG = TAPS / G;
    constexpr int MR = TM / WM / 32, NR = NT / WN / 32;
    constexpr int NSLOT = (HVOX * 2 + 255) / 256;          // (voxel, 8-channel piece) slots per thread
    constexpr int WGRP = G * NT * REC;                      // halfs per weight group
    constexpr int WPIECES = WGRP * 2 / 16;                  // 16-byte pieces per group
    constexpr int WSLOT = (WPIECES + 255) / 256;
    static_assert(WM * WN == 4, "kernel assumes 4 waves");

    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    _Float16* lds_in = smem_h;                  // [HVOX][REC]
    _Float16* lds_w = smem_h + HVOX * REC;      // [2][G][NT][REC]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;

    int brick = blockIdx.x;
    const int bpv = p.bd * p.bh * p.bw;
    const int b = brick / bpv;
    brick -= b * bpv;
    const int oz0 = (brick / (p.bh * p.bw)) * TD;
    const int oy0 = ((brick / p.bw) % p.bh) * TH;
    const int ox0 = (brick % p.bw) * TW;
    const int ntile = blockIdx.y;

    // staging slots: this thread converts channels [8*piece, 8*piece+8) of halo voxels (tid>>1) + j*128
    const int piece = tid & 1;
    int gvox[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        const int hv = (tid >> 1) + j * 128;
        int g = -1;
        if (hv < HVOX) {
            const int hz = hv / (HH * HW), hy = (hv / HW) % HH, hx = hv % HW;
            const int iz = oz0 * S - p.pad + hz, iy = oy0 * S - p.pad + hy, ix = ox0 * S - p.pad + hx;
            if (iz >= 0 && iz < p.lgd && iy >= 0 && iy < p.lgh && ix >= 0 && ix < p.lgw) {
                const int pz = p.ups ? (iz >> 1) : iz, py = p.ups ? (iy >> 1) : iy, px = p.ups ? (ix >> 1) : ix;
                g = ((b * p.ind + pz) * p.inh + py) * p.inw + px;
            }
        }
        gvox[j] = g;
    }

    // per-lane record of the A operand for each 32-row tile at tap (0,0,0); lane half h reads hi at +8h, lo at +16+8h
    int a_off[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const int r = wm * (TM / WM) + mr * 32 + l32;
        const int dz = r / (TH * TW), dy = (r / TW) % TH, dx = r % TW;
        a_off[mr] = ((dz * S * HH + dy * S) * HW + dx * S) * REC + half * 8;
    }
    int b_off[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) b_off[nr] = (wn * (NT / WN) + nr * 32 + l32) * REC + half * 8;

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.0f;

    // weight image: [ntile][chunk][group][G][NT][REC] halfs; a group is a dense run of WPIECES 16-byte pieces
    const f32x4* w_img = reinterpret_cast<const f32x4*>(static_cast<const _Float16*>(p.wpk) +
                                                        (size_t)ntile * p.nchunks * NG * WGRP);
    f32x4 wreg[WSLOT];
    auto fetch_w = [&](int group_index) {
        const f32x4* src = w_img + (size_t)group_index * WPIECES;
#pragma unroll
        for (int i = 0; i < WSLOT; ++i) {
            const int q = tid + i * 256;
            if (q < WPIECES) wreg[i] = src[q];
        }
    };
    auto store_w = [&](int buf) {
        f32x4* dst = reinterpret_cast<f32x4*>(lds_w + buf * WGRP);
#pragma unroll
        for (int i = 0; i < WSLOT; ++i) {
            const int q = tid + i * 256;
            if (q < WPIECES) dst[q] = wreg[i];
        }
    };
    fetch_w(0);

    const bool pro = p.pscale != nullptr;
    for (int ch = 0; ch < p.nchunks; ++ch) {
        const int c0 = ch * CK;
        const float* src;
        int ldc, cb;
        if (c0 < p.c1) { src = p.x1; ldc = p.c1; cb = c0; } else { src = p.x2; ldc = p.c2; cb = c0 - p.c1; }
        const int cpos = cb + piece * 8;
        const bool ok0 = cpos < ldc, ok1 = cpos + 4 < ldc;
        f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
        if (pro) {
            if (ok0) { sc0 = *reinterpret_cast<const f32x4*>(p.pscale + c0 + piece * 8);
                       sh0 = *reinterpret_cast<const f32x4*>(p.pshift + c0 + piece * 8); }
            if (ok1) { sc1 = *reinterpret_cast<const f32x4*>(p.pscale + c0 + piece * 8 + 4);
                       sh1 = *reinterpret_cast<const f32x4*>(p.pshift + c0 + piece * 8 + 4); }
        }
        h8 shi[NSLOT], slo[NSLOT];              // converted before the barrier: the VALU work overlaps the other waves' MFMAs
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (gvox[j] >= 0) {
                const float* q = src + (size_t)gvox[j] * ldc + cpos;
                if (ok0) v0 = *reinterpret_cast<const f32x4*>(q);
                if (ok1) v1 = *reinterpret_cast<const f32x4*>(q + 4);
                if (pro) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (ok0) v0[e] = dm3d_silu(fmaf(v0[e], sc0[e], sh0[e]));
                        if (ok1) v1[e] = dm3d_silu(fmaf(v1[e], sc1[e], sh1[e]));
                    }
                }
            }
            split8(v0, v1, shi[j], slo[j]);
        }
        __syncthreads();                        // all waves are done with the previous chunk's halo and weight buffers
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const int hv = (tid >> 1) + j * 128;
            if (hv < HVOX) {
                _Float16* rec = lds_in + hv * REC + piece * 8;
                *reinterpret_cast<h8*>(rec) = shi[j];
                *reinterpret_cast<h8*>(rec + 16) = slo[j];
            }
        }
        store_w(0);
        __syncthreads();

#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            const bool last_group = g + 1 == NG;
            if (!last_group) fetch_w(ch * NG + g + 1);
            else if (ch + 1 < p.nchunks) fetch_w((ch + 1) * NG);

            const int kd = g / KS, kh = g % KS;                     // KS == 1: g == 0
            const _Float16* wbuf = lds_w + (g & 1) * WGRP;
#pragma unroll
            for (int t = 0; t < G; ++t) {
                const int tap_off = ((kd * HH + kh) * HW + t) * REC;
                h8 ah[MR], al[MR], bh[NR], bl[NR];
#pragma unroll
                for (int mr = 0; mr < MR; ++mr) {
                    const _Float16* q = lds_in + a_off[mr] + tap_off;
                    ah[mr] = *reinterpret_cast<const h8*>(q);
                    al[mr] = *reinterpret_cast<const h8*>(q + 16);
                }
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    const _Float16* q = wbuf + t * (NT * REC) + b_off[nr];
                    bh[nr] = *reinterpret_cast<const h8*>(q);
                    bl[nr] = *reinterpret_cast<const h8*>(q + 16);
                }
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mr], bh[nr], acc[mr][nr], 0, 0, 0);
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mr], bl[nr], acc[mr][nr], 0, 0, 0);
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mr], bh[nr], acc[mr][nr], 0, 0, 0);
                    }
            }
            if (!last_group) {
                store_w((g + 1) & 1);
                __syncthreads();
            }
        }
    }

    const int vrow = p.vec ? (p.vec_idx ? p.vec_idx[b] : b) : 0;
    const int n0 = ntile * NT;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int n = n0 + wn * (NT / WN) + nr * 32 + l32;
        if (n >= p.cout) continue;
        float add = p.bias ? p.bias[n] : 0.0f;
        if (p.vec) add += p.vec[(size_t)vrow * p.vec_ld + n];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (TM / WM) + mr * 32 + dm3d_acc_row(r, half);
                const int oz = oz0 + row / (TH * TW), oy = oy0 + (row / TW) % TH, ox = ox0 + row % TW;
                if (oz < p.od && oy < p.oh && ox < p.ow) {
                    const size_t o = ((((size_t)b * p.od + oz) * p.oh + oy) * p.ow + ox) * p.cout + n;
                    float v = fmaf(acc[mr][nr][r], p.out_scale, add);
                    if (p.relu) v = fmaxf(v, 0.0f);
                    if (p.res) v += p.res[o];
                    p.out[o] = v;
                }
            }
        }
    }
}

template <int TD, int TH, int TW, int S, int KS, int WM, int WN, int MINW>
int launch_h3(ConvArgs& a, hipStream_t st) {
    constexpr int HVOX = ((TD - 1) * S + KS) * ((TH - 1) * S + KS) * ((TW - 1) * S + KS);
    constexpr size_t lds = (size_t)(HVOX * REC + 2 * KS * 64 * REC) * sizeof(_Float16);
    static_assert(lds <= 160 * 1024, "LDS budget");
    a.bd = (a.od + TD - 1) / TD;
    a.bh = (a.oh + TH - 1) / TH;
    a.bw = (a.ow + TW - 1) / TW;
    static bool attr_set = false;
    if (!attr_set) {
        DM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_igemm_h3<TD, TH, TW, S, KS, WM, WN, MINW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid((unsigned)(a.batch * a.bd * a.bh * a.bw), (unsigned)(a.coutpad / 64));
    hipLaunchKernelGGL((conv3d_igemm_h3<TD, TH, TW, S, KS, WM, WN, MINW>), grid, dim3(256), lds, st, a);
    return dm3d_launch_check("conv3d_igemm_h3");
}

// Keras [taps][cin][cout] -> [coutpad/64][cinpad/16][taps][64][REC] halfs: 16 hi | 16 lo | 8 zero, scaled by 2^w_exp
__global__ __launch_bounds__(256) void pack_weights_h3_kernel(const float* __restrict__ w, int taps, int cin, int cout,
                                                              int nchunks, int ntiles, float scale, const float* in_scale,
                                                              _Float16* __restrict__ out) {
    const long nrec = (long)ntiles * nchunks * taps * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nrec * 16; i += (long)gridDim.x * 256) {
        const int k = (int)(i & 15);
        long rec = i >> 4;
        const int nn = (int)(rec % 64);
        const int tap = (int)((rec / 64) % taps);
        const int chunk = (int)((rec / (64L * taps)) % nchunks);
        const int nt = (int)(rec / (64L * taps * nchunks));
        const int ci = chunk * 16 + k, co = nt * 64 + nn;
        float v = 0.f;
        if (ci < cin && co < cout) {
            v = w[((long)tap * cin + ci) * cout + co];
            if (in_scale) v *= in_scale[ci];
            v *= scale;
        }
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        _Float16* r = out + rec * REC;
        r[k] = hi;
        r[16 + k] = lo;
        if (k < 8) r[32 + k] = (_Float16)0.0f;
    }
}

}  // namespace

int dm3d_conv_launch_h3(ConvArgs& a, int which, hipStream_t st) {
    if (which == DM3D_CONV_K1) return launch_h3<4, 8, 8, 1, 1, 4, 1, 2>(a, st);
    if (which == DM3D_CONV_K3S2) return launch_h3<2, 4, 8, 2, 3, 2, 2, 1>(a, st);
    return launch_h3<4, 8, 8, 1, 3, 4, 1, 2>(a, st);
}

extern "C" int64_t dm3d_packed_weight_h3_bytes(int32_t taps, int32_t cin, int32_t cout) {
    if (taps <= 0 || cin <= 0 || cout <= 0) return 0;
    return (int64_t)taps * dm3d_round_up(cout, 64) * (dm3d_round_up(cin, 16) / 16) * REC * (int64_t)sizeof(_Float16);
}

extern "C" int dm3d_pack_weights_h3(const float* keras_kernel, int32_t taps, int32_t cin, int32_t cout, int32_t w_exp,
                                    const float* in_scale, void* packed, void* stream) {
    DM3D_REQUIRE(keras_kernel && packed && taps > 0 && cin > 0 && cout > 0, "pack_weights_h3: bad arguments");
    DM3D_REQUIRE(w_exp >= -100 && w_exp <= 100, "pack_weights_h3: w_exp %d out of range", w_exp);
    DM3D_REQUIRE(dm3d_aligned16(packed), "pack_weights_h3: packed must be 16-byte aligned");
    const int nchunks = (int)(dm3d_round_up(cin, 16) / 16), ntiles = (int)(dm3d_round_up(cout, 64) / 64);
    const long n = (long)ntiles * nchunks * taps * 64 * 16;
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(pack_weights_h3_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream), keras_kernel,
                       taps, cin, cout, nchunks, ntiles, ldexpf(1.0f, w_exp), in_scale, static_cast<_Float16*>(packed));
    return dm3d_launch_check("pack_weights_h3_kernel");
}
