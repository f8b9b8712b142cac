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
// channels; per 16-channel chunk the input halo lives in LDS as 64-byte records of four 16-byte slots
// [hi c0-7 | hi c8-15 | lo c0-7 | lo c8-15]; the k^3 taps walk it by address only; weight slices arrive in groups of 3 taps
// (one (kd,kh) row), double-buffered, one barrier per group.  One k3 tap = 12 MFMAs (384 cycles) per wave against 8
// ds_read_b128, so LDS bandwidth is the scarce resource and every read is laid out conflict-free:
//   * slot s of record v is stored at physical slot s ^ ((v >> 2) & 3): a 16-lane ds_read_b128 group is conflict-free
//     iff its 16 record indices are distinct mod 16;
//   * the halo row stride is padded from 10 to 12 voxels, which makes every residue mod 16 occur exactly twice in a
//     4 x 8 voxel tile, and the MFMA row -> voxel map (row_to_yx) hands one copy to each of the two hardware lane
//     groups {0-3,12-15,20-27} and {4-11,16-19,28-31}.  Weight rows (record = output channel) are conflict-free as is.
#include "dm3d_conv_args.h"
#include "dm3d_h3.h"

namespace {

constexpr int REC = DM3D_REC;

// MFMA tile row i (0..31) -> (dy, dx) inside a 4 x 8 voxel tile.  PAIR = 0: natural order.  PAIR = 2 / 1: the lanes of
// the first ds_read_b128 lane group {0-3,12-15,20-27} take rows dy in {0,PAIR}, the second group takes the other two.
template <int PAIR>
__device__ __forceinline__ void row_to_yx(int i, int& dy, int& dx) {
    if (PAIR == 0) { dy = i >> 3; dx = i & 7; return; }
    const bool g1 = (i >= 4 && i < 12) || (i >= 16 && i < 20) || i >= 28;
    const int idx = g1 ? (i < 12 ? i - 4 : (i < 20 ? i - 8 : i - 16)) : (i < 4 ? i : (i < 16 ? i - 8 : i - 12));
    const int sel = idx >> 3;
    dx = idx & 7;
    if (PAIR == 2) dy = g1 ? 1 + 2 * sel : 2 * sel;
    else dy = g1 ? 2 + sel : sel;
}

template <int TD, int TH, int TW, int S, int KS, int WM, int WN, int MINW, int NRA = 0>
__global__ __launch_bounds__(256, MINW) void conv3d_igemm_h3(const ConvArgs p) {
    constexpr int CK = 16, NT = 64;
    constexpr int TM = TD * TH * TW;
    constexpr int HD = (TD - 1) * S + KS, HH = (TH - 1) * S + KS, HW = (TW - 1) * S + KS;
    constexpr int HVOX = HD * HH * HW;                      // real halo voxels (staged)
    constexpr bool CF = (S == 1 && TH % 4 == 0 && TW == 8); // conflict-free layout available
    constexpr int HWP = (CF && KS >= 2) ? 12 : HW;          // padded row stride of the LDS image
    constexpr int PAIR = CF ? (KS >= 2 ? 2 : 1) : 0;
    constexpr int HREC = HD * HH * HWP;                     // records in the LDS image
    constexpr int MRSTEP = 4 * S * HWP;                     // records between consecutive 32-row tiles of a wave (4 y rows)
    static_assert(TM / WM <= 32 || MRSTEP % 16 == 0, "row tiles of a wave must share the swizzle term");
    static_assert(TM / WM <= TH * TW, "a wave's rows stay inside one z slice");
    constexpr int TAPS = KS * KS * KS;
    constexpr int G = KS;                       // taps per weight group (one kw row), TAPS / G groups per chunk
    constexpr int NG = TAPS / G;
    // NRA > 0: Cout <= 32*NRA, only the first NRA column tiles are computed (conv_in / conv_out); the weight image keeps 64 rows
    constexpr int MR = TM / WM / 32, NR = NRA > 0 ? NRA : NT / WN / 32;
    constexpr int NSLOT = (HVOX * 2 + 255) / 256;          // (voxel, 8-channel piece) slots per thread
    constexpr int WGRP = G * NT * REC;                      // halfs per weight group
    constexpr int WPIECES = WGRP * 2 / 16;                  // 16-byte pieces per group
    constexpr int WSLOT = WPIECES / 256;
    static_assert(WPIECES % 256 == 0, "a weight group is a whole number of 16-byte pieces per thread (no guards: a guarded "
                                      "load costs a serialized vmcnt(0))");
    static_assert(WM * WN == 4, "kernel assumes 4 waves");
    static_assert(TW == 8 && (TM / WM) % 32 == 0 && (TH * TW) % 32 == 0, "a 32-row MFMA tile is 4 x 8 voxels of one z");

    extern __shared__ __attribute__((aligned(16))) _Float16 smem_h[];
    _Float16* lds_in = smem_h;                  // [HREC][REC]
    _Float16* lds_w = smem_h + HREC * REC;      // [2][G][NT][REC]

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
    int padz = p.padz, pady = p.pady, padx = p.padx, ooz = p.ooz, ooy = p.ooy, oox = p.oox;
    const _Float16* wbase = static_cast<const _Float16*>(p.wpk);
    if (p.parity) {                             // uniform: one output parity of upsample(2) + conv k3 per grid.z slice
        const int par = blockIdx.z;
        ooz = par >> 2; ooy = (par >> 1) & 1; oox = par & 1;
        padz = 1 - ooz; pady = 1 - ooy; padx = 1 - oox;
        wbase += (size_t)par * p.w_parity_stride;
    }

    // staging slots: this thread converts channels [8*piece, 8*piece+8) of halo voxels (tid>>1) + j*128
    const int piece = tid & 1;
    int gvox[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        const int hv = (tid >> 1) + j * 128;
        int g = -1;
        if (hv < HVOX) {
            const int hz = hv / (HH * HW), hy = (hv / HW) % HH, hx = hv % HW;
            const int iz = oz0 * S - padz + hz, iy = oy0 * S - pady + hy, ix = ox0 * S - padx + hx;
            if (iz >= 0 && iz < p.ind && iy >= 0 && iy < p.inh && ix >= 0 && ix < p.inw)
                g = ((b * p.ind + iz) * p.inh + iy) * p.inw + ix;
        }
        gvox[j] = g;
    }

    // LDS write position (halfs) of each staging slot's hi piece; the lo piece sits at the same place ^ 16 halfs (slot ^ 2)
    int st_off[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        const int hv = (tid >> 1) + j * 128;
        const int v = (hv / HW) * HWP + hv % HW;                    // (hz*HH + hy)*HWP + hx
        st_off[j] = hv < HVOX ? v * REC + ((piece ^ swz(v)) << 3) : -1;
    }

    // per-lane record index of the A operand for each 32-row tile at tap (0,0,0)
    int a_rec[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const int r0 = wm * (TM / WM) + mr * 32;
        int dy, dx;
        row_to_yx<PAIR>(l32, dy, dx);
        const int dz = r0 / (TH * TW);
        dy += (r0 / TW) % TH;
        a_rec[mr] = (dz * S * HH + dy * S) * HWP + dx * S;
    }
    // weight rows: record = output channel n; slot half (hi) / 2+half (lo), swizzled by n (tap offsets are multiples of 64)
    int b_hi[NR], b_lo[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int n = wn * (NT / WN) + nr * 32 + l32;
        b_hi[nr] = n * REC + ((half ^ swz(n)) << 3);
        b_lo[nr] = n * REC + (((2 + half) ^ swz(n)) << 3);
    }

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.0f;

    // weight image: [ntile][chunk][group][G][NT][REC] halfs; a group is a dense run of WPIECES 16-byte pieces
    const f32x4* w_img = reinterpret_cast<const f32x4*>(wbase + (size_t)ntile * p.nchunks * NG * WGRP);
    f32x4 wreg[WSLOT];
    auto fetch_w = [&](int group_index) {
        const f32x4* src = w_img + (size_t)group_index * WPIECES;
#pragma unroll
        for (int i = 0; i < WSLOT; ++i) wreg[i] = src[tid + i * 256];
    };
    auto store_w = [&](int buf) {
        f32x4* dst = reinterpret_cast<f32x4*>(lds_w + buf * WGRP);
#pragma unroll
        for (int i = 0; i < WSLOT; ++i) dst[tid + i * 256] = wreg[i];
    };
    fetch_w(0);

    // Halo prefetch state: the raw float32 values of the NEXT chunk travel in registers while the current chunk's MFMAs run.
    // Every load is unconditional on a clamped address and masked afterwards: a load inside a divergent branch (or a
    // prefetch whose result is merged across branches) makes hipcc wait vmcnt(0) on the spot, which serialises the
    // round trips; the sched_barrier after each batch keeps the scheduler from sinking the loads to their first use.
    const bool pro = p.pscale != nullptr;
    f32x4 raw0[NSLOT], raw1[NSLOT];
    f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
    bool ok0 = false, ok1 = false;
    auto load_halo = [&](int ch) {
        const int c0 = ch * CK;
        const float* src;
        int ldc, cb;
        if (c0 < p.c1) { src = p.x1; ldc = p.c1; cb = c0; } else { src = p.x2; ldc = p.c2; cb = c0 - p.c1; }
        const int cpos = cb + piece * 8;
        ok0 = cpos < ldc;
        ok1 = cpos + 4 < ldc;
        const int off0 = ok0 ? cpos : 0, off1 = ok1 ? cpos + 4 : 0;
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const float* q = src + (size_t)(gvox[j] >= 0 ? gvox[j] : 0) * ldc;
            raw0[j] = *reinterpret_cast<const f32x4*>(q + off0);
            raw1[j] = *reinterpret_cast<const f32x4*>(q + off1);
        }
        if (pro) {      // uniform branch; channel offsets clamped the same way (c1+c2 >= 8 is guaranteed by the host)
            const int s0 = ok0 ? c0 + piece * 8 : 0, s1 = ok1 ? c0 + piece * 8 + 4 : 0;
            const size_t bo = (size_t)b * p.pro_bstride;                // 0: one folded-BatchNorm vector; else per-sample (GroupNorm)
            sc0 = *reinterpret_cast<const f32x4*>(p.pscale + bo + s0);
            sh0 = *reinterpret_cast<const f32x4*>(p.pshift + bo + s0);
            sc1 = *reinterpret_cast<const f32x4*>(p.pscale + bo + s1);
            sh1 = *reinterpret_cast<const f32x4*>(p.pshift + bo + s1);
        }
    };
    load_halo(0);

    for (int ch = 0; ch < p.nchunks; ++ch) {
        // convert the prefetched halo (fused silu(x*scale+shift), zero padding, float16 hi/lo split) ...
        h8 shi[NSLOT], slo[NSLOT];
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const bool in = gvox[j] >= 0;
            f32x4 v0 = raw0[j], v1 = raw1[j];
            if (pro) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = dm3d_silu(fmaf(v0[e], sc0[e], sh0[e]));
                    v1[e] = dm3d_silu(fmaf(v1[e], sc1[e], sh1[e]));
                }
            }
            split8(v0, v1, (in && ok0) ? 65504.0f : 0.0f, (in && ok1) ? 65504.0f : 0.0f, shi[j], slo[j]);
        }
        __syncthreads();                        // all waves are done with the previous chunk's halo and weight buffers
        // ... and publish it
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            if (st_off[j] >= 0) {
                *reinterpret_cast<h8*>(lds_in + st_off[j]) = shi[j];
                *reinterpret_cast<h8*>(lds_in + (st_off[j] ^ 16)) = slo[j];
            }
        }
        store_w(0);
        __syncthreads();
        load_halo(ch + 1 < p.nchunks ? ch + 1 : ch);            // in flight for the whole chunk (clamped: never behind a branch)
        __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
        for (int g = 0; g < NG; ++g) {
            const bool last_group = g + 1 == NG;
            {   // prefetch the next weight group (next (kd,kh) row, or row 0 of the next chunk), clamped at the very end
                const int nxt = ch * NG + g + 1, lastg = p.nchunks * NG - 1;
                fetch_w(nxt < lastg ? nxt : lastg);
            }
            __builtin_amdgcn_sched_barrier(0);

            const int kd = g / KS, kh = g % KS;                     // KS == 1: g == 0
            const _Float16* wbuf = lds_w + (g & 1) * WGRP;
#pragma unroll
            for (int t = 0; t < G; ++t) {
                const int tap_rec = (kd * HH + kh) * HWP + t;
                h8 ah[MR], al[MR], bh[NR], bl[NR];
                {   // row tiles of one wave are MRSTEP records apart, a multiple of 16, so they share the swizzle term: one
                    // address computation per tap, the rest are immediate offsets (lo slot = hi slot ^ 2)
                    const int v = a_rec[0] + tap_rec;
                    const int hi_off = v * REC + ((half ^ swz(v)) << 3);
                    const int lo_off = hi_off ^ 16;
#pragma unroll
                    for (int mr = 0; mr < MR; ++mr) {
                        ah[mr] = *reinterpret_cast<const h8*>(lds_in + hi_off + mr * (MRSTEP * REC));
                        al[mr] = *reinterpret_cast<const h8*>(lds_in + lo_off + mr * (MRSTEP * REC));
                    }
                }
#pragma unroll
                for (int nr = 0; nr < NR; ++nr) {
                    bh[nr] = *reinterpret_cast<const h8*>(wbuf + t * (NT * REC) + b_hi[nr]);
                    bl[nr] = *reinterpret_cast<const h8*>(wbuf + t * (NT * REC) + b_lo[nr]);
                }
#ifndef DM3D_MFMA_PASS_MAJOR
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) {
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mr], bh[nr], acc[mr][nr], 0, 0, 0);
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mr], bl[nr], acc[mr][nr], 0, 0, 0);
                        acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mr], bh[nr], acc[mr][nr], 0, 0, 0);
                    }
#else
                // pass-major: the three MFMAs into one accumulator are MR*NR issue slots apart (round 2: measured neutral here, two waves per SIMD fill the slots either way)
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mr], bh[nr], acc[mr][nr], 0, 0, 0);
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mr], bl[nr], acc[mr][nr], 0, 0, 0);
#pragma unroll
                for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                    for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mr], bh[nr], acc[mr][nr], 0, 0, 0);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!last_group) {
                store_w((g + 1) & 1);
                __syncthreads();
            }
        }
    }

    // epilogue: + bias + vec[row(b)] -> relu -> + res -> store.  A wave's rows lie in one z slice, so addresses are a
    // uniform 64-bit slice base plus 32-bit per-lane offsets: accumulator register r = 4q + c of lane half h is the voxel
    // (dy, dx0 + c) given by row_to_yx(8q + 4h).  Loads are unconditional on clamped indices; stores are predicated only
    // in partial bricks (uniform branch).
    const int vrow = p.vec ? (p.vec_idx ? p.vec_idx[b] : b) : 0;
    const int n0 = ntile * NT;
    const bool full = oz0 + TD <= p.od && oy0 + TH <= p.oh && ox0 + TW <= p.ow && n0 + NT <= p.cout;
    const int cstep = p.os * p.cout;                            // distance between x-neighbours of the brick in the output
    float amax = 0.0f;                                          // range guard (include/dm3d.h): largest |value| this lane stores
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const int r0 = wm * (TM / WM) + mr * 32;
        const int oz = oz0 + r0 / (TH * TW), oyb = oy0 + (r0 / TW) % TH;
        const bool z_ok = oz < p.od;
        const size_t zbase = (((size_t)b * p.fd + (z_ok ? oz * p.os + ooz : 0)) * p.fh) * p.fw * p.cout;
        float* outz = p.out + zbase;
        const float* resz = p.res ? p.res + zbase : nullptr;
        // PReLU slope tensor is [fd, fh, fw, cout] without a batch axis: same z-slice offset minus the sample base
        const float* prz = p.prelu ? p.prelu + (zbase - (size_t)b * p.fd * p.fh * p.fw * p.cout) : nullptr;
        int rowoff[4], oyq[4], oxq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int dy, dx;
            row_to_yx<PAIR>(8 * q + 4 * half, dy, dx);
            oyq[q] = oyb + dy;
            oxq[q] = ox0 + dx;
            rowoff[q] = ((oyq[q] * p.os + ooy) * p.fw + oxq[q] * p.os + oox) * p.cout;
        }
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) {
            const int n = n0 + wn * (NT / WN) + nr * 32 + l32;
            const bool n_ok = n < p.cout;
            const int nc = n_ok ? n : p.cout - 1;
            float add = p.bias ? p.bias[nc] : 0.0f;
            if (p.vec) add += p.vec[(size_t)vrow * p.vec_ld + nc];
            if (full) {
                float rv[16];
                if (resz) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) rv[r] = resz[rowoff[r >> 2] + (r & 3) * cstep + n];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = rowoff[r >> 2] + (r & 3) * cstep + n;
                    float v = fmaf(acc[mr][nr][r], p.out_scale, add);
                    if (p.relu) v = fmaxf(v, 0.0f);
                    if (prz) { const float al = prz[o]; v = v > 0.0f ? v : al * v; }
                    if (resz) v += rv[r];
                    if (p.relu_out) v = fmaxf(v, 0.0f);
                    DM3D_AMAX(amax, v);
                    outz[o] = v;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int q = r >> 2, c = r & 3;
                    const bool ok = n_ok && z_ok && oyq[q] < p.oh && oxq[q] + c < p.ow;
                    const int o = ok ? rowoff[q] + c * cstep + n : 0;
                    float v = fmaf(acc[mr][nr][r], p.out_scale, add);
                    if (p.relu) v = fmaxf(v, 0.0f);
                    if (prz) { const float al = prz[o]; v = v > 0.0f ? v : al * v; }
                    if (resz) v += resz[o];
                    if (p.relu_out) v = fmaxf(v, 0.0f);
                    if (ok) { DM3D_AMAX(amax, v); outz[o] = v; }
                }
            }
        }
    }
    if (p.range_flag && amax > p.range_limit) *p.range_flag = 1;
}

template <int TD, int TH, int TW, int S, int KS, int WM, int WN, int MINW, int NRA = 0>
int launch_h3(ConvArgs& a, hipStream_t st) {
    constexpr int HW = (TW - 1) * S + KS;
    constexpr int HWP = (S == 1 && TH % 4 == 0 && TW == 8 && KS >= 2) ? 12 : HW;
    constexpr int HREC = ((TD - 1) * S + KS) * ((TH - 1) * S + KS) * HWP;
    constexpr size_t lds = (size_t)(HREC * REC + 2 * KS * 64 * REC) * sizeof(_Float16);
    static_assert(lds <= 160 * 1024, "LDS budget");
    a.bd = (a.od + TD - 1) / TD;
    a.bh = (a.oh + TH - 1) / TH;
    a.bw = (a.ow + TW - 1) / TW;
    static std::atomic<bool> attr_set[64] = {};            // per device: the attribute belongs to the device the launch goes to
    int dev_ = 0;
    DM3D_HIP(hipGetDevice(&dev_));
    DM3D_REQUIRE(dev_ >= 0 && dev_ < 64, "conv: device ordinal %d", dev_);
    if (!attr_set[dev_]) {
        DM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_igemm_h3<TD, TH, TW, S, KS, WM, WN, MINW, NRA>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev_] = true;
    }
    dim3 grid((unsigned)(a.batch * a.bd * a.bh * a.bw), (unsigned)(a.coutpad / 64), a.parity ? 8u : 1u);
    hipLaunchKernelGGL((conv3d_igemm_h3<TD, TH, TW, S, KS, WM, WN, MINW, NRA>), grid, dim3(256), lds, st, a);
    return dm3d_launch_check("conv3d_igemm_h3");
}

// Keras [taps][cin][cout] -> [coutpad/64][cinpad/16][taps][64][REC] halfs, scaled by 2^w_exp: the LDS image of a weight
// group, i.e. slots (hi c0-7, hi c8-15, lo c0-7, lo c8-15) of output channel n stored at physical slot s ^ swz(n)
__global__ __launch_bounds__(256) void pack_weights_h3_kernel(const float* __restrict__ w, int taps, int cin, int cout,
                                                              int nchunks, int ntiles, float scale, const float* in_scale,
                                                              _Float16* __restrict__ out, int up) {
    // up == 1: w is the 3x3x3 kernel of an UpSample conv; up == 2: the [4,4,4,Cout,Cin] kernel of a Conv3DTranspose(k4,s2).
    // In both cases taps == 8 and 8 parity images follow each other (dm3d_up_weight / dm3d_convt_weight)
    const long nrec = (long)ntiles * nchunks * taps * 64;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < nrec * 16 * (up ? 8 : 1); i0 += (long)gridDim.x * 256) {
        const int par = (int)(i0 / (nrec * 16));
        const long i = i0 % (nrec * 16);
        const int k = (int)(i & 15);
        long rec = i >> 4;
        const int nn = (int)(rec % 64);
        const int tap = (int)((rec / 64) % taps);
        const int chunk = (int)((rec / (64L * taps)) % nchunks);
        const int nt = (int)(rec / (64L * taps * nchunks));
        const int ci = chunk * 16 + k, co = nt * 64 + nn;
        float v = 0.f;
        if (ci < cin && co < cout) {
            v = up == 1 ? dm3d_up_weight(w, cin, cout, par, tap, ci, co)
              : up == 2 ? dm3d_convt_weight(w, cin, cout, par, tap, ci, co) : w[((long)tap * cin + ci) * cout + co];
            if (in_scale) v *= in_scale[ci];
            v *= scale;
        }
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        _Float16* r = out + ((long)par * nrec + rec) * REC;
        const int sw = (nn >> 2) & 3;
        r[(((k >> 3) ^ sw) << 3) + (k & 7)] = hi;
        r[(((2 + (k >> 3)) ^ sw) << 3) + (k & 7)] = lo;
    }
}

}  // namespace

int dm3d_conv_launch_h3(ConvArgs& a, int which, hipStream_t st) {
    if (which == DM3D_CONV_UP) return launch_h3<4, 8, 8, 1, 2, 4, 1, 2>(a, st);
    if (which == DM3D_CONV_K1) return launch_h3<4, 8, 8, 1, 1, 4, 1, 2>(a, st);
    if (which == DM3D_CONV_K3S2) return launch_h3<2, 4, 8, 2, 3, 2, 2, 2>(a, st);
    if (which == DM3D_CONV_K4S2) return launch_h3<2, 4, 8, 2, 4, 2, 2, 1>(a, st);
    if (a.cout <= 32) return launch_h3<4, 8, 8, 1, 3, 4, 1, 2, 1>(a, st);     // conv_in / conv_out: half the MFMAs
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
                       taps, cin, cout, nchunks, ntiles, ldexpf(1.0f, w_exp), in_scale, static_cast<_Float16*>(packed), 0);
    return dm3d_launch_check("pack_weights_h3_kernel");
}

extern "C" int64_t dm3d_packed_weight_up_h3_bytes(int32_t cin, int32_t cout) { return 8 * dm3d_packed_weight_h3_bytes(8, cin, cout); }

extern "C" int dm3d_pack_weights_up_h3(const float* keras_kernel, int32_t cin, int32_t cout, int32_t w_exp, void* packed,
                                       void* stream) {
    DM3D_REQUIRE(keras_kernel && packed && cin > 0 && cout > 0, "pack_weights_up_h3: bad arguments");
    DM3D_REQUIRE(w_exp >= -100 && w_exp <= 100 && dm3d_aligned16(packed), "pack_weights_up_h3: w_exp out of range or packed unaligned");
    if (dm3d_conv_weight_layout(3, 1, 1, 0, cout) == DM3D_WL_PAIR)
        return dm3d_pack_h3v2(keras_kernel, 8, cin, cout, w_exp, nullptr, packed, 1, static_cast<hipStream_t>(stream));
    const int nchunks = (int)(dm3d_round_up(cin, 16) / 16), ntiles = (int)(dm3d_round_up(cout, 64) / 64);
    hipLaunchKernelGGL(pack_weights_h3_kernel, dim3(4096), dim3(256), 0, static_cast<hipStream_t>(stream), keras_kernel, 8, cin,
                       cout, nchunks, ntiles, ldexpf(1.0f, w_exp), nullptr, static_cast<_Float16*>(packed), 1);
    return dm3d_launch_check("pack_weights_h3_kernel(up)");
}

extern "C" int dm3d_pack_weights_convt_h3(const float* keras_kernel, int32_t cin, int32_t cout, int32_t w_exp, void* packed,
                                          void* stream) {
    DM3D_REQUIRE(keras_kernel && packed && cin > 0 && cout > 0, "pack_weights_convt_h3: bad arguments");
    DM3D_REQUIRE(w_exp >= -100 && w_exp <= 100 && dm3d_aligned16(packed), "pack_weights_convt_h3: w_exp out of range or packed unaligned");
    if (dm3d_conv_weight_layout(4, 2, 0, 1, cout) == DM3D_WL_PAIR)
        return dm3d_pack_h3v2(keras_kernel, 8, cin, cout, w_exp, nullptr, packed, 2, static_cast<hipStream_t>(stream));
    const int nchunks = (int)(dm3d_round_up(cin, 16) / 16), ntiles = (int)(dm3d_round_up(cout, 64) / 64);
    hipLaunchKernelGGL(pack_weights_h3_kernel, dim3(4096), dim3(256), 0, static_cast<hipStream_t>(stream), keras_kernel, 8, cin,
                       cout, nchunks, ntiles, ldexpf(1.0f, w_exp), nullptr, static_cast<_Float16*>(packed), 2);
    return dm3d_launch_check("pack_weights_h3_kernel(convt)");
}
