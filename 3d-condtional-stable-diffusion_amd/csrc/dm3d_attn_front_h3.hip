// dm3d_attn_front_h3.hip — the front half of a CrossAttentionBlock in ONE launch (reference: networks/conditional_dm3d.py:186-193 and the
// projections of apply_attention, :163-170):
//     y  = relu(proj_in(norm(x)))                       (:186-189; the inference BatchNormalization is folded into proj_in's weights)
//     n1, n2, n3 = LayerNormalization x 3 of y          (:191-193)
//     q | k = Dense_query|key(n1),  v^T = Dense_value(n1)^T,  q2 = Dense_query(n2)          (the self pass's operands and the context pass's query)
// Outputs: y (float32: the residual of the self-attention pass), q|k, v^T, q2 and n3 (the MLP's input) as DM3D_FMT_H2.  n1 and n2 exist in LDS only.
// As separate launches this was proj_in GEMM + layernorm3 + a grouped GEMM: 98 us per block at B = 32, each a single round of 256 tiles whose
// fixed costs nothing overlaps, with y, n1, n2 round-tripping HBM.
//
// Shape (the scheme of dm3d_mlp_h3.hip).  One workgroup (4 waves, one per SIMD) owns 64 rows.  Every product is computed TRANSPOSED,
// out^T[256 x 64] = W . in^T with K = 256: the weights are the MFMA's A operand and arrive as PRE-TILED operand fragments
// (dm3d_pack_front_weights: 1 KB per (32-row tile, 16-k record, hi | lo), lane-ordered) by plain coalesced 16-byte loads straight into
// registers, two records ahead of their MFMAs — no wave needs another wave's weights, so they never touch LDS; the 64 input rows are the B
// operand, read from an LDS image of DM3D_FMT_H2 records ([record][row][64 B], XOR-swizzled slots; record stride 4 KB + 32 B so that the
// row-major stores of the staging pass spread over the banks).  Wave w owns output columns 64w .. 64w+63 (2 x 2 tiles of 32x32, 64 registers).
// A lane then holds groups of four consecutive columns of one row: 16-byte float32 / 8 + 8-byte H2 stores, no transposes.  v^T is the same
// product with the MFMA operands exchanged (the two fragment layouts are identical), which leaves a lane with four consecutive TOKENS of one
// channel — the orientation v^T is stored in.
// LayerNormalization: the row statistics need all 256 columns of a row, i.e. all four waves: two exchanges of 64 x 4 partial sums through
// LDS (mean, then the centred sum of squares — the two-pass form of dm3d_layernorm3).
#include "dm3d_h3.h"
#include <cstdlib>
#include <type_traits>

namespace {

struct FrontArgs {
    const float* x; long ldx;                   // [m][256] float32
    const void* w_in; const float* b_in;        // tiled [256 x 256]
    const void* w_qk; const float* b_qk;        // tiled [512 x 256]: query rows, then key rows
    const void* w_v; const float* b_v;          // tiled [256 x 256]
    const float* g1; const float* be1; const float* g2; const float* be2; const float* g3; const float* be3;
    float eps;
    float* y; long ldy;
    void* qk; long ldqk;                        // H2 [m][512]
    void* vt; long ldvt;                        // H2 [256][m]
    void* q2; long ldq2;                        // H2 [m][256]
    void* n3; long ldn3;                        // H2 [m][256]
    int m;
    int* range_flag; float range_limit;
};

template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// MR: 32-row tiles per workgroup.  2: 64 rows, one workgroup per CU (134 KB of LDS); 1: 32 rows, 68 KB and half the registers, up to two workgroups
// per CU, at twice the weight traffic per row (the launcher picks by the number of rows).
template <int MR>
__global__ __launch_bounds__(256, MR == 1 ? 2 : 1) void attn_front_h3(const FrontArgs p) {
    constexpr int U = 256, TM = 32 * MR, KR = U / 16, RW = TM / 4;      // RW: rows a wave stages
    constexpr int RS = TM * 64 + 32;                            // bytes between records of the LDS image
    constexpr int REGION = KR * RS;                             // 66 048 bytes
    constexpr int NG = KR / 2;                                  // weight groups per pass: two records = 8 fragments each
    constexpr size_t PASS = 4 * KR * 4 * 1024;                  // bytes of one 256-column pass of a tiled weight image

    extern __shared__ __attribute__((aligned(16))) char smem_f[];
    char* reg_a = smem_f;                                       // x, later n2
    char* reg_b = smem_f + REGION;                              // n1
    float* stat = reinterpret_cast<float*>(smem_f + 2 * REGION);        // [2][4 waves][TM rows]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int m0 = blockIdx.x * TM;

    // hi = f16(x) (RNE), lo = f16(x - hi) of two values (dm3d_h3.h split8's instruction sequence)
    auto split2 = [](float x0, float x1, unsigned int& hi, unsigned int& lo) {
        float r0, r1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
    };
    auto clamp = [](float v) { return __builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f); };
    auto lds_barrier = [&]() {                                  // raw barrier behind this wave's own LDS traffic (weight loads stay in flight across it)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- weight stream: five passes of 8 groups; the loads of a group are issued in front of the MFMAs of the group before it (two register
    // sets), those of a pass's first groups during the previous pass's last groups
    const size_t w_lane = (size_t)wave * (KR * 4 * 1024) + lane * 16;
    const char* const w_in = static_cast<const char*>(p.w_in) + w_lane;
    const char* const w_q = static_cast<const char*>(p.w_qk) + w_lane;
    const char* const w_k = w_q + PASS;
    const char* const w_v = static_cast<const char*>(p.w_v) + w_lane;
    // (WSETS - 1 groups ahead.  Measured: one group and three groups of lead take the same 57 us at B = 32, and so does a timing-only build without
    // any MFMA — a lone workgroup per CU runs its phases serially (36 us for a workgroup alone on the chip: x load and staging, five passes each
    // followed by its stores, two statistics exchanges), and 100 MB of output per launch add the rest; the MFMAs (14 us) hide inside the waits)
    constexpr int WSETS = 2;
    h8 wa[WSETS][8];                                            // [set][record i of the group][column tile nr][hi | lo]
    auto load_w = [&](const int set, const char* wp, const int g) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 8; ++t) wa[set][t] = *reinterpret_cast<const h8*>(wp + (size_t)g * 8192 + t * 1024);
    };
#pragma unroll
    for (int g = 0; g < WSETS - 1; ++g) load_w(g, w_in, g);

    // ---- stage x: 64 rows x 256 float32 -> H2 records in region A.  A wave's load instruction reads one whole row (1 KB); lane l converts
    // columns 4l .. 4l+3: record l >> 2, slot (l >> 1) & 1 (lo: + 2), bytes 8 (l & 1) .. + 7 of the slot.
    {
        f32x4 xv[RW];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int row = m0 + wave * RW + j;
            xv[j] = *reinterpret_cast<const f32x4*>(p.x + (size_t)row * p.ldx + 4 * lane);
        }
        const int rec = lane >> 2, slot = (lane >> 1) & 1, sub = (lane & 1) * 8;
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int r = wave * RW + j, sw = (r >> 2) & 3;
            unsigned int h0, l0, h1, l1;
            split2(clamp(xv[j][0]), clamp(xv[j][1]), h0, l0);
            split2(clamp(xv[j][2]), clamp(xv[j][3]), h1, l1);
            char* rp = reg_a + rec * RS + r * 64 + sub;
            *reinterpret_cast<u32x2*>(rp + ((slot ^ sw) << 4)) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(rp + (((2 + slot) ^ sw) << 4)) = u32x2{l0, l1};
        }
    }
    lds_barrier();

    // B-operand fragment offsets (bytes): logical slot `half` (hi; lo: ^ 32) of this lane's row, swizzled by the row
    unsigned b_off[MR];
#pragma unroll
    for (int t = 0; t < MR; ++t) {
        const int ra = t * 32 + l32;
        b_off[t] = (unsigned)(ra * 64 + ((half ^ ((ra >> 2) & 3)) << 4));
    }
    // The MFMA as the compiler's builtin, NOT inline asm: behind an asm MFMA hipcc's hazard pass knows nothing, and it placed a VALU write to a
    // register of the last MFMA's A operand two instructions behind that MFMA (the address arithmetic of the epilogue's bias loads) — on
    // gfx950 the K = 16 MFMA reads its operands over several passes, and one launch in ~20 000 workgroups stored one wrong element of y
    // (tools/chain_trace.py: two chains of one seed differed in ONE word of y at step 10, everything downstream of that row with it).
#define DM3D_MFMA_VV(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)
#define DM3D_MFMA_DRAIN() do { } while (0)
    // One pass: acc[mr][nr] (+)= W[64w + 32nr ..][:] . in[32mr ..][:]^T over K = 256.  SWAP: the operands exchanged (acc^T: lane = column).
    auto pass = [&](auto SWAP_T, f32x16 (&acc)[MR][2], const char* reg, const char* wp, const char* wnext) __attribute__((always_inline)) {
        constexpr bool SWAP = decltype(SWAP_T)::value;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.0f;
        // the B fragments of a record are requested one record ahead of the MFMAs that use them (two register sets)
        h8 bh[2][MR], bl[2][MR];
        auto read_b = [&](const int set, const int kk) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < MR; ++t) {
                bh[set][t] = *reinterpret_cast<const h8*>(reg + kk * RS + b_off[t]);
                bl[set][t] = *reinterpret_cast<const h8*>(reg + kk * RS + (b_off[t] ^ 32u));
            }
        };
        read_b(0, 0);
        static_for<NG>([&](auto G_) __attribute__((always_inline)) {
            constexpr int g = decltype(G_)::value, set = g & (WSETS - 1);          // (NG % WSETS == 0: every pass starts in set 0)
            constexpr int ga = g + WSETS - 1;                                       // the group requested now: WSETS - 1 groups ahead
            if (ga < NG) load_w(ga & (WSETS - 1), wp, ga); else load_w(ga & (WSETS - 1), wnext, ga - NG);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kk = 2 * g + i, bs = i;                       // (two records per group: the record's parity is i)
                if (kk + 1 < KR) read_b(bs ^ 1, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
                const h8 (&ah)[MR] = bh[bs];
                const h8 (&al)[MR] = bl[bs];
                // pass-major over the four tiles (a 32x32x16 that depends on the one issued just before it waits out its latency)
                if constexpr (!SWAP) {
#pragma unroll
                    for (int t = 0; t < 2 * MR; ++t) DM3D_MFMA_VV(acc[t >> 1][t & 1], wa[set][4 * i + 2 * (t & 1)], al[t >> 1]);
#pragma unroll
                    for (int t = 0; t < 2 * MR; ++t) DM3D_MFMA_VV(acc[t >> 1][t & 1], wa[set][4 * i + 2 * (t & 1) + 1], ah[t >> 1]);
#pragma unroll
                    for (int t = 0; t < 2 * MR; ++t) DM3D_MFMA_VV(acc[t >> 1][t & 1], wa[set][4 * i + 2 * (t & 1)], ah[t >> 1]);
                } else {
#pragma unroll
                    for (int t = 0; t < 2 * MR; ++t) DM3D_MFMA_VV(acc[t >> 1][t & 1], al[t >> 1], wa[set][4 * i + 2 * (t & 1)]);
#pragma unroll
                    for (int t = 0; t < 2 * MR; ++t) DM3D_MFMA_VV(acc[t >> 1][t & 1], ah[t >> 1], wa[set][4 * i + 2 * (t & 1) + 1]);
#pragma unroll
                    for (int t = 0; t < 2 * MR; ++t) DM3D_MFMA_VV(acc[t >> 1][t & 1], ah[t >> 1], wa[set][4 * i + 2 * (t & 1)]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        DM3D_MFMA_DRAIN();
    };
    const std::false_type plain_t;
    const std::true_type swap_t;

    float amax = 0.0f;
    // lane (l32, half): row 32 mr + l32; register 4 gq + j of tile (mr, nr): column n0 + j, n0 = 64 w + 32 nr + 8 gq + 4 half
    auto col0 = [&](int nr, int gq) { return wave * 64 + nr * 32 + 8 * gq + 4 * half; };
    // Four consecutive columns of one row -> 16 bytes of DM3D_FMT_H2.  This lane (half h) holds columns 8 q + 4 h .. + 3 of an 8-column group,
    // its partner lane (half h ^ 1, same row) the other four: v_permlane32_swap exchanges the halves' words so that the half-0 lane ends up
    // with the group's eight hi halves (one whole 16-byte slot of the record) and the half-1 lane with its eight lo halves (the slot 32
    // bytes further): one 16-byte store per lane instead of two 8-byte ones.
    auto pack16 = [&](const float (&v)[4]) __attribute__((always_inline)) {
        unsigned int h0, l0, h1, l1;
        split2(clamp(v[0]), clamp(v[1]), h0, l0);
        split2(clamp(v[2]), clamp(v[3]), h1, l1);
        const auto s0 = __builtin_amdgcn_permlane32_swap(h0, l0, false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(h1, l1, false, false);
        return u32x4{s0[0], s1[0], s0[1], s1[1]};
    };
    // n8: first column of the 8-column group (a multiple of 8)
    auto store_h2 = [&](void* base, long ld, int row, int n8, const float (&v)[4]) __attribute__((always_inline)) {
        char* dst = static_cast<char*>(base) + (size_t)row * ld * 4 + (n8 >> 4) * 64 + ((n8 >> 3) & 1) * 16 + 32 * half;
        *reinterpret_cast<u32x4*>(dst) = pack16(v);
    };
    // the same into an LDS image (the B operand of a later pass)
    auto store_lds = [&](char* reg, int r, int n8, const float (&v)[4]) __attribute__((always_inline)) {
        const int sw = (r >> 2) & 3, slot = ((n8 >> 3) & 1) + 2 * half;
        *reinterpret_cast<u32x4*>(reg + (n8 >> 4) * RS + r * 64 + ((slot ^ sw) << 4)) = pack16(v);
    };
    auto col8 = [&](int nr, int gq) { return wave * 64 + nr * 32 + 8 * gq; };

    // ---- pass 0: y = relu(W_in . x + b_in), kept in registers
    f32x16 yv[MR][2];
    pass(plain_t, yv, reg_a, w_in, w_q);
    {
        f32x4 bv[2][4];
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) bv[nr][gq] = *reinterpret_cast<const f32x4*>(p.b_in + col0(nr, gq));
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const int row = m0 + mr * 32 + l32;
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = fmaxf(yv[mr][nr][4 * gq + j] + bv[nr][gq][j], 0.0f);
                        yv[mr][nr][4 * gq + j] = o[j];
                        DM3D_AMAX(amax, o[j]);
                    }
                    *reinterpret_cast<f32x4*>(p.y + (size_t)row * p.ldy + col0(nr, gq)) = o;
                }
        }
    }
    // ---- LayerNormalization statistics of the 64 rows (two-pass): this lane's 32 columns of its two rows, its partner half's, the four waves'
    float mean[MR], rstd[MR];
    {
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            float a = 0.0f;
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
#pragma unroll
                for (int r = 0; r < 16; ++r) a += yv[mr][nr][r];
            a += __shfl_xor(a, 32);
            if (half == 0) stat[wave * TM + mr * 32 + l32] = a;
        }
        lds_barrier();                                          // (also: every wave has finished reading x — region A is free)
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const int r = mr * 32 + l32;
            mean[mr] = ((stat[r] + stat[TM + r]) + (stat[2 * TM + r] + stat[3 * TM + r])) * (1.0f / U);
            float q = 0.0f;
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
#pragma unroll
                for (int i = 0; i < 16; ++i) { const float d = yv[mr][nr][i] - mean[mr]; q = fmaf(d, d, q); }
            q += __shfl_xor(q, 32);
            if (half == 0) stat[4 * TM + wave * TM + r] = q;
        }
        lds_barrier();
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            const int r = mr * 32 + l32;
            const float var = ((stat[4 * TM + r] + stat[5 * TM + r]) + (stat[6 * TM + r] + stat[7 * TM + r])) * (1.0f / U);
            rstd[mr] = rsqrtf(var + p.eps);
        }
    }
    // normalised rows in place, then the three affine copies: n3 -> HBM (the MLP's input), n1 -> region B, n2 -> region A
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) yv[mr][nr][r] = (yv[mr][nr][r] - mean[mr]) * rstd[mr];
    static_for<3>([&](auto I_) __attribute__((always_inline)) {
        constexpr int which = decltype(I_)::value;              // 0: n3, 1: n1, 2: n2
        const float* gam = which == 0 ? p.g3 : which == 1 ? p.g1 : p.g2;
        const float* bet = which == 0 ? p.be3 : which == 1 ? p.be1 : p.be2;
        f32x4 gv[2][4], tv[2][4];
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                gv[nr][gq] = *reinterpret_cast<const f32x4*>(gam + col0(nr, gq));
                tv[nr][gq] = *reinterpret_cast<const f32x4*>(bet + col0(nr, gq));
            }
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = yv[mr][nr][4 * gq + j] * gv[nr][gq][j] + tv[nr][gq][j]; DM3D_AMAX(amax, v[j]); }
                    if constexpr (which == 0) store_h2(p.n3, p.ldn3, m0 + mr * 32 + l32, col8(nr, gq), v);
                    else store_lds(which == 1 ? reg_b : reg_a, mr * 32 + l32, col8(nr, gq), v);
                }
    });
    lds_barrier();                                              // n1 and n2 are visible

    // ---- passes 1, 2: q | k = W_qk . n1 + b_qk  -> qk[:, 0..255 | 256..511]
    f32x16 acc[MR][2];
    static_for<2>([&](auto P_) __attribute__((always_inline)) {
        constexpr int pi = decltype(P_)::value;
        pass(plain_t, acc, reg_b, pi == 0 ? w_q : w_k, pi == 0 ? w_k : w_v);
        f32x4 bv[2][4];
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) bv[nr][gq] = *reinterpret_cast<const f32x4*>(p.b_qk + pi * U + col0(nr, gq));
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = acc[mr][nr][4 * gq + j] + bv[nr][gq][j]; DM3D_AMAX(amax, v[j]); }
                    store_h2(p.qk, p.ldqk, m0 + mr * 32 + l32, pi * U + col8(nr, gq), v);
                }
    });
    // ---- pass 3: v^T = W_v . n1^T + b_v with the operands exchanged: lane = channel 64 w + 32 nr + l32, register 4 gq + j of tile (mr, nr) =
    // token 32 mr + 8 gq + 4 half + j — four consecutive tokens of one channel: 8 + 8 bytes of v^T's row
    {
        pass(swap_t, acc, reg_b, w_v, w_q);
#pragma unroll
        for (int nr = 0; nr < 2; ++nr) {
            const int ch = wave * 64 + nr * 32 + l32;
            const float bias = p.b_v[ch];
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = acc[mr][nr][4 * gq + j] + bias; DM3D_AMAX(amax, v[j]); }
                    store_h2(p.vt, p.ldvt, ch, m0 + mr * 32 + 8 * gq, v);
                }
        }
    }
    // ---- pass 4: q2 = W_q . n2 + b_q (the context pass's query)
    {
        pass(plain_t, acc, reg_a, w_q, w_q);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the redundant tail loads
        f32x4 bv[2][4];
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) bv[nr][gq] = *reinterpret_cast<const f32x4*>(p.b_qk + col0(nr, gq));
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = acc[mr][nr][4 * gq + j] + bv[nr][gq][j]; DM3D_AMAX(amax, v[j]); }
                    store_h2(p.q2, p.ldq2, m0 + mr * 32 + l32, col8(nr, gq), v);
                }
    }
    if (p.range_flag && amax > p.range_limit) *p.range_flag = 1;
#undef DM3D_MFMA_VV
#undef DM3D_MFMA_DRAIN
}

// DM3D_FMT_H2 weight rows W[n][256] -> operand fragments [pass n / 256][wave][record 16][column tile nr][hi | lo][lane][16 B]: one thread per 16-byte piece
__global__ __launch_bounds__(256) void front_tile_weights_kernel(const char* __restrict__ src, char* __restrict__ dst, int n) {
    const long total = (long)n * 256 * 4 / 16;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long q = i;
        const int lane = (int)(q & 63); q >>= 6;
        const int hilo = (int)(q & 1); q >>= 1;
        const int nr = (int)(q & 1); q >>= 1;
        const int rec = (int)(q & 15); q >>= 4;
        const int wave = (int)(q & 3); q >>= 2;
        const int ps = (int)q;
        const long off = (long)(ps * 256 + wave * 64 + nr * 32 + (lane & 31)) * 1024 + rec * 64 + ((lane >> 5) + 2 * hilo) * 16;
        *reinterpret_cast<f32x4*>(dst + i * 16) = *reinterpret_cast<const f32x4*>(src + off);
    }
}

}  // namespace

extern "C" int dm3d_pack_front_weights(const void* w_h2, int32_t n, int32_t units, void* tiled, void* stream) {
    DM3D_REQUIRE(w_h2 && tiled && dm3d_aligned16(w_h2) && dm3d_aligned16(tiled), "pack_front_weights: null or unaligned pointer");
    DM3D_REQUIRE(units == 256 && n > 0 && n % 256 == 0, "pack_front_weights: n=%d units=%d (units must be 256, n a multiple of 256)", n, units);
    hipLaunchKernelGGL(front_tile_weights_kernel, dim3(256), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const char*>(w_h2),
                       static_cast<char*>(tiled), n);
    return dm3d_launch_check("front_tile_weights_kernel");
}

extern "C" int dm3d_attn_front(const dm3d_attn_front_desc* d, void* stream) {
    DM3D_REQUIRE(d != nullptr, "attn_front: null descriptor");
    DM3D_REQUIRE(d->units == 256, "attn_front: units=%d (this kernel is built for 256: the U-Net's attention width; use dm3d_gemm_tn + dm3d_layernorm3_h2 otherwise)", d->units);
    DM3D_REQUIRE(d->m > 0 && d->m % 64 == 0, "attn_front: m=%d must be a positive multiple of 64", d->m);
    // Tile form.  Measured (tools/front_time.py; us per launch at m = 512 / 2 048 / 8 192 / 16 384): 32-row tiles 24 / 26 / 31 / 60, 64-row
    // tiles 36 / 37 / 41 / 54 — a workgroup's latency is what small launches pay (32-row tiles: two thirds of it, and twice the workgroups), the
    // weight traffic per row what a launch that fills the chip twice over pays.  DM3D_FRONT_MR=1|2 forces a form (read per call: an A/B switch).
    const char* mr_env = getenv("DM3D_FRONT_MR");
    const int mr = (mr_env && (mr_env[0] == '1' || mr_env[0] == '2')) ? mr_env[0] - '0' : (d->m <= 8192 ? 1 : 2);
    DM3D_REQUIRE(d->x && d->w_in && d->b_in && d->w_qk && d->b_qk && d->w_v && d->b_v, "attn_front: x / weights / biases must be non-null");
    DM3D_REQUIRE(d->g1 && d->be1 && d->g2 && d->be2 && d->g3 && d->be3, "attn_front: the three LayerNormalization (gamma, beta) pairs must be non-null");
    DM3D_REQUIRE(d->y && d->qk && d->vt && d->q2 && d->n3, "attn_front: y / qk / vt / q2 / n3 must be non-null");
    DM3D_REQUIRE(d->ldx >= 256 && d->ldx % 4 == 0 && d->ldy >= 256 && d->ldy % 4 == 0, "attn_front: ldx=%lld ldy=%lld", (long long)d->ldx, (long long)d->ldy);
    DM3D_REQUIRE(d->ldqk >= 512 && d->ldqk % 16 == 0 && d->ldq2 >= 256 && d->ldq2 % 16 == 0 && d->ldn3 >= 256 && d->ldn3 % 16 == 0 && d->ldvt >= d->m && d->ldvt % 16 == 0,
                 "attn_front: ldqk=%lld ldq2=%lld ldn3=%lld ldvt=%lld (H2 rows: multiples of 16, at least the row length)", (long long)d->ldqk, (long long)d->ldq2,
                 (long long)d->ldn3, (long long)d->ldvt);
    DM3D_REQUIRE(dm3d_aligned16(d->x) && dm3d_aligned16(d->w_in) && dm3d_aligned16(d->w_qk) && dm3d_aligned16(d->w_v) && dm3d_aligned16(d->b_in) && dm3d_aligned16(d->b_qk)
                 && dm3d_aligned16(d->g1) && dm3d_aligned16(d->be1) && dm3d_aligned16(d->g2) && dm3d_aligned16(d->be2) && dm3d_aligned16(d->g3) && dm3d_aligned16(d->be3)
                 && dm3d_aligned16(d->y) && dm3d_aligned16(d->qk) && dm3d_aligned16(d->vt) && dm3d_aligned16(d->q2) && dm3d_aligned16(d->n3),
                 "attn_front: every pointer must be 16-byte aligned");
    FrontArgs a{};
    a.x = d->x; a.ldx = d->ldx; a.w_in = d->w_in; a.b_in = d->b_in; a.w_qk = d->w_qk; a.b_qk = d->b_qk; a.w_v = d->w_v; a.b_v = d->b_v;
    a.g1 = d->g1; a.be1 = d->be1; a.g2 = d->g2; a.be2 = d->be2; a.g3 = d->g3; a.be3 = d->be3; a.eps = d->eps;
    a.y = d->y; a.ldy = d->ldy; a.qk = d->qk; a.ldqk = d->ldqk; a.vt = d->vt; a.ldvt = d->ldvt; a.q2 = d->q2; a.ldq2 = d->ldq2; a.n3 = d->n3; a.ldn3 = d->ldn3;
    a.m = d->m; a.range_flag = d->range_flag; a.range_limit = d->range_limit > 0.0f ? d->range_limit : 65504.0f;
    const size_t lds = (size_t)2 * 16 * (32 * mr * 64 + 32) + (size_t)2 * 4 * 32 * mr * sizeof(float);       // two operand images + the statistics exchange
    static std::atomic<bool> attr_set[64][2] = {};     // (atomic: two host threads may meet in a first launch; the attribute call itself is idempotent)
    int dev = 0;
    DM3D_HIP(hipGetDevice(&dev));
    DM3D_REQUIRE(dev >= 0 && dev < 64, "attn_front: device ordinal %d", dev);
    if (!attr_set[dev][mr - 1]) {
        if (mr == 2) DM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_front_h3<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        else DM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_front_h3<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev][mr - 1] = true;
    }
    if (mr == 2) hipLaunchKernelGGL(attn_front_h3<2>, dim3((unsigned)(d->m / 64)), dim3(256), lds, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(attn_front_h3<1>, dim3((unsigned)(d->m / 32)), dim3(256), lds, static_cast<hipStream_t>(stream), a);
    return dm3d_launch_check("attn_front_h3");
}
