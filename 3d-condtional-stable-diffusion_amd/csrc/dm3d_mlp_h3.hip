// dm3d_mlp_h3.hip — the MLP of a CrossAttentionBlock in ONE launch: out = Dense_1(relu(Dense_0(x))) + res + res2 with the 4u-wide hidden
// activation never leaving the CU (reference: networks/conditional_dm3d.py:132-133 `keras.Sequential([Dense(units * 4, relu), Dense(units)])`
// applied at :194, and the two residual adds of :193-195).  Split-float16 arithmetic with float32-grade results as everywhere (dm3d_h3.h).
//
// Why.  As two GEMMs the hidden tensor (B*L x 4u float32-sized H2 records: 67 MB per block at B = 32) is written and read back, and each
// GEMM re-stages both of its operands per tile; the K = 256 GEMMs of the block are bound by exactly that staging.
//
// Shape.  One workgroup (4 waves, one per SIMD) owns 64 rows of x.  Its x tile — 64 rows x u = 256 columns — lives in REGISTERS for the
// whole launch (the 256 accumulation registers of the wave: the MFMA reads an operand from them directly; accumulators and the other
// fragments take the architectural half), in MFMA operand layout.  The hidden axis is walked in slabs of 128 columns:
//   phase 1   H^T[128 x 64] = relu(W0[slab] . x^T + b0)    K = u:    wave w owns hidden columns 32w .. 32w+31 of the slab (1 x 2 MFMA tiles)
//             -> split and stored to LDS as H2 records (the B operand of phase 2)
//   phase 2   acc_out^T[u x 64] += W1[:, slab] . H^T        K = 128:  wave w owns output columns 64w .. 64w+63 (2 x 2 tiles, 64 registers)
// Both products run transposed (weights = the MFMA's A operand): a lane holds groups of four consecutive COLUMNS of one row, so the hidden
// slab is split two values per instruction and stored 8 bytes at a time, and the output leaves in 16-byte accesses.
// Weights are PRE-TILED into operand fragments (dm3d_pack_mlp_weights, once per weight set): the 1 KB a wave needs for one (32-row tile,
// 16-k record, hi | lo) are contiguous, lane-ordered.  Every wave consumes weights nobody else in the workgroup needs, so they do not pass
// through LDS at all: plain coalesced 16-byte global loads (8 cache lines per instruction) straight into the fragment registers, one group
// of records ahead of the MFMAs that use them.  (First form of this kernel: weights as DM3D_FMT_H2 rows through a ring of LDS granules by
// LDS-DMA — 2 048 one-KB DMA instructions per workgroup at 100+ cycles of issue each, as long as the MFMAs themselves, plus a barrier per
// granule: 78 us per block against 98 for the two GEMMs; in-kernel stamps put a granule at 1 900 cycles for 768 of MFMA.)
// The only workgroup-wide synchronisation left is the hand-over of the H slab: two barriers per slab.
// Optional tail (MlpArgs.w2): the block's proj_out — relu(W2 . a3 + b2) + the block input — as one more K = 256 product in the phase-2 form, with
// the MLP's result a3 handed over through LDS instead of HBM (conditional_dm3d.py:195).
#include "dm3d_h3.h"
#include <cstdlib>
#include <type_traits>

// Diagnostic build only (-DDM3D_MLP_STAMPS; the product library carries none of it): thread 0 of the first 256 workgroups writes s_memtime at
// phase boundaries into a buffer of its own (tools/mlp_time.py stamps).
#ifdef DM3D_MLP_STAMPS
__device__ unsigned long long* g_dbg_stamps_m = nullptr;
extern "C" int dm3d_debug_set_stamps_mlp(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_stamps_m), &p, sizeof(p)); }
#define MSTAMP(i) do { if (g_dbg_stamps_m && threadIdx.x == 0 && blockIdx.x < 256) g_dbg_stamps_m[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MSTAMP(i) do { } while (0)
#endif

namespace {


struct MlpArgs {
    const void* x; long ldx;                    // [m][u] DM3D_FMT_H2 (ld in elements of 4 bytes)
    const void* w0; const float* b0;            // [4u][u] H2, [4u]
    const void* w1; const float* b1;            // [u][4u] H2, [u]
    const float* res; const float* res2; long ldr;
    void* out; long ldo; int out_h2;
    int m;
    int* range_flag; float range_limit;
    const void* w2; const float* b2;            // optional tail: out = relu(W2 . a3 + b2) + res3 with a3 = the MLP's result (never stored)
    const float* res3; long ldr3;
};

template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

template <int U>
__global__ __launch_bounds__(256, 1) void mlp_fused_h3(const MlpArgs p) {
    constexpr int HID = 4 * U, TM = 64, HS = 128, NSLAB = HID / HS;
    constexpr int KR = U / 16;                                  // records per x / W0 row (16)
    constexpr int HR = HS / 16;                                 // records of a slab's hidden columns (8)
    constexpr int T_RS = TM * 64 + 32;                          // record stride of the tail's a3 image (16 records: 66 048 bytes)
    static_assert(U == 256, "built for the U-Net's attention width");
    static_assert(16 * T_RS <= 64 * 1040, "the tail's a3 image overlays the x staging area");

    extern __shared__ __attribute__((aligned(16))) _Float16 smem_m[];
    _Float16* lds_h = smem_m;                                   // [8 records][64 rows][REC]   32 KB (the x staging area of the prologue overlays it: 65 KB)

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, l32 = lane & 31;
    const int m0 = blockIdx.x * TM;

    // ---- weight fragments (dm3d_pack_mlp_weights): w0t [slab][wave][record kk][hi | lo][lane][16 B], w1t [slab][wave][record][nr][hi | lo][lane][16 B]
    const char* w0_lane = static_cast<const char*>(p.w0) + (size_t)wave * (KR * 2 * 1024) + lane * 16;
    const char* w1_lane = static_cast<const char*>(p.w1) + (size_t)wave * (HR * 2 * 2 * 1024) + lane * 16;
    constexpr size_t W0_SLAB = 4 * KR * 2 * 1024, W1_SLAB = 4 * HR * 2 * 2 * 1024;       // bytes per slab
    MSTAMP(0);
    // ---- the x tile into registers, through LDS: read straight from global memory in operand layout a wave's load touches 64 cache lines
    // (20 800 cycles for the tile, in-kernel stamps); as 64 one-KB row copies by LDS-DMA (row pitch 1040 bytes: the operand reads of 32
    // consecutive rows then spread over the banks) and 64 ds_read_b128 per lane it is a tenth of that.  Rows past m re-read the last row.
    {
        constexpr int XP = 1040;
        char* lds_x = reinterpret_cast<char*>(smem_m);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = m0 + wave * 16 + j;
            const char* src = static_cast<const char*>(p.x) + (size_t)(row < p.m ? row : p.m - 1) * p.ldx * 4 + lane * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds_x + (wave * 16 + j) * XP), 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0070);                    // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    h8 xh[2][KR], xl[2][KR];
#pragma unroll
    for (int mr = 0; mr < 2; ++mr) {
        const char* xr = reinterpret_cast<const char*>(smem_m) + (mr * 32 + l32) * 1040 + half * 16;
#pragma unroll
        for (int kk = 0; kk < KR; ++kk) {
            xh[mr][kk] = *reinterpret_cast<const h8*>(xr + kk * 64);
            xl[mr][kk] = *reinterpret_cast<const h8*>(xr + kk * 64 + 32);
        }
    }
    // (pinned in the accumulation file: left to itself hipcc keeps half of the tile there anyway and copies four registers back in front
    // of every MFMA, and moves the accumulators between the two files at every granule)
#pragma unroll
    for (int mr = 0; mr < 2; ++mr)
#pragma unroll
        for (int kk = 0; kk < KR; ++kk) { asm volatile("" : "+a"(xh[mr][kk])); asm volatile("" : "+a"(xl[mr][kk])); }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();                              // everyone has its copy: the staging area becomes the H slab and the ring
    asm volatile("" ::: "memory");
    // x from the accumulation file; accumulators and the other fragments architectural.  Inline asm: hipcc's hazard pass does not see an
    // MFMA in it, so the two pads it would insert are written here (rules and their measurement: tools/isa_hazard.py, tools/micro/mfma_hazards.hip,
    // profiles/r05_mfma_hazards.log; the build runs the checker over this file's ISA, tests/test_host.py):
    //  * `s_nop 1` INSIDE every MFMA statement, in front of the MFMA: a VALU write of A, B or C needs two wait states before the MFMA reads
    //    it, and hipcc materialises the copies of a "+v" accumulator (v_mov_b64 of the zeroed tile) directly in front of the statement —
    //    a stand-alone fence would sit in front of those copies.  Free when the matrix pipe is the pace (8 + 8 of a 32-cycle 32x32x16).
    //  * DM3D_MFMA_DRAIN (P + 4 = 12 wait states, here 32) in front of every vector read of an accumulator.
    // (Round 4 read the failure this guards against as a write-after-read on A / B behind the MFMA; the probe shows that one does not exist
    // on gfx950 — operands are read at issue — and that the read-after-write in FRONT of it does.)
#define DM3D_MFMA_VX(acc, a, b) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b))
#define DM3D_MFMA_VV(acc, a, b) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define DM3D_MFMA_DRAIN() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

    f32x16 acc_out[2][2];                                       // [row tile mr][column tile nr]: lane = row 32 mr + l32, register r = column 32 nr + (r & 3) + 8 (r >> 2) + 4 half of the wave's 64
#pragma unroll
    for (int mr = 0; mr < 2; ++mr)
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_out[mr][nr][r] = 0.0f;

    // H fragment offsets (bytes): logical slot `half` (hi; lo: ^ 32) of this lane's row, swizzled by the row
    unsigned a2_off[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int ra = t * 32 + l32;
        a2_off[t] = (unsigned)(ra * 64 + ((half ^ ((ra >> 2) & 3)) << 4));
    }
    const char* lds_hc = reinterpret_cast<const char*>(lds_h);
    auto lds_barrier = [&]() {                                  // raw barrier behind this wave's own LDS traffic (the weight loads stay in flight across it)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // hi = f16(x) (RNE), lo = f16(x - hi) of two values (dm3d_h3.h split8's instruction sequence)
    auto split2 = [](float x0, float x1, unsigned int& hi, unsigned int& lo) {
        float r0, r1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(x0), "v"(x1));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "v"(x0));
        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "v"(x1));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
    };
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

    // Groups of records: phase 1 in four groups of four records (8 fragment loads), phase 2 in four groups of two records (8 loads); the
    // loads of group g + 1 are issued in front of the MFMAs of group g (two register sets), those of a phase's first group during the
    // previous phase's last group: the stream never stops at a phase or slab boundary.  (Two groups of lead — three live sets — spilled
    // registers and measured 8 % slower.)
    constexpr int G1 = 4, R1 = KR / G1, G2 = 4, R2 = HR / G2;
    h8 wa[2][8];                                                // two sets of 8 fragments
    auto load_p1 = [&](const int set, int slab, const int g) {  // W0 records 4g .. 4g+3 of the slab: [i][hi | lo]
        const char* src = w0_lane + (size_t)(slab < NSLAB ? slab : NSLAB - 1) * W0_SLAB + (size_t)(g * R1) * 2048;
#pragma unroll
        for (int i = 0; i < R1; ++i) {
            wa[set][2 * i] = *reinterpret_cast<const h8*>(src + i * 2048);
            wa[set][2 * i + 1] = *reinterpret_cast<const h8*>(src + i * 2048 + 1024);
        }
    };
    auto load_p2 = [&](const int set, int slab, const int g) {  // W1 records 2g, 2g+1 of the slab: [i][nr][hi | lo]
        const char* src = w1_lane + (size_t)slab * W1_SLAB + (size_t)(g * R2) * 4096;
#pragma unroll
        for (int i = 0; i < R2; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) wa[set][4 * i + t] = *reinterpret_cast<const h8*>(src + i * 4096 + t * 1024);
    };
    load_p1(0, 0, 0);
    MSTAMP(1);
    for (int slab = 0; slab < NSLAB; ++slab) {
        if (slab == 1) MSTAMP(2);
        if (slab == 2) MSTAMP(13);
        // ---- phase 1: H slab^T = relu(W0[slab] . x^T + b0): lane = row, registers = the wave's 32 hidden columns
        f32x16 acc_h[2];
#pragma unroll
        for (int mr = 0; mr < 2; ++mr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc_h[mr][r] = 0.0f;
        f32x4 b0v[4];                                           // bias of columns 8 gq + 4 half .. + 3 of the wave's 32
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) b0v[gq] = *reinterpret_cast<const f32x4*>(p.b0 + slab * HS + wave * 32 + 8 * gq + 4 * half);
        static_for<G1>([&](auto G_) {
            constexpr int g = decltype(G_)::value, set = g & 1;
            if (slab == 1) MSTAMP(3 + g);
            if (g + 1 < G1) load_p1(set ^ 1, slab, g + 1); else load_p2(set ^ 1, slab, 0);
            __builtin_amdgcn_sched_barrier(0);
            // pass-major per record: consecutive MFMAs go to different accumulators (a 32x32x16 that depends on the one issued just before
            // it waits out its latency, and no second wave fills the hole)
#pragma unroll
            for (int i = 0; i < R1; ++i) {
                const int kk = g * R1 + i;
                DM3D_MFMA_VX(acc_h[0], wa[set][2 * i + 1], xh[0][kk]);
                DM3D_MFMA_VX(acc_h[1], wa[set][2 * i + 1], xh[1][kk]);
                DM3D_MFMA_VX(acc_h[0], wa[set][2 * i], xl[0][kk]);
                DM3D_MFMA_VX(acc_h[1], wa[set][2 * i], xl[1][kk]);
                DM3D_MFMA_VX(acc_h[0], wa[set][2 * i], xh[0][kk]);
                DM3D_MFMA_VX(acc_h[1], wa[set][2 * i], xh[1][kk]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        // hidden columns n0 .. n0 + 3, n0 = 32 w + 8 gq + 4 half, of row 32 mr + l32: record n0 >> 4, slot (n0 >> 3) & 1 (lo: + 2), bytes
        // 8 half .. + 7 of the slot: the four hi halves, then the four lo halves, 8 bytes each
        {
            if (slab == 1) MSTAMP(7);
            DM3D_MFMA_DRAIN();
            lds_barrier();                                      // everyone has left the previous slab's H (phase 2 reads)
#pragma unroll
            for (int mr = 0; mr < 2; ++mr) {
                const int row = mr * 32 + l32, sw = (row >> 2) & 3;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_fmed3f(acc_h[mr][4 * gq + j] + b0v[gq][j], 0.0f, 65504.0f);     // relu, and the float16 range
                    unsigned int h0, l0, h1, l1;
                    split2(v[0], v[1], h0, l0);
                    split2(v[2], v[3], h1, l1);
                    const u32x2 hi = {h0, h1}, lo = {l0, l1};
                    char* rec = reinterpret_cast<char*>(lds_h) + ((2 * wave + (gq >> 1)) * TM + row) * 64 + 8 * half;
                    *reinterpret_cast<u32x2*>(rec + (((gq & 1) ^ sw) << 4)) = hi;
                    *reinterpret_cast<u32x2*>(rec + (((2 + (gq & 1)) ^ sw) << 4)) = lo;
                }
            }
            lds_barrier();                                      // the H slab is visible
            if (slab == 1) MSTAMP(8);
        }
        // ---- phase 2: acc_out^T += W1[:, slab] . H^T
        static_for<G2>([&](auto G_) {
            constexpr int g = decltype(G_)::value, set = g & 1;                  // (G1 is even: the first group of phase 2 sits in set 0)
            if (slab == 1) MSTAMP(9 + g);
            if (g + 1 < G2) load_p2(set ^ 1, slab, g + 1); else load_p1(set ^ 1, slab + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < R2; ++i) {
                const int kk = g * R2 + i;
                h8 ah[2], al[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ah[t] = *reinterpret_cast<const h8*>(lds_hc + kk * (TM * 64) + a2_off[t]);
                    al[t] = *reinterpret_cast<const h8*>(lds_hc + kk * (TM * 64) + (a2_off[t] ^ 32u));
                }
                // pass-major over the four tiles; A = the W1 fragment (column tile nr: [nr][hi | lo]), B = the H fragment (row tile mr)
#pragma unroll
                for (int t = 0; t < 4; ++t) DM3D_MFMA_VV(acc_out[t >> 1][t & 1], wa[set][4 * i + 2 * (t & 1)], al[t >> 1]);
#pragma unroll
                for (int t = 0; t < 4; ++t) DM3D_MFMA_VV(acc_out[t >> 1][t & 1], wa[set][4 * i + 2 * (t & 1) + 1], ah[t >> 1]);
#pragma unroll
                for (int t = 0; t < 4; ++t) DM3D_MFMA_VV(acc_out[t >> 1][t & 1], wa[set][4 * i + 2 * (t & 1)], ah[t >> 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    MSTAMP(14);
    DM3D_MFMA_DRAIN();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the redundant tail loads
    if (p.w2) lds_barrier();                                    // the tail's a3 image overlays the H slab: every wave must be past its last phase-2 read

    // ---- epilogue: + b1 + res + res2 -> float32 or DM3D_FMT_H2.  Lane (l32, half) holds row 32 mr + l32 and, per group gq of four
    // registers, columns n0 .. n0 + 3, n0 = 64 w + 32 nr + 8 gq + 4 half: 16-byte residual loads and stores (H2: 8 + 8 bytes).  Every load is
    // unconditional (an absent residual reads the bias vector instead and is discarded) and all of a row tile's loads go out before the first use.
    float amax = 0.0f;
    const bool has_r = p.res != nullptr, has_r2 = p.res2 != nullptr;
#pragma unroll
    for (int mr = 0; mr < 2; ++mr) {
        const int row = m0 + mr * 32 + l32;
        const size_t rrow = (size_t)(row < p.m ? row : p.m - 1);
        f32x4 rv[2][4], rv2[2][4], bv[2][4];
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n0 = wave * 64 + nr * 32 + 8 * gq + 4 * half;
                bv[nr][gq] = *reinterpret_cast<const f32x4*>(p.b1 + n0);
                rv[nr][gq] = *reinterpret_cast<const f32x4*>(has_r ? p.res + rrow * p.ldr + n0 : p.b1 + n0);
                rv2[nr][gq] = *reinterpret_cast<const f32x4*>(has_r2 ? p.res2 + rrow * p.ldr + n0 : p.b1 + n0);
            }
#pragma unroll
        for (int nr = 0; nr < 2; ++nr)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n0 = wave * 64 + nr * 32 + 8 * gq + 4 * half;
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = ((acc_out[mr][nr][4 * gq + j] + bv[nr][gq][j]) + (has_r ? rv[nr][gq][j] : 0.0f)) + (has_r2 ? rv2[nr][gq][j] : 0.0f);
                    DM3D_AMAX(amax, o[j]);
                }
                if (p.w2) {
                    // the tail's B operand: this row's four columns as H2 into the LDS image [record][row][64 B] (record stride 4 KB + 32 B,
                    // slots XOR-swizzled by the row: dm3d_attn_front_h3.hip's layout).  Rows past m hold the last row's values: never stored.
                    unsigned int h0, l0, h1, l1;
                    split2(__builtin_amdgcn_fmed3f(o[0], -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(o[1], -65504.0f, 65504.0f), h0, l0);
                    split2(__builtin_amdgcn_fmed3f(o[2], -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(o[3], -65504.0f, 65504.0f), h1, l1);
                    const int r = mr * 32 + l32, sw = (r >> 2) & 3, slot = (n0 >> 3) & 1;
                    char* rp = reinterpret_cast<char*>(smem_m) + (n0 >> 4) * T_RS + r * 64 + (n0 & 7) * 2;
                    *reinterpret_cast<u32x2*>(rp + ((slot ^ sw) << 4)) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2*>(rp + (((2 + slot) ^ sw) << 4)) = u32x2{l0, l1};
                } else if (row < p.m) {
                    if (p.out_h2) {
                        unsigned int h0, l0, h1, l1;
                        split2(__builtin_amdgcn_fmed3f(o[0], -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(o[1], -65504.0f, 65504.0f), h0, l0);
                        split2(__builtin_amdgcn_fmed3f(o[2], -65504.0f, 65504.0f), __builtin_amdgcn_fmed3f(o[3], -65504.0f, 65504.0f), h1, l1);
                        const u32x2 hi = {h0, h1}, lo = {l0, l1};
                        char* dst = static_cast<char*>(p.out) + (size_t)row * p.ldo * 4 + (n0 >> 4) * 64 + ((n0 >> 3) & 1) * 16 + (n0 & 7) * 2;
                        *reinterpret_cast<u32x2*>(dst) = hi;
                        *reinterpret_cast<u32x2*>(dst + 32) = lo;
                    } else {
                        *reinterpret_cast<f32x4*>(static_cast<char*>(p.out) + ((size_t)row * p.ldo + n0) * 4) = o;
                    }
                }
            }
    }
    // ---- optional tail (the block's proj_out, conditional_dm3d.py:195: Conv3D(units, 1, relu) on a3, + the block input): one more K = 256
    // product in the phase-2 form — weights as pre-tiled fragments (dm3d_pack_front_weights) by plain loads, a3 from the LDS image above
    if (p.w2) {
        lds_barrier();                                          // (every wave is past its last H read; the a3 image is complete)
        const char* w2_lane = static_cast<const char*>(p.w2) + (size_t)wave * (KR * 4 * 1024) + lane * 16;
        auto load_t = [&](const int set, const int g) {
#pragma unroll
            for (int t = 0; t < 8; ++t) wa[set][t] = *reinterpret_cast<const h8*>(w2_lane + (size_t)(g < KR / 2 ? g : KR / 2 - 1) * 8192 + t * 1024);
        };
        load_t(0, 0);
#pragma unroll
        for (int mr = 0; mr < 2; ++mr)
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc_out[mr][nr][r] = 0.0f;
        static_for<KR / 2>([&](auto G_) {
            constexpr int g = decltype(G_)::value, set = g & 1;
            load_t(set ^ 1, g + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kk = 2 * g + i;
                h8 ah[2], al[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    ah[t] = *reinterpret_cast<const h8*>(lds_hc + kk * T_RS + a2_off[t]);
                    al[t] = *reinterpret_cast<const h8*>(lds_hc + kk * T_RS + (a2_off[t] ^ 32u));
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) DM3D_MFMA_VV(acc_out[t >> 1][t & 1], wa[set][4 * i + 2 * (t & 1)], al[t >> 1]);
#pragma unroll
                for (int t = 0; t < 4; ++t) DM3D_MFMA_VV(acc_out[t >> 1][t & 1], wa[set][4 * i + 2 * (t & 1) + 1], ah[t >> 1]);
#pragma unroll
                for (int t = 0; t < 4; ++t) DM3D_MFMA_VV(acc_out[t >> 1][t & 1], wa[set][4 * i + 2 * (t & 1)], ah[t >> 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        DM3D_MFMA_DRAIN();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const bool has_r3 = p.res3 != nullptr;
#pragma unroll
        for (int mr = 0; mr < 2; ++mr) {
            const int row = m0 + mr * 32 + l32;
            const size_t rrow = (size_t)(row < p.m ? row : p.m - 1);
            f32x4 rv[2][4], bv[2][4];
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int n0 = wave * 64 + nr * 32 + 8 * gq + 4 * half;
                    bv[nr][gq] = *reinterpret_cast<const f32x4*>(p.b2 + n0);
                    rv[nr][gq] = *reinterpret_cast<const f32x4*>(has_r3 ? p.res3 + rrow * p.ldr3 + n0 : p.b2 + n0);
                }
#pragma unroll
            for (int nr = 0; nr < 2; ++nr)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int n0 = wave * 64 + nr * 32 + 8 * gq + 4 * half;
                    f32x4 o;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[j] = fmaxf(acc_out[mr][nr][4 * gq + j] + bv[nr][gq][j], 0.0f) + (has_r3 ? rv[nr][gq][j] : 0.0f);
                        DM3D_AMAX(amax, o[j]);
                    }
                    if (row < p.m) *reinterpret_cast<f32x4*>(static_cast<char*>(p.out) + ((size_t)row * p.ldo + n0) * 4) = o;
                }
        }
    }
    if (p.range_flag && amax > p.range_limit) *p.range_flag = 1;
    MSTAMP(15);
}

// DM3D_FMT_H2 weight rows -> operand fragments: one thread per 16-byte piece of the tiled image (layouts at the top of mlp_fused_h3)
__global__ __launch_bounds__(256) void mlp_tile_weights_kernel(const char* __restrict__ src, char* __restrict__ dst, int units, int which) {
    const int hid = 4 * units, kr = units / 16;
    const long total = (long)units * hid * 4 / 16;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        long q = i;
        const int lane = (int)(q & 63); q >>= 6;
        const int hilo = (int)(q & 1); q >>= 1;
        long off;
        if (which == 0) {                   // W0 [hid][units]: [slab][wave][record kk][hi | lo][lane]
            const int kk = (int)(q % kr); q /= kr;
            const int wave = (int)(q & 3); q >>= 2;
            const int slab = (int)q;
            off = (long)(slab * 128 + wave * 32 + (lane & 31)) * (units * 4) + kk * 64 + ((lane >> 5) + 2 * hilo) * 16;
        } else {                            // W1 [units][hid]: [slab][wave][record][nr][hi | lo][lane]
            const int nr = (int)(q & 1); q >>= 1;
            const int rec = (int)(q & 7); q >>= 3;
            const int wave = (int)(q & 3); q >>= 2;
            const int slab = (int)q;
            off = (long)(wave * 64 + nr * 32 + (lane & 31)) * (hid * 4) + (long)(slab * 8 + rec) * 64 + ((lane >> 5) + 2 * hilo) * 16;
        }
        *reinterpret_cast<f32x4*>(dst + i * 16) = *reinterpret_cast<const f32x4*>(src + off);
    }
}

}  // namespace

extern "C" int dm3d_pack_mlp_weights(const void* w_h2, int32_t units, int32_t which, void* tiled, void* stream) {
    DM3D_REQUIRE(w_h2 && tiled && dm3d_aligned16(w_h2) && dm3d_aligned16(tiled), "pack_mlp_weights: null or unaligned pointer");
    DM3D_REQUIRE(units == 256 && (which == 0 || which == 1), "pack_mlp_weights: units=%d which=%d (units must be 256, which 0 | 1)", units, which);
    hipLaunchKernelGGL(mlp_tile_weights_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const char*>(w_h2),
                       static_cast<char*>(tiled), units, which);
    return dm3d_launch_check("mlp_tile_weights_kernel");
}

extern "C" int dm3d_mlp_fused(const dm3d_mlp_desc* d, void* stream) {
    DM3D_REQUIRE(d != nullptr, "mlp_fused: null descriptor");
    DM3D_REQUIRE(d->x && d->w0 && d->b0 && d->w1 && d->b1 && d->out, "mlp_fused: x / w0 / b0 / w1 / b1 / out must be non-null");
    DM3D_REQUIRE(d->units == 256, "mlp_fused: units=%d (this kernel is built for 256: the U-Net's attention width; use dm3d_gemm_tn twice otherwise)", d->units);
    DM3D_REQUIRE(d->m > 0 && d->ldx >= d->units && d->ldx % 16 == 0 && d->ldo >= d->units, "mlp_fused: m=%d ldx=%lld ldo=%lld", d->m, (long long)d->ldx, (long long)d->ldo);
    DM3D_REQUIRE(d->out_fmt == DM3D_FMT_F32 || (d->out_fmt == DM3D_FMT_H2 && d->ldo % 16 == 0), "mlp_fused: bad out_fmt / ldo");
    DM3D_REQUIRE(d->out_fmt == DM3D_FMT_H2 || d->ldo % 4 == 0, "mlp_fused: ldo %% 4 != 0");
    DM3D_REQUIRE(dm3d_aligned16(d->x) && dm3d_aligned16(d->w0) && dm3d_aligned16(d->w1) && dm3d_aligned16(d->out) && dm3d_aligned16(d->b0) && dm3d_aligned16(d->b1),
                 "mlp_fused: x / w0 / w1 / out / b0 / b1 must be 16-byte aligned");
    DM3D_REQUIRE(!d->res2 || d->res, "mlp_fused: res2 needs res");
    DM3D_REQUIRE(!d->res || (d->ldr >= d->units && d->ldr % 4 == 0 && dm3d_aligned16(d->res) && dm3d_aligned16(d->res2)), "mlp_fused: residuals need ldr >= units, ldr %% 4 == 0, 16-byte alignment");
    MlpArgs a{};
    a.x = d->x; a.ldx = d->ldx; a.w0 = d->w0; a.b0 = d->b0; a.w1 = d->w1; a.b1 = d->b1;
    a.res = d->res; a.res2 = d->res2; a.ldr = d->ldr; a.out = d->out; a.ldo = d->ldo; a.out_h2 = d->out_fmt == DM3D_FMT_H2;
    a.m = d->m; a.range_flag = d->range_flag; a.range_limit = d->range_limit > 0.0f ? d->range_limit : 65504.0f;
    if (d->w2) {
        DM3D_REQUIRE(d->b2 && dm3d_aligned16(d->w2) && dm3d_aligned16(d->b2), "mlp_fused: the tail needs w2 and b2, 16-byte aligned");
        DM3D_REQUIRE(d->out_fmt == DM3D_FMT_F32, "mlp_fused: with a tail (w2) the output is float32");
        DM3D_REQUIRE(!d->res3 || (d->ldr3 >= d->units && d->ldr3 % 4 == 0 && dm3d_aligned16(d->res3)), "mlp_fused: res3 needs ldr3 >= units, ldr3 %% 4 == 0, 16-byte alignment");
        a.w2 = d->w2; a.b2 = d->b2; a.res3 = d->res3; a.ldr3 = d->ldr3;
    }
    constexpr size_t lds = 64 * 1040;                       // the x staging area of the prologue; the H slab (32 KB) overlays it afterwards
    static std::atomic<bool> attr_set[64] = {};
    int dev = 0;
    DM3D_HIP(hipGetDevice(&dev));
    DM3D_REQUIRE(dev >= 0 && dev < 64, "mlp_fused: device ordinal %d", dev);
    if (!attr_set[dev]) {
        DM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fused_h3<256>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(mlp_fused_h3<256>, dim3((unsigned)((d->m + 63) / 64)), dim3(256), lds, static_cast<hipStream_t>(stream), a);
    return dm3d_launch_check("mlp_fused_h3");
}
