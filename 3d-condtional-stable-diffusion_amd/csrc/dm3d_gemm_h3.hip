// dm3d_gemm_h3.hip — batched "TN" contraction out[b][m][n] = act(alpha * sum_k A[b][m][k] B[b][n][k] + bias) + res on the
// 16-bit matrix pipe with float32-grade results (split-float16 arithmetic, see dm3d_h3.h).
//
// Replaces the same reference ops as dm3d_gemm.hip (layers.Dense, 1x1 Conv3D, the attention einsums; reference
// networks/conditional_dm3d.py:129-137, 164-180).  A GEMM re-stages its operands for every 64 output columns, so — unlike the
// conv, which amortises one halo over 27 taps — splitting float32 operands while staging would make the kernel VALU-bound
// (5 VALU per MFMA).  Operands therefore normally arrive pre-split in DM3D_FMT_H2 (written by the producing kernel's
// epilogue, by the LayerNorm / softmax kernels, or once at load time for weights) and staging is plain 16-byte copies into
// the XOR-swizzled LDS records; a float32 operand is still accepted (split on the fly) for the tensors that enter a block
// from a convolution.
//
// Workgroup = 256 threads = 2 x 2 waves, K in chunks of 32 (two 16-k records per row), LDS double-buffered, the next chunks prefetched
// into registers during the MFMAs, one barrier per chunk.  Two tile forms (template MR = 32 x 32 MFMA tiles per wave and axis):
//   MR = 2: tile 128 x 128, waves of 64 x 64, two chunks in flight.  The square tile matters: the kernel re-stages both operands every
//           chunk and is bound by LDS write bandwidth (ds_write_b128 ~ 70 B/clk/CU); 128 x 128 stages 341 B per MFMA against 427 B for a
//           256 x 64 tile, and the model's GEMMs at B = 32 (m = B*L = 16384, n = 256...1024) still yield >= 256 workgroups.
//   MR = 1: tile 64 x 64, waves of 32 x 32, EIGHT chunks in flight (16 registers each) — the small-batch form (round 3).  At B = 4 a
//           K = 256 GEMM of the attention block is 32 tiles of 128 x 128 on a 256-CU chip, and each of its 8 chunks (~0.4 us of MFMAs)
//           waited out a memory round trip with two chunks of lookahead: 20 us per launch, 36 launches per step = 22 % of the step
//           (profiles/r03_layers_B4.log).  With every chunk of a K = 256 row requested up front the loop pays one round trip, and four
//           times as many workgroups share the chip.  Chosen when the 128 x 128 form would leave CUs without a workgroup.
#include "dm3d_h3.h"
#include <cstdlib>
#include <type_traits>

// Diagnostic build only (-DDM3D_GEMM_STAMPS, tools/mk_stamp_variants.py -> variants/gst.so; the product library carries none of it): thread 0 of
// the first 2048 workgroups writes s_memtime at phase boundaries (0 entry, 1 first fetch, 2.. chunk steps, 10 epilogue, 11 end) and
// s_memrealtime beside stamps 1 and 10 into a buffer of its own (tools/gemm_stamps.py, tools/kernel_clock.py gemm).
#ifdef DM3D_GEMM_STAMPS
__device__ unsigned long long* g_dbg_stamps = nullptr;
extern "C" int dm3d_debug_set_stamps(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_stamps), &p, sizeof(p)); }
#define GSTAMP(i) do { if (g_dbg_stamps && threadIdx.x == 0 && blockIdx.x < 2048) { g_dbg_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    if ((i) == 1) g_dbg_stamps[blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memrealtime(); \
    if ((i) == 10) g_dbg_stamps[blockIdx.x * 16 + 15] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif

namespace {

struct GemmH3Args {
    const void* a; long lda, sa;        // leading dimensions / batch strides in elements (4 bytes each in both formats)
    const void* b; long ldb, sb;
    void* out; long ldo, so;
    int m, n, k;
    float alpha;
    const float* bias; int bias_m; int act;
    const float* res; long ldr, sr;
    const float* res2;                  // optional second float32 residual with res's layout
    int out_h2;
    int batch;
    int* range_flag; float range_limit;     // H3 range guard (include/dm3d.h)
};

constexpr int MAX_GROUP = 4;
struct GemmGroup {                      // independent problems served by one launch: blockIdx.x is a flat tile index
    GemmH3Args prob[MAX_GROUP];
    int tstart[MAX_GROUP + 1];          // prefix sums of tiles_m * tiles_n * batch
    int tiles_m[MAX_GROUP];
    int count;
};

constexpr int REC = DM3D_REC;
constexpr int KC = 32;

template <bool A_F32, bool B_F32, int MR>
__global__ __launch_bounds__(256, 2) void gemm_tn_h3(const GemmGroup grp) {
    constexpr int TM = 64 * MR, NT = 64 * MR;                     // 2 x 2 waves of (32 MR) x (32 MR): A and B staging balanced
    constexpr int A_BUF = 2 * TM * REC, B_BUF = 2 * NT * REC;     // halfs per buffer (two records per row)
    constexpr int NS = MR == 2 ? 2 : 8;                           // register sets = chunks in flight behind the one being multiplied
    // problem selection (uniform): flat tile index -> (problem, batch, tile_m, tile_n); m tiles vary fastest so that
    // consecutive workgroups share the B tile (the weight operand) through L2
    GSTAMP(0);
    int which = 0;
#pragma unroll
    for (int i = 1; i < MAX_GROUP; ++i) which += (i < grp.count && (int)blockIdx.x >= grp.tstart[i]) ? 1 : 0;
    const GemmH3Args& p = grp.prob[which];
    const int tm = grp.tiles_m[which], tn = (p.n + NT - 1) / NT;
    int local = blockIdx.x - grp.tstart[which];
    const int bz = local / (tm * tn);
    local -= bz * tm * tn;
    const int tile_n = local / tm, tile_m = local - tile_n * tm;
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_g[];
    _Float16* lds_a = smem_g;                   // [2][2][TM][REC]
    _Float16* lds_b = smem_g + 2 * A_BUF;       // [2][2][NT][REC]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int m0 = tile_m * TM, n0 = tile_n * NT;
    const char* A = static_cast<const char*>(p.a) + (size_t)bz * p.sa * 4;
    const char* B = static_cast<const char*>(p.b) + (size_t)bz * p.sb * 4;

    // ---- staging maps (all loads unconditional on clamped rows; rows beyond m / n only feed outputs that are never stored)
    // H2 source: a row's chunk is 128 contiguous bytes = 8 pieces (record kk = w >> 2, slot w & 3)
    // F32 source: item = (row, 8-k group g): two float4 -> hi slot g & 1, lo slot 2 + (g & 1) of record g >> 1
    constexpr int A_ITEMS = (A_F32 ? 1 : 2) * MR;
    constexpr int BI = (B_F32 ? 1 : 2) * MR;
    constexpr int RA = A_F32 ? 2 * A_ITEMS : A_ITEMS, RB = B_F32 ? 2 * BI : BI;
    f32x4 ra[NS][RA], rb[NS][RB];               // NS register sets: NS chunks in flight behind the one being multiplied
    size_t a_src[A_ITEMS];
    int a_dst[A_ITEMS];
#pragma unroll
    for (int i = 0; i < A_ITEMS; ++i) {
        const int q = tid + i * 256;
        if (A_F32) {
            const int row = q >> 2, g = q & 3;
            const int mrow = m0 + row < p.m ? m0 + row : p.m - 1;
            a_src[i] = ((size_t)mrow * p.lda + g * 8) * 4;
            a_dst[i] = ((g >> 1) * TM + row) * REC + (((g & 1) ^ swz(row)) << 3);
        } else {
            const int row = q >> 3, w = q & 7;
            const int mrow = m0 + row < p.m ? m0 + row : p.m - 1;
            a_src[i] = (size_t)mrow * p.lda * 4 + w * 16;
            a_dst[i] = ((w >> 2) * TM + row) * REC + (((w & 3) ^ swz(row)) << 3);
        }
    }
    size_t b_src[BI];
    int b_dst[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int q = tid + i * 256;
        if (B_F32) {
            const int row = q >> 2, g = q & 3;
            const int nrow = n0 + row < p.n ? n0 + row : p.n - 1;
            b_src[i] = ((size_t)nrow * p.ldb + g * 8) * 4;
            b_dst[i] = ((g >> 1) * NT + row) * REC + (((g & 1) ^ swz(row)) << 3);
        } else {
            const int row = q >> 3, w = q & 7;
            const int nrow = n0 + row < p.n ? n0 + row : p.n - 1;
            b_src[i] = (size_t)nrow * p.ldb * 4 + w * 16;
            b_dst[i] = ((w >> 2) * NT + row) * REC + (((w & 3) ^ swz(row)) << 3);
        }
    }

    auto fetch = [&](auto SET, int k0) {        // k0 is clamped by the caller to the last chunk
        constexpr int S = decltype(SET)::value;
        // when k % 32 == 16 the last chunk has no second record: its loads re-read the first one (64 bytes earlier)
        // instead of running past the end of the row; publish() zero-fills the LDS image
        const bool sec = k0 + 16 < p.k;
#pragma unroll
        for (int i = 0; i < A_ITEMS; ++i) {
            const bool second = A_F32 ? (((tid + i * 256) & 3) >= 2) : (((tid + i * 256) & 7) >= 4);
            const size_t koff = (size_t)k0 * 4 - ((!sec && second) ? 64 : 0);
            if (A_F32) {
                ra[S][2 * i] = *reinterpret_cast<const f32x4*>(A + a_src[i] + koff);
                ra[S][2 * i + 1] = *reinterpret_cast<const f32x4*>(A + a_src[i] + koff + 16);
            } else {
                ra[S][i] = *reinterpret_cast<const f32x4*>(A + a_src[i] + koff);
            }
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const bool second = B_F32 ? (((tid + i * 256) & 3) >= 2) : (((tid + i * 256) & 7) >= 4);
            const size_t koff = (size_t)k0 * 4 - ((!sec && second) ? 64 : 0);
            if (B_F32) {
                rb[S][2 * i] = *reinterpret_cast<const f32x4*>(B + b_src[i] + koff);
                rb[S][2 * i + 1] = *reinterpret_cast<const f32x4*>(B + b_src[i] + koff + 16);
            } else {
                rb[S][i] = *reinterpret_cast<const f32x4*>(B + b_src[i] + koff);
            }
        }
    };
    // k tail: the chunk's second record is absent when k % 32 == 16; its LDS image is zero filled.  `live` false (a step past the last
    // chunk: the loop runs whole rounds of NS steps without a branch around its loads): the whole image is zero, the step adds nothing.
    auto publish = [&](auto SET, int buf, bool second_rec, bool live) {
        constexpr int S = decltype(SET)::value;
        _Float16* da = lds_a + buf * A_BUF;
        _Float16* db = lds_b + buf * B_BUF;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < A_ITEMS; ++i) {
            if (A_F32) {
                const bool ok = live && (second_rec || ((tid + i * 256) & 3) < 2);
                h8 hi, lo;
                split8(ra[S][2 * i], ra[S][2 * i + 1], ok ? 65504.0f : 0.0f, ok ? 65504.0f : 0.0f, hi, lo);
                *reinterpret_cast<h8*>(da + a_dst[i]) = hi;
                *reinterpret_cast<h8*>(da + (a_dst[i] ^ 16)) = lo;
            } else {
                const bool ok = live && (second_rec || ((tid + i * 256) & 7) < 4);
                *reinterpret_cast<f32x4*>(da + a_dst[i]) = ok ? ra[S][i] : z;
            }
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            if (B_F32) {
                const bool ok = live && (second_rec || ((tid + i * 256) & 3) < 2);
                h8 hi, lo;
                split8(rb[S][2 * i], rb[S][2 * i + 1], ok ? 65504.0f : 0.0f, ok ? 65504.0f : 0.0f, hi, lo);
                *reinterpret_cast<h8*>(db + b_dst[i]) = hi;
                *reinterpret_cast<h8*>(db + (b_dst[i] ^ 16)) = lo;
            } else {
                const bool ok = live && (second_rec || ((tid + i * 256) & 7) < 4);
                *reinterpret_cast<f32x4*>(db + b_dst[i]) = ok ? rb[S][i] : z;
            }
        }
    };

    // fragment addresses: record = row, hi slot = half ^ swz(row), lo slot = that ^ 2; 32-row tiles are 32 records apart
    const int wm = wave >> 1, wn = wave & 1;
    int a_hi, b_hi;
    {
        const int row = wm * 32 * MR + l32, col = wn * 32 * MR + l32;
        a_hi = row * REC + ((half ^ swz(row)) << 3);
        b_hi = col * REC + ((half ^ swz(col)) << 3);
    }

    f32x16 acc[MR][MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < MR; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.0f;

    auto compute = [&](int buf) {
        const _Float16* la = lds_a + buf * A_BUF;
        const _Float16* lb = lds_b + buf * B_BUF;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h8 ah[MR], al[MR], bh[MR], bl[MR];
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                ah[mr] = *reinterpret_cast<const h8*>(la + a_hi + (kk * TM + mr * 32) * REC);
                al[mr] = *reinterpret_cast<const h8*>(la + (a_hi ^ 16) + (kk * TM + mr * 32) * REC);
            }
#pragma unroll
            for (int nr = 0; nr < MR; ++nr) {
                bh[nr] = *reinterpret_cast<const h8*>(lb + b_hi + (kk * NT + nr * 32) * REC);
                bl[nr] = *reinterpret_cast<const h8*>(lb + (b_hi ^ 16) + (kk * NT + nr * 32) * REC);
            }
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < MR; ++nr) {
                    acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[mr], bh[nr], acc[mr][nr], 0, 0, 0);
                    acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mr], bl[nr], acc[mr][nr], 0, 0, 0);
                    acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[mr], bh[nr], acc[mr][nr], 0, 0, 0);
                }
        }
    };

    // Chunk c travels in register set c % NS and LDS buffer c & 1; while chunk c is multiplied, c+1 .. c+NS are in flight (a K = 256
    // GEMM has only 8 chunks of ~0.4 us each — one chunk of lookahead does not cover a memory round trip).  The loop runs whole rounds of
    // NS steps; loads past the last chunk re-read it (clamped, never behind a branch) and their steps publish zeros.
    const int nchunks = (p.k + KC - 1) / KC, last = nchunks - 1;
    auto clampk = [&](int c) { return (c < last ? c : last) * KC; };
    auto for_sets = [&](auto&& f) {                       // f(integral_constant<j>) for j = 0 .. NS-1: static register-set indices
        f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{});
        if constexpr (NS == 8) {
            f(std::integral_constant<int, 2>{}); f(std::integral_constant<int, 3>{}); f(std::integral_constant<int, 4>{});
            f(std::integral_constant<int, 5>{}); f(std::integral_constant<int, 6>{}); f(std::integral_constant<int, 7>{});
        }
    };
    GSTAMP(1);
    for_sets([&](auto J) { fetch(J, clampk(decltype(J)::value)); });
    __builtin_amdgcn_sched_barrier(0);
    for (int it = 0; it < nchunks; it += NS) {
        for_sets([&](auto J) {
            constexpr int j = decltype(J)::value;
            const int c = it + j;
            publish(J, j & 1, c * KC + 16 < p.k, c < nchunks);
            // chunk c visible; everyone has left chunk c-1 (other buffer).  A raw barrier behind this wave's LDS traffic only: __syncthreads()
            // also drains the vector-memory counter, i.e. it waited out the prefetches that are meant to stay in flight across it — rounds 1-2
            // ran with ONE chunk of lookahead in effect (in-kernel stamps, round 3: 2 250 cycles per chunk for 768 of MFMA)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (c < 8) GSTAMP(2 + c);
            fetch(J, clampk(c + NS));
            __builtin_amdgcn_sched_barrier(0);
            compute(j & 1);
            __builtin_amdgcn_sched_barrier(0);
        });
    }

    // ---- epilogue: lane (l32, half) holds column n and rows acc_row(r, half) of each 32 x 32 tile.  Addresses are a uniform
    // 64-bit tile base plus 32-bit lane offsets.  A small-K GEMM has only ~200 MFMAs per wave, so the epilogue is written for
    // latency: every option is a uniform select (no per-element branch — hipcc waits vmcnt(0) behind each load that sits under
    // one), the residual / row-bias loads of a 64-column half are all issued before the first is used (absent operands read
    // element 0 of A instead and are discarded), and the H2 pair exchange is one DPP move.
    GSTAMP(10);
    const bool full = m0 + TM <= p.m && n0 + NT <= p.n;
    char* O = static_cast<char*>(p.out) + ((size_t)bz * p.so + (size_t)m0 * p.ldo) * 4;
    const bool has_r = p.res != nullptr, has_r2 = p.res2 != nullptr, has_bm = p.bias && p.bias_m, has_bn = p.bias && !p.bias_m;
    const float* dummy = static_cast<const float*>(p.a);
    const float* R = has_r ? p.res + (size_t)bz * p.sr + (size_t)m0 * p.ldr : dummy;
    const float* R2 = p.res2 + (size_t)bz * p.sr + (size_t)m0 * p.ldr;
    const float* Bm = p.bias + m0;
    const float* Bn = has_bn ? p.bias : dummy;
    const int r_mul = has_r ? 1 : 0, bn_mul = has_bn ? 1 : 0;
    const int lrow = wm * 32 * MR + 4 * half;                  // + mr*32 + (r&3) + 8*(r>>2)
    const int ldo = (int)p.ldo, ldr = (int)p.ldr;
    const float lo_bound = p.act == DM3D_ACT_RELU ? 0.0f : -3.4e38f;
    const int row_max = p.m - 1 - m0;                          // last valid row of this tile (partial tiles clamp their loads)
    float amax = 0.0f;
    auto epilogue = [&](auto FULL_T, auto H2_T) {
        constexpr bool FULL = decltype(FULL_T)::value, H2 = decltype(H2_T)::value;
#pragma unroll
        for (int nr = 0; nr < MR; ++nr) {
            const int n = n0 + wn * 32 * MR + nr * 32 + l32;
            const bool n_ok = FULL || n < p.n;
            const int nc = n_ok ? n : p.n - 1;
            const float bnv = Bn[nc * bn_mul];
            const float bn = has_bn ? bnv : 0.0f;
            // H2 column position inside its row: record n >> 4, slot (n >> 3) & 1 (+2 for lo), element n & 7; lanes n and n^1
            // exchange halves so that every lane stores one dword: even lanes the hi pair, odd lanes the lo pair
            const int ocol = H2 ? (n >> 4) * 64 + ((n >> 3) & 1) * 16 + ((n & 7) >> 1) * 4 + (n & 1) * 32 : n * 4;
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) {
                __builtin_amdgcn_sched_barrier(0);                   // one 32 x 32 block at a time: 16 loads in flight, not 64
                float rv[16], v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {                       // the common residual: requested first, unconditionally
                    const int row = lrow + mr * 32 + (r & 3) + 8 * (r >> 2);
                    const int rc = FULL ? row : (row < row_max ? row : row_max);
                    rv[r] = R[(unsigned)((rc * ldr + nc) * r_mul)];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = acc[mr][nr][r] * p.alpha + bn;
                if (has_bm) {                                        // rare options: one uniform branch around a batch of loads
                    float bm[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = lrow + mr * 32 + (r & 3) + 8 * (r >> 2);
                        bm[r] = Bm[(unsigned)(FULL ? row : (row < row_max ? row : row_max))];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += bm[r];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], lo_bound);
                if (p.act == DM3D_ACT_SILU) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = dm3d_silu(v[r]);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += has_r ? rv[r] : 0.0f;
                if (has_r2) {
                    float rv2[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = lrow + mr * 32 + (r & 3) + 8 * (r >> 2);
                        rv2[r] = R2[(unsigned)((FULL ? row : (row < row_max ? row : row_max)) * ldr + nc)];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] += rv2[r];
                }
                {   // range guard: a tree (depth 4) instead of a 16-long dependent chain; rows / columns past the edge of a partial
                    // tile hold bias + residual of clamped addresses — ordinary magnitudes, so they are not masked
                    float t8[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) { t8[r] = 0.0f; DM3D_AMAX(t8[r], v[r]); DM3D_AMAX(t8[r], v[r + 8]); }
                    const float t = fmaxf(fmaxf(fmaxf(t8[0], t8[1]), fmaxf(t8[2], t8[3])), fmaxf(fmaxf(t8[4], t8[5]), fmaxf(t8[6], t8[7])));
                    amax = fmaxf(amax, t);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = lrow + mr * 32 + (r & 3) + 8 * (r >> 2);
                    const float o = v[r];
                    const bool ok = FULL || (n_ok && m0 + row < p.m);
                    if (H2) {
                        const unsigned int mine = split1_bits(o);
                        const unsigned int oth = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xf, 0xf, false);   // lane ^ 1
                        const unsigned int word = (n & 1) ? ((oth >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (oth << 16));
                        if (ok) *reinterpret_cast<unsigned int*>(O + (size_t)(unsigned)(row * ldo) * 4 + ocol) = word;
                    } else {
                        if (ok) *reinterpret_cast<float*>(O + (size_t)(unsigned)(row * ldo) * 4 + ocol) = o;
                    }
                }
            }
        }
    };
    const std::true_type yes;
    const std::false_type no;
    if (full) {
        if (p.out_h2) epilogue(yes, yes); else epilogue(yes, no);
    } else {
        if (p.out_h2) epilogue(no, yes); else epilogue(no, no);
    }
    if (p.range_flag && amax > p.range_limit) *p.range_flag = 1;
    GSTAMP(11);
}

// float32 [rows][k] (ld_src) -> DM3D_FMT_H2 [rows][ld_dst], scaled by 2^exp2, zero filled up to round_up(k, 16)
__global__ __launch_bounds__(256) void split_h2_kernel(const float* __restrict__ src, long rows, int k, long ld_src, float scale,
                                                       _Float16* __restrict__ dst, long ld_dst) {
    const int groups = (int)((k + 15) / 16) * 2;                    // 8-k groups per row incl. padding
    const long total = rows * groups;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / groups;
        const int g = (int)(i % groups);
        f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        const float* s = src + row * ld_src + g * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (g * 8 + e < k) v0[e] = s[e] * scale;
            if (g * 8 + 4 + e < k) v1[e] = s[4 + e] * scale;
        }
        h8 hi, lo;
        split8(v0, v1, 65504.0f, 65504.0f, hi, lo);
        _Float16* rec = dst + row * ld_dst * 2 + (g >> 1) * REC;     // ld_dst elements = 2 halfs each
        *reinterpret_cast<h8*>(rec + (g & 1) * 8) = hi;
        *reinterpret_cast<h8*>(rec + 16 + (g & 1) * 8) = lo;
    }
}

}  // namespace

static int fill_args(const dm3d_gemm_desc* d, GemmH3Args& a) {
    DM3D_REQUIRE(d->a && d->b && d->out, "gemm(h3): a/b/out must be non-null");
    DM3D_REQUIRE(d->m > 0 && d->n > 0 && d->k > 0 && d->batch > 0, "gemm(h3): non-positive extent");
    DM3D_REQUIRE(d->k % 16 == 0 && d->lda % 4 == 0 && d->ldb % 4 == 0, "gemm(h3): k=%d must be a multiple of 16", d->k);
    DM3D_REQUIRE(d->lda >= d->k && d->ldb >= d->k && d->ldo >= d->n, "gemm(h3): leading dimension smaller than the row");
    DM3D_REQUIRE(d->a_fmt == DM3D_FMT_F32 || d->a_fmt == DM3D_FMT_H2, "gemm(h3): bad a_fmt");
    DM3D_REQUIRE(d->b_fmt == DM3D_FMT_F32 || d->b_fmt == DM3D_FMT_H2, "gemm(h3): bad b_fmt");
    DM3D_REQUIRE(d->a_fmt == DM3D_FMT_F32 || (d->lda % 16 == 0 && d->stride_a % 16 == 0), "gemm(h3): H2 operand a needs lda, stride %% 16 == 0");
    DM3D_REQUIRE(d->b_fmt == DM3D_FMT_F32 || (d->ldb % 16 == 0 && d->stride_b % 16 == 0), "gemm(h3): H2 operand b needs ldb, stride %% 16 == 0");
    DM3D_REQUIRE(d->out_fmt == DM3D_FMT_F32 || (d->out_fmt == DM3D_FMT_H2 && d->n % 16 == 0 && d->ldo % 16 == 0 && d->stride_o % 16 == 0),
                 "gemm(h3): H2 output needs n, ldo, stride_o %% 16 == 0");
    DM3D_REQUIRE(dm3d_aligned16(d->a) && dm3d_aligned16(d->b), "gemm(h3): a/b must be 16-byte aligned");
    DM3D_REQUIRE(dm3d_aligned16(d->out) || d->out_fmt == DM3D_FMT_F32, "gemm(h3): H2 output must be 16-byte aligned");
    DM3D_REQUIRE(!d->res || d->ldr >= d->n, "gemm(h3): ldr smaller than n");
    DM3D_REQUIRE(!d->res2 || d->res, "gemm(h3): res2 needs res");
    DM3D_REQUIRE(d->act >= DM3D_ACT_NONE && d->act <= DM3D_ACT_SILU, "gemm(h3): unknown act %d", d->act);
    a.a = d->a; a.lda = d->lda; a.sa = d->stride_a;
    a.b = d->b; a.ldb = d->ldb; a.sb = d->stride_b;
    a.out = d->out; a.ldo = d->ldo; a.so = d->stride_o;
    a.m = d->m; a.n = d->n; a.k = d->k; a.alpha = d->alpha;
    a.bias = d->bias; a.bias_m = d->bias_along_m; a.act = d->act;
    a.res = d->res; a.ldr = d->ldr; a.sr = d->stride_r; a.res2 = d->res2;
    a.out_h2 = d->out_fmt == DM3D_FMT_H2;
    a.batch = d->batch;
    a.range_flag = d->range_flag; a.range_limit = d->range_limit > 0.0f ? d->range_limit : 65504.0f;
    return DM3D_OK;
}

template <int MR>
static int launch_group_mr(GemmGroup& g, bool af, bool bf, hipStream_t st) {
    constexpr int TM = 64 * MR, NT = 64 * MR;
    constexpr size_t lds = (size_t)(2 * 2 * TM * REC + 2 * 2 * NT * REC) * sizeof(_Float16);     // 65536 (MR 2) / 32768 (MR 1)
    static std::atomic<bool> attr_set[64] = {};            // per device: the attribute belongs to the device the launch goes to
    int dev = 0;
    DM3D_HIP(hipGetDevice(&dev));
    DM3D_REQUIRE(dev >= 0 && dev < 64, "gemm(h3): device ordinal %d", dev);
    if (!attr_set[dev]) {
        const void* fns[] = {reinterpret_cast<const void*>(&gemm_tn_h3<false, false, MR>), reinterpret_cast<const void*>(&gemm_tn_h3<true, false, MR>),
                             reinterpret_cast<const void*>(&gemm_tn_h3<false, true, MR>), reinterpret_cast<const void*>(&gemm_tn_h3<true, true, MR>)};
        for (const void* f : fns) DM3D_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    long t = 0;
    for (int i = 0; i < g.count; ++i) {
        g.tstart[i] = (int)t;
        g.tiles_m[i] = (g.prob[i].m + TM - 1) / TM;
        t += (long)g.tiles_m[i] * ((g.prob[i].n + NT - 1) / NT) * g.prob[i].batch;
    }
    for (int i = g.count; i <= MAX_GROUP; ++i) g.tstart[i] = (int)t;
    DM3D_REQUIRE(t > 0 && t < (1l << 31), "gemm(h3): %ld tiles do not fit the grid", t);
    dim3 grid((unsigned)t);
    if (af && bf) hipLaunchKernelGGL((gemm_tn_h3<true, true, MR>), grid, dim3(256), lds, st, g);
    else if (af) hipLaunchKernelGGL((gemm_tn_h3<true, false, MR>), grid, dim3(256), lds, st, g);
    else if (bf) hipLaunchKernelGGL((gemm_tn_h3<false, true, MR>), grid, dim3(256), lds, st, g);
    else hipLaunchKernelGGL((gemm_tn_h3<false, false, MR>), grid, dim3(256), lds, st, g);
    return dm3d_launch_check("gemm_tn_h3");
}

// Tile form of a launch: 64 x 64 tiles with eight chunks in flight when the 128 x 128 form would give fewer workgroups than
// DM3D_GEMM_SMALL_TILES (default 256 = one per CU: at 256 tiles — the proj GEMMs at B = 32 — the 128 x 128 form measured faster, 1.25 vs 1.35 ms per step), else 128 x 128.  DM3D_GEMM_MR = 1 | 2 forces one form (A/B knob, read per call).
static int launch_group(GemmGroup& g, bool af, bool bf, hipStream_t st) {
    long t128 = 0;
    for (int i = 0; i < g.count; ++i) t128 += (long)((g.prob[i].m + 127) / 128) * ((g.prob[i].n + 127) / 128) * g.prob[i].batch;
    const char* e = getenv("DM3D_GEMM_MR");
    const char* th = getenv("DM3D_GEMM_SMALL_TILES");
    const int force = e ? atoi(e) : 0;
    const bool small = force == 1 || (force != 2 && t128 < (th ? atol(th) : 256L));
    return small ? launch_group_mr<1>(g, af, bf, st) : launch_group_mr<2>(g, af, bf, st);
}

int dm3d_gemm_h3_launch(const dm3d_gemm_desc* d, hipStream_t st) {
    GemmGroup g{};
    g.count = 1;
    const int rc = fill_args(d, g.prob[0]);
    if (rc != DM3D_OK) return rc;
    return launch_group(g, d->a_fmt == DM3D_FMT_F32, d->b_fmt == DM3D_FMT_F32, st);
}

extern "C" int dm3d_gemm_tn_group(const dm3d_gemm_desc* descs, int32_t count, void* stream) {
    DM3D_REQUIRE(descs != nullptr && count >= 1 && count <= MAX_GROUP, "gemm_group: count %d not in [1,%d]", count, MAX_GROUP);
    GemmGroup g{};
    g.count = count;
    for (int i = 0; i < count; ++i) {
        DM3D_REQUIRE(descs[i].precision == DM3D_PREC_H3, "gemm_group: only DM3D_PREC_H3 problems can be grouped");
        DM3D_REQUIRE(descs[i].a_fmt == descs[0].a_fmt && descs[i].b_fmt == descs[0].b_fmt,
                     "gemm_group: all problems must share the operand formats");
        const int rc = fill_args(&descs[i], g.prob[i]);
        if (rc != DM3D_OK) return rc;
    }
    return launch_group(g, descs[0].a_fmt == DM3D_FMT_F32, descs[0].b_fmt == DM3D_FMT_F32, static_cast<hipStream_t>(stream));
}

extern "C" int dm3d_split_h2(const float* src, int64_t rows, int32_t k, int64_t ld_src, int32_t exp2, void* dst, int64_t ld_dst,
                             void* stream) {
    DM3D_REQUIRE(src && dst && rows > 0 && k > 0 && ld_src >= k, "split_h2: bad arguments");
    DM3D_REQUIRE(ld_dst % 16 == 0 && ld_dst >= dm3d_round_up(k, 16), "split_h2: ld_dst=%lld must be a multiple of 16 covering k", (long long)ld_dst);
    DM3D_REQUIRE(exp2 >= -100 && exp2 <= 100 && dm3d_aligned16(dst), "split_h2: exp2 out of range or dst unaligned");
    const long total = rows * (long)(dm3d_round_up(k, 16) / 8);
    long g = (total + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(split_h2_kernel, dim3((unsigned)g), dim3(256), 0, static_cast<hipStream_t>(stream), src, (long)rows, k,
                       (long)ld_src, ldexpf(1.0f, exp2), static_cast<_Float16*>(dst), (long)ld_dst);
    return dm3d_launch_check("split_h2_kernel");
}
