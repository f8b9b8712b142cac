// dm3d_attn_h3.hip — fused single-head attention  out = softmax(q k^T * scale) v (+ res)  for the attention levels of the U-Net
// (reference networks/conditional_dm3d.py:163-184: einsum "blc,bLc->blL", softmax over the flattened D*H*W tokens, einsum "blL,bLc->blc";
// networks/dm3d.py:47-61).  One launch per attention pass (several passes per launch: grid.z) instead of score GEMM + row softmax + P.V
// GEMM: the [batch, L, L] probabilities never exist in HBM (134 MB per launch written and re-read at B = 32, L = 512 before).
//
// Flash-style, in the split-float16 arithmetic of dm3d_h3.h (three v_mfma_f32_32x32x16_f16 passes per product, float32 accumulate):
//   * a workgroup = 4 waves owns 128 queries of one sample; a wave owns 32 of them and keeps, in registers, its query rows (the B operand
//     of the score product, 128 VGPRs), the running row maximum / row sum, and O^T [256 channels x 32 queries] (128 accumulator VGPRs);
//   * K and V^T stream through LDS in 32-key tiles (DM3D_FMT_H2 records are the LDS image: staging is 16-byte copies);
//   * per tile the wave computes S^T [32 keys x 32 queries] — TRANSPOSED, so that the accumulator layout (lane = query column) is already
//     the B-operand layout of the second product O^T += V^T P^T: no cross-lane movement of the probabilities at all.  The rows of the
//     S^T tile are assigned to keys through the permutation KEY_OF_ROW below, chosen so that the 8 accumulator registers a lane half
//     feeds to one MFMA k-step are 8 CONSECUTIVE keys — what the V^T fragment (a plain H2 record) contracts with;
//   * online softmax: the row maximum of a query lives in two lanes (l, l^32), exchanged once per tile; O^T is rescaled only when some
//     lane's maximum moved (wave-uniform test); probabilities are split hi/lo in registers;
//   * epilogue: O^T / row sum goes through LDS (transposed back) and leaves as coalesced float32 rows, residual added.
// Needs c == 256, lq % 128 == 0, lk % 32 == 0 and H2 operands; dm3d_attention[_group] falls back to the three-launch form otherwise.
#include <cstdlib>
#include "dm3d_common.h"
#include "dm3d_h3.h"

#ifdef DM3D_ATTN_STAMPS        // tools/attn_stamps.py: cycle stamps of wave 0 of the first workgroups (variant build only)
__device__ unsigned long long* g_attn_stamps = nullptr;
extern "C" int dm3d_debug_set_stamps_attn(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &p, sizeof(p)); }
#define ASTAMP(i) do { if (g_attn_stamps && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) g_attn_stamps[blockIdx.x * 128 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define ASTAMP(i) do { } while (0)
#endif

namespace {

constexpr int REC = DM3D_REC;
constexpr int AC = 256;                     // channels (units of the attention block)
constexpr int QT = 128, KT = 32;            // queries per workgroup, keys per tile
constexpr int K_TILE = (AC / 16) * KT * REC;          // halfs: [16 channel records][32 keys][REC]      = 32 KB
constexpr int V_TILE = (KT / 16) * AC * REC;          // halfs: [2 key records][256 channels][REC]      = 32 KB
constexpr int O_LD = AC + 4;                           // floats per query row of the epilogue image

struct AttnPass {
    const char* q; long ldq;                // bytes are computed from element counts (4 bytes per element in H2)
    const char* k; long ldk, sk;
    const char* vt; long ldv, sv;
    float* out; long ldo;
    const float* res;
};
struct AttnArgs {
    AttnPass pass[4];
    int lq, lk;
    float scale;
};

// S^T row m (MFMA row) holds key KEY_OF_ROW(m) of the tile: with m = 8a + 4h + i (accumulator register 4a + i of lane half h) the key is
// 16 (a >> 1) + 8 h + 4 (a & 1) + i, so registers 8j .. 8j+7 of half h are keys 16 j + 8 h + 0..7
__device__ __forceinline__ int key_of_row(int m) {
    const int a = m >> 3, h = (m >> 2) & 1, i = m & 3;
    return ((a >> 1) << 4) + (h << 3) + ((a & 1) << 2) + i;
}

__global__ __launch_bounds__(256, 1) void attn_fused_h3(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) _Float16 smem_at[];
    // two buffers of { K tile [16][KT][REC], V^T tile [2][AC][REC] }
    const AttnPass& ps = p.pass[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int b = blockIdx.y, q0 = blockIdx.x * QT;
    const int lq = p.lq, lk = p.lk;

    // ---- this lane's query row (B operand of the score product): 16 channel records x (hi, lo)
    h8 qh[16], ql[16];
    {
        const char* qrow = ps.q + ((size_t)b * lq + q0 + wave * 32 + l32) * ps.ldq * 4 + half * 16;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            qh[s] = *reinterpret_cast<const h8*>(qrow + s * 64);
            ql[s] = *reinterpret_cast<const h8*>(qrow + s * 64 + 32);
        }
    }

    // ---- K / V^T tiles go global -> LDS by LDS-DMA (no staging registers: the wave already holds 128 + 128 + 16 of them).  One
    // instruction fills 1 KB = 16 records of the image [record plane][row][REC]; lane l lands in record l >> 2, physical slot l & 3, so it
    // fetches the LOGICAL slot (l & 3) ^ swz(row) of that row — the XOR swizzle is applied on the source address.
    //   K tile  [16 channel records s][32 keys][REC]: instruction (s, key half) — wave w moves s = 4w .. 4w+3 for both halves;
    //   V^T tile [2 key records j][256 channels][REC]: instruction (j, 16-channel block) — wave w moves blocks 4w .. 4w+3 for j = 0, 1.
    // Two LDS buffers: tile t+1 streams in while tile t is multiplied (a tile is ~1.7 us of MFMAs; a 16 KB fill lands in ~1.1 us).
    // Buffer form of the LDS-DMA (round 4, as in dm3d_conv_h3w.hip): one resource descriptor per operand in scalar registers, a per-lane
    // 32-bit offset that never changes, the tile as a scalar offset — the first form added a 64-bit per-lane address for every instruction.
    const char* kbase = ps.k + (size_t)b * ps.sk * 4;
    const char* vbase = ps.vt + (size_t)b * ps.sv * 4;
    const auto k_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(kbase), (short)0, (int)((long)lk * ps.ldk * 4), 0x00020000);
    const auto v_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(vbase), (short)0, (int)((long)AC * ps.ldv * 4), 0x00020000);
    const int r16 = lane >> 2, pslot = lane & 3;
    int k_lane[2];                               // + key0 * ldk * 4 + s * 64
    int v_lane[4];                               // + key0 * 4 + j * 64
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        const int row = kh * 16 + r16;
        k_lane[kh] = (int)(row * ps.ldk * 4) + ((pslot ^ swz(row)) << 4);
    }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int row = (wave * 4 + cb) * 16 + r16;
        v_lane[cb] = (int)(row * ps.ldv * 4) + ((pslot ^ swz(row)) << 4);
    }
    // part p of a tile: this wave issues K record plane (4w + p) for both key halves and V^T
    // block (4w + p) for both key records: 4 instructions.  A tile is issued in four parts between the score MFMAs of the previous tile.
    // (In-kernel stamps, tools/attn_stamps.py: an LDS-DMA instruction takes ~170 cycles to ISSUE — 16 per wave and tile = ~2700 of a
    // tile's ~10 000 cycles, against 3072 cycles of MFMA; spreading them moves that cost, it does not remove it: 83.6 -> 78.9 us per
    // launch.  Register staging would not block the issue port but needs 64 more VGPRs than the 496 this kernel holds.)
    auto fetch_part = [&](int key0, int buf, int part) {
        char* dk = reinterpret_cast<char*>(smem_at) + buf * ((K_TILE + V_TILE) * 2);
        char* dv = dk + K_TILE * 2;
        const int sidx = __builtin_amdgcn_readfirstlane(wave * 4 + part);
        const int koff = __builtin_amdgcn_readfirstlane(key0 * (int)ps.ldk * 4 + sidx * 64), voff = __builtin_amdgcn_readfirstlane(key0 * 4);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(k_rsrc, (__attribute__((address_space(3))) void*)(dk + (sidx * KT + kh * 16) * (REC * 2)), 16,
                                                     k_lane[kh], koff, 0, 0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(v_rsrc, (__attribute__((address_space(3))) void*)(dv + (j * AC + sidx * 16) * (REC * 2)), 16,
                                                     v_lane[part], voff + j * 64, 0, 0);
    };
    auto fetch = [&](int key0, int buf) {
#pragma unroll
        for (int part = 0; part < 4; ++part) fetch_part(key0, buf, part);
    };

    // fragment addresses
    const int krow = key_of_row(l32);
    const int k_hi = krow * REC + ((half ^ swz(krow)) << 3);           // + s * KT * REC
    const int v_hi = l32 * REC + ((half ^ swz(l32)) << 3);             // + (j * AC + mt * 32) * REC   (swz(mt*32 + l32) == swz(l32))

    f32x16 oacc[8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[mt][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    fetch(0, 0);
    const int ntiles = lk / KT;
    for (int t = 0; t < ntiles; ++t) {
        // tile t has landed (every wave waits for its own DMAs, then all meet); everyone is past tile t-1, whose buffer tile t+1 now takes
        if (t < 8) ASTAMP(t * 8 + 0);
        // (the builtin, not inline asm: hipcc's wait-count pass must SEE that nothing is in flight here.  With an asm wait it still counted
        // the query-row loads of the kernel's head as possibly pending at the loop head and guarded every qh[s] / ql[s] with vmcnt(32 - 2s) ..
        // vmcnt(0) — waits that in every later tile drain the next tile's DMA pieces just issued: the score phase took 3 700 cycles for
        // 1 536 of MFMA, tools/attn_stamps.py.)
        __builtin_amdgcn_s_waitcnt(0x0F70);                     // vmcnt(0), expcnt 7, lgkmcnt 15
        __syncthreads();
        if (t < 8) ASTAMP(t * 8 + 1);
        const bool more = t + 1 < ntiles;
        const int nkey0 = (more ? t + 1 : t) * KT, nbuf = (t + 1) & 1;
        __builtin_amdgcn_sched_barrier(0);
        const _Float16* lds_k = smem_at + (t & 1) * (K_TILE + V_TILE);
        const _Float16* lds_v = lds_k + K_TILE;
        if (t < 8) ASTAMP(t * 8 + 2);

        // ---- S^T = K Q^T for this tile: rows = keys (permuted), columns = this wave's queries
        // Three accumulators, one per pass: back-to-back MFMAs into ONE accumulator wait out the full result latency (~64 cycles for a
        // 32x32x16) and this kernel runs one wave per SIMD — nothing else would fill those slots.  Summed once per tile.
        f32x16 sacc, sacc1, sacc2;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = 0.0f; sacc1[r] = 0.0f; sacc2[r] = 0.0f; }
        // (the K fragments of step s + 1 are requested in front of step s's MFMAs: left to hipcc, each pair was read right in front of its
        // first use and the LDS round trip — nothing else to run on this SIMD — stood between every two steps)
        h8 kh = *reinterpret_cast<const h8*>(lds_k + k_hi), kl = *reinterpret_cast<const h8*>(lds_k + (k_hi ^ 16));
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int sn = s < 15 ? s + 1 : 15;
            const h8 kh_n = *reinterpret_cast<const h8*>(lds_k + k_hi + sn * (KT * REC));
            const h8 kl_n = *reinterpret_cast<const h8*>(lds_k + (k_hi ^ 16) + sn * (KT * REC));
            __builtin_amdgcn_sched_barrier(0);
            sacc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[s], sacc1, 0, 0, 0);
            sacc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[s], sacc2, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[s], sacc, 0, 0, 0);
            if ((s & 3) == 1 && more) {                        // the next tile's DMAs, a quarter at a time behind steps 1, 5, 9, 13
                __builtin_amdgcn_sched_barrier(0);
                fetch_part(nkey0, nbuf, s >> 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            kh = kh_n; kl = kl_n;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] += sacc1[r] + sacc2[r];        // small terms first
        if (t < 8) ASTAMP(t * 8 + 3);

        // The V^T fragments run one MFMA pass ahead of their use: the lo set of the tile's first k-step (keys 0-15) is requested HERE, in
        // front of the softmax arithmetic, every later set while the pass before its first use runs (each register right behind the last
        // MFMA that reads it).  Requested where they are used, 16 reads and their round trip stood in front of each k-step's 24 MFMAs with
        // nothing else to issue on this SIMD (P.V phase 2 600 cycles for 1 536 of MFMA, tools/attn_stamps.py).  (Both sets of the first
        // k-step in front of the softmax: 35 spilled registers.)
        h8 vh[8], vl[8];
        auto v_read_h = [&](const int j, const int mt) { vh[mt] = *reinterpret_cast<const h8*>(lds_v + v_hi + (j * AC + mt * 32) * REC); };
        auto v_read_l = [&](const int j, const int mt) { vl[mt] = *reinterpret_cast<const h8*>(lds_v + (v_hi ^ 16) + (j * AC + mt * 32) * REC); };
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) v_read_l(0, mt);
        __builtin_amdgcn_sched_barrier(0);
        // ---- online softmax of column l32 (its 32 keys of this tile sit in lanes l32 and l32 + 32, 16 registers each)
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] *= p.scale; mloc = fmaxf(mloc, sacc[r]); }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        // Lazy reference maximum: softmax is invariant to the value subtracted before exp, so the running reference only has to keep exp()
        // in range.  It moves (and the 128 accumulator registers are rescaled) only when some query's tile maximum exceeds its reference by
        // more than 8 (exp(8) ~ 3e3: far inside float16 / float32 range).  With an exact running maximum 32 queries per wave would trigger the
        // rescale on almost every tile ((1 - 1/t)^32 chance of none moving).
        const bool moved = mloc > m_run + 8.0f;                            // first tile: m_run = -inf
        float alpha = 1.0f;
        if (__builtin_amdgcn_ballot_w64(moved) != 0) {                     // wave-uniform
            const float m_new = fmaxf(m_run, mloc);
            alpha = __expf(m_run - m_new);                                 // exp(-inf) = 0 on the first tile
            m_run = m_new;
            if (t > 0) {
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) oacc[mt][r] *= alpha;
            }
        }
        float psum = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { sacc[r] = __expf(sacc[r] - m_run); psum += sacc[r]; }
        l_run = l_run * alpha + psum;

        if (t < 8) ASTAMP(t * 8 + 4);
        // ---- O^T += V^T P^T: the B operand of k-step j is registers 8j..8j+7 of the probabilities, split in place
        h8 ph[2], pl[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pv = sacc[8 * j + e];
                const _Float16 hi = (_Float16)pv;
                ph[j][e] = hi;
                pl[j][e] = (_Float16)(pv - (float)hi);
            }
        __builtin_amdgcn_sched_barrier(0);
        // pass-major over the 8 channel tiles: consecutive MFMAs write different accumulators
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[mt], ph[0], oacc[mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            v_read_h(0, mt);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[mt], pl[0], oacc[mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            v_read_l(1, mt);                                  // (the lo fragment's last reader issued a pass ago)
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[mt], ph[0], oacc[mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            v_read_h(1, mt);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl[mt], ph[1], oacc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[mt], pl[1], oacc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) oacc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh[mt], ph[1], oacc[mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (t < 8) ASTAMP(t * 8 + 5);
    }
    ASTAMP(120);

    // ---- epilogue: O = O^T^T / row sum (+ res), through LDS so that the stores are whole rows.  All 32 residual pieces of this lane are
    // requested FIRST, in front of the transposition (the query-row registers are dead): as a load under `if (res)` inside the store loop
    // every iteration waited vmcnt(0) — for its own load and, the counter being one for loads and stores, for the previous iteration's
    // store to complete: 21-25 thousand cycles of a 155-thousand-cycle workgroup (tools/attn_stamps.py).
    const size_t row0 = (size_t)b * lq + q0 + wave * 32;
    f32x4 rr[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int idx = lane + i * 64;                                  // 32 rows x 64 float4
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        rr[i] = ps.res ? *reinterpret_cast<const f32x4*>(ps.res + (row0 + (idx >> 6)) * ps.ldo + (idx & 63) * 4) : zero;
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    __syncthreads();                                                    // the K / V images are dead: the float32 image overlays them
    float* lds_o = reinterpret_cast<float*>(smem_at) + wave * (32 * O_LD);      // [32 queries][O_LD]
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            lds_o[l32 * O_LD + mt * 32 + dm3d_acc_row(r, half)] = oacc[mt][r] * inv;
    __builtin_amdgcn_s_waitcnt(0xC07F);                                 // lgkmcnt(0): this wave's own image is complete (no other wave reads it)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int idx = lane + i * 64;
        const int qr = idx >> 6, c4 = idx & 63;
        f32x4 v = *reinterpret_cast<const f32x4*>(lds_o + qr * O_LD + c4 * 4);
        if (ps.res) { v[0] += rr[i][0]; v[1] += rr[i][1]; v[2] += rr[i][2]; v[3] += rr[i][3]; }
        *reinterpret_cast<f32x4*>(ps.out + (row0 + qr) * ps.ldo + c4 * 4) = v;
    }
    ASTAMP(121);
}

}  // namespace

// true when the fused kernel serves this description
static bool attn_fusable(const dm3d_attention_desc* d) {
    return d->precision == DM3D_PREC_H3 && d->fmt == DM3D_FMT_H2 && d->c == AC && d->lq % QT == 0 && d->lk % KT == 0 &&
           (int64_t)AC * d->ldv * 4 < (1ll << 31) && (int64_t)d->lk * d->ldk * 4 < (1ll << 31) &&
           d->ldq % 16 == 0 && d->ldk % 16 == 0 && d->ldv % 16 == 0 && d->stride_k % 16 == 0 && d->stride_vt % 16 == 0 && d->ldo % 4 == 0 &&
           dm3d_aligned16(d->q) && dm3d_aligned16(d->k) && dm3d_aligned16(d->vt) && dm3d_aligned16(d->out) && dm3d_aligned16(d->res);
}

// Launches `count` (<= 4) attention passes of identical (batch, lq, lk, scale) as one grid; returns DM3D_EUNSUPPORTED (without touching
// the error text's meaning for callers that fall back) when a pass does not qualify.
int dm3d_attention_fused_launch(const dm3d_attention_desc* descs, int count, hipStream_t st) {
    if (count < 1 || count > 4) return DM3D_EUNSUPPORTED;
    static const bool off = [] { const char* e = getenv("DM3D_ATTN_FUSED"); return e && e[0] == '0'; }();      // A/B switch
    if (off) return DM3D_EUNSUPPORTED;
    AttnArgs a{};
    for (int i = 0; i < count; ++i) {
        const dm3d_attention_desc* d = &descs[i];
        if (!attn_fusable(d) || d->batch != descs[0].batch || d->lq != descs[0].lq || d->lk != descs[0].lk || d->scale != descs[0].scale)
            return DM3D_EUNSUPPORTED;
        AttnPass& p = a.pass[i];
        p.q = reinterpret_cast<const char*>(d->q); p.ldq = d->ldq;
        p.k = reinterpret_cast<const char*>(d->k); p.ldk = d->ldk; p.sk = d->stride_k;
        p.vt = reinterpret_cast<const char*>(d->vt); p.ldv = d->ldv; p.sv = d->stride_vt;
        p.out = d->out; p.ldo = d->ldo; p.res = d->res;
    }
    a.lq = descs[0].lq; a.lk = descs[0].lk; a.scale = descs[0].scale;
    DM3D_REQUIRE(descs[0].batch <= 65535, "attention: batch %d exceeds grid.y", descs[0].batch);
    constexpr size_t lds_stage = (size_t)2 * (K_TILE + V_TILE) * sizeof(_Float16), lds_out = (size_t)4 * 32 * O_LD * sizeof(float);
    constexpr size_t lds = lds_stage > lds_out ? lds_stage : lds_out;
    static_assert(lds <= 160 * 1024, "LDS");
    static std::atomic<bool> attr_set[64] = {};            // per device: the attribute belongs to the device the launch goes to
    int dev_ = 0;
    DM3D_HIP(hipGetDevice(&dev_));
    DM3D_REQUIRE(dev_ >= 0 && dev_ < 64, "attention: device ordinal %d", dev_);
    if (!attr_set[dev_]) {
        DM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fused_h3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev_] = true;
    }
    dim3 grid((unsigned)(a.lq / QT), (unsigned)descs[0].batch, (unsigned)count);
    hipLaunchKernelGGL(attn_fused_h3, grid, dim3(256), lds, st, a);
    return dm3d_launch_check("attn_fused_h3");
}
