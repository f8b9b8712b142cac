// dm3d_conv_h3v2.hip — (1) the float8 cross-term form ("h3f8", opt-in) of the k3 / stride-1 and 2x2x2 parity Conv3d; (2) what the
// 16x16x32 split-float16 conv kernels share on the host side: the launch bracket (Cin splitting for small grids, zero fill, reduce
// launch), the DM3D_WL_PAIR weight packers, the tile-form predicates.  The default three-pass arithmetic lives in dm3d_conv_h3v3.hip.
//
// Operand geometry (both kernels; helpers in dm3d_conv_h3v2_parts.h):
//   * K = 32 per MFMA = two consecutive taps x 16 channels.  Lane l: row = l & 15, k-group kg = l >> 4; kg >> 1 picks the tap
//     of the pair (= lane half), kg & 1 the 8-channel half (slot of the 64-byte record).  27 taps are padded to 28 (zero weights).
//   * A 16-row tile is a 4 x 4 voxel patch; with the halo row stride padded to 12 the 16 records of a patch are distinct mod 16.
//     The two hardware lane groups of a ds_read_b128, {0-3,12-15,20-27} and {4-11,16-19,28-31}, each mix rows {0-3,12-15} at one
//     slot with rows {4-11} at the other; under the XOR swizzle two such reads collide iff their records differ by +-4 mod 16,
//     so rows {0-3,12-15} take the patch columns dx in {0,1} and rows {4-11} take dx in {2,3} (record mod 4 = dx): conflict-free
//     for every tap.  Weight rows get the same treatment by permuting their LDS position inside each group of 16 (pi_pos).
//   * A wave owns one 8 x 8 z-slice = 2 x 2 patches x 64 output channels = 16 tiles of 16 x 16 (64 accumulator registers).
#include <cstdlib>
#include "dm3d_conv_h3v2_parts.h"

using namespace h3v2;

namespace {

// The "H3F8" arithmetic of dm3d_h3.h: ah.bh on v_mfma_f32_16x16x32_f16, both cross terms on v_mfma_scale_f32_16x16x128_f8f6f4 (K = 128 =
// 4 taps x 16 channels x 2 terms = one weight group; lane k-group kg takes tap kg of the group).  Per group a wave issues 32 float16 + 16
// float8 MFMAs (512 + ~455 cycles) where the three-pass arithmetic issues 96 float16 MFMAs (1536).  8-slice bricks (512 threads, one
// workgroup per CU), weight groups of 4 taps (16 KB by LDS-DMA) through three buffers with two groups of lead, counted vmcnt waits
// and raw barriers.  MODE 0: float32 input as is; 1: float32 input through the fused norm + SiLU prologue; 2: x1 already activated and
// split (DM3D_FMT_H2, written by the producing conv's epilogue).
#define F8READ(p) *reinterpret_cast<const u32x4*>(p)
template <int KS, int MODE>
__global__ __launch_bounds__(512, 2) void conv3d_igemm_h3f8(const ConvArgs p) {
    constexpr int TD = 8, NBUF = 3;
    constexpr int TH = 8, TW = 8, CK = 16, NT = 64, NTHR = TD * 64;
    constexpr int HD = TD - 1 + KS, HH = TH - 1 + KS, HW = TW - 1 + KS, HWP = 12;
    constexpr int HVOX = HD * HH * HW;
    constexpr int G = 4;                                     // taps per weight group (= one K = 128 step of the float8 cross terms)
    constexpr int TAPS = KS * KS * KS, TAPSP = (TAPS + G - 1) / G * G;
    constexpr int NG = TAPSP / G;                            // groups per chunk
    constexpr int NSLOT = (HVOX * 2 + NTHR - 1) / NTHR;
    constexpr int WGRP = G * NT * REC;                       // halfs per weight group (16 KB)
    constexpr int WSLOT = WGRP * 2 / 16 / NTHR;              // 16-byte pieces per thread: 4 (256 threads) or 2 (512)
    static_assert(WGRP * 2 / 16 % NTHR == 0, "weight group must be a whole number of pieces per thread");

    extern __shared__ __attribute__((aligned(16))) _Float16 smem_v2[];
    _Float16* lds_w = smem_v2;                  // [NBUF][G][NT][REC]   (first: every weight read is base + a 16-bit immediate)
    _Float16* lds_in = smem_v2 + NBUF * WGRP;   // [HREC][REC]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, q = (lane >> 4) & 1, row = lane & 15;

    int brick = blockIdx.x;
    const int bpv = p.bd * p.bh * p.bw;
    const int b = brick / bpv;
    brick -= b * bpv;
    const int oz0 = (brick / (p.bh * p.bw)) * TD;
    const int oy0 = ((brick / p.bw) % p.bh) * TH;
    const int ox0 = (brick % p.bw) * TW;
    // blockIdx.y = ntile + ntiles * khalf.  ksplit == 2 (small grids, linear epilogue): this workgroup contracts chunks
    // [c_lo, c_hi) only and adds its partial sums into the zeroed output; the khalf == 0 half also carries bias / vec / residual.
    // 0 + a + b == 0 + b + a in floating point, so the result does not depend on which half arrives first.
    const int ntiles = p.coutpad / NT;
    const int ntile = blockIdx.y % ntiles, khalf = blockIdx.y / ntiles;
    const int c_lo = khalf * (p.nchunks / p.ksplit), c_hi = c_lo + p.nchunks / p.ksplit;
    int padz = p.padz, pady = p.pady, padx = p.padx, ooz = p.ooz, ooy = p.ooy, oox = p.oox;
    const _Float16* wbase = static_cast<const _Float16*>(p.wpk);
    if (p.parity) {
        const int par = blockIdx.z;
        ooz = par >> 2; ooy = (par >> 1) & 1; oox = par & 1;
        padz = 1 - ooz; pady = 1 - ooy; padx = 1 - oox;
        wbase += (size_t)par * p.w_parity_stride;
    }

    // ---- halo staging slots (identical to dm3d_conv_h3.hip)
    const int piece = tid & 1;
    int gvox[NSLOT], st_off[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        const int hv = (tid >> 1) + j * (NTHR / 2);
        int g = -1;
        if (hv < HVOX) {
            const int hz = hv / (HH * HW), hy = (hv / HW) % HH, hx = hv % HW;
            const int iz = oz0 - padz + hz, iy = oy0 - pady + hy, ix = ox0 - padx + hx;
            if (iz >= 0 && iz < p.ind && iy >= 0 && iy < p.inh && ix >= 0 && ix < p.inw)
                g = ((b * p.ind + iz) * p.inh + iy) * p.inw + ix;
        }
        gvox[j] = g;
        const int v = (hv / HW) * HWP + hv % HW;
        st_off[j] = hv < HVOX ? v * REC + ((piece ^ swz(v)) << 3) : -1;
    }

    // ---- operand addressing: this lane's voxel inside a 4 x 4 patch, patch (0,0) of the wave's z-slice, tap (0,0,0)
    const int a_rec0 = (wave * HH + (row & 3)) * HWP + dx_of_row(row);
    // weight rows: LDS position PI(row) inside the 16-column tile, the lane half picks the tap of the pair
    const int b_pos = pi_pos(row);
    const int b_hi = (half * NT + b_pos) * REC + ((q ^ swz(b_pos)) << 3);

    f32x4v acc[4][4];
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[pi][ni] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // weights go global -> LDS by LDS-DMA (no registers, no ds_write): the packed image IS the LDS image, so group gg is a
    // linear 16 KB copy; wave w moves the 1 KB pieces w, w+TD, ...  Buffer = group index mod NBUF, counted from this workgroup's
    // first group (NG is odd for k3, so the phase differs from chunk to chunk: a running counter).
    const char* w_img = reinterpret_cast<const char*>(wbase + (size_t)ntile * p.nchunks * NG * WGRP) + wave * 1024 + lane * 16;
    auto fetch_w = [&](int gg, int buf) {
        const char* src = w_img + (size_t)gg * (WGRP * 2);
        char* dst = reinterpret_cast<char*>(lds_w) + buf * (WGRP * 2) + wave * 1024;
#pragma unroll
        for (int i = 0; i < WSLOT; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * (NTHR * 16)),
                                             (__attribute__((address_space(3))) void*)(dst + i * (NTHR * 16)), 16, 0, 0);
    };
    const int g_first = c_lo * NG, g_end = c_hi * NG;
    fetch_w(g_first, 0);
    // every DMA below is issued unconditionally (past the end the last group is fetched again into a free buffer): with
    // conditional issues hipcc cannot count what is in flight and falls back to vmcnt(0) in front of the halo registers' first use.
    fetch_w(g_first + 1 < g_end ? g_first + 1 : g_end - 1, 1);
    int wb = 0;                                  // buffer of the group about to be consumed

    constexpr bool pro = MODE == 1, xh2 = MODE == 2;
    f32x4 raw0[NSLOT], raw1[NSLOT];
    f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
    bool ok0 = false, ok1 = false;
    // One slot (two 16-byte loads per thread) of chunk ch's halo; all slots of a chunk read the same channel piece.
    auto load_halo_slot = [&](int ch, int j) {
        if (xh2) {                       // record ch of each voxel: hi piece at slot `piece`, lo piece at slot 2 + piece
            const char* qp = reinterpret_cast<const char*>(p.x1) + (size_t)ch * 64 + piece * 16 + (size_t)(gvox[j] >= 0 ? gvox[j] : 0) * ((size_t)p.c1 * 4);
            raw0[j] = *reinterpret_cast<const f32x4*>(qp);
            raw1[j] = *reinterpret_cast<const f32x4*>(qp + 32);
            return;
        }
        const int c0 = ch * CK;
        const float* src;
        int ldc, cb;
        if (c0 < p.c1) { src = p.x1; ldc = p.c1; cb = c0; } else { src = p.x2; ldc = p.c2; cb = c0 - p.c1; }
        const int cpos = cb + piece * 8;
        const int off0 = cpos < ldc ? cpos : 0, off1 = cpos + 4 < ldc ? cpos + 4 : 0;
        const float* qp = src + (size_t)(gvox[j] >= 0 ? gvox[j] : 0) * ldc;
        raw0[j] = *reinterpret_cast<const f32x4*>(qp + off0);
        raw1[j] = *reinterpret_cast<const f32x4*>(qp + off1);
    };
    // what the conversion of chunk ch needs beside the voxels: channel validity and, behind a fused norm, the scale / shift vectors (PLOADS loads)
    constexpr int PLOADS = pro ? 4 : 0;
    auto load_chunk_params = [&](int ch) {
        if (xh2) return;
        const int c0 = ch * CK;
        const int ldc = c0 < p.c1 ? p.c1 : p.c2, cpos = (c0 < p.c1 ? c0 : c0 - p.c1) + piece * 8;
        ok0 = cpos < ldc;
        ok1 = cpos + 4 < ldc;
        if (pro) {
            const int s0 = ok0 ? c0 + piece * 8 : 0, s1 = ok1 ? c0 + piece * 8 + 4 : 0;
            const size_t bo = (size_t)b * p.pro_bstride;
            sc0 = *reinterpret_cast<const f32x4*>(p.pscale + bo + s0);
            sh0 = *reinterpret_cast<const f32x4*>(p.pshift + bo + s0);
            sc1 = *reinterpret_cast<const f32x4*>(p.pscale + bo + s1);
            sh1 = *reinterpret_cast<const f32x4*>(p.pshift + bo + s1);
        }
    };
    auto load_halo = [&](int ch) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) load_halo_slot(ch, j);
        load_chunk_params(ch);
    };
    // The next chunk's halo is requested inside this chunk's MFMA phase.  SPREAD: one slot per group (groups 0 .. NSLOT-1, behind the
    // group's weight DMA) instead of all of them in front of group 0: requests return in order, so with everything issued up front the
    // whole 64 KB per workgroup — 16 MB across the chip, in one burst — had to land before the wait for the DMA issued one group later
    // could pass (two groups, ~2.8 us); measured with cache-resident halos the kernel was 8-10 % faster, i.e. that wait was exposed.
    constexpr bool SPREAD = NG > NSLOT;
    // vector-memory requests issued inside group g of a chunk (beside its weight DMA)
    auto halo_ops = [&](int g) { return SPREAD ? ((g >= 0 && g < NSLOT) ? 2 + (g == 0 ? PLOADS : 0) : 0) : 0; };
    load_halo(c_lo);
    // Drain here, with an instruction hipcc's wait-count pass sees: the chunk loop's header otherwise merges "first chunk: the
    // halo loads are the newest requests" with "later chunks: 7 groups of DMAs were issued behind them" into vmcnt(0) on every iteration.
    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), expcnt / lgkmcnt untouched

    // Conversion of a chunk's halo registers (prologue norm + SiLU, float16 split / float8 operands) into the pieces the LDS image takes,
    // in place.  (Tried: converting chunk ch+1 inside chunk ch's MFMA phase, the two waves of a SIMD in different groups, so that only
    // [barrier, LDS stores, barrier] remain between chunks — 2.5 % slower on the three-pass form, 5 % on the float8 form: the VALU work
    // delays that wave's MFMAs more than the idle boundary costs.)
    auto convert_slot = [&](const int j) {
        {
            const bool in = gvox[j] >= 0;
            h8 shi_j, slo_j;
            f32x4 v0 = raw0[j], v1 = raw1[j];
            if (pro) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v0[e] = dm3d_silu(fmaf(v0[e], sc0[e], sh0[e]));
                    v1[e] = dm3d_silu(fmaf(v1[e], sc1[e], sh1[e]));
                }
            }
            if (xh2) {
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                shi_j = __builtin_bit_cast(h8, in ? v0 : z);
                slo_j = __builtin_bit_cast(h8, in ? v1 : z);
                slo_j = h2_to_x8(shi_j, slo_j);                         // (hi16, lo16) of the hand-off format -> [ah8 | al8]
            } else {
                split8_f8(v0, v1, (in && ok0) ? DM3D_F8_LIMIT : 0.0f, (in && ok1) ? DM3D_F8_LIMIT : 0.0f, shi_j, slo_j);
            }
            raw0[j] = __builtin_bit_cast(f32x4, shi_j);
            raw1[j] = __builtin_bit_cast(f32x4, slo_j);
        }
    };
    auto convert_halo = [&]() {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) convert_slot(j);
    };
    int a_rec = a_rec0;

    for (int ch = c_lo; ch < c_hi; ++ch) {
        asm volatile("" : "+v"(a_rec));      // keeps the 14 per-pair operand addresses from being hoisted out of the chunk loop (spills)
        convert_halo();
        // With weight groups in flight across barriers (NBUF == 3) every barrier is a raw s_barrier behind counted waits:
        // __syncthreads() would drain the VM counter, i.e. wait for the DMAs that are meant to stay in flight (cdna_hip_programming.md,
        // "glds with >1 tile in flight across the barrier").
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's reads of the previous halo image are retired
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            if (st_off[j] >= 0) {
                *reinterpret_cast<h8*>(lds_in + st_off[j]) = __builtin_bit_cast(h8, raw0[j]);
                *reinterpret_cast<h8*>(lds_in + (st_off[j] ^ 16)) = __builtin_bit_cast(h8, raw1[j]);
            }
        }
        {
            // outstanding, oldest first: [this chunk's first group] [its second group] (the first chunk: + its own halo loads, which the
            // conversion above already waited for): the first group must have landed; the halo stores must be visible
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WSLOT) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        const int ch_next = ch + 1 < c_hi ? ch + 1 : ch;           // (past the end: the last chunk again, unconditional like the DMAs)
        if (!SPREAD) load_halo(ch_next);
        __builtin_amdgcn_sched_barrier(0);
        // requests of group g behind its weight DMA
        auto halo_in_group = [&](const int g) {
            if (!SPREAD) return;
            if (g < NSLOT) load_halo_slot(ch_next, g);
            if (g == 0) load_chunk_params(ch_next);
        };

        // end of a weight group: advance the ring, wait for the next group's DMA, one barrier
        auto group_end = [&](const int g) {
            __builtin_amdgcn_sched_barrier(0);
            wb = wb + 1 == NBUF ? 0 : wb + 1;
            const bool last_group = g + 1 == NG;
            if (!last_group) {
                {
                    // group g+1 must have landed; newer than it in the queue: group g+2's DMA and, behind the chunk's first iteration,
                    // the next chunk's halo loads (issued between the DMAs of g+1 and g+2)
                    // (SPREAD: the requests issued beside the DMAs of groups g-1 and g, see halo_ops)
                    const int extra = SPREAD ? halo_ops(g - 1) + halo_ops(g) : (g == 0 ? 2 * NSLOT : 0);
                    if (extra == 0)      asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WSLOT) : "memory");
                    else if (extra == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WSLOT + 2) : "memory");
                    else if (extra == 4) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WSLOT + 4) : "memory");
                    else if (extra == 6) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WSLOT + 6) : "memory");
                    else if (extra == 8) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WSLOT + 8) : "memory");
                    else if (extra == 2 * NSLOT) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WSLOT + 2 * NSLOT) : "memory");
                    else                 asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WSLOT) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
            }
        };
        // ---- the float8 cross-term form's group: 32 float16 MFMAs (ah.bh of the two tap pairs) + 16 float8 MFMAs (one K = 128 step:
        // ah.bl + al.bh of all four taps), software-pipelined by hand: the LDS reads of a batch are issued in front of the previous batch's
        // MFMAs (sched_barrier keeps them there), so the matrix pipe never waits for an operand that was requested a few cycles ago — left
        // to the compiler every batch of reads sat directly in front of its own MFMAs behind an s_waitcnt, and with all eight waves of the
        // workgroup in the same phase (one barrier per group) nothing else filled those gaps.
        // float8 operands: lane (half, q) contributes, as its 32 K-bytes, slot 2 + q (channels 8q .. 8q+7: [ah8 x8 | al8 x8], weights
        // [bl8 x8 | bh8 x8]) of tap `half` of the group's first pair and of its second pair — the very addresses the three-pass form reads
        // its lo operands from, conflict-free like those (a k-group = tap mapping cost 5-7 LDS cycles per read instead of 4).
        typedef int i32x8 __attribute__((ext_vector_type(8)));
        h8 a_carry[4];                                                   // ah of the next group's first pair, read during this group's tail
        auto group_body_f8 = [&](const int g) {
            const bool last_group = g + 1 == NG;
            {
                // group g+2 goes into the buffer group g-1 occupied: every wave is past g-1 (the barrier that opened this iteration)
                const int nxt = ch * NG + g + 2;
                const int b2 = wb + 2 >= 3 ? wb - 1 : wb + 2;
                fetch_w(nxt < g_end ? nxt : g_end - 1, b2);
            }
            halo_in_group(g);
            __builtin_amdgcn_sched_barrier(0);
            const _Float16* wbuf = lds_w + wb * WGRP;
            // halo offset (halfs) of this lane's hi slot: tap pair pr of group gg, patch pi (pad taps re-read the last real tap: zero weights)
            auto a_off = [&](int gg, int pr, int pi) {
                const int ta = gg * G + pr * 2, tb = ta + 1;
                const int tac = ta < TAPS ? ta : TAPS - 1, tbc = tb < TAPS ? tb : TAPS - 1;
                const int rec_a = ((tac / (KS * KS)) * HH + (tac / KS) % KS) * HWP + tac % KS;
                const int rec_b = ((tbc / (KS * KS)) * HH + (tbc / KS) % KS) * HWP + tbc % KS;
                const int v = a_rec + (half ? rec_b : rec_a) + 4 * (pi & 1);
                return v * REC + ((q ^ swz(v)) << 3) + (pi >> 1) * (48 * REC);
            };
            auto rd_a = [&](int gg, int pr, h8 (&a)[4]) {
#pragma unroll
                for (int pi = 0; pi < 4; ++pi) a[pi] = __builtin_bit_cast(h8, F8READ(lds_in + a_off(gg, pr, pi)));
            };
            auto rd_b = [&](int pr, int nb, h8 (&b)[2]) {
#pragma unroll
                for (int k = 0; k < 2; ++k) b[k] = __builtin_bit_cast(h8, F8READ(wbuf + (pr * 2 * NT + (nb * 2 + k) * 16) * REC + b_hi));
            };
            auto rd_a8 = [&](int pi) {
                const u32x4 x0 = F8READ(lds_in + (a_off(g, 0, pi) ^ 16)), x1 = F8READ(lds_in + (a_off(g, 1, pi) ^ 16));
                return i32x8{(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
            };
            auto mma_hi = [&](const h8 (&a)[4], const h8 (&b)[2], int nb) {
#pragma unroll
                for (int pi = 0; pi < 4; ++pi)
#pragma unroll
                    for (int k = 0; k < 2; ++k)
                        acc[pi][nb * 2 + k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[pi], b[k], acc[pi][nb * 2 + k], 0, 0, 0);
            };
            auto mma_f8 = [&](const i32x8& a8, const i32x8 (&b8)[4], int pi) {
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[pi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8[ni], acc[pi][ni], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            };
            h8 a0[4], a1[4], b0[2], b1[2], b2[2], b3[2];
            if (g == 0) {
                rd_a(g, 0, a0);
            } else {
#pragma unroll
                for (int pi = 0; pi < 4; ++pi) a0[pi] = a_carry[pi];
            }
            rd_b(0, 0, b0);
            rd_b(0, 1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mma_hi(a0, b0, 0);
            __builtin_amdgcn_sched_barrier(0);
            rd_a(g, 1, a1);
            rd_b(1, 0, b2);
            __builtin_amdgcn_sched_barrier(0);
            mma_hi(a0, b1, 1);
            __builtin_amdgcn_sched_barrier(0);
            rd_b(1, 1, b3);
            __builtin_amdgcn_sched_barrier(0);
            mma_hi(a1, b2, 0);
            __builtin_amdgcn_sched_barrier(0);
            i32x8 b8[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const u32x4 w0 = F8READ(wbuf + (ni * 16) * REC + (b_hi ^ 16));
                const u32x4 w1 = F8READ(wbuf + (2 * NT + ni * 16) * REC + (b_hi ^ 16));
                b8[ni] = i32x8{(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
            }
            i32x8 x0 = rd_a8(0);
            __builtin_amdgcn_sched_barrier(0);
            mma_hi(a1, b3, 1);
            __builtin_amdgcn_sched_barrier(0);
            i32x8 x1 = rd_a8(1);
            __builtin_amdgcn_sched_barrier(0);
            mma_f8(x0, b8, 0);
            __builtin_amdgcn_sched_barrier(0);
            x0 = rd_a8(2);
            __builtin_amdgcn_sched_barrier(0);
            mma_f8(x1, b8, 1);
            __builtin_amdgcn_sched_barrier(0);
            x1 = rd_a8(3);
            __builtin_amdgcn_sched_barrier(0);
            mma_f8(x0, b8, 2);
            __builtin_amdgcn_sched_barrier(0);
            if (!last_group) rd_a(g + 1, 0, a_carry);
            __builtin_amdgcn_sched_barrier(0);
            mma_f8(x1, b8, 3);
            group_end(g);
        };
#pragma unroll
        for (int g = 0; g < NG; ++g) group_body_f8(g);
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the redundant tail fetches: nothing may land in LDS the skip phase reuses

    const Brick br = {b, oz0, oy0, ox0, ooz, ooy, oox, ntile, khalf};
    if constexpr (KS == 3) {                        // (the launcher admits a skip conv behind k3 / stride 1 only)
        static_assert(KS != 3 || skip_lds_halfs<TD>() <= HD * HH * HWP * REC + NBUF * WGRP, "skip phase LDS carve");
        skip_phase<TD>(p, smem_v2, acc, br);
    }
    epilogue<TD>(p, acc, br);
}

// out = epilogue(sum of the ksplit partial-sum images, added in image order): + bias[c] + vec[row(b)][c] -> ReLU -> PReLU -> + res ->
// ReLU.  One thread per 4 consecutive channels (or per element when cout % 4 != 0).
__global__ __launch_bounds__(256) void conv_split_reduce_kernel(const float* __restrict__ part, int nsplit, long stride, float* __restrict__ out,
                                                                int cout, long per_sample, const float* __restrict__ bias,
                                                                const float* __restrict__ vec, const int* __restrict__ vec_idx, int vec_ld,
                                                                int relu, const float* __restrict__ prelu, const float* __restrict__ res,
                                                                int relu_out, int vec4, int* range_flag, float range_limit) {
    const int w = vec4 ? 4 : 1;
    const long total = stride / w;
    float amax = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long e0 = i * w;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < nsplit; ++s) {
            if (vec4) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(part + (size_t)s * stride + e0);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += q[j];
            } else {
                v[0] += part[(size_t)s * stride + e0];
            }
        }
        const unsigned e32 = (unsigned)e0;                  // the launcher guarantees stride < 2^31
        const int c0 = (int)(e32 % (unsigned)cout);
        const unsigned b = e32 / (unsigned)per_sample;
        const int vrow = vec ? (vec_idx ? vec_idx[b] : (int)b) : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= w) break;
            float x = v[j];
            if (bias) x += bias[c0 + j];
            if (vec) x += vec[(size_t)vrow * vec_ld + c0 + j];
            if (relu) x = fmaxf(x, 0.0f);
            if (prelu) { const float al = prelu[e32 % (unsigned)per_sample + j]; x = x > 0.0f ? x : al * x; }
            if (res) x += res[e0 + j];
            if (relu_out) x = fmaxf(x, 0.0f);
            DM3D_AMAX(amax, x);
            out[e0 + j] = x;
        }
    }
    if (range_flag && amax > range_limit) *range_flag = 1;
}

// Zero fill as a kernel of our own: a hipMemsetAsync captured into the per-step HIP graph becomes a memset node, and replays of
// that graph were observed to race it against the atomic adds of the following conv (two full T = 1000 chains diverged after
// ~90 steps; eager launches and graphs without memset nodes did not).  A kernel node is ordered like every other launch.
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, long n4, long n) {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) reinterpret_cast<f32x4*>(p)[i] = z;
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[n4 * 4 + threadIdx.x] = 0.f;
}

}  // namespace

// Everything around the conv launch itself that the v2 and v3 kernels share: brick counts, Cin splitting for small grids (zero fill +
// two-way atomic add, or raw partial sums into caller scratch + a reduce launch that applies the epilogue), 16-byte epilogue eligibility.
int dm3d_h3v2_pre_launch(ConvArgs& a, int td, bool f8, H3v2Launch& L, hipStream_t st, int force_ksplit) {
    a.bd = (a.od + td - 1) / td;
    a.bh = (a.oh + 7) / 8;
    a.bw = (a.ow + 7) / 8;
    // Small grids (the 8^3 level at B = 32 has 64 bricks x 4 channel tiles = one workgroup per CU, i.e. one wave per SIMD with
    // nothing to hide its barriers and LDS latency behind: in-kernel stamps showed 58 % MFMA occupancy inside the tap loop there;
    // at B = 1 that level has 8 workgroups for 256 CUs) split the Cin chunks over several workgroups per brick.
    const bool with_scratch = a.scratch != nullptr;
    a.ksplit = force_ksplit > 0 ? force_ksplit
             : (td != 4 || a.out_h2 || a.post_scale) ? 1 : dm3d_conv_h3v2_ksplit(a, with_scratch);     // the fused output forms live in the plain epilogue
    const size_t out_elems = (size_t)a.batch * a.fd * a.fh * a.fw * a.cout;
    a.split_atomic = 0;
    a.split_stride = 0;
    {
        auto al16 = [](const void* q) { return (reinterpret_cast<size_t>(q) & 15) == 0; };
        a.epi_vec4 = a.cout % 4 == 0 && al16(a.out) && al16(a.bias) && al16(a.res) && al16(a.prelu) && al16(a.post_scale) && al16(a.post_shift)
                     && al16(a.vec) && (a.vec == nullptr || a.vec_ld % 4 == 0);
    }
    ConvArgs& k = L.k;
    k = a;
    if (f8) k.wpk = a.wpk_f8;
    const bool linear = !a.relu && !a.prelu && !a.relu_out && a.res != a.out && a.x1 != a.out && a.x2 != a.out;
    static const bool no_atomic = [] { const char* e = getenv("DM3D_CONV_NO_ATOMIC"); return e && e[0] == '1'; }();
    const bool atomic2 = a.ksplit == 2 && linear && a.nchunks >= 8 && !(no_atomic && with_scratch);       // cheaper than a reduce launch when two parts suffice
    L.out_elems = out_elems;
    L.reduce = false;
    if (a.ksplit > 1 && (!with_scratch || atomic2)) {         // two halves, order-independent atomic add into the zeroed output
        k.split_atomic = 1;
        long zg = ((long)(out_elems / 4) + 255) / 256;
        if (zg > 4096) zg = 4096;
        if (zg < 1) zg = 1;
        hipLaunchKernelGGL(zero_f32_kernel, dim3((unsigned)zg), dim3(256), 0, st, a.out, (long)(out_elems / 4), (long)out_elems);
        if (int zrc = dm3d_launch_check("zero_f32_kernel")) return zrc;
    } else if (a.ksplit > 1) {                                // raw partial sums -> scratch; epilogue in the reduce launch
        DM3D_REQUIRE((size_t)a.scratch_bytes >= out_elems * sizeof(float) * a.ksplit, "conv: scratch of %ld bytes is too small", a.scratch_bytes);
        DM3D_REQUIRE(out_elems < (1ull << 31), "conv: split-K output of %zu elements overflows the reduce kernel's 32-bit index", out_elems);
        k.out = static_cast<float*>(a.scratch);
        k.split_stride = (long)out_elems;
        k.bias = nullptr; k.vec = nullptr; k.res = nullptr; k.relu = 0; k.prelu = nullptr; k.relu_out = 0;
        k.range_flag = nullptr;                               // the reduce launch checks the finished values
        L.reduce = true;
    }
    return DM3D_OK;
}

int dm3d_h3v2_post_launch(const ConvArgs& a, const H3v2Launch& L, hipStream_t st) {
    if (!L.reduce) return DM3D_OK;
    const size_t out_elems = L.out_elems;
    const long n4 = (long)(out_elems / 4);                    // cout % 4 == 0 is not required of cout: fall back to scalar lanes
    const bool vec4 = a.cout % 4 == 0;
    const long work = vec4 ? n4 : (long)out_elems;
    long g = (work + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(conv_split_reduce_kernel, dim3((unsigned)g), dim3(256), 0, st, static_cast<const float*>(a.scratch), a.ksplit,
                       (long)out_elems, a.out, a.cout, (long)a.fd * a.fh * a.fw * a.cout, a.bias, a.vec, a.vec_idx, a.vec_ld, a.relu, a.prelu,
                       a.res, a.relu_out, vec4 ? 1 : 0, a.range_flag, a.range_limit);
    return dm3d_launch_check("conv_split_reduce_kernel");
}

namespace {

template <int KS, int MODE>
int launch_f8(ConvArgs& a, hipStream_t st) {
    constexpr int TD = 8, NBUF = 3;
    constexpr int HREC = (TD - 1 + KS) * (7 + KS) * 12;
    constexpr size_t lds = (size_t)(HREC * REC + NBUF * 4 * 64 * REC) * sizeof(_Float16);
    static_assert(lds <= 160 * 1024, "one workgroup per CU");
    static bool attr_set = false;
    if (!attr_set) {
        DM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_igemm_h3f8<KS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    H3v2Launch L;
    if (int rc = dm3d_h3v2_pre_launch(a, TD, true, L, st)) return rc;
    dim3 grid((unsigned)(a.batch * a.bd * a.bh * a.bw), (unsigned)(a.coutpad / 64 * a.ksplit), a.parity ? 8u : 1u);
    hipLaunchKernelGGL((conv3d_igemm_h3f8<KS, MODE>), grid, dim3(TD * 64), lds, st, L.k);
    if (int rc = dm3d_launch_check("conv3d_igemm_h3f8")) return rc;
    return dm3d_h3v2_post_launch(a, L, st);
}

// weight image of the v2 kernel: [coutpad/64][cinpad/16][TAPSP][64 positions][REC]; position 16*t16 + PI(c) holds output channel
// 64*ntile + 16*t16 + c; taps >= taps are zero; slots swizzled by the position.  mode: 0 plain, 1 UpSample sums, 2 Conv3DTranspose
__global__ __launch_bounds__(256) void pack_weights_h3v2_kernel(const float* __restrict__ w, int taps, int tapsp, int cin, int cout,
                                                                int nchunks, int ntiles, float scale, const float* in_scale,
                                                                _Float16* __restrict__ out, int mode, int f8) {
    const long nrec = (long)ntiles * nchunks * tapsp * 64;
    const int npar = (mode == 1 || mode == 2) ? 8 : 1;
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < nrec * 16 * npar; i0 += (long)gridDim.x * 256) {
        const int par = (int)(i0 / (nrec * 16));
        const long i = i0 % (nrec * 16);
        const int k = (int)(i & 15);
        const long rec = i >> 4;
        const int pos = (int)(rec % 64);
        const int tap = (int)((rec / 64) % tapsp);
        const int chunk = (int)((rec / (64L * tapsp)) % nchunks);
        const int nt = (int)(rec / (64L * tapsp * nchunks));
        // invert PI inside the group of 16
        const int p16 = pos & 15;
        int c = 0;
        for (int cc = 0; cc < 16; ++cc) if (pi_pos(cc) == p16) c = cc;
        const int ci = chunk * 16 + k, co = nt * 64 + (pos & ~15) + c;
        float v = 0.f;
        if (ci < cin && co < cout && tap < taps) {
            if (mode == 3) {
                // Winograd F(2,3) along x (dm3d_conv_h3w.hip): virtual tap = 2 * step + h, step = 5 * t + tap pair; the pair's two (dz, dy)
                // taps, lane half h picking one: (dz, 0) | (dz, 1) for pairs 0-2, (0, 2) | zero pad, (1, 2) | (2, 2) — two per-lane operand
                // bases serve all five (the kernel's a_pair); transform term t of the tap's three x taps
                const int step = tap >> 1, t = step / 5;
                const int tq = (step % 5) < 3 ? (step % 5) * 3 + (tap & 1) : ((step % 5) == 3 ? ((tap & 1) ? -1 : 2) : ((tap & 1) ? 8 : 5));
                if (tq >= 0) {
                    const float g0 = w[((long)(tq * 3 + 0) * cin + ci) * cout + co], g1 = w[((long)(tq * 3 + 1) * cin + ci) * cout + co],
                                g2 = w[((long)(tq * 3 + 2) * cin + ci) * cout + co];
                    v = t == 0 ? g0 : (t == 1 ? 0.5f * ((g0 + g2) + g1) : (t == 2 ? 0.5f * ((g0 + g2) - g1) : g2));
                }
            } else
            v = mode == 1 ? dm3d_up_weight(w, cin, cout, par, tap, ci, co)
              : mode == 2 ? dm3d_convt_weight(w, cin, cout, par, tap, ci, co) : w[((long)tap * cin + ci) * cout + co];
            if (in_scale) v *= in_scale[ci];
            v *= scale;
        }
        const _Float16 hi = (_Float16)v;
        _Float16* r = out + ((long)par * nrec + rec) * REC;
        const int sw = (pos >> 2) & 3;
        r[(((k >> 3) ^ sw) << 3) + (k & 7)] = hi;
        if (f8) {
            // H3F8 record (dm3d_h3.h): slot 2 + (k >> 3) = [bl8 x8 | bh8 x8] with bl8 = fp8((w - hi) * 4), bh8 = fp8(hi * 2^-11)
            unsigned char* s8 = reinterpret_cast<unsigned char*>(r + (((2 + (k >> 3)) ^ sw) << 3));
            s8[k & 7] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32((v - (float)hi) * 4.0f, 0.0f, 0, false) & 0xff);
            s8[8 + (k & 7)] = (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32((float)hi * 0.00048828125f, 0.0f, 0, false) & 0xff);
        } else {
            r[(((2 + (k >> 3)) ^ sw) << 3) + (k & 7)] = (_Float16)(v - (float)hi);
        }
    }
}

// skip-conv weight image: [coutpad/64][npairs][2 chunks][64 positions][REC] from a Keras 1x1 kernel [cin][cout]; chunk t of pair i holds
// input channels (2i+t)*16 .. +15 (zeros past cin), rows permuted / slots swizzled like the main image
__global__ __launch_bounds__(256) void pack_skip_h3v2_kernel(const float* __restrict__ w, int cin, int cout, int npairs, int ntiles,
                                                             float scale, _Float16* __restrict__ out) {
    const long nrec = (long)ntiles * npairs * 2 * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nrec * 16; i += (long)gridDim.x * 256) {
        const int k = (int)(i & 15);
        const long rec = i >> 4;
        const int pos = (int)(rec % 64);
        const int t = (int)((rec / 64) % 2);
        const int pair = (int)((rec / 128) % npairs);
        const int nt = (int)(rec / (128L * npairs));
        const int p16 = pos & 15;
        int c = 0;
        for (int cc = 0; cc < 16; ++cc) if (pi_pos(cc) == p16) c = cc;
        const int ci = (pair * 2 + t) * 16 + k, co = nt * 64 + (pos & ~15) + c;
        const float v = (ci < cin && co < cout) ? w[(long)ci * cout + co] * scale : 0.0f;
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)(v - (float)hi);
        _Float16* r = out + rec * REC;
        const int sw = (pos >> 2) & 3;
        r[(((k >> 3) ^ sw) << 3) + (k & 7)] = hi;
        r[(((2 + (k >> 3)) ^ sw) << 3) + (k & 7)] = lo;
    }
}

}  // namespace

int64_t dm3d_h3v2_skip_image_bytes(int cin, int cout) {
    return (int64_t)(dm3d_round_up(cin, 32) / 32) * 2 * dm3d_round_up(cout, 64) * REC * (int64_t)sizeof(_Float16);
}

int dm3d_pack_skip_h3v2(const float* keras_kernel, int cin, int cout, int w_exp, void* packed, hipStream_t st) {
    const int npairs = (int)(dm3d_round_up(cin, 32) / 32), ntiles = (int)(dm3d_round_up(cout, 64) / 64);
    hipLaunchKernelGGL(pack_skip_h3v2_kernel, dim3(1024), dim3(256), 0, st, keras_kernel, cin, cout, npairs, ntiles, ldexpf(1.0f, w_exp),
                       static_cast<_Float16*>(packed));
    return dm3d_launch_check("pack_skip_h3v2_kernel");
}

// Workgroups per brick along Cin.  Goal: at least ~2 workgroups per CU (512) while every part keeps >= 2 chunks.  Without scratch
// only the two-way atomic form exists, and only behind a linear epilogue.
int dm3d_conv_h3v2_ksplit(const ConvArgs& a, bool with_scratch) {
    static const int mode = [] { const char* e = getenv("DM3D_CONV_KSPLIT"); return e ? atoi(e) : -1; }();   // 0: never split (A/B, debugging)
    if (mode == 0) return 1;
    const long bd = (a.od + 3) / 4, bh = (a.oh + 7) / 8, bw = (a.ow + 7) / 8;
    const long wgs = (long)a.batch * bd * bh * bw * (a.coutpad / 64) * (a.parity ? 8 : 1);
    static const long wg_limit = [] { const char* e = getenv("DM3D_CONV_SPLIT_WGS"); return e ? atol(e) : 256L; }();   // A/B knob
    if (wgs > wg_limit || a.nchunks < 4) return 1;
    if (!with_scratch) {
        const bool linear = !a.relu && !a.prelu && !a.relu_out && a.res != a.out && a.x1 != a.out && a.x2 != a.out;
        return (linear && a.nchunks >= 8 && a.nchunks % 2 == 0) ? 2 : 1;
    }
    // A/B knobs (read once: dm3d_conv_scratch_bytes and the launch must agree): least chunks per part, workgroups to aim for, most parts
    static const int min_chunks = [] { const char* e = getenv("DM3D_CONV_SPLIT_MINCHUNKS"); return e ? atoi(e) : 2; }();
    static const long target = [] { const char* e = getenv("DM3D_CONV_SPLIT_TARGET"); return e ? atol(e) : 512L; }();
    static const int max_parts = [] { const char* e = getenv("DM3D_CONV_SPLIT_MAXPARTS"); return e ? atoi(e) : 16; }();
    int best = 1;                                             // smallest divisor that fills the chip, else the largest allowed
    for (int d = 2; d <= max_parts; ++d) {
        if (a.nchunks % d != 0 || a.nchunks / d < min_chunks) continue;
        best = d;
        if (wgs * d >= target) break;
    }
    return best;
}

// The float8 cross-term form (F8) serves a launch when the caller supplied the second weight image (wpk_f8) and the grid is large enough
// for 8-slice bricks.  (A fused skip phase keeps its own staging, weight image and three-pass arithmetic; the hand-off output format is
// the epilogue's business: both are independent of the main loop's operand format.)
bool dm3d_conv_h3v2_f8(const ConvArgs& a) {
    static const int min_wgs = [] { const char* e = getenv("DM3D_CONV_WIDE_WGS"); return e ? atoi(e) : 512; }();
    if (!a.wpk_f8) return false;
    if ((a.out_h2 || a.post_scale) && a.od % 8 != 0) return false;     // the fused output forms live in the full-brick epilogue: whole 8-slice bricks
    const long wgs = (long)a.batch * ((a.od + 7) / 8) * ((a.oh + 7) / 8) * ((a.ow + 7) / 8) * (a.coutpad / 64) * (a.parity ? 8 : 1);
    return wgs >= min_wgs;
}

// only the float8 cross-term launches come here (dm3d_conv.hip: dm3d_conv_h3v3_serves() is false for them)
int dm3d_conv_launch_h3v2(ConvArgs& a, int which, hipStream_t st) {
    DM3D_REQUIRE(dm3d_conv_h3v2_f8(a), "conv: dm3d_conv_launch_h3v2 serves the float8 cross-term form only");
    if (which == DM3D_CONV_UP) return a.pscale ? launch_f8<2, 1>(a, st) : launch_f8<2, 0>(a, st);
    if (a.x_h2) return launch_f8<3, 2>(a, st);
    return a.pscale ? launch_f8<3, 1>(a, st) : launch_f8<3, 0>(a, st);
}

// f8: the image of the float8 cross-term form (taps padded to a multiple of 8, records in the H3F8 layout)
int64_t dm3d_h3v2_image_bytes(int taps, int cin, int cout, int f8) {
    const int g = 4, tapsp = (taps + g - 1) / g * g;
    return (int64_t)tapsp * dm3d_round_up(cout, 64) * (dm3d_round_up(cin, 16) / 16) * REC * (int64_t)sizeof(_Float16);
}

int dm3d_pack_h3v2(const float* keras_kernel, int taps, int cin, int cout, int w_exp, const float* in_scale, void* packed, int mode,
                   int f8, hipStream_t st) {
    const int nchunks = (int)(dm3d_round_up(cin, 16) / 16), ntiles = (int)(dm3d_round_up(cout, 64) / 64);
    const int g = 4;
    hipLaunchKernelGGL(pack_weights_h3v2_kernel, dim3(4096), dim3(256), 0, st, keras_kernel, taps, (taps + g - 1) / g * g, cin, cout,
                       nchunks, ntiles, ldexpf(1.0f, w_exp), in_scale, static_cast<_Float16*>(packed), mode, f8);
    return dm3d_launch_check("pack_weights_h3v2_kernel");
}
