// dm3d_conv_h3w.hip — the k3 / stride-1 split-float16 Conv3d with a Winograd F(2,3) transform along x: 36 instead of 54 MFMA k-steps per
// pair of output voxels (9 (dz, dy) taps x 4 transform terms against 27 taps x 2 voxels).
//
// reference op: Conv3D(width, 3, padding="same") behind BatchNormalization + swish, + time-embedding / bias / residual adds
// (networks/conditional_dm3d.py:254-268) — the same launches the free-running kernel of dm3d_conv_h3v3.hip serves; this form takes them when
// the caller supplied the transformed weight image (dm3d_conv_desc.wpk_wino) and the grid is large (dm3d_conv_h3w_serves below).
//
// Why.  Round 3 measured the three-pass split-float16 conv at the chip's power limit, not at a scheduling limit: the same instruction stream
// runs 30 % faster on all-zero operands (2.39 against 1.85 GHz), no reordering of it moved the wall time, and no other
// MFMA operand type is cheaper per product at the accuracy H3 keeps.  What is left is to execute fewer MFMAs.  For outputs y(2i), y(2i+1)
// of a row and inputs d0..d3 = x(2i-1 .. 2i+2), with the three x taps g0, g1, g2 of one (dz, dy, cin, cout):
//     v0 = d0 - d2, v1 = d1 + d2, v2 = d2 - d1, v3 = d1 - d3          (input transform, float32, before the hi / lo split)
//     u0 = g0, u1 = (g0 + g1 + g2) / 2, u2 = (g0 - g1 + g2) / 2, u3 = g2    (weight transform, at pack time)
//     m_t = sum over (dz, dy, cin) of u_t . v_t                         (FOUR implicit GEMMs over 9 taps — the MFMA work)
//     y(2i) = m0 + m1 + m2,  y(2i+1) = m1 - m2 - m3                     (output transform, on the accumulators)
// Every m_t is a float32-accumulated sum of split-float16 products exactly like the direct form's; the transforms add one float32 rounding
// on each side and the output transform two more — the result differs from the direct kernel's in the last bits (same 2e-5 parity bound;
// tests/test_gpu_wino.py), not in kind.  |v_t| <= 2 max|x|: the range the activations may use halves (the host passes range_limit / 2 to
// the producers of such a conv).
//
// Geometry.  One workgroup = 4 waves = one 8 x 8 x 8 brick x 64 output channels; a wave owns TWO z-slices: per transform term t four
// 16-row operand groups (slice s, row group g: rows = 4 y x 4 x-pairs) x four 16-column tiles = 16 accumulator tiles, 64 in all (256
// registers: the kernel runs one wave per SIMD with the accumulators in the AGPR half of the file).  That is the point of the shape: an
// MFMA step of 48 instructions reads 16 fragments, the same LDS bytes per MFMA as the direct kernel (a one-slice wave would read 12 per 24).
// The LDS image holds, per halo row (z, y) of the brick, 16 records [t][x-pair] of 16 channels (hi / lo, 64 bytes) at a pitch of 17
// records (100 rows = 106 KB), slot s of a row's records at physical slot s ^ (y & 2), and row group g of a fragment (MFMA rows 4g .. 4g+3 =
// four y) reads x-pair g ^ (g >> 1) = 0, 1, 3, 2.  ds_read_b128 is served in four groups of 16 lanes that are NOT contiguous — {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31}, the same + 32 (MI355X_MICROARCH.md, LDS) —, i.e. row groups 0 and 3 with channel piece 0 together with row
// groups 1 and 2 with piece 1: with that pairing and this swizzle the 16 reads of a group hit 16 different 16-byte slots of the 256-byte
// bank row.  (Rounds 3-4 had x-pair = g and s ^ (y & 3), laid out for contiguous groups of 16: every A-fragment read was a two-way
// conflict, 8 LDS cycles instead of 4 — SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.31 in profiles/r04_h3_sq.csv; tools/lds_model.py
// replays both layouts against the group table.)  The stores of one record index by the 64 rows of a wave are conflict-free in both
// (pitch 16 would put them all into one 64-byte window).
// Output channels: column j of column tile ni computes channel 64 ntile + 4 j + ni (the packers' choice — dm3d_pack_weights_h3w and the skip conv's
// operand fragments —, NOT the direct kernel's 16 ni + j): a lane's four column tiles are four consecutive channels of one voxel, and the
// epilogue (epilogue_cq, dm3d_conv_h3v2_parts.h) stores 16-byte pieces without a transpose, 256 contiguous bytes per voxel and row group.
// A step = one pair of (dz, dy) taps (the lane half picks the tap; the tenth tap is a zero pad) x one transform term: 20 steps per
// 16-channel chunk, TERM-MAJOR (t = step / 5), three passes of 16 MFMAs each (al.bh, ah.bh, ah.bl) on registers; weights by LDS-DMA
// through a ring of four 8 KB buffers, one barrier per step.  Term-major order is what keeps the staging out of the register file: the
// image's records of term t are dead five steps into the chunk's t-th block, so the NEXT chunk's records of t = 0, 1, 2 are stored as soon
// as they are computed and only its t = 3 quarter (32 registers) waits for the last step's barrier.
//
// One wave per SIMD: what that costs and how the loop is written (measurements: DESIGN.md section 4, tools/micro/mfma_valu_overlap.hip).
// A wave issues in order; an MFMA 16x16x32 holds the vector issue for 8 of its 16 cycles, so 8 cycles of other instructions ride in its
// shadow (two plain VALU, or one transcendental; a ds_read_b128 ~2) — IF they sit between two MFMAs; a burst between passes, a third VALU
// in a gap, an s_nop, a packed-float32 VALU all cost their full issue time, and there is no second wave to absorb them.  So a pass is 16
// gaps of one MFMA plus at most a few instructions: the fragment reads go, one or two at a time, into the registers the last group of four
// MFMAs released; the weight DMA, the halo requests and the staging arithmetic (norm + SiLU, transform, split — plain scalar float32, one
// transcendental or two plain instructions per gap) are spread over the gaps by fixed tables (slot_gap below); finished records are stored
// at pass heads, where the counted LDS waits can account for them.  The MFMAs are inline asm with the accumulator constrained to AGPRs (left
// alone, hipcc's allocator shuffled accumulator tiles between AGPRs, VGPRs and scratch, and a scratch reload in this loop costs a vmcnt(0)).
// First forms of this loop (staging in per-pass bursts; packed VALU; spills) ran at 0.54 MFMA duty and lost to the direct kernel; this one
// holds 0.65-0.68 (the same duty on all-zero operands: in cycles it is bound by single-wave issue) at the 2.0-2.15 GHz the chip's power
// limit leaves on real data (2.38 on zeros; profiles/r03_clocks_under_load.log), and takes 10-21 % less time than the direct kernel from
// 32 input channels up (profiles/r03_wino_ab.log).  The prologue and the epilogue (the shared one, once per slice) run unoverlapped — one
// workgroup per CU — and cost 10 % of a 12-chunk workgroup, a quarter of a 4-chunk one: one-chunk launches stay on the direct kernel.
// A tenth of the MFMAs multiply the zero pad tap (5-9.5 % of the time: profiles/r03_wino_pad_cost.log; DESIGN.md section 8 on removing it).
#include <cstdlib>
#include "dm3d_conv_h3v2_parts.h"

using namespace h3v2;

// Diagnostic build only (-DDM3D_CLOCK_STAMPS, tools/mk_stamp_variants.py -> variants/cck.so; the product library carries none of it): thread 0 of
// a workgroup writes s_memtime / s_memrealtime at kernel entry (0: its first item only — a stamp at the item advance costs the step loop a
// spilled register), around every item's chunk loop (1, 28), inside its epilogue (20, 21) and at its end (29) into a buffer of its own.
#ifdef DM3D_CLOCK_STAMPS
__device__ unsigned long long* g_dbg_stamps_w = nullptr;
extern "C" int dm3d_debug_set_stamps_wino(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_stamps_w), &p, sizeof(p)); }
#ifndef STAMP_MASK
#define STAMP_MASK 0xffffffffu
#endif
#define STAMP(i) do { if (((STAMP_MASK >> (i)) & 1u) && g_dbg_stamps_w && wave == 0 && __builtin_amdgcn_mbcnt_lo(~0u, 0u) == 0u && cur.ntile == 0 && cur.khalf == 0 && item / ny < 4096u) { \
    g_dbg_stamps_w[(item / ny) * 32 + (i)] = __builtin_amdgcn_s_memtime(); \
    if ((i) == 1) g_dbg_stamps_w[(item / ny) * 32 + 30] = __builtin_amdgcn_s_memrealtime(); \
    if ((i) == 28) g_dbg_stamps_w[(item / ny) * 32 + 31] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

namespace {

template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// MODE 0: float32 input as is; 1: float32 input through the fused norm + SiLU prologue; 2: x1 already activated and split (DM3D_FMT_H2)
// The workgroup is PERSISTENT: it walks a list of work items, and the next item's first image and weight steps are staged during the last
// chunk of the current one.  A fused 1x1 skip conv runs between an item's chunk loop and its epilogue WITHOUT touching LDS (skip_tail below).
template <int MODE>
__global__ __launch_bounds__(256, 1) void conv3d_igemm_h3w(const ConvArgs p) {
    constexpr int TD = 8, TH = 8, TW = 8, CK = 16, NT = 64, NW = 4;
    constexpr int HD = TD + 2, HH = TH + 2, HROWS = HD * HH;
    constexpr int RREC = 17;                                        // records per halo row: 16 + 1 (see the store side below)
    constexpr int NS = 20;                                          // steps per chunk: 5 tap pairs x 4 transform terms
    constexpr int WPAIR = 2 * NT * REC;                             // halfs per step's weights (8 KB)
    constexpr int RING = 4;
    constexpr int WSLOT = WPAIR * 2 / 1024 / NW;                    // 1 KB DMA pieces per wave and step: 2
    constexpr int NLD = 5;                                          // steps whose pass B requests two halo voxels each
    constexpr bool pro = MODE == 1, xh2 = MODE == 2;
    constexpr int PLOADS = pro ? 4 : 0;

    extern __shared__ __attribute__((aligned(16))) _Float16 smem_w[];
    _Float16* lds_w = smem_w;                   // [RING][2 taps][NT][REC]
    _Float16* lds_in = smem_w + RING * WPAIR;   // [HROWS][4 t][4 x-pairs (+ 1 pad)][REC]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (uniform: DMA destinations, M0, stay in scalar registers)

    // ---- work items.  An item = (brick, column tile, Cin part); item w = brick * ny + by, by = ntile + ntiles * khalf (column tile fastest:
    // the column tiles of one brick share all of its halo voxels).  XCD-aware as in dm3d_conv_h3v3.hip: dispatch ids d and d + 8 share an
    // XCD, XCD k takes the k-th contiguous eighth of the item list, and inside the XCD the G / 8 workgroups take its items round-robin, so
    // that the workgroups of one L2 work on neighbouring bricks at about the same time.  ksplit == 2 (grids that would leave half the CUs
    // idle: the 8^3 level at B = 32): an item contracts chunks [c_lo, c_hi) only; the two parts of a tile meet behind the chunk loop (the
    // hand-over form, below).
    const int ntiles = p.coutpad / NT;
    const unsigned ny = (unsigned)(ntiles * p.ksplit);
    const int bpv = p.bd * p.bh * p.bw;
    const int cpp = p.nchunks / p.ksplit;                                // chunks per Cin part
    unsigned item, item_end, item_step;
    {
        const unsigned n_items = (unsigned)(p.batch * bpv) * ny, G = gridDim.x, d = blockIdx.x;
        if ((G & 7u) == 0 && (n_items & 7u) == 0) {
            const unsigned per = n_items >> 3;
            item = (d & 7u) * per + (d >> 3); item_end = (d & 7u) * per + per; item_step = G >> 3;
        } else { item = d; item_end = n_items; item_step = G; }
    }
    struct Item { int b, oz0, oy0, ox0, ntile, khalf; };
    auto decode = [&](unsigned w) {
        Item it;
        unsigned brick = w / ny;
        const unsigned by = w - brick * ny;
        it.ntile = (int)(by % (unsigned)ntiles); it.khalf = (int)(by / (unsigned)ntiles);
        it.b = (int)(brick / (unsigned)bpv);
        brick -= (unsigned)(it.b * bpv);
        it.oz0 = (int)(brick / (unsigned)(p.bh * p.bw)) * TD;
        it.oy0 = (int)((brick / (unsigned)p.bw) % (unsigned)p.bh) * TH;
        it.ox0 = (int)(brick % (unsigned)p.bw) * TW;
        // (uniform, but the divisions run on the vector unit: without this hipcc keeps both items and everything derived from them —
        // the weight pointers — in vector registers, twenty of them across the step loop)
        it.b = __builtin_amdgcn_readfirstlane(it.b); it.oz0 = __builtin_amdgcn_readfirstlane(it.oz0); it.oy0 = __builtin_amdgcn_readfirstlane(it.oy0);
        it.ox0 = __builtin_amdgcn_readfirstlane(it.ox0); it.ntile = __builtin_amdgcn_readfirstlane(it.ntile); it.khalf = __builtin_amdgcn_readfirstlane(it.khalf);
        return it;
    };
    Item cur = decode(item);
    // the item's row of the epilogue's vector operand (dm3d_conv_desc.vec_idx[b]): a scalar load issued here, a whole chunk loop before its
    // use — read inside the epilogue it headed a chain of two dependent misses (index, then the row) in front of the first tile
    auto vec_row = [&](const Item& it) { return p.vec ? (p.vec_idx ? p.vec_idx[it.b] : it.b) : 0; };
    int vrow_s = vec_row(cur);
    int c_lo = cur.khalf * cpp, c_hi = c_lo + cpp;
    STAMP(0);

    // ---- staging: thread t owns 16-byte piece t & 1 (8 channels) of halo row t >> 1 = (hz, hy): ten voxels in, sixteen records out.
    // (neighbouring lanes take the two pieces of one voxel: a wave's load instruction touches 32 cache lines, not 64)
    // The lane constants of the step loop — staging piece and LDS store address, operand bases, DMA lane offset — are RE-MADE at the top of
    // every item from an opaque copy of the thread number (lane_setup): defined once in front of the item loop they would be live across
    // the skip tail and the epilogue as well, where every register is taken, and hipcc then spills the whole live range — reloading such
    // a value at each of its uses INSIDE the step loop, behind a vmcnt(0) that drains the weight DMA.
    int piece = 0;
    bool s_act = false;
    unsigned st_addr = 0u, a_pair[2] = {0u, 0u}, wb_hi = 0u;
    int w_voff = 0;
    constexpr int DZB = HH * RREC * REC * 2, DYB = RREC * REC * 2;        // bytes to the halo row one z / one y further
    const unsigned in_addr = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)lds_in;
    const unsigned w_addr = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)lds_w;
    auto lane_setup = [&]() {
        int t = tid;
        asm volatile("" : "+v"(t));
        const int ln = t & 63, half = ln >> 5, q = (ln >> 4) & 1, row = ln & 15;
        // staging: halo row t >> 1, piece t & 1; record k = term * 4 + x-pair of the row sits k * REC halfs further; lo piece: ^ 16 halfs
        piece = t & 1;
        s_act = (t >> 1) < HROWS;
        const int srow = s_act ? (t >> 1) : HROWS - 1;
        st_addr = in_addr + (unsigned)(srow * RREC * REC + ((piece ^ ((srow % HH) & 2)) << 3)) * 2u;
        // operand addressing: lane (half, q, row): row = 4 * xpair + y inside a group, half = tap of the pair, q = 8-channel piece.
        // A step's tap pair = two (dz, dy) taps, the lane half picks one: pairs 0-2 = (dz, 0) | (dz, 1) for dz = 0, 1, 2; pair 3 = (0, 2) | the
        // pad (the voxels of (1, 2) against zero weights); pair 4 = (1, 2) | (2, 2).  So TWO lane-dependent bases serve all five (the swizzled
        // slot depends on dy only) and the pair is an immediate: a_pair[0] + dz * DZB, a_pair[1] + {0, DZB}; hi piece (lo: ^ 32)
        const int ay = row & 3, xq = (row >> 2) ^ (row >> 3);           // row group g of the fragment = x-pair g ^ (g >> 1) (0, 1, 3, 2: see the file comment)
        const unsigned a_base = in_addr + (unsigned)((((2 * wave) * HH + ay) * RREC + xq) * (REC * 2));
        unsigned sl[3];                                                 // physical slot (bytes) of piece q in a row with y = ay + dy
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) sl[dy] = (unsigned)((q ^ ((ay + dy) & 2)) << 4);
        a_pair[0] = a_base + (half ? DYB + sl[1] : sl[0]);
        a_pair[1] = a_base + 2 * DYB + sl[2] + (half ? DZB : 0);
        const int b_pos = pi_pos(row);
        wb_hi = w_addr + (unsigned)((half * NT + b_pos) * REC + ((q ^ swz(b_pos)) << 3)) * 2u;       // hi piece (lo: ^ 32)
        w_voff = wave * 1024 + ln * 16;
    };
    lane_setup();
    // the staging state always describes the image being staged — the next chunk's, which in a persistent workgroup's last chunk is the
    // NEXT item's first image: sample, whether this thread's halo row lies inside the volume, whether halo columns 0 / 9 do, and the voxel
    // index of halo column 0 (rows outside the volume: a clamped row, masked later)
    int sb, gv0;
    bool row_in, x_lo, x_hi;
    auto stage_setup = [&](const Item& it) {
        // (the halo row from an opaque copy of the thread number, once per item: two registers that do not live across the step loop)
        int tid_s = tid;
        asm volatile("" : "+v"(tid_s));
        const int sr = (tid_s >> 1) < HROWS ? (tid_s >> 1) : HROWS - 1, hz = sr / HH, hy = sr % HH;
        const int iz = it.oz0 - 1 + hz, iy = it.oy0 - 1 + hy;
        row_in = iz >= 0 && iz < p.ind && iy >= 0 && iy < p.inh;
        x_lo = it.ox0 > 0; x_hi = it.ox0 + TW < p.inw;
        const int izc = iz < 0 ? 0 : (iz >= p.ind ? p.ind - 1 : iz), iyc = iy < 0 ? 0 : (iy >= p.inh ? p.inh - 1 : iy);
        gv0 = ((it.b * p.ind + izc) * p.inh + iyc) * p.inw + it.ox0 - 1;
        sb = it.b;
    };
    stage_setup(cur);
    f32x4v acc[4][4][4];                                                // [t][2 * slice + row group][column tile]; zeroed in the prologue

    // weights: the packed image IS the LDS image, a step of a column tile is a linear 8 KB copy; wave w moves the 1 KB pieces w, w + 4.
    // Buffer form of the LDS-DMA (buffer_load_dwordx4 ... lds): the image's descriptor in scalar registers, ONE constant lane offset, the
    // step as a scalar byte offset — no per-lane 64-bit pointer to keep (two registers the step loop does not have) or to add to.
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.wpk), (short)0, (int)((long)ntiles * p.nchunks * (NS * WPAIR * 2)), 0x00020000);
    auto w_chunk = [&](int ntile_, int ch_) { return (unsigned)((ntile_ * p.nchunks + ch_) * (NS * WPAIR * 2)); };      // byte offset of a chunk's 20 steps
    unsigned w_cur = w_chunk(cur.ntile, c_lo), w_nxt = w_cur;            // (uniform) the chunk being multiplied / the chunk behind it
    auto fetch_w1 = [&](const unsigned off, int slot, const int i) {
        char* dst = reinterpret_cast<char*>(lds_w) + slot * (WPAIR * 2) + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(dst + i * (NW * 1024)), 16, w_voff,
                                                 (int)(off + i * (NW * 1024)), 0, 0);
    };
#pragma unroll
    for (int i = 0; i < RING; ++i) { fetch_w1(w_cur + i * (WPAIR * 2), i, 0); fetch_w1(w_cur + i * (WPAIR * 2), i, 1); }      // the ring starts full; pass B of step s refills s's buffer with step s + 4  (NS % RING == 0: slot = step & 3)

    f32x4 va[10][2];                                     // the row's ten voxels: raw -> activated float32, in place
    u32x4 oh[4], ol[4];                                  // one transform term's four records (x-pairs), hi and lo pieces
    u32x4 o3h[4], o3l[4];                                // the t = 3 records, kept until the chunk's last barrier
    f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
    bool km[3][2] = {{true, true}, {true, true}, {true, true}};     // keep masks [halo column 0 / 1-8 / 9][channels 0-3 / 4-7 of the piece]: inside the volume and a real channel
    bool ok0 = true, ok1 = true;
    // Loads are never behind a branch (clamped addresses, masked afterwards): a load under a divergent `if` makes hipcc wait vmcnt(0) on the spot.
    auto load_voxel = [&](int ch, int hx) {
        const int hxc = hx == 0 ? (x_lo ? 0 : 1) : (hx == 9 ? (x_hi ? 9 : 8) : hx);
        const unsigned gv = (unsigned)(gv0 + hxc);
        // (addresses = a uniform base + a 32-bit lane offset: the launcher admits tensors below 4 GB only — one register per voxel, nothing
        // for hipcc to hoist out of the chunk loop as 64-bit pairs)
        if (xh2) {                       // record ch of the voxel: hi piece at slot `piece`, lo piece at slot 2 + piece
            const char* base = reinterpret_cast<const char*>(p.x1) + (size_t)ch * 64;
            const unsigned off = gv * ((unsigned)p.c1 * 4u) + (unsigned)piece * 16u;
            va[hx][0] = *reinterpret_cast<const f32x4*>(base + off);
            va[hx][1] = *reinterpret_cast<const f32x4*>(base + off + 32u);
            return;
        }
        const int c0 = ch * CK;
        const float* src;
        int ldc, cb;
        if (c0 < p.c1) { src = p.x1; ldc = p.c1; cb = c0; } else { src = p.x2; ldc = p.c2; cb = c0 - p.c1; }
        const int cpos = cb + piece * 8;
        const int off0 = cpos < ldc ? cpos : 0, off1 = cpos + 4 < ldc ? cpos + 4 : 0;
        const char* base = reinterpret_cast<const char*>(src);
        const unsigned voff = gv * ((unsigned)ldc * 4u);
        va[hx][0] = *reinterpret_cast<const f32x4*>(base + (voff + (unsigned)off0 * 4u));
        va[hx][1] = *reinterpret_cast<const f32x4*>(base + (voff + (unsigned)off1 * 4u));
    };
    auto load_chunk_params = [&](int ch) {
        if (xh2) return;
        const int c0 = ch * CK;
        const int ldc = c0 < p.c1 ? p.c1 : p.c2, cpos = (c0 < p.c1 ? c0 : c0 - p.c1) + piece * 8;
        ok0 = cpos < ldc;
        ok1 = cpos + 4 < ldc;
        if (pro) {
            const int s0 = ok0 ? c0 + piece * 8 : 0, s1 = ok1 ? c0 + piece * 8 + 4 : 0;
            const size_t bo = (size_t)sb * p.pro_bstride;
            sc0 = *reinterpret_cast<const f32x4*>(p.pscale + bo + s0);
            sh0 = *reinterpret_cast<const f32x4*>(p.pshift + bo + s0);
            sc1 = *reinterpret_cast<const f32x4*>(p.pscale + bo + s1);
            sh1 = *reinterpret_cast<const f32x4*>(p.pshift + bo + s1);
        }
    };
    // vector-memory requests issued in pass B of step s beside its weight DMA (a negative s: a late step of the previous chunk)
    auto halo_ops = [&](int s) {
        if (s < 0) s += NS;
        if (s >= NLD) return 0;
        return 4 + (s == 0 ? PLOADS : 0);
    };
    // ---- the staging arithmetic.  Two forms of the same operations in the same order: whole quads / pairs at once for the prologue's
    // first image (act_quad, split_pair: nothing to hide behind yet), and one or two scalar instructions per MFMA gap inside the step loop
    // (slot_gap below — packed float32 VALU costs far more than its count beside MFMAs: tools/micro/mfma_valu_overlap.hip).
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    h8 g_hi, g_lo;
#define DM3D_PIN(x) asm volatile("" :: "v"(x))        /* an ordered use: what feeds it cannot sink below this point (instruction selection
                                                          otherwise sinks such arithmetic to its first use, ten steps later) */
    // channels 4h .. 4h + 3 of halo voxel hx: norm + SiLU (the operations of dm3d_silu in its order), zero outside the volume / past the
    // last channel (the prologue's form: the whole quad at once)
    auto act_quad = [&](const int hx, const int h) {
        const int col = hx == 0 ? 0 : (hx == 9 ? 2 : 1);
        f32x4 v;
        if (xh2) {                       // the split pair back to one float32 per channel
            if (h == 0) { g_hi = __builtin_bit_cast(h8, va[hx][0]); g_lo = __builtin_bit_cast(h8, va[hx][1]); }    // (saved: the float32 values overwrite the halves)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const _Float16 a = g_hi[4 * h + e], b = g_lo[4 * h + e]; v[e] = (float)a + (float)b; }
            if (!km[col][0]) v = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            v = va[hx][h];
            if (pro) {
                const f32x4 y = __builtin_elementwise_fma(v, h ? sc1 : sc0, h ? sh1 : sh0);
                const f32x4 z = y * -1.4426950408889634f;
                f32x4 d;
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = __builtin_amdgcn_exp2f(z[e]);
                d = d + 1.0f;
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = __builtin_amdgcn_rcpf(d[e]);
                v = y * d;
            }
            if (!km[col][h]) v = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        DM3D_PIN(v);
        va[hx][h] = v;
    };
    // channels 2j, 2j + 1 of transform term t of x-pair xt, and their float16 split (no clamp: |term| <= 2 max|x|, and the range guard of
    // this form keeps |x| <= 32752 — include/dm3d.h)
    auto split_pair = [&](const int t, const int xt, const int j) {
        const int h = j >> 1, c = 2 * (j & 1);
        auto d = [&](const int k) { return f32x2{va[2 * xt + k][h][c], va[2 * xt + k][h][c + 1]}; };
        const f32x2 x = t == 0 ? d(0) - d(2) : (t == 1 ? d(1) + d(2) : (t == 2 ? d(2) - d(1) : d(1) - d(3)));
        const float x0 = x[0], x1 = x[1];
        unsigned int a, r;
        float r0, r1;
        // hi = f16(x) (RNE), lo = f16(x - hi): one asm statement (hipcc puts an s_nop between two asm statements that depend on each other)
        asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n\tv_fma_mix_f32 %2, %0, -1.0, %4 op_sel_hi:[1,0,0]\n\t"
                     "v_fma_mix_f32 %3, %0, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_cvt_pk_f16_f32 %1, %2, %3"
                     : "=&v"(a), "=&v"(r), "=&v"(r0), "=&v"(r1) : "v"(x0), "v"(x1));
        if (t == 3) { o3h[xt][j] = a; o3l[xt][j] = r; } else { oh[xt][j] = a; ol[xt][j] = r; }
    };
    auto chunk_masks = [&]() {
        km[0][0] = row_in && x_lo && ok0; km[0][1] = row_in && x_lo && ok1; km[1][0] = row_in && ok0; km[1][1] = row_in && ok1;
        km[2][0] = row_in && x_hi && ok0; km[2][1] = row_in && x_hi && ok1;
    };
    // (inline asm with the lo address made on the spot: as C++ stores hipcc keeps both addresses in registers for the whole loop — and, the
    // register file being full, spills them and reloads them at every store site behind a vmcnt(0))
    auto store_record = [&](const int t, const int xt) {
        if (s_act) {
            unsigned lo_addr;
            const u32x4 dh = t == 3 ? o3h[xt] : oh[xt], dl = t == 3 ? o3l[xt] : ol[xt];
            switch (t * 4 + xt) {
#define DM3D_ST(k_) case k_: asm volatile("ds_write_b128 %1, %2 offset:%4\n\tv_xor_b32 %0, 32, %1\n\tds_write_b128 %0, %3 offset:%4" \
                                          : "=&v"(lo_addr) : "v"(st_addr), "v"(dh), "v"(dl), "n"((k_) * REC * 2) : "memory"); break;
                DM3D_ST(0) DM3D_ST(1) DM3D_ST(2) DM3D_ST(3) DM3D_ST(4) DM3D_ST(5) DM3D_ST(6) DM3D_ST(7)
                DM3D_ST(8) DM3D_ST(9) DM3D_ST(10) DM3D_ST(11) DM3D_ST(12) DM3D_ST(13) DM3D_ST(14) DM3D_ST(15)
#undef DM3D_ST
            }
        }
    };
    // raw barrier behind this wave's own LDS traffic (never drains the vector-memory counter: the DMA fills stay in flight across it)
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: the first chunk's image
#pragma unroll
    for (int hx = 0; hx < 10; ++hx) load_voxel(c_lo, hx);
    load_chunk_params(c_lo);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int pi = 0; pi < 4; ++pi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                acc[t][pi][ni] = f32x4v{0.f, 0.f, 0.f, 0.f};
                asm volatile("" : "+a"(acc[t][pi][ni]));      // zeroed HERE, in the shadow of the first image's loads (hipcc sinks the 256 writes to the first MFMA otherwise)
            }
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0) (with an instruction hipcc's wait-count pass sees): halo and the first four weight steps
    chunk_masks();
#pragma unroll
    for (int hx = 0; hx < 10; ++hx) { act_quad(hx, 0); act_quad(hx, 1); }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int u = 0; u < 16; ++u) split_pair(t, u >> 2, u & 3);
#pragma unroll
        for (int xt = 0; xt < 4; ++xt) store_record(t, xt);
    }
    lds_barrier();

    // ---- the step loop.  A wave alone on its SIMD has only its own MFMAs to hide its other instructions behind: an MFMA holds the vector
    // issue for 8 of its 16 cycles, so 8 cycles of other instructions ride in its shadow — if they are THERE, between two MFMAs, and
    // independent of the neighbours (in-kernel stamps of the first forms of this loop, which issued reads, DMA and staging arithmetic in
    // bursts between the passes: 28 000 cycles per chunk for 15 360 of MFMA).  So a pass is 16 GAPS, each one MFMA plus at most a few
    // instructions:
    //   * fragment reads, one or two per group of four MFMAs, into registers the group just released.  Pass A (al.bh, row-group major)
    //     re-requests bl for this step's pass C behind groups 0-1 and al of the NEXT step behind groups 1-3; pass B (ah.bh, column-tile
    //     major) requests bh of the next step behind each group; pass C (ah.bl, row-group major) ah of the next step.  LDS returns in order:
    //     the head of pass A waits lgkmcnt(4 + stores) (the four ah reads of the previous pass C may be out), the head of pass B — the step's
    //     one barrier — lgkmcnt(4) (al of the next step) and vmcnt(N) as before, pass C nothing.  Addresses cost nothing: five per-lane
    //     bases (one per tap pair, the lane half picks the tap) + immediates for term / row group / slice / ring slot.
    //   * the weight DMA right behind the barrier, the halo requests behind groups 0-1 of pass B, the chunk parameters behind group 2.
    //   * the staging arithmetic, one or two instructions per gap (the tables at slot_gap below); finished records are stored at the head
    //     of the next pass (the waits above count them).
    // The fragment reads are inline asm: behind a `global_load_lds` hipcc guards every visible ds_read result with lgkmcnt(0) (see
    // dm3d_conv_h3v3.hip).  The MFMAs are inline asm with the accumulator constrained to the AGPR file: left to itself hipcc's allocator
    // moved accumulator tiles between AGPRs, VGPRs and scratch around the term blocks, and every scratch reload inside the loop is
    // followed by vmcnt(0), which drains the weight DMA and halo requests in flight.
    h8 ah[4], al[4], bh[4], bl[4];
    // the lo-piece addresses are made where they are used (one v_xor_b32 per step and operand), not kept: registers
    auto lo_of = [](const unsigned a) { unsigned r; asm volatile("v_xor_b32 %0, 32, %1" : "=v"(r) : "v"(a)); return r; };
#define DM3D_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    constexpr int GOFF = 4 * RREC * REC * 2, SOFF = HH * RREC * REC * 2;    // bytes to the second row group / the second slice
    auto a_hi = [&](const int s) { return a_pair[(s % 5) >= 3]; };
    // fragment i (0 .. 3: slice i >> 1, row group i & 1) of step s (term s / 5: 256 bytes per term inside a halo row) from a0 = a_hi(s) or its lo twin
    auto read_a = [&](h8 (&dst)[4], const int s, const int i, const unsigned a0) {
        const int tp = s % 5;
        switch (((s / 5) * 4 + i) * 3 + (tp < 3 ? tp : tp - 3)) {      // (term, fragment, z step of the pair): the immediate offset
#define DM3D_RA1(t_, i_, z_) case ((t_) * 4 + (i_)) * 3 + (z_): DM3D_DSR(dst[i_], a0, (z_) * DZB + (t_) * 4 * REC * 2 + ((i_) >> 1) * SOFF + ((i_) & 1) * GOFF); break;
#define DM3D_RA(t_, i_) DM3D_RA1(t_, i_, 0) DM3D_RA1(t_, i_, 1) DM3D_RA1(t_, i_, 2)
            DM3D_RA(0, 0) DM3D_RA(0, 1) DM3D_RA(0, 2) DM3D_RA(0, 3) DM3D_RA(1, 0) DM3D_RA(1, 1) DM3D_RA(1, 2) DM3D_RA(1, 3)
            DM3D_RA(2, 0) DM3D_RA(2, 1) DM3D_RA(2, 2) DM3D_RA(2, 3) DM3D_RA(3, 0) DM3D_RA(3, 1) DM3D_RA(3, 2) DM3D_RA(3, 3)
#undef DM3D_RA
#undef DM3D_RA1
        }
    };
    // column tile i of the weights in ring slot `slot` from wa = wb_hi or its lo twin
    auto read_b = [&](h8 (&dst)[4], const int slot, const int i, const unsigned wa) {
        switch (slot * 4 + i) {
#define DM3D_RB(k_, i_) case (k_) * 4 + (i_): DM3D_DSR(dst[i_], wa, (k_) * WPAIR * 2 + (i_) * 16 * REC * 2); break;
            DM3D_RB(0, 0) DM3D_RB(0, 1) DM3D_RB(0, 2) DM3D_RB(0, 3) DM3D_RB(1, 0) DM3D_RB(1, 1) DM3D_RB(1, 2) DM3D_RB(1, 3)
            DM3D_RB(2, 0) DM3D_RB(2, 1) DM3D_RB(2, 2) DM3D_RB(2, 3) DM3D_RB(3, 0) DM3D_RB(3, 1) DM3D_RB(3, 2) DM3D_RB(3, 3)
#undef DM3D_RB
        }
    };
    // ---- staging work by gap G = 16 * (3 * step + pass) + g (0 .. 959), at most 8 cycles of vector issue per gap (MI355X_MICROARCH.md,
    // 'vector-instruction ISSUE cost': an MFMA holds the vector issue for 8 of its 16 cycles, a plain VALU for 4, a transcendental for 8,
    // packed float32 VALU far more beside MFMAs — so plain scalar arithmetic, two plain instructions or one transcendental per gap):
    //   207            the next chunk's keep masks
    //   208 .. 567     norm + SiLU, 18 gaps per 4 channels (channels 0-3 / 4-7 of halo voxel hx: quad i = 2 hx + h at 208 + 18 i; voxel pair
    //                  i >> 2 was requested in pass B of step i >> 2, the barrier heads since have waited past it): y = x * scale + shift
    //                  (2 gaps), z = -y log2 e (2), 2^z (4), + 1 (2), 1 / . (4), y * . (2), the mask (2)
    //   576 .. 623     term 0: unit u = x-pair u >> 2, channel pair u & 3 at 576 + 3 u: the term, hi = f16(x) and x0 - hi, x1 - hi and lo.
    //   624 .. 671     term 1                The steps are term-major, so the image's records of term t are dead once every wave is past
    //   688 .. 735     term 2 (pass 43 on)   the barrier of step 5t + 4: a finished record goes to LDS at the head of the next pass (one or
    //   880 .. 927     term 3 (passes 55-57) two transient records in registers, not the whole image); only the four t = 3 records wait in
    //                                        registers for the barrier of the last step (pass 58).
    f32x4 g_y, g_z;
    float g_x0 = 0.f, g_x1 = 0.f, g_r0 = 0.f;
    unsigned int g_a = 0u;
    // ds_write_b128 issued at the head of pass p (the waits count them): the records finished in pass p - 1
    auto stores_at_head = [&](int p) {
        p = (p + 60) % 60;
        if (p == 37 || p == 38 || p == 40 || p == 41 || p == 44 || p == 45) return 2;
        if (p == 39 || p == 42 || p == 46) return 4;
        return p == 58 ? 8 : 0;
    };
    auto head_stores = [&](const int p) {
        if (p == 37) store_record(0, 0); else if (p == 38) store_record(0, 1); else if (p == 39) { store_record(0, 2); store_record(0, 3); }
        else if (p == 40) store_record(1, 0); else if (p == 41) store_record(1, 1); else if (p == 42) { store_record(1, 2); store_record(1, 3); }
        else if (p == 44) store_record(2, 0); else if (p == 45) store_record(2, 1); else if (p == 46) { store_record(2, 2); store_record(2, 3); }
        else if (p == 58) {
#pragma unroll
            for (int xt = 0; xt < 4; ++xt) store_record(3, xt);
        }
    };
    auto slot_gap = [&](const int G) {
        if (G == 207) { chunk_masks(); return; }
        if (G >= 208 && G < 568) {
            const int i = (G - 208) / 18, r = (G - 208) % 18;
            const int hx = i >> 1, h = i & 1, col = hx == 0 ? 0 : (hx == 9 ? 2 : 1);
            if (xh2) {
                // the split pair back to one float32 per channel: (float)hi + (float)lo as ONE v_fma_mix_f32 (hi * 1.0 + lo, both read as
                // float16 halves straight from the loaded registers: the same single rounding).  va[hx][0] / [1] arrive as the hi / lo
                // halves of channels 0-7 and leave as the float32 channels 0-3 / 4-7: channels 0-3 wait in g_y until 4-7, which are
                // written over the lo halves in an order that has consumed them, have read the hi halves.
                auto mix = [&](const int c) {          // channel c of the voxel
                    float d;
                    const float hv = va[hx][0][c >> 1], lv = va[hx][1][c >> 1];
                    if (c & 1) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(hv), "v"(lv));
                    else asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(hv), "v"(lv));
                    return d;
                };
                if (h == 0) {
                    if (r < 2) { g_y[2 * r] = mix(2 * r); g_y[2 * r + 1] = mix(2 * r + 1); }
                } else {
                    if (r < 2) {
                        const float v0 = mix(4 + 2 * r), v1 = mix(5 + 2 * r);
                        va[hx][1][2 * r] = v0; va[hx][1][2 * r + 1] = v1;
                    } else if (r < 4) {
                        const int c = 2 * (r - 2);
                        float v0 = km[col][0] ? g_y[c] : 0.0f, v1 = km[col][0] ? g_y[c + 1] : 0.0f;
                        DM3D_PIN(v0); DM3D_PIN(v1);
                        va[hx][0][c] = v0; va[hx][0][c + 1] = v1;
                    } else if (r < 6) {
                        const int c = 2 * (r - 4);
                        float v0 = km[col][0] ? va[hx][1][c] : 0.0f, v1 = km[col][0] ? va[hx][1][c + 1] : 0.0f;
                        DM3D_PIN(v0); DM3D_PIN(v1);
                        va[hx][1][c] = v0; va[hx][1][c + 1] = v1;
                    }
                }
                return;
            }
            if (!pro) {
                if (r < 2) {
                    const int c = 2 * r;
                    float v0 = km[col][h] ? va[hx][h][c] : 0.0f, v1 = km[col][h] ? va[hx][h][c + 1] : 0.0f;
                    DM3D_PIN(v0); DM3D_PIN(v1);
                    va[hx][h][c] = v0; va[hx][h][c + 1] = v1;
                }
                return;
            }
            const f32x4 sc = h ? sc1 : sc0, sh = h ? sh1 : sh0;
            if (r < 2) {                 // y = x * scale + shift
                const int c = 2 * r;
                g_y[c] = fmaf(va[hx][h][c], sc[c], sh[c]); g_y[c + 1] = fmaf(va[hx][h][c + 1], sc[c + 1], sh[c + 1]);
                DM3D_PIN(g_y[c]); DM3D_PIN(g_y[c + 1]);
            } else if (r < 4) {          // z = -y log2 e
                const int c = 2 * (r - 2);
                g_z[c] = g_y[c] * -1.4426950408889634f; g_z[c + 1] = g_y[c + 1] * -1.4426950408889634f;
                DM3D_PIN(g_z[c]); DM3D_PIN(g_z[c + 1]);
            } else if (r < 8) {          // 2^z
                g_z[r - 4] = __builtin_amdgcn_exp2f(g_z[r - 4]);
                DM3D_PIN(g_z[r - 4]);
            } else if (r < 10) {         // 1 + 2^z
                const int c = 2 * (r - 8);
                g_z[c] = 1.0f + g_z[c]; g_z[c + 1] = 1.0f + g_z[c + 1];
                DM3D_PIN(g_z[c]); DM3D_PIN(g_z[c + 1]);
            } else if (r < 14) {         // 1 / (1 + 2^z)
                g_z[r - 10] = __builtin_amdgcn_rcpf(g_z[r - 10]);
                DM3D_PIN(g_z[r - 10]);
            } else if (r < 16) {         // y / (1 + 2^z): the operations of dm3d_silu in its order
                const int c = 2 * (r - 14);
                g_y[c] = g_y[c] * g_z[c]; g_y[c + 1] = g_y[c + 1] * g_z[c + 1];
                DM3D_PIN(g_y[c]); DM3D_PIN(g_y[c + 1]);
            } else {                     // zero outside the volume / past the last channel
                const int c = 2 * (r - 16);
                float v0 = km[col][h] ? g_y[c] : 0.0f, v1 = km[col][h] ? g_y[c + 1] : 0.0f;
                DM3D_PIN(v0); DM3D_PIN(v1);
                va[hx][h][c] = v0; va[hx][h][c + 1] = v1;
            }
            return;
        }
        // (term 3 last, right in front of the barrier that frees its place: its four records then live three passes, not twelve, beside the
        // ten voxels they are made from — the step loop's register peak)
        if ((G >= 576 && G < 672) || (G >= 688 && G < 736) || (G >= 880 && G < 928)) {
            const int t = G < 624 ? 0 : (G < 672 ? 1 : (G < 736 ? 2 : 3));
            const int q = G - (t == 0 ? 576 : (t == 1 ? 624 : (t == 2 ? 688 : 880)));
            const int u = q / 3, st = q % 3, xt = u >> 2, j = u & 3, h = j >> 1, c = 2 * (j & 1);
            if (st == 0) {
                auto d = [&](const int k, const int e) { return va[2 * xt + k][h][c + e]; };
                g_x0 = t == 0 ? d(0, 0) - d(2, 0) : (t == 1 ? d(1, 0) + d(2, 0) : (t == 2 ? d(2, 0) - d(1, 0) : d(1, 0) - d(3, 0)));
                g_x1 = t == 0 ? d(0, 1) - d(2, 1) : (t == 1 ? d(1, 1) + d(2, 1) : (t == 2 ? d(2, 1) - d(1, 1) : d(1, 1) - d(3, 1)));
                DM3D_PIN(g_x0); DM3D_PIN(g_x1);
            } else if (st == 1) {        // hi = f16(x) (RNE), x0 - hi0
                asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_fma_mix_f32 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]" : "=&v"(g_a), "=&v"(g_r0) : "v"(g_x0), "v"(g_x1));
            } else {                     // x1 - hi1, lo = f16(x - hi)
                unsigned int r;
                float r1;
                asm volatile("v_fma_mix_f32 %1, %2, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_cvt_pk_f16_f32 %0, %4, %1" : "=&v"(r), "=&v"(r1) : "v"(g_a), "v"(g_x1), "v"(g_r0));
                if (t == 3) { o3h[xt][j] = g_a; o3l[xt][j] = r; } else { oh[xt][j] = g_a; ol[xt][j] = r; }
            }
        }
    };
#define DM3D_MFMA(T, PI, NI, A, B) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[T][PI][NI]) : "v"(A[PI]), "v"(B[NI]))
#define DM3D_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(n) : "memory")

    // ---- the item loop
    bool has_next = item + item_step < item_end;
    for (;;) {
    lane_setup();
    // (requested here, not carried over from the previous item's last step: the epilogue wants the registers)
    {
        const unsigned a_lo0 = lo_of(a_hi(0));
        read_a(al, 0, 0, a_lo0); read_a(al, 0, 1, a_lo0); read_a(al, 0, 2, a_lo0); read_a(al, 0, 3, a_lo0);
    }
    read_b(bh, 0, 0, wb_hi); read_b(bh, 0, 1, wb_hi); read_b(bh, 0, 2, wb_hi); read_b(bh, 0, 3, wb_hi);
    read_a(ah, 0, 0, a_hi(0)); read_a(ah, 0, 1, a_hi(0)); read_a(ah, 0, 2, a_hi(0)); read_a(ah, 0, 3, a_hi(0));
    STAMP(1);

    for (int ch = c_lo; ch < c_hi; ++ch) {
        // keeps the per-voxel offsets from being hoisted out of the chunk loop as 64-bit pairs
        asm volatile("" : "+v"(gv0));
        // the chunk staged during this one: the item's next chunk; in its last chunk the NEXT item's first (the staging state moves on to
        // that item's brick: nothing of this item is staged any more); behind the last item the same chunk again (unconditional like the DMAs)
        int ch_next;
        if (ch + 1 < c_hi) { ch_next = ch + 1; w_nxt = w_cur + NS * WPAIR * 2; }
        else if (has_next) {
            // (decoded here, once per item: the next item's coordinates do not live in registers across the chunk loop)
            const Item n = decode(item + item_step);
            stage_setup(n); ch_next = n.khalf * cpp; w_nxt = w_chunk(n.ntile, ch_next);
        }
        else { ch_next = ch; w_nxt = w_cur; }
        // (a generic lambda over integral constants, not `#pragma unroll`: hipcc unrolls a 20-step body of this size only in part, and a step
        // index that is not a constant turns acc[t] into scratch memory)
        static_for<NS>([&](auto S_) {
            constexpr int s = decltype(S_)::value, sn = (s + 1) % NS;
            constexpr int t = s / 5, ws = s & (RING - 1), ws1 = (s + 1) & (RING - 1);
            // ---- pass A: al(s).bh(s), row-group major.  al(s) (requested in pass A of the step before) and bh(s) (in its pass B) are back
            // once at most the four ah reads of its pass C — and the stores at that pass's head — are out.
            {
                constexpr int n_out = 4;
                switch (stores_at_head(3 * s - 1)) {
                case 0: DM3D_WAIT_LGKM(n_out); break;
                case 2: DM3D_WAIT_LGKM(n_out + 2); break;
                case 4: DM3D_WAIT_LGKM(n_out + 4); break;
                default: DM3D_WAIT_LGKM(n_out + 8); break;
                }
            }
            head_stores(3 * s);
            __builtin_amdgcn_sched_barrier(0);
            unsigned b_lo = 0u, a_lo = 0u;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                DM3D_MFMA(t, g >> 2, g & 3, al, bh);
                slot_gap(16 * (3 * s) + g);
                if (g == 2) b_lo = lo_of(wb_hi);
                if (g == 3) { read_b(bl, ws, 0, b_lo); read_b(bl, ws, 1, b_lo); }
                if (g == 6) a_lo = lo_of(a_hi(sn));
                if (g == 7) { read_b(bl, ws, 2, b_lo); read_b(bl, ws, 3, b_lo); read_a(al, sn, 0, a_lo); }
                if (g == 11) { read_a(al, sn, 1, a_lo); read_a(al, sn, 2, a_lo); }
                if (g == 15) read_a(al, sn, 3, a_lo);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- pass B: ah(s).bh(s), column-tile major.  Its head is the step's one barrier: behind this wave's lgkmcnt(4) (its reads of
            // step s's buffer — bh, bl — are back; al of the next step may be out) and vmcnt(N) (its pieces of step s + 1 have landed) it
            // frees step s's buffer for the DMA of step s + 4 and makes step s + 1 visible to everyone.  Newer than this wave's pieces of
            // step s + 1 (issued in B(s - 3)): the pieces of steps s + 2, s + 3 and the halo requests of B(s - 3) .. B(s - 1).
            {
                constexpr int DMA_BEHIND = 2 * WSLOT;
                const int extra = halo_ops(s - 3) + halo_ops(s - 2) + halo_ops(s - 1);
                // (the builtin, not inline asm: hipcc's own wait-count pass must see it.)  simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 4 << 8 | vmcnt[5:4] << 14
#define DM3D_WAIT_VM_LGKM4(n) __builtin_amdgcn_s_waitcnt((((n) & 15) | (7 << 4) | (4 << 8) | ((((n) >> 4) & 3) << 14)))
                switch (extra) {
                case 0:  DM3D_WAIT_VM_LGKM4(DMA_BEHIND); break;
                case 4:  DM3D_WAIT_VM_LGKM4(DMA_BEHIND + 4); break;
                case 8:  DM3D_WAIT_VM_LGKM4(DMA_BEHIND + 8); break;
                case 12: DM3D_WAIT_VM_LGKM4(DMA_BEHIND + 12); break;
                case 16: DM3D_WAIT_VM_LGKM4(DMA_BEHIND + 16); break;
                default: DM3D_WAIT_VM_LGKM4(0); break;
                }
#undef DM3D_WAIT_VM_LGKM4
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            const unsigned w_src = s + RING < NS ? w_cur + (s + RING) * (WPAIR * 2) : w_nxt + (s + RING - NS) * (WPAIR * 2);      // step s + 4
            fetch_w1(w_src, ws, 0);                  // into the buffer step s just left (an LDS-DMA piece costs ~60 cycles of issue: the second one two gaps on)
            head_stores(3 * s + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                DM3D_MFMA(t, g & 3, g >> 2, ah, bh);
                slot_gap(16 * (3 * s + 1) + g);
                if ((g & 3) == 3) read_b(bh, ws1, g >> 2, wb_hi);
                if (g == 1) fetch_w1(w_src, ws, 1);
                if (s < NLD && g == 5) load_voxel(ch_next, 2 * s);
                if (s < NLD && g == 9) load_voxel(ch_next, 2 * s + 1);
                if (s == 0 && g == 13) load_chunk_params(ch_next);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- pass C: ah(s).bl(s), row-group major; requests ah of the next step
            head_stores(3 * s + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                DM3D_MFMA(t, g >> 2, g & 3, ah, bl);
                slot_gap(16 * (3 * s + 2) + g);
                if ((g & 3) == 3) read_a(ah, sn, g >> 2, a_hi(sn));
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        w_cur = w_nxt;
    }
#undef DM3D_MFMA
#undef DM3D_WAIT_LGKM
#undef DM3D_PIN
#undef DM3D_DSR
    STAMP(28);
    const Brick br = {cur.b, cur.oz0, cur.oy0, cur.ox0, p.ooz, p.ooy, p.oox, cur.ntile, cur.khalf};
    {
        // (this wait must stay the first instruction behind the loop: an asm ds_read returns at once, and its destination — dead to hipcc
        // after the last step — is hipcc's to reuse before the data has landed.  The vector-memory counter is left alone: what is in
        // flight are DMA pieces of the next item's first steps, on their way into ring buffers nobody reads before the next barrier.)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        STAMP(20);
        const bool skip = p.s_npairs > 0;
        if (skip) {
            // ---- fused 1x1 conv over a second, raw input (ResidualBlock: out = conv2(...) + Conv3D(width, 1)(concat(x, skip)),
            // conditional_dm3d.py:243-248, 268).  First the output transform, IN PLACE in the accumulator file (y0 over m0, y1 over m1):
            // the skip products are ordinary ones and accumulate into the transformed tiles.
#pragma unroll
            for (int pi = 0; pi < 4; ++pi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const f32x4v m0 = acc[0][pi][ni], m1 = acc[1][pi][ni], m2 = acc[2][pi][ni], m3 = acc[3][pi][ni];
                    acc[0][pi][ni] = (m0 + m1) + m2;
                    acc[1][pi][ni] = (m1 - m2) - m3;
                    asm volatile("" : "+a"(acc[0][pi][ni]), "+a"(acc[1][pi][ni]));
                }
            // The tail, register-direct.  K = 32 per MFMA = two 16-channel chunks of the SAME voxel (the lane half picks the chunk, q the
            // 8-channel piece).  A wave's A operands are ITS OWN brick voxels — nobody else in the workgroup needs them — so they do not
            // pass through LDS: every lane loads the 8 float32 channels of its voxel for each of the wave's eight tiles (slice s, row
            // group g, column parity: rows = (y, x-pair)), splits them in registers and multiplies.  The weights come as operand fragments
            // (dm3d_pack_weights_skip_h3f: 1 KB per (16-column tile, hi | lo), lane-ordered) by plain coalesced loads.  No LDS, no barrier:
            // the next item's image and weight ring stay where they are, and launches with a skip conv are persistent like the others.
            // (The first form of this tail staged voxels and weights through LDS behind two barriers per pair: 8 000 cycles per pair for
            // 1 536 of MFMA, and one work item per workgroup.)
            // (a Cin-split launch spreads the pairs over its parts as it spreads the main loop's chunks: the skip sum is linear like them)
            const int p_lo = p.s_npairs * cur.khalf / p.ksplit, np = p.s_npairs * (cur.khalf + 1) / p.ksplit;
            // (from an opaque copy of the lane number: hipcc otherwise computes the tail's lane constants in front of the item loop and
            // keeps — spills — them across the step loop, which has no register to spare)
            int lane_t = lane;
            asm volatile("" : "+v"(lane_t));
            const int kq = lane_t >> 4, lhalf = kq >> 1, cq = (kq & 1) * 8, row_t = lane_t & 15;
            unsigned sv[8];                                     // voxel index of this lane's row in tile t = 4 s + 2 g + parity
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int z = 2 * wave + (t >> 2), y = 4 * ((t >> 1) & 1) + (row_t & 3), x = 2 * ((row_t >> 2) ^ (row_t >> 3)) + (t & 1);      // (row group -> x-pair as in the main loop)
                sv[t] = (unsigned)(((cur.b * p.ind + cur.oz0 + z) * p.inh + cur.oy0 + y) * p.inw + cur.ox0 + x);
            }
            const char* swf = reinterpret_cast<const char*>(p.swpk_f) + (size_t)cur.ntile * p.s_npairs * 8192 + lane_t * 16;
            f32x4 ar0[2][8], ar1[2][8];
            h8 bw[2][8];
            float lim0[2], lim1[2];
            auto t_load = [&](auto SET, int pp) {
                constexpr int S = decltype(SET)::value;
                const int c0 = (pp * 2 + lhalf) * CK;
                const float* src;
                int ldc, cb;
                if (c0 < p.sc1) { src = p.sx1; ldc = p.sc1; cb = c0; } else { src = p.sx2; ldc = p.sc2; cb = c0 - p.sc1; }
                const int cpos = cb + cq;
                const bool real = src != nullptr && cb < ldc && pp < np;      // a pad chunk past the last channel (or pair) multiplies zeros
                const bool k0 = real && cpos < ldc, k1 = real && cpos + 4 < ldc;
                lim0[S] = k0 ? 65504.0f : 0.0f;
                lim1[S] = k1 ? 65504.0f : 0.0f;
                const char* base = reinterpret_cast<const char*>(real ? src : p.sx1);
                const unsigned ld4 = (unsigned)(real ? ldc : p.sc1) * 4u, o0 = (unsigned)(k0 ? cpos : 0) * 4u, o1 = (unsigned)(k1 ? cpos + 4 : 0) * 4u;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    ar0[S][t] = *reinterpret_cast<const f32x4*>(base + (sv[t] * ld4 + o0));
                    ar1[S][t] = *reinterpret_cast<const f32x4*>(base + (sv[t] * ld4 + o1));
                }
                const char* ws = swf + (size_t)(pp < p.s_npairs ? pp : p.s_npairs - 1) * 8192;
#pragma unroll
                for (int f = 0; f < 8; ++f) bw[S][f] = *reinterpret_cast<const h8*>(ws + f * 1024);
            };
#define DM3D_TMFMA(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
            auto t_step = [&](auto SET) {
                constexpr int S = decltype(SET)::value;
                h8 ahi[8], alo[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) split8(ar0[S][t], ar1[S][t], lim0[S], lim1[S], ahi[t], alo[t]);
                // A VALU write of an MFMA operand needs two wait states before an asm MFMA reads it: hipcc pads one at the statement boundary,
                // and with one the MFMA reads registers 0-1 of the operand STALE in > 99 % of the cases (tools/micro/mfma_hazards.hip,
                // profiles/r05_mfma_hazards.log).  A v_cvt_pk of the split sat one state in front of the first MFMA; it happened to write
                // register 3, which the chip reads a state later (the r04 library's results are right: profiles/r05_skip_tail_err.log) —
                // luck of the register allocation, not construction.  tools/isa_hazard.py checks this file's ISA (tests/test_isa_hazard.py).
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 1" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // pass-major: an accumulator tile is touched once per 32 MFMAs; tile t = 4 s + 2 g + parity -> acc[parity][2 s + g][ni]
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) DM3D_TMFMA(acc[t & 1][t >> 1][ni], alo[t], bw[S][2 * ni]);
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) DM3D_TMFMA(acc[t & 1][t >> 1][ni], ahi[t], bw[S][2 * ni + 1]);
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) DM3D_TMFMA(acc[t & 1][t >> 1][ni], ahi[t], bw[S][2 * ni]);
            };
#undef DM3D_TMFMA
            const std::integral_constant<int, 0> S0;
            const std::integral_constant<int, 1> S1;
            t_load(S0, p_lo);
            for (int pp = p_lo; pp < np; pp += 2) {            // whole rounds of two pairs: a step past the last pair multiplies zeros
                t_load(S1, pp + 1);
                t_step(S0);
                t_load(S0, pp + 2);
                t_step(S1);
            }
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");      // (asm MFMAs: hipcc's hazard pass does not know the accumulators were just written)
        }
        // ---- the shared epilogue, one slice at a time straight from the accumulators (the LDS holds the next item's image and weights):
        // tile (2g + parity, ni) of slice s = rows 4g .. 4g+3, x = 2 * x-pair + parity; without a skip conv the output transform happens here.
        auto slice_tiles = [&](auto S_, f32x4v (&e)[4][4]) {
            constexpr int s = decltype(S_)::value;
            // (pinned in the accumulator file up to here: hipcc otherwise copies both slices' tiles out at the loop exit — 200 registers)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) asm volatile("" : "+a"(acc[t][2 * s + g][ni]));
            if (skip) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) { e[2 * g][ni] = acc[0][2 * s + g][ni]; e[2 * g + 1][ni] = acc[1][2 * s + g][ni]; }
            } else {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        const f32x4v m0 = acc[0][2 * s + g][ni], m1 = acc[1][2 * s + g][ni], m2 = acc[2][2 * s + g][ni], m3 = acc[3][2 * s + g][ni];
                        e[2 * g][ni] = (m0 + m1) + m2;
                        e[2 * g + 1][ni] = (m1 - m2) - m3;
                    }
            }
        };
        // Cin split (the 8^3 level at B = 32): both parts store their transformed tiles, the part that draws the tile's last ticket sums them in
        // part order and runs the epilogue (split_* in dm3d_conv_h3v2_parts.h); the other one goes on to its next item.
        bool finish = true;                                     // (uniform)
        const long tile = (long)(item / ny) * ntiles + cur.ntile;
        SplitTile stile = {};
        if (p.ksplit > 1) {
            stile = split_tile(p, tile, cur.khalf);
            static_for<2>([&](auto S_) {
                __builtin_amdgcn_sched_barrier(0);
                f32x4v e[4][4];
                slice_tiles(S_, e);
                split_store(stile, e, 16 * decltype(S_)::value);
            });
            finish = split_is_last(p, tile, reinterpret_cast<unsigned*>(lds_in + HROWS * RREC * REC));      // (a word behind the image: nobody else's)
        }
        if (finish) {
            float gn[8];                        // fused GroupNormalization statistics of the output (ConvArgs.gn_stats): this lane's four channels over both slices
            const EpiVecs ev = epilogue_cq_vecs(p, br, vrow_s);
#pragma unroll
            for (int i = 0; i < 8; ++i) gn[i] = 0.0f;
            static_for<2>([&](auto S_) {
                constexpr int s = decltype(S_)::value;
                __builtin_amdgcn_sched_barrier(0);
                if (s == 1) STAMP(21);
                f32x4v e[4][4];
                slice_tiles(S_, e);
                if (p.ksplit > 1) split_gather(stile, p, cur.khalf, e, 16 * s);
#ifndef DM3D_EXP_NO_EPILOGUE                    // (timing-only A/B arm, tools/mk_variant_conv.sh: what hiding the whole epilogue could buy at most)
                epilogue_cq<TD>(p, e, br, 2 * wave + s, ev, vrow_s, p.gn_stats ? gn : nullptr);
#else
                if (e[0][0][0] == 12345.678f) p.out[0] = e[1][1][1] + e[2][2][2] + e[3][3][3];
#endif
            });
            if (p.gn_stats) gn_flush_cq<TD>(p, gn, br, 2 * wave, 2);
        }
        STAMP(29);
        if (!has_next) break;
        // the accumulators of the next item: zeroed only now, with every old value dead (zeroing a slice's tiles as soon as they were read
        // made hipcc's allocator move the OTHER slice's tiles out of the accumulator file, through vector registers and scratch memory)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int pi = 0; pi < 4; ++pi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    acc[t][pi][ni] = f32x4v{0.f, 0.f, 0.f, 0.f};
                    asm volatile("" : "+a"(acc[t][pi][ni]));
                }
        item += item_step;
        cur = decode(item);
        vrow_s = vec_row(cur);
        c_lo = cur.khalf * cpp; c_hi = c_lo + cpp;
        has_next = item + item_step < item_end;
    }
    }       // items
}

template <int MODE>
int launch_w(ConvArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)(4 * 2 * 64 * REC + 10 * 10 * 17 * REC) * sizeof(_Float16) + 16;      // 32 KB of weights + 106 KB of image + the split form's ticket word
    static_assert(lds <= 160 * 1024, "one workgroup per CU");
    // (per device: the attribute and the CU count belong to the device the launch goes to)
    static std::atomic<int> cus[64] = {};
    int dev = 0;
    DM3D_HIP(hipGetDevice(&dev));
    DM3D_REQUIRE(dev >= 0 && dev < 64, "conv: device ordinal %d", dev);
    if (cus[dev].load(std::memory_order_acquire) == 0) {       // (two host threads may meet here: both set the same attribute and store the same count)
        DM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_igemm_h3w<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int n = 0;
        DM3D_HIP(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        cus[dev].store(n > 0 ? n : 256, std::memory_order_release);
    }
    const int ncu = cus[dev].load(std::memory_order_relaxed);
    H3v2Launch L;
    if (int rc = dm3d_h3v2_pre_launch(a, 8, L, st, dm3d_conv_h3w_ksplit(a))) return rc;
    L.k.wpk = a.wpk_wino;
    // work items = bricks x column tiles x Cin parts; one persistent workgroup per CU walks its share of the list
    // (DM3D_CONV_WINO_PERSIST=0: one item per workgroup)
    const long items = (long)a.batch * a.bd * a.bh * a.bw * (a.coutpad / 64) * a.ksplit;
    static const bool persist = [] { const char* e = getenv("DM3D_CONV_WINO_PERSIST"); return !(e && e[0] == '0'); }();
    long g = items;
    // (a partition that reports fewer than 8 CUs has no whole XCD round: one workgroup per CU, never an empty grid)
    if (persist && items > ncu) g = (items % 8 == 0 && ncu >= 8) ? (ncu / 8 * 8) : ncu;
    if (g < 1) g = 1;
    if (persist) {                           // test knob (read per call): at most this many workgroups, so that small shapes walk item lists too
        const char* cap = getenv("DM3D_CONV_WINO_GRID");
        if (cap && atol(cap) > 0 && atol(cap) < g) g = atol(cap);
    }
    hipLaunchKernelGGL((conv3d_igemm_h3w<MODE>), dim3((unsigned)g), dim3(256), lds, st, L.k);
    if (int rc = dm3d_launch_check("conv3d_igemm_h3w")) return rc;
    return dm3d_h3v2_post_launch(a, L, st);
}

}  // namespace

// The Winograd form serves a k3 / stride-1 launch when the caller supplied the transformed image (wpk_wino), the volume is whole 8 x 8 x 8
// bricks, a fused skip conv is short or the main loop long (below), the grid gives every CU two workgroups in turn or exactly one
// (the same threshold as the 8-slice bricks of the direct kernel: DM3D_CONV_WIDE_WGS) and Cin is at least 32 (two 16-channel chunks,
// DM3D_CONV_WINO_MINCHUNKS: its unoverlapped prologue and epilogue cost as much as two chunks; at 32 input channels it is still 9 %
// ahead of the direct kernel, at 64 10-13 %, profiles/r03_wino_ab.log).  The input tensors must be below 4 GB (32-bit lane offsets).  DM3D_CONV_WINO=0 (A/B knob, read per
// call): never.
bool dm3d_conv_h3w_serves(const ConvArgs& a, int which) {
    if (!a.wpk_wino || which != DM3D_CONV_K3S1 || a.parity || a.cout <= 32) return false;
    // (a fused skip conv needs its weights as operand fragments: dm3d_conv_desc.skip_wpk_frag; raw skip inputs below 4 GB: 32-bit lane offsets)
    if (a.s_npairs > 0) {
        if (!a.swpk_f) return false;
        const long long svox = (long long)a.batch * a.ind * a.inh * a.inw;
        if (svox * (a.sc1 > a.sc2 ? a.sc1 : a.sc2) * 4 >= (1ll << 32)) return false;
    }
    if (a.od % 8 != 0 || a.oh % 8 != 0 || a.ow % 8 != 0 || a.padz != 1 || a.pady != 1 || a.padx != 1) return false;
    const char* e = getenv("DM3D_CONV_WINO");
    if (e && e[0] == '0') return false;
    const char* mc = getenv("DM3D_CONV_WINO_MINCHUNKS");
    if (a.nchunks < (mc ? atoi(mc) : 2)) return false;
    const long long vox = (long long)a.batch * a.ind * a.inh * a.inw;
    if (vox * (a.c1 > a.c2 ? a.c1 : a.c2) * 4 >= (1ll << 32)) return false;
    const long wgs = (long)a.batch * (a.od / 8) * (a.oh / 8) * (a.ow / 8) * (a.coutpad / 64) * dm3d_conv_h3w_ksplit(a);
    const char* w = getenv("DM3D_CONV_WIDE_WGS");
    const long need = w ? atol(w) : 512L;
    // (one workgroup per CU: exactly 256 workgroups are one full round — B = 4 at 32^3, config 2: 3.21 -> 3.09 ms per step —; between 256
    // and 512 the second round would be part empty, which two small workgroups per CU of the direct kernel handle better)
    return wgs >= need || wgs == 256 || (dm3d_conv_h3w_ksplit(a) > 1 && wgs >= (need < 256 ? need : 256L));
}

// Cin split of the Winograd form: two workgroups per brick and column tile where one would leave at least half of the CUs without work
// (the 8^3 level at B = 32: 32 bricks x 4 column tiles), each contracting half of the chunks (at least eight) and half of a fused skip
// conv's pairs; the halves meet inside the launch (the hand-over form, dm3d_conv_h3v2_parts.h: any epilogue, fused statistics and output
// formats included).  Needs the host's split_counters (and scratch: dm3d_conv_scratch_bytes).  DM3D_CONV_WINO_SPLIT=0: never.
int dm3d_conv_h3w_ksplit(const ConvArgs& a) {
    static const bool off = [] { const char* e = getenv("DM3D_CONV_WINO_SPLIT"); return e && e[0] == '0'; }();
    if (off || !a.split_counters || a.nchunks % 2 != 0 || a.nchunks < 16) return 1;
    const long wgs = (long)a.batch * (a.od / 8) * (a.oh / 8) * (a.ow / 8) * (a.coutpad / 64);
    return wgs <= 128 ? 2 : 1;
}

int dm3d_conv_launch_h3w(ConvArgs& a, int which, hipStream_t st) {
    (void)which;
    if (a.x_h2) return launch_w<2>(a, st);
    return a.pscale ? launch_w<1>(a, st) : launch_w<0>(a, st);
}
