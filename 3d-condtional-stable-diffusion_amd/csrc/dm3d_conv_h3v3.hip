// dm3d_conv_h3v3.hip — the k3 / stride-1 (and 2x2x2 parity) split-float16 Conv3d as FREE-RUNNING, software-pipelined waves.
//
// reference op: Conv3D(width, 3, padding="same") behind BatchNormalization + swish, + time-embedding / bias / residual adds
// (networks/conditional_dm3d.py:254-268), UpSampling3D + Conv3D (:288-296), Concatenate + ResidualBlock on the up path (:394-404).
//
// Same arithmetic, operand geometry, LDS images, packed weight image (DM3D_WL_PAIR), skip-conv tail phase and epilogue as the round-1/2
// kernels (git history; the pieces this kernel shares with the Winograd-x form, dm3d_conv_h3w.hip, are in dm3d_conv_h3v2_parts.h).  What changed is the
// skeleton.  Round 3 measured the round-2 kernel with one stamp pair around its chunk loop (tools/kernel_clock.py, profiles/
// r03_v2_clocks.log): the chip held 1.93-2.02 GHz under it and the matrix pipe was busy 72-75 % of the loop — against 87 % at 1.87 GHz
// for a bare LDS-fed MFMA loop (tools/micro/mfma_shapes) — and 67 % of the launch (SQ_VALU_MFMA_BUSY_CYCLES, profiles/
// r03_v2_sq_conv_pro192.csv).  That kernel alternated a LOAD and a COMPUTE segment between the two waves of a SIMD with a workgroup barrier
// after every 48 MFMAs: ~990 cycles per segment for 768 of MFMA, plus a prologue and an epilogue that nothing overlapped (one 512-thread
// workgroup per CU).  Here:
//   * every wave runs its own pipeline.  A tap pair is three passes of 16 MFMAs on registers — A: al.bh, B: ah.bh, C: ah.bl — and each
//     pass requests the fragments of a LATER pass into the registers the previous pass released: A(p) requests ah(p), bl(p); B(p)
//     requests al(p+1); C(p) requests bh(p+1).  No fragment is double-buffered (64 VGPRs of fragments as before) and every ds_read has
//     256-512 MFMA cycles to land.  A wave alone can keep its SIMD's matrix pipe fed.
//   * two independent 256-thread workgroups per CU (4-slice bricks, 78.8 KB of LDS each): one workgroup's prologue, epilogue, chunk
//     boundary and barrier waits run beside the other's MFMAs, which no structure inside ONE workgroup could give.
//   * weights stream per tap PAIR (8 KB by LDS-DMA) through a ring of four buffers, three pairs of lead (~2 300 MFMA cycles; a fill takes
//     ~1.1 us from issue to landing).  ONE workgroup barrier per pair, at the head of pass B: behind this wave's `lgkmcnt(0)` (its reads of
//     pair p's buffer — bh in C(p-1), bl in A(p) — are back) and `vmcnt(N)` (its pieces of pair p+1 have landed) it (1) frees pair p's
//     buffer for the DMA of pair p+4, issued right after it, and (2) makes pair p+1 visible to everyone, first read in C(p).
//   * the halo image is single: behind the barrier of the chunk's LAST pair every wave has all its voxel reads back, so the next
//     chunk's image (converted in registers during passes C(8..)) is stored during B/C of that pair, one more barrier publishes it.
//   * workgroups are renumbered so that the 8 XCDs each take a contiguous range of (brick, column tile) work: neighbouring bricks
//     share halo voxels and the column tiles of one brick share all of them — served by that XCD's L2 instead of from beyond it.
// Per accumulator the order is al.bh, ah.bh, ah.bl (rounds 1-2: al.bh, ah.bl, ah.bh): float32 accumulation, same error bound, not the same bits.
// What it bought: matrix-pipe duty in the loop 0.73 -> 0.82 with two workgroups per CU — and an in-kernel clock of
// 1.85 instead of 1.95 GHz, i.e. the same wall time; on all-zero operands the same stream runs at 2.39 GHz, 30 % faster.  The split-float16
// conv is bound by what the MFMA array draws on random data, not by its schedule; this kernel is kept because it is one loop for every
// form (k3, parity, 4 / 8 slices, 16 / 32 / 64 columns) and because its work assignment halves the HBM traffic.
#include <cstdlib>
#include "dm3d_conv_h3v2_parts.h"

using namespace h3v2;

// Diagnostic build only (-DDM3D_CLOCK_STAMPS, tools/mk_stamp_variants.py -> variants/cck.so; the product library carries none of it): thread 0 of
// every workgroup writes s_memtime / s_memrealtime at kernel entry (0), when the first halo is back (2), around the chunk loop (1, 28), behind
// the skip phase (19), behind the hand-over store / ticket / gather of a Cin-split launch (20, 21, 22) and at the end (29) into a buffer of its
// own — the in-kernel clock and matrix-pipe duty of tools/kernel_clock.py (MI355X_MICROARCH.md, 'DVFS give-back' item 6) and the phase table
// of tools/split_phases.py.
#ifdef DM3D_CLOCK_STAMPS
__device__ unsigned long long* g_dbg_stamps_c = nullptr;
extern "C" int dm3d_debug_set_stamps_conv(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_stamps_c), &p, sizeof(p)); }
#define STAMP(i) do { const unsigned wg_ = blockIdx.x + gridDim.x * blockIdx.y; \
    if (g_dbg_stamps_c && threadIdx.x == 0 && blockIdx.z == 0 && wg_ < 4096) { \
    g_dbg_stamps_c[wg_ * 32 + (i)] = __builtin_amdgcn_s_memtime(); \
    if ((i) == 0) g_dbg_stamps_c[wg_ * 32 + 24] = __builtin_amdgcn_s_memrealtime(); \
    if ((i) == 20) g_dbg_stamps_c[wg_ * 32 + 26] = __builtin_amdgcn_s_memrealtime(); \
    if ((i) == 29) g_dbg_stamps_c[wg_ * 32 + 25] = __builtin_amdgcn_s_memrealtime(); \
    if ((i) == 1) g_dbg_stamps_c[wg_ * 32 + 30] = __builtin_amdgcn_s_memrealtime(); \
    if ((i) == 28) g_dbg_stamps_c[wg_ * 32 + 31] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

namespace {

// MODE 0: float32 input as is; 1: float32 input through the fused norm + SiLU prologue; 2: x1 already activated and split (DM3D_FMT_H2)
// TD = z-slices per brick = waves per workgroup: 4 (256 threads, two workgroups per CU) or 8 (512 threads, one per CU)
// NCT = 16-column tiles per wave: 4 = the whole 64-column tile of the packed image; 1 / 2 = only its first 16 / 32 columns (conv_out 64 -> 8,
// conv_in 8 -> 32: round 2 ran them on the 32-column 32x32x16 kernel, 4x / 1x the needed MFMA work at 77 / 117 TFLOP/s algorithmic)
template <int KS, int MODE, int TD, int NCT = 4>
__global__ __launch_bounds__(TD * 64, 2) void conv3d_igemm_h3v3(const ConvArgs p) {
    constexpr int TH = 8, TW = 8, CK = 16, NT = 64, NTHR = TD * 64;
    constexpr int HD = TD - 1 + KS, HH = TH - 1 + KS, HW = TW - 1 + KS, HWP = 12;
    constexpr int HVOX = HD * HH * HW;
    constexpr int TAPS = KS * KS * KS, TAPSP = (TAPS + 3) / 4 * 4;   // the packed image pads the taps to groups of 4 (zero weights)
    constexpr int NP = TAPSP / 2;                                    // tap pairs per chunk: 14 (k3) or 4 (parity 2x2x2)
    constexpr int NSLOT = (HVOX * 2 + NTHR - 1) / NTHR;              // halo staging slots per thread: 5 (TD 4) or 4 (TD 8)
    constexpr int WPAIR = 2 * NT * REC;                              // halfs per weight pair (8 KB)
    constexpr int RING = 4;
    constexpr int WSLOT = WPAIR * 2 / 1024 / TD;                     // 1 KB DMA pieces per wave and pair: 2 or 1
    // the next chunk's halo: SPP staging slots are requested in pass B of each of the pairs 0 .. NREQ-1 and converted (in place) in pass C
    // of the pairs CV0 .. CV0+NREQ-1, all before the last pair, whose pass B stores the image (k3: one slot per pair; the 4-pair parity form: two)
    constexpr int SPP = (NSLOT + NP - 3) / (NP - 2);
    constexpr int NREQ = (NSLOT + SPP - 1) / SPP;
    constexpr int CV0 = NP - 1 - NREQ < 8 ? NP - 1 - NREQ : 8;
    static_assert(NREQ <= NP - 2 && CV0 >= 1 && CV0 + NREQ <= NP - 1, "request / conversion schedule of the staging slots");
    static_assert(WPAIR * 2 / 1024 % TD == 0, "a weight pair must be a whole number of 1 KB pieces per wave");

    extern __shared__ __attribute__((aligned(16))) _Float16 smem_v3[];
    _Float16* lds_w = smem_v3;                  // [RING][2 taps][NT][REC]   (first: every weight read is base + a 16-bit immediate)
    _Float16* lds_in = smem_v3 + RING * WPAIR;  // [HREC][REC]

    STAMP(0);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, q = (lane >> 4) & 1, row = lane & 15;

    // ---- XCD-aware work assignment.  Workgroups are dealt round-robin over the 8 XCDs in dispatch order (x fastest, then y), so
    // dispatch ids d and d + 8 share an XCD and its L2.  XCD k takes the k-th contiguous eighth of the (brick, column tile) list, column
    // tile fastest: bricks that share halo voxels, and the column tiles of one brick (which share all of them), run on one L2 at about the
    // same time.  A pure renumbering (a bijection for any grid): placement is never assumed for correctness.
    const int ny = gridDim.y;
    int brick, by;
    {
        const unsigned total = gridDim.x * gridDim.y;
        const unsigned d = blockIdx.x + gridDim.x * blockIdx.y;
        unsigned w = d;
        if ((total & 7u) == 0) w = (d & 7u) * (total >> 3) + (d >> 3);
        brick = (int)(w / (unsigned)ny);
        by = (int)(w - (unsigned)brick * (unsigned)ny);
    }
    const int brick_all = brick;                                    // over the whole batch: the tile number of a Cin-split launch counts with it
    const int bpv = p.bd * p.bh * p.bw;
    const int b = brick / bpv;
    brick -= b * bpv;
    const int oz0 = (brick / (p.bh * p.bw)) * TD;
    const int oy0 = ((brick / p.bw) % p.bh) * TH;
    const int ox0 = (brick % p.bw) * TW;
    // by = ntile + ntiles * khalf.  ksplit > 1 (small grids): this workgroup contracts chunks [c_lo, c_hi) only; the parts of a tile meet in
    // the hand-over form below (split_* of dm3d_conv_h3v2_parts.h).
    const int ntiles = p.coutpad / NT;
    const int ntile = by % ntiles, khalf = by / ntiles;
    const int c_lo = khalf * (p.nchunks / p.ksplit), c_hi = c_lo + p.nchunks / p.ksplit;
    int padz = p.padz, pady = p.pady, padx = p.padx, ooz = p.ooz, ooy = p.ooy, oox = p.oox;
    const _Float16* wbase = static_cast<const _Float16*>(p.wpk);
    if (p.parity) {
        const int par = blockIdx.z;
        ooz = par >> 2; ooy = (par >> 1) & 1; oox = par & 1;
        padz = 1 - ooz; pady = 1 - ooy; padx = 1 - oox;
        wbase += (size_t)par * p.w_parity_stride;
    }

    // ---- halo staging slots: thread t moves 16-byte piece t & 1 of voxels (t >> 1) + j * NTHR/2
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
    int a_rec = (wave * HH + (row & 3)) * HWP + dx_of_row(row);
    // weight rows: LDS position PI(row) inside the 16-column tile, the lane half picks the tap of the pair
    const int b_pos = pi_pos(row);
    const int b_hi = (half * NT + b_pos) * REC + ((q ^ swz(b_pos)) << 3);

    f32x4v acc[4][NCT];
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
#pragma unroll
        for (int ni = 0; ni < NCT; ++ni) acc[pi][ni] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // weights go global -> LDS by LDS-DMA: the packed image IS the LDS image, pair pq of this (column tile) is a linear 8 KB copy; wave w
    // moves the 1 KB pieces w, w + TD.  Ring slot = running pair number mod 4 (NP mod 4 = 2 for k3: the phase differs from chunk to chunk).
    // Buffer form of the LDS-DMA (as in dm3d_conv_h3w.hip): the image's descriptor in scalar registers, ONE constant 32-bit lane offset, the
    // pair as a scalar byte offset.  The flat form (global_load_lds with a 64-bit lane address) is a FLAT-encoded instruction to hipcc's
    // wait-count pass: with one pending it treats the vector-memory counter as out of order and turns every wait of its own for a plain
    // load's result into vmcnt(0) — in the 4-pair parity form, whose staging slots are converted one pair behind their request, that put a
    // vmcnt(0) in front of the kernel's own counted wait in every chunk, draining the weight ring (isa listing of <2, 0, 4, 4>, round 5).
    const auto w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(wbase), (short)0, (int)((long)ntiles * p.nchunks * (NP * WPAIR * 2)), 0x00020000);
    const int w_voff = wave * 1024 + lane * 16;
    // (uniform values that hipcc computed on the vector unit — the divisions of the work assignment — go through readfirstlane: the scalar
    // offset and M0 must be scalar registers, and for a value it cannot prove uniform hipcc builds a waterfall loop around the instruction)
    const unsigned w_tile = (unsigned)__builtin_amdgcn_readfirstlane((ntile * p.nchunks) * (NP * WPAIR * 2));      // byte offset of this column tile's pairs
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int pq_end = c_hi * NP;
    auto fetch_w = [&](int pq, int slot) {              // unconditional (past the end the last pair is fetched again): hipcc can count what is in flight
        const unsigned off = w_tile + (unsigned)__builtin_amdgcn_readfirstlane((pq < pq_end ? pq : pq_end - 1) * (WPAIR * 2));
        char* dst = reinterpret_cast<char*>(lds_w) + __builtin_amdgcn_readfirstlane(slot) * (WPAIR * 2) + wave_u * 1024;
#pragma unroll
        for (int i = 0; i < WSLOT; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (__attribute__((address_space(3))) void*)(dst + i * (TD * 1024)), 16, w_voff,
                                                     (int)(off + i * (TD * 1024)), 0, 0);
    };
    int pq = c_lo * NP;                                  // running pair number of the pair being multiplied
    int ws = 0;                                          // its ring slot
#pragma unroll
    for (int i = 0; i < RING; ++i) fetch_w(pq + i, i);   // the ring starts full: pairs 0 .. 3; pass B of pair pp refills pp's buffer with pair pp + 4

    constexpr bool pro = MODE == 1, xh2 = MODE == 2;
    f32x4 raw0[NSLOT], raw1[NSLOT];
    f32x4 sc0 = {1.f, 1.f, 1.f, 1.f}, sc1 = sc0, sh0 = {0.f, 0.f, 0.f, 0.f}, sh1 = sh0;
    bool ok0 = false, ok1 = false;
    // One slot (two 16-byte loads per thread) of chunk ch's halo; all slots of a chunk read the same channel piece.  Loads are never
    // behind a branch (clamped addresses, masked afterwards): a load under a divergent `if` makes hipcc wait vmcnt(0) on the spot.
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
    // what the conversion of chunk ch needs beside the voxels: channel validity and, behind a fused norm, the scale / shift vectors
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
    // vector-memory requests issued in pass B of pair pp beside its weight DMA: SPP staging slots of the next chunk (+ its parameters with
    // slot 0).  A negative pp is a late pair of the previous chunk (in the first chunk nothing was requested there, but then the pieces that
    // wait is for were drained in the prologue).
    auto halo_ops = [&](int pp) {
        if (pp < 0) pp += NP;
        if (pp >= NREQ) return 0;
        const int n = NSLOT - pp * SPP < SPP ? NSLOT - pp * SPP : SPP;
        return 2 * n + (pp == 0 ? PLOADS : 0);
    };
    // in place: prologue norm + SiLU, float16 split -> the two 16-byte pieces the LDS image takes
    auto convert_slot = [&](const int j) {
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
        } else {
            split8(v0, v1, (in && ok0) ? 65504.0f : 0.0f, (in && ok1) ? 65504.0f : 0.0f, shi_j, slo_j);
        }
        raw0[j] = __builtin_bit_cast(f32x4, shi_j);
        raw1[j] = __builtin_bit_cast(f32x4, slo_j);
    };
    // (the lo piece's address is made on the spot inside the asm pair: as C++ stores hipcc keeps the NSLOT hi AND lo addresses as lane
    // constants across the chunk loop, and in the 4-slice prologue form — 256 registers at two waves per SIMD — spilled six of them: every
    // store of the image then reloaded its address behind a vmcnt(0), i.e. behind the weight DMA pieces just issued)
    const unsigned img_addr = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)lds_in;
    auto store_image = [&]() {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            if (st_off[j] >= 0) {
                const unsigned addr = img_addr + (unsigned)st_off[j] * 2u;
                unsigned lo_addr;
                asm volatile("ds_write_b128 %1, %2\n\tv_xor_b32 %0, 32, %1\n\tds_write_b128 %0, %3"
                             : "=&v"(lo_addr) : "v"(addr), "v"(raw0[j]), "v"(raw1[j]) : "memory");
            }
        }
    };
    // raw barrier behind this wave's own LDS traffic: __syncthreads() would also drain the vector-memory counter, i.e. wait for the
    // LDS-DMA fills that are meant to stay in flight across it
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: the first chunk's image
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) load_halo_slot(c_lo, j);
    load_chunk_params(c_lo);
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0) (with an instruction hipcc's wait-count pass sees): halo and the three weight pairs
    STAMP(2);
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) convert_slot(j);
    store_image();
    lds_barrier();

    // ---- operand reads.  Pair pp of a chunk = taps 2pp, 2pp + 1 (the lane half picks the tap; the pad tap re-reads the last real tap's
    // voxels against zero weights); patch (py, px) of the wave's slice sits 48 * py + 4 * px records further.
    h8 ah[4], al[4], bh[NCT], bl[NCT];
    auto a_offs = [&](const int pp, int& o0, int& o1) {
        const int ta = 2 * pp, tb = ta + 1;
        const int tac = ta < TAPS ? ta : TAPS - 1, tbc = tb < TAPS ? tb : TAPS - 1;
        const int rec_a = ((tac / (KS * KS)) * HH + (tac / KS) % KS) * HWP + tac % KS;
        const int rec_b = ((tbc / (KS * KS)) * HH + (tbc / KS) % KS) * HWP + tbc % KS;
        const int v0 = a_rec + (half ? rec_b : rec_a);
        const int v1 = v0 + 4;
        o0 = v0 * REC + ((q ^ swz(v0)) << 3);
        o1 = v1 * REC + ((q ^ swz(v1)) << 3);
    };
    // The fragment reads are inline asm on purpose.  Behind a `global_load_lds` hipcc's wait-count pass treats the LDS counter as unordered
    // ("pending flat": the DMA is a FLAT-encoded instruction that touches LDS) and guards every use of a ds_read result with lgkmcnt(0) —
    // also rewriting an explicit counted wait to 0 — which would serialise "request for a later pass" and "multiply this pass".  As asm the
    // reads are invisible to that pass; every wait for them below is ours: lgkmcnt(8) in front of pass A (al, bh are back; ah, bl may be
    // out), lgkmcnt(0) at the head of pass B (ah, bl), nothing for pass C.  A fragment register is re-requested right behind the last MFMA
    // that reads it: the MFMA takes its A / B sources in its first cycles, LDS data comes back tens of cycles later.
    const unsigned in_addr = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)lds_in;
    const unsigned w_addr = (unsigned)(size_t)(__attribute__((address_space(3))) _Float16*)lds_w;
#define DM3D_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    auto read_ah = [&](const int pp) {
        int o0, o1; a_offs(pp, o0, o1);
        const unsigned a0 = in_addr + o0 * 2, a1 = in_addr + o1 * 2;
        DM3D_DSR(ah[0], a0, 0); DM3D_DSR(ah[1], a1, 0); DM3D_DSR(ah[2], a0, 48 * REC * 2); DM3D_DSR(ah[3], a1, 48 * REC * 2);
    };
    auto read_al = [&](const int pp) {
        int o0, o1; a_offs(pp, o0, o1);
        const unsigned a0 = in_addr + (o0 ^ 16) * 2, a1 = in_addr + (o1 ^ 16) * 2;
        DM3D_DSR(al[0], a0, 0); DM3D_DSR(al[1], a1, 0); DM3D_DSR(al[2], a0, 48 * REC * 2); DM3D_DSR(al[3], a1, 48 * REC * 2);
    };
    auto read_bh = [&](int slot) {
        const unsigned wa = w_addr + (slot * WPAIR + b_hi) * 2;
        DM3D_DSR(bh[0], wa, 0);
        if constexpr (NCT >= 2) DM3D_DSR(bh[1], wa, 16 * REC * 2);
        if constexpr (NCT == 4) { DM3D_DSR(bh[2], wa, 32 * REC * 2); DM3D_DSR(bh[3], wa, 48 * REC * 2); }
    };
    auto read_bl = [&](int slot) {
        const unsigned wa = w_addr + (slot * WPAIR + (b_hi ^ 16)) * 2;
        DM3D_DSR(bl[0], wa, 0);
        if constexpr (NCT >= 2) DM3D_DSR(bl[1], wa, 16 * REC * 2);
        if constexpr (NCT == 4) { DM3D_DSR(bl[2], wa, 32 * REC * 2); DM3D_DSR(bl[3], wa, 48 * REC * 2); }
    };
#define DM3D_PASS(A, B)                                                                                          \
    _Pragma("unroll") for (int pi_ = 0; pi_ < 4; ++pi_)                                                        \
        _Pragma("unroll") for (int ni_ = 0; ni_ < NCT; ++ni_)                                                  \
            acc[pi_][ni_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[pi_], B[ni_], acc[pi_][ni_], 0, 0, 0)

    read_al(0);
    read_bh(0);
    STAMP(1);

    for (int ch = c_lo; ch < c_hi; ++ch) {
        asm volatile("" : "+v"(a_rec));      // keeps the per-pair operand addresses from being hoisted out of the chunk loop (spills)
        const int ch_next = ch + 1 < c_hi ? ch + 1 : ch;           // (past the end: the last chunk again, unconditional like the DMAs)
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
            const bool last = pp == NP - 1;
            const int ws1 = (ws + 1) & (RING - 1);
            // ---- pass A: al(pp).bh(pp); requests ah(pp), bl(pp)
            read_ah(pp);
            read_bl(ws);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(4 + NCT) : "memory");   // al, bh (requested a pass or more ago) are back; the 4 + NCT reads above may be out
            __builtin_amdgcn_sched_barrier(0);
            DM3D_PASS(al, bh);
            __builtin_amdgcn_sched_barrier(0);
            // ---- pass B: ah(pp).bh(pp).  Its head is the pair's one barrier (see the file comment).
            {
                // newer than this wave's pieces of pair pp+1 (issued in B(pp-3)): the pieces of pairs pp+2, pp+3 and the staging requests
                // of B(pp-3) .. B(pp-1) (negative: the previous chunk's late pairs, which request nothing)
                constexpr int DMA_BEHIND = 2 * WSLOT;
                const int extra = halo_ops(pp - 3) + halo_ops(pp - 2) + halo_ops(pp - 1);
                // (the builtin, not inline asm: hipcc's own wait-count pass must see that every LDS read is back here, or it guards the
                // next passes with lgkmcnt(0) / vmcnt(0) of its own.)  simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 0 << 8 | vmcnt[5:4] << 14
#define DM3D_WAIT_VM_LGKM0(n) __builtin_amdgcn_s_waitcnt((((n) & 15) | (7 << 4) | ((((n) >> 4) & 3) << 14)))
                switch (extra) {
                case 0:  DM3D_WAIT_VM_LGKM0(DMA_BEHIND); break;
                case 2:  DM3D_WAIT_VM_LGKM0(DMA_BEHIND + 2); break;
                case 4:  DM3D_WAIT_VM_LGKM0(DMA_BEHIND + 4); break;
                case 6:  DM3D_WAIT_VM_LGKM0(DMA_BEHIND + 6); break;
                case 8:  DM3D_WAIT_VM_LGKM0(DMA_BEHIND + 8); break;
                case 10: DM3D_WAIT_VM_LGKM0(DMA_BEHIND + 10); break;
                case 12: DM3D_WAIT_VM_LGKM0(DMA_BEHIND + 12); break;
                case 14: DM3D_WAIT_VM_LGKM0(DMA_BEHIND + 14); break;
                case 16: DM3D_WAIT_VM_LGKM0(DMA_BEHIND + 16); break;
                default: DM3D_WAIT_VM_LGKM0(0); break;
                }
#undef DM3D_WAIT_VM_LGKM0
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            fetch_w(pq + RING, ws);                                      // into the buffer pair pp just left
            if (pp < NREQ) {
#pragma unroll
                for (int j = pp * SPP; j < (pp + 1) * SPP && j < NSLOT; ++j) load_halo_slot(ch_next, j);
            }
            if (pp == 0) load_chunk_params(ch_next);
            if (!last) read_al(pp + 1);
            else store_image();                                          // every wave has its voxel reads of this chunk back: the image is free
            __builtin_amdgcn_sched_barrier(0);
            DM3D_PASS(ah, bh);
            __builtin_amdgcn_sched_barrier(0);
            // ---- pass C: ah(pp).bl(pp); requests bh(pp+1) (pair pp+1 is visible since the barrier above)
            read_bh(ws1);
            if (pp >= CV0 && pp < CV0 + NREQ) {                          // the next chunk's staging registers
                // Their requests went out in pass B of pair pp - CV0.  k3: that is >= 8 pairs back and the barrier heads since have waited
                // past them.  The 4-pair parity form converts one pair after the request: say so with a counted wait hipcc's pass sees
                // (newer: the DMA pieces and staging requests of the pairs in between), or it guards the registers with vmcnt(0).
                if constexpr (CV0 < 3) {
                    int newer = 0;
#pragma unroll
                    for (int k_ = pp - CV0 + 1; k_ <= pp; ++k_) newer += WSLOT + halo_ops(k_);
                    switch (newer) {
                    case 1: __builtin_amdgcn_s_waitcnt(0x0F71); break;   // vmcnt(n), expcnt 7, lgkmcnt 15: n | 0x0F70
                    case 2: __builtin_amdgcn_s_waitcnt(0x0F72); break;
                    case 3: __builtin_amdgcn_s_waitcnt(0x0F73); break;
                    case 4: __builtin_amdgcn_s_waitcnt(0x0F74); break;
                    case 5: __builtin_amdgcn_s_waitcnt(0x0F75); break;
                    case 6: __builtin_amdgcn_s_waitcnt(0x0F76); break;
                    case 7: __builtin_amdgcn_s_waitcnt(0x0F77); break;
                    case 8: __builtin_amdgcn_s_waitcnt(0x0F78); break;
                    case 9: __builtin_amdgcn_s_waitcnt(0x0F79); break;
                    case 10: __builtin_amdgcn_s_waitcnt(0x0F7A); break;
                    default: __builtin_amdgcn_s_waitcnt(0x0F70); break;
                    }
                }
#pragma unroll
                for (int j = (pp - CV0) * SPP; j < (pp - CV0 + 1) * SPP && j < NSLOT; ++j) convert_slot(j);
            }
            __builtin_amdgcn_sched_barrier(0);
            DM3D_PASS(ah, bl);
            __builtin_amdgcn_sched_barrier(0);
            if (last) {
                lds_barrier();                                           // the new image is visible
                read_al(0);
                __builtin_amdgcn_sched_barrier(0);
            }
            ws = ws1;
            ++pq;
        }
    }
#undef DM3D_PASS
#undef DM3D_DSR
    STAMP(28);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          // the redundant tail fetches / reads: nothing may land in LDS the skip phase reuses
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const Brick br = {b, oz0, oy0, ox0, ooz, ooy, oox, ntile, khalf};
    if constexpr (KS == 3 && NCT == 4) {            // (the launcher admits a skip conv behind k3 / stride 1 / Cout > 32 only)
        // (the launcher sizes the dynamic LDS for whichever of the two phases needs more: v3_lds_halfs)
        skip_phase<TD>(p, smem_v3, acc, br);
    }
    STAMP(19);
    if (p.ksplit > 1) {                     // (uniform) Cin split: store this part's tiles; the last part of the tile to arrive sums them and goes on
        const long tile = ((long)blockIdx.z * gridDim.x + brick_all) * ntiles + ntile;
        const SplitTile stile = split_tile(p, tile, khalf);
        split_store(stile, acc, 0);
        STAMP(20);
        if (!split_is_last(p, tile, reinterpret_cast<unsigned*>(smem_v3))) return;      // (the LDS is free: every wave is past the barrier above / the skip phase's last)
        STAMP(21);
        split_gather(stile, p, khalf, acc, 0);
        STAMP(22);
    }
    if (p.gn_stats) {                       // fused GroupNormalization statistics of the output (uniform; the launcher admits the 16-byte full-brick form only)
        float gn[NCT * 8];
#pragma unroll
        for (int i = 0; i < NCT * 8; ++i) gn[i] = 0.0f;
        epilogue<TD, NCT>(p, acc, br, -1, 4, -1, gn);
        gn_flush<TD, NCT>(p, gn, br, threadIdx.x >> 6, 1);
    } else {
        epilogue<TD, NCT>(p, acc, br);
    }
    STAMP(29);
}

template <int KS, int MODE, int TD, int NCT = 4>
int launch_v3(ConvArgs& a, hipStream_t st) {
    constexpr int HREC = (TD - 1 + KS) * (7 + KS) * 12;
    constexpr int main_halfs = HREC * REC + 4 * 2 * 64 * REC, skip_halfs = (KS == 3 && NCT == 4) ? skip_lds_halfs<TD>() : 0;
    constexpr size_t lds = (size_t)(main_halfs > skip_halfs ? main_halfs : skip_halfs) * sizeof(_Float16);
    static_assert((TD == 4 ? 2 : 1) * lds <= 160 * 1024, "workgroups per CU x LDS");
    static std::atomic<bool> attr_set[64] = {};            // per device: the attribute belongs to the device the launch goes to
    int dev = 0;
    DM3D_HIP(hipGetDevice(&dev));
    DM3D_REQUIRE(dev >= 0 && dev < 64, "conv: device ordinal %d", dev);
    if (!attr_set[dev]) {
        DM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_igemm_h3v3<KS, MODE, TD, NCT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    H3v2Launch L;
    if (int rc = dm3d_h3v2_pre_launch(a, TD, L, st)) return rc;
    dim3 grid((unsigned)(a.batch * a.bd * a.bh * a.bw), (unsigned)(a.coutpad / 64 * a.ksplit), a.parity ? 8u : 1u);
    hipLaunchKernelGGL((conv3d_igemm_h3v3<KS, MODE, TD, NCT>), grid, dim3(TD * 64), lds, st, L.k);
    if (int rc = dm3d_launch_check("conv3d_igemm_h3v3")) return rc;
    return dm3d_h3v2_post_launch(a, L, st);
}

template <int KS, int MODE>
int launch_td(ConvArgs& a, hipStream_t st) {
    if constexpr (KS == 3) {                        // the narrow column forms: k3 convs with few output channels
        if (a.cout <= 32) {
            DM3D_REQUIRE(a.s_npairs == 0 && !a.out_h2, "conv: the fused skip conv / hand-off output need cout > 32");
            // (4-slice bricks: with 8-slice bricks conv_out / conv_in were 4 % / 11 % slower — these launches are bound by staging the input,
            // 12 or 24 MFMAs per pair against the same halo, not by the matrix pipe: profiles/r03_layers_h3.log)
            return a.cout <= 16 ? launch_v3<KS, MODE, 4, 1>(a, st) : launch_v3<KS, MODE, 4, 2>(a, st);
        }
    }
    return dm3d_conv_h3v3_td(a) == 8 ? launch_v3<KS, MODE, 8>(a, st) : launch_v3<KS, MODE, 4>(a, st);
}

}  // namespace

// Brick depth of a launch.  8 slices (512 threads, one workgroup per CU: half the weight bytes per FLOP, halo factor 1.95 instead of 2.34)
// where the grid still gives every CU at least two such workgroups in turn; 4 slices (256 threads, two independent workgroups per CU,
// Cin splitting for tiny grids) for small grids, the parity convs (4 pairs per chunk: nothing for a wider barrier to amortise) and the
// launches with a fused skip phase.  DM3D_CONV_V3_TD (A/B knob, read per call): 4 = always 4; 8 = 8 wherever the grid allows; else auto.
int dm3d_conv_h3v3_td(const ConvArgs& a) {
    const char* e = getenv("DM3D_CONV_V3_TD");
    const int mode = e ? atoi(e) : 0;
    if (mode == 4) return 4;
    if (mode != 8 && (a.parity || a.s_npairs > 0)) return 4;
    if ((a.out_h2 || a.post_scale) && a.od % 8 != 0) return 4;      // the fused output forms live in the full-brick epilogue: whole bricks
    const long wgs = (long)a.batch * ((a.od + 7) / 8) * ((a.oh + 7) / 8) * ((a.ow + 7) / 8) * (a.coutpad / 64) * (a.parity ? 8 : 1);
    const char* w = getenv("DM3D_CONV_WIDE_WGS");                      // threshold override (tests force the 8-slice forms onto small shapes with 1)
    return wgs >= (w ? atol(w) : 512L) ? 8 : 4;
}

int dm3d_conv_launch_h3v3(ConvArgs& a, int which, hipStream_t st) {
    if (which == DM3D_CONV_UP) return a.pscale ? launch_td<2, 1>(a, st) : launch_td<2, 0>(a, st);
    if (a.x_h2) return launch_td<3, 2>(a, st);
    return a.pscale ? launch_td<3, 1>(a, st) : launch_td<3, 0>(a, st);
}
