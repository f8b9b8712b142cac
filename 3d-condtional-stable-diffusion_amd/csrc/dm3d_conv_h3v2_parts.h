// dm3d_conv_h3v2_parts.h — pieces shared by the two 16x16x32 split-float16 Conv3d kernels (dm3d_conv_h3v3.hip: the free-running
// software-pipelined direct kernel; dm3d_conv_h3w.hip: the persistent Winograd-x kernel, which calls the epilogue once per z-slice):
// operand geometry helpers, the direct kernel's LDS-staged 1x1 skip-conv tail phase, the epilogue and the GroupNormalization partial sums.
// The epilogue works on a [4 patches][column tiles] accumulator of one 8 x 8 z-slice.
#pragma once
#include "dm3d_conv_args.h"
#include "dm3d_h3.h"
#include <type_traits>

namespace h3v2 {

constexpr int REC = DM3D_REC;
typedef float f32x4v __attribute__((ext_vector_type(4)));

// LDS position (within its group of 16) of the weight row read by MFMA column c
__host__ __device__ constexpr int pi_pos(int c) {
    return c < 4 ? (c < 2 ? c : c + 2) : (c >= 12 ? (c < 14 ? c - 4 : c - 2) : (((c - 4) >> 1) * 4 + 2 + ((c - 4) & 1)));
}
// patch column of MFMA row i (rows 0-3 -> 0, 4-7 -> 2, 8-11 -> 3, 12-15 -> 1); patch row is i & 3
__device__ __forceinline__ int dx_of_row(int i) { return (0x1320 >> ((i >> 2) * 4)) & 3; }

// The three passes (al.bh, ah.bl, ah.bh) of the 4 x 2 accumulator tiles one weight batch feeds, tile-major.
#define DM3D_MFMA3_TILES(acc, nb, al, ah, bl, bh)                                                        \
    _Pragma("unroll") for (int pi_ = 0; pi_ < 4; ++pi_)                                                  \
        _Pragma("unroll") for (int k_ = 0; k_ < 2; ++k_) {                                               \
            f32x4v& c_ = acc[pi_][(nb) * 2 + k_];                                                        \
            c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[pi_], bh[k_], c_, 0, 0, 0);                   \
            c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[pi_], bl[k_], c_, 0, 0, 0);                   \
            c_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[pi_], bh[k_], c_, 0, 0, 0);                   \
        }

// geometry of one workgroup's brick, as both kernels compute it
struct Brick {
    int b, oz0, oy0, ox0;            // sample, first output voxel
    int ooz, ooy, oox;               // output offset in the full tensor (parity bits)
    int ntile, khalf;                // 64-column tile, split-K part
};

// LDS bytes the skip phase needs (it overlays the main loop's images once every wave has left them)
template <int TD> constexpr int skip_lds_halfs() { return 2 * TD * 8 * 12 * REC + 3 * 2 * 64 * REC; }

template <int TD>
__device__ __forceinline__ void skip_phase(const ConvArgs& p, _Float16* smem, f32x4v (&acc)[4][4], const Brick& br) {
    constexpr int TH = 8, TW = 8, CK = 16, NT = 64, NTHR = TD * 64, HWP = 12, KS = 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, q = (lane >> 4) & 1, row = lane & 15, piece = tid & 1;
    const int b = br.b, oz0 = br.oz0, oy0 = br.oy0, ox0 = br.ox0, ntile = br.ntile, khalf = br.khalf;
    const int b_pos = pi_pos(row);
    const int b_hi = (half * NT + b_pos) * REC + ((q ^ swz(b_pos)) << 3);
    // ---- fused 1x1 conv over a second, raw input (ResidualBlock: out = conv2(...) + Conv3D(width, 1)(x), conditional_dm3d.py:243-248,
    // 268).  K = 32 per MFMA = two 16-channel chunks of the SAME voxel instead of two taps: chunk 2i goes to LDS region 0, chunk
    // 2i+1 to region 1 (brick voxels only, rows padded to 12 records like the halo so the patch reads stay conflict-free), the
    // lane half picks the region.  One pair of chunks = one barrier pair + 48 MFMAs per wave; the next two pairs' weights (8 KB by
    // LDS-DMA) and voxels (registers) are in flight meanwhile.
    // A split-K launch spreads the pairs over its parts as it spreads the main loop's chunks (the skip sum is linear like them).
    const int s_lo = p.s_npairs * khalf / p.ksplit, s_hi = p.s_npairs * (khalf + 1) / p.ksplit;
    if constexpr (KS == 3) if (s_hi > s_lo) {                      // (the launcher admits a skip conv behind k3 / stride 1 only)
        constexpr int SREC = TD * TH * HWP;                                // 384 records per region
        _Float16* lds_sa = smem;                                        // [2][SREC][REC]            (0 .. 48 KB)
        _Float16* lds_sw = smem + 2 * SREC * REC;                       // [3 buffers][2][NT][REC]   (48 .. 72 KB)
        constexpr int SITEMS = TD * TH * TW * 2;                          // 16-byte pieces per region (= 2 * NTHR: four per thread in all)
        int sgv[4], sst[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int item = tid + j * NTHR, t = item / SITEMS, iv = (item % SITEMS) >> 1;
            const int z = iv >> 6, y = (iv >> 3) & 7, x = iv & 7;
            const bool in = oz0 + z < p.ind && oy0 + y < p.inh && ox0 + x < p.inw;
            sgv[j] = in ? ((b * p.ind + oz0 + z) * p.inh + oy0 + y) * p.inw + ox0 + x : -1;
            const int v = (z * TH + y) * HWP + x;
            sst[j] = (t * SREC + v) * REC + ((piece ^ swz(v)) << 3);       // piece = tid & 1 = item & 1
        }
        // Two register sets and three weight buffers: while pair pp is multiplied, the voxels and weights of pairs pp+1 AND pp+2 are in
        // flight.  (Round 2 kept one pair of lookahead: a pair is 48 MFMAs, ~0.4 us, and every pair waited out most of a memory round trip
        // behind vmcnt(0) — 2 us per pair at small batch, where no other workgroup fills the gap: profiles/r03_layers_B4.log.)  The loop runs
        // whole rounds of two steps without a branch around its loads; a step past the last pair re-reads the last weights against zeros.
        f32x4 sr0[2][4], sr1[2][4];
        bool sok0[2][4], sok1[2][4];
        const int np = s_hi;                                               // this part's pairs: s_lo .. np-1
        auto sload = [&](auto SET, int pp) {
            constexpr int S = decltype(SET)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c0 = (pp * 2 + (tid + j * NTHR) / SITEMS) * CK;
                const float* src;
                int ldc, cb;
                if (c0 < p.sc1) { src = p.sx1; ldc = p.sc1; cb = c0; } else { src = p.sx2; ldc = p.sc2; cb = c0 - p.sc1; }
                const int cpos = cb + piece * 8;
                const bool real = src != nullptr && cb < ldc && pp < np;    // a pad chunk past the last channel (or pair) reads zeros
                sok0[S][j] = real && cpos < ldc;
                sok1[S][j] = real && cpos + 4 < ldc;
                const float* qp = (real ? src : p.sx1) + (size_t)(sgv[j] >= 0 ? sgv[j] : 0) * (real ? ldc : p.sc1);
                sr0[S][j] = *reinterpret_cast<const f32x4*>(qp + (sok0[S][j] ? cpos : 0));
                sr1[S][j] = *reinterpret_cast<const f32x4*>(qp + (sok1[S][j] ? cpos + 4 : 0));
            }
        };
        constexpr int WS = 8 / TD;                                         // 1 KB DMA pieces per wave and pair
        // (buffer form of the LDS-DMA, as the main loop's weights: a FLAT-encoded global_load_lds in flight makes hipcc's wait-count pass treat
        // both counters as out of order and guard every load / LDS-read result with a wait for zero)
        const auto sw_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(static_cast<const void*>(p.swpk)), (short)0, (int)((long)(p.coutpad / NT) * p.s_npairs * (2 * NT * REC * 2)), 0x00020000);
        const int sw_voff = wave * 1024 + lane * 16, wave_u = __builtin_amdgcn_readfirstlane(wave);
        const unsigned sw_tile = (unsigned)__builtin_amdgcn_readfirstlane(ntile * p.s_npairs * (2 * NT * REC * 2));
        auto sdma = [&](int pp, int buf) {                                 // 8 KB per pair (past the end: the last pair again)
            const unsigned off = sw_tile + (unsigned)__builtin_amdgcn_readfirstlane((pp < np ? pp : np - 1) * (2 * NT * REC * 2));
            char* dst = reinterpret_cast<char*>(lds_sw) + __builtin_amdgcn_readfirstlane(buf) * (2 * NT * REC * 2) + wave_u * 1024;
#pragma unroll
            for (int i = 0; i < WS; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(sw_rsrc, (__attribute__((address_space(3))) void*)(dst + i * (TD * 1024)), 16, sw_voff,
                                                         (int)(off + i * (TD * 1024)), 0, 0);
        };
        const int sa_rec = (wave * TH + (row & 3)) * HWP + dx_of_row(row) + half * SREC;
        const int sb_hi = b_hi;                                            // same [2 taps][NT][REC] row layout as a main weight pair
        // vmcnt waits as builtins hipcc's wait-count pass sees (behind an LDS-DMA it guards every load result with vmcnt(0) of its own
        // otherwise).  simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14
#define DM3D_WAIT_VM(n) __builtin_amdgcn_s_waitcnt((((n) & 15) | (7 << 4) | (15 << 8) | ((((n) >> 4) & 3) << 14)))
        const std::integral_constant<int, 0> S0;
        const std::integral_constant<int, 1> S1;
        // every wave has left the main loop's LDS images (the weight buffers and regions below overlay them) and nothing is in flight
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        sdma(s_lo, 0);
        sload(S0, s_lo);
        sdma(s_lo + 1, 1);
        sload(S1, s_lo + 1);
        int wbuf_i = 0;                                                    // weight buffer of the pair being multiplied (pp mod 3)
        auto step = [&](auto SET, int pp) {
            constexpr int S = decltype(SET)::value;
            // this pair's voxels and weights were requested two steps ago (the first two: in front of the loop); newer: the other set's
            // 8 loads and one pair of weights.  The same count in every step: a branch here makes hipcc merge the two states into vmcnt(0).
            DM3D_WAIT_VM(8 + WS);
#pragma unroll
            for (int j = 0; j < 4; ++j) {                                  // in place: the set's registers become the two 16-byte pieces
                h8 hi_, lo_;
                split8(sr0[S][j], sr1[S][j], (sgv[j] >= 0 && sok0[S][j]) ? 65504.0f : 0.0f, (sgv[j] >= 0 && sok1[S][j]) ? 65504.0f : 0.0f, hi_, lo_);
                sr0[S][j] = __builtin_bit_cast(f32x4, hi_);
                sr1[S][j] = __builtin_bit_cast(f32x4, lo_);
            }
            // raw barriers behind this wave's own LDS traffic: __syncthreads() would also drain the vector-memory counter, i.e. wait for
            // the requests that are meant to stay in flight across it
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                  // everyone has left the previous LDS image
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<f32x4*>(lds_sa + sst[j]) = sr0[S][j];
                *reinterpret_cast<f32x4*>(lds_sa + (sst[j] ^ 16)) = sr1[S][j];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                  // the image and this pair's weights are visible
            asm volatile("" ::: "memory");
            {
                const int b2 = wbuf_i + 2 >= 3 ? wbuf_i - 1 : wbuf_i + 2;  // the buffer pair pp-1 left (every wave is past it: the barrier above)
                sdma(pp + 2, b2);
                sload(SET, pp + 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            const _Float16* wbuf = lds_sw + wbuf_i * (2 * NT * REC);
            const int v0 = sa_rec, v1 = sa_rec + 4;
            const int o0 = v0 * REC + ((q ^ swz(v0)) << 3);
            const int o1 = v1 * REC + ((q ^ swz(v1)) << 3);
            h8 ah[4], al[4];
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                ah[py * 2 + 0] = *reinterpret_cast<const h8*>(lds_sa + o0 + py * (48 * REC));
                al[py * 2 + 0] = *reinterpret_cast<const h8*>(lds_sa + (o0 ^ 16) + py * (48 * REC));
                ah[py * 2 + 1] = *reinterpret_cast<const h8*>(lds_sa + o1 + py * (48 * REC));
                al[py * 2 + 1] = *reinterpret_cast<const h8*>(lds_sa + (o1 ^ 16) + py * (48 * REC));
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                h8 bh[2], bl[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    bh[k] = *reinterpret_cast<const h8*>(wbuf + ((nb * 2 + k) * 16) * REC + sb_hi);
                    bl[k] = *reinterpret_cast<const h8*>(wbuf + ((nb * 2 + k) * 16) * REC + (sb_hi ^ 16));
                }
                DM3D_MFMA3_TILES(acc, nb, al, ah, bl, bh);
            }
            __builtin_amdgcn_sched_barrier(0);
            wbuf_i = wbuf_i + 1 == 3 ? 0 : wbuf_i + 1;
        };
        for (int pp = s_lo; pp < np; pp += 2) {
            step(S0, pp);
            step(S1, pp + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the tail requests
#undef DM3D_WAIT_VM
    }

}

// ---- Cin split, hand-over form (round 5; replaces zero fill + atomic adds and scratch + reduce launch).  A launch whose grid would leave
// most of the chip idle runs P workgroups per tile (brick x 64 columns), each contracting 1/P of the chunks.  Every part stores its RAW
// accumulator tiles (MFMA layout, 16 bytes per lane and piece: 1 KB per wave and instruction) into the caller's scratch, write-through
// (sc1: the bytes leave the XCD's L2, no release fence needed), waits for them (vmcnt(0), every storing wave), meets at a workgroup
// barrier, and ONE lane draws a ticket from the tile's counter (relaxed agent-scope add).  The part that draws P - 1 knows all others
// have stored: it sums the parts IN PART ORDER (its own tiles, still in registers, at their place; the others by sc1 loads, which bypass
// this CU's L1) — the result does not depend on which part came last —, zeroes the counter for the next launch and runs the ordinary
// epilogue.  Nobody waits for anybody: a part that is not last is done.  (MI355X_MICROARCH.md, inter-workgroup visibility, the unsharded
// counter row; cdna_hip_programming.md, in-launch split-K reduction.)  scratch layout: [tile][part][piece][256 threads] x 16 bytes.
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
struct SplitTile { __amdgpu_buffer_rsrc_t rs; unsigned lane_off, part_off; };
__device__ __forceinline__ SplitTile split_tile(const ConvArgs& p, const long tile, const int part) {
    SplitTile t;
    char* base = static_cast<char*>(p.scratch) + (size_t)tile * p.ksplit * p.split_tile_floats * 4;      // (uniform)
    t.rs = __builtin_amdgcn_make_buffer_rsrc(base, (short)0, (int)(p.ksplit * p.split_tile_floats * 4), 0x00020000);
    // (from an opaque copy of the thread number: a lane constant hipcc can compute in front of the caller's main loop is one more register
    // alive across it — in the Winograd kernel, whose step loop has none to spare, a spill reloaded behind a vmcnt(0) at every chunk)
    int tx = threadIdx.x;
    asm volatile("" : "+v"(tx));
    t.lane_off = (unsigned)tx * 16u;
    t.part_off = (unsigned)(part * p.split_tile_floats * 4);
    return t;
}
// pieces piece0 .. piece0 + A * B - 1 of this part's image
template <int A, int B>
__device__ __forceinline__ void split_store(const SplitTile& t, const f32x4v (&v)[A][B], const int piece0) {
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int b = 0; b < B; ++b)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v[a][b]), t.rs, t.lane_off + (unsigned)(piece0 + a * B + b) * 4096u, (int)t.part_off, 16);   // aux 16 = sc1
}
// true in the part that drew the tile's last ticket (uniform over the workgroup); word: 4 bytes of LDS nobody else uses right now
__device__ __forceinline__ bool split_is_last(const ConvArgs& p, const long tile, unsigned* word) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave: its stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned tk = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(p.split_counters) + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tk == (unsigned)p.ksplit - 1u)                       // nobody touches the word again in this launch: the next one finds zero
            __hip_atomic_store(reinterpret_cast<unsigned*>(p.split_counters) + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *word = tk;
    }
    __syncthreads();
    const bool last = *word == (unsigned)p.ksplit - 1u;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (no instruction: keeps the compiler from moving the loads above the ticket)
    return last;
}
// v = sum over the parts, in part order, of pieces piece0 ..; on entry v holds this part's own tiles
template <int A, int B>
__device__ __forceinline__ void split_gather(const SplitTile& t, const ConvArgs& p, const int part, f32x4v (&v)[A][B], const int piece0) {
    f32x4v own[A][B];
#pragma unroll
    for (int a = 0; a < A; ++a)
#pragma unroll
        for (int b = 0; b < B; ++b) own[a][b] = v[a][b];
    for (int q = 0; q < p.ksplit; ++q) {                         // (uniform)
        if (q == part) {
            if (q > 0) {
#pragma unroll
                for (int a = 0; a < A; ++a)
#pragma unroll
                    for (int b = 0; b < B; ++b) v[a][b] += own[a][b];
            }
            continue;
        }
        const int qoff = (int)(q * p.split_tile_floats * 4);
        u32x4v x[A][B];
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int b = 0; b < B; ++b) x[a][b] = __builtin_amdgcn_raw_buffer_load_b128(t.rs, t.lane_off + (unsigned)(piece0 + a * B + b) * 4096u, qoff, 16);
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const f32x4v y = __builtin_bit_cast(f32x4v, x[a][b]);
                v[a][b] = q == 0 ? y : v[a][b] + y;
            }
    }
}

// NCT = 16-column tiles per wave the caller computed (4: a whole 64-column tile; 1 / 2: the narrow forms for Cout <= 16 / 32)
// zs: the z-slice of the brick these tiles belong to (-1: the wave index — one slice per wave); tile pi sits at brick column
// xa * (pi & 1) + xb (xb < 0: the direct kernels' patch geometry, 4 * (pi & 1) + dx_of_row; the Winograd form passes 1, 2 * x-pair)
// gn: per-lane partial sums for the fused GroupNormalization statistics (ConvArgs.gn_stats; nullptr: none): [ni][j][sum, sum of squares] of
// the values this lane stores for its four channels of column tile ni — only the 16-byte full-brick form accumulates (the launcher clears
// gn_stats for every other form and runs the stand-alone statistics kernel behind the launch); gn_flush() adds them to the tensor's buffer.
template <int TD, int NCT = 4>
__device__ __forceinline__ void epilogue(const ConvArgs& p, f32x4v (&acc)[4][NCT], const Brick& br, const int zs = -1, const int xa = 4, const int xb = -1,
                                         float* gn = nullptr) {
    constexpr int TH = 8, TW = 8, NT = 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = lane & 15, g4 = lane >> 4;
    const int b = br.b, oz0 = br.oz0, oy0 = br.oy0, ox0 = br.ox0, ooz = br.ooz, ooy = br.ooy, oox = br.oox, ntile = br.ntile;
    // ---- epilogue.  Accumulator register r of tile (patch pi, column tile ni): voxel (dy = 4*(pi>>1) + r, dx = 4*(pi&1) +
    // dx_of_row(4*g4)), output channel ni*16 + row.
    const int vrow = p.vec ? (p.vec_idx ? p.vec_idx[b] : b) : 0;
    const int n0 = ntile * NT;
    const bool full = oz0 + TD <= p.od && oy0 + TH <= p.oh && ox0 + TW <= p.ow && n0 + NT <= p.cout;
    const int oz = oz0 + (zs < 0 ? wave : zs);
    const bool z_ok = oz < p.od;
    const size_t zbase = (((size_t)b * p.fd + (z_ok ? oz * p.os + ooz : 0)) * p.fh) * p.fw * p.cout;
    float* outz = p.out + zbase;
    // (a Cin-split launch runs this once per tile, in the part that drew the last ticket, on the summed tiles: split_* above)
    const float* resz = p.res ? p.res + zbase : nullptr;
    const float* prz = p.prelu ? p.prelu + (zbase - (size_t)b * p.fd * p.fh * p.fw * p.cout) : nullptr;
    const int dxl = xb < 0 ? dx_of_row(4 * g4) : xb;
    const int ystep = p.os * p.fw * p.cout;                              // one brick row further in the output
    float amax = 0.0f;                                                   // range guard: largest |value| this lane stores
    const float rlim = p.range_limit;
#ifndef DM3D_EPILOGUE_SCALAR
    if (full && p.epi_vec4 && !prz) {
        // Full brick, plain stores, aligned operands, no PReLU (the common case; the autoencoder's PReLU convs take the scalar form below: a
        // per-tile slope load under a uniform `if` left an unconditional vmcnt(0) behind it, and with it every tile waited for the previous
        // tile's STORE to complete — 600-800 cycles per tile for every conv of the U-Net, in-kernel stamps of round 3).  The MFMA leaves a lane with ONE channel of FOUR voxels (r = 0..3: brick rows); stored
        // like that every access is 4 bytes per lane — 64 loads + 64 stores per lane with a residual, and the epilogue of a 64 -> 64 conv
        // took 17-25 thousand cycles (in-kernel stamps), bound by the number of memory instructions, not by bytes.  A 4 x 4 transpose
        // inside each quad of lanes (two DPP exchange rounds, 16 VALU per tile) gives a lane FOUR consecutive channels of ONE voxel
        // (row k = n & 3, channels (n & ~3) .. +3): 16-byte accesses, a quarter of the instructions.  Per element the arithmetic and its
        // order are those of the scalar path below (-DDM3D_EPILOGUE_SCALAR), so the results are bit-identical.
        const int k = row & 3, c4 = row & ~3;
        const bool b0 = (row & 1) != 0, b1 = (row & 2) != 0;
        auto xor1 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false)); };
        auto xor2 = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false)); };
        f32x4 rv4[NCT][4];
        if (resz) {                                     // all residual pieces requested before the first one is used
#pragma unroll
            for (int ni = 0; ni < NCT; ++ni)
#pragma unroll
                for (int pi = 0; pi < 4; ++pi)
                    rv4[ni][pi] = *reinterpret_cast<const f32x4*>(resz + (((oy0 + 4 * (pi >> 1) + k) * p.os + ooy) * p.fw
                                                                          + (ox0 + xa * (pi & 1) + dxl) * p.os + oox) * p.cout + n0 + ni * 16 + c4);
        }
        // Every load of this path comes before its first store: vmcnt counts loads and stores together, in order, so a bias / vector load
        // behind the stores of the previous column tile waited out their whole write latency — four serialised round trips per slice,
        // ~20 000 cycles of a 43 000-cycle epilogue where no other wave hides them (in-kernel stamps of the Winograd form, round 3).
        f32x4 add_t[NCT], ps_t[NCT], pt_t[NCT];
#pragma unroll
        for (int ni = 0; ni < NCT; ++ni) {
            const int n = n0 + ni * 16 + c4;
            const f32x4 one = {1.f, 1.f, 1.f, 1.f}, zero = {0.f, 0.f, 0.f, 0.f};
            add_t[ni] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : zero;
            if (p.vec) {
                const f32x4 vv = *reinterpret_cast<const f32x4*>(p.vec + (size_t)vrow * p.vec_ld + n);
#pragma unroll
                for (int j = 0; j < 4; ++j) add_t[ni][j] += vv[j];
            }
            ps_t[ni] = p.post_scale ? *reinterpret_cast<const f32x4*>(p.post_scale + n) : one;
            pt_t[ni] = p.post_scale ? *reinterpret_cast<const f32x4*>(p.post_shift + n) : zero;
        }
        // The two rare options are compile-time forms of the loop: as run-time tests of uniform flags hipcc if-converted them, i.e. every
        // element paid the consumer's SiLU (v_exp_f32, v_rcp_f32) and the DM3D_FMT_H2 split and then selected them away — 164 instructions
        // per tile instead of ~60, 12 500 cycles per slice where no other wave hides them (in-kernel stamps of the Winograd form, round 3).
        auto tiles = [&](auto POST_T, auto H2_T, auto GN_T) {
            constexpr bool POST = decltype(POST_T)::value, H2 = decltype(H2_T)::value, GN = decltype(GN_T)::value;
    #pragma unroll
            for (int ni = 0; ni < NCT; ++ni) {
                const int n = n0 + ni * 16 + c4;             // first of this lane's four channels
                const f32x4 add = add_t[ni], ps = ps_t[ni], pt = pt_t[ni];
                // DM3D_FMT_H2: the hi halves of channels n..n+3 are 8 contiguous bytes of the voxel's record, the lo halves 32 bytes further
                const int h2off = (n >> 4) * 64 + ((n >> 3) & 1) * 16 + (n & 7) * 2 - n * 4;
    #pragma unroll
                for (int pi = 0; pi < 4; ++pi) {
                    float a[4] = {acc[pi][ni][0], acc[pi][ni][1], acc[pi][ni][2], acc[pi][ni][3]};
                    {   // quad transpose: a[j] of lane k  <-  a[k] of lane j
                        float s0 = b0 ? a[0] : a[1], s1 = b0 ? a[2] : a[3];
                        float r0 = xor1(s0), r1 = xor1(s1);
                        if (b0) { a[0] = r0; a[2] = r1; } else { a[1] = r0; a[3] = r1; }
                        s0 = b1 ? a[0] : a[2]; s1 = b1 ? a[1] : a[3];
                        r0 = xor2(s0); r1 = xor2(s1);
                        if (b1) { a[0] = r0; a[1] = r1; } else { a[2] = r0; a[3] = r1; }
                    }
                    const int o = (((oy0 + 4 * (pi >> 1) + k) * p.os + ooy) * p.fw + (ox0 + xa * (pi & 1) + dxl) * p.os + oox) * p.cout + n;
                    f32x4 v4;
    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float v = fmaf(a[j], p.out_scale, add[j]);
                        if (p.relu) v = fmaxf(v, 0.0f);
                        if (resz) v += rv4[ni][pi][j];
                        if (p.relu_out) v = fmaxf(v, 0.0f);
                        if constexpr (POST) v = dm3d_silu(fmaf(v, ps[j], pt[j]));         // the consumer's norm + SiLU, applied once here
                        DM3D_AMAX(amax, v);
                        v4[j] = v;
                        if constexpr (GN) { gn[(ni * 4 + j) * 2] += v; gn[(ni * 4 + j) * 2 + 1] = fmaf(v, v, gn[(ni * 4 + j) * 2 + 1]); }
                    }
                    if constexpr (H2) {
                        const unsigned int w0 = split1_bits(v4[0]), w1 = split1_bits(v4[1]), w2 = split1_bits(v4[2]), w3 = split1_bits(v4[3]);
                        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                        char* dst = reinterpret_cast<char*>(outz) + (size_t)o * 4 + h2off;
                        *reinterpret_cast<u32x2*>(dst) = u32x2{(w0 & 0xffffu) | (w1 << 16), (w2 & 0xffffu) | (w3 << 16)};
                        *reinterpret_cast<u32x2*>(dst + 32) = u32x2{(w0 >> 16) | (w1 & 0xffff0000u), (w2 >> 16) | (w3 & 0xffff0000u)};
                    } else {
                        *reinterpret_cast<f32x4*>(outz + o) = v4;
                    }
                }
            }
        };
        const std::true_type yes_t;
        const std::false_type no_t;
        if (p.post_scale) { if (p.out_h2) tiles(yes_t, yes_t, no_t); else tiles(yes_t, no_t, no_t); }
        else if (p.out_h2) tiles(no_t, yes_t, no_t);
        else if (gn) tiles(no_t, no_t, yes_t);          // (the launcher admits gn_stats with a plain float32 output only)
        else tiles(no_t, no_t, no_t);
        if (p.range_flag && amax > rlim) *p.range_flag = 1;
        return;
    }
#endif
    // The slower forms below start from opaque copies of the lane's channel and column: without them hipcc computes THEIR per-element
    // 64-bit addresses (a thousand instructions, a hundred scratch stores) in front of the branch, i.e. also on the way into the 16-byte
    // form above — 12 500 cycles at the head of every epilogue (in-kernel stamps of the Winograd form, round 3: 15 900 -> 3 400).
    int rowq = row, dxq = dxl;
    asm volatile("" : "+v"(rowq), "+v"(dxq));
    if (full) {
        // full brick, scalar form (unaligned operands, PReLU; -DDM3D_EPILOGUE_SCALAR: the A/B arm of the form above):
        // all 64 residual values of this lane are requested before the first one is used
        float rv[NCT][4][4];
        if (resz) {
#pragma unroll
            for (int ni = 0; ni < NCT; ++ni)
#pragma unroll
                for (int pi = 0; pi < 4; ++pi)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        rv[ni][pi][r] = resz[((oy0 + 4 * (pi >> 1)) * p.os + ooy) * p.fw * p.cout
                                             + ((ox0 + xa * (pi & 1) + dxq) * p.os + oox) * p.cout + n0 + ni * 16 + rowq + r * ystep];
        }
#pragma unroll
        for (int ni = 0; ni < NCT; ++ni) {
            const int n = n0 + ni * 16 + rowq;
            float add = p.bias ? p.bias[n] : 0.0f;
            if (p.vec) add += p.vec[(size_t)vrow * p.vec_ld + n];
            const float ps = p.post_scale ? p.post_scale[n] : 1.0f, pt = p.post_scale ? p.post_shift[n] : 0.0f;
            // DM3D_FMT_H2 position of channel n inside its voxel's row (see dm3d_gemm_h3.hip): lanes n and n^1 exchange halves
            const int h2col = (n >> 4) * 64 + ((n >> 3) & 1) * 16 + ((n & 7) >> 1) * 4 + (n & 1) * 32 - n * 4;
#pragma unroll
            for (int pi = 0; pi < 4; ++pi) {
                const int base = (((oy0 + 4 * (pi >> 1)) * p.os + ooy) * p.fw + (ox0 + xa * (pi & 1) + dxq) * p.os + oox) * p.cout + n;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = base + r * ystep;
                    float v = fmaf(acc[pi][ni][r], p.out_scale, add);
                    if (p.relu) v = fmaxf(v, 0.0f);
                    if (prz) { const float al = prz[o]; v = v > 0.0f ? v : al * v; }
                    if (resz) v += rv[ni][pi][r];
                    if (p.relu_out) v = fmaxf(v, 0.0f);
                    if (p.post_scale) v = dm3d_silu(fmaf(v, ps, pt));                 // the consumer's norm + SiLU, applied once here
                    DM3D_AMAX(amax, v);
                    if (p.out_h2) {
                        const unsigned int mine = split1_bits(v);
                        const unsigned int oth = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xf, 0xf, false);   // lane ^ 1
                        const unsigned int word = (n & 1) ? ((oth >> 16) | (mine & 0xffff0000u)) : ((mine & 0xffffu) | (oth << 16));
                        *reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(outz) + (size_t)o * 4 + h2col) = word;
                    } else {
                        outz[o] = v;
                    }
                }
            }
        }
        if (p.range_flag && amax > rlim) *p.range_flag = 1;
        return;
    }
#pragma unroll
    for (int ni = 0; ni < NCT; ++ni) {
        const int n = n0 + ni * 16 + rowq;
        const bool n_ok = n < p.cout;
        const int nc = n_ok ? n : p.cout - 1;
        float add = p.bias ? p.bias[nc] : 0.0f;
        if (p.vec) add += p.vec[(size_t)vrow * p.vec_ld + nc];
#pragma unroll
        for (int pi = 0; pi < 4; ++pi) {
            const int oyb = oy0 + 4 * (pi >> 1), ox = ox0 + xa * (pi & 1) + dxq;
            const int base = ((oyb * p.os + ooy) * p.fw + ox * p.os + oox) * p.cout + nc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = n_ok && z_ok && oyb + r < p.oh && ox < p.ow;
                const int o = ok ? base + r * ystep : 0;
                float v = fmaf(acc[pi][ni][r], p.out_scale, add);
                if (p.relu) v = fmaxf(v, 0.0f);
                if (prz) { const float al = prz[o]; v = v > 0.0f ? v : al * v; }
                if (resz) v += resz[o];
                if (p.relu_out) v = fmaxf(v, 0.0f);
                if (ok) { DM3D_AMAX(amax, v); outz[o] = v; }
            }
        }
    }
    if (p.range_flag && amax > rlim) *p.range_flag = 1;
}

// Fused GroupNormalization statistics, second half: the per-lane partial sums of epilogue() — this lane's four channels of every column tile —
// are summed over the 16 lanes that hold the same channels (lane bits 0-1: the voxel row inside a quad, bits 4-5: the voxel column group) and
// STORED as this wave's partial (sum, sum of squares) per channel: gn_stats[b][slot][cout][2] float32, slot = the wave's z-slice of the
// brick grid (parity forms: one set of slots per parity) — no atomics (a first form added float64 atomics to one [b][cout][2] buffer: a
// million device-scope atomics per launch cost 13 % of the conv), and the sum dm3d_groupnorm_finalize2 takes over the slots has a fixed order.
// `slices`: z-slices this wave covered (the Winograd form: 2 — the second slice's slot is written as zeros).  One call per wave and brick.
template <int TD, int NCT>
__device__ __forceinline__ void gn_flush(const ConvArgs& p, float (&gn)[NCT * 8], const Brick& br, const int zs, const int slices) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NCT * 8; ++i) {
        float v = gn[i];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));   // lane ^ 1
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));   // lane ^ 2
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        gn[i] = v;
    }
    if ((lane & 0x33) == 0) {
        const int c4 = lane & 12;
        const int bh = p.oh / 8, bw = p.ow / 8, par = br.ooz * 4 + br.ooy * 2 + br.oox;
        const long nslots = (long)p.od * bh * bw * (p.os == 2 ? 8 : 1);
        const long slot = (((long)par * p.od + br.oz0 + zs) * bh + br.oy0 / 8) * bw + br.ox0 / 8;
        float* dst = p.gn_stats + (((size_t)br.b * nslots + slot) * p.cout + br.ntile * 64 + c4) * 2;
#pragma unroll
        for (int ni = 0; ni < NCT; ++ni) {
            *reinterpret_cast<f32x4*>(dst + ni * 32) = f32x4{gn[ni * 8], gn[ni * 8 + 1], gn[ni * 8 + 2], gn[ni * 8 + 3]};
            *reinterpret_cast<f32x4*>(dst + ni * 32 + 4) = f32x4{gn[ni * 8 + 4], gn[ni * 8 + 5], gn[ni * 8 + 6], gn[ni * 8 + 7]};
        }
        for (int k = 1; k < slices; ++k) {                      // slots of further slices this wave covered in the same sums
            float* dz = dst + (size_t)k * bh * bw * p.cout * 2;
#pragma unroll
            for (int ni = 0; ni < NCT; ++ni) {
                *reinterpret_cast<f32x4*>(dz + ni * 32) = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(dz + ni * 32 + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
}

// ---- The Winograd kernel's epilogue (round 5): the CHANNEL-QUAD column mapping.  Which output channel an MFMA column computes is the weight
// packer's choice.  The direct kernel's image puts channel 16 ni + j into column j of column tile ni, so a lane (column j) holds ONE channel of
// four voxels per tile and the 16-byte form above first transposes every tile across a quad of lanes.  The Winograd image
// (dm3d_pack_weights_h3w, and the fused skip conv's operand fragments) puts channel 4 j + ni there instead: the four column tiles of a lane
// are four CONSECUTIVE channels of one voxel — register r of tiles (t, 0..3) is a 16-byte piece as it stands.  No transpose (16 of ~60
// instructions per tile), and the sixteen lanes of a row group write 256 contiguous bytes of a voxel: a store instruction covers 8 whole
// cache lines instead of 16 half lines (timing-only bound of fully coalesced stores: profiles/r05_ab_store_coalescing.log).
// e[2 g + parity][ni][r]: voxel (y = 4 g + r, x = 2 * x-pair + parity) of z-slice zs, channel 64 ntile + 4 j + ni; lane = 16 * g + j with x-pair = g ^ (g >> 1).
// Whole 8 x 8 x 8 bricks only (dm3d_conv_h3w_serves), stride-1 outputs.  gn: this lane's partial (sum, sum of squares) of its four
// channels, [ni][2] (gn_flush_cq below).
// The per-channel operands of epilogue_cq's 16-byte form — bias + vector row, post-norm scale / shift of this lane's four channels — are
// the same for every z-slice of a work item: loaded ONCE per item (epilogue_cq_vecs, in front of the first slice).  Loaded per slice, the
// second slice's requests queued behind the first slice's stores (vmcnt counts loads and stores together, in order) and waited out their
// whole write latency.  vrow: the row of the vector tensor (dm3d_conv_desc.vec_idx), which the caller reads early (a dependent scalar load).
struct EpiVecs { f32x4 add, ps, pt; };
__device__ __forceinline__ EpiVecs epilogue_cq_vecs(const ConvArgs& p, const Brick& br, const int vrow) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int n = br.ntile * 64 + (tid & 15) * 4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f}, one = {1.f, 1.f, 1.f, 1.f};
    EpiVecs ev;
    ev.add = zero; ev.ps = one; ev.pt = zero;
    if (br.ntile * 64 + 64 <= p.cout && p.epi_vec4) {            // (the general form loads element by element itself)
        if (p.bias) ev.add = *reinterpret_cast<const f32x4*>(p.bias + n);
        if (p.vec) {
            const f32x4 vv = *reinterpret_cast<const f32x4*>(p.vec + (size_t)vrow * p.vec_ld + n);
#pragma unroll
            for (int c = 0; c < 4; ++c) ev.add[c] += vv[c];
        }
        if (p.post_scale) { ev.ps = *reinterpret_cast<const f32x4*>(p.post_scale + n); ev.pt = *reinterpret_cast<const f32x4*>(p.post_shift + n); }
    }
    return ev;
}

template <int TD>
__device__ __forceinline__ void epilogue_cq(const ConvArgs& p, f32x4v (&e)[4][4], const Brick& br, const int zs, const EpiVecs& ev, const int vrow, float* gn = nullptr) {
    constexpr int NT = 64;
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                       // (lane constants made here, not alive — spilled — across the caller's main loop)
    const int lane = tid & 63, j4 = (lane & 15) * 4, xp = (lane >> 4) ^ (lane >> 5);       // (accumulator row group g holds x-pair g ^ (g >> 1): the Winograd kernel's fragment order)
    const int n0 = br.ntile * NT, n = n0 + j4;           // this lane's first channel
    const size_t zbase = (((size_t)br.b * p.fd + br.oz0 + zs) * p.fh) * p.fw * p.cout;
    float* const outz = p.out + zbase;
    const float* const resz = p.res ? p.res + zbase : nullptr;
    const float* const prz = p.prelu ? p.prelu + (zbase - (size_t)br.b * p.fd * p.fh * p.fw * p.cout) : nullptr;
    // element offset of voxel (y = oy0 + 4 g + r, x = ox0 + 2 xp + parity), channel n
    auto voff = [&](const int t, const int r) { return ((br.oy0 + 4 * (t >> 1) + r) * p.fw + br.ox0 + 2 * xp + (t & 1)) * p.cout + n; };
    float amax = 0.0f;
#ifndef DM3D_EPILOGUE_SCALAR
    if (n0 + NT <= p.cout && p.epi_vec4 && !prz) {
        f32x4 rv[4][4];                                  // [tile][row]
        if (resz) {                                      // every load before the first store (vmcnt counts both, in order)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) rv[t][r] = *reinterpret_cast<const f32x4*>(resz + voff(t, r));
        }
        const f32x4 add = ev.add, ps = ev.ps, pt = ev.pt;
        const int h2off = (n >> 4) * 64 + ((n >> 3) & 1) * 16 + (n & 7) * 2 - n * 4;       // DM3D_FMT_H2 position of channels n .. n + 3 inside their voxel's row
        auto tiles = [&](auto POST_T, auto H2_T, auto GN_T) {
            constexpr bool POST = decltype(POST_T)::value, H2 = decltype(H2_T)::value, GN = decltype(GN_T)::value;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    f32x4 v4;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v = fmaf(e[t][c][r], p.out_scale, add[c]);
                        if (p.relu) v = fmaxf(v, 0.0f);
                        if (resz) v += rv[t][r][c];
                        if (p.relu_out) v = fmaxf(v, 0.0f);
                        if constexpr (POST) v = dm3d_silu(fmaf(v, ps[c], pt[c]));
                        DM3D_AMAX(amax, v);
                        v4[c] = v;
                        if constexpr (GN) { gn[c * 2] += v; gn[c * 2 + 1] = fmaf(v, v, gn[c * 2 + 1]); }
                    }
                    const int o = voff(t, r);
                    if constexpr (H2) {
                        const unsigned int w0 = split1_bits(v4[0]), w1 = split1_bits(v4[1]), w2 = split1_bits(v4[2]), w3 = split1_bits(v4[3]);
                        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                        char* dst = reinterpret_cast<char*>(outz) + (size_t)o * 4 + h2off;
                        *reinterpret_cast<u32x2*>(dst) = u32x2{(w0 & 0xffffu) | (w1 << 16), (w2 & 0xffffu) | (w3 << 16)};
                        *reinterpret_cast<u32x2*>(dst + 32) = u32x2{(w0 >> 16) | (w1 & 0xffff0000u), (w2 >> 16) | (w3 & 0xffff0000u)};
                    } else {
                        *reinterpret_cast<f32x4*>(outz + o) = v4;
                    }
                }
        };
        const std::true_type yes_t;
        const std::false_type no_t;
        if (p.post_scale) { if (p.out_h2) tiles(yes_t, yes_t, no_t); else tiles(yes_t, no_t, no_t); }
        else if (p.out_h2) tiles(no_t, yes_t, no_t);
        else if (gn) tiles(no_t, no_t, yes_t);
        else tiles(no_t, no_t, no_t);
        if (p.range_flag && amax > p.range_limit) *p.range_flag = 1;
        return;
    }
#endif
    // the general form: a ragged last column tile, unaligned operands, PReLU (the autoencoders' residual units) — element by element
    int nq = n;
    asm volatile("" : "+v"(nq));                         // (keeps this form's address arithmetic out of the 16-byte form's way)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool n_ok = nq + c < p.cout;
        const int nc = n_ok ? nq + c : p.cout - 1;
        float add = p.bias ? p.bias[nc] : 0.0f;
        if (p.vec) add += p.vec[(size_t)vrow * p.vec_ld + nc];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = n_ok ? voff(t, r) - nq + nc : 0;
                float v = fmaf(e[t][c][r], p.out_scale, add);
                if (p.relu) v = fmaxf(v, 0.0f);
                if (prz) { const float al = prz[o]; v = v > 0.0f ? v : al * v; }
                if (resz) v += resz[o];
                if (p.relu_out) v = fmaxf(v, 0.0f);
                if (n_ok) { DM3D_AMAX(amax, v); outz[o] = v; }
            }
    }
    if (p.range_flag && amax > p.range_limit) *p.range_flag = 1;
}

// The fused GroupNormalization statistics of epilogue_cq: a lane's partial sums of its four channels over the tiles of `slices` z-slices are
// added over the four lanes that hold the same channels (the x-pairs: lane bits 4-5) and stored as the slot of z-slice zs
// (gn_stats[b][slot][cout][2], as gn_flush does; the further slices' slots are written as zeros).
template <int TD>
__device__ __forceinline__ void gn_flush_cq(const ConvArgs& p, float (&gn)[8], const Brick& br, const int zs, const int slices) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v = gn[i];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        gn[i] = v;
    }
    if (lane < 16) {
        const int bh = p.oh / 8, bw = p.ow / 8;
        const long nslots = (long)p.od * bh * bw;
        const long slot = ((long)(br.oz0 + zs) * bh + br.oy0 / 8) * bw + br.ox0 / 8;
        float* dst = p.gn_stats + (((size_t)br.b * nslots + slot) * p.cout + br.ntile * 64 + lane * 4) * 2;
        *reinterpret_cast<f32x4*>(dst) = f32x4{gn[0], gn[1], gn[2], gn[3]};
        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{gn[4], gn[5], gn[6], gn[7]};
        for (int k = 1; k < slices; ++k) {
            float* dz = dst + (size_t)k * bh * bw * p.cout * 2;
            *reinterpret_cast<f32x4*>(dz) = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(dz + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

}  // namespace h3v2
