// dm3d_conv_h3_host.hip — what the 16x16x32 split-float16 Conv3d kernels (dm3d_conv_h3v3.hip: the free-running three-pass kernel;
// dm3d_conv_h3w.hip: its Winograd-x form) share on the host side: brick counts and the Cin split of small grids (the hand-over form: raw
// partial tiles into caller scratch, the last part of a tile to arrive sums them and runs the epilogue — one launch, no zero fill, no
// atomics on the output, no reduce launch), the weight packers of the DM3D_WL_PAIR
// geometry (plain / UpSample parity sums / Conv3DTranspose / the Winograd-x transform) and of the fused skip conv's image.
// reference ops: Conv3D / UpSampling3D + Conv3D / Conv3DTranspose weights in Keras layouts (networks/conditional_dm3d.py:238-296,
// networks/vqvae3d_monai.py:373-377).
#include <cstdlib>
#include "dm3d_conv_h3v2_parts.h"

using namespace h3v2;

// Everything around the conv launch itself that the two kernels share: brick counts, the Cin split of small grids, 16-byte epilogue
// eligibility, where the GroupNormalization statistics come from.
long dm3d_conv_split_tiles(const ConvArgs& a, int td) {
    return (long)a.batch * ((a.od + td - 1) / td) * ((a.oh + 7) / 8) * ((a.ow + 7) / 8) * (a.coutpad / 64) * (a.parity ? 8 : 1);
}

int dm3d_h3v2_pre_launch(ConvArgs& a, int td, H3v2Launch& L, hipStream_t st, int force_ksplit) {
    (void)st;
    a.bd = (a.od + td - 1) / td;
    a.bh = (a.oh + 7) / 8;
    a.bw = (a.ow + 7) / 8;
    // Small grids (the 8^3 level at B = 32 has 64 bricks x 4 channel tiles = one workgroup per CU, i.e. one wave per SIMD with
    // nothing to hide its barriers and LDS latency behind; at B = 1 that level has 8 workgroups for 256 CUs) split the Cin chunks over
    // several workgroups per tile.  Round 5: the parts meet INSIDE the launch (split_* in dm3d_conv_h3v2_parts.h) — every epilogue form,
    // the fused output formats and the fused statistics included, since one workgroup sees the finished sums.
    a.ksplit = force_ksplit > 0 ? force_ksplit : (td != 4 ? 1 : dm3d_conv_h3v2_ksplit(a));
    {
        auto al16 = [](const void* q) { return (reinterpret_cast<size_t>(q) & 15) == 0; };
        a.epi_vec4 = a.cout % 4 == 0 && al16(a.out) && al16(a.bias) && al16(a.res) && al16(a.prelu) && al16(a.post_scale) && al16(a.post_shift)
                     && al16(a.vec) && (a.vec == nullptr || a.vec_ld % 4 == 0);
    }
    a.split_tile_floats = 0;
    if (a.ksplit > 1) {
        const long tiles = dm3d_conv_split_tiles(a, td);
        a.split_tile_floats = (long)td * 8 * 8 * 64;                      // a whole brick x 64 columns, whatever part of it is inside the volume
        DM3D_REQUIRE(a.split_counters && a.split_counter_words >= tiles, "conv: a Cin-split launch of %ld tiles needs split_counters of as many zeroed words (have %d)",
                     tiles, a.split_counter_words);
        DM3D_REQUIRE(a.scratch && (size_t)a.scratch_bytes >= (size_t)tiles * a.ksplit * a.split_tile_floats * sizeof(float),
                     "conv: scratch of %ld bytes is too small for %d parts x %ld tiles (dm3d_conv_scratch_bytes)", a.scratch_bytes, a.ksplit, tiles);
        DM3D_REQUIRE((size_t)a.ksplit * a.split_tile_floats * sizeof(float) < (1ull << 31), "conv: %d parts overflow a tile's buffer descriptor", a.ksplit);
    }
    ConvArgs& k = L.k;
    k = a;
    // Fused GroupNormalization statistics: the kernel's 16-byte full-brick epilogue accumulates them (whole bricks, cout % 64 == 0, plain
    // float32 output without PReLU; a split launch's last part runs that epilogue like any other); every other form leaves them to the
    // stand-alone kernel behind the launch (post_launch).
    L.stats_after = false;
    if (a.gn_stats) {
#ifdef DM3D_EPILOGUE_SCALAR
        const bool fused = false;            // (that A/B build compiles the 16-byte epilogue — the only form that accumulates — out)
#else
        const bool fused = a.od % td == 0 && a.oh % 8 == 0 && a.ow % 8 == 0 && a.cout % 64 == 0 && a.epi_vec4 && !a.prelu
                           && !a.out_h2 && !a.post_scale;
#endif
        if (!fused) { k.gn_stats = nullptr; L.stats_after = true; }
    }
    return DM3D_OK;
}

int dm3d_h3v2_post_launch(const ConvArgs& a, const H3v2Launch& L, hipStream_t st) {
    if (L.stats_after) return dm3d_groupnorm_partials(a.out, a.batch, (int64_t)a.fd * a.fh * a.fw, a.cout, a.gn_stats, st);
    return DM3D_OK;
}

namespace {

// weight image of the v2 kernel: [coutpad/64][cinpad/16][TAPSP][64 positions][REC]; position 16*t16 + PI(c) holds output channel
// 64*ntile + 16*t16 + c; taps >= taps are zero; slots swizzled by the position.  mode: 0 plain, 1 UpSample sums, 2 Conv3DTranspose
__global__ __launch_bounds__(256) void pack_weights_h3v2_kernel(const float* __restrict__ w, int taps, int tapsp, int cin, int cout,
                                                                int nchunks, int ntiles, float scale, const float* in_scale,
                                                                _Float16* __restrict__ out, int mode) {
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
        // column c of 16-column tile pos >> 4: output channel 16 tile + c; the Winograd image (mode 3): 4 c + tile — a lane's four column tiles are
        // four consecutive channels (epilogue_cq, dm3d_conv_h3v2_parts.h)
        const int ci = chunk * 16 + k, co = nt * 64 + (mode == 3 ? 4 * c + (pos >> 4) : (pos & ~15) + c);
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
        r[(((2 + (k >> 3)) ^ sw) << 3) + (k & 7)] = (_Float16)(v - (float)hi);
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

// the same 1x1 kernel as MFMA operand fragments: [coutpad/64][npairs][4 column tiles][hi | lo][64 lanes][8 halfs]; lane (column c = lane & 15,
// k group = lane >> 4: chunk 2 pair + (k group >> 1), channels 8 (k group & 1) .. + 7 of it) holds its 8 consecutive k of output channel
// 64 ntile + 4 c + tile (the Winograd kernel's column mapping) — what the v_mfma_f32_16x16x32_f16 B operand of that lane is
__global__ __launch_bounds__(256) void pack_skip_h3f_kernel(const float* __restrict__ w, int cin, int cout, int npairs, int ntiles,
                                                            float scale, _Float16* __restrict__ out) {
    const long npieces = (long)ntiles * npairs * 4 * 2 * 64;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npieces * 8; i += (long)gridDim.x * 256) {
        const int j = (int)(i & 7);
        long q = i >> 3;
        const int lane = (int)(q & 63); q >>= 6;
        const int hilo = (int)(q & 1); q >>= 1;
        const int ni = (int)(q & 3); q >>= 2;
        const int pair = (int)(q % npairs);
        const int nt = (int)(q / npairs);
        const int kg = lane >> 4;
        const int ci = (pair * 2 + (kg >> 1)) * 16 + (kg & 1) * 8 + j, co = nt * 64 + 4 * (lane & 15) + ni;      // (the Winograd kernel's channel-quad mapping)
        const float v = (ci < cin && co < cout) ? w[(long)ci * cout + co] * scale : 0.0f;
        const _Float16 hi = (_Float16)v;
        out[i] = hilo ? (_Float16)(v - (float)hi) : hi;
    }
}

}  // namespace

int dm3d_pack_skip_h3f(const float* keras_kernel, int cin, int cout, int w_exp, void* packed, hipStream_t st) {
    const int npairs = (int)(dm3d_round_up(cin, 32) / 32), ntiles = (int)(dm3d_round_up(cout, 64) / 64);
    hipLaunchKernelGGL(pack_skip_h3f_kernel, dim3(1024), dim3(256), 0, st, keras_kernel, cin, cout, npairs, ntiles, ldexpf(1.0f, w_exp),
                       static_cast<_Float16*>(packed));
    return dm3d_launch_check("pack_skip_h3f_kernel");
}

int64_t dm3d_h3v2_skip_image_bytes(int cin, int cout) {
    return (int64_t)(dm3d_round_up(cin, 32) / 32) * 2 * dm3d_round_up(cout, 64) * REC * (int64_t)sizeof(_Float16);
}

int dm3d_pack_skip_h3v2(const float* keras_kernel, int cin, int cout, int w_exp, void* packed, hipStream_t st) {
    const int npairs = (int)(dm3d_round_up(cin, 32) / 32), ntiles = (int)(dm3d_round_up(cout, 64) / 64);
    hipLaunchKernelGGL(pack_skip_h3v2_kernel, dim3(1024), dim3(256), 0, st, keras_kernel, cin, cout, npairs, ntiles, ldexpf(1.0f, w_exp),
                       static_cast<_Float16*>(packed));
    return dm3d_launch_check("pack_skip_h3v2_kernel");
}

// Workgroups per tile along Cin (the direct kernel's 4-slice form).  Goal: at least ~2 workgroups per CU (512) while every part keeps >= 2
// chunks.  Only with split_counters (the host's statement that it provides the hand-over workspace: dm3d_conv_scratch_bytes).
int dm3d_conv_h3v2_ksplit(const ConvArgs& a) {
    static const int mode = [] { const char* e = getenv("DM3D_CONV_KSPLIT"); return e ? atoi(e) : -1; }();   // 0: never split (A/B, debugging)
    if (mode == 0 || !a.split_counters) return 1;
    const long wgs = dm3d_conv_split_tiles(a, 4);
    static const long wg_limit = [] { const char* e = getenv("DM3D_CONV_SPLIT_WGS"); return e ? atol(e) : 256L; }();   // A/B knob
    if (wgs > wg_limit || a.nchunks < 4) return 1;
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

int64_t dm3d_h3v2_image_bytes(int taps, int cin, int cout) {
    const int g = 4, tapsp = (taps + g - 1) / g * g;
    return (int64_t)tapsp * dm3d_round_up(cout, 64) * (dm3d_round_up(cin, 16) / 16) * REC * (int64_t)sizeof(_Float16);
}

int dm3d_pack_h3v2(const float* keras_kernel, int taps, int cin, int cout, int w_exp, const float* in_scale, void* packed, int mode,
                   hipStream_t st) {
    const int nchunks = (int)(dm3d_round_up(cin, 16) / 16), ntiles = (int)(dm3d_round_up(cout, 64) / 64);
    const int g = 4;
    hipLaunchKernelGGL(pack_weights_h3v2_kernel, dim3(4096), dim3(256), 0, st, keras_kernel, taps, (taps + g - 1) / g * g, cin, cout,
                       nchunks, ntiles, ldexpf(1.0f, w_exp), in_scale, static_cast<_Float16*>(packed), mode);
    return dm3d_launch_check("pack_weights_h3v2_kernel");
}
