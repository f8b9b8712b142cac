// dm3d_conv.hip — Conv3D(padding="same") on NDHWC float32 as an implicit GEMM on v_mfma_f32_32x32x2_f32.
//
// Replaces layers.Conv3D / UpSampling3D+Conv3D / Concatenate+Conv3D and the BatchNormalization+swish in front of them
// (reference networks/conditional_dm3d.py:238-296, 348-353, 410-414).  GEMM view: M = batch*Dout*Hout*Wout output
// voxels, N = Cout, K = k^3 * Cin.
//
// One workgroup (256 threads = 4 waves) owns a TD x TH x TW brick of output voxels times 64 output channels.
//   for each 16-channel chunk of Cin:
//       stage the brick's input halo ((TD-1)*S+k) x ((TH-1)*S+k) x ((TW-1)*S+k) voxels x 16 channels into LDS once,
//       applying the fused prologue silu(x*scale+shift) and the zero padding on the way in;
//       for each of the k^3 taps: stream that tap's [64 cout][16 cin] weight slice into a double-buffered LDS slot
//       (one barrier per tap) and issue the MFMAs; a tap only shifts the LDS read address of the A operand.
// Every input element is fetched from HBM/L2 once per workgroup and chunk (halo overhead 2.3x for a 4x8x8 brick) instead
// of k^3 times, and activation tensors are read in their raw form, so norm/activation/concat/upsample never round-trip
// through HBM.  LDS: 600*20*4 + 2*64*20*4 = 58 KB -> two workgroups per CU, whose staging and MFMA phases overlap.
#include <cstdlib>
#include "dm3d_conv_args.h"
#include <math.h>

namespace {

template <int TD, int TH, int TW, int S, int KS, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv3d_igemm_f32(const ConvArgs p) {
    constexpr int CK = 16, LDV = CK + 4, NT = 64;
    constexpr int TM = TD * TH * TW;
    constexpr int HD = (TD - 1) * S + KS, HH = (TH - 1) * S + KS, HW = (TW - 1) * S + KS;
    constexpr int HVOX = HD * HH * HW;
    constexpr int TAPS = KS * KS * KS;
    constexpr int MR = TM / WM / 32, NR = NT / WN / 32;
    constexpr int NTHR = WM * WN * 64;
    constexpr int NSLOT = (HVOX * 4 + NTHR - 1) / NTHR;
    static_assert(NTHR == 256, "kernel assumes 4 waves");
    static_assert(TM % (WM * 32) == 0 && NT % (WN * 32) == 0, "tile split");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* lds_in = smem;                       // [HVOX][LDV]
    float* lds_w = smem + HVOX * LDV;           // [2][NT][LDV]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int wm = wave / WN, wn = wave % WN;

    // brick -> (sample, origin)
    int brick = blockIdx.x;
    const int bpv = p.bd * p.bh * p.bw;
    const int b = brick / bpv;
    brick -= b * bpv;
    const int oz0 = (brick / (p.bh * p.bw)) * TD;
    const int oy0 = ((brick / p.bw) % p.bh) * TH;
    const int ox0 = (brick % p.bw) * TW;
    const int n0 = blockIdx.y * NT;
    int padz = p.padz, pady = p.pady, padx = p.padx, ooz = p.ooz, ooy = p.ooy, oox = p.oox;
    const float* wbase = static_cast<const float*>(p.wpk);
    if (p.parity) {                             // uniform: one output parity of upsample(2) + conv k3 per grid.z slice
        const int par = blockIdx.z;
        ooz = par >> 2; ooy = (par >> 1) & 1; oox = par & 1;
        padz = 1 - ooz; pady = 1 - ooy; padx = 1 - oox;
        wbase += (size_t)par * p.w_parity_stride;
    }

    // staging slots: this thread copies float4 piece `piece` of halo voxels hv0 + j*64
    const int piece = tid & 3;
    int gvox[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        const int hv = (tid >> 2) + j * (NTHR / 4);
        int g = -1;
        if (hv < HVOX) {
            const int hz = hv / (HH * HW), hy = (hv / HW) % HH, hx = hv % HW;
            const int iz = oz0 * S - padz + hz, iy = oy0 * S - pady + hy, ix = ox0 * S - padx + hx;
            if (iz >= 0 && iz < p.ind && iy >= 0 && iy < p.inh && ix >= 0 && ix < p.inw)
                g = ((b * p.ind + iz) * p.inh + iy) * p.inw + ix;
        }
        gvox[j] = g;
    }

    // per-lane LDS row bases (floats) for the A operand of each 32-row tile, tap (0,0,0)
    const float* a_base[MR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
        const int r = wm * (TM / WM) + mr * 32 + l32;
        const int dz = r / (TH * TW), dy = (r / TW) % TH, dx = r % TW;
        a_base[mr] = lds_in + ((dz * S * HH + dy * S) * HW + dx * S) * LDV;
    }
    int b_off[NR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) b_off[nr] = (wn * (NT / WN) + nr * 32 + l32) * LDV;

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.0f;

    // weight slice (tap, chunk): row n = tid>>2 of 64, float4 piece tid&3 of the 16 input channels
    const float* w_thread = wbase + (size_t)(n0 + (tid >> 2)) * p.cinpad + piece * 4;
    const size_t w_tap_stride = (size_t)p.coutpad * p.cinpad;
    const int w_lds_off = (tid >> 2) * LDV + piece * 4;
    f32x4 wreg = *reinterpret_cast<const f32x4*>(w_thread);      // (tap 0, chunk 0)

    for (int ch = 0; ch < p.nchunks; ++ch) {
        const int c0 = ch * CK;
        const float* src;
        int ldc, cb;
        if (c0 < p.c1) { src = p.x1; ldc = p.c1; cb = c0; } else { src = p.x2; ldc = p.c2; cb = c0 - p.c1; }
        const bool chan_ok = cb + piece * 4 < ldc;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
        const bool pro = p.pscale != nullptr;
        if (pro && chan_ok) {
            sc = *reinterpret_cast<const f32x4*>(p.pscale + (size_t)b * p.pro_bstride + c0 + piece * 4);
            sh = *reinterpret_cast<const f32x4*>(p.pshift + (size_t)b * p.pro_bstride + c0 + piece * 4);
        }
        // unconditional loads on clamped addresses, masked afterwards: a load inside a divergent branch costs one
        // serialized memory round trip each (hipcc waits vmcnt(0) per branch)
        const int coff = chan_ok ? cb + piece * 4 : 0;
        f32x4 hval[NSLOT];
#pragma unroll
        for (int j = 0; j < NSLOT; ++j)
            hval[j] = *reinterpret_cast<const f32x4*>(src + (size_t)(gvox[j] >= 0 ? gvox[j] : 0) * ldc + coff);
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            f32x4 v = hval[j];
            if (pro) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = dm3d_silu(fmaf(v[e], sc[e], sh[e]));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (gvox[j] >= 0 && chan_ok) ? v[e] : 0.0f;
            hval[j] = v;
        }
        __syncthreads();                        // every wave is done with the previous chunk's halo and weight slots
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const int hv = (tid >> 2) + j * (NTHR / 4);
            if (hv < HVOX) *reinterpret_cast<f32x4*>(lds_in + hv * LDV + piece * 4) = hval[j];
        }
        *reinterpret_cast<f32x4*>(lds_w + w_lds_off) = wreg;
        __syncthreads();

        for (int tap = 0; tap < TAPS; ++tap) {
            // prefetch the next weight slice (next tap, or tap 0 of the next chunk)
            const bool last_tap = tap + 1 == TAPS;
            {   // unconditional, clamped: a load behind a branch is waited for on the spot
                int nt = tap + 1, nc = c0;
                if (last_tap) { nt = 0; nc = (ch + 1 < p.nchunks) ? c0 + CK : c0; }
                wreg = *reinterpret_cast<const f32x4*>(w_thread + (size_t)nt * w_tap_stride + nc);
            }

            const int kd = tap / (KS * KS), kh = (tap / KS) % KS, kw = tap % KS;
            const int tap_off = ((kd * HH + kh) * HW + kw) * LDV;
            const float* wbuf = lds_w + (tap & 1) * (NT * LDV);
            const float* a_lds[MR];
            const float* b_lds[NR];
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) a_lds[mr] = a_base[mr] + tap_off;
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) b_lds[nr] = wbuf + b_off[nr];
            dm3d_mma_step<MR, NR, CK>(acc, a_lds, b_lds, half);

            if (!last_tap) {
                *reinterpret_cast<f32x4*>(lds_w + ((tap + 1) & 1) * (NT * LDV) + w_lds_off) = wreg;
                __syncthreads();
            }
        }
    }

    // epilogue: + bias + vec[row(b)] -> relu -> + res -> store.  Lanes 0..31 cover 32 consecutive output channels.
    // Loads are unconditional on clamped indices; only the stores are predicated.
    const int vrow = p.vec ? (p.vec_idx ? p.vec_idx[b] : b) : 0;
    const size_t sample_elems = (size_t)p.fd * p.fh * p.fw * p.cout;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int n = n0 + wn * (NT / WN) + nr * 32 + l32;
        const bool n_ok = n < p.cout;
        const int nc = n_ok ? n : p.cout - 1;
        float add = p.bias ? p.bias[nc] : 0.0f;
        if (p.vec) add += p.vec[(size_t)vrow * p.vec_ld + nc];
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
            size_t o[16];
            bool ok[16];
            float rv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (TM / WM) + mr * 32 + dm3d_acc_row(r, half);
                const int oz = oz0 + row / (TH * TW), oy = oy0 + (row / TW) % TH, ox = ox0 + row % TW;
                ok[r] = n_ok && oz < p.od && oy < p.oh && ox < p.ow;
                o[r] = ok[r] ? ((((size_t)b * p.fd + oz * p.os + ooz) * p.fh + oy * p.os + ooy) * p.fw + ox * p.os + oox) * p.cout + n : 0;
            }
            if (p.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = p.res[o[r]];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mr][nr][r] + add;
                if (p.relu) v = fmaxf(v, 0.0f);
                if (p.prelu) { const float al = p.prelu[o[r] % sample_elems]; v = v > 0.0f ? v : al * v; }
                if (p.res) v += rv[r];
                if (p.relu_out) v = fmaxf(v, 0.0f);
                if (ok[r]) p.out[o[r]] = v;
            }
        }
    }
}

template <int TD, int TH, int TW, int S, int KS, int WM, int WN>
int launch_conv(ConvArgs& a, hipStream_t st) {
    constexpr int LDV = 20;
    constexpr int HVOX = ((TD - 1) * S + KS) * ((TH - 1) * S + KS) * ((TW - 1) * S + KS);
    constexpr size_t lds = (size_t)(HVOX * LDV + 2 * 64 * LDV) * sizeof(float);
    a.bd = (a.od + TD - 1) / TD;
    a.bh = (a.oh + TH - 1) / TH;
    a.bw = (a.ow + TW - 1) / TW;
    static std::atomic<bool> attr_set[64] = {};            // per device: the attribute belongs to the device the launch goes to
    int dev_ = 0;
    DM3D_HIP(hipGetDevice(&dev_));
    DM3D_REQUIRE(dev_ >= 0 && dev_ < 64, "conv: device ordinal %d", dev_);
    if (!attr_set[dev_]) {
        DM3D_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_igemm_f32<TD, TH, TW, S, KS, WM, WN>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev_] = true;
    }
    dim3 grid((unsigned)(a.batch * a.bd * a.bh * a.bw), (unsigned)(a.coutpad / 64), a.parity ? 8u : 1u);
    hipLaunchKernelGGL((conv3d_igemm_f32<TD, TH, TW, S, KS, WM, WN>), grid, dim3(256), lds, st, a);
    return dm3d_launch_check("conv3d_igemm_f32");
}

}  // namespace

extern "C" int dm3d_conv3d_ndhwc(const dm3d_conv_desc* d, void* stream) {
    DM3D_REQUIRE(d != nullptr, "conv: null descriptor");
    DM3D_REQUIRE(d->x1 && d->wpk && d->out, "conv: x1/wpk/out must be non-null");
    DM3D_REQUIRE(d->ksize == 1 || d->ksize == 3 || (d->ksize == 4 && d->stride == 2), "conv: ksize %d not in {1,3} (4 needs stride 2)", d->ksize);
    DM3D_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride %d not in {1,2}", d->stride);
    DM3D_REQUIRE(!d->transpose || (d->ksize == 4 && d->stride == 2 && !d->upsample), "conv: transpose needs ksize 4, stride 2, no upsample");
    DM3D_REQUIRE(!(d->upsample && d->stride != 1), "conv: upsample requires stride 1");
    DM3D_REQUIRE(!(d->ksize == 1 && (d->stride != 1 || d->upsample)), "conv: ksize 1 supports stride 1 without upsample only");
    DM3D_REQUIRE(d->batch > 0 && d->in_d > 0 && d->in_h > 0 && d->in_w > 0 && d->cout > 0, "conv: non-positive extent");
    DM3D_REQUIRE(d->c1 > 0 && d->c1 % 4 == 0 && d->c2 >= 0 && d->c2 % 4 == 0, "conv: c1=%d c2=%d must be multiples of 4", d->c1, d->c2);
    DM3D_REQUIRE((d->c2 == 0) == (d->x2 == nullptr), "conv: x2 and c2 must be given together");
    DM3D_REQUIRE(d->c2 == 0 || d->c1 % 16 == 0, "conv: with a second input c1=%d must be a multiple of 16", d->c1);
    DM3D_REQUIRE((d->pro_scale == nullptr) == (d->pro_shift == nullptr), "conv: pro_scale and pro_shift go together");
    DM3D_REQUIRE(!d->vec || d->vec_ld >= d->cout, "conv: vec_ld %d < cout %d", d->vec_ld, d->cout);
    const void* ptrs[] = {d->x1, d->x2, d->wpk, d->pro_scale, d->pro_shift, d->out, d->res};
    for (const void* q : ptrs) DM3D_REQUIRE(dm3d_aligned16(q), "conv: pointer %p is not 16-byte aligned", q);
    const int64_t vox = (int64_t)d->batch * d->in_d * d->in_h * d->in_w;
    DM3D_REQUIRE(vox < (1ll << 31) / 4, "conv: %lld voxels overflow the 32-bit voxel index", (long long)vox);

    ConvArgs a{};
    a.x1 = d->x1; a.x2 = d->x2; a.c1 = d->c1; a.c2 = d->c2;
    a.ind = d->in_d; a.inh = d->in_h; a.inw = d->in_w;
    // Logical output domain the bricks tile.  upsample: the k3 conv on the nearest-2x upsampled tensor is evaluated as 8
    // 2x2x2 convs (one per output parity) on the low-resolution input, so the domain is the input extent and results are
    // scattered with stride 2 into the full output.
    const bool par_mode = d->upsample || d->transpose;      // both run as 8 parity 2x2x2 convs on the input grid
    const int cstride = par_mode ? 1 : d->stride;
    a.lgd = d->in_d; a.lgh = d->in_h; a.lgw = d->in_w;
    a.od = (d->in_d + cstride - 1) / cstride;
    a.oh = (d->in_h + cstride - 1) / cstride;
    a.ow = (d->in_w + cstride - 1) / cstride;
    a.parity = par_mode ? 1 : 0;
    a.os = par_mode ? 2 : 1;
    a.fd = a.od * a.os; a.fh = a.oh * a.os; a.fw = a.ow * a.os;
    // TF SAME: total = max((out-1)*stride + k - in, 0), zeros in front = total/2 (k=3: stride 1 -> 1; stride 2 -> 0 on even
    // sizes, 1 on odd sizes).  In parity mode the kernel derives the pads from the parity bits.
    auto pad_front = [&](int in, int out) { int t = (out - 1) * cstride + d->ksize - in; return t > 0 ? t / 2 : 0; };
    a.padz = pad_front(a.lgd, a.od); a.pady = pad_front(a.lgh, a.oh); a.padx = pad_front(a.lgw, a.ow);
    const int cin = d->c1 + d->c2;
    a.cinpad = (int)dm3d_round_up(cin, DM3D_CIN_PAD);
    a.coutpad = (int)dm3d_round_up(d->cout, DM3D_COUT_PAD);
    a.nchunks = a.cinpad / 16;
    a.wpk = d->wpk; a.bias = d->bias; a.pscale = d->pro_scale; a.pshift = d->pro_shift; a.pro_bstride = d->pro_batch_stride;
    DM3D_REQUIRE(d->pro_batch_stride == 0 || (d->pro_batch_stride >= d->c1 + d->c2 && d->pro_batch_stride % 4 == 0),
                 "conv: pro_batch_stride must be 0 or a multiple of 4 >= c1+c2");
    a.vec = d->vec; a.vec_idx = d->vec_idx; a.vec_ld = d->vec_ld;
    a.relu = d->relu; a.res = d->res; a.out = d->out; a.cout = d->cout;
    a.prelu = d->prelu_alpha; a.relu_out = d->relu_out;
    a.scratch = d->scratch; a.scratch_bytes = d->scratch_bytes;
    DM3D_REQUIRE((d->split_counters == nullptr) == (d->split_counter_words == 0) && d->split_counter_words >= 0, "conv: split_counters and split_counter_words go together");
    a.split_counters = d->split_counters; a.split_counter_words = d->split_counter_words;
    a.range_flag = d->range_flag; a.range_limit = d->range_limit > 0.0f ? d->range_limit : 65504.0f;
    if (d->wpk_wino) {
        DM3D_REQUIRE(d->precision == DM3D_PREC_H3 && dm3d_aligned16(d->wpk_wino), "conv: wpk_wino needs precision H3 and 16-byte alignment");
        a.wpk_wino = d->wpk_wino;
    }
    DM3D_REQUIRE((d->x1_fmt == DM3D_FMT_F32 || d->x1_fmt == DM3D_FMT_H2) && (d->out_fmt == DM3D_FMT_F32 || d->out_fmt == DM3D_FMT_H2),
                 "conv: unknown x1_fmt / out_fmt");
    DM3D_REQUIRE((d->post_scale == nullptr) == (d->post_shift == nullptr), "conv: post_scale and post_shift go together");
    if (d->x1_fmt == DM3D_FMT_H2 || d->out_fmt == DM3D_FMT_H2 || d->post_scale) {
        DM3D_REQUIRE(d->precision == DM3D_PREC_H3 && d->ksize == 3 && d->stride == 1 && !d->upsample && !d->transpose,
                     "conv: x1_fmt / out_fmt / post_* need precision H3, ksize 3, stride 1, no upsample / transpose");
        DM3D_REQUIRE(d->cout > 32 || (d->out_fmt == DM3D_FMT_F32 && !d->post_scale), "conv: out_fmt H2 / post_* need cout > 32");
        DM3D_REQUIRE(d->x1_fmt != DM3D_FMT_H2 || (d->c2 == 0 && d->c1 % 16 == 0 && !d->pro_scale),
                     "conv: an H2 input needs c1 %% 16 == 0, no second input and no prologue");
        DM3D_REQUIRE(d->out_fmt != DM3D_FMT_H2 || (d->in_d % 4 == 0 && d->in_h % 8 == 0 && d->in_w % 8 == 0 && d->cout % 64 == 0 &&
                                                  !d->prelu_alpha),
                     "conv: an H2 output needs extents of whole 4x8x8 bricks and cout %% 64 == 0");
        DM3D_REQUIRE(!d->post_scale || (d->in_d % 4 == 0 && d->in_h % 8 == 0 && d->in_w % 8 == 0 && d->cout % 64 == 0),
                     "conv: post_scale needs extents of whole 4x8x8 bricks and cout %% 64 == 0");
        DM3D_REQUIRE(dm3d_aligned16(d->post_scale) && dm3d_aligned16(d->post_shift), "conv: post_* must be 16-byte aligned");
        a.x_h2 = d->x1_fmt == DM3D_FMT_H2; a.out_h2 = d->out_fmt == DM3D_FMT_H2;
        a.post_scale = d->post_scale; a.post_shift = d->post_shift;
    }
    if (d->skip_wpk) {
        DM3D_REQUIRE(d->precision == DM3D_PREC_H3 && d->ksize == 3 && d->stride == 1 && !d->upsample && !d->transpose && d->cout > 32,
                     "conv: the fused skip conv needs precision H3, ksize 3, stride 1, no upsample / transpose, cout > 32");
        DM3D_REQUIRE(d->skip_x1 && d->skip_c1 > 0 && d->skip_c1 % 4 == 0 && d->skip_c2 >= 0 && d->skip_c2 % 4 == 0, "conv: skip_c1=%d skip_c2=%d must be multiples of 4", d->skip_c1, d->skip_c2);
        DM3D_REQUIRE((d->skip_c2 == 0) == (d->skip_x2 == nullptr) && (d->skip_c2 == 0 || d->skip_c1 % 16 == 0), "conv: skip_x2 / skip_c2 go together and need skip_c1 %% 16 == 0");
        DM3D_REQUIRE(dm3d_aligned16(d->skip_x1) && dm3d_aligned16(d->skip_x2) && dm3d_aligned16(d->skip_wpk), "conv: skip pointers must be 16-byte aligned");
        a.sx1 = d->skip_x1; a.sx2 = d->skip_x2; a.sc1 = d->skip_c1; a.sc2 = d->skip_c2; a.swpk = d->skip_wpk;
        a.s_npairs = (int)(dm3d_round_up(d->skip_c1 + d->skip_c2, 32) / 32);
        DM3D_REQUIRE(dm3d_aligned16(d->skip_wpk_frag), "conv: skip_wpk_frag must be 16-byte aligned");
        a.swpk_f = d->skip_wpk_frag;
    }
    DM3D_REQUIRE(dm3d_aligned16(d->scratch) && d->scratch_bytes >= 0, "conv: scratch must be 16-byte aligned");
    a.batch = d->batch;
    DM3D_REQUIRE(d->precision == DM3D_PREC_F32 || d->precision == DM3D_PREC_H3, "conv: unknown precision %d", d->precision);
    DM3D_REQUIRE(d->w_exp >= -100 && d->w_exp <= 100, "conv: w_exp %d out of range", d->w_exp);
    a.out_scale = d->precision == DM3D_PREC_H3 ? ldexpf(1.0f, -d->w_exp) : 1.0f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int which = par_mode ? DM3D_CONV_UP
                               : (d->ksize == 1 ? DM3D_CONV_K1 : (d->ksize == 4 ? DM3D_CONV_K4S2 : (d->stride == 2 ? DM3D_CONV_K3S2 : DM3D_CONV_K3S1)));
    if (par_mode) a.w_parity_stride = d->precision == DM3D_PREC_H3
        ? dm3d_packed_weight_h3_bytes(8, cin, d->cout) / 2 : dm3d_packed_weight_elems(8, cin, d->cout);
    // Fused GroupNormalization statistics of the output (gn_stats): the 16x16x32 kernels' full-brick epilogue accumulates them; behind every
    // other kernel or form the stand-alone statistics kernel reads the finished output once (same buffer, same result up to summation order).
    if (d->gn_stats) {
        DM3D_REQUIRE(d->out_fmt == DM3D_FMT_F32 && d->cout % 4 == 0 && dm3d_aligned16(d->gn_stats),
                     "conv: gn_stats needs a float32 output, cout %% 4 == 0 and a 16-byte aligned buffer");
        a.gn_stats = d->gn_stats;
    }
    auto stats_behind = [&](int rc) {
        if (rc != DM3D_OK || !a.gn_stats) return rc;
        return dm3d_groupnorm_partials(a.out, a.batch, (int64_t)a.fd * a.fh * a.fw, a.cout, a.gn_stats, stream);
    };
    if (d->precision != DM3D_PREC_H3) return stats_behind(dm3d_conv_launch_f32(a, which, st));
    const int layout = dm3d_conv_weight_layout(d->ksize, d->stride, d->upsample, d->transpose, d->cout);
    DM3D_REQUIRE(d->w_layout == layout, "conv: w_layout %d but this geometry reads layout %d (dm3d_conv_weight_layout)", d->w_layout, layout);
    if (layout != DM3D_WL_PAIR) return stats_behind(dm3d_conv_launch_h3(a, which, st));
    // same arguments and epilogue: the Winograd-x form (dm3d_conv_h3w.hip, its own weight image) where it is eligible, else the
    // free-running direct form (dm3d_conv_h3v3.hip)
    if (dm3d_conv_h3w_serves(a, which)) return dm3d_conv_launch_h3w(a, which, st);
    return dm3d_conv_launch_h3v3(a, which, st);
}

extern "C" int64_t dm3d_packed_weight_skip_h3p_bytes(int32_t cin, int32_t cout) {
    return (cin > 0 && cout > 0) ? dm3d_h3v2_skip_image_bytes(cin, cout) : 0;
}

extern "C" int dm3d_pack_weights_skip_h3p(const float* keras_kernel, int32_t cin, int32_t cout, int32_t w_exp, void* packed, void* stream) {
    DM3D_REQUIRE(keras_kernel && packed && cin > 0 && cout > 0, "pack_weights_skip_h3p: bad arguments");
    DM3D_REQUIRE(w_exp >= -100 && w_exp <= 100 && dm3d_aligned16(packed), "pack_weights_skip_h3p: w_exp out of range or packed unaligned");
    return dm3d_pack_skip_h3v2(keras_kernel, cin, cout, w_exp, packed, static_cast<hipStream_t>(stream));
}

extern "C" int dm3d_pack_weights_skip_h3f(const float* keras_kernel, int32_t cin, int32_t cout, int32_t w_exp, void* packed, void* stream) {
    DM3D_REQUIRE(keras_kernel && packed && cin > 0 && cout > 0, "pack_weights_skip_h3f: bad arguments");
    DM3D_REQUIRE(w_exp >= -100 && w_exp <= 100 && dm3d_aligned16(packed), "pack_weights_skip_h3f: w_exp out of range or packed unaligned");
    return dm3d_pack_skip_h3f(keras_kernel, cin, cout, w_exp, packed, static_cast<hipStream_t>(stream));
}

extern "C" int32_t dm3d_conv_tile_form(const dm3d_conv_desc* d) {
    if (!d || d->precision != DM3D_PREC_H3) return 0;
    if (dm3d_conv_weight_layout(d->ksize, d->stride, d->upsample, d->transpose, d->cout) != DM3D_WL_PAIR) return 0;
    ConvArgs a{};
    const bool par_mode = d->upsample || d->transpose;
    a.batch = d->batch; a.od = d->in_d; a.oh = d->in_h; a.ow = d->in_w;      // (WL_PAIR convs are stride 1 / parity convs on the input grid)
    a.coutpad = (int)dm3d_round_up(d->cout, DM3D_COUT_PAD);
    a.parity = par_mode ? 1 : 0;
    a.s_npairs = d->skip_wpk ? (int)(dm3d_round_up(d->skip_c1 + d->skip_c2, 32) / 32) : 0;
    a.swpk_f = d->skip_wpk ? d->skip_wpk_frag : nullptr; a.sc1 = d->skip_c1; a.sc2 = d->skip_c2;
    a.wpk_wino = d->wpk_wino; a.cout = d->cout; a.c1 = d->c1; a.c2 = d->c2; a.ind = d->in_d; a.inh = d->in_h; a.inw = d->in_w;
    a.nchunks = (int)(dm3d_round_up(d->c1 + d->c2, DM3D_CIN_PAD) / 16);
    a.x1 = d->x1; a.x2 = d->x2; a.out = d->out; a.res = d->res; a.relu = d->relu; a.relu_out = d->relu_out; a.prelu = d->prelu_alpha;
    a.out_h2 = d->out_fmt == DM3D_FMT_H2; a.post_scale = d->post_scale;
    a.split_counters = d->split_counters; a.split_counter_words = d->split_counter_words;      // (no Cin split without them)
    a.padz = a.pady = a.padx = (d->ksize == 3 && d->stride == 1 && !par_mode) ? 1 : 0;
    if (dm3d_conv_h3w_serves(a, (d->ksize == 3 && d->stride == 1 && !par_mode) ? DM3D_CONV_K3S1 : DM3D_CONV_UP)) return 10;
    return dm3d_conv_h3v3_td(a);
}

// What a Cin-split launch of this descriptor needs (the hand-over form, dm3d_conv_h3v2_parts.h): tiles x parts x a tile's image.  Both
// 16x16x32 kernels are asked (which one serves the launch depends on the weight images the caller passes at launch time): the larger need.
static void split_needs(const dm3d_conv_desc* d, int64_t& bytes, int64_t& words) {
    bytes = words = 0;
    if (!d || d->precision != DM3D_PREC_H3 || d->batch <= 0 || d->cout <= 0 || d->in_d <= 0 || d->in_h <= 0 || d->in_w <= 0) return;
    if (dm3d_conv_weight_layout(d->ksize, d->stride, d->upsample, d->transpose, d->cout) != DM3D_WL_PAIR) return;
    const bool par = d->upsample || d->transpose;
    ConvArgs a{};
    int dummy = 0;
    a.split_counters = &dummy;                               // "the host will provide them": what the split would be
    a.batch = d->batch; a.od = d->in_d; a.oh = d->in_h; a.ow = d->in_w; a.parity = par ? 1 : 0;
    a.coutpad = (int)dm3d_round_up(d->cout, DM3D_COUT_PAD);
    a.nchunks = (int)(dm3d_round_up(d->c1 + d->c2, DM3D_CIN_PAD) / 16);
    const int ks = dm3d_conv_h3v2_ksplit(a);
    if (ks > 1) {
        words = dm3d_conv_split_tiles(a, 4);
        bytes = words * ks * (int64_t)(4 * 8 * 8 * 64) * (int64_t)sizeof(float);
    }
    if (d->ksize == 3 && d->stride == 1 && !par && d->cout > 32 && d->in_d % 8 == 0 && d->in_h % 8 == 0 && d->in_w % 8 == 0 && dm3d_conv_h3w_ksplit(a) > 1) {
        const int64_t tw = dm3d_conv_split_tiles(a, 8), bw = tw * 2 * (int64_t)(8 * 8 * 8 * 64) * (int64_t)sizeof(float);
        if (tw > words) words = tw;
        if (bw > bytes) bytes = bw;
    }
}

extern "C" int64_t dm3d_conv_scratch_bytes(const dm3d_conv_desc* d) {
    int64_t bytes, words;
    split_needs(d, bytes, words);
    return bytes;
}

extern "C" int32_t dm3d_conv_split_counter_words(const dm3d_conv_desc* d) {
    int64_t bytes, words;
    split_needs(d, bytes, words);
    return (int32_t)words;
}

extern "C" int32_t dm3d_conv_weight_layout(int32_t ksize, int32_t stride, int32_t upsample, int32_t transpose, int32_t cout) {
    static const bool pair_off = [] { const char* e = getenv("DM3D_CONV_PAIR"); return e && e[0] == '0'; }();   // A/B switch
    if (pair_off) return DM3D_WL_TAP;
    if (upsample || transpose) return DM3D_WL_PAIR;
    return (ksize == 3 && stride == 1) ? DM3D_WL_PAIR : DM3D_WL_TAP;     // (round 3: also Cout <= 32, the narrow column forms of the 16x16x32 kernel)
}

extern "C" int64_t dm3d_packed_weight_h3p_bytes(int32_t taps, int32_t cin, int32_t cout) {
    if (taps <= 0 || cin <= 0 || cout <= 0) return 0;
    return dm3d_h3v2_image_bytes(taps, cin, cout);
}

extern "C" int dm3d_pack_weights_h3p(const float* keras_kernel, int32_t taps, int32_t cin, int32_t cout, int32_t w_exp,
                                     const float* in_scale, void* packed, int32_t mode, void* stream) {
    DM3D_REQUIRE(keras_kernel && packed && taps > 0 && cin > 0 && cout > 0, "pack_weights_h3p: bad arguments");
    DM3D_REQUIRE(mode >= 0 && mode <= 2 && (mode == 0 || taps == 8), "pack_weights_h3p: mode %d with taps %d", mode, taps);
    DM3D_REQUIRE(w_exp >= -100 && w_exp <= 100, "pack_weights_h3p: w_exp %d out of range", w_exp);
    DM3D_REQUIRE(dm3d_aligned16(packed), "pack_weights_h3p: packed must be 16-byte aligned");
    return dm3d_pack_h3v2(keras_kernel, taps, cin, cout, w_exp, in_scale, packed, mode, static_cast<hipStream_t>(stream));
}

extern "C" int64_t dm3d_packed_weight_h3w_bytes(int32_t cin, int32_t cout) {
    if (cin <= 0 || cout <= 0) return 0;
    return dm3d_h3v2_image_bytes(40, cin, cout);          // 5 tap pairs x 4 transform terms x 2 taps
}

extern "C" int dm3d_pack_weights_h3w(const float* keras_kernel, int32_t cin, int32_t cout, int32_t w_exp, const float* in_scale, void* packed,
                                     void* stream) {
    DM3D_REQUIRE(keras_kernel && packed && cin > 0 && cout > 0, "pack_weights_h3w: bad arguments");
    DM3D_REQUIRE(w_exp >= -100 && w_exp <= 100, "pack_weights_h3w: w_exp %d out of range", w_exp);
    DM3D_REQUIRE(dm3d_aligned16(packed), "pack_weights_h3w: packed must be 16-byte aligned");
    return dm3d_pack_h3v2(keras_kernel, 40, cin, cout, w_exp, in_scale, packed, 3, static_cast<hipStream_t>(stream));
}

int dm3d_conv_launch_f32(ConvArgs& a, int which, hipStream_t st) {
    if (which == DM3D_CONV_UP) return launch_conv<4, 8, 8, 1, 2, 4, 1>(a, st);
    if (which == DM3D_CONV_K1) return launch_conv<4, 8, 8, 1, 1, 4, 1>(a, st);
    if (which == DM3D_CONV_K3S2) return launch_conv<2, 4, 8, 2, 3, 2, 2>(a, st);
    if (which == DM3D_CONV_K4S2) return launch_conv<2, 4, 8, 2, 4, 2, 2>(a, st);
    return launch_conv<4, 8, 8, 1, 3, 4, 1>(a, st);
}
