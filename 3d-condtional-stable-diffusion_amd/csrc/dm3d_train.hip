// dm3d_train.hip — the kernels DiffusionModel.train_step needs beyond the sampling path (reference
// networks/conditional_dm3d.py:471-510): training-mode BatchNormalization (batch statistics, moving averages), the weight
// gradients of Conv3D / Dense as a contraction over voxels on v_mfma_f32_32x32x2_f32, the backward forms of the
// HBM-bound layers (BatchNorm+swish, LayerNormalization, softmax, ReLU / swish), q_sample, the loss and its gradient, Adam.
// Data gradients of Conv3D / Dense are the forward kernels on flipped / transposed weights (dm3d_flip_kernel + the packers).
// Everything here is exact float32 (gradients span too many octaves for the float16 hi/lo split of the sampling path).
#include "dm3d_common.h"
#include <math.h>

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
inline unsigned tgrid(long n, int cap = 4096) {
    long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}
__device__ __forceinline__ float sigmoidf_(float u) { return 1.0f / (1.0f + expf(-u)); }
// d act(u) / du
__device__ __forceinline__ float act_grad(float u, int act) {
    if (act == DM3D_ACT_RELU) return u > 0.0f ? 1.0f : 0.0f;
    if (act == DM3D_ACT_SILU) { const float s = sigmoidf_(u); return s * (1.0f + u * (1.0f - s)); }
    return 1.0f;
}
__device__ __forceinline__ float act_fwd(float u, int act) {
    if (act == DM3D_ACT_RELU) return fmaxf(u, 0.0f);
    if (act == DM3D_ACT_SILU) return u * sigmoidf_(u);
    return u;
}

// element (row, ch) of the channel concatenation [x1 | x2] (layers.Concatenate(axis=-1)([x, skip]), conditional_dm3d.py:396)
struct Cat {
    const float* x1; const float* x2; int c1, c2;
    __device__ __forceinline__ f32x4 load4(long row, int ch) const {      // ch % 4 == 0, c1 % 4 == 0
        return ch < c1 ? *reinterpret_cast<const f32x4*>(x1 + row * c1 + ch) : *reinterpret_cast<const f32x4*>(x2 + row * c2 + (ch - c1));
    }
};

// ---- BatchNormalization(training=True): finish the statistics started by dm3d_groupnorm_stats ----------------------
__global__ __launch_bounds__(256) void bn_finalize_kernel(double* __restrict__ acc, int batch, long voxels, int c, float eps,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                          float* __restrict__ mov_mean, float* __restrict__ mov_var, float momentum,
                                                          int unbiased) {
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch >= c) return;
    double s = 0, q = 0;
    for (int b = 0; b < batch; ++b) {
        double* a = acc + ((size_t)b * c + ch) * 2;
        s += a[0]; q += a[1];
        a[0] = 0; a[1] = 0;                                   // ready for the next use
    }
    const double n = (double)batch * (double)voxels, m = s / n;
    double var = q / n - m * m;
    var = var > 0 ? var : 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)gamma[ch] * rstd;
    scale[ch] = (float)sc;
    shift[ch] = (float)((double)beta[ch] - m * sc);
    mean_out[ch] = (float)m;
    rstd_out[ch] = (float)rstd;
    if (mov_mean) {
        // keras BatchNormalization(momentum): moving = moving*momentum + batch*(1 - momentum).  For rank-5 inputs Keras (TF2
        // behaviour) runs tf.nn.fused_batch_norm, whose batch variance output — the one fed to the moving average — carries
        // Bessel's correction n/(n-1); the normalisation itself uses the biased variance above.
        const double vmov = (unbiased && n > 1) ? var * (n / (n - 1.0)) : var;
        mov_mean[ch] = (float)((double)mov_mean[ch] * momentum + m * (1.0 - (double)momentum));
        mov_var[ch] = (float)((double)mov_var[ch] * momentum + vmov * (1.0 - (double)momentum));
    }
}

// y[row][ch] = act(cat(x1,x2)[row][ch]*scale[ch] + shift[ch])   (scale == NULL: plain concatenation / copy)
__global__ __launch_bounds__(256) void affine_act_cat_kernel(Cat x, long rows, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, int act, float* __restrict__ y) {
    const int ct4 = (x.c1 + x.c2) >> 2;
    const long n4 = rows * ct4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long row = i / ct4;
        const int ch = (int)(i % ct4) * 4;
        f32x4 v = x.load4(row, ch);
        if (scale) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + ch), sh = *reinterpret_cast<const f32x4*>(shift + ch);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_fwd(v[e], act);
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

// backward of a = act(BN_train(cat(x1,x2))): pass 1 — per-channel sums of du and du*xhat (du = g * act'(u)), float64 atomics.
// Thread -> fixed channel quad, strided rows (like groupnorm_stats_kernel).
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const float* __restrict__ g, Cat x, long rows,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                                                double* __restrict__ red) {
    extern __shared__ float sred[];             // [256][8]
    const int ct = x.c1 + x.c2, c4 = ct >> 2;
    const int tid = threadIdx.x;
    const int lanes = c4 < 256 ? c4 : 256, par = 256 / lanes;
    const int q0 = tid % lanes, vl = tid / lanes;
    const long slab = (rows + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * slab, r1 = r0 + slab < rows ? r0 + slab : rows;
    for (int q = q0; q < c4; q += lanes) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, sx = s;
        if (vl < par) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + q * 4), sh = *reinterpret_cast<const f32x4*>(shift + q * 4);
            const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + q * 4), rs = *reinterpret_cast<const f32x4*>(rstd + q * 4);
            for (long r = r0 + vl; r < r1; r += par) {
                const f32x4 xv = x.load4(r, q * 4);
                const f32x4 gv = *reinterpret_cast<const f32x4*>(g + r * ct + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float du = gv[e] * act_grad(fmaf(xv[e], sc[e], sh[e]), act);
                    s[e] += du;
                    sx[e] += du * ((xv[e] - mu[e]) * rs[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { sred[tid * 8 + e] = s[e]; sred[tid * 8 + 4 + e] = sx[e]; }
        __syncthreads();
        if (vl == 0) {
            double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
            for (int k = 0; k < par; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) { ds[e] += sred[(k * lanes + q0) * 8 + e]; dq[e] += sred[(k * lanes + q0) * 8 + 4 + e]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) { atomicAdd(red + (q * 4 + e) * 2, ds[e]); atomicAdd(red + (q * 4 + e) * 2 + 1, dq[e]); }
        }
        __syncthreads();
    }
}

// pass 2: dx = gamma*rstd*(du - sum(du)/N - xhat*sum(du*xhat)/N), added into dx1 / dx2 (either may be NULL: that input needs no gradient)
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const float* __restrict__ g, Cat x, long rows,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd, int act,
                                                               const double* __restrict__ red, float* __restrict__ dx1, float* __restrict__ dx2) {
    const int ct = x.c1 + x.c2, ct4 = ct >> 2;
    const long n4 = rows * ct4;
    const double inv_n = 1.0 / (double)rows;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long row = i / ct4;
        const int ch = (int)(i % ct4) * 4;
        float* dst = ch < x.c1 ? (dx1 ? dx1 + row * x.c1 + ch : nullptr) : (dx2 ? dx2 + row * x.c2 + (ch - x.c1) : nullptr);
        if (!dst) continue;
        const f32x4 xv = x.load4(row, ch);
        const f32x4 gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 o = *reinterpret_cast<f32x4*>(dst);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sc = scale[ch + e], rs = rstd[ch + e];
            const float du = gv[e] * act_grad(fmaf(xv[e], sc, shift[ch + e]), act);
            const float xh = (xv[e] - mean[ch + e]) * rs;
            const float m1 = (float)(red[(ch + e) * 2] * inv_n), m2 = (float)(red[(ch + e) * 2 + 1] * inv_n);
            o[e] += sc * (du - m1 - xh * m2);                 // sc = gamma*rstd
        }
        *reinterpret_cast<f32x4*>(dst) = o;
    }
}

// dgamma[ch] += red[ch][1], dbeta[ch] += red[ch][0]
__global__ void bn_param_grad_kernel(const double* __restrict__ red, int c, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int ch = blockIdx.x * 256 + threadIdx.x;
    if (ch < c) { dgamma[ch] += (float)red[ch * 2 + 1]; dbeta[ch] += (float)red[ch * 2]; }
}

// ---- Conv3D / Dense weight gradient: dW[tap][ci][co] += sum over voxels of a[voxel + tap][ci] * g[voxel][co] ---------
// GEMM view per tap: M = ci, N = co, K = batch*D*H*W voxels — both operands are stored voxel-major / channel-contiguous, which
// is exactly what v_mfma_f32_32x32x2_f32 wants when K is the voxel axis: lane l supplies A[i = l&31][k = l>>5] = a[voxel k][ci0 + i].
// Workgroup: 64 ci x 64 co, the three taps (dx = -1, 0, +1) of one (dz, dy) row (they share the g tile), a slice of the voxel
// range (grid.z folds tap rows x K slices x batch groups); K advances 32 voxels at a time through LDS; zero padding is applied
// while staging.  Partial results are added with float atomics into the Keras-layout gradient ([taps][cin][cout]).
struct WgradArgs {
    const float* a; const float* g; float* dw;
    int cin, cout;                 // channel counts (= row strides of a / g)
    int d, h, w;                   // spatial extent of one sample (ksize 1: d = rows per batch item, h = w = 1)
    int ksize;                     // 1 or 3
    int ntr;                       // tap rows: 9 (k3) or 1 (k1)
    int nsplit;                    // K slices per (tap row, batch item)
    long chunks;                   // 32-voxel chunks per batch item
    int batch;                     // independent problems (ksize 1 only; >1: per-item output)
    long a_bs, g_bs, dw_bs;        // element strides between batch items
};

__global__ __launch_bounds__(256, 2) void wgrad_f32_kernel(const WgradArgs p) {
    constexpr int KT = 32, LDA = 64 + 4;       // row stride 68 floats: rows 16 B aligned; 2-way conflicts at most on the scalar reads
    __shared__ __attribute__((aligned(16))) float lds_a[3][KT * LDA];
    __shared__ __attribute__((aligned(16))) float lds_g[KT * LDA];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
    int z = blockIdx.z;
    const int split = z % p.nsplit; z /= p.nsplit;
    const int tr = z % p.ntr;
    const int bi = z / p.ntr;
    const float* A = p.a + (size_t)bi * p.a_bs;
    const float* G = p.g + (size_t)bi * p.g_bs;
    const int ntap = p.ksize == 3 ? 3 : 1;
    const int dz = p.ksize == 3 ? tr / 3 - 1 : 0, dy = p.ksize == 3 ? tr % 3 - 1 : 0;
    const long vox = (long)p.d * p.h * p.w;                     // voxels per sample; samples are consecutive (the K axis spans all)
    const long c_lo = p.chunks * split / p.nsplit, c_hi = p.chunks * (split + 1) / p.nsplit;

    // staging assignment: thread -> (voxel row r = tid >> 3 [0..31], 8-channel piece = tid & 7 -> two float4 at piece*8, piece*8+4)
    const int srow = tid >> 3, sp = (tid & 7) * 8;
    const int wm = wave >> 1, wn = wave & 1;                    // wave's 32 x 32 quadrant of the 64 x 64 tile
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const bool a_ok0 = ci0 + sp < p.cin, a_ok1 = ci0 + sp + 4 < p.cin;
    const bool g_ok0 = co0 + sp < p.cout, g_ok1 = co0 + sp + 4 < p.cout;
    const long total_vox = p.ksize == 3 ? vox * p.batch : vox;  // k3: `batch` samples laid out consecutively form one K axis

    // Loads are unconditional on clamped addresses and masked afterwards (a load under a divergent branch makes hipcc wait for it on the
    // spot: eight serialised round trips per chunk), and the next chunk's operands travel in registers during this chunk's MFMAs.
    const long vox_total = p.ksize == 3 ? total_vox : vox;
    const int ca0 = a_ok0 ? ci0 + sp : 0, ca1 = a_ok1 ? ci0 + sp + 4 : 0;       // clamped channel offsets
    const int cg0 = g_ok0 ? co0 + sp : 0, cg1 = g_ok1 ? co0 + sp + 4 : 0;
    f32x4 ga, gb, av[3][2];
    bool gm = false, am[3] = {false, false, false};
    auto fetch = [&](long ch) {
        const long v = ch * KT + srow;                          // this thread's voxel (flat over samples for k3)
        gm = v < vox_total;
        const long vc = gm ? v : vox_total - 1;
        ga = *reinterpret_cast<const f32x4*>(G + vc * p.cout + cg0);
        gb = *reinterpret_cast<const f32x4*>(G + vc * p.cout + cg1);
        if (p.ksize == 3) {
            const long s = vc / vox, rem = vc - s * vox;
            const int zz = (int)(rem / ((long)p.h * p.w)), yy = (int)((rem / p.w) % p.h), xx = (int)(rem % p.w);
            const int iz = zz + dz, iy = yy + dy;
            const bool zy_ok = gm && iz >= 0 && iz < p.d && iy >= 0 && iy < p.h;
            const int izc = min(max(iz, 0), p.d - 1), iyc = min(max(iy, 0), p.h - 1);
            const long base = ((s * p.d + izc) * p.h + iyc) * (long)p.w;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int ix = xx + t - 1;
                am[t] = zy_ok && ix >= 0 && ix < p.w;
                const float* q = A + (base + min(max(ix, 0), p.w - 1)) * p.cin;
                av[t][0] = *reinterpret_cast<const f32x4*>(q + ca0);
                av[t][1] = *reinterpret_cast<const f32x4*>(q + ca1);
            }
        } else {
            am[0] = gm;
            const float* q = A + vc * p.cin;
            av[0][0] = *reinterpret_cast<const f32x4*>(q + ca0);
            av[0][1] = *reinterpret_cast<const f32x4*>(q + ca1);
        }
    };
    if (c_lo < c_hi) fetch(c_lo);
    for (long ch = c_lo; ch < c_hi; ++ch) {
        __syncthreads();                                        // previous chunk's MFMAs are done with the tiles
        *reinterpret_cast<f32x4*>(lds_g + srow * LDA + sp) = (gm && g_ok0) ? ga : z4;
        *reinterpret_cast<f32x4*>(lds_g + srow * LDA + sp + 4) = (gm && g_ok1) ? gb : z4;
#pragma unroll
        for (int t = 0; t < 3; ++t) {                           // (static indices: a runtime-indexed register array lives in scratch memory)
            if (t < ntap) {
                *reinterpret_cast<f32x4*>(lds_a[t] + srow * LDA + sp) = (am[t] && a_ok0) ? av[t][0] : z4;
                *reinterpret_cast<f32x4*>(lds_a[t] + srow * LDA + sp + 4) = (am[t] && a_ok1) ? av[t][1] : z4;
            }
        }
        __syncthreads();
        fetch(ch + 1 < c_hi ? ch + 1 : ch);
#pragma unroll
        for (int k = 0; k < KT; k += 2) {
            const float bv = lds_g[(k + half) * LDA + wn * 32 + l32];
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                if (t < ntap) {
                    const float avv = lds_a[t][(k + half) * LDA + wm * 32 + l32];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(avv, bv, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // D layout: register r of lane l = row (r&3) + 8*(r>>2) + 4*half (ci), column l32 (co): 32 consecutive co per row -> coalesced atomics
    const int co = co0 + wn * 32 + l32;
    if (co >= p.cout) return;
    float* DW = p.dw + (size_t)bi * p.dw_bs;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        if (t >= ntap) continue;
        const int tap = p.ksize == 3 ? tr * 3 + t : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = ci0 + wm * 32 + dm3d_acc_row(r, half);
            if (ci < p.cin) unsafeAtomicAdd(DW + ((size_t)tap * p.cin + ci) * p.cout + co, acc[t][r]);
        }
    }
}

// out[group][c] += sum over the group's rows of x[row][c]   (bias gradients: one group; the time-embedding add: one group per sample)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long rows_per_group, int c, float* __restrict__ out,
                                                     long ld_out) {
    __shared__ float sred[256 * 4];
    const int c4 = c >> 2, tid = threadIdx.x;
    const int lanes = c4 < 256 ? c4 : 256, par = 256 / lanes;
    const int q0 = tid % lanes, vl = tid / lanes;
    const long grp = blockIdx.y;
    const long slab = (rows_per_group + gridDim.x - 1) / gridDim.x;
    const long r0 = (long)blockIdx.x * slab, r1 = r0 + slab < rows_per_group ? r0 + slab : rows_per_group;
    const float* xg = x + (size_t)grp * rows_per_group * c;
    for (int q = q0; q < c4; q += lanes) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (vl < par)
            for (long r = r0 + vl; r < r1; r += par) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(xg + r * c + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += t[e];
            }
#pragma unroll
        for (int e = 0; e < 4; ++e) sred[tid * 4 + e] = s[e];
        __syncthreads();
        if (vl == 0) {
            float t[4] = {0, 0, 0, 0};
            for (int k = 0; k < par; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] += sred[(k * lanes + q0) * 4 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) unsafeAtomicAdd(out + grp * ld_out + q * 4 + e, t[e]);
        }
        __syncthreads();
    }
}

// Keras kernel [taps][cin][cout] -> [taps][cout][cin] with the tap order reversed: the kernel whose "same" stride-1 convolution of dy
// is the data gradient of the original convolution (taps == 1: the plain transpose, Dense / 1x1)
__global__ __launch_bounds__(256) void flip_transpose_kernel(const float* __restrict__ w, int taps, int cin, int cout, float* __restrict__ out) {
    const long n = (long)taps * cin * cout;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int ci = (int)(i % cin);
        const int co = (int)((i / cin) % cout);
        const int tap = (int)(i / ((long)cin * cout));
        out[i] = w[((long)(taps - 1 - tap) * cin + ci) * cout + co];
    }
}

// ---- LayerNormalization backward: one wavefront per row (c <= 1024); dx += ..., dgamma / dbeta += per-block partial sums ----
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, long rows, int c, float eps,
                                                            const float* __restrict__ gamma, const float* __restrict__ dy,
                                                            float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float sg[4][1024], sb[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = c >> 2;
    f32x4 pg[4], pb[4];                                          // this lane's partial dgamma / dbeta for its (up to 4) float4 columns
#pragma unroll
    for (int j = 0; j < 4; ++j) { pg[j] = f32x4{0.f, 0.f, 0.f, 0.f}; pb[j] = pg[j]; }
    for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * c);
        const f32x4* dr = reinterpret_cast<const f32x4*>(dy + row * c);
        f32x4 v[4], d[4];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane + j * 64;
            v[j] = i < nv ? xr[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            d[j] = i < nv ? dr[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
        const float mean = wsum(s) / (float)c;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (lane + j * 64 < nv)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float dl = v[j][e] - mean; q += dl * dl; }
        const float rstd = rsqrtf(wsum(q) / (float)c + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane + j * 64;
            if (i < nv) {
                const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (v[j][e] - mean) * rstd;
                    const float dg = d[j][e] * g[e];
                    s1 += dg; s2 += dg * xh;
                    pg[j][e] += d[j][e] * xh; pb[j][e] += d[j][e];
                    v[j][e] = xh; d[j][e] = dg;                  // keep xhat and dy*gamma
                }
            }
        }
        s1 = wsum(s1) / (float)c; s2 = wsum(s2) / (float)c;
        f32x4* ox = reinterpret_cast<f32x4*>(dx + row * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane + j * 64;
            if (i < nv) {
                f32x4 o = ox[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] += rstd * (d[j][e] - s1 - v[j][e] * s2);
                ox[i] = o;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + j * 64;
        if (i < nv)
#pragma unroll
            for (int e = 0; e < 4; ++e) { sg[wave][i * 4 + e] = pg[j][e]; sb[wave][i * 4 + e] = pb[j][e]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += 256) {
        unsafeAtomicAdd(dgamma + i, (sg[0][i] + sg[1][i]) + (sg[2][i] + sg[3][i]));
        unsafeAtomicAdd(dbeta + i, (sb[0][i] + sb[1][i]) + (sb[2][i] + sb[3][i]));
    }
}

// softmax backward in place: dS = scale * P * (dP - sum_j P_j dP_j) per row (scale: the alpha the scores were multiplied by)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, long rows, int cols, long ld,
                                                          float scale) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* pr = p + row * ld;
    float* dr = dp + row * ld;
    float s = 0.f;
    for (int i = lane; i < cols; i += 64) s += pr[i] * dr[i];
    s = wsum(s);
    for (int i = lane; i < cols; i += 64) dr[i] = scale * pr[i] * (dr[i] - s);
}

// dx = dy * act'(pre)  (out may alias dy);  act' evaluated on `ref`: the pre-activation (SiLU) or, for ReLU, pre or post alike
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ ref, const float* __restrict__ dy, float* __restrict__ dx,
                                                      long n4, int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 r = reinterpret_cast<const f32x4*>(ref)[i], d = reinterpret_cast<const f32x4*>(dy)[i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = d[e] * act_grad(r[e], act);
        reinterpret_cast<f32x4*>(dx)[i] = o;
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ dst, const float* __restrict__ src, long n4, long n, float alpha) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 d = reinterpret_cast<f32x4*>(dst)[i];
        const f32x4 s = reinterpret_cast<const f32x4*>(src)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = fmaf(alpha, s[e], d[e]);
        reinterpret_cast<f32x4*>(dst)[i] = d;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] += alpha * src[n4 * 4 + threadIdx.x];
}

__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ p, long n4, long n, float v) {
    const f32x4 z = {v, v, v, v};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) reinterpret_cast<f32x4*>(p)[i] = z;
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[n4 * 4 + threadIdx.x] = v;
}

// dst[b][j][i] = src[b][i][j]  (32 x 32 tiles through LDS)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int rows, int cols, long ld_src, long stride_src,
                                                        float* __restrict__ dst, long ld_dst, long stride_dst) {
    __shared__ float tile[32][33];
    const float* s = src + (size_t)blockIdx.z * stride_src;
    float* d = dst + (size_t)blockIdx.z * stride_dst;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + k * 8, c = c0 + tx;
        tile[ty + k * 8][tx] = (r < rows && c < cols) ? s[(size_t)r * ld_src + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + k * 8, r = r0 + tx;
        if (r < rows && c < cols) d[(size_t)c * ld_dst + r] = tile[tx][ty + k * 8];
    }
}

// dst[r][dst_off + j] (+)= src[r][src_off + j], j < c: a column window of one row-major matrix into another (concat / its backward)
__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, long ld_src, int src_off, float* __restrict__ dst,
                                                        long ld_dst, int dst_off, long rows, int c4, int accumulate) {
    const long n = rows * c4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / c4;
        const int j = (int)(i % c4) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(src + r * ld_src + src_off + j);
        f32x4* d = reinterpret_cast<f32x4*>(dst + r * ld_dst + dst_off + j);
        if (accumulate) { const f32x4 o = *d; v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
        *d = v;
    }
}

// UpSampling3D(size=2), nearest (conditional_dm3d.py:290): dst[b][2z+a][2y+b][2x+c][:] = src[b][z][y][x][:]
__global__ __launch_bounds__(256) void upsample2_kernel(const float* __restrict__ src, float* __restrict__ dst, int batch, int d, int h,
                                                        int w, int c4) {
    const long n = (long)batch * 8 * d * h * w * c4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int cc = (int)(i % c4);
        long v = i / c4;
        const int x = (int)(v % (2 * w)); v /= 2 * w;
        const int y = (int)(v % (2 * h)); v /= 2 * h;
        const int z = (int)(v % (2 * d));
        const long b = v / (2 * d);
        reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(src)[(((b * d + (z >> 1)) * h + (y >> 1)) * w + (x >> 1)) * c4 + cc];
    }
}
// its backward: dst[b][z][y][x][:] += sum of the 8 children of src
__global__ __launch_bounds__(256) void sumpool2_kernel(const float* __restrict__ src, float* __restrict__ dst, int batch, int d, int h,
                                                       int w, int c4) {
    const long n = (long)batch * d * h * w * c4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int cc = (int)(i % c4);
        long v = i / c4;
        const int x = (int)(v % w); v /= w;
        const int y = (int)(v % h); v /= h;
        const int z = (int)(v % d);
        const long b = v / d;
        f32x4 o = reinterpret_cast<f32x4*>(dst)[i];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 t = reinterpret_cast<const f32x4*>(src)[(((b * 2 * d + 2 * z + (k >> 2)) * 2 * h + 2 * y + ((k >> 1) & 1)) * 2 * w + 2 * x + (k & 1)) * c4 + cc];
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += t[e];
        }
        reinterpret_cast<f32x4*>(dst)[i] = o;
    }
}
// dst (zero filled, extent id x ih x iw) gets src[b][o] at position 2*o + off per axis: the dilated output gradient of a stride-2 conv
__global__ __launch_bounds__(256) void dilate2_kernel(const float* __restrict__ src, float* __restrict__ dst, int batch, int od, int oh,
                                                      int ow, int id, int ih, int iw, int offz, int offy, int offx, int c4) {
    const long n = (long)batch * od * oh * ow * c4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int cc = (int)(i % c4);
        long v = i / c4;
        const int x = (int)(v % ow); v /= ow;
        const int y = (int)(v % oh); v /= oh;
        const int z = (int)(v % od);
        const long b = v / od;
        const int zz = 2 * z + offz, yy = 2 * y + offy, xx = 2 * x + offx;
        if (zz < id && yy < ih && xx < iw)
            reinterpret_cast<f32x4*>(dst)[(((b * id + zz) * ih + yy) * iw + xx) * c4 + cc] = reinterpret_cast<const f32x4*>(src)[i];
    }
}

// table[idx[r]][:] += src[r][:]   (Embedding gradient)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, int rows, int c,
                                                               float* __restrict__ table, int table_rows) {
    const long n = (long)rows * c;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int r = (int)(i / c), cc = (int)(i % c);
        const int t = idx[r];
        if (t >= 0 && t < table_rows) unsafeAtomicAdd(table + (size_t)t * c + cc, src[i]);
    }
}

// noisy = sqrt_alpha_bar[t[b]] * latents + sqrt_one_minus_alpha_bar[t[b]] * noise   (conditional_dm3d.py:484-490)
__global__ __launch_bounds__(256) void q_sample_kernel(const float* __restrict__ lat, const float* __restrict__ noise, const int* __restrict__ t,
                                                       const float* __restrict__ sqab, const float* __restrict__ sq1ab, int timesteps,
                                                       float* __restrict__ out, long per4) {
    const int b = blockIdx.y;
    const int tt = min(max(t[b], 0), timesteps - 1);
    const float a = sqab[tt], s = sq1ab[tt];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per4; i += (long)gridDim.x * 256) {
        const f32x4 l = reinterpret_cast<const f32x4*>(lat)[b * per4 + i], z = reinterpret_cast<const f32x4*>(noise)[b * per4 + i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(__fmul_rn(a, l[e]), __fmul_rn(s, z[e]));
        reinterpret_cast<f32x4*>(out)[b * per4 + i] = o;
    }
}

// loss = sum((noise - pred)^2) * inv  (float64 accumulation; inv = 1 / (channels * global_bs * lc^4): the channel mean of
// keras MeanSquaredError, reduction SUM over b*d*h*w, the reference's divisor);  dpred = 2 * (pred - noise) * inv
__global__ __launch_bounds__(256) void mse_loss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ noise, long n4,
                                                            double inv, double* __restrict__ loss, float* __restrict__ dpred) {
    double s = 0;
    const float two_inv = (float)(2.0 * inv);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 p = reinterpret_cast<const f32x4*>(pred)[i], z = reinterpret_cast<const f32x4*>(noise)[i];
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = p[e] - z[e]; s += (double)d * (double)d; g[e] = d * two_inv; }
        if (dpred) reinterpret_cast<f32x4*>(dpred)[i] = g;
    }
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(loss, red[0] * inv);
}

// keras.optimizers.Adam (beta_1, beta_2, epsilon; main_conditional_dm.py:153): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// w -= lr_t * m / (sqrt(v) + eps), lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t) computed by the host
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr_t, float b1, float b2, float eps) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        w[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

}  // namespace

#define TR_ST static_cast<hipStream_t>(stream)

extern "C" int dm3d_batchnorm_finalize(double* acc, int32_t batch, int64_t voxels, int32_t c, float eps, const float* gamma, const float* beta,
                                       float* scale, float* shift, float* mean_out, float* rstd_out, float* moving_mean, float* moving_var,
                                       float momentum, int32_t unbiased_moving, void* stream) {
    DM3D_REQUIRE(acc && gamma && beta && scale && shift && mean_out && rstd_out && batch > 0 && voxels > 0 && c > 0, "batchnorm_finalize: bad arguments");
    DM3D_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), "batchnorm_finalize: moving_mean and moving_var go together");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((c + 255) / 256)), dim3(256), 0, TR_ST, acc, batch, (long)voxels, c, eps, gamma, beta,
                       scale, shift, mean_out, rstd_out, moving_mean, moving_var, momentum, unbiased_moving);
    return dm3d_launch_check("bn_finalize_kernel");
}

static int check_cat(const char* what, const float* x1, int c1, const float* x2, int c2) {
    DM3D_REQUIRE(x1 && c1 > 0 && c1 % 4 == 0 && c2 >= 0 && c2 % 4 == 0 && (c2 == 0) == (x2 == nullptr), "%s: c1=%d c2=%d must be multiples of 4, x2 given iff c2 > 0", what, c1, c2);
    DM3D_REQUIRE(dm3d_aligned16(x1) && dm3d_aligned16(x2), "%s: inputs must be 16-byte aligned", what);
    return DM3D_OK;
}

extern "C" int dm3d_affine_act_cat(const float* x1, int32_t c1, const float* x2, int32_t c2, int64_t rows, const float* scale, const float* shift,
                                   int32_t act, float* y, void* stream) {
    if (int rc = check_cat("affine_act_cat", x1, c1, x2, c2)) return rc;
    DM3D_REQUIRE(y && rows > 0 && dm3d_aligned16(y) && (scale == nullptr) == (shift == nullptr), "affine_act_cat: bad arguments");
    DM3D_REQUIRE(act >= DM3D_ACT_NONE && act <= DM3D_ACT_SILU, "affine_act_cat: unknown act %d", act);
    const Cat x{x1, x2, c1, c2};
    hipLaunchKernelGGL(affine_act_cat_kernel, dim3(tgrid(rows * ((c1 + c2) / 4))), dim3(256), 0, TR_ST, x, (long)rows, scale, shift, act, y);
    return dm3d_launch_check("affine_act_cat_kernel");
}

extern "C" int dm3d_bn_act_bwd(const float* g, const float* x1, int32_t c1, const float* x2, int32_t c2, int64_t rows, const float* scale,
                               const float* shift, const float* mean, const float* rstd, int32_t act, double* red, float* dx1, float* dx2,
                               float* dgamma, float* dbeta, void* stream) {
    if (int rc = check_cat("bn_act_bwd", x1, c1, x2, c2)) return rc;
    DM3D_REQUIRE(g && scale && shift && mean && rstd && red && rows > 0, "bn_act_bwd: null argument");
    DM3D_REQUIRE(act >= DM3D_ACT_NONE && act <= DM3D_ACT_SILU, "bn_act_bwd: unknown act %d", act);
    const Cat x{x1, x2, c1, c2};
    const int ct = c1 + c2, c4 = ct / 4, lanes = c4 < 256 ? c4 : 256;
    long slabs = rows / (256 / lanes * 8);
    slabs = slabs < 1 ? 1 : (slabs > 512 ? 512 : slabs);
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3((unsigned)slabs), dim3(256), 256 * 8 * sizeof(float), TR_ST, g, x, (long)rows, scale, shift,
                       mean, rstd, act, red);
    if (int rc = dm3d_launch_check("bn_act_bwd_reduce_kernel")) return rc;
    if (dx1 || dx2) {
        hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(tgrid(rows * (ct / 4))), dim3(256), 0, TR_ST, g, x, (long)rows, scale, shift, mean, rstd,
                           act, red, dx1, dx2);
        if (int rc = dm3d_launch_check("bn_act_bwd_apply_kernel")) return rc;
    }
    if (dgamma) {
        DM3D_REQUIRE(dbeta != nullptr, "bn_act_bwd: dgamma and dbeta go together");
        hipLaunchKernelGGL(bn_param_grad_kernel, dim3((unsigned)((ct + 255) / 256)), dim3(256), 0, TR_ST, red, ct, dgamma, dbeta);
        return dm3d_launch_check("bn_param_grad_kernel");
    }
    return DM3D_OK;
}

extern "C" int dm3d_wgrad(const dm3d_wgrad_desc* d, void* stream) {
    DM3D_REQUIRE(d != nullptr && d->a && d->g && d->dw, "wgrad: null descriptor / pointer");
    DM3D_REQUIRE(d->ksize == 1 || d->ksize == 3, "wgrad: ksize %d not in {1,3}", d->ksize);
    DM3D_REQUIRE(d->cin > 0 && d->cin % 4 == 0 && d->cout > 0 && d->cout % 4 == 0, "wgrad: cin=%d cout=%d must be multiples of 4", d->cin, d->cout);
    DM3D_REQUIRE(d->batch > 0 && d->in_d > 0 && d->in_h > 0 && d->in_w > 0, "wgrad: non-positive extent");
    DM3D_REQUIRE(dm3d_aligned16(d->a) && dm3d_aligned16(d->g), "wgrad: a / g must be 16-byte aligned");
    DM3D_REQUIRE(d->ksize == 1 || d->per_item_output == 0, "wgrad: per-item outputs exist for ksize 1 only");
    WgradArgs p{};
    p.a = d->a; p.g = d->g; p.dw = d->dw; p.cin = d->cin; p.cout = d->cout; p.ksize = d->ksize;
    p.d = d->in_d; p.h = d->in_h; p.w = d->in_w;
    const long vox = (long)d->in_d * d->in_h * d->in_w;
    int items = 1;
    if (d->ksize == 1 && d->per_item_output) {          // `batch` independent contractions (attention: one per sample)
        items = d->batch;
        p.batch = 1;
        p.a_bs = d->stride_a; p.g_bs = d->stride_g; p.dw_bs = d->stride_dw;
        p.chunks = (vox + 31) / 32;
    } else {                                             // one contraction over every voxel of every sample
        p.batch = d->batch;
        if (d->ksize == 1) { p.d = (int)1; p.h = 1; p.w = 1; }
        p.chunks = (vox * d->batch + 31) / 32;
        if (d->ksize == 1) { DM3D_REQUIRE(vox * d->batch < (1l << 31), "wgrad: too many rows"); p.d = (int)(vox * d->batch); p.batch = 1; }
    }
    p.ntr = d->ksize == 3 ? 9 : 1;
    const long tiles = (long)((d->cin + 63) / 64) * ((d->cout + 63) / 64) * p.ntr * items;
    long ns = (2048 + tiles - 1) / tiles;                // aim at ~2048 workgroups, every K slice >= 4 chunks
    if (ns > p.chunks / 4) ns = p.chunks / 4;
    if (ns < 1) ns = 1;
    if (ns > 1024) ns = 1024;
    p.nsplit = (int)ns;
    const long gz = (long)p.nsplit * p.ntr * items;
    DM3D_REQUIRE(gz <= 65535, "wgrad: grid.z %ld too large", gz);
    dim3 grid((unsigned)((d->cin + 63) / 64), (unsigned)((d->cout + 63) / 64), (unsigned)gz);
    hipLaunchKernelGGL(wgrad_f32_kernel, grid, dim3(256), 0, TR_ST, p);
    return dm3d_launch_check("wgrad_f32_kernel");
}

extern "C" int dm3d_colsum(const float* x, int64_t groups, int64_t rows_per_group, int32_t c, float* out, int64_t ld_out, void* stream) {
    DM3D_REQUIRE(x && out && groups > 0 && groups <= 65535 && rows_per_group > 0 && c > 0 && c % 4 == 0 && ld_out >= c && dm3d_aligned16(x),
                 "colsum: bad arguments (c %% 4 == 0, groups <= 65535)");
    const int c4 = c / 4, lanes = c4 < 256 ? c4 : 256;
    long slabs = rows_per_group / (256 / lanes * 8);
    slabs = slabs < 1 ? 1 : (slabs > 256 ? 256 : slabs);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)slabs, (unsigned)groups), dim3(256), 0, TR_ST, x, (long)rows_per_group, c, out, (long)ld_out);
    return dm3d_launch_check("colsum_kernel");
}

extern "C" int dm3d_flip_transpose(const float* keras_kernel, int32_t taps, int32_t cin, int32_t cout, float* out, void* stream) {
    DM3D_REQUIRE(keras_kernel && out && taps > 0 && cin > 0 && cout > 0 && keras_kernel != out, "flip_transpose: bad arguments");
    hipLaunchKernelGGL(flip_transpose_kernel, dim3(tgrid((long)taps * cin * cout)), dim3(256), 0, TR_ST, keras_kernel, taps, cin, cout, out);
    return dm3d_launch_check("flip_transpose_kernel");
}

extern "C" int dm3d_layernorm_bwd(const float* x, int64_t rows, int32_t c, float eps, const float* gamma, const float* dy, float* dx,
                                  float* dgamma, float* dbeta, void* stream) {
    DM3D_REQUIRE(x && gamma && dy && dx && dgamma && dbeta && rows > 0, "layernorm_bwd: null argument");
    DM3D_REQUIRE(c > 0 && c % 4 == 0 && c <= 1024, "layernorm_bwd: c=%d must be a multiple of 4 and <= 1024", c);
    DM3D_REQUIRE(dm3d_aligned16(x) && dm3d_aligned16(dy) && dm3d_aligned16(dx) && dm3d_aligned16(gamma), "layernorm_bwd: pointers must be 16-byte aligned");
    long g = (rows + 3) / 4;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)g), dim3(256), 0, TR_ST, x, (long)rows, c, eps, gamma, dy, dx, dgamma, dbeta);
    return dm3d_launch_check("layernorm_bwd_kernel");
}

extern "C" int dm3d_softmax_bwd(const float* p, float* dp, int64_t rows, int32_t cols, int64_t ld, float scale, void* stream) {
    DM3D_REQUIRE(p && dp && rows > 0 && cols > 0 && ld >= cols, "softmax_bwd: bad arguments");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, TR_ST, p, dp, (long)rows, cols, (long)ld, scale);
    return dm3d_launch_check("softmax_bwd_kernel");
}

extern "C" int dm3d_act_bwd(const float* ref, const float* dy, float* dx, int64_t n, int32_t act, void* stream) {
    DM3D_REQUIRE(ref && dy && dx && n > 0 && n % 4 == 0, "act_bwd: bad arguments (n %% 4 == 0)");
    DM3D_REQUIRE(act >= DM3D_ACT_NONE && act <= DM3D_ACT_SILU && dm3d_aligned16(ref) && dm3d_aligned16(dy) && dm3d_aligned16(dx), "act_bwd: bad act / alignment");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(tgrid(n / 4)), dim3(256), 0, TR_ST, ref, dy, dx, (long)(n / 4), act);
    return dm3d_launch_check("act_bwd_kernel");
}

extern "C" int dm3d_axpy(float* dst, const float* src, int64_t n, float alpha, void* stream) {
    DM3D_REQUIRE(dst && src && n > 0 && dm3d_aligned16(dst) && dm3d_aligned16(src), "axpy: bad arguments");
    hipLaunchKernelGGL(axpy_kernel, dim3(tgrid(n / 4 + 1)), dim3(256), 0, TR_ST, dst, src, (long)(n / 4), (long)n, alpha);
    return dm3d_launch_check("axpy_kernel");
}

extern "C" int dm3d_fill(float* dst, int64_t n, float value, void* stream) {
    DM3D_REQUIRE(dst && n > 0 && dm3d_aligned16(dst), "fill: bad arguments");
    hipLaunchKernelGGL(fill_kernel, dim3(tgrid(n / 4 + 1)), dim3(256), 0, TR_ST, dst, (long)(n / 4), (long)n, value);
    return dm3d_launch_check("fill_kernel");
}

extern "C" int dm3d_transpose(const float* src, int32_t rows, int32_t cols, int64_t ld_src, int64_t stride_src, float* dst, int64_t ld_dst,
                              int64_t stride_dst, int32_t batch, void* stream) {
    DM3D_REQUIRE(src && dst && rows > 0 && cols > 0 && batch > 0 && batch <= 65535 && ld_src >= cols && ld_dst >= rows, "transpose: bad arguments");
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)batch), dim3(256), 0, TR_ST,
                       src, rows, cols, (long)ld_src, (long)stride_src, dst, (long)ld_dst, (long)stride_dst);
    return dm3d_launch_check("transpose_kernel");
}

extern "C" int dm3d_copy_cols(const float* src, int64_t ld_src, int32_t src_off, float* dst, int64_t ld_dst, int32_t dst_off, int64_t rows,
                              int32_t c, int32_t accumulate, void* stream) {
    DM3D_REQUIRE(src && dst && rows > 0 && c > 0 && c % 4 == 0 && src_off >= 0 && dst_off >= 0 && src_off % 4 == 0 && dst_off % 4 == 0 &&
                 ld_src % 4 == 0 && ld_dst % 4 == 0 && ld_src >= src_off + c && ld_dst >= dst_off + c && dm3d_aligned16(src) && dm3d_aligned16(dst),
                 "copy_cols: bad arguments (c, offsets and leading dimensions are multiples of 4)");
    hipLaunchKernelGGL(copy_cols_kernel, dim3(tgrid(rows * (c / 4))), dim3(256), 0, TR_ST, src, (long)ld_src, src_off, dst, (long)ld_dst, dst_off,
                       (long)rows, c / 4, accumulate);
    return dm3d_launch_check("copy_cols_kernel");
}

extern "C" int dm3d_upsample2(const float* src, float* dst, int32_t batch, int32_t d, int32_t h, int32_t w, int32_t c, void* stream) {
    DM3D_REQUIRE(src && dst && batch > 0 && d > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && dm3d_aligned16(src) && dm3d_aligned16(dst), "upsample2: bad arguments");
    hipLaunchKernelGGL(upsample2_kernel, dim3(tgrid((long)batch * 8 * d * h * w * (c / 4))), dim3(256), 0, TR_ST, src, dst, batch, d, h, w, c / 4);
    return dm3d_launch_check("upsample2_kernel");
}

extern "C" int dm3d_sumpool2_add(const float* src, float* dst, int32_t batch, int32_t d, int32_t h, int32_t w, int32_t c, void* stream) {
    DM3D_REQUIRE(src && dst && batch > 0 && d > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0 && dm3d_aligned16(src) && dm3d_aligned16(dst), "sumpool2_add: bad arguments");
    hipLaunchKernelGGL(sumpool2_kernel, dim3(tgrid((long)batch * d * h * w * (c / 4))), dim3(256), 0, TR_ST, src, dst, batch, d, h, w, c / 4);
    return dm3d_launch_check("sumpool2_kernel");
}

extern "C" int dm3d_dilate2(const float* src, float* dst, int32_t batch, int32_t od, int32_t oh, int32_t ow, int32_t id, int32_t ih, int32_t iw,
                            int32_t offz, int32_t offy, int32_t offx, int32_t c, void* stream) {
    DM3D_REQUIRE(src && dst && batch > 0 && od > 0 && oh > 0 && ow > 0 && c > 0 && c % 4 == 0 && dm3d_aligned16(src) && dm3d_aligned16(dst), "dilate2: bad arguments");
    DM3D_REQUIRE(offz >= 0 && offz <= 1 && offy >= 0 && offy <= 1 && offx >= 0 && offx <= 1, "dilate2: offsets must be 0 or 1");
    const long n = (long)batch * id * ih * iw * c;
    hipLaunchKernelGGL(fill_kernel, dim3(tgrid(n / 4 + 1)), dim3(256), 0, TR_ST, dst, n / 4, n, 0.0f);
    hipLaunchKernelGGL(dilate2_kernel, dim3(tgrid((long)batch * od * oh * ow * (c / 4))), dim3(256), 0, TR_ST, src, dst, batch, od, oh, ow, id, ih, iw,
                       offz, offy, offx, c / 4);
    return dm3d_launch_check("dilate2_kernel");
}

extern "C" int dm3d_scatter_add_rows(const float* src, const int32_t* idx, int32_t rows, int32_t c, float* table, int32_t table_rows, void* stream) {
    DM3D_REQUIRE(src && idx && table && rows > 0 && c > 0 && table_rows > 0, "scatter_add_rows: bad arguments");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(tgrid((long)rows * c)), dim3(256), 0, TR_ST, src, idx, rows, c, table, table_rows);
    return dm3d_launch_check("scatter_add_rows_kernel");
}

extern "C" int dm3d_q_sample(const float* latents, const float* noise, const int32_t* t, const float* sqrt_alpha_bar,
                             const float* sqrt_one_minus_alpha_bar, int32_t timesteps, float* out, int32_t batch, int64_t per_sample, void* stream) {
    DM3D_REQUIRE(latents && noise && t && sqrt_alpha_bar && sqrt_one_minus_alpha_bar && out && timesteps > 0, "q_sample: null argument");
    DM3D_REQUIRE(batch > 0 && batch <= 65535 && per_sample > 0 && per_sample % 4 == 0, "q_sample: batch=%d per_sample=%lld", batch, (long long)per_sample);
    DM3D_REQUIRE(dm3d_aligned16(latents) && dm3d_aligned16(noise) && dm3d_aligned16(out), "q_sample: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(q_sample_kernel, dim3(tgrid(per_sample / 4, 256), (unsigned)batch), dim3(256), 0, TR_ST, latents, noise, t, sqrt_alpha_bar,
                       sqrt_one_minus_alpha_bar, timesteps, out, (long)(per_sample / 4));
    return dm3d_launch_check("q_sample_kernel");
}

extern "C" int dm3d_mse_loss_grad(const float* pred, const float* noise, int64_t n, double inv_divisor, double* loss, float* dpred, void* stream) {
    DM3D_REQUIRE(pred && noise && loss && n > 0 && n % 4 == 0 && dm3d_aligned16(pred) && dm3d_aligned16(noise) && dm3d_aligned16(dpred),
                 "mse_loss_grad: bad arguments (n %% 4 == 0)");
    hipLaunchKernelGGL(mse_loss_grad_kernel, dim3(tgrid(n / 4, 1024)), dim3(256), 0, TR_ST, pred, noise, (long)(n / 4), inv_divisor, loss, dpred);
    return dm3d_launch_check("mse_loss_grad_kernel");
}

extern "C" int dm3d_adam(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2, float eps, void* stream) {
    DM3D_REQUIRE(w && g && m && v && n > 0, "adam: bad arguments");
    hipLaunchKernelGGL(adam_kernel, dim3(tgrid(n)), dim3(256), 0, TR_ST, w, g, m, v, (long)n, lr_t, beta1, beta2, eps);
    return dm3d_launch_check("adam_kernel");
}
