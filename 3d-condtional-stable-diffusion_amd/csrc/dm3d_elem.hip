// dm3d_elem.hip — the HBM-bound companions of the conv/GEMM kernels: LayerNormalization, row softmax (wavefront
// shuffle reductions), per-channel affine+activation, the DDPM posterior update with in-kernel Philox noise, and small
// index utilities.  All accesses are 16 B per lane, coalesced.
#include "dm3d_h3.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- LayerNormalization: one wavefront per row, row kept in registers, up to three affine outputs ----------------
__global__ __launch_bounds__(256) void layernorm3_kernel(const float* __restrict__ x, long rows, int c, float eps,
                                                         const float* g1, const float* b1, float* o1,
                                                         const float* g2, const float* b2, float* o2,
                                                         const float* g3, const float* b3, float* o3) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = c >> 2;                          // float4 per row, <= 256
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * c);
    f32x4 v[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = lane + j * 64;
        v[j] = (i < nv) ? xr[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    const float mean = wave_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (lane + j * 64 < nv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float dlt = v[j][e] - mean; q += dlt * dlt; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)c + eps);
    const float* gs[3] = {g1, g2, g3};
    const float* bs[3] = {b1, b2, b3};
    float* os[3] = {o1, o2, o3};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!os[k]) continue;
        f32x4* orow = reinterpret_cast<f32x4*>(os[k] + row * c);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = lane + j * 64;
            if (i < nv) {
                const f32x4 g = reinterpret_cast<const f32x4*>(gs[k])[i];
                const f32x4 bb = reinterpret_cast<const f32x4*>(bs[k])[i];
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (v[j][e] - mean) * rstd * g[e] + bb[e];
                orow[i] = y;
            }
        }
    }
}

// ---- softmax over the last axis, in place: one wavefront per row, max and sum by cross-lane shuffles --------------
template <int NE>   // elements per lane kept in registers (cols <= 64*NE)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, long rows, int cols, long ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = s + row * ld;
    float v[NE];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int i = lane + j * 64;
        v[j] = (i < cols) ? r[i] : -INFINITY;
        mx = fmaxf(mx, v[j]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        v[j] = (lane + j * 64 < cols) ? __expf(v[j] - mx) : 0.f;
        sum += v[j];
    }
    const float inv = 1.0f / wave_sum(sum);
#pragma unroll
    for (int j = 0; j < NE; ++j) {
        const int i = lane + j * 64;
        if (i < cols) r[i] = v[j] * inv;
    }
}

// long rows: three passes through L1/L2 (a row of a few thousand floats stays cached)
__global__ __launch_bounds__(256) void softmax_rows_stream_kernel(float* __restrict__ s, long rows, int cols, long ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = s + row * ld;
    float mx = -INFINITY;
    for (int i = lane; i < cols; i += 64) mx = fmaxf(mx, r[i]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int i = lane; i < cols; i += 64) sum += __expf(r[i] - mx);
    const float inv = 1.0f / wave_sum(sum);
    for (int i = lane; i < cols; i += 64) r[i] = __expf(r[i] - mx) * inv;
}

// ---- LayerNormalization with DM3D_FMT_H2 outputs: lane owns 8 consecutive channels per group (c <= 1024 -> <= 2 groups) ----
__global__ __launch_bounds__(256) void layernorm3_h2_kernel(const float* __restrict__ x, long rows, int c, float eps,
                                                            const float* g1, const float* b1, _Float16* o1,
                                                            const float* g2, const float* b2, _Float16* o2,
                                                            const float* g3, const float* b3, _Float16* o3) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int ng = c >> 3;                          // 8-channel groups per row
    const float* xr = x + row * c;
    f32x4 v[2][2];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int g = lane + j * 64;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        v[j][0] = g < ng ? *reinterpret_cast<const f32x4*>(xr + g * 8) : z;
        v[j][1] = g < ng ? *reinterpret_cast<const f32x4*>(xr + g * 8 + 4) : z;
#pragma unroll
        for (int e = 0; e < 4; ++e) s += v[j][0][e] + v[j][1][e];
    }
    const float mean = wave_sum(s) / (float)c;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (lane + j * 64 < ng) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d0 = v[j][0][e] - mean, d1 = v[j][1][e] - mean;
                q += d0 * d0 + d1 * d1;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)c + eps);
    const float* gs[3] = {g1, g2, g3};
    const float* bs[3] = {b1, b2, b3};
    _Float16* os[3] = {o1, o2, o3};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (!os[k]) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int g = lane + j * 64;
            if (g < ng) {
                f32x4 y0, y1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y0[e] = (v[j][0][e] - mean) * rstd * gs[k][g * 8 + e] + bs[k][g * 8 + e];
                    y1[e] = (v[j][1][e] - mean) * rstd * gs[k][g * 8 + 4 + e] + bs[k][g * 8 + 4 + e];
                }
                h8 hi, lo;
                split8(y0, y1, 65504.0f, 65504.0f, hi, lo);
                _Float16* rec = os[k] + row * c * 2 + (g >> 1) * DM3D_REC;
                *reinterpret_cast<h8*>(rec + (g & 1) * 8) = hi;
                *reinterpret_cast<h8*>(rec + 16 + (g & 1) * 8) = lo;
            }
        }
    }
}

// ---- softmax over the last axis, result left in place in DM3D_FMT_H2.  One wavefront owns a row: every load of the row has
// completed (the max / sum reductions need them) before the first store, so the in-place format change is race free. ----
__global__ __launch_bounds__(256) void softmax_rows_h2_kernel(float* __restrict__ s, long rows, int cols, long ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = s + row * ld;
    const int ng = cols >> 3;
    f32x4 v[2][2];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int g = lane + j * 64;
        const f32x4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        v[j][0] = g < ng ? *reinterpret_cast<const f32x4*>(r + g * 8) : ninf;
        v[j][1] = g < ng ? *reinterpret_cast<const f32x4*>(r + g * 8 + 4) : ninf;
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fmaxf(v[j][0][e], v[j][1][e]));
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[j][0][e] = __expf(v[j][0][e] - mx);       // exp(-inf) = 0 for the padding groups
            v[j][1][e] = __expf(v[j][1][e] - mx);
            sum += v[j][0][e] + v[j][1][e];
        }
    const float inv = 1.0f / wave_sum(sum);
    _Float16* rh = reinterpret_cast<_Float16*>(r);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int g = lane + j * 64;
        if (g < ng) {
            h8 hi, lo;
            split8(v[j][0] * inv, v[j][1] * inv, 65504.0f, 65504.0f, hi, lo);
            _Float16* rec = rh + (g >> 1) * DM3D_REC;
            *reinterpret_cast<h8*>(rec + (g & 1) * 8) = hi;
            *reinterpret_cast<h8*>(rec + 16 + (g & 1) * 8) = lo;
        }
    }
}

// Rows longer than 1024 (attention over 16^3 = 4096 tokens at 64^3 latents): three passes over the row (max, sum, write), the
// row stays in L2.  Lanes g and g^1 share one 64-byte record and sit in the same wave iteration, so both have loaded their
// float32 halves before either stores float16 pieces over them.
__global__ __launch_bounds__(256) void softmax_rows_h2_stream_kernel(float* __restrict__ s, long rows, int cols, long ld) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = s + row * ld;
    const int ng = cols >> 3, iters = (ng + 63) / 64;
    float mx = -INFINITY;
    for (int j = 0; j < iters; ++j) {
        const int g = lane + j * 64;
        if (g < ng) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(r + g * 8), b = *reinterpret_cast<const f32x4*>(r + g * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fmaxf(a[e], b[e]));
        }
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = 0; j < iters; ++j) {
        const int g = lane + j * 64;
        if (g < ng) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(r + g * 8), b = *reinterpret_cast<const f32x4*>(r + g * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) sum += __expf(a[e] - mx) + __expf(b[e] - mx);
        }
    }
    const float inv = 1.0f / wave_sum(sum);
    _Float16* rh = reinterpret_cast<_Float16*>(r);
    for (int j = 0; j < iters; ++j) {
        const int g = lane + j * 64;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
        if (g < ng) {
            a = *reinterpret_cast<const f32x4*>(r + g * 8);
            b = *reinterpret_cast<const f32x4*>(r + g * 8 + 4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a[e] = __expf(a[e] - mx) * inv;
            b[e] = __expf(b[e] - mx) * inv;
        }
        h8 hi, lo;
        split8(a, b, 65504.0f, 65504.0f, hi, lo);
        __builtin_amdgcn_sched_barrier(0);                      // every load of this iteration is above, every store below
        if (g < ng) {
            _Float16* rec = rh + (g >> 1) * DM3D_REC;
            *reinterpret_cast<h8*>(rec + (g & 1) * 8) = hi;
            *reinterpret_cast<h8*>(rec + 16 + (g & 1) * 8) = lo;
        }
    }
}

// ---- y = act(x*scale[c] + shift[c]) ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, float* __restrict__ y, long n4,
                                                         int c4, const float* scale, const float* shift, int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        const int cc = (int)(i % c4);
        if (scale) {
            const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[cc];
            const f32x4 sh = reinterpret_cast<const f32x4*>(shift)[cc];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dm3d_act(v[e], act);
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

// ---- GroupNormalization statistics: per-(sample, channel) sum and sum of squares, float32 partials per thread, float64 atomics
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const float* __restrict__ x, long voxels, int c, double* __restrict__ acc,
                                                              int c_total, int chan_off) {
    // grid: (voxel slabs, batch).  Thread -> channel quad q = tid % (c/4) (clamped), voxel lane vl = tid / (c/4); every
    // thread owns the same 4 channels for all its voxels, so sums stay in registers; LDS reduces threads sharing a quad.
    extern __shared__ float sred[];             // [256][8]
    const int c4 = c >> 2;
    const int tid = threadIdx.x, b = blockIdx.y;
    const int lanes_per_vox = c4 < 256 ? c4 : 256;            // quads handled per pass
    const int vox_par = 256 / lanes_per_vox;                  // voxels processed in parallel by the block
    const int q0 = tid % lanes_per_vox, vl = tid / lanes_per_vox;
    const long slab = (voxels + gridDim.x - 1) / gridDim.x;
    const long v0 = (long)blockIdx.x * slab, v1 = v0 + slab < voxels ? v0 + slab : voxels;
    for (int q = q0; q < c4; q += lanes_per_vox) {            // c > 1024 loops
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = s;
        if (vl < vox_par) {
            for (long v = v0 + vl; v < v1; v += vox_par) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(x + ((size_t)b * voxels + v) * c + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[e] += t[e]; ss[e] += t[e] * t[e]; }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { sred[tid * 8 + e] = s[e]; sred[tid * 8 + 4 + e] = ss[e]; }
        __syncthreads();
        if (vl == 0) {                                          // one thread per quad folds the vox_par partials
            double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
            for (int k = 0; k < vox_par; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) { ds[e] += sred[(k * lanes_per_vox + q0) * 8 + e]; dq[e] += sred[(k * lanes_per_vox + q0) * 8 + 4 + e]; }
            double* a = acc + ((size_t)b * c_total + chan_off + q * 4) * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) { atomicAdd(a + 2 * e, ds[e]); atomicAdd(a + 2 * e + 1, dq[e]); }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void groupnorm_finalize_kernel(double* __restrict__ acc, long voxels, int c_total, int groups, float eps,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ scale, float* __restrict__ shift) {
    const int b = blockIdx.x, gc = c_total / groups;
    __shared__ double gmean[64], grstd[64];
    for (int g = threadIdx.x; g < groups; g += 256) {
        double s = 0, q = 0;
        for (int i = 0; i < gc; ++i) { s += acc[((size_t)b * c_total + g * gc + i) * 2]; q += acc[((size_t)b * c_total + g * gc + i) * 2 + 1]; }
        const double n = (double)voxels * gc, m = s / n;
        double var = q / n - m * m;
        var = var > 0 ? var : 0;
        gmean[g] = m;
        grstd[g] = 1.0 / sqrt(var + (double)eps);
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c_total; ch += 256) {
        const int g = ch / gc;
        const double sc = (double)gamma[ch] * grstd[g];
        scale[(size_t)b * c_total + ch] = (float)sc;
        shift[(size_t)b * c_total + ch] = (float)((double)beta[ch] - gmean[g] * sc);
        acc[((size_t)b * c_total + ch) * 2] = 0.0;            // ready for the next use (same stream order)
        acc[((size_t)b * c_total + ch) * 2 + 1] = 0.0;
    }
}

// GroupNormalization from per-TENSOR partial statistics (dm3d_conv_desc.gn_stats / dm3d_groupnorm_partials): part[b][slot][c][2] float32
// (sum, sum of squares) over the voxels of slot, nslots = ceil(voxels / 64).  One block per (group, sample): float64 sums over the slots and
// the group's channels — channel ch of concat(x1, x2) lives in part1 (ch < c1) or part2 — then scale / shift of those channels.
__global__ __launch_bounds__(256) void groupnorm_finalize2_kernel(const float* __restrict__ part1, int c1, const float* __restrict__ part2, int c2,
                                                                  long voxels, long nslots, int groups, float eps, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift) {
    const int g = blockIdx.x, b = blockIdx.y, c_total = c1 + c2, gc = c_total / groups;
    __shared__ double red[2][256];
    double s = 0, q = 0;
    // four (sum, sum of squares) pairs per thread in flight, 32-bit index arithmetic: as one dependent 64-bit-indexed load pair per iteration
    // this kernel took 11 us at the 32^3 level (16 serial round trips per thread) — 46 launches per step of the norm="group" U-Net
    const unsigned total = (unsigned)nslots * (unsigned)gc, ugc = (unsigned)gc;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    for (unsigned i0 = threadIdx.x; i0 < total; i0 += 1024u) {
        f32x2 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned i = i0 + 256u * k;
            const bool in = i < total;
            const unsigned ic = in ? i : 0u, slot = ic / ugc;
            const int ch = g * gc + (int)(ic - slot * ugc);
            const float* src = ch < c1 ? part1 + (((size_t)b * nslots + slot) * c1 + ch) * 2 : part2 + (((size_t)b * nslots + slot) * c2 + (ch - c1)) * 2;
            v[k] = *reinterpret_cast<const f32x2*>(src);
            if (!in) v[k] = f32x2{0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { s += (double)v[k][0]; q += (double)v[k][1]; }
    }
    red[0][threadIdx.x] = s; red[1][threadIdx.x] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) { red[0][threadIdx.x] += red[0][threadIdx.x + w]; red[1][threadIdx.x] += red[1][threadIdx.x + w]; }
        __syncthreads();
    }
    const double n = (double)voxels * gc, m = red[0][0] / n;
    double var = red[1][0] / n - m * m;
    var = var > 0 ? var : 0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int i = threadIdx.x; i < gc; i += 256) {
        const int ch = g * gc + i;
        const double sc = (double)gamma[ch] * rstd;
        scale[(size_t)b * c_total + ch] = (float)sc;
        shift[(size_t)b * c_total + ch] = (float)((double)beta[ch] - m * sc);
    }
}

// the stand-alone producer of such partials: block (slot, sample) sums its 64 voxels for every channel
__global__ __launch_bounds__(256) void groupnorm_partials_kernel(const float* __restrict__ x, long voxels, int c, float* __restrict__ part, long nslots) {
    const long slot = blockIdx.x;
    const int b = blockIdx.y;
    const long v0 = slot * 64, v1 = v0 + 64 < voxels ? v0 + 64 : voxels;
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        float s = 0.f, q = 0.f;
        for (long v = v0; v < v1; ++v) { const float t = x[((size_t)b * voxels + v) * c + ch]; s += t; q = fmaf(t, t, q); }
        float* dst = part + (((size_t)b * nslots + slot) * c + ch) * 2;
        dst[0] = s; dst[1] = q;
    }
}

__global__ __launch_bounds__(256) void affine_act_batched_kernel(const float* __restrict__ x, float* __restrict__ y, long per_sample4,
                                                                 int c4, const float* scale, const float* shift, int act) {
    const int b = blockIdx.y;
    const f32x4* xs = reinterpret_cast<const f32x4*>(x) + (size_t)b * per_sample4;
    f32x4* ys = reinterpret_cast<f32x4*>(y) + (size_t)b * per_sample4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < per_sample4; i += (long)gridDim.x * 256) {
        f32x4 v = xs[i];
        const int cc = (int)(i % c4);
        const f32x4 sc = reinterpret_cast<const f32x4*>(scale)[(size_t)b * c4 + cc];
        const f32x4 sh = reinterpret_cast<const f32x4*>(shift)[(size_t)b * c4 + cc];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dm3d_act(fmaf(v[e], sc[e], sh[e]), act);
        ys[i] = v;
    }
}

// ---- Philox4x32-10 + Box-Muller ---------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// four N(0,1) draws for 128-bit counter (i_lo, i_hi, s0, s1) under key seed
__device__ __forceinline__ f32x4 philox_normal4(uint64_t idx, uint32_t s0, uint32_t s1, uint64_t seed) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), s0, s1};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u0 = ((float)c[0] + 0.5f) * 2.3283064365386963e-10f;   // (0,1]: float rounding can reach 1, log(1)=0 is fine
    const float u1 = (float)c[1] * 2.3283064365386963e-10f;
    const float u2 = ((float)c[2] + 0.5f) * 2.3283064365386963e-10f;
    const float u3 = (float)c[3] * 2.3283064365386963e-10f;
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    float s_0, c_0, s_1, c_1;
    sincosf(6.283185307179586f * u1, &s_0, &c_0);
    sincosf(6.283185307179586f * u3, &s_1, &c_1);
    return f32x4{r0 * c_0, r0 * s_0, r1 * c_1, r1 * s_1};
}

__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ x, long n4, uint64_t seed, uint32_t stream_id) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        reinterpret_cast<f32x4*>(x)[i] = philox_normal4((uint64_t)i, stream_id, 0x5eedu, seed);
}

// ---- DDPM posterior (reference conditional_dm3d.py:517-548) and loop body (:571-573), float32 in the written order -----
struct DdpmArgs {
    float* x; const float* eps; const float* noise;
    int batch; long per4;                          // float4 per sample
    const int* t;
    const float *beta, *sqa, *ab, *abp, *sqab, *sqabp, *sq1ab;
    uint64_t seed; int mode;
    float* mean_out; float* var_out;
    const uint64_t* seed_dev; int timesteps;
};

__global__ __launch_bounds__(256) void ddpm_kernel(const DdpmArgs p) {
    const int b = blockIdx.y;
    const int t = min(max(p.t[b], 0), p.timesteps - 1);
    const uint64_t seed = p.seed_dev ? *p.seed_dev : p.seed;
    const float be = p.beta[t], sqa = p.sqa[t], ab = p.ab[t], abp = p.abp[t];
    const float sqab = p.sqab[t], sqabp = p.sqabp[t], sq1ab = p.sq1ab[t];
    const float one_m_ab = __fsub_rn(1.0f, ab), one_m_abp = __fsub_rn(1.0f, abp);
    const float c1 = __fdiv_rn(__fmul_rn(be, sqabp), one_m_ab);             // b*sqab_prev/(1-ab)
    const float c2 = __fdiv_rn(__fmul_rn(one_m_abp, sqa), one_m_ab);        // (1-ab_prev)*sqa/(1-ab)
    const float var = __fdiv_rn(__fmul_rn(one_m_abp, be), one_m_ab);        // (1-ab_prev)*b/(1-ab)
    const float sigma = expf(__fmul_rn(0.5f, logf(fmaxf(var, 1e-20f))));    // tf.exp(0.5*np.log(np.maximum(var,1e-20)))
    if (p.mode == 0 && p.var_out && blockIdx.x == 0 && threadIdx.x == 0) p.var_out[b] = var;
    const long base = (long)b * p.per4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.per4; i += (long)gridDim.x * 256) {
        const f32x4 x = reinterpret_cast<const f32x4*>(p.x)[base + i];
        const f32x4 e = reinterpret_cast<const f32x4*>(p.eps)[base + i];
        f32x4 mean;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x0 = __fdiv_rn(__fsub_rn(x[k], __fmul_rn(sq1ab, e[k])), sqab);
            mean[k] = __fadd_rn(__fmul_rn(c1, x0), __fmul_rn(c2, x[k]));
        }
        if (p.mode == 0) {
            reinterpret_cast<f32x4*>(p.mean_out)[base + i] = mean;
            continue;
        }
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if (t > 0) z = p.noise ? reinterpret_cast<const f32x4*>(p.noise)[base + i]
                               : philox_normal4((uint64_t)(base + i), (uint32_t)t, 0xd1f0u, seed);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // tf.clip_by_value propagates a NaN (conditional_dm3d.py:572); fminf / fmaxf return the other operand for one, which would turn a
            // NaN eps — diverged weights, inf - inf somewhere in the U-Net — into a plausible x = -1 + noise that no later check can see
            const float cl = mean[k] != mean[k] ? mean[k] : fminf(fmaxf(mean[k], -1.0f), 1.0f);
            o[k] = __fadd_rn(cl, __fmul_rn(sigma, z[k]));
        }
        reinterpret_cast<f32x4*>(p.x)[base + i] = o;
    }
}

typedef unsigned int u32x4r __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void range_check_kernel(const float* __restrict__ x, long n4, float limit, int* __restrict__ flag) {
    // |x| compared as integers: for non-negative floats the bit patterns order like the values and every NaN pattern lies above +inf,
    // so a NaN raises the flag too (fmaxf drops NaNs and `amax > limit` is false for one: a NaN produced upstream — NaN or Inf weights of a
    // diverged checkpoint — would otherwise pass unseen)
    unsigned int amax = 0u;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const u32x4r v = reinterpret_cast<const u32x4r*>(x)[i];
        amax = max(max(amax, max(v[0] & 0x7fffffffu, v[1] & 0x7fffffffu)), max(v[2] & 0x7fffffffu, v[3] & 0x7fffffffu));
    }
    if (amax > __float_as_uint(limit)) *flag = 1;
}

__global__ void add_i32_kernel(int* p, int n, int delta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = max(p[i] + delta, 0);
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, int table_rows,
                                                          const int* __restrict__ idx, float* __restrict__ out,
                                                          int rows, int c4) {
    const long n = (long)rows * c4;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int r = (int)(i / c4), cc = (int)(i % c4);
        int src = idx[r];
        src = src < 0 ? 0 : (src >= table_rows ? table_rows - 1 : src);      // clamp like tf.gather on GPU never faults
        reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(table)[(long)src * c4 + cc];
    }
}

// Keras [taps][cin][cout] -> [taps][coutpad][cinpad], optional per-input-channel scale
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, int taps, int cin, int cout,
                                                           int cinpad, int coutpad, const float* in_scale,
                                                           float* __restrict__ out) {
    const long n = (long)taps * coutpad * cinpad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int ci = (int)(i % cinpad);
        const int co = (int)((i / cinpad) % coutpad);
        const int tap = (int)(i / ((long)cinpad * coutpad));
        float v = 0.f;
        if (ci < cin && co < cout) {
            v = w[((long)tap * cin + ci) * cout + co];
            if (in_scale) v *= in_scale[ci];
        }
        out[i] = v;
    }
}

// upsample variant: 8 parity images of a taps=8 conv (see dm3d_up_weight)
__global__ __launch_bounds__(256) void pack_weights_up_kernel(const float* __restrict__ w, int cin, int cout, int cinpad,
                                                              int coutpad, float* __restrict__ out, int convt) {
    const long per = (long)8 * coutpad * cinpad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < 8 * per; i += (long)gridDim.x * 256) {
        const int par = (int)(i / per);
        const long j = i % per;
        const int ci = (int)(j % cinpad);
        const int co = (int)((j / cinpad) % coutpad);
        const int tap = (int)(j / ((long)cinpad * coutpad));
        out[i] = (ci < cin && co < cout) ? (convt ? dm3d_convt_weight(w, cin, cout, par, tap, ci, co)
                                                  : dm3d_up_weight(w, cin, cout, par, tap, ci, co)) : 0.f;
    }
}

// ---- VectorQuantizer.get_code_indices (reference networks/vqvae3d_monai.py:164-177): distances = |z|^2 + |e_k|^2 - 2 z.e_k in
// float32 in that order, argmin with the lowest index on ties.  sim = z.E comes from the GEMM; one wavefront per row. ----
__global__ __launch_bounds__(256) void vq_assign_kernel(const float* __restrict__ z, long rows, int d, const float* __restrict__ sim,
                                                        int k, const float* __restrict__ esq, int* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float zs = 0.f;
    for (int i = lane; i < d; i += 64) { const float v = z[row * d + i]; zs += v * v; }
    zs = wave_sum(zs);
    float best = INFINITY;
    int bi = 0x7fffffff;
    const float* srow = sim + row * k;
    for (int j = lane; j < k; j += 64) {
        const float dist = __fsub_rn(__fadd_rn(zs, esq[j]), __fmul_rn(2.0f, srow[j]));
        if (dist < best) { best = dist; bi = j; }           // ascending j per lane: first minimum kept
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) idx[row] = bi;
}

inline unsigned grid_for(long n, int cap = 256 * 8) {
    long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int dm3d_layernorm3(const float* x, int64_t rows, int32_t c, float eps,
                               const float* g1, const float* b1, float* o1,
                               const float* g2, const float* b2, float* o2,
                               const float* g3, const float* b3, float* o3, void* stream) {
    DM3D_REQUIRE(x && rows > 0, "layernorm: x null or rows <= 0");
    DM3D_REQUIRE(c > 0 && c % 4 == 0 && c <= 1024, "layernorm: c=%d must be a multiple of 4 and <= 1024", c);
    DM3D_REQUIRE(o1 || o2 || o3, "layernorm: no output");
    DM3D_REQUIRE((!o1 || (g1 && b1)) && (!o2 || (g2 && b2)) && (!o3 || (g3 && b3)), "layernorm: output without gamma/beta");
    const void* ptrs[] = {x, g1, b1, o1, g2, b2, o2, g3, b3, o3};
    for (const void* q : ptrs) DM3D_REQUIRE(dm3d_aligned16(q), "layernorm: pointer %p is not 16-byte aligned", q);
    hipLaunchKernelGGL(layernorm3_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, (long)rows, c, eps, g1, b1, o1, g2, b2, o2, g3, b3, o3);
    return dm3d_launch_check("layernorm3_kernel");
}

extern "C" int dm3d_softmax_rows(float* s, int64_t rows, int32_t cols, int64_t ld, void* stream) {
    DM3D_REQUIRE(s && rows > 0 && cols > 0 && ld >= cols, "softmax: bad arguments rows=%lld cols=%d ld=%lld",
                 (long long)rows, cols, (long long)ld);
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (cols <= 64) hipLaunchKernelGGL(softmax_rows_kernel<1>, grid, block, 0, st, s, (long)rows, cols, (long)ld);
    else if (cols <= 256) hipLaunchKernelGGL(softmax_rows_kernel<4>, grid, block, 0, st, s, (long)rows, cols, (long)ld);
    else if (cols <= 512) hipLaunchKernelGGL(softmax_rows_kernel<8>, grid, block, 0, st, s, (long)rows, cols, (long)ld);
    else if (cols <= 1024) hipLaunchKernelGGL(softmax_rows_kernel<16>, grid, block, 0, st, s, (long)rows, cols, (long)ld);
    else hipLaunchKernelGGL(softmax_rows_stream_kernel, grid, block, 0, st, s, (long)rows, cols, (long)ld);
    return dm3d_launch_check("softmax_rows_kernel");
}

extern "C" int dm3d_affine_act(const float* x, float* y, int64_t rows, int32_t c, const float* scale, const float* shift,
                               int32_t act, void* stream) {
    DM3D_REQUIRE(x && y && rows > 0 && c > 0 && c % 4 == 0, "affine_act: bad arguments (c=%d must be a multiple of 4)", c);
    DM3D_REQUIRE((scale == nullptr) == (shift == nullptr), "affine_act: scale and shift go together");
    DM3D_REQUIRE(act >= DM3D_ACT_NONE && act <= DM3D_ACT_SILU, "affine_act: unknown act %d", act);
    DM3D_REQUIRE(dm3d_aligned16(x) && dm3d_aligned16(y) && dm3d_aligned16(scale) && dm3d_aligned16(shift),
                 "affine_act: pointers must be 16-byte aligned");
    const long n4 = (long)rows * (c / 4);
    hipLaunchKernelGGL(affine_act_kernel, dim3(grid_for(n4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n4,
                       c / 4, scale, shift, act);
    return dm3d_launch_check("affine_act_kernel");
}

extern "C" int dm3d_ddpm_update(const dm3d_ddpm_desc* d, void* stream) {
    DM3D_REQUIRE(d != nullptr, "ddpm: null descriptor");
    DM3D_REQUIRE(d->x && d->eps && d->t, "ddpm: x/eps/t must be non-null");
    DM3D_REQUIRE(d->batch > 0 && d->batch <= 65535 && d->per_sample > 0 && d->per_sample % 4 == 0,
                 "ddpm: batch=%d per_sample=%lld (must be a positive multiple of 4)", d->batch, (long long)d->per_sample);
    DM3D_REQUIRE(d->beta && d->sqrt_alpha && d->alpha_bar && d->alpha_bar_prev && d->sqrt_alpha_bar &&
                 d->sqrt_alpha_bar_prev && d->sqrt_one_minus_alpha_bar, "ddpm: a Betas table is null");
    DM3D_REQUIRE(d->mode == 0 || d->mode == 1, "ddpm: mode %d not in {0,1}", d->mode);
    DM3D_REQUIRE(d->timesteps > 0, "ddpm: timesteps=%d", d->timesteps);
    DM3D_REQUIRE(d->mode == 1 || d->mean_out, "ddpm: mode 0 needs mean_out");
    DM3D_REQUIRE(dm3d_aligned16(d->x) && dm3d_aligned16(d->eps) && dm3d_aligned16(d->noise) && dm3d_aligned16(d->mean_out),
                 "ddpm: pointers must be 16-byte aligned");
    DdpmArgs a{};
    a.x = d->x; a.eps = d->eps; a.noise = d->noise; a.batch = d->batch; a.per4 = d->per_sample / 4; a.t = d->t;
    a.beta = d->beta; a.sqa = d->sqrt_alpha; a.ab = d->alpha_bar; a.abp = d->alpha_bar_prev; a.sqab = d->sqrt_alpha_bar;
    a.sqabp = d->sqrt_alpha_bar_prev; a.sq1ab = d->sqrt_one_minus_alpha_bar;
    a.seed = d->seed; a.mode = d->mode; a.mean_out = d->mean_out; a.var_out = d->var_out;
    a.seed_dev = d->seed_dev; a.timesteps = d->timesteps;
    dim3 grid(grid_for(a.per4, 256), (unsigned)d->batch);
    hipLaunchKernelGGL(ddpm_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return dm3d_launch_check("ddpm_kernel");
}

extern "C" int dm3d_range_check(const float* x, int64_t n, float limit, int32_t* flag, void* stream) {
    DM3D_REQUIRE(x && flag && n > 0 && n % 4 == 0 && dm3d_aligned16(x), "range_check: x/flag null, x unaligned or n=%lld not a positive multiple of 4", (long long)n);
    hipLaunchKernelGGL(range_check_kernel, dim3(grid_for(n / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long)(n / 4), limit, flag);
    return dm3d_launch_check("range_check_kernel");
}

extern "C" int dm3d_add_i32(int32_t* p, int32_t n, int32_t delta, void* stream) {
    DM3D_REQUIRE(p && n > 0, "add_i32: bad arguments");
    hipLaunchKernelGGL(add_i32_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream), p, n, delta);
    return dm3d_launch_check("add_i32_kernel");
}

extern "C" int dm3d_randn(float* x, int64_t n, uint64_t seed, uint32_t stream_id, void* stream) {
    DM3D_REQUIRE(x && n > 0 && n % 4 == 0 && dm3d_aligned16(x), "randn: x null/unaligned or n=%lld not a positive multiple of 4", (long long)n);
    hipLaunchKernelGGL(randn_kernel, dim3(grid_for(n / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long)(n / 4), seed, stream_id);
    return dm3d_launch_check("randn_kernel");
}

extern "C" int dm3d_gather_rows(const float* table, int32_t table_rows, const int32_t* idx, float* out, int32_t rows,
                                int32_t c, void* stream) {
    DM3D_REQUIRE(table && idx && out && table_rows > 0 && rows > 0 && c > 0 && c % 4 == 0, "gather_rows: bad arguments");
    DM3D_REQUIRE(dm3d_aligned16(table) && dm3d_aligned16(out), "gather_rows: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((long)rows * (c / 4))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       table, table_rows, idx, out, rows, c / 4);
    return dm3d_launch_check("gather_rows_kernel");
}

extern "C" int64_t dm3d_packed_weight_elems(int32_t taps, int32_t cin, int32_t cout) {
    if (taps <= 0 || cin <= 0 || cout <= 0) return 0;
    return (int64_t)taps * dm3d_round_up(cout, DM3D_COUT_PAD) * dm3d_round_up(cin, DM3D_CIN_PAD);
}

extern "C" int dm3d_pack_weights(const float* keras_kernel, int32_t taps, int32_t cin, int32_t cout, const float* in_scale,
                                 float* packed, void* stream) {
    DM3D_REQUIRE(keras_kernel && packed && taps > 0 && cin > 0 && cout > 0, "pack_weights: bad arguments");
    const int cinpad = (int)dm3d_round_up(cin, DM3D_CIN_PAD), coutpad = (int)dm3d_round_up(cout, DM3D_COUT_PAD);
    const long n = (long)taps * cinpad * coutpad;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), keras_kernel,
                       taps, cin, cout, cinpad, coutpad, in_scale, packed);
    return dm3d_launch_check("pack_weights_kernel");
}

extern "C" int dm3d_layernorm3_h2(const float* x, int64_t rows, int32_t c, float eps,
                                  const float* g1, const float* b1, void* o1,
                                  const float* g2, const float* b2, void* o2,
                                  const float* g3, const float* b3, void* o3, void* stream) {
    DM3D_REQUIRE(x && rows > 0, "layernorm_h2: x null or rows <= 0");
    DM3D_REQUIRE(c > 0 && c % 16 == 0 && c <= 1024, "layernorm_h2: c=%d must be a multiple of 16 and <= 1024", c);
    DM3D_REQUIRE(o1 || o2 || o3, "layernorm_h2: no output");
    DM3D_REQUIRE((!o1 || (g1 && b1)) && (!o2 || (g2 && b2)) && (!o3 || (g3 && b3)), "layernorm_h2: output without gamma/beta");
    const void* ptrs[] = {x, o1, o2, o3};
    for (const void* q : ptrs) DM3D_REQUIRE(dm3d_aligned16(q), "layernorm_h2: pointer %p is not 16-byte aligned", q);
    hipLaunchKernelGGL(layernorm3_h2_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, (long)rows, c, eps, g1, b1, static_cast<_Float16*>(o1), g2, b2, static_cast<_Float16*>(o2), g3, b3,
                       static_cast<_Float16*>(o3));
    return dm3d_launch_check("layernorm3_h2_kernel");
}

extern "C" int dm3d_softmax_rows_h2(float* s, int64_t rows, int32_t cols, int64_t ld, void* stream) {
    DM3D_REQUIRE(s && rows > 0 && cols > 0 && ld >= cols, "softmax_h2: bad arguments");
    DM3D_REQUIRE(cols % 16 == 0 && ld % 16 == 0 && dm3d_aligned16(s),
                 "softmax_h2: cols=%d must be a multiple of 16, ld %% 16 == 0, s 16-byte aligned", cols);
    if (cols > 1024) {
        hipLaunchKernelGGL(softmax_rows_h2_stream_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           s, (long)rows, cols, (long)ld);
        return dm3d_launch_check("softmax_rows_h2_stream_kernel");
    }
    hipLaunchKernelGGL(softmax_rows_h2_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       s, (long)rows, cols, (long)ld);
    return dm3d_launch_check("softmax_rows_h2_kernel");
}

extern "C" int64_t dm3d_packed_weight_up_elems(int32_t cin, int32_t cout) { return 8 * dm3d_packed_weight_elems(8, cin, cout); }

extern "C" int dm3d_pack_weights_up(const float* keras_kernel, int32_t cin, int32_t cout, float* packed, void* stream) {
    DM3D_REQUIRE(keras_kernel && packed && cin > 0 && cout > 0, "pack_weights_up: bad arguments");
    const int cinpad = (int)dm3d_round_up(cin, DM3D_CIN_PAD), coutpad = (int)dm3d_round_up(cout, DM3D_COUT_PAD);
    hipLaunchKernelGGL(pack_weights_up_kernel, dim3(grid_for((long)64 * cinpad * coutpad)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), keras_kernel, cin, cout, cinpad, coutpad, packed, 0);
    return dm3d_launch_check("pack_weights_up_kernel");
}

extern "C" int dm3d_pack_weights_convt(const float* keras_kernel, int32_t cin, int32_t cout, float* packed, void* stream) {
    DM3D_REQUIRE(keras_kernel && packed && cin > 0 && cout > 0, "pack_weights_convt: bad arguments");
    const int cinpad = (int)dm3d_round_up(cin, DM3D_CIN_PAD), coutpad = (int)dm3d_round_up(cout, DM3D_COUT_PAD);
    hipLaunchKernelGGL(pack_weights_up_kernel, dim3(grid_for((long)64 * cinpad * coutpad)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), keras_kernel, cin, cout, cinpad, coutpad, packed, 1);
    return dm3d_launch_check("pack_weights_up_kernel(convt)");
}

extern "C" int dm3d_vq_assign(const float* z, int64_t rows, int32_t d, const float* sim, int32_t k, const float* esq,
                              int32_t* idx, void* stream) {
    DM3D_REQUIRE(z && sim && esq && idx && rows > 0 && d > 0 && k > 0, "vq_assign: bad arguments");
    hipLaunchKernelGGL(vq_assign_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), z,
                       (long)rows, d, sim, k, esq, idx);
    return dm3d_launch_check("vq_assign_kernel");
}

extern "C" int dm3d_groupnorm_stats(const float* x, int32_t batch, int64_t voxels, int32_t c, double* acc, int32_t c_total,
                                    int32_t chan_off, void* stream) {
    DM3D_REQUIRE(x && acc && batch > 0 && batch <= 65535 && voxels > 0 && c > 0 && c % 4 == 0, "groupnorm_stats: bad arguments (c %% 4 == 0)");
    DM3D_REQUIRE(chan_off >= 0 && chan_off % 4 == 0 && chan_off + c <= c_total, "groupnorm_stats: channel window outside c_total");
    DM3D_REQUIRE(dm3d_aligned16(x), "groupnorm_stats: x must be 16-byte aligned");
    const int c4 = c / 4, lanes = c4 < 256 ? c4 : 256;
    long slabs = voxels / (256 / lanes * 16);                // >= 16 voxels per thread
    slabs = slabs < 1 ? 1 : (slabs > 64 ? 64 : slabs);
    hipLaunchKernelGGL(groupnorm_stats_kernel, dim3((unsigned)slabs, (unsigned)batch), dim3(256), 256 * 8 * sizeof(float),
                       static_cast<hipStream_t>(stream), x, (long)voxels, c, acc, c_total, chan_off);
    return dm3d_launch_check("groupnorm_stats_kernel");
}

extern "C" int dm3d_groupnorm_finalize(double* acc, int32_t batch, int64_t voxels, int32_t c_total, int32_t groups, float eps,
                                       const float* gamma, const float* beta, float* scale, float* shift, void* stream) {
    DM3D_REQUIRE(acc && gamma && beta && scale && shift && batch > 0 && voxels > 0, "groupnorm_finalize: bad arguments");
    DM3D_REQUIRE(groups > 0 && groups <= 64 && c_total % groups == 0, "groupnorm_finalize: groups=%d must divide c=%d (<= 64)", groups, c_total);
    hipLaunchKernelGGL(groupnorm_finalize_kernel, dim3((unsigned)batch), dim3(256), 0, static_cast<hipStream_t>(stream), acc,
                       (long)voxels, c_total, groups, eps, gamma, beta, scale, shift);
    return dm3d_launch_check("groupnorm_finalize_kernel");
}

extern "C" int64_t dm3d_groupnorm_partials_bytes(int32_t batch, int64_t voxels, int32_t c) {
    return (batch > 0 && voxels > 0 && c > 0) ? (int64_t)batch * ((voxels + 63) / 64) * c * 2 * (int64_t)sizeof(float) : 0;
}

extern "C" int dm3d_groupnorm_partials(const float* x, int32_t batch, int64_t voxels, int32_t c, float* part, void* stream) {
    DM3D_REQUIRE(x && part && batch > 0 && batch <= 65535 && voxels > 0 && c > 0, "groupnorm_partials: bad arguments");
    const long nslots = (long)((voxels + 63) / 64);
    DM3D_REQUIRE(nslots < (1l << 31), "groupnorm_partials: too many slots");
    hipLaunchKernelGGL(groupnorm_partials_kernel, dim3((unsigned)nslots, (unsigned)batch), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       (long)voxels, c, part, nslots);
    return dm3d_launch_check("groupnorm_partials_kernel");
}

extern "C" int dm3d_groupnorm_finalize2(const float* part1, int32_t c1, const float* part2, int32_t c2, int32_t batch, int64_t voxels, int32_t groups,
                                        float eps, const float* gamma, const float* beta, float* scale, float* shift, void* stream) {
    DM3D_REQUIRE(part1 && c1 > 0 && gamma && beta && scale && shift && batch > 0 && batch <= 65535 && voxels > 0, "groupnorm_finalize2: bad arguments");
    DM3D_REQUIRE((part2 != nullptr) == (c2 > 0) && c2 >= 0, "groupnorm_finalize2: part2 / c2 go together");
    DM3D_REQUIRE(groups > 0 && groups <= 64 && (c1 + c2) % groups == 0, "groupnorm_finalize2: groups=%d must divide c=%d (<= 64)", groups, c1 + c2);
    hipLaunchKernelGGL(groupnorm_finalize2_kernel, dim3((unsigned)groups, (unsigned)batch), dim3(256), 0, static_cast<hipStream_t>(stream), part1, c1,
                       part2, c2, (long)voxels, (long)((voxels + 63) / 64), groups, eps, gamma, beta, scale, shift);
    return dm3d_launch_check("groupnorm_finalize2_kernel");
}

extern "C" int dm3d_affine_act_batched(const float* x, float* y, int32_t batch, int64_t rows_per_sample, int32_t c,
                                       const float* scale, const float* shift, int32_t act, void* stream) {
    DM3D_REQUIRE(x && y && scale && shift && batch > 0 && batch <= 65535 && rows_per_sample > 0 && c > 0 && c % 4 == 0,
                 "affine_act_batched: bad arguments (c %% 4 == 0)");
    DM3D_REQUIRE(act >= DM3D_ACT_NONE && act <= DM3D_ACT_SILU, "affine_act_batched: unknown act %d", act);
    DM3D_REQUIRE(dm3d_aligned16(x) && dm3d_aligned16(y) && dm3d_aligned16(scale) && dm3d_aligned16(shift), "affine_act_batched: pointers must be 16-byte aligned");
    const long per4 = (long)rows_per_sample * (c / 4);
    hipLaunchKernelGGL(affine_act_batched_kernel, dim3(grid_for(per4, 256), (unsigned)batch), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, y, per4, c / 4, scale, shift, act);
    return dm3d_launch_check("affine_act_batched_kernel");
}
