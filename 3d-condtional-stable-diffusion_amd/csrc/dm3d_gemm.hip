// dm3d_gemm.hip — batched float32 "TN" contraction on v_mfma_f32_32x32x2_f32:
//     out[b][m][n] = act(alpha * sum_k A[b][m][k] * B[b][n][k] + bias) + res[b][m][n]
// Replaces layers.Dense on the last axis, the 1x1 Conv3D projections of the attention blocks and the two attention
// einsums (reference networks/conditional_dm3d.py:129-137, 164-180, 251, 301-304, 313; networks/dm3d.py:46-62).
//
// Workgroup = 256 threads (4 waves), tile 256 (m) x 64 (n), K in chunks of 32.  Both operand tiles are register-
// prefetched one chunk ahead (global loads in flight during the MFMAs) and staged through one LDS buffer.
#include "dm3d_common.h"

namespace {

struct GemmArgs {
    const float* a; long lda, sa;
    const float* b; long ldb, sb;
    float* out; long ldo, so;
    int m, n, k;
    float alpha;
    const float* bias; int bias_m; int act;
    const float* res; long ldr, sr;
    const float* res2;
    int ksplit, batch;          // ksplit > 1: grid.z = batch * ksplit, each part contracts a K range and adds into the zeroed output
};

__global__ __launch_bounds__(256, 2) void gemm_tn_f32(const GemmArgs p) {
    constexpr int CK = 32, LDV = CK + 4, TM = 256, NT = 64, MR = 2, NR = 2;
    constexpr int A_SLOTS = TM * (CK / 4) / 256;   // 8 float4 per thread
    constexpr int B_SLOTS = NT * (CK / 4) / 256;   // 2
    __shared__ __attribute__((aligned(16))) float lds_a[TM * LDV];
    __shared__ __attribute__((aligned(16))) float lds_b[NT * LDV];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l32 = lane & 31;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * NT, bz = blockIdx.z % p.batch, part = blockIdx.z / p.batch;
    const int nck = (p.k + CK - 1) / CK;
    const int k_begin = (int)((long)nck * part / p.ksplit) * CK, k_end = part + 1 == p.ksplit ? p.k : (int)((long)nck * (part + 1) / p.ksplit) * CK;
    const float* A = p.a + (size_t)bz * p.sa;
    const float* B = p.b + (size_t)bz * p.sb;

    const int piece = tid & 7;                      // float4 piece of the 32-wide chunk
    const int row0 = tid >> 3;                      // + j*32
    f32x4 ra[A_SLOTS], rb[B_SLOTS];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto fetch = [&](int k0) {
        const bool kok = k0 + piece * 4 < k_end;
#pragma unroll
        for (int j = 0; j < A_SLOTS; ++j) {
            const int m = m0 + row0 + j * 32;
            ra[j] = (kok && m < p.m) ? *reinterpret_cast<const f32x4*>(A + (size_t)m * p.lda + k0 + piece * 4) : zero4;
        }
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) {
            const int n = n0 + row0 + j * 32;
            rb[j] = (kok && n < p.n) ? *reinterpret_cast<const f32x4*>(B + (size_t)n * p.ldb + k0 + piece * 4) : zero4;
        }
    };

    f32x16 acc[MR][NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.0f;

    const float* a_lds[MR];
    const float* b_lds[NR];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) a_lds[mr] = lds_a + (wave * 64 + mr * 32 + l32) * LDV;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) b_lds[nr] = lds_b + (nr * 32 + l32) * LDV;

    fetch(k_begin);
    for (int k0 = k_begin; k0 < k_end; k0 += CK) {
        __syncthreads();                            // previous chunk's MFMAs have read the LDS tiles
#pragma unroll
        for (int j = 0; j < A_SLOTS; ++j) *reinterpret_cast<f32x4*>(lds_a + (row0 + j * 32) * LDV + piece * 4) = ra[j];
#pragma unroll
        for (int j = 0; j < B_SLOTS; ++j) *reinterpret_cast<f32x4*>(lds_b + (row0 + j * 32) * LDV + piece * 4) = rb[j];
        __syncthreads();
        if (k0 + CK < k_end) fetch(k0 + CK);
        dm3d_mma_step<MR, NR, CK>(acc, a_lds, b_lds, half);
    }

    float* O = p.out + (size_t)bz * p.so;
    const float* R = p.res ? p.res + (size_t)bz * p.sr : nullptr;
    const float* R2 = p.res2 ? p.res2 + (size_t)bz * p.sr : nullptr;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
        const int n = n0 + nr * 32 + l32;
        if (n >= p.n) continue;
        const bool lead = part == 0;                    // split-K: part 0 carries bias and residuals
        const float bn = (p.bias && !p.bias_m && lead) ? p.bias[n] : 0.0f;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wave * 64 + mr * 32 + dm3d_acc_row(r, half);
                if (m < p.m) {
                    float v = acc[mr][nr][r] * p.alpha + bn;
                    if (p.bias && p.bias_m && lead) v += p.bias[m];
                    v = dm3d_act(v, p.act);
                    if (R && lead) v += R[(size_t)m * p.ldr + n];
                    if (R2 && lead) v += R2[(size_t)m * p.ldr + n];
                    if (p.ksplit > 1) unsafeAtomicAdd(&O[(size_t)m * p.ldo + n], v);
                    else O[(size_t)m * p.ldo + n] = v;
                }
            }
        }
    }
}

}  // namespace

int dm3d_gemm_h3_launch(const dm3d_gemm_desc* d, hipStream_t st);     // dm3d_gemm_h3.hip
extern "C" int dm3d_fill(float* dst, int64_t n, float value, void* stream);    // dm3d_train.hip

extern "C" int dm3d_gemm_tn(const dm3d_gemm_desc* d, void* stream) {
    DM3D_REQUIRE(d != nullptr, "gemm: null descriptor");
    DM3D_REQUIRE(d->a && d->b && d->out, "gemm: a/b/out must be non-null");
    DM3D_REQUIRE(d->m > 0 && d->n > 0 && d->k > 0 && d->batch > 0, "gemm: non-positive extent m=%d n=%d k=%d batch=%d",
                 d->m, d->n, d->k, d->batch);
    DM3D_REQUIRE(d->k % 4 == 0 && d->lda % 4 == 0 && d->ldb % 4 == 0, "gemm: k=%d lda=%lld ldb=%lld must be multiples of 4",
                 d->k, (long long)d->lda, (long long)d->ldb);
    DM3D_REQUIRE(d->stride_a % 4 == 0 && d->stride_b % 4 == 0, "gemm: batch strides of a/b must be multiples of 4");
    DM3D_REQUIRE(d->lda >= d->k && d->ldb >= d->k && d->ldo >= d->n, "gemm: leading dimension smaller than the row");
    DM3D_REQUIRE(!d->res || d->ldr >= d->n, "gemm: ldr smaller than n");
    DM3D_REQUIRE(!d->res2 || d->res, "gemm: res2 needs res");
    DM3D_REQUIRE(dm3d_aligned16(d->a) && dm3d_aligned16(d->b), "gemm: a/b must be 16-byte aligned");
    DM3D_REQUIRE(d->act >= DM3D_ACT_NONE && d->act <= DM3D_ACT_SILU, "gemm: unknown act %d", d->act);
    DM3D_REQUIRE(d->batch <= 65535, "gemm: batch %d exceeds grid.z", d->batch);
    DM3D_REQUIRE(d->precision == DM3D_PREC_F32 || d->precision == DM3D_PREC_H3, "gemm: unknown precision %d", d->precision);
    if (d->precision == DM3D_PREC_H3) return dm3d_gemm_h3_launch(d, static_cast<hipStream_t>(stream));
    DM3D_REQUIRE(d->a_fmt == DM3D_FMT_F32 && d->b_fmt == DM3D_FMT_F32 && d->out_fmt == DM3D_FMT_F32,
                 "gemm: the float32 kernel takes float32 operands only");
    GemmArgs a{};
    a.a = d->a; a.lda = d->lda; a.sa = d->stride_a;
    a.b = d->b; a.ldb = d->ldb; a.sb = d->stride_b;
    a.out = d->out; a.ldo = d->ldo; a.so = d->stride_o;
    a.m = d->m; a.n = d->n; a.k = d->k; a.alpha = d->alpha;
    a.bias = d->bias; a.bias_m = d->bias_along_m; a.act = d->act;
    a.res = d->res; a.ldr = d->ldr; a.sr = d->stride_r; a.res2 = d->res2;
    // A long contraction feeding a handful of tiles (the data gradient of ContextMLP's Dense(h*w*d*c): m = batch, n = 128, k = 131072)
    // would leave the chip idle behind a few serial K loops: split K over workgroups that add into the zeroed output (linear epilogues only)
    a.ksplit = 1; a.batch = d->batch;
    const long tiles = (long)((d->m + 255) / 256) * ((d->n + 63) / 64) * d->batch;
    if (tiles < 128 && d->k >= 8192 && d->act == DM3D_ACT_NONE && d->batch == 1 && d->ldo == d->n && d->res != d->out) {
        long ks = 512 / tiles;
        if (ks > d->k / 1024) ks = d->k / 1024;
        if (ks > 1) {
            a.ksplit = (int)ks;
            if (int rc = dm3d_fill(d->out, (int64_t)d->m * d->n, 0.0f, stream)) return rc;
        }
    }
    dim3 grid((unsigned)((d->m + 255) / 256), (unsigned)((d->n + 63) / 64), (unsigned)(d->batch * a.ksplit));
    hipLaunchKernelGGL(gemm_tn_f32, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return dm3d_launch_check("gemm_tn_f32");
}

// ---- dm3d_attention: the three launches of one attention product behind one entry (see include/dm3d.h)
extern "C" int64_t dm3d_attention_workspace_bytes(int32_t batch, int32_t lq, int32_t lk) {
    if (batch <= 0 || lq <= 0 || lk <= 0) return 0;
    return (int64_t)batch * lq * dm3d_round_up(lk, 16) * (int64_t)sizeof(float);
}

int dm3d_attention_fused_launch(const dm3d_attention_desc* descs, int count, hipStream_t st);      // dm3d_attn_h3.hip

extern "C" int dm3d_attention_group(const dm3d_attention_desc* descs, int32_t count, void* scratch, void* stream) {
    DM3D_REQUIRE(descs != nullptr && count >= 1 && count <= 4, "attention_group: count %d not in [1,4]", count);
    const int rc = dm3d_attention_fused_launch(descs, count, static_cast<hipStream_t>(stream));
    if (rc != DM3D_EUNSUPPORTED) return rc;
    for (int i = 0; i < count; ++i)                       // the three-launch form, one pass after the other on the shared scratch
        if (int r = dm3d_attention(&descs[i], scratch, stream)) return r;
    return DM3D_OK;
}

extern "C" int dm3d_attention(const dm3d_attention_desc* d, void* scratch, void* stream) {
    DM3D_REQUIRE(d != nullptr, "attention: null descriptor");
    {
        const int rc = dm3d_attention_fused_launch(d, 1, static_cast<hipStream_t>(stream));
        if (rc != DM3D_EUNSUPPORTED) return rc;
    }
    DM3D_REQUIRE(scratch != nullptr, "attention: this shape runs as score product + softmax + P.V and needs scratch");
    DM3D_REQUIRE(d->q && d->k && d->vt && d->out, "attention: q/k/vt/out must be non-null");
    DM3D_REQUIRE(d->batch > 0 && d->lq > 0 && d->lk > 0 && d->c > 0, "attention: non-positive extent");
    DM3D_REQUIRE(d->fmt == DM3D_FMT_F32 || (d->fmt == DM3D_FMT_H2 && d->precision == DM3D_PREC_H3), "attention: H2 operands need precision H3");
    DM3D_REQUIRE(d->fmt == DM3D_FMT_F32 || d->lk % 16 == 0, "attention: H2 operands need lk %% 16 == 0");
    DM3D_REQUIRE(dm3d_aligned16(scratch), "attention: scratch must be 16-byte aligned");
    const bool h2 = d->fmt == DM3D_FMT_H2;
    const int64_t lds = dm3d_round_up(d->lk, 16);             // row stride of the probabilities
    float* p = static_cast<float*>(scratch);
    dm3d_gemm_desc g{};
    g.a = d->q; g.lda = d->ldq; g.stride_a = (int64_t)d->lq * d->ldq;
    g.b = d->k; g.ldb = d->ldk; g.stride_b = d->stride_k;
    g.out = p; g.ldo = lds; g.stride_o = (int64_t)d->lq * lds;
    g.m = d->lq; g.n = d->lk; g.k = d->c; g.batch = d->batch; g.alpha = d->scale;
    g.precision = d->precision; g.a_fmt = g.b_fmt = d->fmt; g.out_fmt = DM3D_FMT_F32;
    int rc = dm3d_gemm_tn(&g, stream);
    if (rc) return rc;
    rc = h2 ? dm3d_softmax_rows_h2(p, (int64_t)d->batch * d->lq, d->lk, lds, stream)
            : dm3d_softmax_rows(p, (int64_t)d->batch * d->lq, d->lk, lds, stream);
    if (rc) return rc;
    dm3d_gemm_desc v{};
    v.a = p; v.lda = lds; v.stride_a = (int64_t)d->lq * lds;
    v.b = d->vt; v.ldb = d->ldv; v.stride_b = d->stride_vt;
    v.out = d->out; v.ldo = d->ldo; v.stride_o = (int64_t)d->lq * d->ldo;
    v.m = d->lq; v.n = d->c; v.k = d->lk; v.batch = d->batch; v.alpha = 1.0f;
    v.res = d->res; v.ldr = d->ldo; v.stride_r = (int64_t)d->lq * d->ldo;
    v.precision = d->precision; v.a_fmt = h2 ? DM3D_FMT_H2 : DM3D_FMT_F32; v.b_fmt = d->fmt; v.out_fmt = DM3D_FMT_F32;
    return dm3d_gemm_tn(&v, stream);
}
