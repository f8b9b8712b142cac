/* dm3d.h — C ABI of libdm3d_hip.so: the MI355X (gfx950) kernels behind the 3D latent-diffusion denoising path.
 *
 * The reference (aayush9400/3D-Condtional-Stable-Diffusion) has no FFI / plugin boundary of its own: the path sits
 * behind Keras layer objects (networks/conditional_dm3d.py:324-415 build_model, :517-575 sample/generate).  This
 * header is therefore the build-defined replacement boundary: one entry point per stock
 * TensorFlow/Keras op family the reference's hot path executes.  Each entry cites the reference call it replaces.
 *
 * Conventions
 *   - plain C: raw device pointers, explicit sizes, a hipStream_t passed as void*; no torch types.
 *   - every function enqueues on the caller's stream and returns without synchronising (graph-capturable);
 *     nothing allocates, frees or retains pointers.  The caller owns every buffer.
 *   - return 0 on success, a negative DM3D_E* code otherwise; dm3d_last_error() gives the thread-local text.
 *   - all tensors float32, activations NDHWC (Keras channels_last), pointers 16-byte aligned.
 */
#ifndef DM3D_H
#define DM3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DM3D_VERSION 111          /* major*100 + minor; the descriptor structs grew in 101 (w_layout), 102 (scratch), 103 (skip_*),
                                     104 (dm3d_attention), 105 (x1_fmt / out_fmt / post_*), 106 (ddpm seed_dev; conv/gemm range_flag; the
                                     training entries), 107 (conv wpk_f8: a float8 cross-term form, removed again in 109), 108 (conv wpk_wino: the Winograd-x form), 109 (wpk_f8 and
                                     dm3d_pack_weights_h3f8 are gone; the Winograd-x image pairs its taps differently; dm3d_mlp_fused; conv skip_wpk_frag, gn_stats; dm3d_groupnorm_finalize2), 110 (dm3d_attn_front), 111 (conv split_counters: the Cin split
                                     meets inside the launch; dm3d_conv_split_counter_words): a host built against an older header must be rebuilt */

#define DM3D_OK            0
#define DM3D_EINVAL       -1      /* bad argument (shape, alignment, null pointer) */
#define DM3D_EUNSUPPORTED -2      /* valid request the kernels do not implement */
#define DM3D_EHIP         -3      /* a HIP runtime call failed (text holds hipGetErrorString) */

#define DM3D_ACT_NONE 0
#define DM3D_ACT_RELU 1
#define DM3D_ACT_SILU 2

/* arithmetic of the contraction kernels.
 *   F32: v_mfma_f32_32x32x2_f32, exact float32 products and accumulation (157 TFLOP/s peak).
 *   H3 : every float32 operand x is split on the fly into two float16 terms x = hi + lo (|x - hi - lo| <= 2^-22 |x|),
 *        and a.b is evaluated as ah.bh + ah.bl + al.bh with three v_mfma_f32_32x32x16_f16 passes accumulating in
 *        float32 (dropped term al.bl <= 2^-22 |a.b|): float32-grade results at 1/3 of the 16-bit MFMA rate.  Weights
 *        are pre-scaled by 2^w_exp (undone exactly in the epilogue) so their lo terms stay normal float16 numbers;
 *        activations are clamped to +-65504 before the split. */
#define DM3D_PREC_F32 0
#define DM3D_PREC_H3  1

/* storage formats of a [rows][k] matrix (same 4 bytes per element, leading dimensions always counted in elements):
 *   F32: row-major float32.
 *   H2 : the float16 hi/lo split of DM3D_PREC_H3 made once by the producer instead of by every consumer: each run of 16
 *        consecutive k of a row is one 64-byte record [hi k0-7 | hi k8-15 | lo k0-7 | lo k8-15] — exactly the LDS record
 *        of the H3 kernels, so their staging degenerates to 16-byte copies.  ld % 16 == 0. */
#define DM3D_FMT_F32 0
#define DM3D_FMT_H2  1

/* padded extents of packed weights */
#define DM3D_COUT_PAD 64
#define DM3D_CIN_PAD  16

int         dm3d_version(void);
const char* dm3d_last_error(void);
/* 1 when a gfx950 device is visible to the calling process, 0 otherwise (never fails). */
int         dm3d_device_ok(void);

/* ---- weights ------------------------------------------------------------------------------------------------
 * Keras kernels are [kd,kh,kw,Cin,Cout] (Conv3D, conditional_dm3d.py:257-259) or [in,out] (Dense, :251; taps=1).
 * The kernels consume them as [taps][CoutPad][CinPad] (K contiguous per output channel, zero padded to
 * DM3D_COUT_PAD / DM3D_CIN_PAD) so that one (tap, Cin-chunk) slice is a dense LDS image.
 * in_scale (optional, [cin]) multiplies the rows — used to fold an inference BatchNormalization that directly
 * precedes a 1x1 conv into its weights (CrossAttentionBlock norm -> proj_in, :187-188). */
int64_t dm3d_packed_weight_elems(int32_t taps, int32_t cin, int32_t cout);
int     dm3d_pack_weights(const float* keras_kernel, int32_t taps, int32_t cin, int32_t cout,
                          const float* in_scale, float* packed, void* stream);

/* H3 image: [CoutPad/64][CinPad/16][taps][64][32 halfs] = per output channel one 64-byte LDS record (hi c0-7, hi c8-15,
 * lo c0-7, lo c8-15; the four 16-byte slots XOR-swizzled by the row), weights multiplied by 2^w_exp before the split.
 * dm3d_packed_weight_h3_bytes gives the buffer size. */
int64_t dm3d_packed_weight_h3_bytes(int32_t taps, int32_t cin, int32_t cout);
int     dm3d_pack_weights_h3(const float* keras_kernel, int32_t taps, int32_t cin, int32_t cout, int32_t w_exp,
                             const float* in_scale, void* packed, void* stream);

/* UpSample (UpSampling3D(2) + Conv3D k3, conditional_dm3d.py:288-296): a 3x3x3 conv on a nearest-2x upsampled tensor equals,
 * per output parity (a,b,c), a 2x2x2 conv on the low-resolution tensor whose taps are sums of the k3 taps that read the same
 * source voxel — 8 taps instead of 27.  These pack the [3,3,3,Cin,Cout] kernel into the 8 parity images that
 * dm3d_conv3d_ndhwc expects in wpk when upsample == 1 (for H3, choose w_exp from the summed taps: up to 8x the largest tap). */
int64_t dm3d_packed_weight_up_elems(int32_t cin, int32_t cout);
int     dm3d_pack_weights_up(const float* keras_kernel, int32_t cin, int32_t cout, float* packed, void* stream);
int64_t dm3d_packed_weight_up_h3_bytes(int32_t cin, int32_t cout);
int     dm3d_pack_weights_up_h3(const float* keras_kernel, int32_t cin, int32_t cout, int32_t w_exp, void* packed, void* stream);
/* (the _up_h3 and _convt_h3 packers write whichever layout dm3d_conv_weight_layout names for their conv, see below) */

/* Conv3DTranspose(k=4, strides=2, padding="same") kernels are [4,4,4,Cout,Cin] in Keras; packed as 8 parity images of a
 * taps=8 conv like the UpSample case (same buffer sizes: dm3d_packed_weight_up_elems / _up_h3_bytes with cin, cout). */
int     dm3d_pack_weights_convt(const float* keras_kernel, int32_t cin, int32_t cout, float* packed, void* stream);
int     dm3d_pack_weights_convt_h3(const float* keras_kernel, int32_t cin, int32_t cout, int32_t w_exp, void* packed, void* stream);

/* Second H3 weight layout, DM3D_WL_PAIR: the image read by the v_mfma_f32_16x16x32_f16 conv kernels (every k3/s1 conv — round 3: also cout <= 32, the narrow column forms —,
 * UpSample, Conv3DTranspose): taps padded to a multiple of 4 with zeros and consumed two at a time, rows permuted inside each
 * group of 16 output channels so that the operand reads are bank-conflict free.  mode 0: plain [taps,Cin,Cout] kernel;
 * mode 1: UpSample [3,3,3,Cin,Cout] -> 8 parity images of 8 taps; mode 2: Conv3DTranspose [4,4,4,Cout,Cin] -> 8 parity images.
 * dm3d_packed_weight_h3p_bytes is the size of ONE image (multiply by 8 for modes 1 and 2, where taps must be 8). */
#define DM3D_WL_TAP   0
#define DM3D_WL_PAIR  1
int64_t dm3d_packed_weight_h3p_bytes(int32_t taps, int32_t cin, int32_t cout);
int     dm3d_pack_weights_h3p(const float* keras_kernel, int32_t taps, int32_t cin, int32_t cout, int32_t w_exp,
                              const float* in_scale, void* packed, int32_t mode, void* stream);
/* image for dm3d_conv_desc.wpk_wino from a [3,3,3,Cin,Cout] kernel: the DM3D_WL_PAIR record format, per 16-channel chunk 20 steps
 * (5 pairs of (dz, dy) taps — (dz, 0) | (dz, 1) for dz = 0, 1, 2; (0, 2) | zero; (1, 2) | (2, 2) —) x 4 transform terms u0 = g0, u1 = (g0+g1+g2)/2, u2 = (g0-g1+g2)/2, u3 = g2 of the x taps */
int64_t dm3d_packed_weight_h3w_bytes(int32_t cin, int32_t cout);
int     dm3d_pack_weights_h3w(const float* keras_kernel, int32_t cin, int32_t cout, int32_t w_exp, const float* in_scale, void* packed,
                              void* stream);
/* the layout dm3d_conv3d_ndhwc wants in wpk for a DM3D_PREC_H3 conv of this geometry (what w_layout must say) */
int32_t dm3d_conv_weight_layout(int32_t ksize, int32_t stride, int32_t upsample, int32_t transpose, int32_t cout);
/* image of a 1x1 kernel [cin, cout] for dm3d_conv_desc.skip_wpk (two 16-channel chunks per MFMA k-step) */
int64_t dm3d_packed_weight_skip_h3p_bytes(int32_t cin, int32_t cout);
int     dm3d_pack_weights_skip_h3p(const float* keras_kernel, int32_t cin, int32_t cout, int32_t w_exp, void* packed, void* stream);
/* the same kernel as MFMA operand fragments (same byte size): [cout tile of 64][pair of 16-channel chunks][16-column tile][hi | lo][lane][16 B],
 * for dm3d_conv_desc.skip_wpk_frag — the Winograd-x form reads its skip weights with plain coalesced loads, without LDS */
int     dm3d_pack_weights_skip_h3f(const float* keras_kernel, int32_t cin, int32_t cout, int32_t w_exp, void* packed, void* stream);

/* ---- Conv3D(padding="same") as implicit GEMM on MFMA ---------------------------------------------------------
 * Replaces layers.Conv3D for k=3/s=1 (:257-259, :348-353, :412-414), k=3/s=2 (DownSample :274-285, TF SAME pad 0
 * before / 1 after on even sizes), k=1 (ResidualBlock skip :245-248) and UpSampling3D(2)+Conv3D (UpSample :288-296,
 * upsample=1 reads the low-resolution tensor at index>>1).  Fused around it:
 *   prologue  x -> silu(x*pro_scale[c] + pro_shift[c])   (inference BatchNormalization + swish, :255-256, 262-263,
 *             410-411), applied before zero padding;
 *   also      k=4/s=2 (VQ-VAE Encoder downsampling, vqvae3d_monai.py:266-273; SAME pad 1/1) and its transpose
 *             (Decoder, :373-377): Conv3DTranspose k4 s2 is evaluated per output parity as a 2x2x2 conv on the input
 *             (parity 0 reads sources {i-1,i} with taps w[3],w[1]; parity 1 reads {i,i+1} with w[2],w[0]);
 *   concat    channels of x1 then x2 (layers.Concatenate(axis=-1)([x, skip]), :396) without materialising it;
 *   epilogue  + bias[co] + vec[row(b)][co] (the time-embedding Dense broadcast-add, :250-253, 260) then optional
 *             ReLU, then + res (layers.Add with the residual, :268). */
typedef struct dm3d_conv_desc {
    const float* x1;            /* [batch, in_d, in_h, in_w, c1] */
    const float* x2;            /* optional second input, same spatial shape, c2 channels (NULL if c2 == 0) */
    int32_t c1, c2;             /* c1 % 4 == 0; with x2: c1 % 16 == 0 and c2 % 4 == 0 */
    int32_t batch;
    int32_t in_d, in_h, in_w;   /* physical extent of x1/x2 */
    int32_t upsample;           /* 1: convolve the nearest-2x upsampled tensor (stride must be 1) */
    int32_t ksize;              /* 1 or 3; 4 with stride 2 (and with transpose) */
    int32_t stride;             /* 1 or 2 */
    const float* wpk;           /* dm3d_pack_weights[_h3] output for (ksize^3, c1+c2, cout); upsample: dm3d_pack_weights_up[_h3] */
    const float* bias;          /* [cout] or NULL */
    const float* pro_scale;     /* [c1+c2] or NULL (both or neither) */
    const float* pro_shift;
    const float* vec;           /* [rows, vec_ld] or NULL */
    const int32_t* vec_idx;     /* [batch] row of vec per sample, or NULL for row = sample index */
    int32_t vec_ld;
    int32_t relu;               /* 1: ReLU after bias/vec, before res */
    const float* res;           /* [batch, out_d, out_h, out_w, cout] or NULL */
    float* out;                 /* [batch, out_d, out_h, out_w, cout], out = ceil(in*(upsample?2:1)/stride) */
    int32_t cout;
    int32_t precision;          /* DM3D_PREC_F32: wpk from dm3d_pack_weights; DM3D_PREC_H3: wpk from dm3d_pack_weights_h3 */
    int32_t w_exp;              /* H3 only: the power-of-two exponent the weights were packed with */
    /* autoencoder bracket (networks/vqvae3d_monai.py): */
    const float* prelu_alpha;   /* keras PReLU() with its default full-shape slope [out_d,out_h,out_w,cout] (shared by the batch),
                                   applied after bias/vec/relu and before res: v = v > 0 ? v : alpha*v; or NULL */
    int32_t relu_out;           /* 1: ReLU after the residual add (VQVAEResidualUnit: ReLU(x + PReLU(BN(conv)))) */
    int32_t transpose;          /* 1: Conv3DTranspose(k=4, strides=2, padding="same"): out = 2*in; wpk from
                                   dm3d_pack_weights_convt[_h3]; ksize must be 4, stride 2, no upsample */
    int64_t pro_batch_stride;   /* elements between consecutive samples' pro_scale / pro_shift vectors: 0 = one vector for
                                   the batch (folded BatchNormalization), c1+c2 = per-sample vectors (GroupNormalization,
                                   written by dm3d_groupnorm_finalize) */
    int32_t w_layout;           /* H3 only: DM3D_WL_TAP or DM3D_WL_PAIR, must equal dm3d_conv_weight_layout(...) */
    /* ResidualBlock skip path fused into this launch (DM3D_PREC_H3, DM3D_WL_PAIR, ksize 3, stride 1, no upsample only):
       out += Conv3D(cout, 1)(concat(skip_x1, skip_x2)) on the raw tensors (no prologue), i.e. layers.Add()([x, residual]) with
       residual = layers.Conv3D(width, kernel_size=1)(x) (conditional_dm3d.py:243-248, 268) without a launch, a tensor or a
       residual read of its own.  skip_wpk from dm3d_pack_weights_skip_h3p, packed with THIS conv's w_exp; add the skip bias into
       bias.  All NULL / 0 when unused. */
    const float* skip_x1; const float* skip_x2; int32_t skip_c1, skip_c2; const void* skip_wpk;
    /* A tensor with exactly one consumer (ResidualBlock: conv1 -> BatchNormalization -> swish -> conv2, :255-267) can skip the
       float32 round trip: the producer applies the consumer's folded norm + swish once per element and stores DM3D_FMT_H2
       (post_scale / post_shift per output channel, out_fmt = DM3D_FMT_H2), the consumer (x1_fmt = DM3D_FMT_H2, no x2, no
       prologue) stages plain copies.  k3 / stride 1 / DM3D_WL_PAIR convs only; an H2 output additionally needs extents that
       are whole 4x8x8 bricks and cout % 64 == 0.  Zero / NULL when unused. */
    int32_t x1_fmt, out_fmt;
    const float* post_scale; const float* post_shift;
    void* scratch;              /* optional workspace (16-byte aligned) of scratch_bytes bytes, or NULL: see split_counters below.
                                   dm3d_conv_scratch_bytes(d) says how much a descriptor can use (0: it never splits). */
    int64_t scratch_bytes;
    /* DM3D_PREC_H3 range guard.  An H3 consumer clamps float32 operands to the float16 range (+-65504) before the hi/lo split — a
       silent difference from the reference's float32 arithmetic if a value ever got there.  With range_flag set, this launch
       writes 1 to *range_flag when any |output value| exceeds range_limit (0 = 65504; hosts pass the smaller bound that also
       covers a consumer's folded norm: (65504 - max|shift|) / max|scale|).  The host reads the flag once per generate() / forward and raises instead of returning clamped results
       (rerun with DM3D_PREC_F32).  NULL: no check. */
    int32_t* range_flag; float range_limit;
    /* Optional weight image of the Winograd F(2,3)-along-x form of a DM3D_PREC_H3 k3 / stride-1 conv with cout > 32 (dm3d_pack_weights_h3w,
       packed with THIS conv's w_exp): two neighbouring outputs of a row from four transformed inputs — 36 instead of 54 MFMA k-steps per
       output pair, same split-float16 products and float32 accumulation (results differ from the direct form in the last bits only).
       The kernel uses it when the volume is whole 8x8x8 bricks, Cin >= 32, a fused skip conv is short and there are enough bricks (a launch of
       at most 128 of its work items with Cin >= 256 splits Cin two ways: split_counters below); one
       persistent workgroup per CU walks the list of (brick, column tile, Cin part) items
       (dm3d_conv_tile_form() == 10); otherwise wpk serves the launch as before.  The transformed inputs are up to 2 max|x|: producers of
       such a conv must keep |x| <= 32752 (pass range_limit <= 32752 to them).  NULL: never. */
    const void* wpk_wino;
    /* Optional second image of the fused skip conv's kernel (dm3d_pack_weights_skip_h3f, THIS conv's w_exp), beside skip_wpk: with it the
       Winograd-x form also serves launches that carry a skip conv — as a register-direct tail between an item's chunk loop and its
       epilogue (no LDS: such launches are persistent like the others); without it they stay on the direct kernel.  NULL: never. */
    const void* skip_wpk_frag;
    /* Optional: fused GroupNormalization statistics of the OUTPUT tensor: gn_stats[batch][slots][cout][2] float32 = partial (sum, sum of
       squares) per (sample, channel) over the voxels of a slot, slots = ceil(voxels / 64), dm3d_groupnorm_partials_bytes() bytes — the
       input dm3d_groupnorm_finalize2 needs to turn this tensor into a consumer's per-sample scale / shift, without a pass of its own over the
       tensor.  The 16x16x32 kernels' full-brick epilogue stores a wave's sums as the slot of its z-slice while it stores the output (no
       atomics; every slot is written); behind any other kernel or form the library runs dm3d_groupnorm_partials on the finished output
       itself.  Float32 output only (cout % 4 == 0, 16-byte aligned buffer).  NULL: none. */
    float* gn_stats;
    /* Cin split of small grids (DM3D_PREC_H3, DM3D_WL_PAIR).  A conv whose grid would leave most of the chip idle (small batches; the 8^3
       level at B = 32) runs up to 16 workgroups per tile (brick x 64 output channels), each contracting a share of the input channels.  The
       parts meet INSIDE the launch: each stores its raw accumulator tiles into `scratch`, draws a ticket from the tile's word of
       split_counters, and the part that draws the last one sums all parts in part order (results do not depend on timing) and applies the
       epilogue — every epilogue form, no zero fill, no atomic adds on the output, no second launch.  split_counters: at least
       dm3d_conv_split_counter_words(d) int32 words, ZERO before the first launch that uses them; every launch leaves them zero, so one
       buffer serves every conv of a stream.  Without split_counters (NULL / 0) no conv splits; with them a conv that wants to split
       requires scratch of dm3d_conv_scratch_bytes(d) bytes (DM3D_EINVAL otherwise). */
    int32_t* split_counters; int32_t split_counter_words;
} dm3d_conv_desc;

int     dm3d_conv3d_ndhwc(const dm3d_conv_desc* d, void* stream);
int64_t dm3d_conv_scratch_bytes(const dm3d_conv_desc* d);
int32_t dm3d_conv_split_counter_words(const dm3d_conv_desc* d);     /* words of split_counters a descriptor of this shape can use (0: it never splits) */
/* Which tile form of the 16x16x32 conv kernels serves this descriptor: 8 (8 z-slices per brick, 512 threads, one workgroup per CU — launches
 * with enough bricks to give every CU two such workgroups in turn), 4 (4 slices, 256 threads, two workgroups per CU: small grids, the parity
 * form, launches with a fused skip conv, Cout <= 32), 10 (the Winograd-x form: wpk_wino given and eligible; conv3d_igemm_h3w<MODE>), 0 (another kernel).  Profiling
 * harnesses use it to name the instantiation a launch runs (rocprofv3 lists conv3d_igemm_h3v3<KS, MODE, TD, NCT> and conv3d_igemm_h3w<MODE>). */
int32_t dm3d_conv_tile_form(const dm3d_conv_desc* d);

/* ---- Dense / einsum contractions: out[b][m][n] = act(alpha * sum_k A[b][m][k]*B[b][n][k] + bias) + res -------
 * Both operands K-contiguous ("TN").  Replaces layers.Dense on the last axis (:131-137, 164-169, 251, 301-304, 313),
 * the 1x1 Conv3D projections of CrossAttentionBlock (:129-130) and the two attention einsums "blc,bLc->blL" /
 * "blL,bLc->blc" (:177-180; U:51-61) as batched calls (stride_* in elements; 0 broadcasts an operand). */
typedef struct dm3d_gemm_desc {
    const float* a; int64_t lda; int64_t stride_a;
    const float* b; int64_t ldb; int64_t stride_b;
    float* out;     int64_t ldo; int64_t stride_o;
    int32_t m, n, k, batch;     /* k % 4 == 0, lda/ldb % 4 == 0 */
    float alpha;
    const float* bias;          /* [n] (or [m] when bias_along_m) or NULL */
    int32_t bias_along_m;
    int32_t act;                /* DM3D_ACT_* applied after bias */
    const float* res; int64_t ldr; int64_t stride_r;   /* added after act, or NULL */
    int32_t precision;          /* DM3D_PREC_F32 (all formats must be DM3D_FMT_F32) or DM3D_PREC_H3 */
    int32_t a_fmt, b_fmt;       /* H3: DM3D_FMT_F32 (split while staging) or DM3D_FMT_H2 (pre-split); k % 16 == 0 */
    int32_t out_fmt;            /* H3: DM3D_FMT_F32 or DM3D_FMT_H2 (n % 16 == 0, ldo % 16 == 0); res is always float32 */
    const float* res2;          /* optional second float32 residual, same ldr / stride_r as res (needs res) */
    int32_t* range_flag; float range_limit;     /* H3 only: as in dm3d_conv_desc */
} dm3d_gemm_desc;

int dm3d_gemm_tn(const dm3d_gemm_desc* d, void* stream);
/* Up to 4 independent DM3D_PREC_H3 contractions with identical operand formats in ONE launch (grid.z enumerates the problems'
 * batches).  The attention blocks' GEMMs are small (m = B*L rows, one workgroup per CU each); issuing the independent ones
 * together (q|k, v^T, q2 and the MLP hidden layer; the two score products; the two P.V products) fills the chip. */
int dm3d_gemm_tn_group(const dm3d_gemm_desc* descs, int32_t count, void* stream);

/* ---- the MLP of a CrossAttentionBlock in one launch (round 4): out = Dense_1(relu(Dense_0(x))) + res + res2
 * (conditional_dm3d.py:132-133 `keras.Sequential([Dense(units * 4, relu), Dense(units)])` applied at :194 with the residual adds of
 * :193-195).  DM3D_PREC_H3 arithmetic; x [m][units] in DM3D_FMT_H2 (ldx in 4-byte elements, a multiple of 16); w0 / w1 = the images
 * dm3d_pack_mlp_weights makes of the DM3D_FMT_H2 matrices [4 units][units] (which = 0) and [units][4 units] (which = 1), same byte size:
 * every (32-row tile, 16-k record, hi | lo) operand fragment contiguous in lane order, so that the kernel streams weights with plain
 * coalesced loads; biases and residuals float32 (16-byte aligned, ldr % 4 == 0); out float32 or DM3D_FMT_H2.  The 4 units-wide hidden
 * activation stays in LDS.  units == 256 only (DM3D_EINVAL otherwise: issue the two dm3d_gemm_tn calls instead). */
int dm3d_pack_mlp_weights(const void* w_h2, int32_t units, int32_t which, void* tiled, void* stream);
typedef struct dm3d_mlp_desc {
    const void* x; int64_t ldx;
    const void* w0; const float* b0;
    const void* w1; const float* b1;
    const float* res; const float* res2; int64_t ldr;      /* optional float32 residuals [m][units] (res2 needs res) */
    void* out; int64_t ldo; int32_t out_fmt;
    int32_t m, units;
    int32_t* range_flag; float range_limit;                /* as in dm3d_gemm_desc */
    /* optional tail (ABI 110): out = relu(W2 . a3 + b2) + res3 with a3 = the result above, which is then NOT stored (out is float32
     * [m][units]) — the block's proj_out and its residual add (conditional_dm3d.py:195).  w2: dm3d_pack_front_weights image of W2[units][units] */
    const void* w2; const float* b2;
    const float* res3; int64_t ldr3;
} dm3d_mlp_desc;
int dm3d_mlp_fused(const dm3d_mlp_desc* d, void* stream);

/* ---- front half of a CrossAttentionBlock in one launch (csrc/dm3d_attn_front_h3.hip; reference networks/conditional_dm3d.py:186-193,
 * 163-170):  y = relu(x . W_in^T + b_in)  (float32 out: the residual of the self-attention pass; an inference BatchNormalization in front
 * is folded into W_in / b_in by the caller),  n_i = LayerNormalization_i(y) for the block's three norms,  q|k = n1 . W_qk^T + b_qk,
 * v^T = (n1 . W_v^T + b_v)^T,  q2 = n2 . W_qk[0:units]^T + b_qk[0:units],  n3 as is — all but y in DM3D_FMT_H2; n1 and n2 never leave
 * the CU.  Replaces dm3d_gemm_tn (proj_in) + dm3d_layernorm3_h2 + a dm3d_gemm_tn_group of three.  units == 256, m % 64 == 0,
 * DM3D_PREC_H3 arithmetic.  Weights are operand-fragment images made once by dm3d_pack_front_weights from the DM3D_FMT_H2 rows
 * W[n][units] (dm3d_split_h2 of the [n][units] float32 weight; n = 256 for w_in / w_v, 512 for w_qk: query rows, then key rows);
 * same byte count as the source. */
typedef struct dm3d_attn_front_desc {
    const float* x; int64_t ldx;                           /* [m][units] float32 */
    const void* w_in; const float* b_in;
    const void* w_qk; const float* b_qk;                   /* b_qk: [2 units] */
    const void* w_v; const float* b_v;
    const float* g1; const float* be1;                     /* LayerNormalization (gamma, beta) of norm1 / norm2 / norm3, [units] each */
    const float* g2; const float* be2;
    const float* g3; const float* be3;
    float eps;
    float* y; int64_t ldy;                                 /* [m][units] float32 */
    void* qk; int64_t ldqk;                                /* DM3D_FMT_H2 [m][2 units] */
    void* vt; int64_t ldvt;                                /* DM3D_FMT_H2 [units][m] */
    void* q2; int64_t ldq2;                                /* DM3D_FMT_H2 [m][units] */
    void* n3; int64_t ldn3;                                /* DM3D_FMT_H2 [m][units] */
    int32_t m, units;
    int32_t* range_flag; float range_limit;                /* as in dm3d_gemm_desc */
} dm3d_attn_front_desc;
int dm3d_pack_front_weights(const void* w_h2, int32_t n, int32_t units, void* tiled, void* stream);
int dm3d_attn_front(const dm3d_attn_front_desc* d, void* stream);

/* dst(H2) = split(src * 2^exp2): one-time conversion of static operands (weights, context keys/values). k % 16 == 0 is
 * not required of src: columns k..round_up(k,16) of dst are zero filled; ld_dst % 16 == 0, ld_dst >= round_up(k,16). */
int dm3d_split_h2(const float* src, int64_t rows, int32_t k, int64_t ld_src, int32_t exp2, void* dst, int64_t ld_dst,
                  void* stream);

/* ---- LayerNormalization (eps passed; Keras default 1e-3) ------------------------------------------------------
 * CrossAttentionBlock normalises the same tensor three times (norm1/2/3, :191-193): one pass computes the row
 * statistics once and writes up to three affine outputs.  c % 4 == 0, c <= 1024. */
int dm3d_layernorm3(const float* x, int64_t rows, int32_t c, float eps,
                    const float* g1, const float* b1, float* o1,
                    const float* g2, const float* b2, float* o2,
                    const float* g3, const float* b3, float* o3, void* stream);
/* same with the outputs written in DM3D_FMT_H2 (c % 16 == 0): they only feed Dense layers */
int dm3d_layernorm3_h2(const float* x, int64_t rows, int32_t c, float eps,
                       const float* g1, const float* b1, void* o1,
                       const float* g2, const float* b2, void* o2,
                       const float* g3, const float* b3, void* o3, void* stream);

/* ---- GroupNormalization(groups, epsilon) statistics (the variant the reference keeps commented out, conditional_dm3d.py:77,
 * 254, 261, 409; keras default epsilon 1e-3).  Per-sample statistics cannot be folded at load time, so two small launches
 * turn a tensor into the per-(sample, channel) scale / shift that the conv prologue (pro_batch_stride = c) or
 * dm3d_affine_act_batched then applies, fused with swish:
 *   dm3d_groupnorm_stats:    acc[b][chan_off + c][0..1] += (sum, sum of squares) of x[b, :, c] in float64 (x [batch, voxels, c];
 *                            call once per concatenated input with its channel offset);  acc must be zero on entry.
 *   dm3d_groupnorm_finalize: group moments from acc (float64), scale[b][c] = gamma[c]*rstd, shift[b][c] = beta[c] - mean*scale;
 *                            acc is zeroed again for the next use. */
int dm3d_groupnorm_stats(const float* x, int32_t batch, int64_t voxels, int32_t c, double* acc, int32_t c_total,
                         int32_t chan_off, void* stream);
int dm3d_groupnorm_finalize(double* acc, int32_t batch, int64_t voxels, int32_t c_total, int32_t groups, float eps,
                            const float* gamma, const float* beta, float* scale, float* shift, void* stream);
/* the same from per-TENSOR partial statistics of up to two concatenated inputs: part [batch][slots][c][2] float32, slots = ceil(voxels / 64),
 * each slot the (sum, sum of squares) of its voxels per channel — what dm3d_conv_desc.gn_stats or dm3d_groupnorm_partials left there.
 * Float64 sums over the slots in a fixed order (no atomics anywhere on this path); nothing is cleared: a tensor with several consumers is
 * summed once. */
int64_t dm3d_groupnorm_partials_bytes(int32_t batch, int64_t voxels, int32_t c);
int dm3d_groupnorm_partials(const float* x, int32_t batch, int64_t voxels, int32_t c, float* part, void* stream);
int dm3d_groupnorm_finalize2(const float* part1, int32_t c1, const float* part2, int32_t c2, int32_t batch, int64_t voxels, int32_t groups,
                             float eps, const float* gamma, const float* beta, float* scale, float* shift, void* stream);
/* y[b][r][c] = act(x[b][r][c]*scale[b][c] + shift[b][c]) : dm3d_affine_act with per-sample vectors. c % 4 == 0. */
int dm3d_affine_act_batched(const float* x, float* y, int32_t batch, int64_t rows_per_sample, int32_t c, const float* scale,
                            const float* shift, int32_t act, void* stream);

/* ---- tf.nn.softmax(scores, -1) in place, one wavefront per row (shuffle reductions) (:178; U:56) ------------ */
int dm3d_softmax_rows(float* s, int64_t rows, int32_t cols, int64_t ld, void* stream);
/* same, result left in place in DM3D_FMT_H2 (cols % 16 == 0, ld % 16 == 0; rows longer than 1024 take a three-pass streaming
 * form): it only feeds the P.V contraction */
int dm3d_softmax_rows_h2(float* s, int64_t rows, int32_t cols, int64_t ld, void* stream);

/* ---- single-head attention  out = softmax(q k^T * scale) v (+ res)  per sample (:163-184; U:47-61) in one call: the score
 * product, the row softmax and the P.V product that a host otherwise issues itself, on caller-provided scratch.  L = D*H*W
 * flattened tokens, single head (the reference never uses num_heads > 1).
 *   q   [batch, lq, c]       row stride ldq
 *   k   [batch | 1, lk, c]   row stride ldk; stride_k = 0 broadcasts one context's keys to the whole batch
 *   vt  [batch | 1, c, lk]   the value tensor TRANSPOSED (it is the K-contiguous operand of P.V); row stride ldv, stride_vt
 *   out [batch, lq, c]       row stride ldo; res: optional float32 tensor with out's layout, added to the result
 *   fmt DM3D_FMT_F32 (precision F32 or H3) or DM3D_FMT_H2 (precision H3): format of q, k and vt; out and res are float32.
 *   scratch: >= dm3d_attention_workspace_bytes(batch, lq, lk) bytes, 16-byte aligned (the [batch, lq, lk] probabilities). */
typedef struct dm3d_attention_desc {
    const float* q;  int64_t ldq;
    const float* k;  int64_t ldk; int64_t stride_k;
    const float* vt; int64_t ldv; int64_t stride_vt;
    float* out;      int64_t ldo;
    const float* res;
    int32_t batch, lq, lk, c;
    float scale;                /* units^-0.5 in the reference */
    int32_t precision, fmt;
} dm3d_attention_desc;
int64_t dm3d_attention_workspace_bytes(int32_t batch, int32_t lq, int32_t lk);
int     dm3d_attention(const dm3d_attention_desc* d, void* scratch, void* stream);
/* With DM3D_FMT_H2 operands, c == 256, lq % 128 == 0 and lk % 32 == 0 (the reference's 8^3 attention level: L = 512, units 256) the
 * call above is ONE fused launch: K / V^T stream through LDS in 32-key tiles, the scores and an online softmax (row maximum and sum kept
 * per query, accumulator rescaled when the maximum moves) live in registers, the [batch, lq, lk] probabilities never reach HBM and
 * scratch is not touched (it may be NULL).  Other shapes run as score product + row softmax + P.V on scratch.
 * dm3d_attention_group: up to 4 passes of identical (batch, lq, lk, scale) — the self- and the cross-attention pass of a
 * CrossAttentionBlock — in one grid (the three-launch form, when needed, runs them one after the other on the shared scratch). */
int     dm3d_attention_group(const dm3d_attention_desc* descs, int32_t count, void* scratch, void* stream);

/* ---- y = act(x*scale[c] + shift[c]) over the last axis (inference BatchNormalization of AttentionBlock, U:45;
 * swish of the time embedding, :250); scale/shift may be NULL (identity). */
int dm3d_affine_act(const float* x, float* y, int64_t rows, int32_t c, const float* scale, const float* shift,
                    int32_t act, void* stream);

/* *flag = 1 if any |x[i]| > limit (the H3 range guard for a tensor no dm3d kernel produced, e.g. the caller's x_t). n % 4 == 0. */
int dm3d_range_check(const float* x, int64_t n, float limit, int32_t* flag, void* stream);

/* ---- DDPM posterior step: DiffusionModel.sample + the loop body of generate (:517-548, 571-573) --------------
 * Coefficients are gathered at t[b] and combined in float32 in the reference's order.
 *   mode 0 (sample):   mean_out = posterior mean, var_out[b] = posterior variance (no clip, no noise).
 *   mode 1 (generate): x <- clip(mean,-1,1) + sqrt(max(var,1e-20)) * z, z = noise (if given) or Philox N(0,1)
 *                      keyed by (seed, t[b], element); z = 0 where t[b] == 0. */
typedef struct dm3d_ddpm_desc {
    float* x;                   /* [batch, per_sample] x_t (updated in place in mode 1) */
    const float* eps;           /* predicted noise */
    const float* noise;         /* optional injected z, same shape */
    int32_t batch; int64_t per_sample;     /* per_sample % 4 == 0 */
    const int32_t* t;           /* [batch] device */
    int32_t timesteps;
    const float *beta, *sqrt_alpha, *alpha_bar, *alpha_bar_prev, *sqrt_alpha_bar, *sqrt_alpha_bar_prev,
                *sqrt_one_minus_alpha_bar;     /* [timesteps] device (Betas, :215-235) */
    uint64_t seed;
    int32_t mode;
    float* mean_out; float* var_out;
    const uint64_t* seed_dev;   /* optional: the Philox key is read from device memory instead of `seed`, so one captured step graph
                                   serves every seed (the host rewrites the scalar between chains).  t[b] is clamped to
                                   [0, timesteps) before any table is indexed. */
} dm3d_ddpm_desc;

int dm3d_ddpm_update(const dm3d_ddpm_desc* d, void* stream);

/* p[i] = max(p[i] + delta, 0) (the loop counter of generate kept on the device so a captured step replays unchanged; it
 * saturates at 0, so a step issued past the end of a chain never indexes row -1 of a table). */
int dm3d_add_i32(int32_t* p, int32_t n, int32_t delta, void* stream);
/* x ~ N(0,1) from Philox keyed by (seed, stream_id): the x_T draw of generate (:555). n % 4 == 0. */
int dm3d_randn(float* x, int64_t n, uint64_t seed, uint32_t stream_id, void* stream);
/* out[r][:] = table[idx[r]][:] (tf.gather / Embedding lookup, :358, 518-538). c % 4 == 0. */
int dm3d_gather_rows(const float* table, int32_t table_rows, const int32_t* idx, float* out, int32_t rows,
                     int32_t c, void* stream);

/* ---- VectorQuantizer.get_code_indices (vqvae3d_monai.py:164-177): idx[r] = argmin_k (|z_r|^2 + esq[k] - 2*sim[r][k]), float32 in
 * that order, lowest index on ties.  sim = z.E [rows, k] comes from dm3d_gemm_tn; esq[k] = |e_k|^2.  The quantised vectors are
 * then dm3d_gather_rows(E^T, idx). */
int dm3d_vq_assign(const float* z, int64_t rows, int32_t d, const float* sim, int32_t k, const float* esq, int32_t* idx,
                   void* stream);

/* ================= training (DiffusionModel.train_step, conditional_dm3d.py:471-510; compile() at main_conditional_dm.py:149-154) ==========
 * The forward pass of training reuses the entries above in DM3D_PREC_F32 (gradients span too many octaves for the float16 split);
 * data gradients of Conv3D / Dense are those same entries on flipped / transposed weights (dm3d_flip_transpose + dm3d_pack_weights).
 * What follows is what training needs in addition.  All tensors float32; "+=" outputs are accumulated into (the caller zeroes
 * gradient buffers once per step: a weight used twice, or a tensor with two consumers, collects both contributions). */

/* BatchNormalization(training=True) (network(..., training=True), :493): per-channel batch statistics over (B,D,H,W).
 * dm3d_groupnorm_stats (above) accumulates per-(sample, channel) sums into acc [batch][c][2] (float64, zero on entry; call it once per
 * concatenated input with its channel offset); this finishes them: mean, biased variance -> scale = gamma*rstd, shift = beta - mean*scale
 * (the vectors the conv prologue / dm3d_affine_act_cat apply), mean_out / rstd_out kept for the backward pass, and — when
 * moving_mean / moving_var are given — the Keras moving averages moving = moving*momentum + batch*(1 - momentum) (momentum 0.99;
 * unbiased_moving = 1 feeds the variance with Bessel's correction n/(n-1), as tf.nn.fused_batch_norm — what Keras runs for rank-5
 * inputs — does).  acc is zeroed again. */
int dm3d_batchnorm_finalize(double* acc, int32_t batch, int64_t voxels, int32_t c, float eps, const float* gamma, const float* beta,
                            float* scale, float* shift, float* mean_out, float* rstd_out, float* moving_mean, float* moving_var,
                            float momentum, int32_t unbiased_moving, void* stream);
/* y[row][:] = act(concat(x1[row], x2[row]) * scale + shift)  (x2 NULL / c2 0: one input; scale NULL: plain concatenation).
 * y is [rows][c1+c2]: the normalised, activated, concatenated tensor a training step keeps for the weight gradient. */
int dm3d_affine_act_cat(const float* x1, int32_t c1, const float* x2, int32_t c2, int64_t rows, const float* scale, const float* shift,
                        int32_t act, float* y, void* stream);
/* Backward of y = act(BatchNorm_train(concat(x1, x2))) given g = dL/dy [rows][c1+c2]: du = g*act'(x*scale+shift);
 * red[c][2] (float64, zero on entry) receives (sum du, sum du*xhat);  dx (+=) = scale*(du - mean(du) - xhat*mean(du*xhat)) written
 * to dx1 / dx2 (either NULL: no gradient wanted);  dgamma (+=) = sum du*xhat, dbeta (+=) = sum du (both or neither). */
int dm3d_bn_act_bwd(const float* g, const float* x1, int32_t c1, const float* x2, int32_t c2, int64_t rows, const float* scale,
                    const float* shift, const float* mean, const float* rstd, int32_t act, double* red, float* dx1, float* dx2,
                    float* dgamma, float* dbeta, void* stream);

/* Weight gradient of Conv3D(k in {1,3}, stride 1, "same") / Dense in the Keras layout:  dw[tap][ci][co] += sum over samples and voxels
 * of a[voxel + tap - 1][ci] * g[voxel][co]  (a: the layer's input [batch, in_d, in_h, in_w, cin]; g: dL/d output, same extent, cout
 * channels).  A contraction over voxels on v_mfma_f32_32x32x2_f32; partial sums meet through float atomics (dw must hold the running
 * gradient, zero at the start of a step).  ksize 1 covers Dense (rows = batch*in_d*in_h*in_w).  per_item_output = 1 (ksize 1): `batch`
 * independent products with their own output each (strides in elements) — the attention gradients dV = P^T dO, dK = dS^T Q per sample.
 * A stride-2 conv's gradient is this on the dilated output gradient (dm3d_dilate2); an UpSample conv's on the upsampled input. */
typedef struct dm3d_wgrad_desc {
    const float* a; const float* g; float* dw;
    int32_t batch, in_d, in_h, in_w, cin, cout, ksize;
    int32_t per_item_output; int64_t stride_a, stride_g, stride_dw;
} dm3d_wgrad_desc;
int dm3d_wgrad(const dm3d_wgrad_desc* d, void* stream);
/* out[group][:c] += column sums of x over the group's rows (x [groups*rows_per_group][c]): bias gradients (one group) and the gradient
 * of the per-sample time-embedding vector a conv epilogue adds (one group per sample, ld_out = row stride of the vector table). */
int dm3d_colsum(const float* x, int64_t groups, int64_t rows_per_group, int32_t c, float* out, int64_t ld_out, void* stream);
/* Keras kernel [taps][cin][cout] -> [taps][cout][cin] with the taps reversed: the kernel whose stride-1 "same" convolution of dL/dy is
 * dL/dx of the original layer (taps 1: the transpose). */
int dm3d_flip_transpose(const float* keras_kernel, int32_t taps, int32_t cin, int32_t cout, float* out, void* stream);

/* LayerNormalization backward (eps as in the forward): dx += rstd*(dy*gamma - mean(dy*gamma) - xhat*mean(dy*gamma*xhat)),
 * dgamma += sum dy*xhat, dbeta += sum dy.  c % 4 == 0, c <= 1024. */
int dm3d_layernorm_bwd(const float* x, int64_t rows, int32_t c, float eps, const float* gamma, const float* dy, float* dx, float* dgamma,
                       float* dbeta, void* stream);
/* softmax backward in place: dp <- scale * p * (dp - sum_j p_j dp_j) per row; scale = the factor the logits carried (units^-0.5). */
int dm3d_softmax_bwd(const float* p, float* dp, int64_t rows, int32_t cols, int64_t ld, float scale, void* stream);
/* dx = dy * act'(ref): ref = pre-activation (swish) or pre/post alike (ReLU).  dx may alias dy.  n % 4 == 0. */
int dm3d_act_bwd(const float* ref, const float* dy, float* dx, int64_t n, int32_t act, void* stream);
int dm3d_axpy(float* dst, const float* src, int64_t n, float alpha, void* stream);          /* dst += alpha*src */
int dm3d_fill(float* dst, int64_t n, float value, void* stream);
/* dst[b][j][i] = src[b][i][j]: the K-contiguous copies the TN contraction needs of K / V in the attention backward */
int dm3d_transpose(const float* src, int32_t rows, int32_t cols, int64_t ld_src, int64_t stride_src, float* dst, int64_t ld_dst,
                   int64_t stride_dst, int32_t batch, void* stream);
/* dst[r][dst_off + j] = (accumulate ? dst : 0) + src[r][src_off + j] for j < c: a column window of one row-major matrix into another
 * (Concatenate, and the split of its gradient).  c, offsets, leading dimensions % 4 == 0. */
int dm3d_copy_cols(const float* src, int64_t ld_src, int32_t src_off, float* dst, int64_t ld_dst, int32_t dst_off, int64_t rows, int32_t c,
                   int32_t accumulate, void* stream);
/* UpSampling3D(2) materialised (its conv then needs a weight gradient over the upsampled tensor), and its backward (8 children summed, +=) */
int dm3d_upsample2(const float* src, float* dst, int32_t batch, int32_t d, int32_t h, int32_t w, int32_t c, void* stream);
int dm3d_sumpool2_add(const float* src, float* dst, int32_t batch, int32_t d, int32_t h, int32_t w, int32_t c, void* stream);
/* dst [batch, id, ih, iw, c] = 0 except dst[2*o + off] = src[o]: the output gradient of a stride-2 conv spread over the input grid
 * (off = 1 - pad_front per axis), after which data and weight gradients are those of a stride-1 conv. */
int dm3d_dilate2(const float* src, float* dst, int32_t batch, int32_t od, int32_t oh, int32_t ow, int32_t id, int32_t ih, int32_t iw,
                 int32_t offz, int32_t offy, int32_t offx, int32_t c, void* stream);
/* table[idx[r]][:] += src[r][:]  (Embedding gradient, :358) */
int dm3d_scatter_add_rows(const float* src, const int32_t* idx, int32_t rows, int32_t c, float* table, int32_t table_rows, void* stream);
/* noisy = sqrt_alpha_bar[t[b]]*latents + sqrt_one_minus_alpha_bar[t[b]]*noise  (:484-490) */
int dm3d_q_sample(const float* latents, const float* noise, const int32_t* t, const float* sqrt_alpha_bar,
                  const float* sqrt_one_minus_alpha_bar, int32_t timesteps, float* out, int32_t batch, int64_t per_sample, void* stream);
/* *loss += sum((noise - pred)^2) * inv_divisor (float64);  dpred = 2*(pred - noise)*inv_divisor (or NULL).  With
 * inv_divisor = 1/(channels * global_bs * lc^4) this is keras MeanSquaredError(reduction=SUM) / loss_reduction_factor (:496-499). */
int dm3d_mse_loss_grad(const float* pred, const float* noise, int64_t n, double inv_divisor, double* loss, float* dpred, void* stream);
/* keras.optimizers.Adam step over a flat parameter buffer: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; w -= lr_t*m/(sqrt(v)+eps),
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t) (the host supplies it; Keras defaults b1 0.9, b2 0.999, eps 1e-7). */
int dm3d_adam(float* w, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2, float eps, void* stream);

/* ---- HIP graph capture of one denoising step (replaces the eager per-op Python loop of generate, :559-573) -- */
int dm3d_graph_begin(void* stream);
int dm3d_graph_end(void* stream, void** graph_exec_out);
int dm3d_graph_launch(void* graph_exec, void* stream);
int dm3d_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* DM3D_H */
