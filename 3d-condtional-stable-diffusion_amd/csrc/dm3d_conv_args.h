// dm3d_conv_args.h — launch arguments shared by the fp32 and the split-fp16 ("H3") implicit-GEMM Conv3d kernels.
#pragma once
#include "dm3d_common.h"

struct ConvArgs {
    const float* x1; const float* x2; int c1, c2;
    int ind, inh, inw;        // physical input extent
    int lgd, lgh, lgw;        // logical extent seen by the conv (2x when upsampling)
    int od, oh, ow;
    int padz, pady, padx;     // zero voxels in front of index 0 per axis (TF SAME; parity mode: 1 - parity bit)
    int os;                   // output stride in the full output tensor (2 in parity mode, else 1)
    int ooz, ooy, oox;        // output offset in the full output tensor (the parity bits)
    int fd, fh, fw;           // extent of the full output tensor (od*os ...)
    int parity;               // 1: blockIdx.z enumerates the 8 output parities of a nearest-2x upsample + k3 conv, each a
                              //    2x2x2 conv on the low-resolution input with pre-summed weights
    long w_parity_stride;     // elements (fp32) / halfs (h3) between the packed weight images of two parities
    const void* wpk; int cinpad, coutpad;
    const float* bias; const float* pscale; const float* pshift; long pro_bstride;
    const float* vec; const int* vec_idx; int vec_ld;
    int relu; const float* res; float* out; int cout;
    const float* prelu; int relu_out;      // PReLU slope [od*os, oh*os, ow*os, cout] before res; ReLU after res
    int bd, bh, bw;           // bricks per volume along d, h, w
    int nchunks;
    int batch;
    float out_scale;          // H3: 2^-w_exp, applied to the accumulator
    int ksplit;               // h3v2: workgroups per brick along Cin (each contracts nchunks / ksplit chunks); > 1: the hand-over form
                              // (dm3d_conv_h3v2_parts.h, split_*): every part stores its raw accumulator tiles into scratch, the part that
                              // draws the last ticket of its tile sums them in part order and runs the epilogue
    int* split_counters;      // one ticket word per tile (brick x column tile x parity): zero before the launch, zero again behind it
    int split_counter_words;
    long split_tile_floats;   // floats of one part's image of one tile in scratch ([tile][part][piece][thread] x 4)
    void* scratch; long scratch_bytes;
    const float* post_scale; const float* post_shift;   // h3v2: out = silu(out*post_scale[c] + post_shift[c]) after everything else
    int out_h2;               // h3v2: store the output in DM3D_FMT_H2 (full bricks, cout % 64 == 0 only)
    int x_h2;                 // h3v2: x1 arrives in DM3D_FMT_H2 (c1 % 16 == 0, no x2, no prologue)
    // h3v2 only: a 1x1 conv over a second (raw, un-normalised) input accumulated into the same tile (ResidualBlock skip path)
    const float* sx1; const float* sx2; int sc1, sc2; const void* swpk; int s_npairs;
    const void* swpk_f;                      // optional: the skip weights as operand fragments (dm3d_pack_weights_skip_h3f): the Winograd-x form's tail   // s_npairs = round_up(sc1+sc2, 32) / 32
    const void* wpk_wino;                    // optional weight image of the Winograd-x form (dm3d_conv_h3w.hip)
    int* range_flag; float range_limit;      // H3 range guard (include/dm3d.h): *range_flag = 1 if any |output| > range_limit
    float* gn_stats;                         // fused GroupNormalization statistics of the OUTPUT (include/dm3d.h): [batch][slots][cout][2] partial (sum, sum of squares)
    int epi_vec4;                            // h3v2: every epilogue operand is 16-byte aligned (cout, vec_ld % 4 == 0): 16-byte epilogue accesses
};

// which tile configuration a (ksize, stride) pair uses
enum { DM3D_CONV_K3S1 = 0, DM3D_CONV_K3S2 = 1, DM3D_CONV_K1 = 2, DM3D_CONV_UP = 3, DM3D_CONV_K4S2 = 4 };

int dm3d_conv_launch_f32(ConvArgs& a, int which, hipStream_t st);
int dm3d_conv_launch_h3(ConvArgs& a, int which, hipStream_t st);
int dm3d_conv_launch_h3v3(ConvArgs& a, int which, hipStream_t st);     // DM3D_WL_PAIR weights; which in {K3S1, UP}: the free-running form (dm3d_conv_h3v3.hip)
struct H3v2Launch { ConvArgs k; bool stats_after; };       // what pre_launch decided: the kernel's own arguments; the stand-alone statistics pass behind it
int dm3d_h3v2_pre_launch(ConvArgs& a, int td, H3v2Launch& L, hipStream_t st, int force_ksplit = 0);    // force_ksplit > 0: the caller's Cin split
int dm3d_h3v2_post_launch(const ConvArgs& a, const H3v2Launch& L, hipStream_t st);
int dm3d_conv_h3v2_ksplit(const ConvArgs& a);            // parts along Cin the direct kernel's 4-slice form would use (1: no split_counters, or the grid is large)
long dm3d_conv_split_tiles(const ConvArgs& a, int td);   // tiles of a launch in td-slice bricks: bricks x column tiles x parities
int dm3d_conv_h3v3_td(const ConvArgs& a);              // z-slices per brick (4 or 8) the free-running kernel takes for this launch
int64_t dm3d_h3v2_skip_image_bytes(int cin, int cout);
int dm3d_pack_skip_h3v2(const float* keras_kernel, int cin, int cout, int w_exp, void* packed, hipStream_t st);
int dm3d_pack_skip_h3f(const float* keras_kernel, int cin, int cout, int w_exp, void* packed, hipStream_t st);      // operand-fragment order       // split factor the launch would choose
int64_t dm3d_h3v2_image_bytes(int taps, int cin, int cout);
int dm3d_pack_h3v2(const float* keras_kernel, int taps, int cin, int cout, int w_exp, const float* in_scale, void* packed, int mode,
                   hipStream_t st);
bool dm3d_conv_h3w_serves(const ConvArgs& a, int which);    // true: the Winograd-x form (wpk_wino, dm3d_conv_h3w.hip) serves this launch
int dm3d_conv_launch_h3w(ConvArgs& a, int which, hipStream_t st);
int dm3d_conv_h3w_ksplit(const ConvArgs& a);             // workgroups per brick along Cin the Winograd form would use (1 or 2)
