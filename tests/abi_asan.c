#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include "dm3d.h"
/* host-side paths of every entry with hostile arguments (no device needed: each must refuse before any launch) */
int main(void) {
    int bad = 0;
    dm3d_conv_desc c; memset(&c, 0, sizeof c);
    bad += dm3d_conv3d_ndhwc(&c, NULL) != 0;
    bad += dm3d_conv3d_ndhwc(NULL, NULL) != 0;
    float* buf = (float*)aligned_alloc(64, 4096);
    c.x1 = buf; c.wpk = buf; c.out = buf; c.c1 = 6; c.batch = 1; c.in_d = c.in_h = c.in_w = 2; c.cout = 4; c.ksize = 3; c.stride = 1;
    bad += dm3d_conv3d_ndhwc(&c, NULL) != 0;            /* c1 not a multiple of 4 */
    c.c1 = 4; c.ksize = 5;
    bad += dm3d_conv3d_ndhwc(&c, NULL) != 0;            /* unsupported kernel size */
    c.ksize = 3; c.x1 = (const float*)((char*)buf + 4);
    bad += dm3d_conv3d_ndhwc(&c, NULL) != 0;            /* misaligned pointer */
    dm3d_gemm_desc g; memset(&g, 0, sizeof g);
    bad += dm3d_gemm_tn(&g, NULL) != 0;
    bad += dm3d_gemm_tn_group(&g, 9, NULL) != 0;
    dm3d_attention_desc a; memset(&a, 0, sizeof a);
    bad += dm3d_attention(&a, buf, NULL) != 0;
    dm3d_ddpm_desc d; memset(&d, 0, sizeof d);
    bad += dm3d_ddpm_update(&d, NULL) != 0;
    bad += dm3d_layernorm3(NULL, 1, 4, 1e-3f, 0, 0, 0, 0, 0, 0, 0, 0, 0, NULL) != 0;
    bad += dm3d_softmax_rows(NULL, 1, 4, 4, NULL) != 0;
    bad += dm3d_pack_weights_h3(NULL, 27, 4, 4, 0, NULL, NULL, NULL) != 0;
    bad += dm3d_pack_weights_skip_h3p(buf, 4, 4, 1000, buf, NULL) != 0;
    printf("refused %d of 13; sizes %lld %lld %lld %lld; last error: %s\n", bad, (long long)dm3d_packed_weight_h3_bytes(27, 64, 64),
           (long long)dm3d_packed_weight_h3p_bytes(27, 64, 64), (long long)dm3d_packed_weight_skip_h3p_bytes(192, 64),
           (long long)dm3d_conv_scratch_bytes(&c), dm3d_last_error());
    free(buf);
    return bad == 13 ? 0 : 1;
}
