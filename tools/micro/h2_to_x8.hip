// h2_to_x8 (float8 operands from a stored DM3D_FMT_H2 pair) against split8_f8 (from float32) on the same values
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../../3d-condtional-stable-diffusion_amd/csrc/dm3d_h3.h"
__global__ void k(const float* in, unsigned* out) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(in + threadIdx.x * 8), v1 = *reinterpret_cast<const f32x4*>(in + threadIdx.x * 8 + 4);
    h8 hi, x8;
    split8_f8(v0, v1, DM3D_F8_LIMIT, DM3D_F8_LIMIT, hi, x8);
    unsigned w[8];
    for (int e = 0; e < 4; ++e) { w[e] = split1_bits(v0[e]); w[4 + e] = split1_bits(v1[e]); }
    u32x4 ph, pl;
    for (int e = 0; e < 4; ++e) {
        ph[e] = (w[2 * e] & 0xffffu) | (w[2 * e + 1] << 16);
        pl[e] = (w[2 * e] >> 16) | (w[2 * e + 1] & 0xffff0000u);
    }
    const h8 y8 = h2_to_x8(__builtin_bit_cast(h8, ph), __builtin_bit_cast(h8, pl));
    const u32x4 a = __builtin_bit_cast(u32x4, x8), b = __builtin_bit_cast(u32x4, y8), c = __builtin_bit_cast(u32x4, hi);
    for (int e = 0; e < 4; ++e) { out[threadIdx.x * 16 + e] = a[e]; out[threadIdx.x * 16 + 4 + e] = b[e]; out[threadIdx.x * 16 + 8 + e] = c[e]; out[threadIdx.x * 16 + 12 + e] = ph[e]; }
}
int main() {
    float h[16] = {1.0f, -0.27f, 0.5003f, 2.71828f, 0.001234f, -1.5e-4f, 0.9999f, 3.14159f, 100.3f, -7.77f, 0.06f, 20.1f, 1e-6f, 0.33333f, -0.12345f, 5.5555f};
    float* d; unsigned* o; unsigned ho[32];
    (void)hipMalloc(&d, sizeof h); (void)hipMalloc(&o, sizeof ho);
    (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(2), 0, 0, d, o);
    (void)hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    for (int t = 0; t < 2; ++t) {
        printf("split8_f8: %08x %08x %08x %08x   h2_to_x8: %08x %08x %08x %08x   %s\n", ho[t * 16], ho[t * 16 + 1], ho[t * 16 + 2], ho[t * 16 + 3],
               ho[t * 16 + 4], ho[t * 16 + 5], ho[t * 16 + 6], ho[t * 16 + 7], memcmp(ho + t * 16, ho + t * 16 + 4, 16) ? "DIFFER" : "same");
        printf("   hi: %08x %08x %08x %08x   hi via split1_bits: %08x %08x %08x %08x\n", ho[t * 16 + 8], ho[t * 16 + 9], ho[t * 16 + 10], ho[t * 16 + 11],
               ho[t * 16 + 12], ho[t * 16 + 13], ho[t * 16 + 14], ho[t * 16 + 15]);
    }
    return 0;
}
