// What do the gfx950 float -> fp8 conversions do?  (saturation, the scale operand's direction)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, float scale) {
    const float a = in[threadIdx.x * 2], b = in[threadIdx.x * 2 + 1];
    out[threadIdx.x * 4 + 0] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    s2 old = {0, 0};
    s2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, b, scale, false);
    out[threadIdx.x * 4 + 1] = (unsigned)(unsigned short)r[0];
    h2 hv = {(_Float16)a, (_Float16)b};
    s2 r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, hv, scale, false);
    out[threadIdx.x * 4 + 2] = (unsigned)(unsigned short)r2[0];
    out[threadIdx.x * 4 + 3] = 0;
}
static float f8dec(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    if (e == 15 && m == 7) return s ? -__builtin_nanf("") : __builtin_nanf("");
    const float mag = e == 0 ? m / 8.0f * 0.015625f : (1.0f + m / 8.0f) * __builtin_ldexpf(1.0f, e - 7);
    return s ? -mag : mag;
}
int main() {
    float h[16] = {1.0f, 3.0f, 500.0f, 1.0e5f, 1.0e-3f, -449.0f, 0.3f, 20.0f, 447.0f, 460.0f, 480.0f, 2.0e-3f, 6.0e4f, -7.0f, 0.06f, 100.0f};
    float* d; unsigned* o; unsigned ho[32];
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof ho);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d, o, 4.0f);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 2; ++j)
            printf("x = %-10g  cvt_pk_fp8_f32 -> %-8g  scalef32(4.0) f32 -> %-8g  scalef32(4.0) f16 -> %-8g\n", h[2 * i + j], f8dec((ho[i * 4] >> (8 * j)) & 255),
                   f8dec((ho[i * 4 + 1] >> (8 * j)) & 255), f8dec((ho[i * 4 + 2] >> (8 * j)) & 255));
    // small inputs scaled up (the al * 2^11 conversion): float16 subnormal sources (< 6.1e-5) through the f16-source form
    float g[16] = {2.44e-4f, 1.22e-4f, 6.2e-5f, 6.0e-5f, 3.0e-5f, 1.5e-5f, 4.0e-6f, 1.0e-6f, -2.0e-4f, -5.0e-5f, -8.0e-6f, 3.3e-5f, 9.0e-5f, 2.0e-5f, 1.0e-4f, 5.0e-7f};
    hipMemcpy(d, g, sizeof g, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(8), 0, 0, d, o, 0.00048828125f);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 2; ++j)
            printf("x = %-10g (x * 2^11 = %-8g) scalef32(2^-11) f32 -> %-8g  scalef32(2^-11) f16 -> %-8g\n", g[2 * i + j], g[2 * i + j] * 2048.0f,
                   f8dec((ho[i * 4 + 1] >> (8 * j)) & 255), f8dec((ho[i * 4 + 2] >> (8 * j)) & 255));
    return 0;
}
