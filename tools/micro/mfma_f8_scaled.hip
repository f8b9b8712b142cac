// Microbenchmark / layout check of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands and unit block scales (gfx950):
//   1. operand layout: lane l supplies row (A) / column (B) l & 15 and the 32 consecutive k of k-group l >> 4 (byte j of the 32-byte
//      fragment = k 32 (l >> 4) + j); checked with exact small-integer data against a host product;
//   2. sustained rate against v_mfma_f32_16x16x32_f16, register operands, 16 accumulator tiles per wave, 2 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f8_scaled mfma_f8_scaled.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef int i8v __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// e4m3fn encoding of a small integer |v| <= 15 (exact)
__host__ __device__ inline uint8_t f8_of_int(int v) {
    if (v == 0) return 0;
    const uint8_t s = v < 0 ? 0x80 : 0;
    int a = v < 0 ? -v : v, e = 0;
    while ((a >> (e + 1)) != 0) ++e;                 // a in [2^e, 2^(e+1))
    const int mant = ((a << 3) >> e) & 7;            // 3 mantissa bits (exact for a <= 15)
    return s | (uint8_t)(((e + 7) << 3) | mant);
}

__global__ void check(const uint8_t* A, const uint8_t* B, float* C) {      // A [16][128], B [16 cols][128] (K contiguous), C [16][16]
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    i8v a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = *reinterpret_cast<const int*>(A + r * 128 + g * 32 + j * 4);
        b[j] = *reinterpret_cast<const int*>(B + r * 128 + g * 32 + j * 4);
    }
    f4v c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int i = 0; i < 4; ++i) C[(g * 4 + i) * 16 + r] = c[i];             // row = 4 (l >> 4) + i, column = l & 15
}

template <int F8>
__global__ __launch_bounds__(256, 2) void rate(const int* src, float* out, int iters) {
    const int lane = threadIdx.x;
    f4v acc[4][4] = {};
    i8v a8[4], b8[4];
    h8 a16[4], b16[4];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 8; ++j) { a8[i][j] = src[(lane * 8 + i * 64 + j) & 4095]; b8[i][j] = src[(lane * 8 + i * 64 + j + 2048) & 4095]; }
        for (int j = 0; j < 8; ++j) { a16[i][j] = (_Float16)((a8[i][j] & 255) * 0.001f); b16[i][j] = (_Float16)((b8[i][j] & 255) * 0.001f); }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (F8) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i], b8[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a16[i], b16[j], acc[i][j], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    uint8_t hA[16 * 128], hB[16 * 128];
    int iA[16 * 128], iB[16 * 128];
    srand(3);
    for (int i = 0; i < 16 * 128; ++i) { iA[i] = rand() % 17 - 8; iB[i] = rand() % 13 - 6; hA[i] = f8_of_int(iA[i]); hB[i] = f8_of_int(iB[i]); }
    uint8_t *dA, *dB; float* dC;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    float hC[256];
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        long ref = 0;
        for (int k = 0; k < 128; ++k) ref += (long)iA[m * 128 + k] * iB[n * 128 + k];
        if ((float)ref != hC[m * 16 + n]) { if (bad < 5) printf("mismatch C[%d][%d] = %g, expected %ld\n", m, n, hC[m * 16 + n], ref); ++bad; }
    }
    printf("layout check: %d of 256 outputs wrong\n", bad);

    int* src; float* out;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, 512 * 256 * 4);
    int h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (0x38 + (rand() & 7)) * 0x01010101;     // e4m3 values around 1
    hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep)
        for (int f8 = 0; f8 < 2; ++f8) {
            const int iters = f8 ? 20000 : 80000;                                    // same K in total
            hipEventRecord(e0);
            if (f8) hipLaunchKernelGGL(rate<1>, dim3(512), dim3(256), 0, 0, src, out, iters);
            else hipLaunchKernelGGL(rate<0>, dim3(512), dim3(256), 0, 0, src, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 512.0 * 4 * (double)iters * 16 * (f8 ? 65536.0 : 16384.0);
            printf("%s: %.2f ms  %.0f TFLOP/s\n", f8 ? "16x16x128 f8 scaled" : "16x16x32 f16", ms, flops / ms / 1e9);
        }
    return 0;
}
