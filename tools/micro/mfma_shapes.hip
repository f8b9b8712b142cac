// Microbenchmark: sustained f16 MFMA rate of the two gfx950 shapes on random data, operands re-read from LDS each step
// (ds_read_b128), 2 waves per SIMD, same 64x64 output tile per wave.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shapes mfma_shapes.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 2) void k(const _Float16* __restrict__ src, float* out, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[16384];     // 32 KB
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = src[(blockIdx.x * 16384 + i) & 0xfffff];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    // in-kernel clock (MI355X_MICROARCH.md 'DVFS give-back' item 6): delta(s_memtime) / delta(s_memrealtime) x 100 MHz around the loop
    unsigned long long t0 = 0, r0 = 0;
    if (clk && threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    if (SHAPE == 32) {
        f16v acc[2][2] = {};
        for (int it = 0; it < iters; ++it) {
            const int base = ((it * 7) & 31) * 512;
            h8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const h8*>(lds + ((base + (i * 64 + lane) * 8) & 16383));
                b[i] = *reinterpret_cast<const h8*>(lds + ((base + 4096 + (i * 64 + lane) * 8) & 16383));
            }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    } else {
        f4v acc[4][4] = {};
        for (int it = 0; it < iters; ++it) {      // one iteration = K 32 = two of the K-16 iterations above
            const int base = ((it * 7) & 31) * 512;
            h8 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const h8*>(lds + ((base + (i * 64 + lane) * 8) & 16383));
                b[i] = *reinterpret_cast<const h8*>(lds + ((base + 4096 + (i * 64 + lane) * 8) & 16383));
            }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    }
    if (clk && threadIdx.x == 0) { clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0; clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    _Float16* src; float* out;
    const int N = 1 << 20;
    _Float16* h = (_Float16*)malloc(N * 2);
    srand(1);
    for (int i = 0; i < N; ++i) h[i] = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    hipMalloc(&src, N * 2); hipMalloc(&out, 512 * 256 * 4);
    hipMemcpy(src, h, N * 2, hipMemcpyHostToDevice);
    unsigned long long* clk; hipMalloc(&clk, 512 * 2 * 8);
    unsigned long long hc[1024];
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 8; ++rep) {       // ~2.5 s of back-to-back launches; the last rounds give the sustained clock
        for (int shape = 0; shape < 2; ++shape) {
            const int iters32 = 40000;
            hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(k<32>, dim3(512), dim3(256), 0, 0, src, out, iters32, clk);
            else hipLaunchKernelGGL(k<16>, dim3(512), dim3(256), 0, 0, src, out, iters32 / 2, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 512.0 * 4 * iters32 * 12 * 32768.0;       // per wave per K-16 step: 12 MFMAs of 32x32x16
            hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost);
            double g[512]; for (int i = 0; i < 512; ++i) g[i] = hc[2 * i + 1] ? (double)hc[2 * i] / (double)hc[2 * i + 1] * 0.1 : 0.0;
            for (int i = 0; i < 512; ++i) for (int j = i + 1; j < 512; ++j) if (g[j] < g[i]) { double t = g[i]; g[i] = g[j]; g[j] = t; }
            const double mf = shape == 0 ? iters32 * 12.0 * 32 * 2 : iters32 / 2 * 48.0 * 16 * 2;      // MFMA cycles of one SIMD (two waves)
            printf("%s: %.2f ms  %.0f TFLOP/s executed | in-kernel clock median %.3f GHz (p10 %.3f, p90 %.3f) | MFMA duty %.3f\n", shape == 0 ? "32x32x16" : "16x16x32", ms,
                   flops / ms / 1e9, g[256], g[51], g[460], mf / (double)hc[2 * 256]);
        }
    }
    return 0;
}
