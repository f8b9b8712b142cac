// Microbenchmark: how much of the MFMA array's power (hence, under the chip's power limit, its rate) depends on the operand BITS?
// The f16 16x16x32 loop of mfma_dtypes.hip on random float16 operands whose low MASKA / MASKB mantissa bits are cleared in A / B.
// Round 3 question: would `lo` terms of the float16 split with fewer significant bits make the two cross-term passes cheaper?
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_bits mfma_bits.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void k(const unsigned int* __restrict__ src, float* out, int iters, unsigned long long* clk, unsigned maskA, unsigned maskB) {
    __shared__ __attribute__((aligned(16))) unsigned int lds[8192];     // 32 KB: first half A operands, second half B operands
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = src[(blockIdx.x * 8192 + i) & 0xfffff] & (((i & 4095) >= 2048) ? maskB : maskA);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    f4v acc[4][4] = {};
    for (int it = 0; it < iters; ++it) {
        const int base = ((it * 7) & 1) * 4096;
        u4v a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = *reinterpret_cast<const u4v*>(lds + base + ((((it * 5) & 7) * 256 + (i * 64 + lane) * 4) & 2047));
            b[i] = *reinterpret_cast<const u4v*>(lds + base + 2048 + ((((it * 3) & 7) * 256 + (i * 64 + lane) * 4) & 2047));
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a[i]), __builtin_bit_cast(h8, b[j]), acc[i][j], 0, 0, 0);
    }
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0; clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int N = 1 << 20;
    unsigned int* h = (unsigned int*)malloc(N * 4);
    unsigned int* src; float* out; unsigned long long* clk;
    (void)hipMalloc(&src, N * 4); (void)hipMalloc(&out, 512 * 256 * 4); (void)hipMalloc(&clk, 1024 * 8);
    srand(1);
    for (int i = 0; i < N; ++i) {           // two random float16 per word, exponents around 1, both signs
        unsigned int w = 0;
        for (int b = 0; b < 2; ++b) {
            const unsigned int r = (unsigned int)rand();
            w |= (((r & 0x8000u) | ((0x0Cu + ((r >> 10) & 3u)) << 10) | (r & 0x3ffu)) & 0xffffu) << (16 * b);
        }
        h[i] = w;
    }
    (void)hipMemcpy(src, h, N * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    const int drops[][2] = {{0, 0}, {3, 0}, {3, 3}, {5, 0}, {5, 5}, {8, 0}, {8, 8}, {10, 10}, {0, 0}};
    for (auto& d : drops) {
        const unsigned ma = 0xffffu & ~((1u << d[0]) - 1), mb = 0xffffu & ~((1u << d[1]) - 1);
        float ms = 0; unsigned long long hc[1024];
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, src, out, iters, clk, ma | (ma << 16), mb | (mb << 16));
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
        }
        (void)hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost);
        double g[512];
        for (int i = 0; i < 512; ++i) g[i] = hc[2 * i + 1] ? (double)hc[2 * i] / (double)hc[2 * i + 1] * 0.1 : 0.0;
        for (int i = 0; i < 512; ++i) for (int j = i + 1; j < 512; ++j) if (g[j] < g[i]) { double t = g[i]; g[i] = g[j]; g[j] = t; }
        const double ops = 512.0 * 4 * iters * 48.0 * 2.0 * 16 * 16 * 32;
        printf("A: low %2d mantissa bits cleared, B: %2d | %7.2f ms  %6.0f TFLOP/s | clock median %.3f GHz\n", d[0], d[1], ms, ops / ms / 1e9, g[256]);
    }
    return 0;
}
