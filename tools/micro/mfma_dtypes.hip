// Microbenchmark: which MFMA datatype the chip can run fastest under its power limit.  Same loop for every type: operands re-read from
// LDS each step (ds_read_b128), 48 MFMAs of the 16x16 shape per step on 16 accumulators, 2 waves per SIMD, RANDOM data; per type the
// wall rate, the in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz, MI355X_MICROARCH.md 'DVFS give-back' item 6) and
// the matrix-pipe duty.  Round 3 question: the split-float16 conv is energy-bound (same instruction stream on zeros: 2.39 GHz, on
// random data: 1.83-1.9 GHz) — is any other operand type cheaper per MFMA?
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_dtypes mfma_dtypes.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));
typedef int i8v __attribute__((ext_vector_type(8)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

enum { F16 = 0, BF16 = 1, I8 = 2, FP8 = 3, F8SC = 4, F4SC = 5 };

template <int T>
__global__ __launch_bounds__(256, 2) void k(const unsigned int* __restrict__ src, float* out, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) unsigned int lds[8192];     // 32 KB
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = src[(blockIdx.x * 8192 + i) & 0xfffff];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = 0, r0 = 0;
    if (threadIdx.x == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    f4v acc[4][4] = {};
    i4v iacc[4][4] = {};
    for (int it = 0; it < iters; ++it) {
        const int base = ((it * 7) & 31) * 256;
        u4v a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = *reinterpret_cast<const u4v*>(lds + ((base + (i * 64 + lane) * 4) & 8191));
            b[i] = *reinterpret_cast<const u4v*>(lds + ((base + 2048 + (i * 64 + lane) * 4) & 8191));
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (T == F16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a[i]), __builtin_bit_cast(h8, b[j]), acc[i][j], 0, 0, 0);
                    if (T == BF16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8, a[i]), __builtin_bit_cast(b8, b[j]), acc[i][j], 0, 0, 0);
                    if (T == I8) iacc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i4v, a[i]), __builtin_bit_cast(i4v, b[j]), iacc[i][j], 0, 0, 0);
                    if (T == FP8) {
                        const long al = ((long)a[i][1] << 32) | a[i][0], bl = ((long)b[j][1] << 32) | b[j][0];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(al, bl, acc[i][j], 0, 0, 0);
                    }
                    if (T == F8SC || T == F4SC) {          // K = 128: 32 operand bytes per lane (the 16 read + a rotation of them: same LDS traffic per MFMA as above)
                        const i8v a8 = {(int)a[i][0], (int)a[i][1], (int)a[i][2], (int)a[i][3], (int)a[i][1], (int)a[i][2], (int)a[i][3], (int)a[i][0]};
                        const i8v b8 = {(int)b[j][0], (int)b[j][1], (int)b[j][2], (int)b[j][3], (int)b[j][2], (int)b[j][3], (int)b[j][0], (int)b[j][1]};
                        // cbsz / blgp: 0 = fp8 e4m3, 4 = fp4 e2m1; unit block scales
                        if (T == F8SC) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                        else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[i][j], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                    }
                }
    }
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = __builtin_amdgcn_s_memtime() - t0; clk[blockIdx.x * 2 + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r] + (float)iacc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int T>
static void run(const char* name, double k_per_mfma, double cyc_per_mfma, const unsigned int* src, float* out, unsigned long long* clk, int zeros) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    float ms = 0; unsigned long long hc[1024];
    for (int rep = 0; rep < 6; ++rep) {          // ~2 s of back-to-back launches; the last one is reported
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<T>, dim3(512), dim3(256), 0, 0, src, out, iters, clk);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    (void)hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost);
    double g[512];
    for (int i = 0; i < 512; ++i) g[i] = hc[2 * i + 1] ? (double)hc[2 * i] / (double)hc[2 * i + 1] * 0.1 : 0.0;
    for (int i = 0; i < 512; ++i) for (int j = i + 1; j < 512; ++j) if (g[j] < g[i]) { double t = g[i]; g[i] = g[j]; g[j] = t; }
    const double mfmas = 512.0 * 4 * iters * 48.0;                       // wave-MFMAs in the launch
    const double ops = mfmas * 2.0 * 16 * 16 * k_per_mfma;
    const double duty = iters * 48.0 * cyc_per_mfma * 2 / (double)hc[2 * 256];
    printf("%-26s %s: %7.2f ms  %6.0f Tops/s | clock median %.3f GHz (p10 %.3f, p90 %.3f) | pipe duty %.3f | %.2f us per 1000 wave-MFMAs per SIMD\n", name,
           zeros ? "zeros " : "random", ms, ops / ms / 1e9, g[256], g[51], g[460], duty, ms * 1e3 / (iters * 48.0 * 2 / 1000.0));
}

int main() {
    const int N = 1 << 20;
    unsigned int* h = (unsigned int*)malloc(N * 4);
    unsigned int* src; float* out; unsigned long long* clk;
    (void)hipMalloc(&src, N * 4); (void)hipMalloc(&out, 512 * 256 * 4); (void)hipMalloc(&clk, 1024 * 8);
    for (int zeros = 0; zeros < 2; ++zeros) {
        srand(1);
        // random bits with sane exponents for the float types: every byte in [0x30, 0x47] / [0xB0, 0xC7] keeps f16 / bf16 / fp8 values finite and O(1)
        for (int i = 0; i < N; ++i) {
            unsigned int w = 0;
            for (int b = 0; b < 4; ++b) {
                unsigned int byte = (unsigned int)(rand() & 0xff);
                if (b & 1) byte = (byte & 0x80) | (0x30 + (byte & 0x7f) % 0x18);      // the high byte of each half-word: sign + exponent
                w |= byte << (8 * b);
            }
            h[i] = zeros ? 0u : w;
        }
        (void)hipMemcpy(src, h, N * 4, hipMemcpyHostToDevice);
        run<F16>("f16 16x16x32", 32, 16, src, out, clk, zeros);
        run<BF16>("bf16 16x16x32", 32, 16, src, out, clk, zeros);
        run<I8>("i8 16x16x64", 64, 16, src, out, clk, zeros);
        run<FP8>("fp8 16x16x32", 32, 16, src, out, clk, zeros);
        run<F8SC>("f8f6f4 16x16x128 fp8", 128, 32, src, out, clk, zeros);
        run<F4SC>("f8f6f4 16x16x128 fp4", 128, 16, src, out, clk, zeros);
    }
    return 0;
}
