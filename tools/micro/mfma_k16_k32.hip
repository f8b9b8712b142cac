// Microbenchmark: cycles per MFMA of v_mfma_f32_16x16x32_f16 against v_mfma_f32_16x16x16_f16 on gfx950, one wave per SIMD, operands in
// registers, 16 independent accumulators (no dependent-issue stalls).  Question behind it ('left on the table'): the
// Winograd-x conv pads each term's nine taps to ten, so one K=32 step in five multiplies a real tap by a zero one — would a K=16 MFMA on
// those steps cost half a K=32 one?  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_k16_k32 mfma_k16_k32.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int K>
__global__ __launch_bounds__(256, 1) void k(const _Float16* __restrict__ src, float* out, int iters, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    h8 a8 = *reinterpret_cast<const h8*>(src + lane * 8), b8 = *reinterpret_cast<const h8*>(src + 512 + lane * 8);
    h4 a4 = {a8[0], a8[1], a8[2], a8[3]}, b4 = {b8[0], b8[1], b8[2], b8[3]};
    f4v acc[16] = {};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // inline asm: sixteen equal products would otherwise be folded into one chain with copies
            if (K == 32) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a8), "v"(b8));
            else asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a4), "v"(b4));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

// 32x32x16: K = 16 with twice the output tile — one (dz, dy) tap x 16 channels per step needs no pad tap at all
__global__ __launch_bounds__(256, 1) void k3216(const _Float16* __restrict__ src, float* out, int iters, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    h8 a8 = *reinterpret_cast<const h8*>(src + lane * 8), b8 = *reinterpret_cast<const h8*>(src + 512 + lane * 8);
    f16v acc[8] = {};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a8), "v"(b8));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

int main() {
    _Float16 h[1024];
    for (int i = 0; i < 1024; ++i) h[i] = (_Float16)(((i * 37) % 101) / 101.f - 0.5f);
    _Float16* src; float* out; unsigned long long* clk;
    hipMalloc(&src, sizeof(h)); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 256 * 8);
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 20000;
    unsigned long long hc[256];
    for (int rep = 0; rep < 3; ++rep)
        for (int kk = 0; kk < 3; ++kk) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (kk == 0) hipLaunchKernelGGL(k<32>, dim3(256), dim3(256), 0, 0, src, out, iters, clk);
            else if (kk == 1) hipLaunchKernelGGL(k<16>, dim3(256), dim3(256), 0, 0, src, out, iters, clk);
            else hipLaunchKernelGGL(k3216, dim3(256), dim3(256), 0, 0, src, out, iters, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(hc, clk, sizeof(hc), hipMemcpyDeviceToHost);
            double cyc = 0; for (int i = 0; i < 256; ++i) cyc += (double)hc[i]; cyc /= 256;
            const double n = (double)iters * (kk == 2 ? 8 : 16), fl = kk == 0 ? 2.0 * 16 * 16 * 32 : (kk == 1 ? 2.0 * 16 * 16 * 16 : 2.0 * 32 * 32 * 16);
            printf("%s f16: %.2f ms, %.2f cycles per MFMA per wave (s_memtime), %.1f TFLOP/s on 256 CUs x 4 waves\n",
                   kk == 0 ? "16x16x32" : (kk == 1 ? "16x16x16" : "32x32x16"), ms, cyc / n, 256.0 * 4 * n * fl / (ms * 1e-3) / 1e12);
        }
    return 0;
}
