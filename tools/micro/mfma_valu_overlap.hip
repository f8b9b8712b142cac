// Microbenchmark: can ONE wave overlap its own VALU / LDS instructions with its own MFMAs?  One wave per SIMD (256 threads per CU, launch
// bounds 1), a loop of 16 independent v_mfma_f32_16x16x32_f16 with K independent instructions behind each (K = 0 .. 4; VALU v_fma_f32 on
// private registers, or ds_read_b128), timed with s_memtime: cycles per MFMA.  If the shadow of an MFMA (16 cycles of matrix pipe, 4 of
// issue) can hold the wave's other instructions, K <= 3 costs nothing; if the wave issues strictly serially, every instruction adds 4.
// Round 4 (KIND 2): K ds_read_b128 per TWELVE MFMAs, spread evenly — the fragment-read ratios of the conv forms: 4 (the Winograd-x kernel and
// the direct kernel: 16 reads per 48 MFMAs), 8 and 10 (what F(2,3) x F(2,3) tilings would need: 2 x 2 or 4 x 1 tiles per term and wave).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int K, int KIND, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k(float* out, unsigned long long* clk, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += WAVES * 64) lds[i] = (float)i;
    __syncthreads();
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    f4v acc[16] = {};
    float x[4] = {1.0f, 2.0f, 3.0f, 4.0f};
    f4v r[4] = {};
    const unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds + (threadIdx.x & 63) * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
            if (KIND == 2) {                    // K reads per 12 MFMAs (16 MFMAs per iteration: the pattern runs over 48 = three iterations' worth of phase)
                const int ph = m % 12;
                if ((ph + 1) * K / 12 != ph * K / 12) asm volatile("ds_read_b128 %0, %1" : "=v"(r[m & 3]) : "v"(la));
                continue;
            }
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[j]));
                else asm volatile("ds_read_b128 %0, %1" : "=v"(r[j]) : "v"(la));
            }
        }
        if (KIND >= 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int m = 0; m < 16; ++m) s += acc[m][0] + acc[m][3];
    for (int j = 0; j < 4; ++j) s += x[j] + r[j][0];
    out[blockIdx.x * WAVES * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int K, int KIND, int WAVES>
static void run(float* out, unsigned long long* clk) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<K, KIND, WAVES>), dim3(256), dim3(WAVES * 64), 0, 0, out, clk, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[256];
    (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long s = 0;
    for (int i = 0; i < 256; ++i) s += h[i];
    if (KIND == 2) {
        int reads = 0;
        for (int m = 0; m < 16; ++m) { const int ph = m % 12; reads += (ph + 1) * K / 12 != ph * K / 12; }
        printf("waves per SIMD %d, ds_read_b128 x %d per 12 MFMAs (%d reads per 16-MFMA iteration = %.2f per MFMA): %.2f cycles per MFMA per wave\n", WAVES / 4, K,
               reads, reads / 16.0, (double)s / 256.0 / (iters * 16.0));
        return;
    }
    printf("waves per SIMD %d, %s x %d per MFMA: %.0f shader cycles (s_memtime) ... %.2f cycles per MFMA per wave\n", WAVES / 4, KIND ? "ds_read_b128" : "v_fma_f32", K,
           (double)s / 256.0, (double)s / 256.0 / (iters * 16.0));
}

int main() {
    float* out; unsigned long long* clk;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&clk, 256 * 8);
    run<0, 0, 4>(out, clk); run<1, 0, 4>(out, clk); run<2, 0, 4>(out, clk); run<3, 0, 4>(out, clk); run<4, 0, 4>(out, clk);
    run<1, 1, 4>(out, clk); run<2, 1, 4>(out, clk);
    run<0, 0, 8>(out, clk); run<2, 0, 8>(out, clk); run<4, 0, 8>(out, clk); run<2, 1, 8>(out, clk);
    run<4, 2, 4>(out, clk); run<6, 2, 4>(out, clk); run<8, 2, 4>(out, clk); run<10, 2, 4>(out, clk); run<12, 2, 4>(out, clk);
    return 0;
}
