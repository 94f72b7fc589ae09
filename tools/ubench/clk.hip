// Developer micro-benchmark (gfx950): sustained FP64 VALU issue rate with the whole chip busy -> the effective
// shader clock under an FP64 load (full rate = one wave64 v_fma_f64 per 4 clk per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/clk.hip -o /tmp/clk && /tmp/clk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_ind_fma(double *out, int iters, double b, double c) {
    const long long c0 = clock64(), w0 = wall_clock64();
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
            "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
            "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
            "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(b), "v"(c));
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[0] = a0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {  // shader-clock ticks vs 100 MHz wall ticks over the kernel
        out[1] = (double)(clock64() - c0);
        out[2] = (double)(wall_clock64() - w0);
    }
}
__global__ __launch_bounds__(256) void k_ind_fma32(float *out, int iters, float b, float c) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
            "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
            "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
            "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(b), "v"(c));
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = a0;
}

int main() {
    double *out; hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass)
    for (int wps : {1, 2, 4}) {  // waves per SIMD
        const int blocks = 256 * wps;  // 4 waves per block = one per SIMD
        const int iters = 1500000 / wps;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (pass == 0) hipLaunchKernelGGL(k_ind_fma, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9);
            else hipLaunchKernelGGL(k_ind_fma32, dim3(blocks), dim3(256), 0, 0, (float *)out, iters, 1.0000001f, 1e-9f);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = (double)iters * 16 * wps;
            double cw[3]; hipMemcpy(cw, out, 24, hipMemcpyDeviceToHost);
            if (pass == 0) printf("   clock64/wall_clock64 = %.3f (x100 MHz if wall is the 100 MHz counter)\n", cw[1] / cw[2]);
            printf("%s waves/SIMD=%d  %.1f ms  %.3f G wave-instr/s per SIMD -> %.2f GHz if 4 clk/instr; %.1f TFLOP/s\n",
                   pass ? "f32" : "f64", wps, ms, instr_per_simd / ms / 1e6, 4 * instr_per_simd / ms / 1e6,
                   instr_per_simd * 1024 * 128 / ms / 1e9);
        }
    }
    return 0;
}
