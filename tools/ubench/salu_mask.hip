// Developer micro-benchmark (gfx950), VERDICT r1 #6: can the per-cell mismatch compare leave the vector unit?
// The production cell makes its mismatch mask with one v_cmp per cell (into an SGPR pair, one cell ahead).  The read
// base of a lane changes every step, the haplotype base of a (lane, column) never does: with 2-bit base codes the mask
// of column k is (X0 ^ Y0k) | (X1 ^ Y1k), where X0/X1 are the lane masks of the two code bits of the READ base (two
// v_cmp per STEP) and Y0k/Y1k the constant lane masks of column k's HAPLOTYPE base: three scalar instructions per
// cell instead of one vector instruction.  This file times whole cell bodies, counted in CELLS (not instructions):
//   f64_vcmp   fma fma | exec<-mask | mul | exec<--1 | v_cmp -> next mask | mul fma fma        7 VALU + 2 SALU  (production)
//   f64_salu   fma fma | exec<-mask | mul | exec<--1 | s_xor s_xor s_or   | mul fma fma        6 VALU + 5 SALU
//   f32_vcmp   fma fma | v_cmp | v_cndmask | mul mul fma fma                                   8 VALU           (production)
//   f32_salu   fma fma | s_xor s_xor s_or | v_cndmask (SGPR mask) | mul mul fma fma            7 VALU + 3 SALU
// The SGPR cost is what decides whether it can be used: 4 SGPRs per column (K = 19: 76 of the 102 a wave has).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/salu_mask.hip -o /tmp/salu_mask && /tmp/salu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define FIN(T)                                                                                                       \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == (T)12345.678) out[0] = (double)a0;                                    \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                                                       \
        out[1] = (double)(clock64() - c0);                                                                           \
        out[2] = (double)(wall_clock64() - w0);                                                                      \
    }
#define X8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)

// ---- f64 ---------------------------------------------------------------------------------------------------------
#define D_VCMP(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n v_fma_f64 %" #i ", %" #i ", %8, %9\n s_mov_b64 exec, s[20:21]\n v_mul_f64 %" #i ", %" #i ", %8\n s_mov_b64 exec, -1\n v_cmp_ne_u32_e64 s[20:21], %10, %11\n v_mul_f64 %" #i ", %" #i ", %8\n v_fma_f64 %" #i ", %" #i ", %8, %9\n v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define D_SALU(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n v_fma_f64 %" #i ", %" #i ", %8, %9\n s_mov_b64 exec, s[20:21]\n v_mul_f64 %" #i ", %" #i ", %8\n s_mov_b64 exec, -1\n s_xor_b64 s[22:23], s[24:25], s[28:29]\n s_xor_b64 s[20:21], s[26:27], s[30:31]\n s_or_b64 s[20:21], s[20:21], s[22:23]\n v_mul_f64 %" #i ", %" #i ", %8\n v_fma_f64 %" #i ", %" #i ", %8, %9\n v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define KD(name, BODY)                                                                                               \
    __global__ __launch_bounds__(256) void name(double *out, int iters, double B, double C, uint32_t ux) {            \
        double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        double b = B, c = C;                                                                                         \
        uint32_t vy = threadIdx.x & 3;                                                                               \
        const long long c0 = clock64(), w0 = wall_clock64();                                                         \
        asm volatile("v_cmp_ne_u32_e64 s[20:21], %0, %1\n s_mov_b64 s[24:25], s[20:21]\n s_not_b64 s[26:27], s[20:21]\n" \
                     "s_mov_b64 s[28:29], 0x5555\n s_mov_b64 s[30:31], 0x3333" : : "s"(ux), "v"(vy)                    \
                     : "s20", "s21", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "scc");                  \
        for (int i = 0; i < iters; ++i)                                                                              \
            asm volatile(X8(BODY) X8(BODY)                                                                           \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)               \
                         : "v"(b), "v"(c), "s"(ux), "v"(vy)                                                           \
                         : "vcc", "scc", "s20", "s21", "s22", "s23");                                                 \
        FIN(double)                                                                                                  \
    }
KD(k_f64_vcmp, D_VCMP)
KD(k_f64_salu, D_SALU)

// ---- f32 ---------------------------------------------------------------------------------------------------------
#define S_VCMP(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n v_cmp_ne_u32_e64 s[20:21], %10, %11\n v_cndmask_b32_e64 v60, 1.0, %8, s[20:21]\n v_mul_f32 %" #i ", %" #i ", v60\n v_mul_f32 %" #i ", %" #i ", %8\n v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define S_SALU(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n s_xor_b64 s[22:23], s[24:25], s[28:29]\n s_xor_b64 s[20:21], s[26:27], s[30:31]\n s_or_b64 s[20:21], s[20:21], s[22:23]\n v_cndmask_b32_e64 v60, 1.0, %8, s[20:21]\n v_mul_f32 %" #i ", %" #i ", v60\n v_mul_f32 %" #i ", %" #i ", %8\n v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define KS(name, BODY)                                                                                               \
    __global__ __launch_bounds__(256) void name(double *out, int iters, double B, double C, uint32_t ux) {            \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b = (float)B, c = (float)C;                                                                            \
        uint32_t vy = threadIdx.x & 3;                                                                               \
        const long long c0 = clock64(), w0 = wall_clock64();                                                         \
        asm volatile("v_cmp_ne_u32_e64 s[20:21], %0, %1\n s_mov_b64 s[24:25], s[20:21]\n s_not_b64 s[26:27], s[20:21]\n" \
                     "s_mov_b64 s[28:29], 0x5555\n s_mov_b64 s[30:31], 0x3333" : : "s"(ux), "v"(vy)                    \
                     : "s20", "s21", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "scc");                  \
        for (int i = 0; i < iters; ++i)                                                                              \
            asm volatile(X8(BODY) X8(BODY)                                                                           \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)               \
                         : "v"(b), "v"(c), "s"(ux), "v"(vy)                                                           \
                         : "vcc", "scc", "s20", "s21", "s22", "s23", "v60");                                          \
        FIN(float)                                                                                                   \
    }
KS(k_f32_vcmp, S_VCMP)
KS(k_f32_salu, S_SALU)

typedef void (*kern_t)(double *, int, double, double, uint32_t);
static void run(const char *name, kern_t k, int valu, int salu) {
    double *out;
    hipMalloc(&out, 64);
    for (int wps : {1, 2, 3, 4}) {
        const int blocks = 256 * wps, iters = 100000 / wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9, 3u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9, 3u);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double cw[3];
        hipMemcpy(cw, out, 24, hipMemcpyDeviceToHost);
        const double wave_cells = (double)iters * 16 * wps;  // cell bodies per SIMD (each is 64 lane-cells)
        const double ghz = cw[1] / cw[2] * 0.1;
        printf("%-10s %d VALU + %d SALU  waves/SIMD=%d  %.2f ms  %.2f clk per cell body at %.2f GHz  -> %.1f G lane-cells/s per SIMD, %.2f T/s per chip\n",
               name, valu, salu, wps, ms, ms * 1e6 * ghz / wave_cells, ghz, wave_cells * 64 / ms / 1e6, wave_cells * 64 * 1024 / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    run("f64_vcmp", k_f64_vcmp, 7, 2);
    run("f64_salu", k_f64_salu, 6, 5);
    run("f32_vcmp", k_f32_vcmp, 8, 0);
    run("f32_salu", k_f32_salu, 7, 3);
    return 0;
}
