// Developer micro-benchmark (gfx950): VALU issue interval (shader clocks per wave64 instruction) of the instruction
// forms the PairHMM row update uses, with 1 / 2 / 4 waves per SIMD on a full chip.  Clocks are real shader clocks
// (clock64() = s_memtime), so DVFS does not distort the numbers; the clock itself is reported too.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/issue.hip -o /tmp/issue && /tmp/issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define DECL double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
             double b = B, c = C; uint32_t vy = threadIdx.x & 3; const long long c0 = clock64(), w0 = wall_clock64();
#define FIN  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[0] = a0; \
             if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (double)(clock64() - c0); out[2] = (double)(wall_clock64() - w0); }
#define OPS8(fmt) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)
#define REGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(B), "s"(ux), "v"(vy) : "vcc"

#define K(name, F)                                                                                   \
    __global__ __launch_bounds__(256) void name(double *out, int iters, double B, double C, uint32_t ux) { \
        DECL                                                                                         \
        for (int i = 0; i < iters; ++i) asm volatile(OPS8(F) OPS8(F) REGS);                          \
        FIN                                                                                          \
    }
#define F_FMA3(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define F_FMA_S(i) "v_fma_f64 %" #i ", %" #i ", %10, %9\n"
#define F_FMA_AA(i) "v_fma_f64 %" #i ", %" #i ", %" #i ", %9\n"
#define F_MUL2(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define F_MUL_S(i) "v_mul_f64 %" #i ", %" #i ", %10\n"
#define F_ADD2(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define F_MOV(i) "v_mov_b64 %" #i ", %8\n"
#define F_CMPX(i) "v_cmpx_ne_u32_e32 vcc, %11, %12\n v_mul_f64 %" #i ", %" #i ", %8\n s_mov_b64 exec, -1\n"
#define F_CMP(i) "v_cmp_ne_u32_e32 vcc, %11, %12\n v_mul_f64 %" #i ", %" #i ", %8\n"
// the 7-op cell body shape: fma fma cmpx mul(masked) mul fma fma
#define F_CELL(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n v_fma_f64 %" #i ", %" #i ", %8, %9\n v_cmpx_ne_u32_e32 vcc, %11, %12\n v_mul_f64 %" #i ", %" #i ", %8\n s_mov_b64 exec, -1\n v_mul_f64 %" #i ", %" #i ", %8\n v_fma_f64 %" #i ", %" #i ", %8, %9\n v_fma_f64 %" #i ", %" #i ", %8, %9\n"
// select variants: SGPR-mask form (v_cmp to an SGPR pair, EXEC loaded by the SALU), singly and in groups of four
#define F_SMASK(i) "v_cmp_ne_u32_e64 s[20:21], %11, %12\n s_mov_b64 exec, s[20:21]\n v_mul_f64 %" #i ", %" #i ", %8\n s_mov_b64 exec, -1\n"
#define F_CMPX_IL(i) "v_cmpx_ne_u32_e32 vcc, %11, %12\n v_mul_f64 %" #i ", %" #i ", %8\n s_mov_b64 exec, -1\n v_fma_f64 %" #i ", %" #i ", %8, %9\n"
__global__ __launch_bounds__(256) void k_smask4(double *out, int iters, double B, double C, uint32_t ux) {
    DECL
    for (int i = 0; i < iters; ++i)
        asm volatile(
            "v_cmp_ne_u32_e64 s[20:21], %11, %12\n v_cmp_ne_u32_e64 s[22:23], %11, %12\n v_cmp_ne_u32_e64 s[24:25], %11, %12\n v_cmp_ne_u32_e64 s[26:27], %11, %12\n"
            "s_mov_b64 exec, s[20:21]\n v_mul_f64 %0, %0, %8\n s_mov_b64 exec, s[22:23]\n v_mul_f64 %1, %1, %8\n"
            "s_mov_b64 exec, s[24:25]\n v_mul_f64 %2, %2, %8\n s_mov_b64 exec, s[26:27]\n v_mul_f64 %3, %3, %8\n s_mov_b64 exec, -1\n"
            "v_cmp_ne_u32_e64 s[20:21], %11, %12\n v_cmp_ne_u32_e64 s[22:23], %11, %12\n v_cmp_ne_u32_e64 s[24:25], %11, %12\n v_cmp_ne_u32_e64 s[26:27], %11, %12\n"
            "s_mov_b64 exec, s[20:21]\n v_mul_f64 %4, %4, %8\n s_mov_b64 exec, s[22:23]\n v_mul_f64 %5, %5, %8\n"
            "s_mov_b64 exec, s[24:25]\n v_mul_f64 %6, %6, %8\n s_mov_b64 exec, s[26:27]\n v_mul_f64 %7, %7, %8\n s_mov_b64 exec, -1\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(b), "v"(c), "s"(B), "s"(ux), "v"(vy)
            : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    FIN
}
#undef REGS
#define REGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "s"(B), "s"(ux), "v"(vy) : "vcc", "s20", "s21"
K(k_smask, F_SMASK)
K(k_cmpx_il, F_CMPX_IL)
K(k_fma3, F_FMA3)
K(k_fma_s, F_FMA_S)
K(k_fma_aa, F_FMA_AA)
K(k_mul2, F_MUL2)
K(k_mul_s, F_MUL_S)
K(k_add2, F_ADD2)
K(k_mov, F_MOV)
K(k_cmpx, F_CMPX)
K(k_cmp, F_CMP)
K(k_cell, F_CELL)

// the same cell body in f32 (what an f32-first mode like the reference's gkl arm would issue) and with packed f32
// for the four FMAs / the I multiply (two columns per instruction; select and D chain stay per column)
__global__ __launch_bounds__(256) void k_cell_f32(double *out, int iters, double B, double C, uint32_t ux) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = (float)B, c = (float)C; uint32_t vy = threadIdx.x & 3;
    const long long c0 = clock64(), w0 = wall_clock64();
#define F_CELL32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n v_cmpx_ne_u32_e32 vcc, %11, %12\n v_mul_f32 %" #i ", %" #i ", %8\n s_mov_b64 exec, -1\n v_mul_f32 %" #i ", %" #i ", %8\n v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n"
    for (int i = 0; i < iters; ++i)
        asm volatile(OPS8(F_CELL32) OPS8(F_CELL32)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(B), "s"(ux), "v"(vy) : "vcc");
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = a0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (double)(clock64() - c0); out[2] = (double)(wall_clock64() - w0); }
}

// f32 cell with a v_cndmask select (32-bit values need only one): cmp, cndmask, mul instead of cmpx, masked mul, s_mov
__global__ __launch_bounds__(256) void k_cell_f32_cnd(double *out, int iters, double B, double C, uint32_t ux) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = (float)B, c = (float)C; uint32_t vy = threadIdx.x & 3;
    const long long c0 = clock64(), w0 = wall_clock64();
#define F_CELL32C(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n v_cmp_ne_u32_e32 vcc, %11, %12\n v_cndmask_b32_e32 v60, 1.0, %8, vcc\n v_mul_f32 %" #i ", %" #i ", v60\n v_mul_f32 %" #i ", %" #i ", %8\n v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n"
    for (int i = 0; i < iters; ++i)
        asm volatile(OPS8(F_CELL32C) OPS8(F_CELL32C)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(B), "s"(ux), "v"(vy) : "vcc", "v60");
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = a0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (double)(clock64() - c0); out[2] = (double)(wall_clock64() - w0); }
}

// f64 cell with a v_cndmask select: fma fma cmp cnd cnd mul mul fma fma (9 VALU, three of them 32-bit)
__global__ __launch_bounds__(256) void k_cell_f64_cnd(double *out, int iters, double B, double C, uint32_t ux) {
    DECL
    uint32_t blo = (uint32_t)__double2loint(b), bhi = (uint32_t)__double2hiint(b);
    for (int i = 0; i < iters; ++i) {
#define F_C64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n v_fma_f64 %" #i ", %" #i ", %8, %9\n v_cmp_ne_u32_e32 vcc, %11, %12\n v_cndmask_b32_e32 v60, 0, %13, vcc\n v_cndmask_b32_e32 v61, %15, %14, vcc\n v_mul_f64 %" #i ", %" #i ", v[60:61]\n v_mul_f64 %" #i ", %" #i ", %8\n v_fma_f64 %" #i ", %" #i ", %8, %9\n v_fma_f64 %" #i ", %" #i ", %8, %9\n"
        asm volatile(OPS8(F_C64) OPS8(F_C64)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(B), "s"(ux), "v"(vy), "v"(blo), "v"(bhi), "v"(0x3ff00000u) : "vcc", "v60", "v61");
    }
    FIN
}

// f64 cell with the compare hoisted one cell ahead into an SGPR pair (no VALU write of EXEC): per cell
//   fma, fma, s_mov exec <- mask, masked mul, s_mov exec <- -1, v_cmp (next cell's mask), mul, fma
__global__ __launch_bounds__(256) void k_cell_f64_hoist(double *out, int iters, double B, double C, uint32_t ux) {
    DECL
#define F_HA(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n v_fma_f64 %" #i ", %" #i ", %8, %9\n s_mov_b64 exec, s[20:21]\n v_mul_f64 %" #i ", %" #i ", %8\n s_mov_b64 exec, -1\n v_cmp_ne_u32_e64 s[22:23], %11, %12\n v_mul_f64 %" #i ", %" #i ", %8\n v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define F_HB(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n v_fma_f64 %" #i ", %" #i ", %8, %9\n s_mov_b64 exec, s[22:23]\n v_mul_f64 %" #i ", %" #i ", %8\n s_mov_b64 exec, -1\n v_cmp_ne_u32_e64 s[20:21], %11, %12\n v_mul_f64 %" #i ", %" #i ", %8\n v_fma_f64 %" #i ", %" #i ", %8, %9\n"
    asm volatile("v_cmp_ne_u32_e64 s[20:21], %0, %1" : : "s"(ux), "v"(vy) : "s20", "s21");
    for (int i = 0; i < iters; ++i)
        asm volatile(F_HA(0) F_HB(1) F_HA(2) F_HB(3) F_HA(4) F_HB(5) F_HA(6) F_HB(7) F_HA(0) F_HB(1) F_HA(2) F_HB(3) F_HA(4) F_HB(5) F_HA(6) F_HB(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(B), "s"(ux), "v"(vy) : "vcc", "s20", "s21", "s22", "s23");
    FIN
}

// as above, but the masked multiplies of four cells are done back to back (four SGPR masks, ONE exec restore)
__global__ __launch_bounds__(256) void k_cell_f64_hoist4(double *out, int iters, double B, double C, uint32_t ux) {
    DECL
#define F_P1(i, m) "v_fma_f64 %" #i ", %" #i ", %8, %9\n v_fma_f64 %" #i ", %" #i ", %8, %9\n v_cmp_ne_u32_e64 " m ", %11, %12\n v_mul_f64 %" #i ", %" #i ", %8\n v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define F_P2(i, m) "s_mov_b64 exec, " m "\n v_mul_f64 %" #i ", %" #i ", %8\n"
    for (int i = 0; i < iters; ++i)
        asm volatile(F_P1(0, "s[20:21]") F_P1(1, "s[22:23]") F_P1(2, "s[24:25]") F_P1(3, "s[26:27]")
                     F_P2(0, "s[20:21]") F_P2(1, "s[22:23]") F_P2(2, "s[24:25]") F_P2(3, "s[26:27]") "s_mov_b64 exec, -1\n"
                     F_P1(4, "s[20:21]") F_P1(5, "s[22:23]") F_P1(6, "s[24:25]") F_P1(7, "s[26:27]")
                     F_P2(4, "s[20:21]") F_P2(5, "s[22:23]") F_P2(6, "s[24:25]") F_P2(7, "s[26:27]") "s_mov_b64 exec, -1\n"
                     F_P1(0, "s[20:21]") F_P1(1, "s[22:23]") F_P1(2, "s[24:25]") F_P1(3, "s[26:27]")
                     F_P2(0, "s[20:21]") F_P2(1, "s[22:23]") F_P2(2, "s[24:25]") F_P2(3, "s[26:27]") "s_mov_b64 exec, -1\n"
                     F_P1(4, "s[20:21]") F_P1(5, "s[22:23]") F_P1(6, "s[24:25]") F_P1(7, "s[26:27]")
                     F_P2(4, "s[20:21]") F_P2(5, "s[22:23]") F_P2(6, "s[24:25]") F_P2(7, "s[26:27]") "s_mov_b64 exec, -1\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(B), "s"(ux), "v"(vy) : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
    FIN
}

// f32 cell with the hoisted SGPR mask (the form that wins in f64)
__global__ __launch_bounds__(256) void k_cell_f32_hoist(double *out, int iters, double B, double C, uint32_t ux) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = (float)B, c = (float)C; uint32_t vy = threadIdx.x & 3;
    const long long c0 = clock64(), w0 = wall_clock64();
#define G_HA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n s_mov_b64 exec, s[20:21]\n v_mul_f32 %" #i ", %" #i ", %8\n s_mov_b64 exec, -1\n v_cmp_ne_u32_e64 s[22:23], %11, %12\n v_mul_f32 %" #i ", %" #i ", %8\n v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define G_HB(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n v_fma_f32 %" #i ", %" #i ", %8, %9\n s_mov_b64 exec, s[22:23]\n v_mul_f32 %" #i ", %" #i ", %8\n s_mov_b64 exec, -1\n v_cmp_ne_u32_e64 s[20:21], %11, %12\n v_mul_f32 %" #i ", %" #i ", %8\n v_fma_f32 %" #i ", %" #i ", %8, %9\n"
    asm volatile("v_cmp_ne_u32_e64 s[20:21], %0, %1" : : "s"(ux), "v"(vy) : "s20", "s21");
    for (int i = 0; i < iters; ++i)
        asm volatile(G_HA(0) G_HB(1) G_HA(2) G_HB(3) G_HA(4) G_HB(5) G_HA(6) G_HB(7) G_HA(0) G_HB(1) G_HA(2) G_HB(3) G_HA(4) G_HB(5) G_HA(6) G_HB(7)
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c), "s"(B), "s"(ux), "v"(vy) : "vcc", "s20", "s21", "s22", "s23");
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = a0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[1] = (double)(clock64() - c0); out[2] = (double)(wall_clock64() - w0); }
}

typedef void (*kern_t)(double *, int, double, double, uint32_t);
static void run(const char *name, kern_t k, double valu_per_group) {
    double *out; hipMalloc(&out, 64);
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps, iters = (int)(400000 / wps / (valu_per_group > 4 ? 4 : 1));
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9, 3u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9, 3u);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double cw[3]; hipMemcpy(cw, out, 24, hipMemcpyDeviceToHost);
        // block 0's own clocks: its SIMD ran `wps` waves side by side for the whole time (all blocks are equal)
        const double valu = (double)iters * 16 * valu_per_group * wps;
        const double ghz = cw[1] / cw[2] * 0.1;
        printf("%-36s waves/SIMD=%d  block0: %.2f clk/VALU per SIMD | kernel: %.2f ms -> %.2f clk/VALU per SIMD at %.2f GHz (%.3f G/s)\n",
               name, wps, cw[1] / valu, ms, ms * 1e6 * ghz / valu, ghz, valu / ms / 1e6);
    }
    hipFree(out);
}

int main() {
    run("v_fma_f64 v,v,v,v", k_fma3, 1);
    run("v_fma_f64 v,v,s,v", k_fma_s, 1);
    run("v_fma_f64 v,v,v(same),v", k_fma_aa, 1);
    run("v_mul_f64 v,v,v", k_mul2, 1);
    run("v_mul_f64 v,v,s", k_mul_s, 1);
    run("v_add_f64 v,v,v", k_add2, 1);
    run("v_mov_b64", k_mov, 1);
    run("cmpx + mul + s_mov exec (2 VALU)", k_cmpx, 2);
    run("cmp + mul (2 VALU)", k_cmp, 2);
    run("7-op cell body (7 VALU + s_mov)", k_cell, 7);
    run("v_cmp->sgpr, s_mov exec, mul, s_mov", k_smask, 2);
    run("same, groups of 4 (one restore)", (kern_t)k_smask4, 1);  // 16 VALU per iteration: count as 8+8 over 16 groups
    run("cmpx block + 1 independent fma", k_cmpx_il, 3);
    run("7-op cell body in f32", k_cell_f32, 7);
    run("f32 cell, cndmask select (8 VALU)", k_cell_f32_cnd, 8);
    run("f32 cell, hoisted v_cmp -> SGPR mask (7 VALU)", k_cell_f32_hoist, 7);
    run("f64 cell, cndmask select (9 VALU)", k_cell_f64_cnd, 9);
    run("f64 cell, hoisted v_cmp -> SGPR mask (7 VALU)", k_cell_f64_hoist, 7);
    run("f64 cell, masks of 4 cells, one restore (7 VALU)", k_cell_f64_hoist4, 7);
    return 0;
}
