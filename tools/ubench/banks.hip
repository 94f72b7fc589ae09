// Developer micro-benchmark (gfx950): does the VGPR bank of the three f64 FMA sources matter?
// Explicit physical registers; accumulators v[40+4i:41+4i] (bank pair 0/1 if banks = index mod 4).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/banks.hip -o /tmp/banks && /tmp/banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CLOB "v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71"
#define INIT "v_mov_b32 v28, %0\n v_mov_b32 v29, %1\n v_mov_b32 v30, %0\n v_mov_b32 v31, %1\n v_mov_b32 v32, %0\n v_mov_b32 v33, %1\n v_mov_b32 v34, %0\n v_mov_b32 v35, %1\n v_mov_b32 v36, %0\n v_mov_b32 v37, %1\n v_mov_b32 v38, %0\n v_mov_b32 v39, %1\n" \
             "v_mov_b32 v40, %0\n v_mov_b32 v41, %1\n v_mov_b32 v42, %0\n v_mov_b32 v43, %1\n v_mov_b32 v44, %0\n v_mov_b32 v45, %1\n v_mov_b32 v46, %0\n v_mov_b32 v47, %1\n v_mov_b32 v48, %0\n v_mov_b32 v49, %1\n v_mov_b32 v50, %0\n v_mov_b32 v51, %1\n" \
             "v_mov_b32 v52, %0\n v_mov_b32 v53, %1\n v_mov_b32 v54, %0\n v_mov_b32 v55, %1\n v_mov_b32 v56, %0\n v_mov_b32 v57, %1\n v_mov_b32 v58, %0\n v_mov_b32 v59, %1\n v_mov_b32 v60, %0\n v_mov_b32 v61, %1\n v_mov_b32 v62, %0\n v_mov_b32 v63, %1\n" \
             "v_mov_b32 v64, %0\n v_mov_b32 v65, %1\n v_mov_b32 v66, %0\n v_mov_b32 v67, %1\n v_mov_b32 v68, %0\n v_mov_b32 v69, %1\n v_mov_b32 v70, %0\n v_mov_b32 v71, %1\n"
// 8 accumulators on bank pair (0,1): v[40:41], v[44:45], ... v[68:69]
#define ACC8(F) F("40:41") F("44:45") F("48:49") F("52:53") F("56:57") F("60:61") F("64:65") F("68:69")
// 8 accumulators alternating bank pairs: v[40:41], v[42:43], ...
#define ACC8ALT(F) F("40:41") F("42:43") F("44:45") F("46:47") F("48:49") F("50:51") F("52:53") F("54:55")

#define KERN(name, BODY)                                                                        \
    __global__ __launch_bounds__(256) void name(double *out, int iters, int lo, int hi) {      \
        const long long c0 = clock64();                                                         \
        asm volatile(INIT : : "v"(lo), "v"(hi) : CLOB);                                         \
        for (int i = 0; i < iters; ++i) asm volatile(BODY BODY : : : CLOB);                     \
        double r;                                                                               \
        asm volatile("v_mov_b32 %0, v40\n" : "=v"(lo) : : CLOB);                                \
        if (lo == 12345) out[3] = 1.0;                                                          \
        if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = (double)(clock64() - c0);             \
        (void)r;                                                                                \
    }
#define F_ALL01(a) "v_fma_f64 v[" a "], v[" a "], v[32:33], v[36:37]\n"   /* src0 01, src1 01, src2 01 */
#define F_01_23_01(a) "v_fma_f64 v[" a "], v[" a "], v[30:31], v[36:37]\n" /* src0 01, src1 23, src2 01 */
#define F_01_23_23(a) "v_fma_f64 v[" a "], v[" a "], v[30:31], v[34:35]\n" /* src0 01, src1 23, src2 23 */
#define F_FMAC(a) "v_fmac_f64_e32 v[" a "], v[30:31], v[34:35]\n"          /* acc += b*c : src0 23, src1 23, src2=dst 01 */
#define F_FMAC2(a) "v_fmac_f64_e32 v[" a "], v[30:31], v[32:33]\n"         /* src0 23, src1 01, dst 01 */
#define F_MUL_01_01(a) "v_mul_f64 v[" a "], v[" a "], v[32:33]\n"
#define F_MUL_01_23(a) "v_mul_f64 v[" a "], v[" a "], v[30:31]\n"
#define F_FMA_ACC2(a) "v_fma_f64 v[" a "], v[30:31], v[34:35], v[" a "]\n"   /* VOP3, accumulator as src2 (what fmac does) */
#define F_FMAC_E64(a) "v_fmac_f64_e64 v[" a "], v[30:31], v[34:35]\n"
KERN(k_fma_acc2, ACC8(F_FMA_ACC2))
KERN(k_fmac_e64, ACC8(F_FMAC_E64))
KERN(k_all01, ACC8(F_ALL01))
KERN(k_01_23_01, ACC8(F_01_23_01))
KERN(k_01_23_23, ACC8(F_01_23_23))
KERN(k_alt_01_23_01, ACC8ALT(F_01_23_01))
KERN(k_fmac, ACC8(F_FMAC))
KERN(k_fmac2, ACC8(F_FMAC2))
KERN(k_mul_same, ACC8(F_MUL_01_01))
KERN(k_mul_diff, ACC8(F_MUL_01_23))

typedef void (*kern_t)(double *, int, int, int);
static void run(const char *name, kern_t k) {
    double *out; hipMalloc(&out, 64);
    for (int wps : {1, 2}) {
        const int blocks = 256 * wps, iters = 300000 / wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 0x11111111, 0x3ff00000);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 0x11111111, 0x3ff00000);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double valu = (double)iters * 16 * wps;
        printf("%-44s waves/SIMD=%d  %.3f G instr/s per SIMD  (%.2f ms)\n", name, wps, valu / ms / 1e6, ms);
    }
    hipFree(out);
}
int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run("fma  VOP3 b*c + acc (acc = src2 = dst)", k_fma_acc2);
        run("fmac e64 acc += b*c", k_fmac_e64);
        run("fma  src0 01 src1 01 src2 01", k_all01);
        run("fma  src0 01 src1 23 src2 01", k_01_23_01);
        run("fma  src0 01 src1 23 src2 23", k_01_23_23);
        run("fma  (acc alternating) 01/23 23 01", k_alt_01_23_01);
        run("fmac src0 23 src1 23 dst 01", k_fmac);
        run("fmac src0 23 src1 01 dst 01", k_fmac2);
        run("mul  src0 01 src1 01", k_mul_same);
        run("mul  src0 01 src1 23", k_mul_diff);
    }
    return 0;
}
