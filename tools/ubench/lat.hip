// Developer micro-benchmark (gfx950): issue / dependent latencies of the instructions the PairHMM row update is
// made of.  One wave per block, one block: pure latency; 2 waves on one SIMD are not forced here.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lat.hip -o gpurun_out/lat && gpurun_out/lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x

__global__ void k_dep_fma(double *out, uint64_t *clk, int iters, double b, double c) {
    double a = out[threadIdx.x];
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));)
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
__global__ void k_ind_fma(double *out, uint64_t *clk, int iters, double b, double c) {
    double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
            "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
            "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
            "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(b), "v"(c));
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
// two interleaved dependent chains
__global__ void k_dep2_fma(double *out, uint64_t *clk, int iters, double b, double c) {
    double a0 = out[threadIdx.x], a1 = a0 + 1;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        REP16(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));)
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a0 + a1;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
// the EXEC-masked select block, independent destinations
__global__ void k_cmpx_block(double *out, uint64_t *clk, int iters, double b, uint32_t x) {
    double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    uint32_t y = threadIdx.x & 3;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define BLK(r) asm volatile("v_cmpx_ne_u32_e32 vcc, %1, %2\n v_mul_f64 %0, %3, %0\n s_mov_b64 exec, -1" : "+v"(r) : "v"(x), "v"(y), "v"(b) : "vcc");
        BLK(a0) BLK(a1) BLK(a2) BLK(a3) BLK(a0) BLK(a1) BLK(a2) BLK(a3) BLK(a0) BLK(a1) BLK(a2) BLK(a3) BLK(a0) BLK(a1) BLK(a2) BLK(a3)
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
// same select with v_cmp + 2 x v_cndmask + v_mul
__global__ void k_cnd_block(double *out, uint64_t *clk, int iters, double b, uint32_t x) {
    double a0 = out[threadIdx.x], a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    uint32_t y = threadIdx.x & 3;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#define BLC(r) { double pr = (x != y + (uint32_t)i) ? b : 1.0; asm volatile("v_mul_f64 %0, %1, %0" : "+v"(r) : "v"(pr)); }
        BLC(a0) BLC(a1) BLC(a2) BLC(a3) BLC(a0) BLC(a1) BLC(a2) BLC(a3) BLC(a0) BLC(a1) BLC(a2) BLC(a3) BLC(a0) BLC(a1) BLC(a2) BLC(a3)
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}
// dependent DPP + fma (the cross-lane hop of the D chain)
__global__ void k_dep_dpp(double *out, uint64_t *clk, int iters, double b, double c) {
    double a = out[threadIdx.x];
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        REP16({ int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x111, 0xf, 0xf, true);
                int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x111, 0xf, 0xf, true);
                double l = __hiloint2double(hi, lo);
                asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a) : "v"(l), "v"(b), "v"(c)); })
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = a;
    if (threadIdx.x == 0) clk[0] = t1 - t0;
}

template <class F>
static void run(const char *name, F f, int waves, double per_iter_instr) {
    double *out; uint64_t *clk;
    hipMalloc(&out, 4096 * sizeof(double)); hipMalloc(&clk, 8);
    hipMemset(out, 0, 4096 * sizeof(double));
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(out, clk, 10, waves);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f(out, clk, iters, waves);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    uint64_t c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    // s_memtime / readcyclecounter runs at a fixed 100 MHz on gfx9: use wall time * 2.4 GHz as the clock estimate
    printf("%-34s waves/blk=%d  %.1f ns/iter  ~%.1f clk/instr-group (at 2.4 GHz, %g groups/iter)\n", name, waves,
           ms * 1e6 / iters, ms * 1e6 / iters * 2.4 / per_iter_instr, per_iter_instr);
    hipFree(out); hipFree(clk);
}

int main() {
    for (int waves : {1, 4, 8}) {  // waves per block on ONE CU: 1 = one SIMD one wave; 4 = one wave per SIMD; 8 = two per SIMD
        run("dependent v_fma_f64", [](double *o, uint64_t *c, int it, int w) { hipLaunchKernelGGL(k_dep_fma, dim3(1), dim3(64 * w), 0, 0, o, c, it, 1.0000001, 1e-9); }, waves, 16);
        run("2 interleaved dependent chains", [](double *o, uint64_t *c, int it, int w) { hipLaunchKernelGGL(k_dep2_fma, dim3(1), dim3(64 * w), 0, 0, o, c, it, 1.0000001, 1e-9); }, waves, 32);
        run("8 independent v_fma_f64", [](double *o, uint64_t *c, int it, int w) { hipLaunchKernelGGL(k_ind_fma, dim3(1), dim3(64 * w), 0, 0, o, c, it, 1.0000001, 1e-9); }, waves, 16);
        run("cmpx+mul+s_mov block (4 dst)", [](double *o, uint64_t *c, int it, int w) { hipLaunchKernelGGL(k_cmpx_block, dim3(1), dim3(64 * w), 0, 0, o, c, it, 1.0000001, 2u); }, waves, 16);
        run("cmp+2cndmask+mul block (4 dst)", [](double *o, uint64_t *c, int it, int w) { hipLaunchKernelGGL(k_cnd_block, dim3(1), dim3(64 * w), 0, 0, o, c, it, 1.0000001, 2u); }, waves, 16);
        run("dependent dpp+dpp+fma", [](double *o, uint64_t *c, int it, int w) { hipLaunchKernelGGL(k_dep_dpp, dim3(1), dim3(64 * w), 0, 0, o, c, it, 1.0000001, 1e-9); }, waves, 16);
    }
    return 0;
}
