// Developer micro-benchmark (gfx950): the Smith-Waterman cell body of phmm_sw_kernels.hip alone -- K cells per step, the
// row state running through them, no memory traffic, no branches -- at 1 ... 8 waves per SIMD on a full chip: shader
// clocks per wave-level cell and per VALU instruction (16 per cell), i.e. the ceiling the kernel's inner loop has.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/sw_cell.hip -o /tmp/sw_cell && /tmp/sw_cell
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int K>
__global__ __launch_bounds__(64) void sw_cells(int32_t *out, int steps, int32_t xm, int32_t xmm, int32_t xo, int32_t xe, uint32_t seed) {
    asm volatile("" : "+s"(xm), "+s"(xmm));
    int32_t up_a[K], up_b[K], bgv[K], bb[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        up_a[k] = up_b[k] = 0;
        bgv[k] = INT32_MIN / 2;
        bb[k] = (threadIdx.x * 7 + k * 3 + seed) & 3;
    }
    int32_t diag = 0, o_sw = 0, o_bgh = 0;
    uint32_t acc_c = 0, acc_e = 0, sink = 0;
    const long long c0 = clock64(), w0 = wall_clock64();
    auto step = [&](int t, const int32_t (&up)[K], int32_t (&o)[K]) {
        int32_t left = __builtin_amdgcn_update_dpp(0, o_sw, 0x111, 0xf, 0xf, true);
        int32_t h_bg = __builtin_amdgcn_update_dpp(0, o_bgh, 0x111, 0xf, 0xf, true);
        const int32_t a_base = (t * 5 + (int)seed) & 3;
        const int32_t diag_next = left;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int32_t d = k ? up[k - 1] : diag;
            const int32_t sd = d + (a_base == bb[k] ? xm : xmm);
            const int32_t pv = up[k] + xo, ev = bgv[k] + xe;
            acc_e = __builtin_amdgcn_alignbit(acc_e, (uint32_t)(ev - pv), 31);
            bgv[k] = max(pv, ev);
            const int32_t ph = left + xo + 1, eh = h_bg + xe;
            acc_e = __builtin_amdgcn_alignbit(acc_e, (uint32_t)(eh - ph), 31);
            h_bg = max(ph, eh);
            const int32_t cx = max(sd, max(h_bg, bgv[k]));
            acc_c = __builtin_amdgcn_alignbit((uint32_t)cx, acc_c, 2);
            left = o[k] = cx & ~3;
        }
        sink ^= acc_c + acc_e;
        diag = diag_next;
        o_sw = left;
        o_bgh = h_bg;
    };
    for (int t = 0; t < steps; t += 2) {
        step(t, up_a, up_b);
        step(t + 1, up_b, up_a);
    }
    const long long c1 = clock64();
    if (sink == 0x12345678u) out[2] = up_a[0];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = (int32_t)(c1 - c0);
        out[1] = (int32_t)(wall_clock64() - w0);  // 100 MHz
    }
}

template <int K>
void run(int32_t *d, int waves_per_simd) {
    const int steps = 4000;
    // occupancy is set through dynamic LDS: 160 KB per CU / (4 x waves per SIMD) per block of one wave; the grid is
    // exactly what the chip holds at once, so the kernel's duration is the time the SIMDs needed for all of it
    const size_t lds = 160 * 1024 / (4 * waves_per_simd) - 512;
    hipFuncSetAttribute(reinterpret_cast<const void *>(sw_cells<K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(sw_cells<K>, dim3(blocks), dim3(64), lds, 0, d, steps, 42, -58, -120, -20, 3u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(sw_cells<K>, dim3(blocks), dim3(64), lds, 0, d, steps, 42, -58, -120, -20, 3u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    int32_t h[4];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const double mhz = (double)h[0] / ((double)h[1] / 100.0);                    // shader clock while the kernel ran
    const double clk_cell = ms * 1e-3 * mhz * 1e6 / ((double)steps * K * waves_per_simd);  // SIMD clocks per wave-level cell
    printf("K=%2d waves/SIMD=%d: %.3f ms at %.0f MHz, one wave saw %6.1f clk per step; %5.2f clk per cell of the SIMD (%.2f per VALU instruction, 16 + 2/K per cell) -> %.2f TCUPS\n",
           K, waves_per_simd, ms, mhz, (double)h[0] / steps, clk_cell, clk_cell / (16.0 + 2.0 / K),
           (double)blocks * 64 * steps * K / (ms * 1e-3) / 1e12);
}

int main() {
    int32_t *d;
    hipMalloc(&d, 64);
    for (int w : {1, 2, 3, 4, 5, 8}) run<10>(d, w);
    for (int w : {1, 2, 3}) run<19>(d, w);
    return 0;
}
