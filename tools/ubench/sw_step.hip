// Developer micro-benchmark (gfx950): the Smith-Waterman step of phmm_sw_kernels.hip rebuilt piece by piece, to see which
// piece costs what the bare cell body (tools/ubench/sw_cell.hip) does not: FORM 0 = two row sets (ping-pong, the round-2
// body), 1 = one row set updated in place (the kernel's), 2 = 1 + flags in two accumulator pairs with the packed last word
// and the dwordx3 store, 3 = 2 + the base of the sweep read from LDS a step ahead + column-0 selects + last-column masks.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/sw_step.hip -o tools/ubench/sw_step && tools/ubench/sw_step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int K, int FORM>
__global__ __launch_bounds__(64) void sw_steps(int32_t *out, uint32_t *slab, size_t slab_stride, int steps, int32_t xm, int32_t xmm, int32_t xo, int32_t xe, uint32_t seed) {
    extern __shared__ unsigned char smem[];
    asm volatile("" : "+s"(xm), "+s"(xmm));
    const int lane = threadIdx.x, l = lane & 7;
    for (int i = lane; i < 2048; i += 64) smem[i] = (unsigned char)((i * 7 + seed) & 3);
    __syncthreads();
    int32_t up[K], up2[K], bgv[K];
    uint32_t bb[K], kmask[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        up[k] = up2[k] = 0;
        bgv[k] = INT32_MIN / 2;
        bb[k] = (threadIdx.x * 7 + k * 3 + seed) & 3;
        kmask[k] = k == 16 ? ~0u : 0u;
        asm volatile("" : "+v"(kmask[k]));
    }
    constexpr int NH = (K + 15) / 16, REM = K % 16;
    int32_t diag = 0, o_sw = 0, o_bgh = 0, lc_score = INT32_MIN, lc_row = 0;
    uint32_t acc_c[NH] = {}, acc_e[NH] = {}, sink = 0;
    uint32_t *bt = slab + (size_t)blockIdx.x * slab_stride + threadIdx.x * 3;
    int32_t a_next = smem[0];
    const int32_t x_open_s = xo, x_open_l = xo + 1;
    const long long c0 = clock64(), w0 = wall_clock64();
    auto step = [&](int t, int32_t (&U)[K], int32_t (&O)[K]) {
        int32_t left = __builtin_amdgcn_update_dpp(0, o_sw, 0x111, 0xf, 0xf, true);
        int32_t h_bg = __builtin_amdgcn_update_dpp(0, o_bgh, 0x111, 0xf, 0xf, true);
        const int i = t - l + 1;
        int32_t a_base;
        if constexpr (FORM >= 3) {
            a_base = a_next;
            a_next = smem[max(i, 0) & 2047];
            left = l == 0 ? 0 : left;
            h_bg = l == 0 ? (INT32_MIN / 2 | 1) : h_bg;
        } else {
            a_base = (t * 5 + (int)seed) & 3;
        }
        const int32_t diag_next = left;
        auto score = [&](int k) { return (uint32_t)a_base == bb[k] ? xm : xmm; };
        if constexpr (FORM == 0) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int32_t d = k ? U[k - 1] : diag;
                const int32_t sd = d + score(k);
                const int32_t pv = U[k] + xo, ev = bgv[k] + xe;
                acc_e[0] = __builtin_amdgcn_alignbit(acc_e[0], (uint32_t)(ev - pv), 31);
                bgv[k] = max(pv, ev);
                const int32_t ph = left + xo + 1, eh = h_bg + xe;
                acc_e[0] = __builtin_amdgcn_alignbit(acc_e[0], (uint32_t)(eh - ph), 31);
                h_bg = max(ph, eh);
                const int32_t cx = max(sd, max(h_bg, bgv[k]));
                acc_c[0] = __builtin_amdgcn_alignbit((uint32_t)cx, acc_c[0], 2);
                left = O[k] = cx & ~3;
            }
            sink ^= acc_c[0] + acc_e[0];
        } else {
            int32_t step_diag = diag + score(0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int32_t pv = U[k] + x_open_s;
                const int32_t next_diag = k + 1 < K ? U[k] + score(k + 1) : 0;
                const int32_t ev = bgv[k] + xe;
                acc_e[FORM >= 2 ? k / 16 : 0] = __builtin_amdgcn_alignbit(acc_e[FORM >= 2 ? k / 16 : 0], (uint32_t)(ev - pv), 31);
                bgv[k] = max(pv, ev);
                const int32_t ph = left + x_open_l, eh = h_bg + xe;
                acc_e[FORM >= 2 ? k / 16 : 0] = __builtin_amdgcn_alignbit(acc_e[FORM >= 2 ? k / 16 : 0], (uint32_t)(eh - ph), 31);
                h_bg = max(ph, eh);
                const int32_t cx = max(step_diag, max(h_bg, bgv[k]));
                acc_c[FORM >= 2 ? k / 16 : 0] = __builtin_amdgcn_alignbit((uint32_t)cx, acc_c[FORM >= 2 ? k / 16 : 0], 2);
                left = U[k] = cx & ~3;
                step_diag = next_diag;
            }
            if constexpr (FORM >= 2) {
                typedef uint32_t v3 __attribute__((ext_vector_type(3)));
                constexpr uint32_t LO = (1u << (2 * (REM & 15))) - 1u;
                v3 v = {acc_c[0], acc_e[0], (acc_c[NH - 1] & ~(~0u >> (2 * (REM & 15)))) | (acc_e[NH - 1] & LO)};
                uint32_t *r = bt + (size_t)t * 192;
                asm volatile("global_store_dwordx3 %0, %1, off nt" ::"v"(r), "v"(v) : "memory");
            } else {
                sink ^= acc_c[0] + acc_e[0];
            }
            if constexpr (FORM >= 3) {
                int32_t v = 0;
#pragma unroll
                for (int k = 0; k < K; ++k) v |= U[k] & (int32_t)kmask[k];
                const bool take = l == 7 && i >= 1 && v >= lc_score;
                lc_score = take ? v : lc_score;
                lc_row = take ? i : lc_row;
            }
        }
        diag = diag_next;
        o_sw = left;
        o_bgh = h_bg;
    };
    for (int t = 0; t < steps; t += 2) {
        step(t, up, up2);
        step(t + 1, FORM == 0 ? up2 : up, FORM == 0 ? up : up2);
    }
    const long long c1 = clock64();
    if (sink == 0x12345678u || lc_row == 0x7fffffff) out[2] = up[0] + lc_score;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = (int32_t)(c1 - c0);
        out[1] = (int32_t)(wall_clock64() - w0);  // 100 MHz
    }
}

template <int FORM>
void run(int32_t *d, uint32_t *slab, size_t stride, int waves_per_simd) {
    constexpr int K = 19;
    const int steps = 1232;
    const size_t lds = 160 * 1024 / (4 * waves_per_simd) - 512;
    hipFuncSetAttribute(reinterpret_cast<const void *>(sw_steps<K, FORM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((sw_steps<K, FORM>), dim3(blocks), dim3(64), lds, 0, d, slab, stride, steps, 42, -58, -120, -20, 3u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((sw_steps<K, FORM>), dim3(blocks), dim3(64), lds, 0, d, slab, stride, steps, 42, -58, -120, -20, 3u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    int32_t h[4];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const double mhz = (double)h[0] / ((double)h[1] / 100.0);
    const char *names[] = {"two row sets (round-2 body)", "one row set, in place", "+ flag words, dwordx3 store", "+ LDS base, column 0, last-column masks"};
    printf("waves/SIMD=%d %-42s: %.3f ms at %.0f MHz, one wave %6.1f clk per step -> %.2f TCUPS\n", waves_per_simd, names[FORM], ms, mhz,
           (double)h[0] / steps, (double)blocks * 64 * steps * K / (ms * 1e-3) / 1e12);
}

int main() {
    int32_t *d;
    hipMalloc(&d, 64);
    const size_t stride = (size_t)1232 * 192 + 64;
    uint32_t *slab;
    if (hipMalloc(&slab, stride * 4 * 256 * 4 * 4) != hipSuccess) return 1;
    for (int w : {1, 2, 3, 4}) {
        run<0>(d, slab, stride, w);
        run<1>(d, slab, stride, w);
        run<2>(d, slab, stride, w);
        run<3>(d, slab, stride, w);
    }
    return 0;
}
