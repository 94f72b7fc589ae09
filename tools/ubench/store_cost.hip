// Developer micro-benchmark (gfx950): what the backtrack-flag stores cost the Smith-Waterman sweep.  The cell body of
// tools/ubench/sw_cell.hip (19 cells per step) plus, per step and wave, 768 bytes of flags written as (1) three dword stores
// 256 bytes apart, (2) one dwordx3 per lane, (3) the same into a 12 KB region per block (L2-resident), (4) plain stores --
// against (0) no store at all; 4 096 waves (4 per SIMD), 1 232 steps each = the production kernel's shape.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/store_cost.hip -o tools/ubench/store_cost && tools/ubench/store_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

template <int K, int MODE>
__global__ __launch_bounds__(64) void sw_cells(int32_t *out, uint32_t *slab, size_t slab_stride, int steps, int32_t xm, int32_t xmm, int32_t xo, int32_t xe, uint32_t seed) {
    asm volatile("" : "+s"(xm), "+s"(xmm));
    int32_t up_a[K], up_b[K], bgv[K], bb[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        up_a[k] = up_b[k] = 0;
        bgv[k] = INT32_MIN / 2;
        bb[k] = (threadIdx.x * 7 + k * 3 + seed) & 3;
    }
    int32_t diag = 0, o_sw = 0, o_bgh = 0;
    uint32_t acc_c = 0, acc_e = 0, acc_x = 0, sink = 0;
    uint32_t *bt = slab + (size_t)blockIdx.x * slab_stride + (MODE == 1 ? threadIdx.x : threadIdx.x * 3);
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
        acc_x += acc_c ^ acc_e;
        if constexpr (MODE == 0) {
            sink ^= acc_c + acc_e + acc_x;
        } else if constexpr (MODE == 1) {   // three dword stores, [dword][lane]
            uint32_t *r = bt + (size_t)t * 192;
            __builtin_nontemporal_store(acc_c, r);
            __builtin_nontemporal_store(acc_e, r + 64);
            __builtin_nontemporal_store(acc_x, r + 128);
        } else {                            // one dwordx3 per lane, [lane][dword]; MODE 3: wrapped into 12 KB per block
            typedef uint32_t v3 __attribute__((ext_vector_type(3)));
            v3 v = {acc_c, acc_e, acc_x};
            uint32_t *r = bt + (size_t)(MODE == 3 ? (t & 15) : t) * 192;
            if constexpr (MODE == 4) { *(r) = acc_c; r[1] = acc_e; r[2] = acc_x; }   // plain (temporal) stores
            else asm volatile("global_store_dwordx3 %0, %1, off nt" ::"v"(r), "v"(v) : "memory");
        }
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


template <int MODE>
void run(int32_t *d, uint32_t *slab, size_t stride, int waves_per_simd) {
    constexpr int K = 19;
    const int steps = 1232;
    const size_t lds = 160 * 1024 / (4 * waves_per_simd) - 512;
    hipFuncSetAttribute(reinterpret_cast<const void *>(sw_cells<K, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((sw_cells<K, MODE>), dim3(blocks), dim3(64), lds, 0, d, slab, stride, steps, 42, -58, -120, -20, 3u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((sw_cells<K, MODE>), dim3(blocks), dim3(64), lds, 0, d, slab, stride, steps, 42, -58, -120, -20, 3u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    int32_t h[4];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const double mhz = (double)h[0] / ((double)h[1] / 100.0);
    const char *names[] = {"no store", "3 x dword, 256 B apart", "dwordx3 per lane", "dwordx3, 12 KB region per block", "3 plain dword stores per lane"};
    printf("waves/SIMD=%d %-34s: %.3f ms at %.0f MHz -> %.2f TCUPS; %.1f GB written, %.0f GB/s\n", waves_per_simd, names[MODE], ms, mhz,
           (double)blocks * 64 * steps * K / (ms * 1e-3) / 1e12, MODE ? blocks * (double)steps * 768 / 1e9 : 0.0,
           MODE ? blocks * (double)steps * 768 / 1e9 / (ms * 1e-3) : 0.0);
}

int main() {
    int32_t *d;
    hipMalloc(&d, 64);
    const size_t stride = (size_t)1232 * 192 + 64;   // dwords per block
    uint32_t *slab;
    if (hipMalloc(&slab, stride * 4 * 256 * 4 * 5) != hipSuccess) return 1;
    for (int w : {2, 4, 5}) {
        run<0>(d, slab, stride, w);
        run<1>(d, slab, stride, w);
        run<2>(d, slab, stride, w);
        run<3>(d, slab, stride, w);
        run<4>(d, slab, stride, w);
    }
    return 0;
}
