// Developer micro-benchmark (gfx950): issue rate of single VALU instructions on a full chip -- 8 independent chains per
// lane, N waves per SIMD -- in SIMD clocks per wave64 instruction.  What decides whether a 32-bit integer recurrence
// (Smith-Waterman) or its exact restatement in f32 (scores below 2^24) is the cheaper one, and what packed f32 buys.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/rates.hip -o /tmp/rates && /tmp/rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

enum Op { ADD_U32, MAX_I32, MAX3_I32, ADD_F32, MAX_F32, MAX3_F32, FMA_F32, PK_FMA_F32, PK_ADD_F32, PK_MAX_F16, ALIGNBIT, CNDMASK, AND_B32, CMP_U32, FMA_F64,
          SUB_F32, MUL_F32, PK_MUL_F32, ADD_I16_PK, MAX_I16_PK,
          SUB_CO, ADDC_CO, CMP_CND, LSHL_ADD, LSHL_OR, OR_B32, XOR_B32, LSHLREV, MAD_U24, ADD3, MAX_U32, MIN_I32, BFE, AND_OR, MOV, MOV_DPP, SUBCO_ADDC_CND, ADD_SDWA, SUB_ALIGN_MAX, CMP_ADDC_CND, N_OPS };
static const char *kNames[] = {"v_add_u32", "v_max_i32", "v_max3_i32", "v_add_f32", "v_max_f32", "v_max3_f32", "v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32",
                               "v_pk_max_f16", "v_alignbit_b32", "v_cndmask_b32", "v_and_b32", "v_cmp_ne_u32", "v_fma_f64", "v_sub_f32", "v_mul_f32", "v_pk_mul_f32",
                               "v_pk_add_i16", "v_pk_max_i16",
                               "v_sub_co_u32", "v_addc_co_u32", "v_cmp+v_cndmask (pair)", "v_lshl_add_u32", "v_lshl_or_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32",
                               "v_mad_u32_u24", "v_add3_u32", "v_max_u32", "v_min_i32", "v_bfe_u32", "v_and_or_b32", "v_mov_b32", "v_mov_b32 dpp row_shr:1",
                               "sub_co+addc+cndmask (triple)", "v_add_u32 sdwa", "sub+alignbit+max (triple)", "cmp+addc+cndmask (triple)"};

template <int OP>
__global__ __launch_bounds__(64) void rate(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[8];
    double d[8];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = threadIdx.x * 3 + i + seed;
        d[i] = 1.0 + a[i] * 1e-9;
        p[i] = f2{(float)a[i], 1.f + i};
    }
    uint32_t b = seed * 7 + 1, c = seed + 5;
    f2 pb = {1.0001f, 0.9999f}, pc = {1e-3f, 2e-3f};
    double db = 1.0000001, dc = 1e-9;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#define ONE(i)                                                                                                             \
    if constexpr (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                               \
    else if constexpr (OP == MAX_I32) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                          \
    else if constexpr (OP == MAX3_I32) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));            \
    else if constexpr (OP == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                          \
    else if constexpr (OP == SUB_F32) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                          \
    else if constexpr (OP == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                          \
    else if constexpr (OP == MAX_F32) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                          \
    else if constexpr (OP == MAX3_F32) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));            \
    else if constexpr (OP == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));              \
    else if constexpr (OP == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));      \
    else if constexpr (OP == PK_ADD_F32) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));                   \
    else if constexpr (OP == PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));                   \
    else if constexpr (OP == PK_MAX_F16) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                    \
    else if constexpr (OP == ADD_I16_PK) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                    \
    else if constexpr (OP == MAX_I16_PK) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));                    \
    else if constexpr (OP == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[i]) : "v"(b));                \
    else if constexpr (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );              \
    else if constexpr (OP == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                          \
    else if constexpr (OP == CMP_U32) asm volatile("v_cmp_ne_u32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");              \
    else if constexpr (OP == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(dc));          \
    else if constexpr (OP == SUB_CO) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");           \
    else if constexpr (OP == ADDC_CO) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a[i]) : : "vcc");            \
    else if constexpr (OP == CMP_CND) asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"); \
    else if constexpr (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));                 \
    else if constexpr (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));                   \
    else if constexpr (OP == OR_B32) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                            \
    else if constexpr (OP == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                          \
    else if constexpr (OP == LSHLREV) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a[i]));                                \
    else if constexpr (OP == MAD_U24) asm volatile("v_mad_u32_u24 %0, %0, 4, %1" : "+v"(a[i]) : "v"(b));                   \
    else if constexpr (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                \
    else if constexpr (OP == MAX_U32) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                          \
    else if constexpr (OP == MIN_I32) asm volatile("v_min_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                          \
    else if constexpr (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 3, 20" : "+v"(a[i]));                                     \
    else if constexpr (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));            \
    else if constexpr (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));                                   \
    else if constexpr (OP == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i])); \
    else if constexpr (OP == SUBCO_ADDC_CND) asm volatile("v_sub_co_u32 %2, vcc, %0, %1\n\tv_addc_co_u32 %3, s[20:21], %3, %3, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]), "+v"(b), "=&v"(c), "+v"(a[(i + 1) & 7]) : : "vcc", "s20", "s21"); \
    else if constexpr (OP == SUB_ALIGN_MAX) asm volatile("v_sub_u32 %2, %0, %1\n\tv_alignbit_b32 %3, %3, %2, 31\n\tv_max_i32 %0, %0, %1" : "+v"(a[i]), "+v"(b), "=&v"(c), "+v"(a[(i + 1) & 7]) : : ); \
    else if constexpr (OP == CMP_ADDC_CND) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_addc_co_u32 %2, s[20:21], %2, %2, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]), "+v"(b), "+v"(a[(i + 1) & 7]) : : "vcc", "s20", "s21"); \
    else if constexpr (OP == ADD_SDWA) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(a[i]) : "v"(b));
            REP8(ONE)
#undef ONE
        }
    }
    const long long c1 = clock64();
    uint32_t sink = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) sink ^= a[i] ^ (uint32_t)d[i] ^ (uint32_t)p[i].x ^ (uint32_t)p[i].y;
    if (sink == 0x12345678u) out[2] = sink;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[0] = (uint32_t)(c1 - c0);
        out[1] = (uint32_t)(wall_clock64() - w0);
    }
}

template <int OP>
void run(uint32_t *dbuf, int wps) {
    const int iters = 20000;
    const size_t lds = 160 * 1024 / (4 * wps) - 512;
    hipFuncSetAttribute(reinterpret_cast<const void *>(rate<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * 4 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(64), lds, 0, dbuf, iters, 3u);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(64), lds, 0, dbuf, iters, 3u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    uint32_t h[4];
    hipMemcpy(h, dbuf, sizeof h, hipMemcpyDeviceToHost);
    const double mhz = (double)h[0] / ((double)h[1] / 100.0);
    const double instr = (double)iters * 32 * wps;  // per SIMD
    printf("%-16s %d waves/SIMD: %5.2f clk per wave64 instruction  (%.0f MHz, %.3f ms)\n", kNames[OP], wps, ms * 1e-3 * mhz * 1e6 / instr, mhz, ms);
}

template <int OP>
void all(uint32_t *d) {
    if constexpr (OP < N_OPS) {
        if (!getenv("RATES_FROM") || OP >= atoi(getenv("RATES_FROM")))
            for (int w : {1, 2, 4}) run<OP>(d, w);
        all<OP + 1>(d);
    }
}

int main() {
    uint32_t *d;
    hipMalloc(&d, 64);
    all<0>(d);
    return 0;
}
