// Sustained issue rate on gfx950 of the packed-f16 / dot / mixed-precision VALU opcodes a cheaper-arithmetic EASU would use
// (companion of scripts/valu_peak.hip; same method: 8 independent chains per wave, 4096 x 256 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = __uint_as_float(0x3c003c01u), c = __uint_as_float(0x38003800u);      // packed halves (1.0, 1.001) and (0.5, 0.5)
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
        {
#define LVK_ASM8(INS) do { asm volatile(INS : "+v"(a0) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a1) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a2) : "v"(b), "v"(c)); \
                          asm volatile(INS : "+v"(a3) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a4) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a5) : "v"(b), "v"(c)); \
                          asm volatile(INS : "+v"(a6) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a7) : "v"(b), "v"(c)); } while (0)
#define LVK_MIX(X, Y) do { asm volatile(X : "+v"(a0) : "v"(b), "v"(c)); asm volatile(Y : "+v"(a4) : "v"(b), "v"(c)); asm volatile(X : "+v"(a1) : "v"(b), "v"(c)); \
                           asm volatile(Y : "+v"(a5) : "v"(b), "v"(c)); asm volatile(X : "+v"(a2) : "v"(b), "v"(c)); asm volatile(Y : "+v"(a6) : "v"(b), "v"(c)); \
                           asm volatile(X : "+v"(a3) : "v"(b), "v"(c)); asm volatile(Y : "+v"(a7) : "v"(b), "v"(c)); } while (0)
            if (MODE == 0) LVK_ASM8("v_fma_f32 %0, %0, %1, %2");
            if (MODE == 1) LVK_ASM8("v_pk_fma_f16 %0, %0, %1, %2");
            if (MODE == 2) LVK_ASM8("v_pk_mul_f16 %0, %0, %1");
            if (MODE == 3) LVK_ASM8("v_pk_add_f16 %0, %0, %1");
            if (MODE == 4) LVK_ASM8("v_pk_min_f16 %0, %0, %1");
            if (MODE == 5) LVK_ASM8("v_pk_max_f16 %0, %0, %1");
            if (MODE == 6) LVK_ASM8("v_dot2_f32_f16 %0, %1, %2, %0");
            if (MODE == 7) LVK_ASM8("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]");
            if (MODE == 8) LVK_ASM8("v_cvt_pkrtz_f16_f32 %0, %0, %1");
            if (MODE == 9) LVK_ASM8("v_cvt_f16_f32 %0, %0");
            if (MODE == 10) LVK_ASM8("v_cvt_f32_f16 %0, %0");
            if (MODE == 11) LVK_ASM8("v_rcp_f16 %0, %0");
            if (MODE == 12) LVK_ASM8("v_rsq_f16 %0, %0");
            if (MODE == 13) LVK_ASM8("v_dot2_i32_i16 %0, %1, %2, %0");
            if (MODE == 14) LVK_ASM8("v_dot4_i32_i8 %0, %1, %2, %0");
            if (MODE == 15) LVK_ASM8("v_pk_mul_lo_u16 %0, %0, %1");
            if (MODE == 16) LVK_ASM8("v_pk_add_u16 %0, %0, %1");
            if (MODE == 17) LVK_MIX("v_pk_fma_f16 %0, %0, %1, %2", "v_fma_f32 %0, %0, %1, %2");
            if (MODE == 18) LVK_MIX("v_pk_fma_f16 %0, %0, %1, %2", "v_perm_b32 %0, %0, %1, %2");
            if (MODE == 19) LVK_MIX("v_dot2_f32_f16 %0, %1, %2, %0", "v_pk_fma_f16 %0, %0, %1, %2");
            if (MODE == 20) LVK_ASM8("v_fma_mixlo_f16 %0, %1, %2, %0");
            if (MODE == 21) LVK_ASM8("v_pk_fma_f16 %0, %0, %1, %2 op_sel_hi:[1,0,1]");
            if (MODE == 22) LVK_ASM8("v_mad_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int MODE> void run(const char* name)
{
    float* d; (void)hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 500, blocks = 4096;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f); (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = (double)blocks * 256 * iters * 16 * 8;
    printf("%-34s %8.3f ms  %7.2f T lane-instr/s\n", name, ms, lane_ops / ms / 1e9);
    (void)hipFree(d);
}
int main()
{
    run<0>("v_fma_f32"); run<1>("v_pk_fma_f16"); run<2>("v_pk_mul_f16"); run<3>("v_pk_add_f16"); run<4>("v_pk_min_f16"); run<5>("v_pk_max_f16");
    run<6>("v_dot2_f32_f16"); run<7>("v_fma_mix_f32 (f16,f16,f32)"); run<8>("v_cvt_pkrtz_f16_f32"); run<9>("v_cvt_f16_f32"); run<10>("v_cvt_f32_f16");
    run<11>("v_rcp_f16"); run<12>("v_rsq_f16"); run<13>("v_dot2_i32_i16"); run<14>("v_dot4_i32_i8"); run<15>("v_pk_mul_lo_u16"); run<16>("v_pk_add_u16");
    run<17>("pk_fma_f16 : fma_f32 1:1"); run<18>("pk_fma_f16 : perm 1:1"); run<19>("dot2_f32_f16 : pk_fma_f16 1:1"); run<20>("v_fma_mixlo_f16"); run<21>("v_pk_fma_f16 op_sel"); 
    return 0;
}
