// Micro-benchmark: sustained VALU issue rate on gfx950 for the instruction classes the EASU kernel uses.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_peak scripts/valu_peak.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0000001f, c = 0.5f;
    v2f p0{a0, a1}, p1{a2, a3}, p2{a4, a5}, p3{a6, a7}; const v2f pb{b, b}, pc{c, c};
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
        {
            if (MODE == 0) { a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
                             a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c); }
            if (MODE == 1) { p0 = __builtin_elementwise_fma(p0, pb, pc); p1 = __builtin_elementwise_fma(p1, pb, pc); p2 = __builtin_elementwise_fma(p2, pb, pc); p3 = __builtin_elementwise_fma(p3, pb, pc);
                             p0 = __builtin_elementwise_fma(p0, pb, pc); p1 = __builtin_elementwise_fma(p1, pb, pc); p2 = __builtin_elementwise_fma(p2, pb, pc); p3 = __builtin_elementwise_fma(p3, pb, pc); }
            if (MODE == 2) { a0 = __builtin_fminf(a0 * b, c + a1); a1 = __builtin_fminf(a1 * b, c + a2); a2 = __builtin_fminf(a2 * b, c + a3); a3 = __builtin_fminf(a3 * b, c + a0);
                             a4 = __builtin_fminf(a4 * b, c + a5); a5 = __builtin_fminf(a5 * b, c + a6); a6 = __builtin_fminf(a6 * b, c + a7); a7 = __builtin_fminf(a7 * b, c + a4); }
#define LVK_ASM8(INS) do { asm volatile(INS : "+v"(a0) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a1) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a2) : "v"(b), "v"(c)); \
                          asm volatile(INS : "+v"(a3) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a4) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a5) : "v"(b), "v"(c)); \
                          asm volatile(INS : "+v"(a6) : "v"(b), "v"(c)); asm volatile(INS : "+v"(a7) : "v"(b), "v"(c)); } while (0)
            if (MODE == 3) LVK_ASM8("v_cvt_i32_f32 %0, %0");
            if (MODE == 4) LVK_ASM8("v_cvt_f32_ubyte1 %0, %0");
            if (MODE == 5) LVK_ASM8("v_lshl_or_b32 %0, %0, 1, %1");
            if (MODE == 6) LVK_ASM8("v_perm_b32 %0, %0, %1, %2");
            if (MODE == 7) LVK_ASM8("v_max3_f32 %0, %0, %1, %2");
            if (MODE == 8) LVK_ASM8("v_max_f32 %0, %0, %1");
            if (MODE == 9) LVK_ASM8("v_mov_b32 %0, %1");
            if (MODE == 10) LVK_ASM8("v_add_u32 %0, %0, %1");
            if (MODE == 11) LVK_ASM8("v_lshlrev_b32 %0, 1, %0");
            if (MODE == 12) LVK_ASM8("v_cndmask_b32 %0, %0, %1, vcc");
            if (MODE == 13) LVK_ASM8("v_bfe_u32 %0, %0, 8, 8");
            if (MODE == 14) LVK_ASM8("v_mul_f32 %0, %0, %1");
            if (MODE == 15) LVK_ASM8("v_and_or_b32 %0, %0, %1, %2");
            if (MODE == 16) LVK_ASM8("v_rcp_f32 %0, %0");
            if (MODE == 17) LVK_ASM8("v_min_f32 %0, %0, %1");
            if (MODE == 18) LVK_ASM8("v_add_f32 %0, %0, %1");
            if (MODE == 19) LVK_ASM8("v_sub_f32 %0, %1, %0");
            if (MODE == 20) LVK_ASM8("v_fma_f32 %0, %0, %1, %2");
            if (MODE == 21) LVK_ASM8("v_floor_f32 %0, %0");
            if (MODE == 22) LVK_ASM8("v_lshrrev_b32 %0, 8, %0");
            if (MODE == 23) LVK_ASM8("v_cvt_f32_i32 %0, %0");
            if (MODE == 24) LVK_ASM8("v_med3_f32 %0, %0, %1, %2");
            if (MODE == 25) LVK_ASM8("v_mad_u32_u24 %0, %0, %1, %2");
            if (MODE == 26) LVK_ASM8("v_alignbit_b32 %0, %0, %1, 8");
            // two instruction classes interleaved 1:1 on independent registers (X on a0..a3, Y on a4..a7)
#define LVK_MIX(X, Y) do { asm volatile(X : "+v"(a0) : "v"(b), "v"(c)); asm volatile(Y : "+v"(a4) : "v"(b), "v"(c)); asm volatile(X : "+v"(a1) : "v"(b), "v"(c)); \
                           asm volatile(Y : "+v"(a5) : "v"(b), "v"(c)); asm volatile(X : "+v"(a2) : "v"(b), "v"(c)); asm volatile(Y : "+v"(a6) : "v"(b), "v"(c)); \
                           asm volatile(X : "+v"(a3) : "v"(b), "v"(c)); asm volatile(Y : "+v"(a7) : "v"(b), "v"(c)); } while (0)
            // 3:1
#define LVK_MIX31(X, Y) do { asm volatile(X : "+v"(a0) : "v"(b), "v"(c)); asm volatile(X : "+v"(a1) : "v"(b), "v"(c)); asm volatile(X : "+v"(a2) : "v"(b), "v"(c)); \
                           asm volatile(Y : "+v"(a6) : "v"(b), "v"(c)); asm volatile(X : "+v"(a3) : "v"(b), "v"(c)); asm volatile(X : "+v"(a4) : "v"(b), "v"(c)); \
                           asm volatile(X : "+v"(a5) : "v"(b), "v"(c)); asm volatile(Y : "+v"(a7) : "v"(b), "v"(c)); } while (0)
            if (MODE == 30) LVK_MIX("v_fma_f32 %0, %0, %1, %2", "v_max_f32 %0, %0, %1");
            if (MODE == 31) LVK_MIX31("v_fma_f32 %0, %0, %1, %2", "v_max_f32 %0, %0, %1");
            if (MODE == 32) LVK_MIX("v_fma_f32 %0, %0, %1, %2", "v_cvt_f32_ubyte1 %0, %0");
            if (MODE == 33) LVK_MIX("v_max_f32 %0, %0, %1", "v_cvt_f32_ubyte1 %0, %0");
            if (MODE == 34) LVK_MIX("v_fma_f32 %0, %0, %1, %2", "v_perm_b32 %0, %0, %1, %2");
            if (MODE == 35) LVK_ASM8("v_max_f32_e64 %0, %0, %1");
            if (MODE == 36) LVK_ASM8("v_and_b32 %0, %0, %1");
            if (MODE == 37) LVK_ASM8("v_or_b32 %0, %0, %1");
            if (MODE == 38) LVK_ASM8("v_sub_u32 %0, %0, %1");
            if (MODE == 39) LVK_ASM8("v_mul_u32_u24 %0, %0, %1");
            if (MODE == 40) LVK_ASM8("v_lshlrev_b32 %0, 8, %0");
            if (MODE == 41) LVK_ASM8("v_lshrrev_b32 %0, 1, %0");
            if (MODE == 42) LVK_ASM8("v_cndmask_b32 %0, %0, %1, s[10:11]");
            if (MODE == 43) LVK_ASM8("v_cmp_lt_f32 vcc, %0, %1");
            if (MODE == 44) LVK_ASM8("v_fmac_f32 %0, %1, %2");
            if (MODE == 45) LVK_ASM8("v_mac_f32 %0, %1, %2");
            if (MODE == 46) LVK_ASM8("v_add3_u32 %0, %0, %1, %2");
            if (MODE == 47) LVK_ASM8("v_lshl_add_u32 %0, %0, 1, %1");
            if (MODE == 48) LVK_ASM8("v_mul_f32 %0, 0x3b808081, %0");
            if (MODE == 49) LVK_ASM8("v_ashrrev_i32 %0, 1, %0");
            if (MODE == 50) LVK_ASM8("v_xor_b32 %0, %0, %1");
            if (MODE == 51) LVK_ASM8("v_mad_f32 %0, %0, %1, %2");
            if (MODE == 52) LVK_ASM8("v_fma_f32 %0, -%0, %1, |%2|");
            if (MODE == 53) LVK_ASM8("v_add_f32 %0, -%0, %1");
            if (MODE == 54) LVK_ASM8("v_mul_f32 %0, %0, %1 clamp");
            if (MODE == 55) LVK_ASM8("v_add_f32_e64 %0, |%0|, %1");
            // round 6: SDWA byte selects on a fast opcode (the byte -> 2^23 + byte trick of the remap's unpack) and their mixes
            if (MODE == 56) LVK_ASM8("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD");
            if (MODE == 57) LVK_MIX("v_fma_f32 %0, %0, %1, %2", "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD");
            if (MODE == 58) LVK_MIX("v_mul_f32 %0, %0, %1", "v_cvt_f32_ubyte1 %0, %0");
            if (MODE == 59) LVK_MIX31("v_fma_f32 %0, %0, %1, %2", "v_cvt_f32_ubyte1 %0, %0");
            if (MODE == 60) LVK_MIX31("v_fma_f32 %0, %0, %1, %2", "v_min_f32 %0, %0, %1");
            if (MODE == 61) LVK_ASM8("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD");
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE> void run(const char* name, double ops_per_iter_per_lane, double flops_per_op)
{
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 500, blocks = 4096;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f); hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = (double)blocks * 256 * iters * ops_per_iter_per_lane;
    printf("%-28s %8.3f ms  %7.2f T lane-instr/s  %7.2f TFLOP/s  (cycles per wave64 instr per SIMD @2.4GHz: %.2f)\n", name, ms,
           lane_ops / ms / 1e9, lane_ops * flops_per_op / ms / 1e9, 2.4e9 * 1024.0 * 64.0 / (lane_ops / (ms * 1e-3)));
    hipFree(d);
}

int main()
{
    run<0>("v_fma_f32 (8 indep chains)", 16 * 8, 2);
    run<1>("v_pk_fma_f32 (4 indep x2)", 16 * 8, 4);
    run<2>("mul+add+min mix", 16 * 8 * 3, 1);
    run<3>("v_cvt_i32_f32", 16 * 8, 1);
    run<4>("v_cvt_f32_ubyte1", 16 * 8, 1);
    run<5>("v_lshl_or_b32", 16 * 8, 1);
    run<6>("v_perm_b32", 16 * 8, 1);
    run<7>("v_max3_f32", 16 * 8, 1);
    run<8>("v_max_f32", 16 * 8, 1);
    run<9>("v_mov_b32", 16 * 8, 1);
    run<10>("v_add_u32", 16 * 8, 1);
    run<11>("v_lshlrev_b32", 16 * 8, 1);
    run<12>("v_cndmask_b32", 16 * 8, 1);
    run<13>("v_bfe_u32", 16 * 8, 1);
    run<14>("v_mul_f32", 16 * 8, 1);
    run<15>("v_and_or_b32", 16 * 8, 1);
    run<16>("v_rcp_f32", 16 * 8, 1);
    run<17>("v_min_f32", 16 * 8, 1);
    run<18>("v_add_f32", 16 * 8, 1);
    run<19>("v_sub_f32", 16 * 8, 1);
    run<20>("v_fma_f32 (asm)", 16 * 8, 2);
    run<21>("v_floor_f32", 16 * 8, 1);
    run<22>("v_lshrrev_b32", 16 * 8, 1);
    run<23>("v_cvt_f32_i32", 16 * 8, 1);
    run<24>("v_med3_f32", 16 * 8, 1);
    run<25>("v_mad_u32_u24", 16 * 8, 1);
    run<26>("v_alignbit_b32", 16 * 8, 1);
    run<30>("fma:max 1:1", 16 * 8, 1);
    run<31>("fma:max 3:1", 16 * 8, 1);
    run<32>("fma:cvt_ubyte 1:1", 16 * 8, 1);
    run<33>("max:cvt_ubyte 1:1", 16 * 8, 1);
    run<34>("fma:perm 1:1", 16 * 8, 1);
    run<35>("v_max_f32_e64", 16 * 8, 1);
    run<36>("v_and_b32", 16 * 8, 1);
    run<37>("v_or_b32", 16 * 8, 1);
    run<38>("v_sub_u32", 16 * 8, 1);
    run<39>("v_mul_u32_u24", 16 * 8, 1);
    run<40>("v_lshlrev_b32 by 8", 16 * 8, 1);
    run<41>("v_lshrrev_b32 by 1", 16 * 8, 1);
    run<42>("v_cndmask_b32 sgpr mask", 16 * 8, 1);
    run<43>("v_cmp_lt_f32", 16 * 8, 1);
    run<44>("v_fmac_f32", 16 * 8, 2);
    run<46>("v_add3_u32", 16 * 8, 1);
    run<47>("v_lshl_add_u32", 16 * 8, 1);
    run<48>("v_mul_f32 literal", 16 * 8, 1);
    run<49>("v_ashrrev_i32", 16 * 8, 1);
    run<50>("v_xor_b32", 16 * 8, 1);
    run<52>("v_fma_f32 neg/abs mods", 16 * 8, 2);
    run<53>("v_add_f32 neg mod", 16 * 8, 1);
    run<54>("v_mul_f32 clamp", 16 * 8, 1);
    run<55>("v_add_f32_e64 abs", 16 * 8, 1);
    run<56>("v_or_b32_sdwa BYTE_1", 16 * 8, 1);
    run<57>("fma:or_sdwa 1:1", 16 * 8, 1);
    run<58>("mul:cvt_ubyte 1:1", 16 * 8, 1);
    run<59>("fma:cvt_ubyte 3:1", 16 * 8, 1);
    run<60>("fma:min 3:1", 16 * 8, 1);
    run<61>("v_add_u32_sdwa BYTE_2", 16 * 8, 1);
    return 0;
}
