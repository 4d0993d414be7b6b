// Probe (round 6): VGPR bank / operand-port conflicts of gfx950's VALU.  scripts/probes/valu_pairing.hip showed that the cadence of a stream of
// independent v_fma_f32 depends on WHICH registers its three sources are (2.3 ... 4.2 cycles per instruction at 8 waves per SIMD).  Here the
// registers are explicit: 8 independent accumulators A_i (dst = src0), multiplier B (src1), addend C (src2).
// Build: hipcc --offload-arch=gfx950 -O3 -w -o scripts/probes/vgpr_bank_bin scripts/probes/vgpr_bank.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39"

// one unrolled body of 8 instructions, repeated 16 x per loop trip
#define BODY(I0, I1, I2, I3, I4, I5, I6, I7) ".rept 16\n" I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n.endr\n"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    // every register the bodies touch gets a finite value
    asm volatile(".irp r,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31\n v_mov_b32 v\\r, 1.0\n.endr\n"
                 "v_mov_b32 v32, 0.5\n v_mov_b32 v33, 0.5\n v_mov_b32 v34, 0.5\n v_mov_b32 v35, 0.5\n v_mov_b32 v36, 0.5\n v_mov_b32 v37, 0.5\n v_mov_b32 v38, 0.5\n v_mov_b32 v39, 0.5\n" ::: CLOB);
    for (int i = 0; i < iters; i++)
    {
        // accumulators in bank 0 (v0, v4, ... v28) unless stated
#define F(A, B, C) "v_fma_f32 v" #A ", v" #A ", v" #B ", v" #C
#define FS(A, B, C) "v_fma_f32 v" #A ", v" #B ", v" #A ", v" #C          /* accumulator as src1 */
#define M(A, B) "v_mul_f32 v" #A ", v" #A ", v" #B
#define MS(A, B) "v_mul_f32 v" #A ", v" #B ", v" #A
#define FMAC(A, B, C) "v_fmac_f32 v" #A ", v" #B ", v" #C
        if (MODE == 0) asm volatile(BODY(F(0,33,34), F(4,33,34), F(8,33,34), F(12,33,34), F(16,33,34), F(20,33,34), F(24,33,34), F(28,33,34)) ::: CLOB);   // banks 0,1,2
        if (MODE == 1) asm volatile(BODY(F(0,32,34), F(4,32,34), F(8,32,34), F(12,32,34), F(16,32,34), F(20,32,34), F(24,32,34), F(28,32,34)) ::: CLOB);   // src0 = src1 bank
        if (MODE == 2) asm volatile(BODY(F(0,33,36), F(4,33,36), F(8,33,36), F(12,33,36), F(16,33,36), F(20,33,36), F(24,33,36), F(28,33,36)) ::: CLOB);   // src0 = src2 bank
        if (MODE == 3) asm volatile(BODY(F(0,33,37), F(4,33,37), F(8,33,37), F(12,33,37), F(16,33,37), F(20,33,37), F(24,33,37), F(28,33,37)) ::: CLOB);   // src1 = src2 bank
        if (MODE == 4) asm volatile(BODY(F(0,32,36), F(4,32,36), F(8,32,36), F(12,32,36), F(16,32,36), F(20,32,36), F(24,32,36), F(28,32,36)) ::: CLOB);   // all bank 0
        if (MODE == 5) asm volatile(BODY(F(0,34,33), F(4,34,33), F(8,34,33), F(12,34,33), F(16,34,33), F(20,34,33), F(24,34,33), F(28,34,33)) ::: CLOB);   // banks 0,2,1
        if (MODE == 6) asm volatile(BODY(F(0,33,35), F(4,33,35), F(8,33,35), F(12,33,35), F(16,33,35), F(20,33,35), F(24,33,35), F(28,33,35)) ::: CLOB);   // banks 0,1,3 (src1, src2 same parity)
        if (MODE == 7) asm volatile(BODY(F(0,33,33), F(4,33,33), F(8,33,33), F(12,33,33), F(16,33,33), F(20,33,33), F(24,33,33), F(28,33,33)) ::: CLOB);   // src1 = src2 same register
        if (MODE == 8) asm volatile(BODY(F(0,33,34), F(1,33,34), F(2,33,34), F(3,33,34), F(4,33,34), F(5,33,34), F(6,33,34), F(7,33,34)) ::: CLOB);        // accumulators in consecutive registers
        if (MODE == 9) asm volatile(BODY(M(0,33), M(4,33), M(8,33), M(12,33), M(16,33), M(20,33), M(24,33), M(28,33)) ::: CLOB);                            // VOP2, banks 0,1
        if (MODE == 10) asm volatile(BODY(M(0,32), M(4,32), M(8,32), M(12,32), M(16,32), M(20,32), M(24,32), M(28,32)) ::: CLOB);                           // VOP2, same bank
        if (MODE == 11) asm volatile(BODY(FMAC(0,33,34), FMAC(4,33,34), FMAC(8,33,34), FMAC(12,33,34), FMAC(16,33,34), FMAC(20,33,34), FMAC(24,33,34), FMAC(28,33,34)) ::: CLOB);  // fmac, banks (dst 0) 1,2
        if (MODE == 12) asm volatile(BODY(FMAC(0,33,37), FMAC(4,33,37), FMAC(8,33,37), FMAC(12,33,37), FMAC(16,33,37), FMAC(20,33,37), FMAC(24,33,37), FMAC(28,33,37)) ::: CLOB);  // fmac, src0 = src1 bank
        if (MODE == 13) asm volatile(BODY(FMAC(0,32,34), FMAC(4,32,34), FMAC(8,32,34), FMAC(12,32,34), FMAC(16,32,34), FMAC(20,32,34), FMAC(24,32,34), FMAC(28,32,34)) ::: CLOB);  // fmac, src0 = dst bank
        if (MODE == 14) asm volatile(BODY(FS(0,33,34), FS(4,33,34), FS(8,33,34), FS(12,33,34), FS(16,33,34), FS(20,33,34), FS(24,33,34), FS(28,33,34)) ::: CLOB);                  // accumulator as src1: banks 1,0,2
        if (MODE == 15) asm volatile(BODY("v_fma_f32 v0, v0, s8, v34", "v_fma_f32 v4, v4, s8, v34", "v_fma_f32 v8, v8, s8, v34", "v_fma_f32 v12, v12, s8, v34",
                                          "v_fma_f32 v16, v16, s8, v34", "v_fma_f32 v20, v20, s8, v34", "v_fma_f32 v24, v24, s8, v34", "v_fma_f32 v28, v28, s8, v34") ::: CLOB, "s8");  // scalar multiplier
        if (MODE == 16) asm volatile(BODY("v_fma_f32 v0, v0, s8, v36", "v_fma_f32 v4, v4, s8, v36", "v_fma_f32 v8, v8, s8, v36", "v_fma_f32 v12, v12, s8, v36",
                                          "v_fma_f32 v16, v16, s8, v36", "v_fma_f32 v20, v20, s8, v36", "v_fma_f32 v24, v24, s8, v36", "v_fma_f32 v28, v28, s8, v36") ::: CLOB, "s8");  // scalar multiplier, src0 = src2 bank
        // result written to a different bank than it is read from / accumulators spread over the banks
        if (MODE == 17) asm volatile(BODY(F(0,34,35), F(1,34,35), F(4,34,35), F(5,34,35), F(8,34,35), F(9,34,35), F(12,34,35), F(13,34,35)) ::: CLOB);     // acc banks 0,1 ; B 2, C 3
        if (MODE == 18) asm volatile(BODY(F(0,33,34), F(4,33,34), F(8,33,34), F(12,33,34), F(0,33,34), F(4,33,34), F(8,33,34), F(12,33,34)) ::: CLOB);     // 4 chains, banks 0,1,2
        if (MODE == 19) asm volatile(BODY(F(0,33,34), F(4,33,34), F(0,33,34), F(4,33,34), F(0,33,34), F(4,33,34), F(0,33,34), F(4,33,34)) ::: CLOB);       // 2 chains, banks 0,1,2
        if (MODE == 20) asm volatile(BODY(F(0,33,34), F(0,33,34), F(0,33,34), F(0,33,34), F(0,33,34), F(0,33,34), F(0,33,34), F(0,33,34)) ::: CLOB);       // 1 chain, banks 0,1,2
#define G8(FMT0, FMT1, FMT2, FMT3, FMT4, FMT5, FMT6, FMT7) asm volatile(BODY(FMT0, FMT1, FMT2, FMT3, FMT4, FMT5, FMT6, FMT7) ::: CLOB, "s8", "s9")
        // VOP2 with a scalar / literal / inline-constant operand; fmamk / fmaak; VOP3-encoded mul with clamp; min / max / cvt with equal banks
        if (MODE == 21) G8("v_mul_f32 v0, s8, v0", "v_mul_f32 v4, s8, v4", "v_mul_f32 v8, s8, v8", "v_mul_f32 v12, s8, v12", "v_mul_f32 v16, s8, v16", "v_mul_f32 v20, s8, v20", "v_mul_f32 v24, s8, v24", "v_mul_f32 v28, s8, v28");
        if (MODE == 22) G8("v_add_f32 v0, s8, v0", "v_add_f32 v4, s8, v4", "v_add_f32 v8, s8, v8", "v_add_f32 v12, s8, v12", "v_add_f32 v16, s8, v16", "v_add_f32 v20, s8, v20", "v_add_f32 v24, s8, v24", "v_add_f32 v28, s8, v28");
        if (MODE == 23) G8("v_fmac_f32 v0, s8, v33", "v_fmac_f32 v4, s8, v33", "v_fmac_f32 v8, s8, v33", "v_fmac_f32 v12, s8, v33", "v_fmac_f32 v16, s8, v33", "v_fmac_f32 v20, s8, v33", "v_fmac_f32 v24, s8, v33", "v_fmac_f32 v28, s8, v33");
        if (MODE == 24) G8("v_fmamk_f32 v0, v0, 0x3f800001, v34", "v_fmamk_f32 v4, v4, 0x3f800001, v34", "v_fmamk_f32 v8, v8, 0x3f800001, v34", "v_fmamk_f32 v12, v12, 0x3f800001, v34",
                           "v_fmamk_f32 v16, v16, 0x3f800001, v34", "v_fmamk_f32 v20, v20, 0x3f800001, v34", "v_fmamk_f32 v24, v24, 0x3f800001, v34", "v_fmamk_f32 v28, v28, 0x3f800001, v34");
        if (MODE == 25) G8("v_fmaak_f32 v0, v0, v33, 0x3f000001", "v_fmaak_f32 v4, v4, v33, 0x3f000001", "v_fmaak_f32 v8, v8, v33, 0x3f000001", "v_fmaak_f32 v12, v12, v33, 0x3f000001",
                           "v_fmaak_f32 v16, v16, v33, 0x3f000001", "v_fmaak_f32 v20, v20, v33, 0x3f000001", "v_fmaak_f32 v24, v24, v33, 0x3f000001", "v_fmaak_f32 v28, v28, v33, 0x3f000001");
        if (MODE == 26) G8("v_fma_f32 v0, v0, v33, 1.0", "v_fma_f32 v4, v4, v33, 1.0", "v_fma_f32 v8, v8, v33, 1.0", "v_fma_f32 v12, v12, v33, 1.0", "v_fma_f32 v16, v16, v33, 1.0", "v_fma_f32 v20, v20, v33, 1.0", "v_fma_f32 v24, v24, v33, 1.0", "v_fma_f32 v28, v28, v33, 1.0");
        if (MODE == 27) G8("v_fma_f32 v0, v0, 0.5, v34", "v_fma_f32 v4, v4, 0.5, v34", "v_fma_f32 v8, v8, 0.5, v34", "v_fma_f32 v12, v12, 0.5, v34", "v_fma_f32 v16, v16, 0.5, v34", "v_fma_f32 v20, v20, 0.5, v34", "v_fma_f32 v24, v24, 0.5, v34", "v_fma_f32 v28, v28, 0.5, v34");
        if (MODE == 28) G8("v_mul_f32_e64 v0, v0, v33 clamp", "v_mul_f32_e64 v4, v4, v33 clamp", "v_mul_f32_e64 v8, v8, v33 clamp", "v_mul_f32_e64 v12, v12, v33 clamp", "v_mul_f32_e64 v16, v16, v33 clamp", "v_mul_f32_e64 v20, v20, v33 clamp", "v_mul_f32_e64 v24, v24, v33 clamp", "v_mul_f32_e64 v28, v28, v33 clamp");
        if (MODE == 29) G8("v_mul_f32_e64 v0, v0, v32 clamp", "v_mul_f32_e64 v4, v4, v32 clamp", "v_mul_f32_e64 v8, v8, v32 clamp", "v_mul_f32_e64 v12, v12, v32 clamp", "v_mul_f32_e64 v16, v16, v32 clamp", "v_mul_f32_e64 v20, v20, v32 clamp", "v_mul_f32_e64 v24, v24, v32 clamp", "v_mul_f32_e64 v28, v28, v32 clamp");
        if (MODE == 30) G8("v_fma_f32 v0, s8, v0, v34", "v_fma_f32 v4, s8, v4, v34", "v_fma_f32 v8, s8, v8, v34", "v_fma_f32 v12, s8, v12, v34", "v_fma_f32 v16, s8, v16, v34", "v_fma_f32 v20, s8, v20, v34", "v_fma_f32 v24, s8, v24, v34", "v_fma_f32 v28, s8, v28, v34");
        if (MODE == 31) G8("v_fma_f32 v0, v0, v33, s8", "v_fma_f32 v4, v4, v33, s8", "v_fma_f32 v8, v8, v33, s8", "v_fma_f32 v12, v12, v33, s8", "v_fma_f32 v16, v16, v33, s8", "v_fma_f32 v20, v20, v33, s8", "v_fma_f32 v24, v24, v33, s8", "v_fma_f32 v28, v28, v33, s8");
        // mixed stream: half of the instructions of the slow (same-bank) form, half of the fast one
        if (MODE == 32) G8(F(0,32,34), F(4,33,34), F(8,32,34), F(12,33,34), F(16,32,34), F(20,33,34), F(24,32,34), F(28,33,34));
        if (MODE == 33) G8(F(0,32,34), F(4,33,34), F(8,33,34), F(12,33,34), F(16,32,34), F(20,33,34), F(24,33,34), F(28,33,34));
        // the write side: results go to bank 1 while sources come from banks 0, 2, 3 (dst != src0)
        if (MODE == 34) G8("v_fma_f32 v1, v0, v34, v35", "v_fma_f32 v5, v4, v34, v35", "v_fma_f32 v9, v8, v34, v35", "v_fma_f32 v13, v12, v34, v35", "v_fma_f32 v17, v16, v34, v35", "v_fma_f32 v21, v20, v34, v35", "v_fma_f32 v25, v24, v34, v35", "v_fma_f32 v29, v28, v34, v35");
        if (MODE == 35) G8("v_sub_f32 v0, v0, v33", "v_sub_f32 v4, v4, v33", "v_sub_f32 v8, v8, v33", "v_sub_f32 v12, v12, v33", "v_sub_f32 v16, v16, v33", "v_sub_f32 v20, v20, v33", "v_sub_f32 v24, v24, v33", "v_sub_f32 v28, v28, v33");
        if (MODE == 36) G8("v_sub_f32 v0, v0, v32", "v_sub_f32 v4, v4, v32", "v_sub_f32 v8, v8, v32", "v_sub_f32 v12, v12, v32", "v_sub_f32 v16, v16, v32", "v_sub_f32 v20, v20, v32", "v_sub_f32 v24, v24, v32", "v_sub_f32 v28, v28, v32");
        // one instruction of a half-rate form among full-rate neighbours: what does it cost the stream?
        if (MODE == 37) G8("v_fma_f32 v0, v0, s8, v34", F(4,33,34), F(8,33,34), F(12,33,34), "v_fma_f32 v16, v16, s8, v34", F(20,33,34), F(24,33,34), F(28,33,34));
        if (MODE == 38) G8("v_fma_f32 v0, v0, s8, v34", F(4,33,34), F(8,33,34), F(12,33,34), F(16,33,34), F(20,33,34), F(24,33,34), F(28,33,34));
        if (MODE == 39) G8("v_mul_f32 v0, s8, v0", F(4,33,34), F(8,33,34), F(12,33,34), F(16,33,34), F(20,33,34), F(24,33,34), F(28,33,34));
        if (MODE == 40) G8(F(0,32,34), F(4,33,34), F(8,33,34), F(12,33,34), F(16,33,34), F(20,33,34), F(24,33,34), F(28,33,34));
        if (MODE == 41) G8("v_max_f32 v0, v0, v33", F(4,33,34), F(8,33,34), F(12,33,34), F(16,33,34), F(20,33,34), F(24,33,34), F(28,33,34));
        if (MODE == 42) G8("v_cvt_f32_ubyte1 v0, v0", F(4,33,34), F(8,33,34), F(12,33,34), "v_max_f32 v16, v16, v33", F(20,33,34), F(24,33,34), F(28,33,34));
        if (MODE == 43) G8("v_mov_b32 v0, s8", F(4,33,34), F(8,33,34), F(12,33,34), F(16,33,34), F(20,33,34), F(24,33,34), F(28,33,34));
        if (MODE == 44) G8("v_cmp_lt_f32 vcc, v0, v33", F(4,33,34), F(8,33,34), F(12,33,34), F(16,33,34), F(20,33,34), F(24,33,34), F(28,33,34));
        if (MODE == 45) G8("v_cmp_lt_f32_e64 s[10:11], v0, v33", F(4,33,34), F(8,33,34), F(12,33,34), F(16,33,34), F(20,33,34), F(24,33,34), F(28,33,34));
        if (MODE == 46) G8("s_mov_b32 s9, s8", F(4,33,34), F(8,33,34), F(12,33,34), F(16,33,34), F(20,33,34), F(24,33,34), F(28,33,34));
        if (MODE == 47) G8("s_nop 0", F(4,33,34), F(8,33,34), F(12,33,34), "s_nop 0", F(20,33,34), F(24,33,34), F(28,33,34));
        if (MODE == 48) G8("v_rcp_f32 v0, v0", F(4,33,34), F(8,33,34), F(12,33,34), F(16,33,34), F(20,33,34), F(24,33,34), F(28,33,34));
    }
    float r;
    asm volatile("v_add_f32 %0, v0, v4\n v_add_f32 %0, %0, v8\n v_add_f32 %0, %0, v1\n v_add_f32 %0, %0, v12" : "=v"(r) :: CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE> void run(const char* name)
{
    float* d; (void)hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-52s", name);
    for (int wps : {2, 4, 8})
    {
        const int iters = 400, blocks = 256 * wps * 4;
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), wps == 8 ? 0 : 160 * 1024 / wps - 256, 0, d, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), wps == 8 ? 0 : 160 * 1024 / wps - 256, 0, d, iters); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  w%d: %5.2f", wps, ms * 1e-3 / (4.0 * wps * iters * 16 * 8) * 2.4e9);
    }
    printf("\n");
    (void)hipFree(d);
}

extern __shared__ float s_pad[];
int main()
{
    (void)hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define RUN(M, NAME) (void)hipFuncSetAttribute((const void*)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); run<M>(NAME)
    printf("cycles @2.4 GHz per wave64 instruction per SIMD; 8 independent accumulators in bank 0 (v0, v4, ...) unless stated\n");
    RUN(0, "fma A(b0) B(b1) C(b2)");
    RUN(1, "fma A(b0) B(b0) C(b2)");
    RUN(2, "fma A(b0) B(b1) C(b0)");
    RUN(3, "fma A(b0) B(b1) C(b1)");
    RUN(4, "fma A(b0) B(b0) C(b0)");
    RUN(5, "fma A(b0) B(b2) C(b1)");
    RUN(6, "fma A(b0) B(b1) C(b3)");
    RUN(7, "fma A(b0) B = C same register");
    RUN(8, "fma A = v0..v7 B(b1) C(b2)");
    RUN(9, "mul A(b0) B(b1)");
    RUN(10, "mul A(b0) B(b0)");
    RUN(11, "fmac D(b0) += v33(b1) * v34(b2)");
    RUN(12, "fmac D(b0) += v33(b1) * v37(b1)");
    RUN(13, "fmac D(b0) += v32(b0) * v34(b2)");
    RUN(14, "fma acc as src1: B(b1) A(b0) C(b2)");
    RUN(15, "fma A(b0) s8 C(b2)");
    RUN(16, "fma A(b0) s8 C(b0)");
    RUN(17, "fma A in banks 0,1  B(b2) C(b3)");
    RUN(18, "fma 4 chains b0,b1,b2");
    RUN(19, "fma 2 chains b0,b1,b2");
    RUN(20, "fma 1 chain b0,b1,b2");
    RUN(21, "mul A(b0) = s8 * A");
    RUN(22, "add A(b0) = s8 + A");
    RUN(23, "fmac A(b0) += s8 * v33(b1)");
    RUN(24, "fmamk A = A * literal + v34(b2)");
    RUN(25, "fmaak A = A * v33(b1) + literal");
    RUN(26, "fma A = A * v33 + 1.0 (inline constant)");
    RUN(27, "fma A = A * 0.5 + v34 (inline constant)");
    RUN(28, "mul_e64 clamp A(b0) B(b1)");
    RUN(29, "mul_e64 clamp A(b0) B(b0)");
    RUN(30, "fma A = s8 * A + v34 (scalar as src0)");
    RUN(31, "fma A = A * v33 + s8 (scalar as src2)");
    RUN(32, "fma half same-bank src0/src1, half not (alternating)");
    RUN(33, "fma quarter same-bank src0/src1");
    RUN(34, "fma dst(b1) = A(b0) * v34(b2) + v35(b3)");
    RUN(35, "sub A(b0) - v33(b1)");
    RUN(36, "sub A(b0) - v32(b0)");
    RUN(37, "1 of 4: fma with scalar src1, rest plain fma");
    RUN(38, "1 of 8: fma with scalar src1");
    RUN(39, "1 of 8: mul with scalar src0 (VOP2)");
    RUN(40, "1 of 8: fma src0/src1 same bank");
    RUN(41, "1 of 8: max (4-cycle class)");
    RUN(42, "2 of 8: cvt_ubyte, max (4-cycle class)");
    RUN(43, "1 of 8: v_mov from scalar");
    RUN(44, "1 of 8: v_cmp -> vcc");
    RUN(45, "1 of 8: v_cmp_e64 -> sgpr pair");
    RUN(46, "7 fma + s_mov");
    RUN(47, "6 fma + 2 s_nop");
    RUN(48, "1 of 8: v_rcp_f32");
    return 0;
}
