// GENERATED probe (round 6): throughput (8 independent chains, 8 waves per SIMD) and latency (one dependent chain, 1 wave per SIMD) of every VALU
// opcode form the remap kernels contain.  cycles at a nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","s8","s9","s10","s11","vcc"
template <int T, int DEP> __global__ __launch_bounds__(256) void k(float* out, int iters)
{
    asm volatile(".irp r,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,35,36,37,38,39\n v_mov_b32 v\\r, 1.5\n.endr\n s_mov_b32 s8, 0x3fc00000\n s_mov_b64 s[10:11], -1\n v_mov_b32 v35, 1\n v_mov_b32 v36, 1\n v_mov_b32 v37, 0\n" ::: CLOB);
    for (int i = 0; i < iters; i++)
    {
        if (T == 0 && DEP == 0) asm volatile(".rept 16\n v_fma_f32 v0, v0, v33, v34\n v_fma_f32 v2, v2, v33, v34\n v_fma_f32 v4, v4, v33, v34\n v_fma_f32 v6, v6, v33, v34\n v_fma_f32 v8, v8, v33, v34\n v_fma_f32 v10, v10, v33, v34\n v_fma_f32 v12, v12, v33, v34\n v_fma_f32 v14, v14, v33, v34\n.endr\n" ::: CLOB);
        if (T == 0 && DEP == 1) asm volatile(".rept 16\n v_fma_f32 v0, v0, v33, v34\n v_fma_f32 v0, v0, v33, v34\n v_fma_f32 v0, v0, v33, v34\n v_fma_f32 v0, v0, v33, v34\n v_fma_f32 v0, v0, v33, v34\n v_fma_f32 v0, v0, v33, v34\n v_fma_f32 v0, v0, v33, v34\n v_fma_f32 v0, v0, v33, v34\n.endr\n" ::: CLOB);
        if (T == 1 && DEP == 0) asm volatile(".rept 16\n v_mul_f32 v0, v0, v33\n v_mul_f32 v2, v2, v33\n v_mul_f32 v4, v4, v33\n v_mul_f32 v6, v6, v33\n v_mul_f32 v8, v8, v33\n v_mul_f32 v10, v10, v33\n v_mul_f32 v12, v12, v33\n v_mul_f32 v14, v14, v33\n.endr\n" ::: CLOB);
        if (T == 1 && DEP == 1) asm volatile(".rept 16\n v_mul_f32 v0, v0, v33\n v_mul_f32 v0, v0, v33\n v_mul_f32 v0, v0, v33\n v_mul_f32 v0, v0, v33\n v_mul_f32 v0, v0, v33\n v_mul_f32 v0, v0, v33\n v_mul_f32 v0, v0, v33\n v_mul_f32 v0, v0, v33\n.endr\n" ::: CLOB);
        if (T == 2 && DEP == 0) asm volatile(".rept 16\n v_rcp_f32 v0, v0\n v_rcp_f32 v2, v2\n v_rcp_f32 v4, v4\n v_rcp_f32 v6, v6\n v_rcp_f32 v8, v8\n v_rcp_f32 v10, v10\n v_rcp_f32 v12, v12\n v_rcp_f32 v14, v14\n.endr\n" ::: CLOB);
        if (T == 2 && DEP == 1) asm volatile(".rept 16\n v_rcp_f32 v0, v0\n v_rcp_f32 v0, v0\n v_rcp_f32 v0, v0\n v_rcp_f32 v0, v0\n v_rcp_f32 v0, v0\n v_rcp_f32 v0, v0\n v_rcp_f32 v0, v0\n v_rcp_f32 v0, v0\n.endr\n" ::: CLOB);
        if (T == 3 && DEP == 0) asm volatile(".rept 16\n v_fract_f32 v0, v0\n v_fract_f32 v2, v2\n v_fract_f32 v4, v4\n v_fract_f32 v6, v6\n v_fract_f32 v8, v8\n v_fract_f32 v10, v10\n v_fract_f32 v12, v12\n v_fract_f32 v14, v14\n.endr\n" ::: CLOB);
        if (T == 3 && DEP == 1) asm volatile(".rept 16\n v_fract_f32 v0, v0\n v_fract_f32 v0, v0\n v_fract_f32 v0, v0\n v_fract_f32 v0, v0\n v_fract_f32 v0, v0\n v_fract_f32 v0, v0\n v_fract_f32 v0, v0\n v_fract_f32 v0, v0\n.endr\n" ::: CLOB);
        if (T == 4 && DEP == 0) asm volatile(".rept 16\n v_frexp_mant_f32 v0, v0\n v_frexp_mant_f32 v2, v2\n v_frexp_mant_f32 v4, v4\n v_frexp_mant_f32 v6, v6\n v_frexp_mant_f32 v8, v8\n v_frexp_mant_f32 v10, v10\n v_frexp_mant_f32 v12, v12\n v_frexp_mant_f32 v14, v14\n.endr\n" ::: CLOB);
        if (T == 4 && DEP == 1) asm volatile(".rept 16\n v_frexp_mant_f32 v0, v0\n v_frexp_mant_f32 v0, v0\n v_frexp_mant_f32 v0, v0\n v_frexp_mant_f32 v0, v0\n v_frexp_mant_f32 v0, v0\n v_frexp_mant_f32 v0, v0\n v_frexp_mant_f32 v0, v0\n v_frexp_mant_f32 v0, v0\n.endr\n" ::: CLOB);
        if (T == 5 && DEP == 0) asm volatile(".rept 16\n v_frexp_exp_i32_f32 v0, v0\n v_frexp_exp_i32_f32 v2, v2\n v_frexp_exp_i32_f32 v4, v4\n v_frexp_exp_i32_f32 v6, v6\n v_frexp_exp_i32_f32 v8, v8\n v_frexp_exp_i32_f32 v10, v10\n v_frexp_exp_i32_f32 v12, v12\n v_frexp_exp_i32_f32 v14, v14\n.endr\n" ::: CLOB);
        if (T == 5 && DEP == 1) asm volatile(".rept 16\n v_frexp_exp_i32_f32 v0, v0\n v_frexp_exp_i32_f32 v0, v0\n v_frexp_exp_i32_f32 v0, v0\n v_frexp_exp_i32_f32 v0, v0\n v_frexp_exp_i32_f32 v0, v0\n v_frexp_exp_i32_f32 v0, v0\n v_frexp_exp_i32_f32 v0, v0\n v_frexp_exp_i32_f32 v0, v0\n.endr\n" ::: CLOB);
        if (T == 6 && DEP == 0) asm volatile(".rept 16\n v_ldexp_f32 v0, v0, v35\n v_ldexp_f32 v2, v2, v35\n v_ldexp_f32 v4, v4, v35\n v_ldexp_f32 v6, v6, v35\n v_ldexp_f32 v8, v8, v35\n v_ldexp_f32 v10, v10, v35\n v_ldexp_f32 v12, v12, v35\n v_ldexp_f32 v14, v14, v35\n.endr\n" ::: CLOB);
        if (T == 6 && DEP == 1) asm volatile(".rept 16\n v_ldexp_f32 v0, v0, v35\n v_ldexp_f32 v0, v0, v35\n v_ldexp_f32 v0, v0, v35\n v_ldexp_f32 v0, v0, v35\n v_ldexp_f32 v0, v0, v35\n v_ldexp_f32 v0, v0, v35\n v_ldexp_f32 v0, v0, v35\n v_ldexp_f32 v0, v0, v35\n.endr\n" ::: CLOB);
        if (T == 7 && DEP == 0) asm volatile(".rept 16\n v_cvt_i32_f32 v0, v0\n v_cvt_i32_f32 v2, v2\n v_cvt_i32_f32 v4, v4\n v_cvt_i32_f32 v6, v6\n v_cvt_i32_f32 v8, v8\n v_cvt_i32_f32 v10, v10\n v_cvt_i32_f32 v12, v12\n v_cvt_i32_f32 v14, v14\n.endr\n" ::: CLOB);
        if (T == 7 && DEP == 1) asm volatile(".rept 16\n v_cvt_i32_f32 v0, v0\n v_cvt_i32_f32 v0, v0\n v_cvt_i32_f32 v0, v0\n v_cvt_i32_f32 v0, v0\n v_cvt_i32_f32 v0, v0\n v_cvt_i32_f32 v0, v0\n v_cvt_i32_f32 v0, v0\n v_cvt_i32_f32 v0, v0\n.endr\n" ::: CLOB);
        if (T == 8 && DEP == 0) asm volatile(".rept 16\n v_cvt_f32_i32 v0, v0\n v_cvt_f32_i32 v2, v2\n v_cvt_f32_i32 v4, v4\n v_cvt_f32_i32 v6, v6\n v_cvt_f32_i32 v8, v8\n v_cvt_f32_i32 v10, v10\n v_cvt_f32_i32 v12, v12\n v_cvt_f32_i32 v14, v14\n.endr\n" ::: CLOB);
        if (T == 8 && DEP == 1) asm volatile(".rept 16\n v_cvt_f32_i32 v0, v0\n v_cvt_f32_i32 v0, v0\n v_cvt_f32_i32 v0, v0\n v_cvt_f32_i32 v0, v0\n v_cvt_f32_i32 v0, v0\n v_cvt_f32_i32 v0, v0\n v_cvt_f32_i32 v0, v0\n v_cvt_f32_i32 v0, v0\n.endr\n" ::: CLOB);
        if (T == 9 && DEP == 0) asm volatile(".rept 16\n v_cvt_f32_ubyte1 v0, v0\n v_cvt_f32_ubyte1 v2, v2\n v_cvt_f32_ubyte1 v4, v4\n v_cvt_f32_ubyte1 v6, v6\n v_cvt_f32_ubyte1 v8, v8\n v_cvt_f32_ubyte1 v10, v10\n v_cvt_f32_ubyte1 v12, v12\n v_cvt_f32_ubyte1 v14, v14\n.endr\n" ::: CLOB);
        if (T == 9 && DEP == 1) asm volatile(".rept 16\n v_cvt_f32_ubyte1 v0, v0\n v_cvt_f32_ubyte1 v0, v0\n v_cvt_f32_ubyte1 v0, v0\n v_cvt_f32_ubyte1 v0, v0\n v_cvt_f32_ubyte1 v0, v0\n v_cvt_f32_ubyte1 v0, v0\n v_cvt_f32_ubyte1 v0, v0\n v_cvt_f32_ubyte1 v0, v0\n.endr\n" ::: CLOB);
        if (T == 10 && DEP == 0) asm volatile(".rept 16\n v_min_f32 v0, v0, v33\n v_min_f32 v2, v2, v33\n v_min_f32 v4, v4, v33\n v_min_f32 v6, v6, v33\n v_min_f32 v8, v8, v33\n v_min_f32 v10, v10, v33\n v_min_f32 v12, v12, v33\n v_min_f32 v14, v14, v33\n.endr\n" ::: CLOB);
        if (T == 10 && DEP == 1) asm volatile(".rept 16\n v_min_f32 v0, v0, v33\n v_min_f32 v0, v0, v33\n v_min_f32 v0, v0, v33\n v_min_f32 v0, v0, v33\n v_min_f32 v0, v0, v33\n v_min_f32 v0, v0, v33\n v_min_f32 v0, v0, v33\n v_min_f32 v0, v0, v33\n.endr\n" ::: CLOB);
        if (T == 11 && DEP == 0) asm volatile(".rept 16\n v_med3_f32 v0, v0, v33, v34\n v_med3_f32 v2, v2, v33, v34\n v_med3_f32 v4, v4, v33, v34\n v_med3_f32 v6, v6, v33, v34\n v_med3_f32 v8, v8, v33, v34\n v_med3_f32 v10, v10, v33, v34\n v_med3_f32 v12, v12, v33, v34\n v_med3_f32 v14, v14, v33, v34\n.endr\n" ::: CLOB);
        if (T == 11 && DEP == 1) asm volatile(".rept 16\n v_med3_f32 v0, v0, v33, v34\n v_med3_f32 v0, v0, v33, v34\n v_med3_f32 v0, v0, v33, v34\n v_med3_f32 v0, v0, v33, v34\n v_med3_f32 v0, v0, v33, v34\n v_med3_f32 v0, v0, v33, v34\n v_med3_f32 v0, v0, v33, v34\n v_med3_f32 v0, v0, v33, v34\n.endr\n" ::: CLOB);
        if (T == 12 && DEP == 0) asm volatile(".rept 16\n v_min3_f32 v0, v0, v33, v34\n v_min3_f32 v2, v2, v33, v34\n v_min3_f32 v4, v4, v33, v34\n v_min3_f32 v6, v6, v33, v34\n v_min3_f32 v8, v8, v33, v34\n v_min3_f32 v10, v10, v33, v34\n v_min3_f32 v12, v12, v33, v34\n v_min3_f32 v14, v14, v33, v34\n.endr\n" ::: CLOB);
        if (T == 12 && DEP == 1) asm volatile(".rept 16\n v_min3_f32 v0, v0, v33, v34\n v_min3_f32 v0, v0, v33, v34\n v_min3_f32 v0, v0, v33, v34\n v_min3_f32 v0, v0, v33, v34\n v_min3_f32 v0, v0, v33, v34\n v_min3_f32 v0, v0, v33, v34\n v_min3_f32 v0, v0, v33, v34\n v_min3_f32 v0, v0, v33, v34\n.endr\n" ::: CLOB);
        if (T == 13 && DEP == 0) asm volatile(".rept 16\n v_mad_u32_u24 v0, v0, v33, v34\n v_mad_u32_u24 v2, v2, v33, v34\n v_mad_u32_u24 v4, v4, v33, v34\n v_mad_u32_u24 v6, v6, v33, v34\n v_mad_u32_u24 v8, v8, v33, v34\n v_mad_u32_u24 v10, v10, v33, v34\n v_mad_u32_u24 v12, v12, v33, v34\n v_mad_u32_u24 v14, v14, v33, v34\n.endr\n" ::: CLOB);
        if (T == 13 && DEP == 1) asm volatile(".rept 16\n v_mad_u32_u24 v0, v0, v33, v34\n v_mad_u32_u24 v0, v0, v33, v34\n v_mad_u32_u24 v0, v0, v33, v34\n v_mad_u32_u24 v0, v0, v33, v34\n v_mad_u32_u24 v0, v0, v33, v34\n v_mad_u32_u24 v0, v0, v33, v34\n v_mad_u32_u24 v0, v0, v33, v34\n v_mad_u32_u24 v0, v0, v33, v34\n.endr\n" ::: CLOB);
        if (T == 14 && DEP == 0) asm volatile(".rept 16\n v_mul_u32_u24 v0, v0, v33\n v_mul_u32_u24 v2, v2, v33\n v_mul_u32_u24 v4, v4, v33\n v_mul_u32_u24 v6, v6, v33\n v_mul_u32_u24 v8, v8, v33\n v_mul_u32_u24 v10, v10, v33\n v_mul_u32_u24 v12, v12, v33\n v_mul_u32_u24 v14, v14, v33\n.endr\n" ::: CLOB);
        if (T == 14 && DEP == 1) asm volatile(".rept 16\n v_mul_u32_u24 v0, v0, v33\n v_mul_u32_u24 v0, v0, v33\n v_mul_u32_u24 v0, v0, v33\n v_mul_u32_u24 v0, v0, v33\n v_mul_u32_u24 v0, v0, v33\n v_mul_u32_u24 v0, v0, v33\n v_mul_u32_u24 v0, v0, v33\n v_mul_u32_u24 v0, v0, v33\n.endr\n" ::: CLOB);
        if (T == 15 && DEP == 0) asm volatile(".rept 16\n v_mul_lo_u32 v0, v0, v33\n v_mul_lo_u32 v2, v2, v33\n v_mul_lo_u32 v4, v4, v33\n v_mul_lo_u32 v6, v6, v33\n v_mul_lo_u32 v8, v8, v33\n v_mul_lo_u32 v10, v10, v33\n v_mul_lo_u32 v12, v12, v33\n v_mul_lo_u32 v14, v14, v33\n.endr\n" ::: CLOB);
        if (T == 15 && DEP == 1) asm volatile(".rept 16\n v_mul_lo_u32 v0, v0, v33\n v_mul_lo_u32 v0, v0, v33\n v_mul_lo_u32 v0, v0, v33\n v_mul_lo_u32 v0, v0, v33\n v_mul_lo_u32 v0, v0, v33\n v_mul_lo_u32 v0, v0, v33\n v_mul_lo_u32 v0, v0, v33\n v_mul_lo_u32 v0, v0, v33\n.endr\n" ::: CLOB);
        if (T == 16 && DEP == 0) asm volatile(".rept 16\n v_lshl_add_u64 v[0:1], v[0:1], 0, v[36:37]\n v_lshl_add_u64 v[2:3], v[2:3], 0, v[36:37]\n v_lshl_add_u64 v[4:5], v[4:5], 0, v[36:37]\n v_lshl_add_u64 v[6:7], v[6:7], 0, v[36:37]\n v_lshl_add_u64 v[8:9], v[8:9], 0, v[36:37]\n v_lshl_add_u64 v[10:11], v[10:11], 0, v[36:37]\n v_lshl_add_u64 v[12:13], v[12:13], 0, v[36:37]\n v_lshl_add_u64 v[14:15], v[14:15], 0, v[36:37]\n.endr\n" ::: CLOB);
        if (T == 16 && DEP == 1) asm volatile(".rept 16\n v_lshl_add_u64 v[0:1], v[0:1], 0, v[36:37]\n v_lshl_add_u64 v[0:1], v[0:1], 0, v[36:37]\n v_lshl_add_u64 v[0:1], v[0:1], 0, v[36:37]\n v_lshl_add_u64 v[0:1], v[0:1], 0, v[36:37]\n v_lshl_add_u64 v[0:1], v[0:1], 0, v[36:37]\n v_lshl_add_u64 v[0:1], v[0:1], 0, v[36:37]\n v_lshl_add_u64 v[0:1], v[0:1], 0, v[36:37]\n v_lshl_add_u64 v[0:1], v[0:1], 0, v[36:37]\n.endr\n" ::: CLOB);
        if (T == 17 && DEP == 0) asm volatile(".rept 16\n v_alignbit_b32 v0, v0, v33, 8\n v_alignbit_b32 v2, v2, v33, 8\n v_alignbit_b32 v4, v4, v33, 8\n v_alignbit_b32 v6, v6, v33, 8\n v_alignbit_b32 v8, v8, v33, 8\n v_alignbit_b32 v10, v10, v33, 8\n v_alignbit_b32 v12, v12, v33, 8\n v_alignbit_b32 v14, v14, v33, 8\n.endr\n" ::: CLOB);
        if (T == 17 && DEP == 1) asm volatile(".rept 16\n v_alignbit_b32 v0, v0, v33, 8\n v_alignbit_b32 v0, v0, v33, 8\n v_alignbit_b32 v0, v0, v33, 8\n v_alignbit_b32 v0, v0, v33, 8\n v_alignbit_b32 v0, v0, v33, 8\n v_alignbit_b32 v0, v0, v33, 8\n v_alignbit_b32 v0, v0, v33, 8\n v_alignbit_b32 v0, v0, v33, 8\n.endr\n" ::: CLOB);
        if (T == 18 && DEP == 0) asm volatile(".rept 16\n v_lshl_or_b32 v0, v0, 8, v33\n v_lshl_or_b32 v2, v2, 8, v33\n v_lshl_or_b32 v4, v4, 8, v33\n v_lshl_or_b32 v6, v6, 8, v33\n v_lshl_or_b32 v8, v8, 8, v33\n v_lshl_or_b32 v10, v10, 8, v33\n v_lshl_or_b32 v12, v12, 8, v33\n v_lshl_or_b32 v14, v14, 8, v33\n.endr\n" ::: CLOB);
        if (T == 18 && DEP == 1) asm volatile(".rept 16\n v_lshl_or_b32 v0, v0, 8, v33\n v_lshl_or_b32 v0, v0, 8, v33\n v_lshl_or_b32 v0, v0, 8, v33\n v_lshl_or_b32 v0, v0, 8, v33\n v_lshl_or_b32 v0, v0, 8, v33\n v_lshl_or_b32 v0, v0, 8, v33\n v_lshl_or_b32 v0, v0, 8, v33\n v_lshl_or_b32 v0, v0, 8, v33\n.endr\n" ::: CLOB);
        if (T == 19 && DEP == 0) asm volatile(".rept 16\n v_and_or_b32 v0, v0, v33, v34\n v_and_or_b32 v2, v2, v33, v34\n v_and_or_b32 v4, v4, v33, v34\n v_and_or_b32 v6, v6, v33, v34\n v_and_or_b32 v8, v8, v33, v34\n v_and_or_b32 v10, v10, v33, v34\n v_and_or_b32 v12, v12, v33, v34\n v_and_or_b32 v14, v14, v33, v34\n.endr\n" ::: CLOB);
        if (T == 19 && DEP == 1) asm volatile(".rept 16\n v_and_or_b32 v0, v0, v33, v34\n v_and_or_b32 v0, v0, v33, v34\n v_and_or_b32 v0, v0, v33, v34\n v_and_or_b32 v0, v0, v33, v34\n v_and_or_b32 v0, v0, v33, v34\n v_and_or_b32 v0, v0, v33, v34\n v_and_or_b32 v0, v0, v33, v34\n v_and_or_b32 v0, v0, v33, v34\n.endr\n" ::: CLOB);
        if (T == 20 && DEP == 0) asm volatile(".rept 16\n v_perm_b32 v0, v0, v33, v34\n v_perm_b32 v2, v2, v33, v34\n v_perm_b32 v4, v4, v33, v34\n v_perm_b32 v6, v6, v33, v34\n v_perm_b32 v8, v8, v33, v34\n v_perm_b32 v10, v10, v33, v34\n v_perm_b32 v12, v12, v33, v34\n v_perm_b32 v14, v14, v33, v34\n.endr\n" ::: CLOB);
        if (T == 20 && DEP == 1) asm volatile(".rept 16\n v_perm_b32 v0, v0, v33, v34\n v_perm_b32 v0, v0, v33, v34\n v_perm_b32 v0, v0, v33, v34\n v_perm_b32 v0, v0, v33, v34\n v_perm_b32 v0, v0, v33, v34\n v_perm_b32 v0, v0, v33, v34\n v_perm_b32 v0, v0, v33, v34\n v_perm_b32 v0, v0, v33, v34\n.endr\n" ::: CLOB);
        if (T == 21 && DEP == 0) asm volatile(".rept 16\n v_bitop3_b16 v0, v0, v33, v34 bitop3:0xec\n v_bitop3_b16 v2, v2, v33, v34 bitop3:0xec\n v_bitop3_b16 v4, v4, v33, v34 bitop3:0xec\n v_bitop3_b16 v6, v6, v33, v34 bitop3:0xec\n v_bitop3_b16 v8, v8, v33, v34 bitop3:0xec\n v_bitop3_b16 v10, v10, v33, v34 bitop3:0xec\n v_bitop3_b16 v12, v12, v33, v34 bitop3:0xec\n v_bitop3_b16 v14, v14, v33, v34 bitop3:0xec\n.endr\n" ::: CLOB);
        if (T == 21 && DEP == 1) asm volatile(".rept 16\n v_bitop3_b16 v0, v0, v33, v34 bitop3:0xec\n v_bitop3_b16 v0, v0, v33, v34 bitop3:0xec\n v_bitop3_b16 v0, v0, v33, v34 bitop3:0xec\n v_bitop3_b16 v0, v0, v33, v34 bitop3:0xec\n v_bitop3_b16 v0, v0, v33, v34 bitop3:0xec\n v_bitop3_b16 v0, v0, v33, v34 bitop3:0xec\n v_bitop3_b16 v0, v0, v33, v34 bitop3:0xec\n v_bitop3_b16 v0, v0, v33, v34 bitop3:0xec\n.endr\n" ::: CLOB);
        if (T == 22 && DEP == 0) asm volatile(".rept 16\n v_lshlrev_b32 v0, 8, v0\n v_lshlrev_b32 v2, 8, v2\n v_lshlrev_b32 v4, 8, v4\n v_lshlrev_b32 v6, 8, v6\n v_lshlrev_b32 v8, 8, v8\n v_lshlrev_b32 v10, 8, v10\n v_lshlrev_b32 v12, 8, v12\n v_lshlrev_b32 v14, 8, v14\n.endr\n" ::: CLOB);
        if (T == 22 && DEP == 1) asm volatile(".rept 16\n v_lshlrev_b32 v0, 8, v0\n v_lshlrev_b32 v0, 8, v0\n v_lshlrev_b32 v0, 8, v0\n v_lshlrev_b32 v0, 8, v0\n v_lshlrev_b32 v0, 8, v0\n v_lshlrev_b32 v0, 8, v0\n v_lshlrev_b32 v0, 8, v0\n v_lshlrev_b32 v0, 8, v0\n.endr\n" ::: CLOB);
        if (T == 23 && DEP == 0) asm volatile(".rept 16\n v_lshrrev_b32 v0, 8, v0\n v_lshrrev_b32 v2, 8, v2\n v_lshrrev_b32 v4, 8, v4\n v_lshrrev_b32 v6, 8, v6\n v_lshrrev_b32 v8, 8, v8\n v_lshrrev_b32 v10, 8, v10\n v_lshrrev_b32 v12, 8, v12\n v_lshrrev_b32 v14, 8, v14\n.endr\n" ::: CLOB);
        if (T == 23 && DEP == 1) asm volatile(".rept 16\n v_lshrrev_b32 v0, 8, v0\n v_lshrrev_b32 v0, 8, v0\n v_lshrrev_b32 v0, 8, v0\n v_lshrrev_b32 v0, 8, v0\n v_lshrrev_b32 v0, 8, v0\n v_lshrrev_b32 v0, 8, v0\n v_lshrrev_b32 v0, 8, v0\n v_lshrrev_b32 v0, 8, v0\n.endr\n" ::: CLOB);
        if (T == 24 && DEP == 0) asm volatile(".rept 16\n v_add_u32 v0, v0, v33\n v_add_u32 v2, v2, v33\n v_add_u32 v4, v4, v33\n v_add_u32 v6, v6, v33\n v_add_u32 v8, v8, v33\n v_add_u32 v10, v10, v33\n v_add_u32 v12, v12, v33\n v_add_u32 v14, v14, v33\n.endr\n" ::: CLOB);
        if (T == 24 && DEP == 1) asm volatile(".rept 16\n v_add_u32 v0, v0, v33\n v_add_u32 v0, v0, v33\n v_add_u32 v0, v0, v33\n v_add_u32 v0, v0, v33\n v_add_u32 v0, v0, v33\n v_add_u32 v0, v0, v33\n v_add_u32 v0, v0, v33\n v_add_u32 v0, v0, v33\n.endr\n" ::: CLOB);
        if (T == 25 && DEP == 0) asm volatile(".rept 16\n v_fma_f32 v0, s8, v0, v34\n v_fma_f32 v2, s8, v2, v34\n v_fma_f32 v4, s8, v4, v34\n v_fma_f32 v6, s8, v6, v34\n v_fma_f32 v8, s8, v8, v34\n v_fma_f32 v10, s8, v10, v34\n v_fma_f32 v12, s8, v12, v34\n v_fma_f32 v14, s8, v14, v34\n.endr\n" ::: CLOB);
        if (T == 25 && DEP == 1) asm volatile(".rept 16\n v_fma_f32 v0, s8, v0, v34\n v_fma_f32 v0, s8, v0, v34\n v_fma_f32 v0, s8, v0, v34\n v_fma_f32 v0, s8, v0, v34\n v_fma_f32 v0, s8, v0, v34\n v_fma_f32 v0, s8, v0, v34\n v_fma_f32 v0, s8, v0, v34\n v_fma_f32 v0, s8, v0, v34\n.endr\n" ::: CLOB);
        if (T == 26 && DEP == 0) asm volatile(".rept 16\n v_add_f32 v0, s8, v0\n v_add_f32 v2, s8, v2\n v_add_f32 v4, s8, v4\n v_add_f32 v6, s8, v6\n v_add_f32 v8, s8, v8\n v_add_f32 v10, s8, v10\n v_add_f32 v12, s8, v12\n v_add_f32 v14, s8, v14\n.endr\n" ::: CLOB);
        if (T == 26 && DEP == 1) asm volatile(".rept 16\n v_add_f32 v0, s8, v0\n v_add_f32 v0, s8, v0\n v_add_f32 v0, s8, v0\n v_add_f32 v0, s8, v0\n v_add_f32 v0, s8, v0\n v_add_f32 v0, s8, v0\n v_add_f32 v0, s8, v0\n v_add_f32 v0, s8, v0\n.endr\n" ::: CLOB);
        if (T == 27 && DEP == 0) asm volatile(".rept 16\n v_mov_b32 v0, s8\n v_mov_b32 v2, s8\n v_mov_b32 v4, s8\n v_mov_b32 v6, s8\n v_mov_b32 v8, s8\n v_mov_b32 v10, s8\n v_mov_b32 v12, s8\n v_mov_b32 v14, s8\n.endr\n" ::: CLOB);
        if (T == 27 && DEP == 1) asm volatile(".rept 16\n v_mov_b32 v0, s8\n v_mov_b32 v0, s8\n v_mov_b32 v0, s8\n v_mov_b32 v0, s8\n v_mov_b32 v0, s8\n v_mov_b32 v0, s8\n v_mov_b32 v0, s8\n v_mov_b32 v0, s8\n.endr\n" ::: CLOB);
        if (T == 28 && DEP == 0) asm volatile(".rept 16\n v_cmp_gt_i32_e64 s[10:11], s8, v0\n v_cmp_gt_i32_e64 s[10:11], s8, v2\n v_cmp_gt_i32_e64 s[10:11], s8, v4\n v_cmp_gt_i32_e64 s[10:11], s8, v6\n v_cmp_gt_i32_e64 s[10:11], s8, v8\n v_cmp_gt_i32_e64 s[10:11], s8, v10\n v_cmp_gt_i32_e64 s[10:11], s8, v12\n v_cmp_gt_i32_e64 s[10:11], s8, v14\n.endr\n" ::: CLOB);
        if (T == 28 && DEP == 1) asm volatile(".rept 16\n v_cmp_gt_i32_e64 s[10:11], s8, v0\n v_cmp_gt_i32_e64 s[10:11], s8, v0\n v_cmp_gt_i32_e64 s[10:11], s8, v0\n v_cmp_gt_i32_e64 s[10:11], s8, v0\n v_cmp_gt_i32_e64 s[10:11], s8, v0\n v_cmp_gt_i32_e64 s[10:11], s8, v0\n v_cmp_gt_i32_e64 s[10:11], s8, v0\n v_cmp_gt_i32_e64 s[10:11], s8, v0\n.endr\n" ::: CLOB);
        if (T == 29 && DEP == 0) asm volatile(".rept 16\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cmp_lt_f32_e64 s[10:11], v2, v33\n v_cmp_lt_f32_e64 s[10:11], v4, v33\n v_cmp_lt_f32_e64 s[10:11], v6, v33\n v_cmp_lt_f32_e64 s[10:11], v8, v33\n v_cmp_lt_f32_e64 s[10:11], v10, v33\n v_cmp_lt_f32_e64 s[10:11], v12, v33\n v_cmp_lt_f32_e64 s[10:11], v14, v33\n.endr\n" ::: CLOB);
        if (T == 29 && DEP == 1) asm volatile(".rept 16\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n.endr\n" ::: CLOB);
        if (T == 30 && DEP == 0) asm volatile(".rept 16\n v_cndmask_b32_e64 v0, v0, 1.0, s[10:11]\n v_cndmask_b32_e64 v2, v2, 1.0, s[10:11]\n v_cndmask_b32_e64 v4, v4, 1.0, s[10:11]\n v_cndmask_b32_e64 v6, v6, 1.0, s[10:11]\n v_cndmask_b32_e64 v8, v8, 1.0, s[10:11]\n v_cndmask_b32_e64 v10, v10, 1.0, s[10:11]\n v_cndmask_b32_e64 v12, v12, 1.0, s[10:11]\n v_cndmask_b32_e64 v14, v14, 1.0, s[10:11]\n.endr\n" ::: CLOB);
        if (T == 30 && DEP == 1) asm volatile(".rept 16\n v_cndmask_b32_e64 v0, v0, 1.0, s[10:11]\n v_cndmask_b32_e64 v0, v0, 1.0, s[10:11]\n v_cndmask_b32_e64 v0, v0, 1.0, s[10:11]\n v_cndmask_b32_e64 v0, v0, 1.0, s[10:11]\n v_cndmask_b32_e64 v0, v0, 1.0, s[10:11]\n v_cndmask_b32_e64 v0, v0, 1.0, s[10:11]\n v_cndmask_b32_e64 v0, v0, 1.0, s[10:11]\n v_cndmask_b32_e64 v0, v0, 1.0, s[10:11]\n.endr\n" ::: CLOB);
        if (T == 31 && DEP == 0) asm volatile(".rept 16\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cndmask_b32_e64 v0, v0, v33, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], v2, v33\n v_cndmask_b32_e64 v2, v2, v33, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], v4, v33\n v_cndmask_b32_e64 v4, v4, v33, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], v6, v33\n v_cndmask_b32_e64 v6, v6, v33, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], v8, v33\n v_cndmask_b32_e64 v8, v8, v33, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], v10, v33\n v_cndmask_b32_e64 v10, v10, v33, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], v12, v33\n v_cndmask_b32_e64 v12, v12, v33, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], v14, v33\n v_cndmask_b32_e64 v14, v14, v33, s[10:11]\n.endr\n" ::: CLOB);
        if (T == 31 && DEP == 1) asm volatile(".rept 16\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cndmask_b32_e64 v0, v0, v33, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cndmask_b32_e64 v0, v0, v33, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cndmask_b32_e64 v0, v0, v33, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cndmask_b32_e64 v0, v0, v33, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cndmask_b32_e64 v0, v0, v33, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cndmask_b32_e64 v0, v0, v33, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cndmask_b32_e64 v0, v0, v33, s[10:11]\n v_cmp_lt_f32_e64 s[10:11], v0, v33\n v_cndmask_b32_e64 v0, v0, v33, s[10:11]\n.endr\n" ::: CLOB);
        if (T == 32 && DEP == 0) asm volatile(".rept 16\n v_cvt_i32_f32_sdwa v0, v0 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n v_cvt_i32_f32_sdwa v2, v2 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n v_cvt_i32_f32_sdwa v4, v4 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n v_cvt_i32_f32_sdwa v6, v6 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n v_cvt_i32_f32_sdwa v8, v8 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n v_cvt_i32_f32_sdwa v10, v10 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n v_cvt_i32_f32_sdwa v12, v12 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n v_cvt_i32_f32_sdwa v14, v14 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n.endr\n" ::: CLOB);
        if (T == 32 && DEP == 1) asm volatile(".rept 16\n v_cvt_i32_f32_sdwa v0, v0 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n v_cvt_i32_f32_sdwa v0, v0 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n v_cvt_i32_f32_sdwa v0, v0 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n v_cvt_i32_f32_sdwa v0, v0 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n v_cvt_i32_f32_sdwa v0, v0 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n v_cvt_i32_f32_sdwa v0, v0 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n v_cvt_i32_f32_sdwa v0, v0 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n v_cvt_i32_f32_sdwa v0, v0 dst_sel:WORD_1 dst_unused:UNUSED_PAD src0_sel:DWORD\n.endr\n" ::: CLOB);
        if (T == 33 && DEP == 0) asm volatile(".rept 16\n v_add_u32_sdwa v0, v0, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u32_sdwa v2, v2, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u32_sdwa v4, v4, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u32_sdwa v6, v6, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u32_sdwa v8, v8, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u32_sdwa v10, v10, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u32_sdwa v12, v12, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u32_sdwa v14, v14, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n.endr\n" ::: CLOB);
        if (T == 33 && DEP == 1) asm volatile(".rept 16\n v_add_u32_sdwa v0, v0, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u32_sdwa v0, v0, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u32_sdwa v0, v0, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u32_sdwa v0, v0, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u32_sdwa v0, v0, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u32_sdwa v0, v0, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u32_sdwa v0, v0, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n v_add_u32_sdwa v0, v0, v33 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n.endr\n" ::: CLOB);
        if (T == 34 && DEP == 0) asm volatile(".rept 16\n v_sub_f32 v0, v0, v33\n v_sub_f32 v2, v2, v33\n v_sub_f32 v4, v4, v33\n v_sub_f32 v6, v6, v33\n v_sub_f32 v8, v8, v33\n v_sub_f32 v10, v10, v33\n v_sub_f32 v12, v12, v33\n v_sub_f32 v14, v14, v33\n.endr\n" ::: CLOB);
        if (T == 34 && DEP == 1) asm volatile(".rept 16\n v_sub_f32 v0, v0, v33\n v_sub_f32 v0, v0, v33\n v_sub_f32 v0, v0, v33\n v_sub_f32 v0, v0, v33\n v_sub_f32 v0, v0, v33\n v_sub_f32 v0, v0, v33\n v_sub_f32 v0, v0, v33\n v_sub_f32 v0, v0, v33\n.endr\n" ::: CLOB);
        if (T == 35 && DEP == 0) asm volatile(".rept 16\n v_max_f32_e64 v0, |v0|, |v33|\n v_max_f32_e64 v2, |v2|, |v33|\n v_max_f32_e64 v4, |v4|, |v33|\n v_max_f32_e64 v6, |v6|, |v33|\n v_max_f32_e64 v8, |v8|, |v33|\n v_max_f32_e64 v10, |v10|, |v33|\n v_max_f32_e64 v12, |v12|, |v33|\n v_max_f32_e64 v14, |v14|, |v33|\n.endr\n" ::: CLOB);
        if (T == 35 && DEP == 1) asm volatile(".rept 16\n v_max_f32_e64 v0, |v0|, |v33|\n v_max_f32_e64 v0, |v0|, |v33|\n v_max_f32_e64 v0, |v0|, |v33|\n v_max_f32_e64 v0, |v0|, |v33|\n v_max_f32_e64 v0, |v0|, |v33|\n v_max_f32_e64 v0, |v0|, |v33|\n v_max_f32_e64 v0, |v0|, |v33|\n v_max_f32_e64 v0, |v0|, |v33|\n.endr\n" ::: CLOB);
        if (T == 36 && DEP == 0) asm volatile(".rept 16\n v_mul_f32_e64 v0, v0, v33 clamp\n v_mul_f32_e64 v2, v2, v33 clamp\n v_mul_f32_e64 v4, v4, v33 clamp\n v_mul_f32_e64 v6, v6, v33 clamp\n v_mul_f32_e64 v8, v8, v33 clamp\n v_mul_f32_e64 v10, v10, v33 clamp\n v_mul_f32_e64 v12, v12, v33 clamp\n v_mul_f32_e64 v14, v14, v33 clamp\n.endr\n" ::: CLOB);
        if (T == 36 && DEP == 1) asm volatile(".rept 16\n v_mul_f32_e64 v0, v0, v33 clamp\n v_mul_f32_e64 v0, v0, v33 clamp\n v_mul_f32_e64 v0, v0, v33 clamp\n v_mul_f32_e64 v0, v0, v33 clamp\n v_mul_f32_e64 v0, v0, v33 clamp\n v_mul_f32_e64 v0, v0, v33 clamp\n v_mul_f32_e64 v0, v0, v33 clamp\n v_mul_f32_e64 v0, v0, v33 clamp\n.endr\n" ::: CLOB);
        if (T == 37 && DEP == 0) asm volatile(".rept 16\n v_fmamk_f32 v0, v0, 0x3ecccccd, v34\n v_fmamk_f32 v2, v2, 0x3ecccccd, v34\n v_fmamk_f32 v4, v4, 0x3ecccccd, v34\n v_fmamk_f32 v6, v6, 0x3ecccccd, v34\n v_fmamk_f32 v8, v8, 0x3ecccccd, v34\n v_fmamk_f32 v10, v10, 0x3ecccccd, v34\n v_fmamk_f32 v12, v12, 0x3ecccccd, v34\n v_fmamk_f32 v14, v14, 0x3ecccccd, v34\n.endr\n" ::: CLOB);
        if (T == 37 && DEP == 1) asm volatile(".rept 16\n v_fmamk_f32 v0, v0, 0x3ecccccd, v34\n v_fmamk_f32 v0, v0, 0x3ecccccd, v34\n v_fmamk_f32 v0, v0, 0x3ecccccd, v34\n v_fmamk_f32 v0, v0, 0x3ecccccd, v34\n v_fmamk_f32 v0, v0, 0x3ecccccd, v34\n v_fmamk_f32 v0, v0, 0x3ecccccd, v34\n v_fmamk_f32 v0, v0, 0x3ecccccd, v34\n v_fmamk_f32 v0, v0, 0x3ecccccd, v34\n.endr\n" ::: CLOB);
    }
    float r; asm volatile("v_add_f32 %0, v0, v2\n v_add_f32 %0, %0, v4" : "=v"(r) :: CLOB);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
extern __shared__ float pad[];
template <int T> void run(const char* name, int per)
{
    float* d; (void)hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-34s", name);
    float ms;
    {   // throughput: 8 waves per SIMD, independent
        const int iters = 200, blocks = 256 * 8 * 4;
        hipLaunchKernelGGL((k<T, 0>), dim3(blocks), dim3(256), 0, 0, d, 10); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); hipLaunchKernelGGL((k<T, 0>), dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("  throughput %6.2f", ms * 1e-3 / (4.0 * 8 * iters * 128 * per) * 2.4e9);
    }
    {   // latency: one wave per SIMD (one 256-thread block per CU), one dependent chain
        (void)hipFuncSetAttribute((const void*)k<T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        const int iters = 200, blocks = 256;
        hipLaunchKernelGGL((k<T, 1>), dim3(blocks), dim3(256), 150 * 1024, 0, d, 10); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); hipLaunchKernelGGL((k<T, 1>), dim3(blocks), dim3(256), 150 * 1024, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("   latency %6.2f\n", ms * 1e-3 / (1.0 * iters * 128 * per) * 2.4e9);
    }
}
int main()
{
    printf("cycles @2.4 GHz nominal per wave64 instruction: throughput per SIMD (8 waves x 8 independent chains) / latency (1 wave, dependent chain)\n");
    run<0>("fma", 1);
    run<1>("mul", 1);
    run<2>("rcp", 1);
    run<3>("fract", 1);
    run<4>("frexp_mant", 1);
    run<5>("frexp_exp", 1);
    run<6>("ldexp", 1);
    run<7>("cvt_i32_f32", 1);
    run<8>("cvt_f32_i32", 1);
    run<9>("cvt_f32_ubyte1", 1);
    run<10>("min", 1);
    run<11>("med3", 1);
    run<12>("min3", 1);
    run<13>("mad_u32_u24", 1);
    run<14>("mul_u32_u24", 1);
    run<15>("mul_lo_u32", 1);
    run<16>("lshl_add_u64", 1);
    run<17>("alignbit", 1);
    run<18>("lshl_or", 1);
    run<19>("and_or", 1);
    run<20>("perm", 1);
    run<21>("bitop3_b16", 1);
    run<22>("lshlrev", 1);
    run<23>("lshrrev", 1);
    run<24>("add_u32", 1);
    run<25>("fma sgpr", 1);
    run<26>("add sgpr", 1);
    run<27>("mov sgpr", 1);
    run<28>("cmp_e64 (sgpr src, sgpr dst)", 1);
    run<29>("cmp_e64 vgprs", 1);
    run<30>("cndmask_e64", 1);
    run<31>("cmp+cndmask pair (2 instr)", 2);
    run<32>("cvt_i32_f32 sdwa", 1);
    run<33>("add_u32 sdwa", 1);
    run<34>("sub", 1);
    run<35>("max_e64 abs", 1);
    run<36>("mul clamp", 1);
    run<37>("fmamk", 1);
    return 0;
}
