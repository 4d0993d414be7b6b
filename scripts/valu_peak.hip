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
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int MODE> void run(const char* name, double ops_per_iter_per_lane, double flops_per_op)
{
    float* d; hipMalloc(&d, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 4096;
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
    return 0;
}
