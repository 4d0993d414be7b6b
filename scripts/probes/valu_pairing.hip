// Probe (round 6): when do two VALU instructions share one 4-cycle issue round of a gfx950 SIMD?  profiles/r06_remap_ifetch.txt shows
// SQ_ACTIVE_INST_VALU2 = 0.41 x SQ_ACTIVE_INST_VALU for the remap: 41 % of its VALU instructions are the second of a pair.  This probe
// measures cycles per wave64 VALU instruction per SIMD for instruction streams of known dependence structure at 1 / 2 / 4 / 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -w -o scripts/probes/valu_pairing_bin scripts/probes/valu_pairing.hip
#include <hip/hip_runtime.h>
#include <cstdio>

extern __shared__ float s_pad[];

#define A(INS, R) asm volatile(INS : "+v"(R) : "v"(b), "v"(c))
#define FMA "v_fma_f32 %0, %0, %1, %2"
#define MAXI "v_max_f32 %0, %0, %1"
#define CVT "v_cvt_f32_ubyte1 %0, %0"

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0000001f, c = 0.5f;
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
        {
            if (MODE == 0) { A(FMA, a0); A(FMA, a0); A(FMA, a0); A(FMA, a0); A(FMA, a0); A(FMA, a0); A(FMA, a0); A(FMA, a0); }          // 1 chain
            if (MODE == 1) { A(FMA, a0); A(FMA, a1); A(FMA, a0); A(FMA, a1); A(FMA, a0); A(FMA, a1); A(FMA, a0); A(FMA, a1); }          // 2 chains alternating
            if (MODE == 2) { A(FMA, a0); A(FMA, a0); A(FMA, a1); A(FMA, a1); A(FMA, a0); A(FMA, a0); A(FMA, a1); A(FMA, a1); }          // 2 chains, dependent neighbours
            if (MODE == 3) { A(FMA, a0); A(FMA, a1); A(FMA, a2); A(FMA, a3); A(FMA, a0); A(FMA, a1); A(FMA, a2); A(FMA, a3); }          // 4 chains
            if (MODE == 4) { A(FMA, a0); A(FMA, a1); A(FMA, a2); A(FMA, a3); A(FMA, a4); A(FMA, a5); A(FMA, a6); A(FMA, a7); }          // 8 chains
            if (MODE == 5) { A(MAXI, a0); A(MAXI, a0); A(MAXI, a0); A(MAXI, a0); A(MAXI, a0); A(MAXI, a0); A(MAXI, a0); A(MAXI, a0); }  // slow, 1 chain
            if (MODE == 6) { A(FMA, a0); A(FMA, a0); A(FMA, a0); A(MAXI, a0); A(FMA, a0); A(FMA, a0); A(FMA, a0); A(MAXI, a0); }        // 3:1, 1 chain
            if (MODE == 7) { A(FMA, a0); A(FMA, a1); A(FMA, a2); A(MAXI, a3); A(FMA, a4); A(FMA, a5); A(FMA, a6); A(MAXI, a7); }        // 3:1, 8 chains
            if (MODE == 8) { A(FMA, a0); A(CVT, a1); A(MAXI, a2); A(FMA, a3); A(FMA, a4); A(MAXI, a5); A(CVT, a6); A(FMA, a7); }        // slow-slow neighbours, 8 chains
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (iters < 0 ? s_pad[threadIdx.x] : 0.0f);
}

template <int MODE> void run(const char* name)
{
    float* d; (void)hipMalloc(&d, 8192 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    printf("%-34s", name);
    for (int wps : {1, 2, 3, 4, 6, 8})
    {
        // a block = 4 waves = one per SIMD; LDS per block sized so that exactly `wps` blocks fit a CU (160 KB)
        const size_t lds = wps == 8 ? 0 : (size_t)(160 * 1024 / wps) - 256;
        const int iters = 400, blocks = 256 * wps * 4;              // 4 full rounds of the chip
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, 10, 1.0f);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, d, iters, 1.0f); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_simd = 4.0 * wps * iters * 16 * 8;    // 4 rounds x wps waves x instructions per wave
        // the clock is not known here (1.9-2.4 GHz depending on the load); print ns per instruction per SIMD and cycles at 2.4 GHz
        printf("  w%d: %5.2f", wps, ms * 1e-3 / instr_per_simd * 2.4e9);
    }
    printf("   (cycles @2.4 GHz per wave64 instr per SIMD)\n");
    (void)hipFree(d);
}

int main()
{
    run<0>("fma, 1 chain");
    run<1>("fma, 2 chains alternating");
    run<2>("fma, 2 chains, dependent pairs");
    run<3>("fma, 4 chains");
    run<4>("fma, 8 chains");
    run<5>("max, 1 chain");
    run<6>("fma:max 3:1, 1 chain");
    run<7>("fma:max 3:1, 8 chains");
    run<8>("fma/cvt/max, slow neighbours");
    return 0;
}
