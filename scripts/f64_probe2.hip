// fp64 VALU throughput on gfx950 by operand kind, one wavefront per SIMD: product of two VGPR pairs, product + difference pairs as in the
// band factorisation, fused multiply-add.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters, double seed)
{
    const int tid = threadIdx.x;
    double a[16], b[16];
    for (int i = 0; i < 16; i++) { a[i] = 1.0 + i + tid * 1e-3; b[i] = 1.0 + seed * (i + 1) * 1e-9; }
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = a[i] * b[i];                  // VGPR x VGPR
    }
    long long t1 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = a[i] - b[i] * b[(i + 1) & 15];   // mul + sub (separately rounded: -ffp-contract=off)
    }
    long long t2 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = __builtin_fma(-b[i], b[(i + 1) & 15], a[i]);
    }
    long long t3 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = a[i] + b[i];
    }
    long long t4 = clock64();
    double s = 0; for (int i = 0; i < 16; i++) s += a[i];
    out[tid + blockIdx.x * 256] = s;
    if (tid == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
}
int main()
{
    double* o; long long* c; hipMalloc(&o, 1024 * 256 * 8); hipMalloc(&c, 64);
    const int iters = 1000;
    for (int blocks : {1, 1024})
    {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, o, c, iters, 1.0); hipDeviceSynchronize(); }
        long long h[4]; hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
        printf("%4d block(s) of 4 waves: ticks per wave-op: mul(v,v) %.2f | mul+sub pair %.2f per pair | fma %.2f | add(v,v) %.2f\n", blocks,
               h[0] / (16.0 * iters), h[1] / (16.0 * iters), h[2] / (16.0 * iters), h[3] / (16.0 * iters));
    }
    return 0;
}
