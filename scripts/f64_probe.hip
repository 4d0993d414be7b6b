// fp64 VALU issue rate and dependent latency, LDS read latency, workgroup barrier cost on gfx950 (one wavefront per SIMD, like k_mesh_solve).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters)
{
    __shared__ double lds[1024];
    const int tid = threadIdx.x;
    lds[tid] = tid; lds[tid + 256] = 1.0; __syncthreads();
    double a[16]; for (int i = 0; i < 16; i++) a[i] = 1.0 + i + tid * 1e-3;
    double x = 1.0 + tid * 1e-6;
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = a[i] * 1.0000001;          // 16 independent v_mul_f64
    }
    long long t1 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) x = x * 1.0000001;                // dependent chain
    }
    long long t2 = clock64();
    int idx = tid;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) idx = (int)lds[idx & 255] & 255;   // dependent LDS reads (+ convert)
    }
    long long t3 = clock64();
    for (int it = 0; it < iters; it++) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    long long t4 = clock64();
    double y = x;
    for (int it = 0; it < iters; it++) { y = 1.0 / (y + 1.5); }
    long long t5 = clock64();
    double s = 0; for (int i = 0; i < 16; i++) s += a[i];
    out[tid] = s + x + idx + y;
    if (tid == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; }
}
int main()
{
    double* o; long long* c; hipMalloc(&o, 256 * 8); hipMalloc(&c, 64);
    const int iters = 1000;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, o, c, iters); hipDeviceSynchronize(); }
    long long h[5]; hipMemcpy(h, c, 40, hipMemcpyDeviceToHost);
    printf("clock64 ticks per op (one wave per SIMD): independent v_mul_f64 %.2f, dependent v_mul_f64 %.2f, dependent LDS read+cvt %.2f, barrier (4 waves) %.2f, dependent 1/x f64 %.2f\n",
           h[0] / (16.0 * iters), h[1] / (16.0 * iters), h[2] / (4.0 * iters), h[3] / (double)iters, h[4] / (double)iters);
    return 0;
}
