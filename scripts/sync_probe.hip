// How much later than a kernel's last store to pinned host memory does hipStreamSynchronize return?  On MI355X / ROCm 7.2: 6-9 us.
// (Tried in round 2: the last kernel of the tracker chain posting a sequence number to pinned memory, the push polling it instead of
// calling hipStreamSynchronize.  In the pipeline it bought nothing -- 7 890 vs 7 945 frames/s: the next HIP call pays the runtime's
// completion handling anyway -- and with a store-completion wait in place of __threadfence_system() (whose L2 write-back cost 10 us
// next to the bulk stream's kernels) a parity test caught a stale read.  Not kept.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(volatile int* flag, int seq, int spin)
{
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) { }
    if (threadIdx.x == 0) { __threadfence_system(); *flag = seq; }
}
int main()
{
    int* flag; if (hipHostMalloc(&flag, 64, hipHostMallocDefault) != hipSuccess) return 1;
    *flag = 0;
    hipStream_t st; (void)hipStreamCreate(&st);
    using clk = std::chrono::steady_clock;
    for (int spin : {2000, 8000})      // kernel duration in 100 MHz ticks: 20 us, 80 us
    {
        std::vector<double> a, b, c;
        for (int i = 1; i <= 300; i++)
        {
            const auto t0 = clk::now();
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, st, flag, i + spin * 1000, spin);
            const auto t1 = clk::now();
            while (*(volatile int*)flag != i + spin * 1000) { __builtin_ia32_pause(); }
            const auto t2 = clk::now();
            (void)hipStreamSynchronize(st);
            const auto t3 = clk::now();
            a.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
            b.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count());
            c.push_back(std::chrono::duration<double, std::micro>(t3 - t2).count());
        }
        auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("kernel %d us: launch call %.2f us, flag seen %.2f us after launch, hipStreamSynchronize returns %.2f us after the flag (medians)\n", spin / 100, med(a), med(b), med(c));
    }
    // a third way: spin on hipStreamQuery
    for (int spin : {2000, 8000})
    {
        std::vector<double> b, c;
        for (int i = 1; i <= 300; i++)
        {
            const auto t0 = clk::now();
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, st, flag, i + 7 + spin * 1000, spin);
            while (*(volatile int*)flag != i + 7 + spin * 1000) { __builtin_ia32_pause(); }
            const auto t2 = clk::now();
            while (hipStreamQuery(st) != hipSuccess) { }
            const auto t3 = clk::now();
            b.push_back(std::chrono::duration<double, std::micro>(t2 - t0).count());
            c.push_back(std::chrono::duration<double, std::micro>(t3 - t2).count());
        }
        std::sort(b.begin(), b.end()); std::sort(c.begin(), c.end());
        printf("kernel %d us: flag seen %.2f us after launch, a hipStreamQuery spin succeeds %.2f us after the flag (medians)\n", spin / 100, b[b.size() / 2], c[c.size() / 2]);
    }
    for (int spin : {2000, 8000})
    {
        std::vector<double> b;
        for (int i = 1; i <= 300; i++)
        {
            const auto t0 = clk::now();
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, st, flag, -i, spin);
            while (hipStreamQuery(st) != hipSuccess) { }
            const auto t3 = clk::now();
            b.push_back(std::chrono::duration<double, std::micro>(t3 - t0).count());
        }
        std::sort(b.begin(), b.end());
        printf("kernel %d us: launch + hipStreamQuery spin %.2f us (median)\n", spin / 100, b[b.size() / 2]);
    }
    // and the plain way: launch + synchronize, no polling
    for (int spin : {2000, 8000})
    {
        std::vector<double> b;
        for (int i = 1; i <= 300; i++)
        {
            const auto t0 = clk::now();
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, st, flag, -i, spin);
            (void)hipStreamSynchronize(st);
            const auto t3 = clk::now();
            b.push_back(std::chrono::duration<double, std::micro>(t3 - t0).count());
        }
        std::sort(b.begin(), b.end());
        printf("kernel %d us: launch + hipStreamSynchronize %.2f us (median)\n", spin / 100, b[b.size() / 2]);
    }
    return 0;
}
