// Probe: GPU-side gaps between dependent small kernels on one stream -- plain launches vs a captured hipGraph.
// hipcc --offload-arch=gfx950 -O3 -o gap_probe gap_probe.hip && ./gap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void spin(long long cycles, int* sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = (int)t0;
}

int main()
{
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* d; CK(hipMalloc(&d, 4));
    int* h; CK(hipHostMalloc(&h, 4, hipHostMallocDefault));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int us[7] = {7, 16, 8, 45, 5, 11, 40};                 // the tracker chain's kernel durations
    const int grid[7] = {544, 135, 600, 1856, 1, 128, 1};
    const int blk[7] = {256, 256, 256, 64, 1024, 256, 256};
    // wall_clock64 ticks at 100 MHz
    auto chain = [&](bool host_write) {
        for (int k = 0; k < 7; k++)
            hipLaunchKernelGGL(spin, dim3(grid[k]), dim3(blk[k]), 0, s, (long long)us[k] * 100, (host_write && (k == 4 || k == 6)) ? h : d);
    };
    for (int mode = 0; mode < 4; mode++)
    {
        const bool graph = mode & 1, host_write = mode & 2;
        hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
        if (graph)
        {
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            chain(host_write);
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        float best = 1e9f, sum = 0; double host_us = 0;
        for (int it = 0; it < 60; it++)
        {
            CK(hipStreamSynchronize(s));
            const auto t0 = std::chrono::steady_clock::now();
            CK(hipEventRecord(a, s));
            if (graph) CK(hipGraphLaunch(ge, s)); else chain(host_write);
            CK(hipEventRecord(b, s));
            const auto t1 = std::chrono::steady_clock::now();
            CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (it >= 10) { best = ms < best ? ms : best; sum += ms; host_us += std::chrono::duration<double, std::micro>(t1 - t0).count(); }
        }
        std::printf("%-6s host_write=%d: GPU chain %.1f us avg (best %.1f; kernels sum 132), host submit %.1f us\n", graph ? "graph" : "stream", (int)host_write,
                    sum / 50 * 1e3, best * 1e3, host_us / 50);
    }
    return 0;
}
