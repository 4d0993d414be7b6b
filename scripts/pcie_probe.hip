// Host <-> device link probe for the host-fed stream ("p99 ms/frame including H2D of the input and D2H of the output
// when frames are host-resident" -- how Modules/OBS-Plugin/Interop/FrameIngest.cpp:415-474,567-602 feeds the filter).  One 4K I420 frame
// is 12.44 MB each way.  Measures, per direction and for both at once:
//   * hipMemcpyAsync on 1 / 2 / 4 streams (the copy engines; planes split over the streams),
//   * copy KERNELS reading / writing pinned host memory directly (what a zero-copy ingest / remap sink would do), for several grid sizes
//     and for coherent (hipHostMallocDefault) and non-coherent (hipHostMallocNonCoherent) pinned memory,
//   * a copy engine one way with a kernel the other way.
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/pcie_probe_bin scripts/pcie_probe.hip     Output kept as profiles/r03_pcie_probe.txt
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__global__ void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}
// four independent 16-byte loads in flight per thread before the stores
__global__ void k_copy4(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride)
    {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const size_t N = (size_t)3840 * 2160 * 3 / 2;      // one I420 frame
    const int ITERS = 40;
    void *h_in[2], *h_out[2], *d_in, *d_out;
    CK(hipHostMalloc(&h_in[0], N, hipHostMallocDefault)); CK(hipHostMalloc(&h_out[0], N, hipHostMallocDefault));
    CK(hipHostMalloc(&h_in[1], N, hipHostMallocNonCoherent)); CK(hipHostMalloc(&h_out[1], N, hipHostMallocNonCoherent));
    CK(hipMalloc(&d_in, N)); CK(hipMalloc(&d_out, N));
    for (int k = 0; k < 2; k++) { for (size_t i = 0; i < N; i++) ((unsigned char*)h_in[k])[i] = (unsigned char)(i * 7 + k); }
    hipStream_t s[8];
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    auto timed = [&](const char* name, auto&& body) {
        for (int w = 0; w < 3; w++) body();
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int it = 0; it < ITERS; it++) body();
        CK(hipDeviceSynchronize());
        const double dt = (now() - t0) / ITERS;
        std::printf("%-78s %7.3f ms/frame  %6.1f GB/s per direction\n", name, dt * 1e3, N / dt / 1e9);
    };
    auto sdma = [&](bool h2d, int ns, int base, int mem) {
        const size_t chunk = (N / ns + 255) / 256 * 256;
        for (int k = 0; k < ns; k++)
        {
            const size_t off = (size_t)k * chunk, len = off >= N ? 0 : (N - off < chunk ? N - off : chunk);
            if (!len) continue;
            if (h2d) CK(hipMemcpyAsync((char*)d_in + off, (char*)h_in[mem] + off, len, hipMemcpyHostToDevice, s[base + k]));
            else CK(hipMemcpyAsync((char*)h_out[mem] + off, (char*)d_out + off, len, hipMemcpyDeviceToHost, s[base + k]));
        }
    };
    auto kern = [&](bool h2d, int blocks, int stream, int mem, bool four) {
        const uint4* src = (const uint4*)(h2d ? h_in[mem] : d_out); uint4* dst = (uint4*)(h2d ? d_in : h_out[mem]);
        if (four) hipLaunchKernelGGL(k_copy4, dim3(blocks), dim3(256), 0, s[stream], src, dst, N / 16);
        else hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, s[stream], src, dst, N / 16);
    };
    char name[160];
    std::printf("== copy engines (hipMemcpyAsync, pinned coherent memory)\n");
    for (int ns : {1, 2, 4}) { std::snprintf(name, sizeof name, "H2D, %d stream(s)", ns); timed(name, [&] { sdma(true, ns, 0, 0); }); }
    for (int ns : {1, 2, 4}) { std::snprintf(name, sizeof name, "D2H, %d stream(s)", ns); timed(name, [&] { sdma(false, ns, 0, 0); }); }
    for (int ns : {1, 2, 4}) { std::snprintf(name, sizeof name, "H2D + D2H at once, %d stream(s) each", ns); timed(name, [&] { sdma(true, ns, 0, 0); sdma(false, ns, 4, 0); }); }
    timed("H2D + D2H at once, 1 stream each, non-coherent pinned memory", [&] { sdma(true, 1, 0, 1); sdma(false, 1, 4, 1); });
    std::printf("== copy kernels on pinned host memory (zero copy)\n");
    for (int mem : {0, 1})
        for (int blocks : {64, 256, 1024, 4096})
        {
            std::snprintf(name, sizeof name, "kernel H2D, %4d blocks, %s", blocks, mem ? "non-coherent" : "coherent"); timed(name, [&] { kern(true, blocks, 0, mem, false); });
            std::snprintf(name, sizeof name, "kernel D2H, %4d blocks, %s", blocks, mem ? "non-coherent" : "coherent"); timed(name, [&] { kern(false, blocks, 1, mem, false); });
        }
    for (int blocks : {256, 1024})
    {
        std::snprintf(name, sizeof name, "kernel H2D x4 loads in flight, %4d blocks, coherent", blocks); timed(name, [&] { kern(true, blocks, 0, 0, true); });
        std::snprintf(name, sizeof name, "kernel H2D + kernel D2H at once, %4d blocks each, coherent", blocks); timed(name, [&] { kern(true, blocks, 0, 0, false); kern(false, blocks, 1, 0, false); });
        std::snprintf(name, sizeof name, "kernel H2D + kernel D2H at once, %4d blocks each, non-coherent", blocks); timed(name, [&] { kern(true, blocks, 0, 1, false); kern(false, blocks, 1, 1, false); });
    }
    std::printf("== mixed\n");
    timed("copy engine H2D + kernel D2H (1024 blocks)", [&] { sdma(true, 1, 0, 0); kern(false, 1024, 1, 0, false); });
    timed("kernel H2D (1024 blocks) + copy engine D2H", [&] { kern(true, 1024, 0, 0, false); sdma(false, 1, 4, 0); });
    timed("copy engines H2D x2 + D2H x2 + kernel H2D of a second frame", [&] { sdma(true, 2, 0, 0); sdma(false, 2, 4, 0); kern(true, 256, 2, 1, false); });
    // single-frame latencies (one at a time, synchronised)
    std::printf("== one transfer at a time (launch to completion, mean of %d)\n", ITERS);
    auto lat = [&](const char* nm, auto&& body) {
        double acc = 0;
        for (int it = 0; it < ITERS + 3; it++) { CK(hipDeviceSynchronize()); const double t0 = now(); body(); CK(hipDeviceSynchronize()); if (it >= 3) acc += now() - t0; }
        std::printf("%-78s %7.3f ms\n", nm, acc / ITERS * 1e3);
    };
    lat("H2D copy engine, 1 stream", [&] { sdma(true, 1, 0, 0); });
    lat("H2D copy engine, 4 streams", [&] { sdma(true, 4, 0, 0); });
    lat("H2D kernel, 1024 blocks", [&] { kern(true, 1024, 0, 0, false); });
    lat("D2H copy engine, 1 stream", [&] { sdma(false, 1, 0, 0); });
    lat("D2H copy engine, 4 streams", [&] { sdma(false, 4, 0, 0); });
    lat("D2H kernel, 1024 blocks", [&] { kern(false, 1024, 1, 0, false); });
    // correctness of the last kernel copies
    CK(hipMemcpy(d_out, h_in[0], N, hipMemcpyHostToDevice)); kern(false, 1024, 1, 0, false); CK(hipDeviceSynchronize());
    size_t bad = 0; for (size_t i = 0; i < N; i += 4099) bad += ((unsigned char*)h_out[0])[i] != ((unsigned char*)h_in[0])[i];
    std::printf("verify: %zu mismatches\n", bad);
    return 0;
}
