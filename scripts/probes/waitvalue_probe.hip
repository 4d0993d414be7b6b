// Can the output remap be put on the bulk stream BEFORE the tracker chain's synchronisation, behind a stream wait-value the host releases once
// the path smoother has the matrix (round-4 VERDICT item 4)?  What this probe measures, on two non-blocking streams like the filter's:
//   (1) the host cost of hipStreamWaitValue32 + a kernel launch behind it (paid while the host waits for the chain anyway);
//   (2) the host cost of releasing it (a store to the flag -- host memory the GPU polls -- against hipStreamWriteValue32 / a launch);
//   (3) release -> kernel start -> kernel end latency, against a plain launch -> end;
//   (4) whether kernels of ANOTHER stream run while the wait is pending (if they queue behind it the chain would deadlock against its own
//       output remap), with and without a long kernel in front of the wait.
// Development probe, not part of the library:  hipcc --offload-arch=gfx950 -O2 -o waitvalue_probe waitvalue_probe.hip && timeout 60 ./waitvalue_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_mark(volatile unsigned* out, unsigned v) { if (threadIdx.x == 0 && blockIdx.x == 0) *out = v; }
__global__ void k_work(float* p, int n, int iters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = p[i];
    for (int k = 0; k < iters; k++) v = v * 1.0001f + 0.5f;
    p[i] = v;
}
// reads its "matrix" from memory the host fills before the release, like a pre-launched remap would
__global__ void k_consume(const float* H, float* out) { if (threadIdx.x < 9 && blockIdx.x == 0) out[threadIdx.x] = H[threadIdx.x] * 2.0f; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t track, bulk;
    CK(hipStreamCreateWithFlags(&track, hipStreamNonBlocking));
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&bulk, hipStreamNonBlocking, lo));
    unsigned* flag; CK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault)); *flag = 0;
    unsigned* sig = nullptr;
    const bool have_sig = hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory) == hipSuccess;
    if (!have_sig) (void)hipGetLastError();
    volatile unsigned* mark; CK(hipHostMalloc((void**)&mark, 64, hipHostMallocDefault)); *mark = 0;
    float* H; CK(hipHostMalloc((void**)&H, 64, hipHostMallocDefault));
    float* out; CK(hipHostMalloc((void**)&out, 64, hipHostMallocDefault));
    float* buf; const int n = 256 * 1024; CK(hipMalloc(&buf, n * 4)); CK(hipMemset(buf, 0, n * 4));
    for (int r = 0; r < 20; r++) { hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, track, mark, 1u); hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, bulk, buf, n, 10); }
    CK(hipDeviceSynchronize());

    // ---- (4) first: does another stream make progress while a wait is pending on `bulk`?
    *flag = 0; *mark = 0;
    CK(hipStreamWaitValue32(bulk, flag, 1u, hipStreamWaitValueEq, 0xffffffffu));
    hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, bulk, mark + 1, 7u);
    hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, track, mark, 5u);
    double t0 = now_us(); bool other_ran = false;
    while (now_us() - t0 < 200000.0) { if (*mark == 5u) { other_ran = true; break; } }
    std::printf("(4) a kernel on the tracking stream while a wait-value is pending on the bulk stream: %s (%.0f us)\n", other_ran ? "RUNS" : "BLOCKED -- would deadlock the push", now_us() - t0);
    *flag = 1; CK(hipStreamSynchronize(bulk)); CK(hipStreamSynchronize(track));
    if (!other_ran) { std::printf("stopping here\n"); return 0; }
    // the same with a long kernel running on the bulk stream in front of the wait (the previous frame's remap)
    *flag = 0; *mark = 0;
    hipLaunchKernelGGL(k_work, dim3(n / 256), dim3(256), 0, bulk, buf, n, 20000);
    CK(hipStreamWaitValue32(bulk, flag, 1u, hipStreamWaitValueEq, 0xffffffffu));
    hipLaunchKernelGGL(k_mark, dim3(1), dim3(64), 0, bulk, mark + 1, 7u);
    int chain_done = 0; t0 = now_us();
    for (int k = 0; k < 6; k++) hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, track, buf, 64 * 256, 50);
    CK(hipStreamSynchronize(track)); chain_done = 1;
    std::printf("    a 6-kernel chain on the tracking stream behind a busy bulk stream + pending wait: done in %.0f us (%d)\n", now_us() - t0, chain_done);
    *flag = 1; CK(hipStreamSynchronize(bulk));

    // ---- (1)-(3): timing, 2000 rounds shaped like a push: [pre-enqueue wait + kernel] ... host works ... [release] ... sync
    const int reps = 2000;
    double t_pre = 0, t_rel = 0, t_end = 0, t_plain_launch = 0, t_plain_end = 0;
    for (int r = 0; r < reps; r++)
    {
        *flag = 0;
        double a = now_us();
        CK(hipStreamWaitValue32(bulk, flag, (unsigned)(r + 1), hipStreamWaitValueEq, 0xffffffffu));
        hipLaunchKernelGGL(k_consume, dim3(1), dim3(64), 0, bulk, H, out);
        double b = now_us();
        // (the chain would run here; give the queue processor time to reach the wait)
        while (now_us() - b < 40.0) { }
        for (int q = 0; q < 9; q++) H[q] = (float)(r + q);
        double c = now_us();
        __atomic_store_n(flag, (unsigned)(r + 1), __ATOMIC_RELEASE);
        double d = now_us();
        CK(hipStreamSynchronize(bulk));
        double e = now_us();
        if (out[3] != 2.0f * (float)(r + 3)) { std::printf("wrong value consumed at round %d\n", r); return 1; }
        t_pre += b - a; t_rel += d - c; t_end += e - d;
        // plain: launch when the matrix is known
        a = now_us();
        hipLaunchKernelGGL(k_consume, dim3(1), dim3(64), 0, bulk, H, out);
        b = now_us();
        CK(hipStreamSynchronize(bulk));
        t_plain_launch += b - a; t_plain_end += now_us() - b;
    }
    std::printf("(1) pre-enqueue (wait-value on host memory + launch): %.2f us of host time, off the critical path\n", t_pre / reps);
    std::printf("(2) release = one host store: %.3f us;   (3) release -> kernel done + sync returned: %.2f us\n", t_rel / reps, t_end / reps);
    std::printf("    plain: launch %.2f us of host time ON the critical path, launch returned -> kernel done + sync returned: %.2f us\n", t_plain_launch / reps, t_plain_end / reps);
    if (have_sig)
    {
        double t_pre2 = 0, t_rel2 = 0, t_end2 = 0;
        for (int r = 0; r < reps; r++)
        {
            double a = now_us();
            CK(hipStreamWaitValue32(bulk, sig, (unsigned)(r + 1), hipStreamWaitValueEq, 0xffffffffu));
            hipLaunchKernelGGL(k_consume, dim3(1), dim3(64), 0, bulk, H, out);
            double b = now_us();
            while (now_us() - b < 40.0) { }
            double c = now_us();
            CK(hipStreamWriteValue32(track, sig, (unsigned)(r + 1), 0));
            double d = now_us();
            CK(hipStreamSynchronize(bulk));
            t_pre2 += b - a; t_rel2 += d - c; t_end2 += now_us() - d;
        }
        std::printf("    signal memory + hipStreamWriteValue32 on the other stream: pre %.2f us, release call %.2f us, release -> done %.2f us\n", t_pre2 / reps, t_rel2 / reps, t_end2 / reps);
    }
    std::printf("done\n");
    return 0;
}
