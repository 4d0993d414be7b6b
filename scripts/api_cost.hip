// Host-side cost of the HIP calls the per-frame path is made of (us per call, idle streams).  hipcc scripts/api_cost.hip -o scripts/api_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_nop() {}
template <class F> double us_per(F f, int n = 2000)
{
    for (int i = 0; i < 50; i++) f();
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; i++) f();
    auto t1 = std::chrono::steady_clock::now();
    (void)hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}
int main()
{
    hipStream_t a, b; hipEvent_t e, e2;
    (void)hipStreamCreateWithFlags(&a, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); (void)hipEventCreateWithFlags(&e2, hipEventDisableTiming);
    std::printf("hipStreamQuery (idle)            %.2f us\n", us_per([&] { (void)hipStreamQuery(a); }));
    std::printf("hipEventRecord                   %.2f us\n", us_per([&] { (void)hipEventRecord(e, a); }));
    std::printf("hipEventRecord + StreamWaitEvent %.2f us\n", us_per([&] { (void)hipEventRecord(e, a); (void)hipStreamWaitEvent(b, e, 0); }));
    std::printf("hipEventQuery                    %.2f us\n", us_per([&] { (void)hipEventQuery(e); }));
    std::printf("kernel launch                    %.2f us\n", us_per([&] { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, a); }));
    std::printf("launch + hipStreamSynchronize    %.2f us\n", us_per([&] { hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, a); (void)hipStreamSynchronize(a); }));
    std::printf("hipStreamSynchronize (idle)      %.2f us\n", us_per([&] { (void)hipStreamSynchronize(a); }));
    std::printf("record + hipEventSynchronize     %.2f us\n", us_per([&] { (void)hipEventRecord(e2, a); (void)hipEventSynchronize(e2); }));
    return 0;
}
