// Host cost of putting a 6-kernel dependent chain on a stream, three ways: plain launches, a hipGraph replayed as is, and a hipGraph with two
// of its kernel nodes re-pointed before every replay (what a push would need: the caller's planes change every frame).  Development probe,
// not part of the library.   hipcc --offload-arch=gfx950 -O2 -o launch_cost launch_cost.hip && ./launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Args { const float* in; float* out; int n; int pad[21]; };          // ~100 bytes, like the tracker's kernels

__global__ void k_step(Args a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.n) a.out[i] = a.in[i] * 1.0001f + 1.0f;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    const int N = 6, n = 64 * 256, reps = 2000;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float* buf[2]; CK(hipMalloc(&buf[0], n * 4)); CK(hipMalloc(&buf[1], n * 4)); CK(hipMemset(buf[0], 0, n * 4));
    float* alt; CK(hipMalloc(&alt, n * 4)); CK(hipMemset(alt, 0, n * 4));

    auto chain = [&](const float* first) {
        for (int k = 0; k < N; k++) { Args a{}; a.in = k == 0 ? first : buf[(k + 1) & 1]; a.out = buf[k & 1]; a.n = n; hipLaunchKernelGGL(k_step, dim3(n / 256), dim3(256), 0, s, a); }
    };
    for (int r = 0; r < 50; r++) chain(buf[1]);
    CK(hipStreamSynchronize(s));

    // (a) plain launches, the stream drained after every chain (the push's shape: launch, wait, host turn)
    double t_launch = 0, t_total = 0;
    for (int r = 0; r < reps; r++)
    {
        const double t0 = now_us(); chain(r & 1 ? alt : buf[1]); const double t1 = now_us();
        CK(hipStreamSynchronize(s)); const double t2 = now_us();
        t_launch += t1 - t0; t_total += t2 - t0;
    }
    std::printf("plain launches : %6.2f us to launch %d kernels (%.2f each), %6.2f us until the chain has run\n", t_launch / reps, N, t_launch / reps / N, t_total / reps);

    // (b) the same chain captured once, replayed
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal)); chain(buf[1]); CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 50; r++) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    t_launch = t_total = 0;
    for (int r = 0; r < reps; r++)
    {
        const double t0 = now_us(); CK(hipGraphLaunch(ge, s)); const double t1 = now_us();
        CK(hipStreamSynchronize(s)); const double t2 = now_us();
        t_launch += t1 - t0; t_total += t2 - t0;
    }
    std::printf("graph replay   : %6.2f us to launch, %6.2f us until the chain has run\n", t_launch / reps, t_total / reps);

    // (c) two kernel nodes re-pointed before every replay
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
    std::vector<hipGraphNode_t> kn;
    for (auto nd : nodes) { hipGraphNodeType t; CK(hipGraphNodeGetType(nd, &t)); if (t == hipGraphNodeTypeKernel) kn.push_back(nd); }
    if (kn.size() >= 2)
    {
        t_launch = t_total = 0;
        for (int r = 0; r < reps; r++)
        {
            const double t0 = now_us();
            for (int j = 0; j < 2; j++)
            {
                hipKernelNodeParams p{}; CK(hipGraphKernelNodeGetParams(kn[j], &p));
                Args a = *reinterpret_cast<Args*>(p.kernelParams[0]); if (j == 0) a.in = r & 1 ? alt : buf[1];
                void* kp[1] = {&a}; p.kernelParams = kp;
                CK(hipGraphExecKernelNodeSetParams(ge, kn[j], &p));
            }
            CK(hipGraphLaunch(ge, s)); const double t1 = now_us();
            CK(hipStreamSynchronize(s)); const double t2 = now_us();
            t_launch += t1 - t0; t_total += t2 - t0;
        }
        std::printf("graph, 2 nodes re-pointed: %6.2f us to launch, %6.2f us until the chain has run\n", t_launch / reps, t_total / reps);
    }
    // (d) GPU-side length of the chain by events, plain against graph
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    CK(hipEventRecord(e0, s)); for (int r = 0; r < 200; r++) chain(buf[1]); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("back to back, plain: %.2f us per chain on the GPU\n", ms * 1000 / 200);
    CK(hipEventRecord(e0, s)); for (int r = 0; r < 200; r++) CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("back to back, graph: %.2f us per chain on the GPU\n", ms * 1000 / 200);
    // (e) ONE launch onto a stream that is idle at that moment (the bulk stream's situation when the remap is launched), by stream priority and
    //     argument size; the other stream keeps running a chain meanwhile
    {
        int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        hipStream_t s_lo, s_def; CK(hipStreamCreateWithPriority(&s_lo, hipStreamNonBlocking, lo)); CK(hipStreamCreateWithFlags(&s_def, hipStreamNonBlocking));
        struct Big { const float* in; float* out; int n; int pad[45]; };      // ~200 bytes
        auto one = [&](hipStream_t q, const char* what) -> int {
            double t_call = 0; const int R = 500;
            for (int r = 0; r < R; r++)
            {
                chain(buf[1]);                                             // the other stream is busy
                CK(hipStreamSynchronize(q));                               // ... and this one idle
                Args a{}; a.in = alt; a.out = alt; a.n = n;
                const double t0 = now_us(); hipLaunchKernelGGL(k_step, dim3(n / 256), dim3(256), 0, q, a); t_call += now_us() - t0;
                CK(hipStreamSynchronize(s));
            }
            std::printf("one launch onto an idle %s stream: %.2f us\n", what, t_call / R);
            return 0;
        };
        // (f) the same launch when the stream's previous command was an event record that another stream then waited for (what every remap launch follows)
        {
            hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            double t_call = 0, t_rec = 0; const int R = 500;
            for (int r = 0; r < R; r++)
            {
                chain(buf[1]);
                CK(hipStreamSynchronize(s_lo));
                Args a{}; a.in = alt; a.out = alt; a.n = n;
                double t0 = now_us(); hipLaunchKernelGGL(k_step, dim3(n / 256), dim3(256), 0, s_lo, a); t_call += now_us() - t0;
                t0 = now_us(); CK(hipEventRecord(ev, s_lo)); t_rec += now_us() - t0;
                CK(hipStreamWaitEvent(s, ev, 0));
                CK(hipStreamSynchronize(s));
            }
            std::printf("launch after an event record on that stream: %.2f us; the event record itself: %.2f us\n", t_call / R, t_rec / R);
        }
        if (one(s_def, "default-priority")) return 1;
        if (one(s_lo, "lowest-priority")) return 1;
        if (one(s, "(the busy one itself, idle now)")) return 1;
    }
    return 0;
}
