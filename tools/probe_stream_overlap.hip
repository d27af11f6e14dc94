// Do two HIP streams really run their kernels concurrently on this box?  Stream A: n short single-wave kernels back to back;
// stream B: the same.  Concurrent queues -> wall time ~ one stream's; serialised / time-sliced queues -> the sum.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_stream_overlap.hip -o /tmp/probe_so && /tmp/probe_so
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float *out, int iters)
{
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.678f) out[0] = v;
}
static double run(int nA, int gA, int itA, int nB, int gB, int itB, bool two)
{
    static hipStream_t sa = nullptr, sb = nullptr;
    static float *buf = nullptr;
    if (!sa) { (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking); (void)hipMalloc(&buf, 1024); }
    hipEvent_t e0, e1, eb;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&eb);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, sa);
    (void)hipStreamWaitEvent(sb, e0, 0);
    for (int i = 0; i < (nA > nB ? nA : nB); ++i) {
        if (i < nA) hipLaunchKernelGGL(spin, dim3(gA), dim3(256), 0, sa, buf, itA);
        if (i < nB) hipLaunchKernelGGL(spin, dim3(gB), dim3(256), 0, two ? sb : sa, buf, itB);
    }
    (void)hipEventRecord(eb, sb);
    (void)hipStreamWaitEvent(sa, eb, 0);
    (void)hipEventRecord(e1, sa);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
// The engine's pattern: stream A runs a step's kernels and records an event; stream B waits for it and runs 80 short kernels
// (the registration); stream A goes straight on with the next step.  Does A's next step start under B?
static void pattern(int itA)
{
    hipStream_t sa, sb;
    (void)hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    float *buf; (void)hipMalloc(&buf, 1024);
    hipEvent_t e0, ready, a_done, b_done;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&ready); (void)hipEventCreate(&a_done); (void)hipEventCreate(&b_done);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, sa);
        hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, sa, buf, itA);          // "matcher(k)"
        (void)hipEventRecord(ready, sa);
        (void)hipStreamWaitEvent(sb, ready, 0);
        for (int i = 0; i < 80; ++i) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, sb, buf, 1500);     // "registration(k)"
        (void)hipEventRecord(b_done, sb);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, sa, buf, 1500);     // "matcher(k+1)" small kernels
        hipLaunchKernelGGL(spin, dim3(2048), dim3(256), 0, sa, buf, itA);                                 // ... and its big one
        (void)hipEventRecord(a_done, sa);
        (void)hipEventSynchronize(a_done); (void)hipEventSynchronize(b_done);
        float ta, tb, tr;
        (void)hipEventElapsedTime(&tr, e0, ready); (void)hipEventElapsedTime(&ta, e0, a_done); (void)hipEventElapsedTime(&tb, e0, b_done);
        printf("pattern: big kernel done at %.2f ms; stream B (80 short kernels) done at %.2f ms; stream A's next step done at %.2f ms\n", tr, tb, ta);
    }
}
int main()
{
    pattern(60000);
    for (int rep = 0; rep < 2; ++rep) {
        printf("A: 100 x (1 WG, ~20 us)   B: 100 x (1 WG, ~20 us):   one stream %.2f ms, two streams %.2f ms\n", run(100, 1, 20000, 100, 1, 20000, false),
               run(100, 1, 20000, 100, 1, 20000, true));
        printf("A: 2 x (2048 WGs, long)    B: 100 x (64 WGs, ~20 us): one stream %.2f ms, two streams %.2f ms\n", run(2, 2048, 400000, 100, 64, 20000, false),
               run(2, 2048, 400000, 100, 64, 20000, true));
    }
    return 0;
}
