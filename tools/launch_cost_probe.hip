// Probe: host cost of a kernel launch right after the stream has gone idle against in a busy stream (why does the first sweep after a
// synchronisation cost the host 190 - 290 us instead of 100?).  Build: hipcc --offload-arch=gfx950 -O3 -o launch_cost_probe launch_cost_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
struct Big { double v[32]; };
__global__ void k_small(double* p, Big b) { if (threadIdx.x == 0 && b.v[0] == 12345.0) p[0] = b.v[1]; }
__global__ void k_busy(double* p, int iters) { double x = threadIdx.x; for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-9; if (x == 12345.0) p[0] = x; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    double* d; hipMalloc(&d, 64);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    Big b{}; 
    for (int mode = 0; mode < 3; ++mode) {      // 0: stream idle before the burst; 1: a 2 ms kernel running on the stream; 2: idle, but polled (hipStreamQuery) instead of a blocking sync
        std::vector<double> first, rest;
        for (int rep = 0; rep < 20; ++rep) {
            if (mode == 2) { while (hipStreamQuery(s) == hipErrorNotReady) {} } else hipStreamSynchronize(s);
            if (mode == 1) hipLaunchKernelGGL(k_busy, dim3(256), dim3(256), 0, s, d, 200000);
            double t[41]; t[0] = now();
            for (int i = 0; i < 40; ++i) { hipLaunchKernelGGL(k_small, dim3(8), dim3(64), 0, s, d, b); t[i + 1] = now(); }
            for (int i = 0; i < 40; ++i) (i < 10 ? first : rest).push_back(t[i + 1] - t[i]);
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("%-52s launches 1-10: median %.2f us   launches 11-40: median %.2f us\n", mode == 0 ? "stream idle (blocking synchronize before)" : mode == 1 ? "a long kernel running on the stream" : "stream idle (polled with hipStreamQuery before)", med(first), med(rest));
    }
    hipStreamSynchronize(s);
    return 0;
}
