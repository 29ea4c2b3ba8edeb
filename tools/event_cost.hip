// Microbenchmark: what timing a kernel costs the stream -- hipEventRecord before and after it (barrier packets) against
// start / stop events attached to the dispatch itself (hipExtLaunchKernelGGL).
// Build: hipcc --offload-arch=gfx950 -O3 tools/event_cost.hip -o tools/event_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(unsigned* out, int iters) {
    unsigned x = threadIdx.x;
    for (int i = 0; i < iters; ++i) x = x * 1664525u + 1013904223u;
    if (x == 12345u) out[0] = x;
}

int main() {
    const int n = 2000, iters = 2000;
    unsigned* d;
    hipMalloc(&d, 4);
    std::vector<hipEvent_t> ev(2 * n + 2);
    for (auto& e : ev) hipEventCreate(&e);
    hipStream_t s;
    hipStreamCreate(&s);
    auto run = [&](int mode) {
        hipStreamSynchronize(s);
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) {
            if (mode == 0) {
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, d, iters);
            } else if (mode == 1) {  // one event per boundary
                hipEventRecord(ev[i], s);
                hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, d, iters);
            } else {
                hipExtLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, ev[2 * i], ev[2 * i + 1], 0, d, iters);
            }
        }
        if (mode == 1) hipEventRecord(ev[n], s);
        hipStreamSynchronize(s);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        double sum = 0;
        if (mode == 1) for (int i = 0; i < n; ++i) { float ms; hipEventElapsedTime(&ms, ev[i], ev[i + 1]); sum += ms; }
        if (mode == 2) for (int i = 0; i < n; ++i) { float ms; hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]); sum += ms; }
        printf("mode %d: %.2f us per kernel (wall), event-measured %.2f us per kernel\n", mode, us / n, sum * 1e3 / n);
    };
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 3; ++m) run(m);
    return 0;
}
