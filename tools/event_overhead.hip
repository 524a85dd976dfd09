// What does timing every launch cost?  N launches of a ~150 us kernel back to back on one stream: (a) no events, (b) hipEventRecord
// before and after every launch (two marker packets per launch), (c) hipExtLaunchKernelGGL with start / stop events (the dispatch's
// own timestamps, no marker packets).  Prints the wall time of the N launches and the mean kernel time each method reports.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o tools/event_overhead tools/event_overhead.hip && tools/event_overhead
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void spin(float *out, int iters)
{
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) out[0] = v;
}

int main()
{
    float *out;
    hipMalloc(&out, 4);
    const int N = 4000, iters = 6000;
    std::vector<hipEvent_t> ev(2 * N);
    for (auto &e : ev) hipEventCreate(&e);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) {
                if (mode == 1) hipEventRecord(ev[2 * i], 0);
                if (mode == 2) hipExtLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, 0, ev[2 * i], ev[2 * i + 1], 0, out, iters);
                else hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, 0, out, iters);
                if (mode == 1) hipEventRecord(ev[2 * i + 1], 0);
            }
            hipDeviceSynchronize();
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            double sum = 0.0;
            int bad = 0;
            if (mode) {
                for (int i = 0; i < N; ++i) {
                    float e = 0.f;
                    if (hipEventElapsedTime(&e, ev[2 * i], ev[2 * i + 1]) != hipSuccess) ++bad;
                    sum += e;
                }
            }
            printf("mode %d (%s): %d launches in %.2f ms = %.2f us per launch; events report %.2f us per kernel (%d unreadable)\n", mode,
                   mode == 0 ? "no events" : mode == 1 ? "hipEventRecord pairs" : "hipExtLaunchKernelGGL events", N, ms, ms * 1e3 / N, sum * 1e3 / N, bad);
        }
    }
    return 0;
}
