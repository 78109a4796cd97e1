// How long do back-to-back DEPENDENT launches on one stream take when the kernels do (almost) nothing?  The floor under any multi-launch
// optimiser step:  hipcc -O3 --offload-arch=gfx950 tools/probe/launch_gap_probe.hip -o tools/probe/launch_gap_probe.out
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void empty_k(int *p) { if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 1; }
__global__ void touch_k(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
int main() {
    float *buf; hipMalloc(&buf, 1 << 22);
    hipStream_t s; hipStreamCreate(&s);
    for (int grid : {1, 128, 298, 1024}) {
        for (int kind = 0; kind < 2; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                const int n = 4000;
                hipStreamSynchronize(s);
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < n; ++i) {
                    if (kind == 0) hipLaunchKernelGGL(empty_k, dim3(grid), dim3(256), 0, s, (int *)nullptr);
                    else hipLaunchKernelGGL(touch_k, dim3(grid), dim3(256), 0, s, buf, grid * 256);
                }
                hipStreamSynchronize(s);
                double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
                if (rep) printf("grid %4d x 256, %s kernel: %.2f us per launch (back to back, one stream)\n", grid, kind ? "touch (4 B per thread, RMW)" : "empty", us);
            }
        }
    }
    // the same inside a captured graph (one graph = 200 launches)
    for (int grid : {128, 298}) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(touch_k, dim3(grid), dim3(256), 0, s, buf, grid * 256);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (20 * 200);
        printf("grid %4d x 256, touch kernel in a hipGraph of 200: %.2f us per launch\n", grid, us);
    }
    return 0;
}
