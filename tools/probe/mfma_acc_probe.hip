// Micro-probe: issue rate of v_mfma_f32_16x16x4_f32 with the accumulators in AGPRs vs VGPRs, by number of independent accumulators.
// hipcc --offload-arch=gfx950 -O3 -o mfma_acc_probe mfma_acc_probe.hip && ./mfma_acc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using v4 = float __attribute__((ext_vector_type(4)));

template <int NACC, bool AGPR, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) probe(float *out, int iters) {
    v4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4){0.f, 0.f, 0.f, 0.f};
    float a[4], b[13];
    for (int i = 0; i < 4; ++i) a[i] = 1.0f + threadIdx.x + i;
    for (int i = 0; i < 13; ++i) b[i] = 0.5f * i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NACC; ++k) {
            if (AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[k]) : "v"(a[k & 3]), "v"(b[k % 13]));
            else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a[k & 3]), "v"(b[k % 13]));
        }
    }
    v4 s = (v4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC, bool AGPR, int WAVES>
void run(const char *name, int grid) {
    float *out;
    hipMalloc(&out, sizeof(float) * grid * 64 * WAVES);
    const int iters = 4096 * 16 / NACC;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<NACC, AGPR, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, out, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<NACC, AGPR, WAVES>), dim3(grid), dim3(64 * WAVES), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * NACC;                      // MFMAs per wave
    const double tf = mf * 2048.0 * grid * WAVES / (ms * 1e-3) / 1e12;
    printf("%-34s grid %4d waves/WG %d: %8.3f ms  %6.1f ns/MFMA/wave  %6.1f TFLOP/s\n", name, grid, WAVES, ms, ms * 1e6 / mf, tf);
    hipFree(out);
}

int main() {
    run<4, false, 4>("4 acc VGPR", 256);
    run<4, true, 4>("4 acc AGPR", 256);
    run<13, false, 4>("13 acc VGPR", 256);
    run<13, true, 4>("13 acc AGPR", 256);
    run<26, false, 4>("26 acc VGPR", 256);
    run<26, true, 4>("26 acc AGPR", 256);
    run<52, false, 4>("52 acc VGPR", 256);
    run<52, true, 4>("52 acc AGPR", 256);
    run<52, true, 4>("52 acc AGPR", 250);
    run<52, true, 4>("52 acc AGPR", 512);
    run<26, true, 4>("26 acc AGPR", 512);
    run<4, false, 4>("4 acc VGPR", 512);
    run<4, false, 4>("4 acc VGPR", 1024);
    run<4, false, 1>("4 acc VGPR 1 wave/WG", 256);
    run<4, false, 1>("4 acc VGPR 1 wave/WG", 1024);
    return 0;
}
