// HBM WRITE rate: what a store-bound kernel (the bf16 decode of the wide models writes 10,000 B per frame for 100 B read) can reach.
// Contiguous 1-KiB stores per wave instruction, and the decode kernels' pattern (a wave writes 16 rows x 64 B of a 2,500-column float32 table).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_write tools/probe/hbm_write_probe.hip && /tmp/hbm_write
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int F = 2500, ROWB = F * 4, NCH = ROWB / 64;
template <int MODE>
__global__ void __launch_bounds__(256) wr(char *__restrict__ base, int64_t nrows) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (int64_t)gridDim.x * 4;
    const v4 val = {1.f, 2.f, 3.f, (float)lane};
    if constexpr (MODE == 0) {
        const int64_t total = nrows * ROWB / 1024;
        for (int64_t i = wave; i < total; i += nwave) *(v4 *)(base + i * 1024 + lane * 16) = val;
    } else if constexpr (MODE == 1) {
        const int64_t ntile = nrows / 16;
        for (int64_t t = wave; t < ntile; t += nwave) {
            char *tb = base + t * 16 * (int64_t)ROWB + (int64_t)(lane >> 2) * ROWB + (lane & 3) * 16;
            for (int c = 0; c < NCH; ++c) *(v4 *)(tb + (int64_t)c * 64) = val;
        }
    } else {      // MODE = bytes per row and instruction (256: the aligned decode path's windows -- 4 rows x 256 B; 512: 2 rows x 512 B), a wave owns 32 rows
        constexpr int LPR = MODE / 16, RPI = 64 / LPR, NW = ROWB / MODE;
        const int64_t ngrp = nrows / 32;
        for (int64_t t = wave; t < ngrp; t += nwave) {
            char *tb = base + t * 32 * (int64_t)ROWB + (int64_t)(lane / LPR) * ROWB + (lane % LPR) * 16;
            for (int c = 0; c < NW; ++c)
#pragma unroll
                for (int k = 0; k < 32 / RPI; ++k) *(v4 *)(tb + (int64_t)k * RPI * ROWB + (int64_t)c * MODE) = val;
        }
    }
}
template <int MODE> void run(char *buf, int64_t nrows, const char *name, int wgs) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((wr<MODE>), dim3(wgs), dim3(256), 0, 0, buf, nrows);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("%-60s %4d workgroups: %7.1f us per pass = %5.2f TB/s\n", name, wgs, 1e3 * ms / 10, nrows * (double)(MODE == 0 ? ROWB : MODE == 1 ? NCH * 64 : (ROWB / MODE) * MODE) / (ms / 10 * 1e-3) / 1e12);
    }
}
int main() {
    const int64_t nrows = 131072;
    char *buf;
    (void)hipMalloc(&buf, nrows * ROWB + 4096);
    for (int wgs : {512, 1024, 2048}) {
        run<0>(buf, nrows, "contiguous 1-KiB stores", wgs);
        run<1>(buf, nrows, "16 rows x 64 B per store instruction", wgs);
        run<256>(buf, nrows, "4 rows x 256 B (the aligned decode path's windows)", wgs);
        run<512>(buf, nrows, "2 rows x 512 B", wgs);
    }
    return 0;
}
