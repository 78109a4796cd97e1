// How fast does a row-panel walk STORE?  A workgroup owns 128 rows of a row-major float32 table (C4: 2500 columns, 10,000-byte rows
// -- not a multiple of 128), every wave its 32 rows, walking along them; a wave-instruction stores 16 bytes per lane, 64 lanes cover
// RPI rows x RUN contiguous bytes (RPI * RUN = 1024): RUN = 64 (a C tile as it stands: 16 rows x 64 B), 256 (the decode kernel's
// staged segments), 512, 1024.  NT: non-temporal stores.  Build: hipcc --offload-arch=gfx950 -O3 hbm_store_pattern_probe.hip -o ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4 __attribute__((ext_vector_type(4)));
template <int RUN, bool NT>
__global__ void __launch_bounds__(256) walk(float *__restrict__ x, int64_t n, int F) {
    constexpr int RPI = 1024 / RUN, LPR = RUN / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ngroup = n / 128;
    const int rowbytes = F * 4, steps = rowbytes / RUN;
    const v4 val = {1.f, 2.f, 3.f, (float)lane};
    for (int64_t grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
        char *base = (char *)x + (grp * 128 + wave * 32) * (int64_t)rowbytes;
        for (int s = 0; s < steps; ++s)
#pragma unroll
            for (int rb = 0; rb < 32 / RPI; ++rb) {
                v4 *p = (v4 *)(base + (int64_t)(rb * RPI + lane / LPR) * rowbytes + s * RUN + (lane % LPR) * 16);
                if (NT) __builtin_nontemporal_store(val, p); else *p = val;
            }
    }
}
template <int RUN, bool NT> void run(float *x, int64_t n, int F, int grid) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((walk<RUN, NT>), dim3(grid), dim3(256), 0, 0, x, n, F);
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((walk<RUN, NT>), dim3(grid), dim3(256), 0, 0, x, n, F);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("store run %5d B/row%s, grid %4d, F %d: %.3f ms  %.2f TB/s\n", RUN, NT ? " nt" : "   ", grid, F, ms, (double)n * (F * 4 / RUN) * RUN / ms / 1e9);
}
int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 131072;
    float *x;
    (void)hipMalloc(&x, n * 2560 * 4);
    for (int F : {2500, 2560})
        for (int grid : {256, 1024}) {
            run<64, false>(x, n, F, grid); run<256, false>(x, n, F, grid); run<512, false>(x, n, F, grid); run<1024, false>(x, n, F, grid);
            run<256, true>(x, n, F, grid); run<1024, true>(x, n, F, grid);
        }
    return 0;
}
