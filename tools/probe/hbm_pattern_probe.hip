// How does the HBM read rate of a row-panel walk depend on the contiguous run fetched per row?
// A workgroup (256 threads) owns 128 rows of a row-major float32 table (row = F floats, C4: 2500) and walks along them;
// every wave-instruction loads 16 bytes per lane, 64 lanes cover RPI rows x RUN contiguous bytes (RPI * RUN = 1024):
//   RUN =   64: 16 rows x 64 B  (the MFMA-operand-shaped loads of wide_bf16_encode_*: lane (i, g) -> 16 B of row i)
//   RUN =  128:  8 rows x 128 B (one full cache line per row)
//   RUN =  256 / 512 / 1024: 4 / 2 / 1 rows per instruction
// Each wave keeps DEPTH instructions in flight.  Build: hipcc --offload-arch=gfx950 -O3 hbm_pattern_probe.hip -o hbm_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4 __attribute__((ext_vector_type(4)));
template <int RUN, int DEPTH, bool MFMA_SHAPED = false>
__global__ void __launch_bounds__(256) walk(const float *__restrict__ x, int64_t n, int F, float *__restrict__ out) {
    constexpr int RPI = 1024 / RUN;                 // rows per instruction
    constexpr int LPR = RUN / 16;                   // lanes per row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ngroup = n / 128;
    const int rowbytes = F * 4;
    const int steps_per_row = rowbytes / RUN;       // full runs only
    v4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
        // the wave's 32 rows, RPI at a time; per position along the row: 32 / RPI instructions
        const char *base = (const char *)x + (grp * 128 + wave * 32) * (int64_t)rowbytes;
        const int total = steps_per_row * (32 / RPI);
        v4 buf[DEPTH];
        auto addr = [&](int k) {
            const int s = k / (32 / RPI), rb = k % (32 / RPI);
            // MFMA_SHAPED: lane (i, g) = (lane & 15, lane >> 4) -> 16 bytes g of row i: the same 16 rows x 64 B per instruction, but
            // ADJACENT LANES IN DIFFERENT ROWS (what a B operand of v_mfma_f32_16x16x32 loaded straight from a row-major table looks like)
            const int row = MFMA_SHAPED ? rb * RPI + (lane & 15) : rb * RPI + lane / LPR;
            const int piece = MFMA_SHAPED ? (lane >> 4) : lane % LPR;
            return (const v4 *)(base + (int64_t)row * rowbytes + s * RUN + piece * 16);
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) buf[d] = *addr(d);
        for (int k = 0; k < total; k += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                acc += buf[d];
                const int kn = k + d + DEPTH;
                buf[d] = *addr(kn < total ? kn : d);
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += buf[d];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}
template <int RUN, int DEPTH, bool M = false> void run(const float *x, int64_t n, int F, float *out, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((walk<RUN, DEPTH, M>), dim3(grid), dim3(256), 0, 0, x, n, F, out);
    hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((walk<RUN, DEPTH, M>), dim3(grid), dim3(256), 0, 0, x, n, F, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double bytes = (double)n * (F * 4 / RUN) * RUN;
    printf("run %5d B/row%s, %2d loads in flight per wave, grid %4d: %.3f ms  %.2f TB/s\n", RUN, M ? " (lane = row + 16 piece)" : "", DEPTH, grid, ms, bytes / ms / 1e9);
}
int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 131072;
    const int F = argc > 2 ? atoi(argv[2]) : 2500;
    float *x, *out;
    hipMalloc(&x, n * F * 4); hipMalloc(&out, 4096);
    hipMemset(x, 0, n * F * 4);
    for (int grid : {256, 512, 1024}) {
        run<64, 8, true>(x, n, F, out, grid); run<64, 16, true>(x, n, F, out, grid);
        run<64, 8>(x, n, F, out, grid); run<64, 16>(x, n, F, out, grid);
        run<128, 8>(x, n, F, out, grid); run<128, 16>(x, n, F, out, grid);
        run<256, 8>(x, n, F, out, grid); run<256, 16>(x, n, F, out, grid);
        run<512, 8>(x, n, F, out, grid); run<512, 16>(x, n, F, out, grid);
        run<1024, 8>(x, n, F, out, grid); run<1024, 16>(x, n, F, out, grid);
    }
    return 0;
}
