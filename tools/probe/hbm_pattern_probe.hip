// HBM read rate against the ADDRESS PATTERN of a wave's 1-KiB load instruction, on a table of float32 rows of F = 2500 columns (10,000 B per
// row, C4's frames): what the bf16 wide encode's row loaders request (16 rows x 64 B per instruction, the second half of each 128-B line by
// the next instruction) against contiguous streams and against 2 rows x 512 B / 4 rows x 256 B per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_pattern tools/probe/hbm_pattern_probe.hip && /tmp/hbm_pattern
// Every pattern reads each byte of the table exactly once per pass; 512 workgroups x 256 threads, 8 loads in flight per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int F = 2500, ROWB = F * 4;
// MODE 0: contiguous 1 KiB per instruction over the whole table.
// MODE p > 0: an instruction covers R = 1024 / P rows x P bytes (P = 64, 128, 256, 512, 1024): lane l -> row l / (P / 16), byte 16 (l % (P / 16)).
// A wave walks a 16-row tile chunk after chunk (P bytes of each row per step, in instructions of R rows), tile after tile.
template <int P>
__global__ void __launch_bounds__(256) rd(const char *__restrict__ base, int64_t nrows, float *out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (int64_t)gridDim.x * 4;
    v4 acc = {0.f, 0.f, 0.f, 0.f};
    if (P == 0) {
        const int64_t total = nrows * ROWB / 1024;      // 1-KiB pieces
        for (int64_t i = wave; i + 7 * nwave < total; i += 8 * nwave) {
            v4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = *(const v4 *)(base + (i + u * nwave) * 1024 + lane * 16);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += t[u];
        }
    } else {
        constexpr int PP = P > 0 ? P : 64, LPR = PP / 16, R = 64 / LPR;      // lanes per row, rows per instruction
        const int64_t ntile = nrows / 16;
        constexpr int NCH = ROWB / PP;                   // whole chunks per row (the tail of a row is skipped: < 3 % of the bytes)
        for (int64_t tile = wave; tile < ntile; tile += nwave) {
            const char *tb = base + tile * 16 * (int64_t)ROWB + (int64_t)(lane / LPR) * ROWB + (lane % LPR) * 16;
            for (int c = 0; c + 8 / (16 / R) <= NCH; c += 8 / (16 / R) > 0 ? 8 / (16 / R) : 1) {
                v4 t[8];
                // 16 / R instructions cover the tile's 16 rows for one chunk; 8 loads in flight = 8 / (16 / R) chunks (at least one)
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int ch = c + u / (16 / R), sub = u % (16 / R);
                    t[u] = *(const v4 *)(tb + (int64_t)sub * R * ROWB + (int64_t)ch * PP);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += t[u];
            }
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = acc[0];
}
template <int P> void run(const char *buf, int64_t nrows, float *out, const char *name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(rd<P>, dim3(512), dim3(256), 0, 0, buf, nrows, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("%-44s %7.1f us per pass = %5.2f TB/s\n", name, 1e3 * ms / 10, nrows * (double)ROWB / (ms / 10 * 1e-3) / 1e12);
    }
}
int main() {
    const int64_t nrows = 131072;      // 1.31 GB
    char *buf; float *out;
    hipMalloc(&buf, nrows * ROWB + 4096); hipMemset(buf, 0, nrows * ROWB + 4096); hipMalloc(&out, 64);
    run<0>(buf, nrows, out, "contiguous, 1 KiB per instruction");
    run<1024>(buf, nrows, out, "1 row x 1,024 B per instruction");
    run<512>(buf, nrows, out, "2 rows x 512 B");
    run<256>(buf, nrows, out, "4 rows x 256 B");
    run<128>(buf, nrows, out, "8 rows x 128 B");
    run<64>(buf, nrows, out, "16 rows x 64 B (the encode kernel's loaders)");
    return 0;
}
