// HBM read rate of C4's row table (float32 rows of 2,500 columns) when a wave requests 16 rows x 64 B per instruction (the wide bf16 encode's
// pattern) against HOW MANY CONSECUTIVE 64-B PIECES OF THE SAME ROWS it has in flight: K consecutive chunks of T row tiles at a time
// (K x T = 8 loads in flight per lane).  K = 8, T = 1 is tools/probe/hbm_pattern_probe.hip's walk (1 KiB of a row requested back to back);
// the encode kernels have K = 2..3 chunks of their rows in flight and come back to the same rows a chunk time (1 - 2 us) later.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_loc tools/probe/hbm_locality_probe.hip && /tmp/hbm_loc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v4 __attribute__((ext_vector_type(4)));
constexpr int F = 2500, ROWB = F * 4, NCH = ROWB / 64;      // 156 whole 64-B pieces per row
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
// M = bf16 MFMAs (16 cycles each) issued per batch of 8 loads: does MFMA activity beside the stream change what HBM delivers?
template <int K, int T, int M = 0>
__global__ void __launch_bounds__(256) rd(const char *__restrict__ base, int64_t nrows, float *out) {
    v4 macc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    bf8 ma, mb;
    for (int i = 0; i < 8; ++i) { ma[i] = (__bf16)(float)(threadIdx.x + i); mb[i] = (__bf16)1.0f; }
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (int64_t)gridDim.x * 4;
    const int64_t ntile = nrows / 16;
    v4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t t0 = wave * T; t0 + T <= ntile; t0 += nwave * T) {
        const char *tb = base + t0 * 16 * (int64_t)ROWB + (int64_t)(lane >> 2) * ROWB + (lane & 3) * 16;
        for (int c = 0; c + K <= NCH; c += K) {
            v4 x[T][K];
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int k = 0; k < K; ++k) x[t][k] = *(const v4 *)(tb + (int64_t)t * 16 * ROWB + (int64_t)(c + k) * 64);
#pragma unroll
            for (int m = 0; m < M; ++m) macc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ma, mb, macc[m & 3], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int k = 0; k < K; ++k) acc += x[t][k];
        }
    }
    if (M > 0) acc += macc[0] + macc[1] + macc[2] + macc[3];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = acc[0];
}
template <int K, int T, int M = 0> void run(const char *buf, int64_t nrows, float *out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL((rd<K, T, M>), dim3(512), dim3(256), 0, 0, buf, nrows, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("K = %d consecutive 64-B pieces of T = %d row tiles in flight, %2d bf16 MFMAs per 8 loads: %7.1f us per pass = %5.2f TB/s\n", K, T, M, 1e3 * ms / 10,
                             nrows * (double)(NCH * 64) / (ms / 10 * 1e-3) / 1e12);
    }
}
int main() {
    const int64_t nrows = 131072;
    char *buf; float *out;
    (void)hipMalloc(&buf, nrows * ROWB + 4096); (void)hipMemset(buf, 0, nrows * ROWB + 4096); (void)hipMalloc(&out, 64);
    run<8, 1>(buf, nrows, out);
    run<4, 2>(buf, nrows, out);
    run<2, 4>(buf, nrows, out);
    run<1, 8>(buf, nrows, out);
    run<2, 4, 2>(buf, nrows, out);
    run<2, 4, 8>(buf, nrows, out);
    run<2, 4, 26>(buf, nrows, out);
    run<8, 1, 8>(buf, nrows, out);
    run<8, 1, 26>(buf, nrows, out);
    return 0;
}
