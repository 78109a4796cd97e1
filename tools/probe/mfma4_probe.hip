// Probe of v_mfma_f32_4x4x1_16B_f32 on gfx950: operand / result lane maps (exact integer data) and issue rate at one wave per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma4_probe tools/probe/mfma4_probe.hip && /tmp/mfma4_probe
// Candidate for the small-batch chain: 16 independent 4x4 outer products per instruction = 64 output features x 4 batch rows x 1 k.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using v4 = float __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4 mfma4(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ v4 mfma16(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__global__ void layout(float *out) {
    const int lane = threadIdx.x;
    // A = 1000 + lane, B = lane: D[r] of lane l = A(lane of (block, i=r)) * B(lane l) if the guessed map holds
    v4 d = mfma4(1000.f + lane, (float)(lane + 1), (v4){0.f, 0.f, 0.f, 0.f});
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}
template <int KIND>
__global__ void __launch_bounds__(256) rate(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    v4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4){0.f, 0.f, 0.f, 0.f};
    float a = 1.f + lane, b = 0.5f * lane;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(a), "+v"(b));
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (KIND == 0) acc[k & 7] = mfma4(a, b, acc[k & 7]);
            if (KIND == 1) acc[k & 1] = mfma4(a, b, acc[k & 1]);
            if (KIND == 2) acc[0] = mfma4(a, b, acc[0]);
            if (KIND == 3) acc[k & 3] = mfma16(a, b, acc[k & 3]);
        }
    }
    v4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main() {
    float *d;
    hipMalloc(&d, 1 << 22);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d);
    std::vector<float> h(256);
    hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const float want = (1000.f + 4 * (l / 4) + r) * (l + 1);   // D[i = r][j = l % 4] of block l / 4 = A[i] * B[j]
            if (h[l * 4 + r] != want) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, r, h[l * 4 + r], want); ++bad; }
        }
    printf("layout guess (A: lane 4b+i, B: lane 4b+j, D: lane 4b+j reg i): %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    const char *names[] = {"4x4x1_16B, 8 accumulators", "4x4x1_16B, 2 accumulators", "4x4x1_16B, 1 accumulator (dependent)", "16x16x4, 4 accumulators"};
    const double flop[] = {512, 512, 512, 2048};
    for (int kind = 0; kind < 4; ++kind) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 2000;
        auto launch = [&]() {
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, d, iters);
            if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, d, iters);
            if (kind == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256), 0, 0, d, iters);
            if (kind == 3) hipLaunchKernelGGL(rate<3>, dim3(256), dim3(256), 0, 0, d, iters);
        };
        launch();
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = 256.0 * 4 * iters * 64;      // wave-instructions
        printf("%-40s %.3f ms  %.1f TFLOP/s  (%.1f cycles per instruction and SIMD at 2.4 GHz)\n", names[kind], ms, n * flop[kind] / ms / 1e9,
               ms * 1e-3 * 2.4e9 / (iters * 64.0));
    }
    return 0;
}
