// Is the bf16 inference chain's VALU work (LeakyReLU + conversion, ~2.65 VALU per 16x16x32 MFMA) better hidden behind
// v_mfma_f32_32x32x16_bf16 (holds the SIMD's vector issue for 8 of its 32 cycles) than behind v_mfma_f32_16x16x32_bf16 (8 of 16)?
// Equal FLOPs and equal VALU work per iteration: 4 x 16x16x32 vs 2 x 32x32x16, NV VALU epilogue values; 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_probe tools/probe/mfma_shape_probe.hip && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
using v4 = float __attribute__((ext_vector_type(4)));
using v16 = float __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float lrelu(float a) { return __builtin_amdgcn_fmed3f(a, 0.01f * a, 3.402823466e38f); }
template <int SHAPE, int NV>      // NV = activation values (per lane) finished per iteration: 2 VALU + 0.5 cvt each
__global__ void __launch_bounds__(512) probe(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    bf8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * (lane - e)); }
    v4 c4[4]; v16 c16[2];
    for (int i = 0; i < 4; ++i) c4[i] = (v4){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) c16[i][r] = 0.f;
    float vals[16];
    for (int i = 0; i < 16; ++i) vals[i] = 0.01f * (lane + i);
    unsigned sink = 0;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(a), "+v"(b));
        if (SHAPE == 16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) c4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c4[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) c16[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c16[i], 0, 0, 0);
        }
        // epilogue work of a PREVIOUS tile (independent of the MFMAs above): LeakyReLU + pairwise bf16 conversion
#pragma unroll
        for (int i = 0; i < NV; i += 2) {
            const float x0 = lrelu(vals[i & 15] + (float)it), x1 = lrelu(vals[(i + 1) & 15] - (float)it);
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            const bf2 pk = {(__bf16)x0, (__bf16)x1};
            sink ^= __builtin_bit_cast(unsigned, pk);
        }
    }
    float s = (float)sink;
    for (int i = 0; i < 4; ++i) s += c4[i][0] + c4[i][3];
    for (int i = 0; i < 2; ++i) s += c16[i][0] + c16[i][15];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int SHAPE, int NV> float run(float *d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<SHAPE, NV>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<SHAPE, NV>), dim3(256), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float *d; hipMalloc(&d, 256 * 512 * 4);
    const int it = 20000;
    const double flop = 256.0 * 8 * it * 4 * 16384;
#define ROW(NV) { float a = run<16, NV>(d, it), b = run<32, NV>(d, it); \
    printf("%2d values per 4 (2) MFMAs = %.2f VALU per 16x16x32: 16x16x32 %.3f ms (%.0f TF)   32x32x16 %.3f ms (%.0f TF)   ratio %.2f\n", NV, 2.5 * NV / 4, a, flop / a / 1e9, b, flop / b / 1e9, a / b); }
    ROW(0) ROW(2) ROW(4) ROW(6) ROW(8) ROW(10) ROW(12)
    return 0;
}
