// Micro-probe: what sustains the v_mfma_f32_16x16x4_f32 issue rate on gfx950 at one wave per SIMD?
// hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using v4 = float __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4 mfma(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// VAR 0: 4 independent accumulators, register operands
// VAR 1: 2 accumulators alternating
// VAR 2: 2 accumulators, A operand from an 8-deep buffer-load ring (1 KB fragment per 4 MFMAs), B in registers
// VAR 3: VAR 2 + B operand rotated through 13 register tiles (as the chain does)
// VAR 4: VAR 1 + 2 VALU ops per 4 MFMAs
// VAR 5: 1 accumulator (dependent chain)
template <int VAR>
__global__ void __launch_bounds__(256) probe(const v4 *w, float *out, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63;
    v4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4){0.f, 0.f, 0.f, 0.f};
    v4 bt[13];
    for (int i = 0; i < 13; ++i) bt[i] = (v4){1.f + lane, 2.f, 3.f, 4.f + i};
    float a0 = 1.0f + lane, b0 = 2.0f, x = 0.5f * lane;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f pk = (v2f){1.0f, 1.0f};
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)w, 0, 1 << 20, 0x00020000);
    int voff = lane * 16;
    v4 ring[8];
    if (VAR == 2 || VAR == 3)
        for (int i = 0; i < 8; ++i) ring[i] = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, i * 1024, 0));
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(voff));
        if (VAR == 0) {
#pragma unroll
            for (int k = 0; k < 64; ++k) acc[k & 3] = mfma(a0, b0, acc[k & 3]);
        } else if (VAR == 1) {
#pragma unroll
            for (int k = 0; k < 64; ++k) acc[k & 1] = mfma(a0, b0, acc[k & 1]);
        } else if (VAR == 5) {
#pragma unroll
            for (int k = 0; k < 64; ++k) acc[0] = mfma(a0, b0, acc[0]);
        } else if (VAR >= 6 && VAR <= 13) {
            // cost of one extra instruction of a given kind per MFMA (2 accumulators, register operands)
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                acc[k & 1] = mfma(a0, b0, acc[k & 1]);
                if (VAR == 6) asm volatile("s_mov_b32 s40, 0x1234" ::: "s40");
                if (VAR == 7) asm volatile("s_waitcnt vmcnt(7)");
                if (VAR == 8) asm volatile("v_mov_b32 %0, %0" : "+v"(x));
                if (VAR == 9) asm volatile("s_nop 0");
                if (VAR == 10) {   // one independent 16-byte LDS read per MFMA (result never waited for inside the loop body)
                    v4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(voff)); asm volatile("" :: "v"(t));
                }
                if (VAR == 11) {   // one independent 16-byte buffer load per MFMA (L1-resident address)
                    v4 t = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (k & 7) * 1024, 0));
                    asm volatile("" :: "v"(t));
                }
                if (VAR == 12) asm volatile("ds_write_b32 %0, %1" :: "v"(voff), "v"(x));
                if (VAR == 13) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(pk));
            }
        } else if (VAR == 4) {
#pragma unroll
            for (int k = 0; k < 64; ++k) {
                acc[k & 1] = mfma(a0, b0, acc[k & 1]);
                if ((k & 3) == 3) { x = x * 0.01f; x = fmaxf(x, 1.0f); }
            }
        } else {
#pragma unroll
            for (int f = 0; f < 16; f += 2) {   // 16 fragments = 64 MFMAs
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float bA = VAR == 3 ? bt[(f) % 13][r] : b0, bB = VAR == 3 ? bt[(f + 1) % 13][r] : b0;
                    acc[0] = mfma(ring[f % 8][r], bA, acc[0]);
                    acc[1] = mfma(ring[(f + 1) % 8][r], bB, acc[1]);
                }
                ring[f % 8] = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, ((f + 8) % 512) * 1024, 0));
                ring[(f + 1) % 8] = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, ((f + 9) % 512) * 1024, 0));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    v4 s = acc[0] + acc[1] + acc[2] + acc[3];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + x + pk[0];
}

template <int VAR> void run(const v4 *w, float *out, const char *name, int lds) {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute((const void *)probe<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<VAR>, dim3(grid), dim3(256), lds, 0, w, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<VAR>, dim3(grid), dim3(256), lds, 0, w, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfmas = (double)grid * 4 * iters * 64;
    double tf = mfmas * 2048.0 / (ms * 1e-3) / 1e12;
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 64);
    printf("%-52s %8.3f ms  %7.1f TFLOP/s  (%5.1f%% of 157.3)  ~%.1f cycles/MFMA @2.4GHz\n", name, ms, tf, 100 * tf / 157.3, cyc);
}

int main() {
    v4 *w; float *out;
    hipMalloc(&w, 1 << 20); hipMemset(w, 0, 1 << 20); hipMalloc(&out, 256 * 256 * 4);
    const int lds = 100000;   // one workgroup per CU = one wave per SIMD
    run<0>(w, out, "4 independent accumulators, register operands", lds);
    run<1>(w, out, "2 alternating accumulators", lds);
    run<5>(w, out, "1 accumulator (dependent chain)", lds);
    run<4>(w, out, "2 accumulators + 2 VALU per 4 MFMAs", lds);
    run<2>(w, out, "2 accumulators, A from 8-deep buffer-load ring", lds);
    run<3>(w, out, "ring + B rotating over 13 register tiles", lds);
    run<6>(w, out, "2 acc + 1 s_mov_b32 per MFMA", lds);
    run<7>(w, out, "2 acc + 1 s_waitcnt per MFMA", lds);
    run<8>(w, out, "2 acc + 1 v_mov_b32 per MFMA", lds);
    run<9>(w, out, "2 acc + 1 s_nop 0 per MFMA", lds);
    run<10>(w, out, "2 acc + 1 ds_read_b128 per MFMA", lds);
    run<11>(w, out, "2 acc + 1 buffer_load_dwordx4 per MFMA", lds);
    run<12>(w, out, "2 acc + 1 ds_write_b32 per MFMA", lds);
    run<13>(w, out, "2 acc + 1 v_pk_mul_f32 per MFMA", lds);
    return 0;
}
