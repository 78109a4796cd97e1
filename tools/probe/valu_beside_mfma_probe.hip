// What does ONE extra instruction cost a wave that issues v_mfma_f32_16x16x32_bf16 back to back, ONE wave per SIMD (the bf16 training
// kernels' regime) -- by instruction type?  Loop body: 4 MFMAs (independent accumulators) + K instructions of one type on registers
// the MFMAs do not touch; cycles per iteration by s_memtime, K = 0, 4, 8.  Also with TWO waves per SIMD (512-thread workgroups).
// hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe tools/probe/valu_beside_mfma_probe.hip && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
using v4 = float __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

typedef unsigned u4v __attribute__((ext_vector_type(4)));
template <int OP> __device__ __forceinline__ void op(float &x, float &y, v2f &p, v2f &q, unsigned &u, unsigned &w, __attribute__((address_space(3))) unsigned *l) {
    if (OP == 13) { u4v t = {u, w, u, w}; asm volatile("ds_write_b128 %0, %1" :: "v"(l), "v"(t) : "memory"); }
    if (OP == 14) { u4v t; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(8)" : "=v"(t) : "v"(l) : "memory"); u ^= t[0]; }
    if (OP == 15) { v2f t; asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(8)" : "=v"(t) : "v"(l) : "memory"); q = t; }
    if (OP == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q));
    if (OP == 2) asm volatile("v_maximum3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    if (OP == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u) : "v"(x), "v"(y));
    if (OP == 5) asm volatile("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(u) : "v"(w), "v"(u));
    if (OP == 6) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(u) : "v"(w), "v"(u));
    if (OP == 7) asm volatile("ds_write_b64 %0, %1" :: "v"(l), "v"(p) : "memory");
    if (OP == 8) asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(8)" : "=v"(q) : "v"(l) : "memory");
    if (OP == 9) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    if (OP == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(u) : "v"(w));
    if (OP == 11) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(p), "v"(q));       // (result to a fresh operand set: same as 1)
    if (OP == 12) asm volatile("s_nop 0");
}
template <int OP, int K, int NM>
__global__ void probe(unsigned long long *out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) unsigned lds[8192];
    const int lane = threadIdx.x & 63;
    bf8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e) + seed); b[e] = (__bf16)(0.002f * (lane - e)); }
    v4 c[4];
    for (int i = 0; i < 4; ++i) c[i] = (v4){0.f, 0.f, 0.f, 0.f};
    float x = seed + lane, y = 0.99f;
    float xs[4] = {x, x + 1, x + 2, x + 3};
    v2f p = {x, y}, q = {0.5f, 0.25f};
    unsigned u = lane, w = 15;
    __attribute__((address_space(3))) unsigned *l = (__attribute__((address_space(3))) unsigned *)lds + 4 * threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c[i & 3]) : "v"(a), "v"(b));
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (OP == 16) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(xs[k & 3]) : "v"(y));
                    else if (OP == 17) asm volatile("v_maximum3_f32 %0, %0, %1, %1" : "+v"(xs[k & 3]) : "v"(y));
                    else op<OP>(x, y, p, q, u, w, l);
                }
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = x + p[0] + q[1] + (float)u + xs[0] + xs[1] + xs[2] + xs[3];
    for (int i = 0; i < 4; ++i) s += c[i][0];
    if (s == 12345.678f) out[1] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int OP, int K, int NM> double run(unsigned long long *d, int threads) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<OP, K, NM>), dim3(256), dim3(threads), 0, 0, d, iters, 0.5f);
    hipLaunchKernelGGL((probe<OP, K, NM>), dim3(256), dim3(threads), 0, 0, d, iters, 0.5f);
    hipDeviceSynchronize();
    unsigned long long t;
    hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
    return (double)t / (iters * 4.0 * NM);          // cycles per MFMA slot
}
const char *names[] = {"v_mul_f32", "v_pk_mul_f32", "v_maximum3_f32", "v_max_f32", "v_cvt_pk_bf16_f32", "v_pk_ashrrev_i16", "v_bfi_b32",
                       "ds_write_b64", "ds_read_b64", "v_fma_f32", "v_mov_b32", "v_pk_mul_f32 (b)", "s_nop 0", "ds_write_b128", "ds_read_b128",
                       "ds_read_b64_tr_b16", "v_mul_f32 x4 indep", "v_max3 x4 indep"};
template <int OP> void row(unsigned long long *d) {
    for (int threads : {256, 512}) {
        const double k0 = run<OP, 0, 4>(d, threads), k1 = run<OP, 1, 4>(d, threads), k2 = run<OP, 2, 4>(d, threads), k4 = run<OP, 4, 4>(d, threads);
        printf("%-20s %d waves/SIMD: cycles per MFMA slot with 0 / 1 / 2 / 4 of them per MFMA: %6.1f %6.1f %6.1f %6.1f   -> %5.1f per instruction (from 4)\n",
               names[OP], threads / 256, k0, k1, k2, k4, (k4 - k0) / 4);
    }
}
int main() {
    unsigned long long *d;
    hipMalloc(&d, 64);
    row<0>(d); row<1>(d); row<2>(d); row<3>(d); row<4>(d); row<5>(d); row<6>(d); row<7>(d); row<8>(d); row<9>(d); row<10>(d); row<12>(d); row<13>(d); row<14>(d); row<15>(d); row<16>(d); row<17>(d);
    return 0;
}
