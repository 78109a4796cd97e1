// What a CU's vector-memory path delivers when a row stream from HBM shares it with an L2-resident fragment stream, and whether
// direct-to-LDS loads (buffer_load_dwordx4 ... lds) stream as fast as register loads.  Sizing probe for the bf16 encode of the wide
// models (wide_bf16_encode_*: 16 KB of float32 rows + 13 KB of weight fragments per 128 rows and 32 features).
//   workgroup = W waves; per "chunk" every wave loads XI 1-KiB pieces of the row table (MFMA-shaped: 16 rows x 64 B per instruction)
//   and FI 1-KiB pieces of a 1-MB fragment table (all waves the same pieces: L2 hits after the first touch);
//   MODE 0: register loads, 1: direct-to-LDS loads (inline asm, hand-counted vmcnt);  DEPTH chunks in flight per wave.
// Build: hipcc --offload-arch=gfx950 -O3 hbm_stream_mix_probe.hip -o hbm_stream_mix_probe.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma(unsigned lds_addr, int voff, __amdgpu_buffer_rsrc_t rs, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int XI, int FI, int DEPTH, int MODE>
__global__ void __launch_bounds__(512) k(const float *__restrict__ x, int64_t nrows, int F, const float *__restrict__ frag, float *__restrict__ out,
                                         int chunks_per_group) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = blockDim.x >> 6;
    const int rows_per_wave = 16 * (XI / 2);                 // XI pieces of 16 rows x 64 B = 16 rows x 128 B per pair
    const int64_t ngroup = nrows / (rows_per_wave * W);
    const __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc((void *)frag, 0, 1 << 20, 0x00020000);
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds + wave * (DEPTH + 1) * (XI + FI) * 1024;
    v4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int64_t grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + (grp * W + wave) * rows_per_wave * (int64_t)F), 0, 0x7fffffff, 0x00020000);
        const int xo = (lane & 15) * F * 4 + 16 * (lane >> 4);
        v4 buf[MODE == 0 ? DEPTH : 1][MODE == 0 ? XI + FI : 1];
        auto issue = [&](int c, int slot) {
#pragma unroll
            for (int i = 0; i < XI; ++i) {
                const int soff = (i >> 1) * 16 * F * 4 + c * 128 + (i & 1) * 64;
                if (MODE == 0) buf[MODE == 0 ? slot : 0][MODE == 0 ? i : 0] = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xo, soff, 0));
                else dma(lbase + (slot * (XI + FI) + i) * 1024, xo, xrs, soff);
            }
#pragma unroll
            for (int i = 0; i < FI; ++i) {
                const int soff = ((c * 13 + i) & 1023) * 1024;
                if (MODE == 0) buf[MODE == 0 ? slot : 0][MODE == 0 ? XI + i : 0] = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(frs, lane * 16, soff, 0));
                else dma(lbase + (slot * (XI + FI) + XI + i) * 1024, lane * 16, frs, soff);
            }
        };
        if (MODE == 0) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) issue(d, d);
            for (int c = 0; c < chunks_per_group; c += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
                    for (int i = 0; i < XI + FI; ++i) acc += buf[d][i];
                    issue(c + d + DEPTH < chunks_per_group ? c + d + DEPTH : d, d);
                }
            }
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
#pragma unroll
                for (int i = 0; i < XI + FI; ++i) acc += buf[d][i];
        } else {
            for (int d = 0; d < DEPTH; ++d) issue(d, d);
            int slot = 0;
            for (int c = 0; c < chunks_per_group; ++c) {
                // all but the DEPTH - 1 youngest chunks have landed
                if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "i"((XI + FI) * 1) : "memory");
                if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "i"((XI + FI) * 2) : "memory");
                if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(%0)" :: "i"((XI + FI) * 3) : "memory");
                if (DEPTH == 6) asm volatile("s_waitcnt vmcnt(%0)" :: "i"((XI + FI) * 5) : "memory");
                acc += *(const v4 *)(lds + wave * (DEPTH + 1) * (XI + FI) * 1024 + slot * (XI + FI) * 1024 + lane * 16);
                const int ns = slot == 0 ? DEPTH : slot - 1;      // ring of DEPTH + 1 slots: refill the one consumed LAST iteration
                (void)ns;
                issue(c + DEPTH < chunks_per_group ? c + DEPTH : c, (slot + DEPTH) % (DEPTH + 1));
                slot = (slot + 1) % (DEPTH + 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[threadIdx.x] = acc[0];
}
template <int XI, int FI, int DEPTH, int MODE> void run(const float *x, int64_t n, int F, const float *frag, float *out, int W) {
    const int cpg = F * 4 / 128;
    auto fn = k<XI, FI, DEPTH, MODE>;
    const int ldsb = MODE ? W * (DEPTH + 1) * (XI + FI) * 1024 : 0;
    if (ldsb > 160 * 1024) { printf("XI %d FI %d depth %d mode %d W %d: LDS %d > 160 KiB, skipped\n", XI, FI, DEPTH, MODE, W, ldsb); return; }
    (void)hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(fn, dim3(256), dim3(64 * W), ldsb, 0, x, n, F, frag, out, cpg);
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(fn, dim3(256), dim3(64 * W), ldsb, 0, x, n, F, frag, out, cpg);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const int rpw = 16 * (XI / 2);
    const double rows = (double)(n / (rpw * W)) * rpw * W;
    printf("%s, %d waves/CU, per wave and chunk %d KiB of rows + %2d KiB of fragments, %d chunks ahead: %.3f ms  rows %.2f TB/s (+ fragments %.2f TB/s)\n",
           MODE ? "direct-to-LDS" : "register loads", W, XI, FI, DEPTH, ms, rows * cpg * 128 / ms / 1e9, rows / rpw * cpg * FI * 1024.0 / ms / 1e9);
}
int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 131072;
    const int F = 2500;
    float *x, *frag, *out;
    (void)hipMalloc(&x, n * F * 4); (void)hipMalloc(&frag, 2 << 20); (void)hipMalloc(&out, 4096);
    (void)hipMemset(x, 0, n * F * 4); (void)hipMemset(frag, 0, 2 << 20);
    for (int W : {2, 4, 8}) {
        run<4, 0, 3, 0>(x, n, F, frag, out, W);  run<4, 0, 6, 0>(x, n, F, frag, out, W);
        run<4, 4, 3, 0>(x, n, F, frag, out, W);  run<4, 2, 3, 0>(x, n, F, frag, out, W);  run<4, 13, 2, 0>(x, n, F, frag, out, W);
        run<4, 0, 3, 1>(x, n, F, frag, out, W);  run<4, 0, 6, 1>(x, n, F, frag, out, W);
        run<4, 4, 3, 1>(x, n, F, frag, out, W);  run<4, 2, 3, 1>(x, n, F, frag, out, W);
    }
    return 0;
}
