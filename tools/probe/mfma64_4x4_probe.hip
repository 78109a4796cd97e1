// Probe of v_mfma_f64_4x4x4_4b_f64 on gfx950: operand / result lane maps found EMPIRICALLY (one-hot A lane x one-hot B lane -> which
// D lane lights up) and the issue rate at one wave per SIMD, beside v_mfma_f64_16x16x4_f64.
// hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma64_4x4_probe tools/probe/mfma64_4x4_probe.hip && /tmp/mfma64_4x4_probe
// Candidate for the fp64 small-batch chain (fused64.hip): 4 independent 4x4x4 products per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using d4 = double __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ d4 mfma16(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// block (la, lb): A = 1 on lane la only, B = 1 on lane lb only; out[la][lb] = bit mask of D lanes that are non-zero
__global__ void onehot(unsigned long long *out) {
    const int lane = threadIdx.x, la = blockIdx.x, lb = blockIdx.y;
    const double d = mfma4(lane == la ? 1.0 : 0.0, lane == lb ? 1.0 : 0.0, 0.0);
    const unsigned long long m = __ballot(d != 0.0);
    if (lane == 0) out[la * 64 + lb] = m;
}
template <int KIND>
__global__ void __launch_bounds__(256) rate(double *out, int iters) {
    const int lane = threadIdx.x & 63;
    double acc[8];
    d4 acc16[4];
    for (int i = 0; i < 8; ++i) acc[i] = 0.0;
    for (int i = 0; i < 4; ++i) acc16[i] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + lane, b = 0.5 * lane;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(a), "+v"(b));
#pragma unroll
        for (int k = 0; k < 64; ++k) {
            if (KIND == 0) acc[k & 7] = mfma4(a, b, acc[k & 7]);
            if (KIND == 1) acc[k & 1] = mfma4(a, b, acc[k & 1]);
            if (KIND == 2) acc[0] = mfma4(a, b, acc[0]);
            if (KIND == 3) acc16[k & 3] = mfma16(a, b, acc16[k & 3]);
            if (KIND == 4) acc[k & 3] = mfma4(a, b, acc[k & 3]);
        }
    }
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    for (int i = 0; i < 4; ++i) s += acc16[i][0] + acc16[i][1] + acc16[i][2] + acc16[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    unsigned long long *dm;
    double *d;
    hipMalloc(&dm, 64 * 64 * 8);
    hipMalloc(&d, 1 << 22);
    hipLaunchKernelGGL(onehot, dim3(64, 64), dim3(64), 0, 0, dm);
    std::vector<unsigned long long> m(4096);
    hipMemcpy(m.data(), dm, 4096 * 8, hipMemcpyDeviceToHost);
    // hypothesis: A lane = 16 k + 4 b + i ... print the raw table compactly instead: for every A lane, the B lanes it meets and the D lane
    int singles = 0, multi = 0;
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            const unsigned long long v = m[la * 64 + lb];
            if (!v) continue;
            if (__builtin_popcountll(v) == 1) { printf(" B%d->D%d", lb, __builtin_ctzll(v)); ++singles; }
            else { printf(" B%d->mask%llx", lb, v); ++multi; }
        }
        printf("\n");
    }
    printf("non-zero (A lane, B lane) pairs: %d with one D lane, %d with several\n", singles, multi);
    // check the guess: A lane = 16 k + 4 b... evaluated by three candidate maps
    struct Map { const char *name; int (*a)(int, int, int); int (*b)(int, int, int); int (*dd)(int, int, int); };
    const Map maps[] = {
        {"A: lane 16k+4b+i  B: lane 16k+4b+j  D: lane 16i+4b+j", [](int b, int i, int k) { return 16 * k + 4 * b + i; }, [](int b, int k, int j) { return 16 * k + 4 * b + j; }, [](int b, int i, int j) { return 16 * i + 4 * b + j; }},
        {"A: lane 16b+4k+i  B: lane 16b+4k+j  D: lane 16b+4i+j", [](int b, int i, int k) { return 16 * b + 4 * k + i; }, [](int b, int k, int j) { return 16 * b + 4 * k + j; }, [](int b, int i, int j) { return 16 * b + 4 * i + j; }},
        {"A: lane 16k+4b+i  B: lane 16k+4b+j  D: lane 16b+4i+j... (mixed)", [](int b, int i, int k) { return 16 * k + 4 * b + i; }, [](int b, int k, int j) { return 16 * k + 4 * b + j; }, [](int b, int i, int j) { return 16 * b + 4 * i + j; }},
        {"A: lane 4k+i+16b  B: lane 4k+j+16b  D: lane 16b+4j+i", [](int b, int i, int k) { return 16 * b + 4 * k + i; }, [](int b, int k, int j) { return 16 * b + 4 * k + j; }, [](int b, int i, int j) { return 16 * b + 4 * j + i; }},
        {"A: lane 16k+4b+i  B: lane 16k+4b+j  D: lane 16j+4b+i", [](int b, int i, int k) { return 16 * k + 4 * b + i; }, [](int b, int k, int j) { return 16 * k + 4 * b + j; }, [](int b, int i, int j) { return 16 * j + 4 * b + i; }},
    };
    for (const Map &mp : maps) {
        int bad = 0, cnt = 0;
        std::vector<unsigned long long> want(4096, 0);
        for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int k = 0; k < 4; ++k)
            want[mp.a(b, i, k) * 64 + mp.b(b, k, j)] |= 1ull << mp.dd(b, i, j);
        for (int q = 0; q < 4096; ++q) { bad += want[q] != m[q]; cnt += m[q] != 0; }
        printf("map [%s]: %s (%d mismatching pairs of %d live)\n", mp.name, bad ? "WRONG" : "ok", bad, cnt);
    }
    const char *names[] = {"f64 4x4x4_4b, 8 accumulators", "f64 4x4x4_4b, 2 accumulators", "f64 4x4x4_4b, 1 accumulator (dependent)", "f64 16x16x4, 4 accumulators", "f64 4x4x4_4b, 4 accumulators"};
    const double flop[] = {512, 512, 512, 2048, 512};
    for (int kind = 0; kind < 5; ++kind) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 1000;
        auto launch = [&]() {
            if (kind == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, d, iters);
            if (kind == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, d, iters);
            if (kind == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256), 0, 0, d, iters);
            if (kind == 3) hipLaunchKernelGGL(rate<3>, dim3(256), dim3(256), 0, 0, d, iters);
            if (kind == 4) hipLaunchKernelGGL(rate<4>, dim3(256), dim3(256), 0, 0, d, iters);
        };
        launch();
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = 256.0 * 4 * iters * 64;      // wave-instructions
        printf("%-42s %.3f ms  %.1f TFLOP/s  (%.1f cycles per instruction and SIMD at 2.4 GHz)\n", names[kind], ms, n * flop[kind] / ms / 1e9,
               ms * 1e-3 * 2.4e9 / (iters * 64.0));
    }
    return 0;
}
