// Sliced-Wasserstein regulariser of the latent batch (utils.loss_function_swae / compute_swd,
// reference utils.py:27-91; reachable from training.fit through config.custom_loss_function,
// training.py:73-80).  For every random projection s (a unit vector of the latent space) the batch's
// latent codes z and the prior samples are projected, both projections are SORTED ACROSS THE BATCH and
// compared rank by rank:
//     swd = reg_weight * mean_{s,k} (sort(z P_s)_k - sort(prior P_s)_k)^2        (wasserstein_deg = 2)
// The random draws (prior, projections) are INPUTS: the caller owns the generator, exactly as the
// reference leaves them to torch's global RNG.
//   swd_sort_k  one workgroup per projection: project (n x d dot products), bitonic sort in LDS of
//               (value, row) pairs for z and of the values for the prior, per-projection loss partial and
//               the gradient w.r.t. the projected values scattered back to row order: G[s][row].
//   swd_dz_k    dz[row][k] = sum_s G[s][row] * P[s][k] in a fixed order (no atomics: reproducible).
// Batches above 4096 rows (CFD_project_animation_config.py:19: batch_size = 6000) run the same bitonic network in passes
// through global memory (swd_project_k / swd_local_k / swd_global_k / swd_finish_k below).
#include "bamd_internal.hpp"

namespace bamd {
namespace {

template <typename T> __device__ __forceinline__ T pos_inf();
template <> __device__ __forceinline__ float pos_inf<float>() { return __int_as_float(0x7f800000); }
template <> __device__ __forceinline__ double pos_inf<double>() { return __longlong_as_double(0x7ff0000000000000LL); }

template <typename T>
__global__ void __launch_bounds__(256) swd_sort_k(const T *__restrict__ z, const T *__restrict__ prior,
                                                  const T *__restrict__ proj, int n, int d, int m, double scale,
                                                  T *__restrict__ G, double *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *av = (T *)lds_raw;            // m projected latent values
    T *bv = av + m;                  // m projected prior values
    int *ai = (int *)(bv + m);       // row of every latent value
    __shared__ double red[256];
    const int s = blockIdx.x, tid = threadIdx.x;
    const T *P = proj + (int64_t)s * d;
    for (int i = tid; i < m; i += 256) {
        T a = pos_inf<T>(), b = pos_inf<T>();
        if (i < n) {
            a = (T)0; b = (T)0;
            for (int k = 0; k < d; ++k) {
                a += z[(int64_t)i * d + k] * P[k];
                b += prior[(int64_t)i * d + k] * P[k];
            }
        }
        av[i] = a; bv[i] = b; ai[i] = i;
    }
    __syncthreads();
    for (int k = 2; k <= m; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < m / 2; t += 256) {
                const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
                const bool up = (lo & k) == 0;
                {
                    const T x = av[lo], y = av[hi];
                    const int xi = ai[lo], yi = ai[hi];
                    const bool gt = x > y || (x == y && xi > yi);
                    if (gt == up) { av[lo] = y; av[hi] = x; ai[lo] = yi; ai[hi] = xi; }
                }
                {
                    const T x = bv[lo], y = bv[hi];
                    if ((x > y) == up) { bv[lo] = y; bv[hi] = x; }
                }
            }
            __syncthreads();
        }
    double acc = 0.0;
    for (int i = tid; i < n; i += 256) {
        const T w = av[i] - bv[i];
        acc += (double)w * (double)w;
        G[(int64_t)s * n + ai[i]] = (T)(2.0 * scale) * w;
    }
    red[tid] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) part[s] = red[0];
}

// ---- batches that do not fit one workgroup's LDS (n > 4096): the same bitonic network in passes through global memory --------
// Every projection's m = 2^p padded values live in global scratch (value + row for z, value for the prior).  Steps with
// partner distance j < kSwdChunk run inside LDS (one workgroup per chunk of kSwdChunk consecutive elements); steps with
// j >= kSwdChunk are one compare-exchange per thread on global memory.  CFD_project_animation trains at batch_size = 6000.
constexpr int kSwdChunk = 4096;
template <typename T>
__global__ void __launch_bounds__(256) swd_project_k(const T *__restrict__ z, const T *__restrict__ prior, const T *__restrict__ proj, int n,
                                                     int d, int m, T *__restrict__ AV, T *__restrict__ BV, int *__restrict__ AI) {
    const int s = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const T *P = proj + (int64_t)s * d;
    T a = pos_inf<T>(), b = pos_inf<T>();
    if (i < n) {
        a = (T)0; b = (T)0;
        for (int k = 0; k < d; ++k) {
            a += z[(int64_t)i * d + k] * P[k];
            b += prior[(int64_t)i * d + k] * P[k];
        }
    }
    AV[(int64_t)s * m + i] = a; BV[(int64_t)s * m + i] = b; AI[(int64_t)s * m + i] = i;
}
template <typename T> __device__ __forceinline__ void swd_cmpx(T &x, T &y, int &xi, int &yi, bool up) {
    const bool gt = x > y || (x == y && xi > yi);
    if (gt == up) { const T t = x; x = y; y = t; const int ti = xi; xi = yi; yi = ti; }
}
// all steps (k, j) with k in [k_lo, k_hi], j < min(k, chunk): inside LDS, one chunk per workgroup (direction from the GLOBAL index)
template <typename T>
__global__ void __launch_bounds__(256) swd_local_k(T *__restrict__ AV, T *__restrict__ BV, int *__restrict__ AI, int m, int k_lo, int k_hi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T *av = (T *)lds_raw;
    T *bv = av + kSwdChunk;
    int *ai = (int *)(bv + kSwdChunk);
    const int64_t base = (int64_t)blockIdx.y * m + (int64_t)blockIdx.x * kSwdChunk;
    const int g0 = blockIdx.x * kSwdChunk;
    for (int i = threadIdx.x; i < kSwdChunk; i += 256) { av[i] = AV[base + i]; bv[i] = BV[base + i]; ai[i] = AI[base + i]; }
    __syncthreads();
    for (int k = k_lo; k <= k_hi; k <<= 1)
        for (int j = (k >> 1) < kSwdChunk ? (k >> 1) : (kSwdChunk >> 1); j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < kSwdChunk / 2; t += 256) {
                const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
                const bool up = ((g0 + lo) & k) == 0;
                swd_cmpx(av[lo], av[hi], ai[lo], ai[hi], up);
                const T x = bv[lo], y = bv[hi];
                if ((x > y) == up) { bv[lo] = y; bv[hi] = x; }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < kSwdChunk; i += 256) { AV[base + i] = av[i]; BV[base + i] = bv[i]; AI[base + i] = ai[i]; }
}
template <typename T>
__global__ void __launch_bounds__(256) swd_global_k(T *__restrict__ AV, T *__restrict__ BV, int *__restrict__ AI, int m, int k, int j) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= m / 2) return;
    const int64_t base = (int64_t)blockIdx.y * m;
    const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
    const bool up = (lo & k) == 0;
    T x = AV[base + lo], y = AV[base + hi];
    int xi = AI[base + lo], yi = AI[base + hi];
    const bool gt = x > y || (x == y && xi > yi);
    if (gt == up) { AV[base + lo] = y; AV[base + hi] = x; AI[base + lo] = yi; AI[base + hi] = xi; }
    x = BV[base + lo]; y = BV[base + hi];
    if ((x > y) == up) { BV[base + lo] = y; BV[base + hi] = x; }
}
// per projection: loss partial + gradient w.r.t. the projected values, scattered back to row order
template <typename T>
__global__ void __launch_bounds__(256) swd_finish_k(const T *__restrict__ AV, const T *__restrict__ BV, const int *__restrict__ AI, int n, int m,
                                                    double scale, T *__restrict__ G, double *__restrict__ part) {
    __shared__ double red[256];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int64_t base = (int64_t)s * m;
    double acc = 0.0;
    for (int i = tid; i < n; i += 256) {
        const T w = AV[base + i] - BV[base + i];
        acc += (double)w * (double)w;
        G[(int64_t)s * n + AI[base + i]] = (T)(2.0 * scale) * w;
    }
    red[tid] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) part[s] = red[0];
}

template <typename T>
__global__ void __launch_bounds__(256) swd_dz_k(const T *__restrict__ G, const T *__restrict__ proj, int n, int d, int ns,
                                                T *__restrict__ dz) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // e = k * n + row: G reads coalesce over rows
    if (e >= (int64_t)n * d) return;
    const int k = (int)(e / n), row = (int)(e - (int64_t)k * n);
    T acc = (T)0;
    for (int s = 0; s < ns; ++s) acc += G[(int64_t)s * n + row] * proj[(int64_t)s * d + k];
    dz[(int64_t)row * d + k] = acc;
}

__global__ void __launch_bounds__(256) swd_loss_k(const double *__restrict__ part, int ns, double scale, double *__restrict__ out) {
    __shared__ double sh[256];
    const double s = block_sum_fixed(part, ns, sh);
    if (threadIdx.x == 0) *out = s * scale;
}

template <typename T>
int swd_T(const void *z, const void *prior, const void *proj, int n, int d, int ns, double reg_weight, double *loss_out,
          void *dz_out, DevBuf &scratch, hipStream_t s) {
    int m = 2;
    while (m < n) m <<= 1;
    const size_t lds = (size_t)m * (2 * sizeof(T) + sizeof(int));
    const size_t g_bytes = ((size_t)ns * n * sizeof(T) + 255) & ~(size_t)255;
    int rc = scratch.ensure(g_bytes + sizeof(double) * ns);
    if (rc) return rc;
    T *G = (T *)scratch.p;
    double *part = (double *)((char *)scratch.p + g_bytes);
    const double scale = reg_weight / ((double)ns * (double)n);           // reg_weight * mean over (projections x rows)
    if (m > kSwdChunk) {
        // multi-pass: scratch = [G | part | AV | BV | AI]
        const size_t p_bytes = (sizeof(double) * ns + 255) & ~(size_t)255, v_bytes = ((size_t)ns * m * sizeof(T) + 255) & ~(size_t)255;
        rc = scratch.ensure(g_bytes + p_bytes + 2 * v_bytes + (size_t)ns * m * sizeof(int));
        if (rc) return rc;
        G = (T *)scratch.p;
        part = (double *)((char *)scratch.p + g_bytes);
        T *AV = (T *)((char *)scratch.p + g_bytes + p_bytes), *BV = (T *)((char *)AV + v_bytes);
        int *AI = (int *)((char *)BV + v_bytes);
        const size_t clds = (size_t)kSwdChunk * (2 * sizeof(T) + sizeof(int));
        BAMD_HIP(hipFuncSetAttribute((const void *)swd_local_k<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds));
        hipLaunchKernelGGL(swd_project_k<T>, dim3(m / 256, ns), dim3(256), 0, s, (const T *)z, (const T *)prior, (const T *)proj, n, d, m, AV, BV, AI);
        hipLaunchKernelGGL(swd_local_k<T>, dim3(m / kSwdChunk, ns), dim3(256), clds, s, AV, BV, AI, m, 2, kSwdChunk);
        for (int k = 2 * kSwdChunk; k <= m; k <<= 1) {
            for (int j = k >> 1; j >= kSwdChunk; j >>= 1)
                hipLaunchKernelGGL(swd_global_k<T>, dim3(m / 512, ns), dim3(256), 0, s, AV, BV, AI, m, k, j);
            hipLaunchKernelGGL(swd_local_k<T>, dim3(m / kSwdChunk, ns), dim3(256), clds, s, AV, BV, AI, m, k, k);
        }
        hipLaunchKernelGGL(swd_finish_k<T>, dim3(ns), dim3(256), 0, s, (const T *)AV, (const T *)BV, (const int *)AI, n, m, scale, G, part);
    } else {
        BAMD_HIP(hipFuncSetAttribute((const void *)swd_sort_k<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(swd_sort_k<T>, dim3(ns), dim3(256), lds, s, (const T *)z, (const T *)prior, (const T *)proj, n, d, m,
                           scale, G, part);
    }
    hipLaunchKernelGGL(swd_dz_k<T>, dim3((unsigned)(((int64_t)n * d + 255) / 256)), dim3(256), 0, s, (const T *)G,
                       (const T *)proj, n, d, ns, (T *)dz_out);
    hipLaunchKernelGGL(swd_loss_k, dim3(1), dim3(256), 0, s, (const double *)part, ns, scale, loss_out);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

}  // namespace

int launch_swd(const void *z, const void *prior, const void *proj, int dtype, int64_t n, int d, int ns, double reg_weight,
               double *loss_out, void *dz_out, hipStream_t s) {
    BAMD_REQUIRE(z && prior && proj && loss_out && dz_out && d > 0 && ns > 0, "bad arguments");
    BAMD_REQUIRE(n >= 2 && n <= (1 << 20), "the sliced-Wasserstein kernel sorts one batch per projection: 2 <= n_rows <= 1048576");
    DevBuf &scratch = scratch_for(2, s);   // keyed by (device, stream): see elementwise.hip
    if (dtype == BAMD_F64) return swd_T<double>(z, prior, proj, (int)n, d, ns, reg_weight, loss_out, dz_out, scratch, s);
    return swd_T<float>(z, prior, proj, (int)n, d, ns, reg_weight, loss_out, dz_out, scratch, s);
}

}  // namespace bamd
