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

__global__ void swd_loss_k(const double *__restrict__ part, int ns, double scale, double *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < ns; ++i) s += part[i];
        *out = s * scale;
    }
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
    BAMD_HIP(hipFuncSetAttribute((const void *)swd_sort_k<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(swd_sort_k<T>, dim3(ns), dim3(256), lds, s, (const T *)z, (const T *)prior, (const T *)proj, n, d, m,
                       scale, G, part);
    hipLaunchKernelGGL(swd_dz_k<T>, dim3((unsigned)(((int64_t)n * d + 255) / 256)), dim3(256), 0, s, (const T *)G,
                       (const T *)proj, n, d, ns, (T *)dz_out);
    hipLaunchKernelGGL(swd_loss_k, dim3(1), dim3(64), 0, s, (const double *)part, ns, scale, loss_out);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

}  // namespace

int launch_swd(const void *z, const void *prior, const void *proj, int dtype, int64_t n, int d, int ns, double reg_weight,
               double *loss_out, void *dz_out, hipStream_t s) {
    BAMD_REQUIRE(z && prior && proj && loss_out && dz_out && d > 0 && ns > 0, "bad arguments");
    BAMD_REQUIRE(n >= 2 && n <= 4096, "the sliced-Wasserstein kernel sorts one batch in LDS: 2 <= n_rows <= 4096");
    DevBuf &scratch = scratch_for(2, s);   // keyed by (device, stream): see elementwise.hip
    if (dtype == BAMD_F64) return swd_T<double>(z, prior, proj, (int)n, d, ns, reg_weight, loss_out, dz_out, scratch, s);
    return swd_T<float>(z, prior, proj, (int)n, d, ns, reg_weight, loss_out, dz_out, scratch, s);
}

}  // namespace bamd
