// HBM-bound helper kernels either side of the autoencoder: column min/max, (un)normalisation,
// dtype conversion, the per-row EMD metric and the fused Adam step.  gfx950 only.
#include "bamd_internal.hpp"

#include <map>
#include <mutex>
#include <tuple>

namespace bamd {

template <typename T>
__device__ __forceinline__ double ld(const void *p, int64_t i) {
    return (double)reinterpret_cast<const T *>(p)[i];
}

// ---- column min/max -----------------------------------------------------------------------------
// data_processing.find_minmax (data_processing.py:113-130).  Rows are contiguous, so a workgroup
// reads R = 256/tcols whole rows per pass (one coalesced segment) and every thread keeps the running
// min/max of ONE column.  Algorithmic traffic: n*c*sizeof(T) bytes read once.
// np.min / np.max PROPAGATE a NaN (data_processing.py:124-125 calls them per column): a column with a NaN cell has min = max =
// range = NaN in the reference, while `v < mn ? v : mn` would silently skip it.  The streaming loop keeps a per-thread "saw a
// NaN" flag (one unordered compare per element) and poisons its running pair at the end; the combines below use the sticky form.
__device__ __forceinline__ void mm_merge(double a, double b, double &mn, double &mx) {
    mn = (a < mn || a != a) ? a : mn;      // once mn is a NaN no comparison is true: it stays
    mx = (b > mx || b != b) ? b : mx;
}

template <typename T>
__global__ void __launch_bounds__(256) minmax_partial(const T *__restrict__ x, int64_t n, int c,
                                                      int tcols, double *__restrict__ part) {
    const int R = 256 / tcols;
    const int lc = threadIdx.x % tcols, lr = threadIdx.x / tcols;
    const int col = blockIdx.y * tcols + lc;
    double mn = INFINITY, mx = -INFINITY;
    bool nan = false;
    if (lr < R && col < c) {
        const int64_t step = (int64_t)gridDim.x * R;
        int64_t r = (int64_t)blockIdx.x * R + lr;
        for (; r + 7 * step < n; r += 8 * step) {      // 8 independent loads in flight per thread (HBM latency ~2 us)
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (double)x[(r + u * step) * c + col];
#pragma unroll
            for (int u = 0; u < 8; ++u) { mn = v[u] < mn ? v[u] : mn; mx = v[u] > mx ? v[u] : mx; nan |= v[u] != v[u]; }
        }
        for (; r < n; r += step) {
            double v = (double)x[r * c + col];
            mn = v < mn ? v : mn;
            mx = v > mx ? v : mx;
            nan |= v != v;
        }
    }
    if (nan) mn = mx = __builtin_nan("");
    __shared__ double smn[256], smx[256];
    smn[threadIdx.x] = mn;
    smx[threadIdx.x] = mx;
    __syncthreads();
    if (lr == 0 && col < c) {
        for (int k = 1; k < R; ++k) mm_merge(smn[k * tcols + lc], smx[k * tcols + lc], mn, mx);
        part[((int64_t)blockIdx.x * 2 + 0) * c + col] = mn;
        part[((int64_t)blockIdx.x * 2 + 1) * c + col] = mx;
    }
}

// one workgroup per column: 256 threads stride over the block partials, LDS tree (min / max are exact in any order; a NaN
// partial wins every merge)
__global__ void __launch_bounds__(256) minmax_final(const double *__restrict__ part, int nblk, int c,
                                                    double *__restrict__ features, int raw) {
    __shared__ double smn[256], smx[256];
    const int col = blockIdx.x;
    double mn = INFINITY, mx = -INFINITY;
    for (int b = threadIdx.x; b < nblk; b += 256) mm_merge(part[((int64_t)b * 2 + 0) * c + col], part[((int64_t)b * 2 + 1) * c + col], mn, mx);
    smn[threadIdx.x] = mn; smx[threadIdx.x] = mx;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            double a = smn[threadIdx.x], b = smx[threadIdx.x];
            mm_merge(smn[threadIdx.x + st], smx[threadIdx.x + st], a, b);
            smn[threadIdx.x] = a; smx[threadIdx.x] = b;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        features[col] = smn[0];
        features[c + col] = raw ? smx[0] : smx[0] - smn[0];   // raw: [min ; max] for a cross-rank min / max reduction
    }
}

// Handle-free kernels keep grow-only scratch buffers keyed by (purpose, device, STREAM): two streams never share a
// partials buffer (stream B's partial kernel would overwrite what stream A's final kernel is still reading), and a
// buffer is only ever re-used -- or, on growth, freed -- behind work of its own stream (hipFree waits for the device).
DevBuf &scratch_for(int purpose, hipStream_t s) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, hipStream_t>, DevBuf> bufs;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    return bufs[std::make_tuple(purpose, dev, s)];
}

int launch_minmax(const void *x, int dtype, int64_t n, int c, double *features, hipStream_t s, bool raw) {
    BAMD_REQUIRE(x && features && n > 0 && c > 0, "bad arguments");
    const int tcols = c < 256 ? c : 256;
    const int R = 256 / tcols;
    int64_t want = (n + R - 1) / R;
    int nblk = (int)(want < 1024 ? want : 1024);
    int ncb = (c + tcols - 1) / tcols;
    DevBuf &scratch = scratch_for(0, s);
    int rc = scratch.ensure((size_t)nblk * 2 * c * sizeof(double));
    if (rc) return rc;
    double *part = (double *)scratch.p;
    dim3 grid(nblk, ncb);
    if (dtype == BAMD_F64)
        hipLaunchKernelGGL(minmax_partial<double>, grid, dim3(256), 0, s, (const double *)x, n, c,
                           tcols, part);
    else
        hipLaunchKernelGGL(minmax_partial<float>, grid, dim3(256), 0, s, (const float *)x, n, c,
                           tcols, part);
    hipLaunchKernelGGL(minmax_final, dim3(c), dim3(256), 0, s, part, nblk, c, features, raw ? 1 : 0);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

// ---- normalise / renormalise / convert ----------------------------------------------------------
template <typename TI, typename TO>
__global__ void normalize_k(const TI *__restrict__ x, int64_t count, int c,
                            const double *__restrict__ f, TO *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        int col = (int)(i % c);
        // (x - min) / (max - min) in float64, exactly as data_processing.py:147-152
        out[i] = (TO)(((double)x[i] - f[col]) / f[c + col]);
    }
}

template <typename TI>
__global__ void renormalize_k(const TI *__restrict__ x, int64_t count, int c,
                              const double *__restrict__ f, const uint8_t *__restrict__ mask,
                              double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        int col = (int)(i % c);
        // norm*range + min with two roundings (numpy, data_processing.py:203): no FMA contraction
        double v = __dadd_rn(__dmul_rn((double)x[i], f[c + col]), f[col]);
        if (mask && mask[col]) v = trunc(v);  // astype(int) written back to float64 (baler.py:431)
        out[i] = v;
    }
}

template <typename TI, typename TO>
__global__ void convert_k(const TI *__restrict__ x, TO *__restrict__ out, int64_t count) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (TO)x[i];
}

static inline int ew_grid(int64_t count) {
    int64_t b = (count + 255) / 256;
    return (int)(b < 2048 ? (b < 1 ? 1 : b) : 2048);
}

int launch_normalize(const void *x, int dtype, int64_t n, int c, const double *f, void *out,
                     int out_dtype, hipStream_t s) {
    BAMD_REQUIRE(x && f && out && n >= 0 && c > 0, "bad arguments");
    int64_t count = n * c;
    if (count == 0) return BAMD_OK;
    dim3 g(ew_grid(count)), b(256);
    if (dtype == BAMD_F64 && out_dtype == BAMD_F64)
        hipLaunchKernelGGL((normalize_k<double, double>), g, b, 0, s, (const double *)x, count, c, f, (double *)out);
    else if (dtype == BAMD_F64)
        hipLaunchKernelGGL((normalize_k<double, float>), g, b, 0, s, (const double *)x, count, c, f, (float *)out);
    else if (out_dtype == BAMD_F64)
        hipLaunchKernelGGL((normalize_k<float, double>), g, b, 0, s, (const float *)x, count, c, f, (double *)out);
    else
        hipLaunchKernelGGL((normalize_k<float, float>), g, b, 0, s, (const float *)x, count, c, f, (float *)out);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

int launch_renormalize(const void *x, int dtype, int64_t n, int c, const double *f,
                       const uint8_t *mask, double *out, hipStream_t s) {
    BAMD_REQUIRE(x && f && out && n >= 0 && c > 0, "bad arguments");
    int64_t count = n * c;
    if (count == 0) return BAMD_OK;
    dim3 g(ew_grid(count)), b(256);
    if (dtype == BAMD_F64)
        hipLaunchKernelGGL(renormalize_k<double>, g, b, 0, s, (const double *)x, count, c, f, mask, out);
    else
        hipLaunchKernelGGL(renormalize_k<float>, g, b, 0, s, (const float *)x, count, c, f, mask, out);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

int launch_convert(const void *src, int sd, void *dst, int dd, int64_t count, hipStream_t s) {
    if (count == 0) return BAMD_OK;
    dim3 g(ew_grid(count)), b(256);
    if (sd == BAMD_F64 && dd == BAMD_F64)
        hipLaunchKernelGGL((convert_k<double, double>), g, b, 0, s, (const double *)src, (double *)dst, count);
    else if (sd == BAMD_F64)
        hipLaunchKernelGGL((convert_k<double, float>), g, b, 0, s, (const double *)src, (float *)dst, count);
    else if (dd == BAMD_F64)
        hipLaunchKernelGGL((convert_k<float, double>), g, b, 0, s, (const float *)src, (double *)dst, count);
    else
        hipLaunchKernelGGL((convert_k<float, float>), g, b, 0, s, (const float *)src, (float *)dst, count);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

// ---- per-row EMD (utils.py:112-119) -------------------------------------------------------------
// wasserstein_distance of two equal-size, unit-weight samples = mean |sorted(a) - sorted(b)|.
// One thread per row, insertion sort of <= 64 values; two-stage fixed-order sum (deterministic).
template <typename T>
__global__ void __launch_bounds__(256) emd_partial(const T *__restrict__ x, const T *__restrict__ r,
                                                   int64_t n, int c, double *__restrict__ part) {
    double acc = 0.0;
    for (int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; row < n;
         row += (int64_t)gridDim.x * blockDim.x) {
        double a[64], b[64];
        for (int k = 0; k < c; ++k) {
            double va = (double)x[row * c + k], vb = (double)r[row * c + k];
            int i = k;
            while (i > 0 && a[i - 1] > va) { a[i] = a[i - 1]; --i; }
            a[i] = va;
            i = k;
            while (i > 0 && b[i - 1] > vb) { b[i] = b[i - 1]; --i; }
            b[i] = vb;
        }
        double sacc = 0.0;
        for (int k = 0; k < c; ++k) sacc += fabs(a[k] - b[k]);
        acc += sacc / (double)c;
    }
    __shared__ double sh[256];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

__global__ void sum_partials_k(const double *__restrict__ part, int n, double *__restrict__ out,
                               double scale, int accumulate) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double sacc = 0.0;
        for (int i = 0; i < n; ++i) sacc += part[i];
        sacc *= scale;
        *out = accumulate ? *out + sacc : sacc;
    }
}


int launch_emd_rows(const void *x, const void *r, int dtype, int64_t n, int c, double *out,
                    hipStream_t s) {
    BAMD_REQUIRE(x && r && out && n > 0 && c > 0 && c <= 64, "bad arguments (n_cols must be <= 64)");
    int nblk = (int)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
    DevBuf &scratch = scratch_for(1, s);
    int rc = scratch.ensure(sizeof(double) * nblk);
    if (rc) return rc;
    double *part = (double *)scratch.p;
    if (dtype == BAMD_F64)
        hipLaunchKernelGGL(emd_partial<double>, dim3(nblk), dim3(256), 0, s, (const double *)x, (const double *)r, n, c, part);
    else
        hipLaunchKernelGGL(emd_partial<float>, dim3(nblk), dim3(256), 0, s, (const float *)x, (const float *)r, n, c, part);
    hipLaunchKernelGGL(sum_partials_k, dim3(1), dim3(64), 0, s, part, nblk, out, 1.0, 0);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

// ---- error-bounded deltas (helper.py:442-470, 708-718) -----------------------------------------------
// numpy semantics, element by element, in the arrays' own dtype T: err = (recon - x) / x * 100, +-inf -> 0, a NaN
// never exceeds the bound; delta = float16(recon) - float16(x) rounded to float16 (np.subtract(..., dtype=float16)
// casts BOTH operands first).  double -> float16 must round ONCE: go through float with round-to-odd (truncate and
// set the sticky bit), then the hardware's float -> half round-to-nearest-even.
__device__ __forceinline__ _Float16 to_f16(float f) { return (_Float16)f; }
__device__ __forceinline__ _Float16 to_f16(double d) {
    float f = __double2float_rz(d);
    if ((double)f != d && isfinite(f)) f = __uint_as_float(__float_as_uint(f) | 1u);
    return (_Float16)f;
}
template <typename T>
__global__ void __launch_bounds__(256) error_deltas_k(const T *__restrict__ x, const T *__restrict__ recon, int64_t n, T bound,
                                                      uint8_t *__restrict__ flags, _Float16 *__restrict__ deltas) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const T v = x[i], d = recon[i];
        T err = (d - v) / v * (T)100;
        if (isinf(err)) err = (T)0;
        flags[i] = fabs(err) > bound ? 1 : 0;
        deltas[i] = to_f16(d) - to_f16(v);
    }
}
template <typename T>
__global__ void __launch_bounds__(256) apply_deltas_k(T *__restrict__ out, int n_cols, const int64_t *__restrict__ rows,
                                                      const int32_t *__restrict__ cols, const _Float16 *__restrict__ deltas,
                                                      int64_t count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[rows[i] * n_cols + cols[i]] -= (T)deltas[i];
}

int launch_error_deltas(const void *x, const void *recon, int dtype, int64_t n, double bound, uint8_t *flags, void *deltas,
                        hipStream_t s) {
    BAMD_REQUIRE(n >= 0 && (n == 0 || (x && recon && flags && deltas)), "bad arguments");
    if (n == 0) return BAMD_OK;
    const int grid = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    if (dtype == BAMD_F64)
        hipLaunchKernelGGL(error_deltas_k<double>, dim3(grid), dim3(256), 0, s, (const double *)x, (const double *)recon, n,
                           bound, flags, (_Float16 *)deltas);
    else
        hipLaunchKernelGGL(error_deltas_k<float>, dim3(grid), dim3(256), 0, s, (const float *)x, (const float *)recon, n,
                           (float)bound, flags, (_Float16 *)deltas);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

int launch_apply_deltas(void *out, int dtype, int n_cols, const int64_t *rows, const int32_t *cols, const void *deltas,
                        int64_t count, hipStream_t s) {
    BAMD_REQUIRE(count >= 0 && n_cols > 0 && (count == 0 || (out && rows && cols && deltas)), "bad arguments");
    if (count == 0) return BAMD_OK;
    const dim3 grid((unsigned)((count + 255) / 256));
    if (dtype == BAMD_F64)
        hipLaunchKernelGGL(apply_deltas_k<double>, grid, dim3(256), 0, s, (double *)out, n_cols, rows, cols,
                           (const _Float16 *)deltas, count);
    else
        hipLaunchKernelGGL(apply_deltas_k<float>, grid, dim3(256), 0, s, (float *)out, n_cols, rows, cols,
                           (const _Float16 *)deltas, count);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

// ---- fused Adam ---------------------------------------------------------------------------------
// torch.optim.Adam single-tensor step (training.py:266; torch/optim/adam.py _single_tensor_adam) over
// ONE flat buffer: p, g, m, v are read once and p, m, v written once (28 B/param in fp32).  The
// per-element arithmetic runs in float64 and is rounded to the storage type once, so the fp32 mode
// differs from the fp64 reference by storage rounding only.  zero_grad needs no work: the next
// bamd_fwd_bwd overwrites the gradient buffer.
template <typename T>
__global__ void __launch_bounds__(256) adam_k(T *__restrict__ p, T *__restrict__ pcopy,
                                              const T *__restrict__ g, T *__restrict__ m,
                                              T *__restrict__ v, int64_t np, double b1, double b2,
                                              double eps, double step_size, double bc2_sqrt,
                                              double *loss_accum, const int *__restrict__ sc_off,
                                              const int *__restrict__ sc_idx, T *__restrict__ packed) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) {
        double gi = (double)g[i];
        double mi = (double)m[i], vi = (double)v[i];
        mi = mi + (gi - mi) * (1.0 - b1);          // exp_avg.lerp_(grad, 1 - beta1)
        vi = vi * b2 + (1.0 - b2) * gi * gi;       // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
        double denom = sqrt(vi) / bc2_sqrt + eps;
        double pi = (double)p[i] - step_size * (mi / denom);
        m[i] = (T)mi;
        v[i] = (T)vi;
        p[i] = (T)pi;
        if (pcopy) pcopy[i] = (T)pi;
        if (packed)   // refresh every copy of this parameter in the MFMA-fragment-packed buffer (fused pack)
            for (int k = sc_off[i]; k < sc_off[i + 1]; ++k) packed[sc_idx[k]] = (T)pi;
    }
    if (loss_accum && i == 0) *loss_accum += (double)g[np];
}

int launch_adam(void *params, void *pcopy, const void *grads, void *m, void *v, int64_t np,
                size_t esize, const bamd_adam &hp, double *loss_accum, const int *sc_off, const int *sc_idx,
                void *packed, hipStream_t s) {
    double bc1 = 1.0 - pow(hp.beta1, (double)hp.step);
    double bc2 = 1.0 - pow(hp.beta2, (double)hp.step);
    double step_size = hp.lr / bc1;
    double bc2_sqrt = sqrt(bc2);
    dim3 g((unsigned)((np + 255) / 256)), b(256);
    if (esize == 8)
        hipLaunchKernelGGL(adam_k<double>, g, b, 0, s, (double *)params, (double *)pcopy, (const double *)grads,
                           (double *)m, (double *)v, np, hp.beta1, hp.beta2, hp.eps, step_size, bc2_sqrt, loss_accum, sc_off, sc_idx, (double *)packed);
    else
        hipLaunchKernelGGL(adam_k<float>, g, b, 0, s, (float *)params, (float *)pcopy, (const float *)grads,
                           (float *)m, (float *)v, np, hp.beta1, hp.beta2, hp.eps, step_size, bc2_sqrt, loss_accum, sc_off, sc_idx, (float *)packed);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

}  // namespace bamd
