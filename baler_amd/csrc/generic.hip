// Generic layer-by-layer path: any layer widths (CFD_dense_AE(2500,25), the 512-column encoder ...),
// fp32 (v_mfma_f32_16x16x4_f32) or fp64 (v_mfma_f64_16x16x4_f64).  One LDS-tiled MFMA GEMM kernel
// serves the three products of a Linear layer through operand descriptors and fused epilogues:
//   forward   Y  = lrelu(X W^T + b)                      (aten::addmm + aten::leaky_relu)
//   input grad dZ_prev = (dZ W) * lrelu'(Y_prev)          (aten::mm + aten::leaky_relu_backward)
//   weight grad [dW | db] = dZ^T [X | 1]                  (aten::mm + aten::sum), split over row
//              ranges into per-split slabs that are summed in a fixed order (deterministic).
// The fused register-chained kernels in fused.hip are the fast path for narrow models; this path is
// the general fallback and the independent cross-check for them.  gfx950 only.
#include "bamd_internal.hpp"
#include "fused.hpp"

#include <cstdlib>

namespace bamd {

template <typename T> struct MF;
template <> struct MF<float> {
    using v4 = __attribute__((ext_vector_type(4))) float;
    static __device__ __forceinline__ v4 mma(float a, float b, v4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D map of the f32 16x16 forms: col = lane&15, row = (lane>>4)*4 + reg
    static __device__ __forceinline__ int crow(int reg, int lane) { return (lane >> 4) * 4 + reg; }
};
template <> struct MF<double> {
    using v4 = __attribute__((ext_vector_type(4))) double;
    static __device__ __forceinline__ v4 mma(double a, double b, v4 c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // f64 16x16x4 uses its own map: col = lane&15, row = (lane>>4) + 4*reg
    static __device__ __forceinline__ int crow(int reg, int lane) { return (lane >> 4) + 4 * reg; }
};

// element(o, k) of a GEMM operand: p[o*s_o + k*s_k] inside [0,n_o) x [k_lo,k_hi), else 0;
// outer index == ones_o yields 1 (the appended ones-column that turns db into a GEMM column).
template <typename T> struct Opnd {
    const T *p;
    int64_t s_o, s_k, n_o, ones_o;
    __device__ __forceinline__ T get(int64_t o, int64_t k, int64_t k_hi) const {
        if (k >= k_hi) return (T)0;
        if (o < n_o) return p[o * s_o + k * s_k];
        if (o == ones_o) return (T)1;
        return (T)0;
    }
};

enum { EPI_FWD = 0, EPI_DX = 1, EPI_DW = 2, EPI_FWD_LOSS = 3 };   // FWD_LOSS: last layer of a training step: writes dL/drecon, sums the loss

template <typename T> struct Epi {
    T *out;            // FWD/DX: output matrix
    int64_t ld;        // its leading dimension
    int64_t n_rows, n_cols;
    const T *bias;     // FWD
    int act;           // FWD: apply leaky relu
    const T *ymask;    // DX: post-activation output of the previous layer (sign = pre-activation sign)
    int64_t ld_mask;
    const T *add;      // DX: extra gradient added to the result (latent regulariser injected at the bottleneck)
    int64_t ld_add;
    T *gw, *gb;        // DW: slab bases of this layer's dW and db
    int64_t kin;       // DW: in-features (column kin of the product is db)
    int64_t slab_stride;
    int64_t rows_per_split;
    const T *xref;     // FWD_LOSS: the (normalised) input rows the reconstruction is compared with, leading dimension ld
    double *loss_part; // FWD_LOSS: one partial sum of (r - x)^2 per workgroup
    double grad_scale; // FWD_LOSS: 2 / n_cols
};

template <typename T, int EPI>
__device__ __forceinline__ void epilogue(const Epi<T> &e, int64_t row, int64_t col, T val, double &lsum) {
    if (EPI == EPI_FWD_LOSS) {
        // utils.py:195-199 fused into de4's store: out = dL/drecon = 2 (r - x) / C, loss partial += (r - x)^2
        if (row < e.n_rows && col < e.n_cols) {
            val += e.bias[col];
            if (e.act) val = val > (T)0 ? val : val * (T)kSlope;
            const T d = val - e.xref[row * e.ld + col];
            lsum += (double)d * (double)d;
            e.out[row * e.ld + col] = (T)(e.grad_scale * (double)d);
        }
    } else if (EPI == EPI_FWD) {
        if (row < e.n_rows && col < e.n_cols) {
            val += e.bias[col];
            if (e.act) val = val > (T)0 ? val : val * (T)kSlope;
            e.out[row * e.ld + col] = val;
        }
    } else if (EPI == EPI_DX) {
        if (row < e.n_rows && col < e.n_cols) {
            if (e.ymask) val = e.ymask[row * e.ld_mask + col] > (T)0 ? val : val * (T)kSlope;
            if (e.add) val += e.add[row * e.ld_add + col];
            e.out[row * e.ld + col] = val;
        }
    } else {
        const int64_t so = (int64_t)blockIdx.z * e.slab_stride;
        if (row < e.n_rows) {
            if (col < e.kin) e.gw[so + row * e.kin + col] = val;
            else if (col == e.kin) e.gb[so + row] = val;
        }
    }
}

// fixed-order workgroup sum of the per-thread loss terms -> one partial per workgroup (row-major over the grid)
__device__ __forceinline__ void block_loss(double lsum, double *__restrict__ part) {
    __shared__ double lsh[256];
    lsh[threadIdx.x] = lsum;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) lsh[threadIdx.x] += lsh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = lsh[0];
}

template <typename T, int EPI, bool A_KFAST, bool B_KFAST>
__global__ void __launch_bounds__(256) gemm_k(Opnd<T> A, Opnd<T> B, int64_t kred, Epi<T> e) {
    using v4 = typename MF<T>::v4;
    __shared__ __attribute__((aligned(16))) T As[64][20];
    __shared__ __attribute__((aligned(16))) T Bs[64][20];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int64_t i0 = (int64_t)blockIdx.y * 64, j0 = (int64_t)blockIdx.x * 64;
    int64_t k_lo = 0, k_hi = kred;
    if (EPI == EPI_DW) {
        k_lo = (int64_t)blockIdx.z * e.rows_per_split;
        k_hi = k_lo + e.rows_per_split < kred ? k_lo + e.rows_per_split : kred;
    }
    v4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (v4){0, 0, 0, 0};

    for (int64_t k0 = k_lo; k0 < k_hi; k0 += 16) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int idx = tid + r * 256;
            int oa = A_KFAST ? (idx >> 4) : (idx & 63), ka = A_KFAST ? (idx & 15) : (idx >> 6);
            As[oa][ka] = A.get(i0 + oa, k0 + ka, k_hi);
            int ob = B_KFAST ? (idx >> 4) : (idx & 63), kb = B_KFAST ? (idx & 15) : (idx >> 6);
            Bs[ob][kb] = B.get(j0 + ob, k0 + kb, k_hi);
        }
        __syncthreads();
        // lane (o = lane&15, g = lane>>4) takes k = 4g..4g+3 of the 16-deep chunk for the four
        // MFMA steps; A and B use the same k permutation, so the sum over k is unchanged.
        T a4[2][4], b4[2][4];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a4[t][r] = As[wr * 32 + t * 16 + (lane & 15)][4 * (lane >> 4) + r];
                b4[t][r] = Bs[wc * 32 + t * 16 + (lane & 15)][4 * (lane >> 4) + r];
            }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = MF<T>::mma(a4[mt][r], b4[nt][r], acc[mt][nt]);
        __syncthreads();
    }

    double lsum = 0.0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                epilogue<T, EPI>(e, i0 + wr * 32 + mt * 16 + MF<T>::crow(reg, lane), j0 + wc * 32 + nt * 16 + (lane & 15),
                                 acc[mt][nt][reg], lsum);
    if (EPI == EPI_FWD_LOSS) block_loss(lsum, e.loss_part);
}

// 128 x 128 x 16 tile, 2 x 2 waves of 4 x 4 MFMA tiles each, double-buffered LDS with register staging: the global
// loads of chunk c+1 are in flight while chunk c feeds 64 MFMAs per wave (8 ds_read_b128 per 64 MFMAs), one barrier per
// chunk.  Used when both outer dimensions are >= 96 (the wide layers of CFD_dense_AE / the 512-column model, their
// input- and weight-gradient products); 16-wide sub-tiles entirely outside the matrix are skipped.
template <typename T, int EPI, bool A_KFAST, bool B_KFAST>
__global__ void __launch_bounds__(256) gemm_big_k(Opnd<T> A, Opnd<T> B, int64_t kred, Epi<T> e) {
    using v4 = typename MF<T>::v4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    // per operand two staging buffers of 2560 elements.  K-fast operands: [128 outer][16 k] with the four 16-byte k groups
    // of row o XOR-swizzled by (o >> 1) & 3: conflict-free 16-byte stores (8 lanes = 2 rows = 32 banks) AND fragment reads
    // (8 lanes = 8 rows, k group g: 8 distinct 4-bank groups); a padded [128][20] layout had 2-way store conflicts.
    // K-slow operands (outer index contiguous in memory): [16 k][132] (128 outer + 4 pad), so the
    // 4 consecutive outer elements a thread loads are ONE 16-byte LDS store -- transposing them into the [outer][k] layout
    // took four 4-byte stores that land 16-deep on two banks (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.89 on the
    // weight-gradient product) -- and the MFMA fragments are read as four 4-byte loads (lane groups on disjoint bank halves).
    T *As = (T *)smem_raw;
    T *Bs = As + 2 * 2560;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int64_t i0 = (int64_t)blockIdx.y * 128, j0 = (int64_t)blockIdx.x * 128;
    int64_t k_lo = 0, k_hi = kred;
    if (EPI == EPI_DW) {
        k_lo = (int64_t)blockIdx.z * e.rows_per_split;
        k_hi = k_lo + e.rows_per_split < kred ? k_lo + e.rows_per_split : kred;
    }
    const int64_t na = (A.ones_o >= 0 ? A.ones_o + 1 : A.n_o), nb = (B.ones_o >= 0 ? B.ones_o + 1 : B.n_o);
    bool live_m[4], live_n[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        live_m[t] = i0 + wr * 64 + 16 * t < na;
        live_n[t] = j0 + wc * 64 + 16 * t < nb;
    }
    v4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v4){0, 0, 0, 0};
    // Staging registers: 8 elements of A and 8 of B per thread per chunk.  Fast path (interior tile, full chunk, 4-element
    // aligned rows): two vector loads per operand -- K-fast operands take 4 consecutive k of one row, K-slow operands
    // (outer index contiguous) take 4 consecutive rows of one k; otherwise bounds-checked scalar loads.
    T ra[8], rb[8];
    const bool vecA = (i0 + 128 <= A.n_o) && ((A_KFAST ? A.s_o : A.s_k) % 4 == 0) && (((uintptr_t)A.p) % (4 * sizeof(T)) == 0) &&
                      (A_KFAST ? A.s_k == 1 : A.s_o == 1);
    const bool vecB = (j0 + 128 <= B.n_o) && ((B_KFAST ? B.s_o : B.s_k) % 4 == 0) && (((uintptr_t)B.p) % (4 * sizeof(T)) == 0) &&
                      (B_KFAST ? B.s_k == 1 : B.s_o == 1);
    auto gload_one = [&](const Opnd<T> &X, bool KF, bool vec, int64_t o0, int64_t k0, T (&reg)[8]) {
        if (vec && k0 + 16 <= k_hi && (KF ? (k0 % 4 == 0) : true)) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v4 v;
                if (KF) v = *(const v4 *)(X.p + (o0 + (tid >> 2) + 64 * h) * X.s_o + k0 + 4 * (tid & 3));
                else v = *(const v4 *)(X.p + (o0 + 4 * (tid & 31)) + (k0 + (tid >> 5) + 8 * h) * X.s_k);
#pragma unroll
                for (int j = 0; j < 4; ++j) reg[4 * h + j] = v[j];
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t o = KF ? o0 + (tid >> 2) + 64 * h : o0 + 4 * (tid & 31) + j;
                    const int64_t k = KF ? k0 + 4 * (tid & 3) + j : k0 + (tid >> 5) + 8 * h;
                    reg[4 * h + j] = X.get(o, k, k_hi);
                }
        }
    };
    auto gload = [&](int64_t k0) {
        gload_one(A, A_KFAST, vecA, i0, k0, ra);
        gload_one(B, B_KFAST, vecB, j0, k0, rb);
    };
    auto lstore_one = [&](T *S, bool KF, const T (&reg)[8]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = reg[4 * h + j];
            if (KF) { const int o = (tid >> 2) + 64 * h; *(v4 *)&S[o * 16 + 4 * ((tid & 3) ^ ((o >> 1) & 3))] = v; }
            else *(v4 *)&S[((tid >> 5) + 8 * h) * 132 + 4 * (tid & 31)] = v;
        }
    };
    auto lstore = [&](int buf) {
        lstore_one(As + buf * 2560, A_KFAST, ra);
        lstore_one(Bs + buf * 2560, B_KFAST, rb);
    };
    if (k_lo < k_hi) {
        gload(k_lo);
        lstore(0);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t k0 = k_lo; k0 < k_hi; k0 += 16, buf ^= 1) {
        const bool more = k0 + 16 < k_hi;
        if (more) gload(k0 + 16);
        T a4[4][4], b4[4][4];
        auto frag = [&](const T *S, bool KF, int o0, T (&f)[4]) {      // 16 outer x 16 k fragment of one 16-wide sub-tile
            if (KF) {
                const int o = o0 + (lane & 15);
                const v4 v = *(const v4 *)&S[o * 16 + 4 * ((lane >> 4) ^ ((o >> 1) & 3))];
#pragma unroll
                for (int r = 0; r < 4; ++r) f[r] = v[r];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) f[r] = S[(4 * (lane >> 4) + r) * 132 + o0 + (lane & 15)];
            }
        };
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            frag(As + buf * 2560, A_KFAST, wr * 64 + t * 16, a4[t]);
            frag(Bs + buf * 2560, B_KFAST, wc * 64 + t * 16, b4[t]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    if (live_m[mt] && live_n[nt]) acc[mt][nt] = MF<T>::mma(a4[mt][r], b4[nt][r], acc[mt][nt]);
        if (more) lstore(buf ^ 1);
        __syncthreads();
    }
    double lsum = 0.0;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                epilogue<T, EPI>(e, i0 + wr * 64 + mt * 16 + MF<T>::crow(reg, lane), j0 + wc * 64 + nt * 16 + (lane & 15),
                                 acc[mt][nt][reg], lsum);
    if (EPI == EPI_FWD_LOSS) block_loss(lsum, e.loss_part);
}

// dZ_L = 2 (R - X)/C and per-block partial sums of (R - X)^2 (utils.py:195-199 and its autograd).
template <typename T>
__global__ void __launch_bounds__(256) loss_grad_k(const T *__restrict__ r, const T *__restrict__ x,
                                                   int64_t count, double inv_c, T *__restrict__ dz,
                                                   double *__restrict__ part) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (int64_t)gridDim.x * blockDim.x) {
        T d = r[i] - x[i];
        acc += (double)d * (double)d;
        if (dz) dz[i] = (T)(2.0 * (double)d * inv_c);
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

template <typename TO>
__global__ void __launch_bounds__(256) loss_final_k(const double *__restrict__ part, int n, double scale, TO *dst,
                                                    int accumulate) {
    // fixed-order tree over <= 1024 partials (one block)
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        s = sh[0] * scale;
        *dst = accumulate ? (TO)((double)*dst + s) : (TO)s;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) reduce_slabs_k(const T *__restrict__ slabs, int nslab,
                                                      int64_t np, int64_t stride, T *__restrict__ g,
                                                      int accumulate) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= np) return;
    T s = accumulate ? g[i] : (T)0;
    for (int k = 0; k < nslab; ++k) s += slabs[(int64_t)k * stride + i];
    g[i] = s;
}

template <typename T>
__global__ void __launch_bounds__(256) colmean_k(const T *__restrict__ y, int64_t n, int c,
                                                 double *__restrict__ out) {
    // mean over rows of column blockIdx.x (fixed-order tree; diagnostics only)
    int col = blockIdx.x;
    double acc = 0.0;
    for (int64_t r = threadIdx.x; r < n; r += blockDim.x) acc += (double)y[r * c + col];
    __shared__ double sh[256];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[col] = sh[0] / (double)n;
}

__global__ void fill_nan_k(double *p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = NAN;
}

// ---- few rows (the reference's batch sizes on the layer-wise path: CFD configs 1 .. 85 rows in float64) -----------------------------
// The tiled kernels above give a 64 x 64 / 128 x 128 output tile to a workgroup that walks the whole contraction: a 60 x 200 product
// over K = 2500 is FOUR workgroups on a 256-CU chip (~40 us per launch, 35 launches per step).  Here a workgroup owns ONE 16 x 16 output
// tile, its four waves take a quarter of the contraction each (operands in MFMA layout straight from memory, `kSmallPF` steps of loads
// ahead of the MFMAs) and add their partial tiles through LDS in wave order (fixed: reproducible); FWD / FWD_LOSS / DX epilogues as above.
constexpr int kSmallPF = 8;
template <typename T, int EPI>
__global__ void __launch_bounds__(256) gemm_small_k(Opnd<T> A, Opnd<T> B, int64_t kred, Epi<T> e) {
    using v4 = typename MF<T>::v4;
    __shared__ __attribute__((aligned(16))) v4 red[3][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, o = lane & 15, kg = lane >> 4;
    const int64_t i0 = (int64_t)blockIdx.y * 16, j0 = (int64_t)blockIdx.x * 16;
    const int64_t steps = (kred + 3) / 4, per = (steps + 3) / 4;               // contraction steps of four; this wave's range
    const int64_t s_lo = w * per, s_hi = s_lo + per < steps ? s_lo + per : steps;
    v4 acc = (v4){0, 0, 0, 0};
    T a[kSmallPF], b[kSmallPF];
    for (int64_t st = s_lo; st < s_hi; st += kSmallPF) {
#pragma unroll
        for (int u = 0; u < kSmallPF; ++u) {
            const int64_t k = (st + u) * 4 + kg;
            const bool ok = st + u < s_hi;
            a[u] = ok ? A.get(i0 + o, k, kred) : (T)0;
            b[u] = ok ? B.get(j0 + o, k, kred) : (T)0;
        }
#pragma unroll
        for (int u = 0; u < kSmallPF; ++u) acc = MF<T>::mma(a[u], b[u], acc);
    }
    if (w > 0) red[w - 1][lane] = acc;
    __syncthreads();
    double lsum = 0.0;
    if (w == 0) {
        acc = ((acc + red[0][lane]) + red[1][lane]) + red[2][lane];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) epilogue<T, EPI>(e, i0 + MF<T>::crow(reg, lane), j0 + o, acc[reg], lsum);
    }
    if (EPI == EPI_FWD_LOSS) block_loss(lsum, e.loss_part);
}
static int64_t gemm_small_rows() {      // BALER_AMD_GEMM_SMALL_ROWS: most rows for the one-tile-per-workgroup kernel (0: off)
    // (C4 in float32, layer-wise step: 256 rows 965 -> 195 us, 512: 1004 -> 281, 1024: 1052 -> 477)
    return env_ll("BALER_AMD_GEMM_SMALL_ROWS", 1024);
}

// pick the tile: 128 x 128 (double-buffered) when both outer extents are large enough, else 64 x 64
template <typename T, int EPI, bool AK, bool BK>
static void launch_gemm(const Opnd<T> &A, const Opnd<T> &B, int64_t kred, const Epi<T> &e, int64_t outer_a, int64_t outer_b,
                        unsigned nz, hipStream_t s) {
    // weight-gradient products of narrow layers (e.g. 50 x 101 outputs) reduce over tens of thousands of rows: the 64 x 64
    // kernel's element-wise K-slow loads made them the slowest launches of a CFD_dense_AE step (103 us each for 0.2 GFLOP);
    // the big-tile kernel skips its dead 16-wide sub-tiles and loads 16 bytes per lane
    const bool long_dw = EPI == EPI_DW && kred >= 4096 && outer_a >= 16 && outer_b >= 16;
    if constexpr (EPI != EPI_DW) {
        if (outer_a <= gemm_small_rows()) {      // (outer_a = the rows of the batch in the forward and input-gradient products)
            dim3 grid((unsigned)((outer_b + 15) / 16), (unsigned)((outer_a + 15) / 16), 1);
            hipLaunchKernelGGL((gemm_small_k<T, EPI>), grid, dim3(256), 0, s, A, B, kred, e);
            return;
        }
    }
    if ((outer_a >= 96 && outer_b >= 96) || long_dw) {
        const int lds = 4 * 128 * 20 * (int)sizeof(T);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)gemm_big_k<T, EPI, AK, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr_set = true;
        }
        dim3 grid((unsigned)((outer_b + 127) / 128), (unsigned)((outer_a + 127) / 128), nz);
        hipLaunchKernelGGL((gemm_big_k<T, EPI, AK, BK>), grid, dim3(256), lds, s, A, B, kred, e);
    } else {
        dim3 grid((unsigned)((outer_b + 63) / 64), (unsigned)((outer_a + 63) / 64), nz);
        hipLaunchKernelGGL((gemm_k<T, EPI, AK, BK>), grid, dim3(256), 0, s, A, B, kred, e);
    }
}

// ---- weight gradients with one short side (float32) --------------------------------------------------------------------------
// [dW | db] = dZ^T [X | 1] of an autoencoder layer always has one side of at most a few hundred columns (200 x 2501, 2500 x 201,
// 100 x 201 ...) and reduces over tens of thousands of rows.  The LDS-tiled kernel pays 128-wide tile padding on the short side
// (201 -> 256) and needs many row splits to fill the chip with 128 x 128 tiles.  Here a wave keeps TQ x PT accumulator tiles in
// registers -- ALL PT 16-column tiles of the short side "P" against TQ tiles of the other side "Q" -- for its whole row range, and
// both operands are read in MFMA layout straight from the row-major matrices: the reduction index of v_mfma_f32_16x16x4_f32 is
// the ROW (step s, lane group g -> row 4 s + g of a 16-row block), lane i of a group the column, so every operand register is
// one coalesced dword load (64 contiguous bytes per lane group), no transpose and no LDS.  The four waves of a workgroup share
// the P loads through the L1; operands of the next 16-row block are loaded while the current block multiplies (two register
// sets, no copies).  P_IS_N: P = dZ (n side), Q = [X | 1]; otherwise P = [X | 1], Q = dZ.  The ones column is a select on the
// tile that holds column K.  Output: this split's slab [n * K + k | n] (fixed order reduction by reduce_layers_k).
template <int PT, int TQ, bool P_IS_N>
__global__ void __launch_bounds__(256) dw_short_k(const float *__restrict__ dzm, const float *__restrict__ xm, int N, int K, int64_t rows,
                                                  int64_t rps, float *__restrict__ slab, int64_t slab_size) {
    using v4 = MF<float>::v4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const float *__restrict__ pm = P_IS_N ? dzm : xm;
    const float *__restrict__ qm = P_IS_N ? xm : dzm;
    const int DP = P_IS_N ? N : K, DQ = P_IS_N ? K : N;            // stored widths
    const int CQ = P_IS_N ? K + 1 : N;                             // logical columns of Q
    const int qt0 = ((int)blockIdx.x * 4 + wave) * TQ;
    if (qt0 * 16 >= CQ) return;
    const int64_t r_begin = (int64_t)blockIdx.y * rps;
    const int64_t r_end = r_begin + rps < rows ? r_begin + rps : rows;
    if (r_begin >= r_end) return;
    // per-lane element offsets inside a 16-row block (columns beyond the stored width re-read the last column: their
    // accumulators are padding and never stored)
    // Operand loads go through buffer resources based at the split's first row: ONE byte offset per lane and operand side in a
    // VGPR (tile t = +64 t bytes, an immediate; PT = tiles(logical columns), so only the LAST P tile can reach beyond the stored
    // width and has its own clamped offset), block and step offsets in SGPRs -- no 64-bit address arithmetic on the vector unit
    // (global loads cost one v_lshl_add_u64 each).  The host keeps rows-per-split x width below 2^29 elements.
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)(pm + r_begin * DP), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)(qm + r_begin * DQ), 0, 0x7fffffff, 0x00020000);
    int offq[TQ];
    const int offp0 = (g * DP + i) * 4;
    const int offpl = (g * DP + (16 * (PT - 1) + i < DP ? 16 * (PT - 1) + i : DP - 1)) * 4;
#pragma unroll
    for (int u = 0; u < TQ; ++u) { const int c = 16 * (qt0 + u) + i; offq[u] = (g * DQ + (c < DQ ? c : DQ - 1)) * 4; }
    auto ldp = [&](int voff, int soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, voff, soff, 0)); };
    auto ldq = [&](int voff, int soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rq, voff, soff, 0)); };
    const bool p_one = !P_IS_N && 16 * (PT - 1) + i == K;          // this lane's column of the LAST P tile is the ones column
    bool q_one[TQ];
#pragma unroll
    for (int u = 0; u < TQ; ++u) q_one[u] = P_IS_N && 16 * (qt0 + u) + i == K;
    v4 acc[TQ][PT];
#pragma unroll
    for (int u = 0; u < TQ; ++u)
#pragma unroll
        for (int t = 0; t < PT; ++t) acc[u][t] = (v4){0.f, 0.f, 0.f, 0.f};
    struct Frags { v4 p[PT]; v4 q[TQ]; };
    auto load = [&](Frags &f, int64_t rb, bool tail) {      // rb = first row of the block, relative to r_begin
        const int sp = (int)rb * DP * 4, sq = (int)rb * DQ * 4;
        if (!tail) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
#pragma unroll
                for (int u = 0; u < TQ; ++u) f.q[u][s4] = ldq(offq[u], sq + s4 * 16 * DQ);
#pragma unroll
                for (int t = 0; t < PT; ++t) f.p[t][s4] = ldp(t == PT - 1 ? offpl : offp0 + 64 * t, sp + s4 * 16 * DP);
            }
        } else {            // last block of the range: rows beyond it read the last row and contribute zero through dZ
            const int64_t nr = r_end - r_begin;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const bool ok = rb + 4 * s4 + g < nr;
                const int back = ok ? 0 : (int)(rb + 4 * s4 + g - (nr - 1));
#pragma unroll
                for (int u = 0; u < TQ; ++u) {
                    const float v = ldq(offq[u] - back * DQ * 4, sq + s4 * 16 * DQ);
                    f.q[u][s4] = (!P_IS_N && !ok) ? 0.f : v;
                }
#pragma unroll
                for (int t = 0; t < PT; ++t) {
                    const float v = ldp((t == PT - 1 ? offpl : offp0 + 64 * t) - back * DP * 4, sp + s4 * 16 * DP);
                    f.p[t][s4] = (P_IS_N && !ok) ? 0.f : v;
                }
            }
        }
    };
    auto mma = [&](Frags &f) {
        if (!P_IS_N) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) f.p[PT - 1][s4] = p_one ? 1.0f : f.p[PT - 1][s4];
        } else {
#pragma unroll
            for (int u = 0; u < TQ; ++u)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) f.q[u][s4] = q_one[u] ? 1.0f : f.q[u][s4];
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int u = 0; u < TQ; ++u)
#pragma unroll
                for (int t = 0; t < PT; ++t) acc[u][t] = MF<float>::mma(f.q[u][s4], f.p[t][s4], acc[u][t]);
    };
    Frags fa, fb;
    const int64_t nfull = (r_end - r_begin) >> 4;          // full 16-row blocks; the loop below never sees the ragged one
    if (nfull > 0) {
        load(fa, 0, false);
        int64_t b = 0;
        for (; b + 2 < nfull; b += 2) {
            load(fb, 16 * (b + 1), false);          // issue the next block's loads BEFORE this block's MFMAs (hipcc sinks them otherwise)
            __builtin_amdgcn_sched_barrier(0);
            mma(fa);
            __builtin_amdgcn_sched_barrier(0);
            load(fa, 16 * (b + 2), false);
            __builtin_amdgcn_sched_barrier(0);
            mma(fb);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (b + 2 == nfull) {
            load(fb, 16 * (b + 1), false);
            mma(fa);
            mma(fb);
        } else {
            mma(fa);
        }
    }
    if ((r_end - r_begin) & 15) {
        load(fa, 16 * nfull, true);
        mma(fa);
    }
    // C map: register r of lane (g, i) = (q column 4 g + r of the tile, p column i)
    float *out = slab + (int64_t)blockIdx.y * slab_size;
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        const int q0 = 16 * (qt0 + u) + 4 * g;
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int pc = 16 * t + i;
            if (P_IS_N) {                                   // n = p column, k = q column: the lane's 4 registers are 4 consecutive k
                if (pc < N) {
                    if ((K & 3) == 0 && q0 + 3 < K) {
                        *(v4 *)(out + (int64_t)pc * K + q0) = acc[u][t];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (q0 + r < K) out[(int64_t)pc * K + q0 + r] = acc[u][t][r];
                            else if (q0 + r == K) out[(int64_t)N * K + pc] = acc[u][t][r];
                        }
                    }
                }
            } else {                                        // n = q column, k = p column: lanes i are 16 consecutive k
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (q0 + r < N) {
                        if (pc < K) out[(int64_t)(q0 + r) * K + pc] = acc[u][t][r];
                        else if (pc == K) out[(int64_t)N * K + q0 + r] = acc[u][t][r];
                    }
                }
            }
        }
    }
}

// The two WIDE weight gradients of a wide model (200 x 2501, 2500 x 201): the same structure with 16-byte operand loads.  The
// vector-memory pipe issues one wave instruction per ~16 cycles whatever its width, so dword operand loads (60 per 104 MFMAs and
// wave above) make the four waves of a CU VMEM-issue bound (measured: MFMA busy 45 %, 8 % of the cycles in s_waitcnt).  Here lane
// (g, i) loads FOUR consecutive columns 4 i .. 4 i + 3 of row 4 s + g: register j of that load is the step-s operand of the
// "strided tile" {64 G + 4 i + j : i = 0..15} of column group G -- a legal MFMA operand, since the column-to-lane map of an
// operand is free as long as the epilogue knows it.  The short side (193..208 logical columns) is 3 groups (12 strided tiles) +
// 1 plain tile of dword loads, the other side one group per wave: 20 loads per 208 MFMAs.  52 accumulator tiles (AGPRs), two
// operand sets of 68 registers.  Needs widths and K divisible by 4 (16-byte rows; the ones column starts a 4-column group).
template <bool P_IS_N>
__global__ void __launch_bounds__(256) dw_wide_k(const float *__restrict__ dzm, const float *__restrict__ xm, int N, int K, int64_t rows,
                                                 int64_t rps, float *__restrict__ slab, int64_t slab_size) {
    using v4 = MF<float>::v4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const float *__restrict__ pm = P_IS_N ? dzm : xm;
    const float *__restrict__ qm = P_IS_N ? xm : dzm;
    const int DP = P_IS_N ? N : K, DQ = P_IS_N ? K : N;
    const int CQ = P_IS_N ? K + 1 : N;
    const int gq = (int)blockIdx.x * 4 + wave;                     // this wave's column group of Q
    if (64 * gq >= CQ) return;
    const int64_t r_begin = (int64_t)blockIdx.y * rps;
    const int64_t r_end = r_begin + rps < rows ? r_begin + rps : rows;
    if (r_begin >= r_end) return;
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)(pm + r_begin * DP), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)(qm + r_begin * DQ), 0, 0x7fffffff, 0x00020000);
    // columns beyond the stored width re-read the last 4 columns (padding accumulators, never stored)
    const int cq = 64 * gq + 4 * i;
    const int offq = (g * DQ + (cq + 4 <= DQ ? cq : DQ - 4)) * 4;
    const int offpg = (g * DP + 4 * i) * 4;                                       // group G: + 256 G bytes
    const int offpl = (g * DP + (192 + i < DP ? 192 + i : DP - 1)) * 4;           // the plain tile: columns 192 + i
    const bool q_one = P_IS_N && cq == K;            // register 0 of this lane's Q load is the ones column
    const bool p_one = !P_IS_N && 192 + i == K;
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t &r, int voff, int soff) {
        return __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    v4 acc[4][13];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 13; ++t) acc[u][t] = (v4){0.f, 0.f, 0.f, 0.f};
    struct Frags { v4 pg[3][4]; v4 pl; v4 q[4]; };                 // [group][step], plain tile [step], [step]
    auto load = [&](Frags &f, int64_t rb, bool tail) {
        const int sp = (int)rb * DP * 4, sq = (int)rb * DQ * 4;
        const int64_t nr = r_end - r_begin;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            int back_p = 0, back_q = 0;
            bool ok = true;
            if (tail) {                                            // rows beyond the range read its last row, dZ zeroed
                ok = rb + 4 * s4 + g < nr;
                const int back = ok ? 0 : (int)(rb + 4 * s4 + g - (nr - 1));
                back_p = back * DP * 4;
                back_q = back * DQ * 4;
            }
            f.q[s4] = ld4(rq, offq - back_q, sq + s4 * 16 * DQ);
#pragma unroll
            for (int G = 0; G < 3; ++G) f.pg[G][s4] = ld4(rp, offpg + 256 * G - back_p, sp + s4 * 16 * DP);
            f.pl[s4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, offpl - back_p, sp + s4 * 16 * DP, 0));
            if (tail && !ok) {
                if (P_IS_N) {
#pragma unroll
                    for (int G = 0; G < 3; ++G) f.pg[G][s4] = (v4){0.f, 0.f, 0.f, 0.f};
                    f.pl[s4] = 0.f;
                } else {
                    f.q[s4] = (v4){0.f, 0.f, 0.f, 0.f};
                }
            }
        }
    };
    auto mma = [&](Frags &f) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            if (P_IS_N) f.q[s4][0] = q_one ? 1.0f : f.q[s4][0];
            else f.pl[s4] = p_one ? 1.0f : f.pl[s4];
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int G = 0; G < 3; ++G)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[u][4 * G + j] = MF<float>::mma(f.q[s4][u], f.pg[G][s4][j], acc[u][4 * G + j]);
                acc[u][12] = MF<float>::mma(f.q[s4][u], f.pl[s4], acc[u][12]);
            }
    };
    Frags fa, fb;
    const int64_t nfull = (r_end - r_begin) >> 4;
    if (nfull > 0) {
        load(fa, 0, false);
        int64_t b = 0;
        for (; b + 2 < nfull; b += 2) {
            load(fb, 16 * (b + 1), false);          // issue the next block's loads BEFORE this block's MFMAs (hipcc sinks them otherwise)
            __builtin_amdgcn_sched_barrier(0);
            mma(fa);
            __builtin_amdgcn_sched_barrier(0);
            load(fa, 16 * (b + 2), false);
            __builtin_amdgcn_sched_barrier(0);
            mma(fb);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (b + 2 == nfull) {
            load(fb, 16 * (b + 1), false);
            mma(fa);
            mma(fb);
        } else {
            mma(fa);
        }
    }
    if ((r_end - r_begin) & 15) {
        load(fa, 16 * nfull, true);
        mma(fa);
    }
    // C map: register r of lane (g, i) in accumulator (u, t): Q column 64 gq + 16 g + 4 r + u; P column 64 G + 4 i + j for the
    // strided tile t = 4 G + j, 192 + i for the plain tile t = 12
    float *out = slab + (int64_t)blockIdx.y * slab_size;
    if (P_IS_N) {                  // n = P column, k = Q column: the four Q tiles of a register are 4 consecutive k
#pragma unroll
        for (int t = 0; t < 13; ++t) {
            const int n = t < 12 ? 64 * (t >> 2) + 4 * i + (t & 3) : 192 + i;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k0 = 64 * gq + 16 * g + 4 * r;
                if (k0 + 3 < K) {
                    *(v4 *)(out + (int64_t)n * K + k0) = (v4){acc[0][t][r], acc[1][t][r], acc[2][t][r], acc[3][t][r]};
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (k0 + u < K) out[(int64_t)n * K + k0 + u] = acc[u][t][r];
                        else if (k0 + u == K) out[(int64_t)N * K + n] = acc[u][t][r];
                    }
                }
            }
        }
    } else {                       // n = Q column, k = P column: the four strided tiles of a group are 4 consecutive k
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 64 * gq + 16 * g + 4 * r + u;
                if (n >= N) continue;
#pragma unroll
                for (int G = 0; G < 3; ++G)
                    *(v4 *)(out + (int64_t)n * K + 64 * G + 4 * i) = (v4){acc[u][4 * G][r], acc[u][4 * G + 1][r], acc[u][4 * G + 2][r], acc[u][4 * G + 3][r]};
                const int k = 192 + i;
                if (k < K) out[(int64_t)n * K + k] = acc[u][12][r];
                else if (k == K) out[(int64_t)N * K + n] = acc[u][12][r];
            }
    }
}

// The same two products on v_mfma_f32_16x16x32_bf16, for BAMD_MODE_BF16 handles: operands rounded to bfloat16 from the float32 rows on
// load, float32 accumulation, float32 partial gradients.  The contraction index of that MFMA holds EIGHT consecutive batch rows per
// lane: lane (g, i) loads columns 4 i .. 4 i + 3 of rows 8 g + e (e = 0..7) of a 32-row block -- 16-byte loads, 4 rows x 256 contiguous
// bytes per instruction -- and register j of the eight loads, packed by four v_cvt_pk_bf16_f32, is the operand of strided tile j.
// Column-to-lane maps, accumulators and epilogue are dw_wide_k's.  52 MFMAs (832 cycles) per 40 KiB loaded: the kernel is bound by the
// rows it streams (the float32 launch: MFMA busy 73-75 %).
// Q16: the Q side (dZ of the last layer, P_IS_N = false) is stored as bfloat16: 8-byte loads of the lane's four columns, the k slots
// assembled with v_perm_b32 instead of conversions.
template <bool P_IS_N, bool Q16 = false>
__global__ void __launch_bounds__(256) dw_wide_bf16_k(const float *__restrict__ dzm, const float *__restrict__ xm, int N, int K, int64_t rows,
                                                      int64_t rps, float *__restrict__ slab, int64_t slab_size) {
    static_assert(!(Q16 && P_IS_N), "the bfloat16 operand is dZ on the Q side");
    using v4 = MF<float>::v4;
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    constexpr int qes = Q16 ? 2 : 4;                               // bytes per stored element of Q
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const float *__restrict__ pm = P_IS_N ? dzm : xm;
    const float *__restrict__ qm = P_IS_N ? xm : dzm;
    const int DP = P_IS_N ? N : K, DQ = P_IS_N ? K : N;
    const int CQ = P_IS_N ? K + 1 : N;
    const int gq = (int)blockIdx.x * 4 + wave;
    if (64 * gq >= CQ) return;
    const int64_t r_begin = (int64_t)blockIdx.y * rps;
    const int64_t r_end = r_begin + rps < rows ? r_begin + rps : rows;
    if (r_begin >= r_end) return;
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)(pm + r_begin * DP), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)qm + r_begin * DQ * qes), 0, 0x7fffffff, 0x00020000);
    const int cq = 64 * gq + 4 * i;
    const int offq = (8 * g * DQ + (cq + 4 <= DQ ? cq : DQ - 4)) * qes;
    const int offpg = (8 * g * DP + 4 * i) * 4;
    const int offpl = (8 * g * DP + (192 + i < DP ? 192 + i : DP - 1)) * 4;
    const bool q_one = P_IS_N && cq == K;
    const bool p_one = !P_IS_N && 192 + i == K;
    auto ld4 = [&](const __amdgpu_buffer_rsrc_t &r, int voff, int soff) {
        return __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    v4 acc[4][13];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < 13; ++t) acc[u][t] = (v4){0.f, 0.f, 0.f, 0.f};
    struct Raw { v4 pg[3][8]; float pl[8]; v4 q[8]; u2 q16[8]; };  // [group][row e], plain tile [row e], [row e] (q16: Q16)
    struct Pk { bf8 p[13]; bf8 q[4]; };
    const int64_t nr = r_end - r_begin;
    auto load = [&](Raw &f, int64_t rb, bool tail) {
        const int sp = (int)rb * DP * 4, sq = (int)rb * DQ * qes;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int back_p = 0, back_q = 0;
            bool ok = true;
            if (tail) {                                            // rows beyond the range read its last row, dZ zeroed
                ok = rb + 8 * g + e < nr;
                const int back = ok ? 0 : (int)(rb + 8 * g + e - (nr - 1));
                back_p = back * DP * 4;
                back_q = back * DQ * qes;
            }
            if constexpr (Q16) f.q16[e] = __builtin_bit_cast(u2, __builtin_amdgcn_raw_buffer_load_b64(rq, offq - back_q, sq + e * qes * DQ, 0));
            else f.q[e] = ld4(rq, offq - back_q, sq + e * 4 * DQ);
#pragma unroll
            for (int G = 0; G < 3; ++G) f.pg[G][e] = ld4(rp, offpg + 256 * G - back_p, sp + e * 4 * DP);
            f.pl[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, offpl - back_p, sp + e * 4 * DP, 0));
            if (tail && !ok) {
                if (P_IS_N) {
#pragma unroll
                    for (int G = 0; G < 3; ++G) f.pg[G][e] = (v4){0.f, 0.f, 0.f, 0.f};
                    f.pl[e] = 0.f;
                } else {
                    f.q[e] = (v4){0.f, 0.f, 0.f, 0.f};
                    f.q16[e] = (u2){0u, 0u};
                }
            }
        }
    };
    auto pack8 = [&](float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
        const bf2 p0 = {(__bf16)a0, (__bf16)a1}, p1 = {(__bf16)a2, (__bf16)a3}, p2 = {(__bf16)a4, (__bf16)a5}, p3 = {(__bf16)a6, (__bf16)a7};
        const u4 w = {__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1), __builtin_bit_cast(unsigned, p2),
                      __builtin_bit_cast(unsigned, p3)};
        return __builtin_bit_cast(bf8, w);
    };
    auto pack = [&](Pk &k, Raw &f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (P_IS_N) f.q[e][0] = q_one ? 1.0f : f.q[e][0];
            else f.pl[e] = p_one ? 1.0f : f.pl[e];
        }
#pragma unroll
        for (int G = 0; G < 3; ++G)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                k.p[4 * G + j] = pack8(f.pg[G][0][j], f.pg[G][1][j], f.pg[G][2][j], f.pg[G][3][j], f.pg[G][4][j], f.pg[G][5][j], f.pg[G][6][j],
                                       f.pg[G][7][j]);
        k.p[12] = pack8(f.pl[0], f.pl[1], f.pl[2], f.pl[3], f.pl[4], f.pl[5], f.pl[6], f.pl[7]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (Q16) {      // column u of rows e = 2 d, 2 d + 1 -> dword d of the operand: halves picked by v_perm_b32
                u4 w;
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    w[d] = __builtin_amdgcn_perm(f.q16[2 * d + 1][u >> 1], f.q16[2 * d][u >> 1], (u & 1) ? 0x07060302u : 0x05040100u);
                k.q[u] = __builtin_bit_cast(bf8, w);
            } else {
                k.q[u] = pack8(f.q[0][u], f.q[1][u], f.q[2][u], f.q[3][u], f.q[4][u], f.q[5][u], f.q[6][u], f.q[7][u]);
            }
        }
    };
    auto mma = [&](const Pk &k) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 13; ++t) acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k.q[u], k.p[t], acc[u][t], 0, 0, 0);
    };
    Raw ra;
    Pk pk;
    const int64_t nfull = nr >> 5;
    if (nfull > 0) {
        load(ra, 0, false);
        for (int64_t b = 0; b < nfull; ++b) {
            pack(pk, ra);
            __builtin_amdgcn_sched_barrier(0);
            if (b + 1 < nfull) load(ra, 32 * (b + 1), false);      // the next block's rows stream in behind this block's MFMAs
            __builtin_amdgcn_sched_barrier(0);
            mma(pk);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (nr & 31) {
        load(ra, 32 * nfull, true);
        pack(pk, ra);
        mma(pk);
    }
    // C map: as dw_wide_k (register r of lane (g, i) in accumulator (u, t): Q column 64 gq + 16 g + 4 r + u; P column 64 G + 4 i + j for
    // the strided tile t = 4 G + j, 192 + i for the plain tile t = 12)
    float *out = slab + (int64_t)blockIdx.y * slab_size;
    if (P_IS_N) {
#pragma unroll
        for (int t = 0; t < 13; ++t) {
            const int n = t < 12 ? 64 * (t >> 2) + 4 * i + (t & 3) : 192 + i;
            if (n >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k0 = 64 * gq + 16 * g + 4 * r;
                if (k0 + 3 < K) {
                    *(v4 *)(out + (int64_t)n * K + k0) = (v4){acc[0][t][r], acc[1][t][r], acc[2][t][r], acc[3][t][r]};
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (k0 + u < K) out[(int64_t)n * K + k0 + u] = acc[u][t][r];
                        else if (k0 + u == K) out[(int64_t)N * K + n] = acc[u][t][r];
                    }
                }
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 64 * gq + 16 * g + 4 * r + u;
                if (n >= N) continue;
#pragma unroll
                for (int G = 0; G < 3; ++G)
                    *(v4 *)(out + (int64_t)n * K + 64 * G + 4 * i) = (v4){acc[u][4 * G][r], acc[u][4 * G + 1][r], acc[u][4 * G + 2][r], acc[u][4 * G + 3][r]};
                const int k = 192 + i;
                if (k < K) out[(int64_t)n * K + k] = acc[u][12][r];
                else if (k == K) out[(int64_t)N * K + n] = acc[u][12][r];
            }
    }
}

// dw_short_k's plain tiles on the bf16 MFMA, for the wide weight gradients of a BAMD_MODE_BF16 handle whose rows are not 16-byte
// aligned (the exafel blocks' 625 columns): lane (g, i) loads column 16 t + i of rows 8 g + e (e = 0..7) of a 32-row block with dword
// loads and packs the eight values of a tile into the MFMA's k slots.  120 loads per 26 MFMAs: issue-bound on the vector-memory
// pipe (~63 us at 131,072 blocks), still well under the float32 kernel's MFMA time (364-390 us).
template <int PT, int TQ, bool P_IS_N>
__global__ void __launch_bounds__(256) dw_short_bf16_k(const float *__restrict__ dzm, const float *__restrict__ xm, int N, int K, int64_t rows,
                                                       int64_t rps, float *__restrict__ slab, int64_t slab_size) {
    using v4 = MF<float>::v4;
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const float *__restrict__ pm = P_IS_N ? dzm : xm;
    const float *__restrict__ qm = P_IS_N ? xm : dzm;
    const int DP = P_IS_N ? N : K, DQ = P_IS_N ? K : N;
    const int CQ = P_IS_N ? K + 1 : N;
    const int qt0 = ((int)blockIdx.x * 4 + wave) * TQ;
    if (qt0 * 16 >= CQ) return;
    const int64_t r_begin = (int64_t)blockIdx.y * rps;
    const int64_t r_end = r_begin + rps < rows ? r_begin + rps : rows;
    if (r_begin >= r_end) return;
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void *)(pm + r_begin * DP), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)(qm + r_begin * DQ), 0, 0x7fffffff, 0x00020000);
    int offq[TQ];
    const int offp0 = (8 * g * DP + i) * 4;
    const int offpl = (8 * g * DP + (16 * (PT - 1) + i < DP ? 16 * (PT - 1) + i : DP - 1)) * 4;
#pragma unroll
    for (int u = 0; u < TQ; ++u) { const int c = 16 * (qt0 + u) + i; offq[u] = (8 * g * DQ + (c < DQ ? c : DQ - 1)) * 4; }
    auto ldp = [&](int voff, int soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rp, voff, soff, 0)); };
    auto ldq = [&](int voff, int soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rq, voff, soff, 0)); };
    const bool p_one = !P_IS_N && 16 * (PT - 1) + i == K;
    bool q_one[TQ];
#pragma unroll
    for (int u = 0; u < TQ; ++u) q_one[u] = P_IS_N && 16 * (qt0 + u) + i == K;
    v4 acc[TQ][PT];
#pragma unroll
    for (int u = 0; u < TQ; ++u)
#pragma unroll
        for (int t = 0; t < PT; ++t) acc[u][t] = (v4){0.f, 0.f, 0.f, 0.f};
    struct Raw { float p[PT][8]; float q[TQ][8]; };
    struct Pk { bf8 p[PT]; bf8 q[TQ]; };
    const int64_t nr = r_end - r_begin;
    auto load = [&](Raw &f, int64_t rb, bool tail) {
        const int sp = (int)rb * DP * 4, sq = (int)rb * DQ * 4;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bool ok = true;
            int back = 0;
            if (tail) {                                            // rows beyond the range read its last row, dZ zeroed
                ok = rb + 8 * g + e < nr;
                back = ok ? 0 : (int)(rb + 8 * g + e - (nr - 1));
            }
#pragma unroll
            for (int u = 0; u < TQ; ++u) {
                const float v = ldq(offq[u] - back * DQ * 4, sq + e * 4 * DQ);
                f.q[u][e] = (tail && !P_IS_N && !ok) ? 0.f : v;
            }
#pragma unroll
            for (int t = 0; t < PT; ++t) {
                const float v = ldp((t == PT - 1 ? offpl : offp0 + 64 * t) - back * DP * 4, sp + e * 4 * DP);
                f.p[t][e] = (tail && P_IS_N && !ok) ? 0.f : v;
            }
        }
    };
    auto pack8 = [&](const float (&a)[8]) {
        const bf2 p0 = {(__bf16)a[0], (__bf16)a[1]}, p1 = {(__bf16)a[2], (__bf16)a[3]}, p2 = {(__bf16)a[4], (__bf16)a[5]}, p3 = {(__bf16)a[6], (__bf16)a[7]};
        const u4 w = {__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1), __builtin_bit_cast(unsigned, p2),
                      __builtin_bit_cast(unsigned, p3)};
        return __builtin_bit_cast(bf8, w);
    };
    auto pack = [&](Pk &k, Raw &f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (!P_IS_N) f.p[PT - 1][e] = p_one ? 1.0f : f.p[PT - 1][e];
            else {
#pragma unroll
                for (int u = 0; u < TQ; ++u) f.q[u][e] = q_one[u] ? 1.0f : f.q[u][e];
            }
        }
#pragma unroll
        for (int t = 0; t < PT; ++t) k.p[t] = pack8(f.p[t]);
#pragma unroll
        for (int u = 0; u < TQ; ++u) k.q[u] = pack8(f.q[u]);
    };
    auto mma = [&](const Pk &k) {
#pragma unroll
        for (int u = 0; u < TQ; ++u)
#pragma unroll
            for (int t = 0; t < PT; ++t) acc[u][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k.q[u], k.p[t], acc[u][t], 0, 0, 0);
    };
    Raw ra;
    Pk pk;
    const int64_t nfull = nr >> 5;
    if (nfull > 0) {
        load(ra, 0, false);
        for (int64_t b = 0; b < nfull; ++b) {
            pack(pk, ra);
            __builtin_amdgcn_sched_barrier(0);
            if (b + 1 < nfull) load(ra, 32 * (b + 1), false);
            __builtin_amdgcn_sched_barrier(0);
            mma(pk);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (nr & 31) {
        load(ra, 32 * nfull, true);
        pack(pk, ra);
        mma(pk);
    }
    // C map: register r of lane (g, i) = (q column 4 g + r of the tile, p column i), as dw_short_k
    float *out = slab + (int64_t)blockIdx.y * slab_size;
#pragma unroll
    for (int u = 0; u < TQ; ++u) {
        const int q0 = 16 * (qt0 + u) + 4 * g;
#pragma unroll
        for (int t = 0; t < PT; ++t) {
            const int pc = 16 * t + i;
            if (P_IS_N) {
                if (pc < N) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (q0 + r < K) out[(int64_t)pc * K + q0 + r] = acc[u][t][r];
                        else if (q0 + r == K) out[(int64_t)N * K + pc] = acc[u][t][r];
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (q0 + r < N) {
                        if (pc < K) out[(int64_t)(q0 + r) * K + pc] = acc[u][t][r];
                        else if (pc == K) out[(int64_t)N * K + q0 + r] = acc[u][t][r];
                    }
                }
            }
        }
    }
}

// grads[off + j] (+)= sum over the layer's splits of slab_l[split][j], splits in fixed order
struct ReducePlan {
    int64_t off[9];        // parameter offsets of the layers (off[L] = parameter count); at most 8 layers
    int64_t base[8];       // float offset of layer l's slabs
    int nsplit[8];
    int L;
};
__global__ void __launch_bounds__(256) reduce_layers_k(const float *__restrict__ slabs, ReducePlan pl, float *__restrict__ gout, int accumulate) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= pl.off[pl.L]) return;
    int l = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) l += (k < pl.L && j >= pl.off[k]) ? 1 : 0;
    const int64_t size = pl.off[l + 1] - pl.off[l];
    const float *p = slabs + pl.base[l] + (j - pl.off[l]);
    float acc = accumulate ? gout[j] : 0.f;
    const int ns = pl.nsplit[l];
    int k = 0;
    for (; k + 8 <= ns; k += 8) {          // eight loads in flight, added in split order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(k + u) * size];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; k < ns; ++k) acc += p[(int64_t)k * size];
    gout[j] = acc;
}

// ---- host side ----------------------------------------------------------------------------------
template <typename T> struct Work {
    T *x0;
    std::vector<T *> y;  // y[l] = output of layer l-1 (y[0] = x0)
    std::vector<T *> dz; // dz[l] = dL/d(pre-activation of layer l): rows x dims[l + 1] (training only)
    int64_t chunk;
};

template <typename T>
static int carve(bamd_handle *h, int64_t n, bool need_grad, Work<T> &wk) {
    int64_t per_row = h->dims[0] + h->sum_dims + (need_grad ? h->sum_dims : 0);
    // activation workspace budget: large enough that one chunk fills the chip even for wide models
    // (CFD_dense_AE: 42 KB of activations per row); 288 GB of HBM per GPU make 4 GB a small price
    static const int64_t budget = getenv("BALER_AMD_WORKSPACE_MB") ? atoll(getenv("BALER_AMD_WORKSPACE_MB")) << 20 : ((int64_t)4 << 30);
    int64_t chunk = budget / (per_row * (int64_t)sizeof(T));
    chunk = chunk < 1024 ? 1024 : chunk;
    chunk = chunk > (1 << 20) ? (1 << 20) : chunk;
    chunk &= ~(int64_t)63;
    if (chunk > n) chunk = n;
    // every buffer starts on a 16-byte boundary (odd chunk x odd width would leave the next one 4-byte aligned: the kernels'
    // 16-byte operand loads would then rely on the unaligned-access mode and split their transactions)
    auto up4 = [](int64_t e) { return (e + 3) & ~(int64_t)3; };
    int64_t total = 0;
    for (int l = 0; l <= h->L; ++l) total += up4(chunk * h->dims[l]);
    for (int l = 0; l < h->L && need_grad; ++l) total += up4(chunk * h->dims[l + 1]);
    int rc = h->work.ensure((size_t)total * sizeof(T));
    if (rc) return rc;
    T *p = (T *)h->work.p;
    wk.chunk = chunk;
    wk.y.assign(h->L + 1, nullptr);
    for (int l = 0; l <= h->L; ++l) {
        wk.y[l] = p;
        p += up4(chunk * h->dims[l]);
    }
    wk.x0 = wk.y[0];
    wk.dz.assign(h->L, nullptr);
    for (int l = 0; l < h->L && need_grad; ++l) {
        wk.dz[l] = p;
        p += up4(chunk * h->dims[l + 1]);
    }
    return BAMD_OK;
}

template <typename T>
static int stage_input(bamd_handle *h, const void *x, int x_dtype, int64_t row0, int64_t rows, int width,
                       const double *features, T *dst, hipStream_t s) {
    size_t es = x_dtype == BAMD_F64 ? 8 : 4;
    const char *src = (const char *)x + (size_t)row0 * width * es;
    int td = sizeof(T) == 8 ? BAMD_F64 : BAMD_F32;
    if (features) return launch_normalize(src, x_dtype, rows, width, features, dst, td, s);
    return launch_convert(src, x_dtype, dst, td, rows * width, s);
}

template <typename T>
static void launch_fwd_layer(bamd_handle *h, int l, const T *xin, T *yout, int64_t rows, hipStream_t s) {
    const T *P = (const T *)h->params.p;
    int K = h->dims[l], N = h->dims[l + 1];
    Opnd<T> A{xin, K, 1, rows, -1};
    Opnd<T> B{P + h->w_off[l], K, 1, N, -1};
    Epi<T> e{};
    e.out = yout; e.ld = N; e.n_rows = rows; e.n_cols = N;
    e.bias = P + h->b_off[l]; e.act = h->has_act(l) ? 1 : 0;
    launch_gemm<T, EPI_FWD, true, true>(A, B, (int64_t)K, e, rows, N, 1, s);
}

template <typename T>
static int forward_T(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features,
                     int l0, int l1, void *out, int out_dtype, const double *renorm,
                     const uint8_t *int_mask, hipStream_t s) {
    Work<T> wk;
    int rc = carve<T>(h, n, false, wk);
    if (rc) return rc;
    int td = sizeof(T) == 8 ? BAMD_F64 : BAMD_F32;
    int win = h->dims[l0], wout = h->dims[l1];
    size_t oes = out_dtype == BAMD_F64 ? 8 : 4;
    for (int64_t r0 = 0; r0 < n; r0 += wk.chunk) {
        int64_t rows = n - r0 < wk.chunk ? n - r0 : wk.chunk;
        rc = stage_input<T>(h, x, x_dtype, r0, rows, win, features, wk.y[l0], s);
        if (rc) return rc;
        for (int l = l0; l < l1; ++l) launch_fwd_layer<T>(h, l, wk.y[l], wk.y[l + 1], rows, s);
        char *dst = (char *)out + (size_t)r0 * wout * oes;
        if (renorm) {
            if (out_dtype != BAMD_F64) { set_error("decode with features needs a float64 output"); return BAMD_ERR_INVALID; }
            rc = launch_renormalize(wk.y[l1], td, rows, wout, renorm, int_mask, (double *)dst, s);
        } else {
            rc = launch_convert(wk.y[l1], td, dst, out_dtype, rows * wout, s);
        }
        if (rc) return rc;
    }
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

int generic_forward(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features,
                    int l0, int l1, void *out, int out_dtype, const double *renorm,
                    const uint8_t *int_mask, hipStream_t s) {
    if (h->esize == 8) return forward_T<double>(h, x, x_dtype, n, features, l0, l1, out, out_dtype, renorm, int_mask, s);
    return forward_T<float>(h, x, x_dtype, n, features, l0, l1, out, out_dtype, renorm, int_mask, s);
}

template <typename T>
static int forward_loss_T(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features,
                          void *recon, int recon_dtype, double *loss_sum, hipStream_t s) {
    Work<T> wk;
    int rc = carve<T>(h, n, false, wk);
    if (rc) return rc;
    int td = sizeof(T) == 8 ? BAMD_F64 : BAMD_F32;
    int c = h->dims[0];
    rc = h->lossp.ensure(sizeof(double) * 1024);
    if (rc) return rc;
    size_t oes = recon_dtype == BAMD_F64 ? 8 : 4;
    int chunk_i = 0;
    for (int64_t r0 = 0; r0 < n; r0 += wk.chunk, ++chunk_i) {
        int64_t rows = n - r0 < wk.chunk ? n - r0 : wk.chunk;
        rc = stage_input<T>(h, x, x_dtype, r0, rows, c, features, wk.x0, s);
        if (rc) return rc;
        for (int l = 0; l < h->L; ++l) launch_fwd_layer<T>(h, l, wk.y[l], wk.y[l + 1], rows, s);
        int64_t count = rows * c;
        int nblk = (int)((count + 255) / 256 < 1024 ? (count + 255) / 256 : 1024);
        hipLaunchKernelGGL(loss_grad_k<T>, dim3(nblk), dim3(256), 0, s, wk.y[h->L], wk.x0, count, 1.0 / c,
                           (T *)nullptr, (double *)h->lossp.p);
        hipLaunchKernelGGL(loss_final_k<double>, dim3(1), dim3(256), 0, s, (const double *)h->lossp.p, nblk,
                           1.0 / c, loss_sum, chunk_i > 0 ? 1 : 0);
        if (recon) {
            rc = launch_convert(wk.y[h->L], td, (char *)recon + (size_t)r0 * c * oes, recon_dtype, count, s);
            if (rc) return rc;
        }
    }
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

int generic_forward_loss(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features,
                         void *recon, int recon_dtype, double *loss_sum, hipStream_t s) {
    if (h->esize == 8) return forward_loss_T<double>(h, x, x_dtype, n, features, recon, recon_dtype, loss_sum, s);
    return forward_loss_T<float>(h, x, x_dtype, n, features, recon, recon_dtype, loss_sum, s);
}

// short-side weight-gradient kernels: usable when every layer has a side of 1, 2, 4, 7 or 13 16-column tiles
static int short_tiles(int cols) {
    const int t = (cols + 15) / 16;
    return (t == 1 || t == 2 || t == 4 || t == 7 || t == 13) ? t : 0;
}
struct ShortPlan {
    bool ok = false;
    bool p_is_n[8], wide[8];
    int pt[8], ncol[8], nsplit[8];
    int64_t rps[8];
    ReducePlan rp;
    int64_t total = 0;     // slab floats
};
static ShortPlan plan_short_dw(const bamd_handle *h, int64_t rows) {
    ShortPlan pl;
    const char *e = getenv("BALER_AMD_SHORT_DW");
    constexpr int64_t min_rows = 128;   // C4 at 512 frames: 796 -> 580 us per pass; 60 frames: no difference
    if ((e && e[0] == '0') || h->L > 8 || rows < min_rows) return pl;
    constexpr int TQ = 2;
    int64_t base = 0;
    for (int l = 0; l < h->L; ++l) {
        const int K = h->dims[l], N = h->dims[l + 1];
        // P = the side with fewer tiles among those that fit (more Q tiles = more workgroup columns)
        const int tk = short_tiles(K + 1), tn = short_tiles(N);
        if (!tk && !tn) return pl;
        const bool pn = tk == 0 || (tn != 0 && N < K + 1);
        pl.p_is_n[l] = pn;
        pl.pt[l] = pn ? tn : tk;
        const int cq = pn ? K + 1 : N;
        // 13 tiles on the short side and 16-byte rows: the kernel with 16-byte operand loads (one 64-column group of Q per wave,
        // one workgroup per CU); otherwise two 16-column tiles of Q per wave, two workgroups per CU
        pl.wide[l] = pl.pt[l] == 13 && K % 4 == 0 && N % 4 == 0 && (pn ? N : K) >= 192;
        const int per_wg = pl.wide[l] ? 256 : 16 * 4 * TQ;
        pl.ncol[l] = (cq + per_wg - 1) / per_wg;
        // every split at least 64 rows and a multiple of 16
        int64_t ns = (pl.wide[l] ? 256 : 512) / pl.ncol[l];        // rounded DOWN: one workgroup more than the chip holds doubles the time
        ns = ns > rows / 64 ? rows / 64 : ns;
        ns = ns < 1 ? 1 : (ns > 256 ? 256 : ns);
        int64_t rps = ((rows + ns - 1) / ns + 15) & ~(int64_t)15;
        const int64_t wmax = K > N ? K : N;
        if (rps * wmax >= ((int64_t)1 << 29)) return pl;       // 32-bit byte offsets inside a split
        ns = (rows + rps - 1) / rps;
        pl.nsplit[l] = (int)ns;
        pl.rps[l] = rps;
        pl.rp.off[l] = h->w_off[l];
        base = (base + 3) & ~(int64_t)3;       // 16-byte aligned slabs: dw_wide_k stores float4 (its layers have N K + N = 0 mod 4)
        pl.rp.base[l] = base;
        pl.rp.nsplit[l] = (int)ns;
        base += ns * ((int64_t)N * K + N);
    }
    pl.rp.off[h->L] = h->nparams;
    pl.rp.L = h->L;
    pl.total = base;
    pl.ok = true;
    return pl;
}
template <int PT>
static void launch_dw_short(bool p_is_n, const float *dz, const float *xm, int N, int K, int64_t rows, int64_t rps, float *slab, dim3 grid,
                            hipStream_t s) {
    const int64_t size = (int64_t)N * K + N;
    if (p_is_n) hipLaunchKernelGGL((dw_short_k<PT, 2, true>), grid, dim3(256), 0, s, dz, xm, N, K, rows, rps, slab, size);
    else hipLaunchKernelGGL((dw_short_k<PT, 2, false>), grid, dim3(256), 0, s, dz, xm, N, K, rows, rps, slab, size);
}
static void run_dw_short(const ShortPlan &pl, int l, const float *dz, const float *xm, int N, int K, int64_t rows, float *slabs, bool bf16,
                         bool dz16, hipStream_t s) {
    const dim3 grid((unsigned)pl.ncol[l], (unsigned)pl.nsplit[l]);
    float *slab = slabs + pl.rp.base[l];
    if (pl.wide[l]) {
        const int64_t size = (int64_t)N * K + N;
        if (bf16) {      // BAMD_MODE_BF16 handles: the two wide weight gradients on the bf16 MFMA (BALER_AMD_BF16_WIDE_TRAIN=0: float32)
            if (pl.p_is_n[l]) hipLaunchKernelGGL((dw_wide_bf16_k<true>), grid, dim3(256), 0, s, dz, xm, N, K, rows, pl.rps[l], slab, size);
            else if (dz16) hipLaunchKernelGGL((dw_wide_bf16_k<false, true>), grid, dim3(256), 0, s, dz, xm, N, K, rows, pl.rps[l], slab, size);
            else hipLaunchKernelGGL((dw_wide_bf16_k<false>), grid, dim3(256), 0, s, dz, xm, N, K, rows, pl.rps[l], slab, size);
            return;
        }
        if (pl.p_is_n[l]) hipLaunchKernelGGL((dw_wide_k<true>), grid, dim3(256), 0, s, dz, xm, N, K, rows, pl.rps[l], slab, size);
        else hipLaunchKernelGGL((dw_wide_k<false>), grid, dim3(256), 0, s, dz, xm, N, K, rows, pl.rps[l], slab, size);
        return;
    }
    if (bf16 && pl.pt[l] == 13) {      // a wide layer whose rows are not 16-byte aligned (625 columns), BAMD_MODE_BF16 handle
        const int64_t size = (int64_t)N * K + N;
        if (pl.p_is_n[l]) hipLaunchKernelGGL((dw_short_bf16_k<13, 2, true>), grid, dim3(256), 0, s, dz, xm, N, K, rows, pl.rps[l], slab, size);
        else hipLaunchKernelGGL((dw_short_bf16_k<13, 2, false>), grid, dim3(256), 0, s, dz, xm, N, K, rows, pl.rps[l], slab, size);
        return;
    }
    switch (pl.pt[l]) {
    case 1: launch_dw_short<1>(pl.p_is_n[l], dz, xm, N, K, rows, pl.rps[l], slab, grid, s); break;
    case 2: launch_dw_short<2>(pl.p_is_n[l], dz, xm, N, K, rows, pl.rps[l], slab, grid, s); break;
    case 4: launch_dw_short<4>(pl.p_is_n[l], dz, xm, N, K, rows, pl.rps[l], slab, grid, s); break;
    case 7: launch_dw_short<7>(pl.p_is_n[l], dz, xm, N, K, rows, pl.rps[l], slab, grid, s); break;
    default: launch_dw_short<13>(pl.p_is_n[l], dz, xm, N, K, rows, pl.rps[l], slab, grid, s); break;
    }
}

// ---- all weight gradients of a SMALL float32 batch in one launch ----------------------------------------------------------------------
// The reference trains its wide models with batches of 1 .. 85 rows (CFD_project_still: 60): eight split-K GEMM launches + a slab
// reduction for ~0.3 GFLOP were 101 + 5 of the step's 220 us.  Here ONE wave owns a 16 x 16 tile of [dW | db] = dZ^T [X | 1] of one
// layer and contracts over ALL rows of the batch (operands in MFMA layout straight from the row-major matrices, as dw_short_k: the
// reduction index of v_mfma_f32_16x16x4_f32 is the row); results go straight to the canonical gradient vector (= or +=).
static int64_t dw_small_rows() {      // BALER_AMD_DW_SMALL_ROWS (C4: 512 rows 226 -> 180 us, 1024 rows 235 -> 237 with the one launch)
    return env_ll("BALER_AMD_DW_SMALL_ROWS", 768);
}
template <typename T> struct SmallDwPlan {
    const T *dz[8], *x[8];
    int N[8], K[8], kt[8], tile0[9];       // tiles of layer l: [tile0[l], tile0[l + 1]) = nt(l) x kt(l), index = nt * kt(l) + kt
    int64_t w_off[8], b_off[8];
    int L;
};
// `ad.on`: the optimiser step of exactly these parameters in the same launch (elementwise.hip adam_k's arithmetic, operation for
// operation: bamd_train_step == bamd_fwd_bwd + bamd_adam_step to the last bit), the refresh of their packed copies included
template <typename T> struct SmallAdamT {
    T *p, *pcopy, *m, *v, *packed;
    const int *sc_off, *sc_idx;
    double *loss_accum;
    double b1, b2, eps, step_size, bc2_sqrt;
    int64_t np;
    int on;
};
using SmallAdam = SmallAdamT<float>;
// The LAST workgroup is loss_final_k: the fixed-order sum of the forward launch's loss partials -> grads[np] (and the caller's running
// loss when the optimiser step rides along): one launch fewer per step.
template <typename T>
__global__ void __launch_bounds__(256) dw_small_all_k(SmallDwPlan<T> pl, int64_t rows, T *__restrict__ grads, int accumulate, SmallAdamT<T> ad,
                                                      const double *__restrict__ loss_part, int nloss, double loss_scale, int64_t np) {
    using v4 = typename MF<T>::v4;
    if (blockIdx.x == gridDim.x - 1) {
        __shared__ double sh[256];
        double s = 0.0;
        for (int k = threadIdx.x; k < nloss; k += 256) s += loss_part[k];
        sh[threadIdx.x] = s;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            s = sh[0] * loss_scale;
            const T lv = accumulate ? (T)((double)grads[np] + s) : (T)s;
            grads[np] = lv;
            if (ad.on && ad.loss_accum) *ad.loss_accum += (double)lv;
        }
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= pl.tile0[pl.L]) return;
    int l = 0;
#pragma unroll
    for (int j = 1; j < 8; ++j) if (j < pl.L && tile >= pl.tile0[j]) l = j;
    const int N = pl.N[l], K = pl.K[l], KT = pl.kt[l];
    const int t = tile - pl.tile0[l], nt = t / KT, kt = t - nt * KT;
    const int ncol = 16 * nt + i, kcol = 16 * kt + i;                 // this lane's dZ column (A operand) / X column (B operand)
    const T *pa = pl.dz[l] + (ncol < N ? ncol : N - 1);
    const T *pb = pl.x[l] + (kcol < K ? kcol : K - 1);
    const bool a_live = ncol < N, b_live = kcol < K, b_one = kcol == K;
    v4 acc = (v4){0, 0, 0, 0};
    T a[4], b[4];
    auto load = [&](int64_t r0) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int64_t r = r0 + 4 * s4 + g;
            const bool ok = r < rows;
            const int64_t rr = ok ? r : rows - 1;
            const T av = pa[rr * N], bv = pb[rr * K];
            a[s4] = (ok && a_live) ? av : (T)0;
            b[s4] = b_one ? (T)1 : (b_live ? bv : (T)0);
        }
    };
    load(0);
    for (int64_t r0 = 0; r0 < rows; r0 += 16) {
        T a0[4], b0[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) { a0[s4] = a[s4]; b0[s4] = b[s4]; }
        if (r0 + 16 < rows) load(r0 + 16);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) acc = MF<T>::mma(a0[s4], b0[s4], acc);
    }
    // C map: register r of lane (i, g) = [dZ column 16 nt + crow(r)][X column 16 kt + i] (crow: 4 g + r in float32, g + 4 r in float64)
    int64_t pidx[4];
    T gv[4], pm[4], pv[4], pp[4];
    int so0[4], so1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {      // every load of the four elements first (the optimiser state is four dependent round trips otherwise)
        const int n = 16 * nt + MF<T>::crow(r, lane);
        pidx[r] = -1;
        if (n < N && kcol < K) pidx[r] = pl.w_off[l] + (int64_t)n * K + kcol;
        else if (n < N && kcol == K) pidx[r] = pl.b_off[l] + n;
        const int64_t q = pidx[r] < 0 ? 0 : pidx[r];
        gv[r] = accumulate ? grads[q] + acc[r] : acc[r];
        if (ad.on) {
            pm[r] = ad.m[q]; pv[r] = ad.v[q]; pp[r] = ad.p[q];
            so0[r] = ad.packed ? ad.sc_off[q] : 0; so1[r] = ad.packed ? ad.sc_off[q + 1] : 0;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (pidx[r] < 0) continue;
        grads[pidx[r]] = gv[r];
        if (ad.on) {
            const double gi = (double)gv[r];
            double mi = (double)pm[r], vi = (double)pv[r];
            mi = mi + (gi - mi) * (1.0 - ad.b1);
            vi = vi * ad.b2 + (1.0 - ad.b2) * gi * gi;
            const double denom = sqrt(vi) / ad.bc2_sqrt + ad.eps;
            const double pn = (double)pp[r] - ad.step_size * (mi / denom);
            ad.m[pidx[r]] = (T)mi;
            ad.v[pidx[r]] = (T)vi;
            ad.p[pidx[r]] = (T)pn;
            if (ad.pcopy) ad.pcopy[pidx[r]] = (T)pn;
            for (int k = so0[r]; k < so1[r]; ++k) ad.packed[ad.sc_idx[k]] = (T)pn;
        }
    }
}

// Which launches a chunk of `rows` rows takes.  ONE definition: the optimiser-step precheck of fwd_bwd_T and its body must agree, or a
// step whose weight gradients do not run on dw_small_all_k would return BAMD_OK without having applied Adam.
struct ChunkRoute {
    bool wide;        // row-local work on the two fused wide launches of fused.hip
    bool small_wide;  // ... on their split float32 forms (also on a BF16 handle)
    bool bf16_dw;     // BF16 handle whose weight gradients keep the per-layer launches (bf16 operands where the layer is wide)
    bool dw_small;    // every weight gradient (+ the loss sum, + Adam when asked) in ONE launch: dw_small_all_k
};
template <typename T>
static ChunkRoute route_chunk(const bamd_handle *h, int64_t rows) {
    ChunkRoute r{};
    // (BALER_AMD_WIDE_LAYERWISE_ROWS, default 0 = never: up to that many rows the one-tile-per-workgroup kernels of this file would take
    // every layer of a wide model too -- measured: C4 optimiser step at 60 rows 148 us against 125 on the split launches of fused.hip)
    const int64_t lw_rows = env_ll("BALER_AMD_WIDE_LAYERWISE_ROWS", 0);
    r.wide = sizeof(T) == 4 && fused_wide_train(h) && !(rows <= lw_rows && rows <= gemm_small_rows() && h->mode != BAMD_MODE_BF16);
    r.small_wide = r.wide && fused_wide_small(h, rows);      // (a BF16 handle's small batches run the float32 split launches)
    r.bf16_dw = h->mode == BAMD_MODE_BF16 && !env_off("BALER_AMD_BF16_WIDE_TRAIN") && !r.small_wide;
    r.dw_small = !r.bf16_dw && rows <= dw_small_rows() && h->L <= 8;
    return r;
}

template <typename T>
static int fwd_bwd_T(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features,
                     void *grads_v, const void *latent_grad, hipStream_t s, const SmallAdam *adam = nullptr) {
    // `adam`: the caller wants the optimiser step in the weight-gradient launch: one chunk of a small float32 batch whose weight
    // gradients run on dw_small_all_k (checked before the workspace is carved: a large batch of a fused shape only asks)
    if (adam && !(sizeof(T) == 4 && route_chunk<T>(h, n).dw_small)) return BAMD_ERR_UNSUPPORTED;
    Work<T> wk;
    int rc = carve<T>(h, n, true, wk);
    if (rc) return rc;
    if (adam && wk.chunk < n) return BAMD_ERR_UNSUPPORTED;
    const T *P = (const T *)h->params.p;
    T *grads = (T *)grads_v;
    int c = h->dims[0];
    const int64_t np = h->nparams;
    // number of row splits of the weight-gradient product (slab memory bounded to ~256 MB)
    int64_t max_split = (int64_t)(1u << 28) / (np * (int64_t)sizeof(T));
    max_split = max_split < 1 ? 1 : (max_split > 64 ? 64 : max_split);
    rc = h->lossp.ensure(sizeof(double) * 1024);
    if (rc) return rc;
    int chunk_i = 0;
    for (int64_t r0 = 0; r0 < n; r0 += wk.chunk, ++chunk_i) {
        int64_t rows = n - r0 < wk.chunk ? n - r0 : wk.chunk;
        int64_t nsplit = (rows + 255) / 256;
        nsplit = nsplit > max_split ? max_split : nsplit;
        int64_t rps = ((rows + nsplit - 1) / nsplit + 15) & ~(int64_t)15;
        nsplit = (rows + rps - 1) / rps;
        ShortPlan sp;
        if constexpr (sizeof(T) == 4) sp = plan_short_dw(h, rows);
        rc = h->slabs.ensure(sp.ok ? (size_t)sp.total * sizeof(float) : (size_t)(nsplit * np) * sizeof(T));
        if (rc) return rc;
        T *slabs = (T *)h->slabs.p;
        // rows that already have the compute type are used where they lie (no staging copy: 328 MB per 32k CFD frames)
        const bool in_place = !features && x_dtype == (sizeof(T) == 8 ? BAMD_F64 : BAMD_F32);
        const T *x0 = in_place ? (const T *)x + r0 * c : wk.x0;
        if (!in_place) {
            rc = stage_input<T>(h, x, x_dtype, r0, rows, c, features, wk.x0, s);
            if (rc) return rc;
        }
        // wide models in float32: the row-local work (forward, loss, input-gradient chain) as two fused launches (fused.hip)
        const ChunkRoute route = route_chunk<T>(h, rows);
        const bool wide = route.wide, small_wide = route.small_wide;
        // BF16 handles: dL/drecon stored as bfloat16 when its two readers take it that way (BALER_AMD_BF16_DZ16=0: float32)
        bool dz16 = false;
        if constexpr (sizeof(T) == 4) {
            const char *e = getenv("BALER_AMD_BF16_WIDE_TRAIN"), *e16 = getenv("BALER_AMD_BF16_DZ16");
            const int ll = h->L - 1;
            dz16 = wide && !small_wide && h->mode == BAMD_MODE_BF16 && !(e && e[0] == '0') && !(e16 && e16[0] == '0') && sp.ok && sp.wide[ll] &&
                   !sp.p_is_n[ll] && h->dims[ll + 1] % 4 == 0;
            if (wide) fused_wide_set_dz16(h, dz16);
        }
        int nblk = 0;
        if (wide) {
            rc = h->lossp.ensure(sizeof(double) * 4096);
            if (rc) return rc;
            rc = fused_wide_train_forward(h, (const float *)x0, rows, (float *const *)wk.y.data(), (float *)wk.dz[h->L - 1],
                                          (double *)h->lossp.p, &nblk, s);
            if (rc) return rc;
        } else {
            launch_fwd_layer<T>(h, 0, x0, wk.y[1], rows, s);
            for (int l = 1; l + 1 < h->L; ++l) launch_fwd_layer<T>(h, l, wk.y[l], wk.y[l + 1], rows, s);
            {   // last layer with the loss fused into its store: dz = 2 (r - x) / C, one loss partial per workgroup (the separate
                // loss pass re-read r and x and wrote dz: 1 GB for 32k CFD frames)
                const int l = h->L - 1, K = h->dims[l], N = h->dims[l + 1];
                const bool big = rows >= 96 && N >= 96;
                const int64_t tile = rows <= gemm_small_rows() ? 16 : (big ? 128 : 64);      // (as launch_gemm picks)
                nblk = (int)(((N + tile - 1) / tile) * ((rows + tile - 1) / tile));
                rc = h->lossp.ensure(sizeof(double) * (size_t)(nblk > 1024 ? nblk : 1024));
                if (rc) return rc;
                Opnd<T> A{wk.y[l], K, 1, rows, -1};
                Opnd<T> B{P + h->w_off[l], K, 1, N, -1};
                Epi<T> e{};
                e.out = wk.dz[l]; e.ld = N; e.n_rows = rows; e.n_cols = N;
                e.bias = P + h->b_off[l]; e.act = h->has_act(l) ? 1 : 0;
                e.xref = x0; e.loss_part = (double *)h->lossp.p; e.grad_scale = 2.0 / c;
                launch_gemm<T, EPI_FWD_LOSS, true, true>(A, B, (int64_t)K, e, rows, N, 1, s);
            }
        }
        const bool dw_small = route.dw_small;      // every weight gradient in ONE launch, behind the input-gradient chain
        if (adam && !dw_small) { set_error("fwd_bwd_T: optimiser step asked of a chunk that is not on dw_small_all_k"); return BAMD_ERR_UNSUPPORTED; }
        if (!dw_small)      // (that launch also sums the loss partials)
            hipLaunchKernelGGL(loss_final_k<T>, dim3(1), dim3(256), 0, s, (const double *)h->lossp.p, nblk, 1.0 / c,
                               grads + np, chunk_i > 0 ? 1 : 0);
        if (wide) {
            rc = fused_wide_train_backward(h, rows, (float *const *)wk.y.data(), (float *const *)wk.dz.data(),
                                           latent_grad ? (const float *)latent_grad + r0 * h->dims[h->L / 2] : nullptr, s);
            if (rc) return rc;
        }
        // small float32 batches of a model on the fused row-local launches: every weight gradient in ONE launch, straight into `grads`
        for (int l = h->L - 1; l >= 0; --l) {
            int K = h->dims[l], N = h->dims[l + 1];
            const T *dz = wk.dz[l];
            // [dW | db] = dZ^T [X | 1], reduced over this chunk's rows in nsplit fixed slabs
            if (dw_small) {
            } else if (sp.ok) {
                if constexpr (sizeof(T) == 4) {
                    const char *e = getenv("BALER_AMD_BF16_WIDE_TRAIN");
                    const bool bf16 = wide && !small_wide && h->mode == BAMD_MODE_BF16 && !(e && e[0] == '0');
                    run_dw_short(sp, l, dz, l == 0 ? x0 : wk.y[l], N, K, rows, slabs, bf16, dz16 && l == h->L - 1, s);
                }
            } else {
                Opnd<T> A{dz, 1, N, N, -1};
                Opnd<T> B{l == 0 ? x0 : wk.y[l], 1, K, K, K};
                Epi<T> e{};
                e.n_rows = N; e.kin = K; e.gw = slabs + h->w_off[l]; e.gb = slabs + h->b_off[l];
                e.slab_stride = np; e.rows_per_split = rps;
                launch_gemm<T, EPI_DW, false, false>(A, B, rows, e, N, K + 1, (unsigned)nsplit, s);
            }
            if (l > 0 && !wide) {
                // dZ_{l-1} = (dZ_l W_l) * lrelu'(Y_{l-1})
                Opnd<T> A{dz, N, 1, rows, -1};
                Opnd<T> B{P + h->w_off[l], 1, K, K, -1};
                Epi<T> e{};
                e.out = wk.dz[l - 1]; e.ld = K; e.n_rows = rows; e.n_cols = K;
                e.ymask = h->has_act(l - 1) ? wk.y[l] : nullptr; e.ld_mask = K;
                if (latent_grad && l == h->L / 2) { e.add = (const T *)latent_grad + r0 * K; e.ld_add = K; }   // dL/dz of the caller's regulariser
                launch_gemm<T, EPI_DX, true, false>(A, B, (int64_t)N, e, rows, K, 1, s);
            }
        }
        if (dw_small) {      // (any model on this path: the row-major activations / gradients are the same)
            SmallDwPlan<T> pl{};
            pl.L = h->L;
            int t0 = 0;
            for (int l = 0; l < h->L; ++l) {
                pl.dz[l] = (const T *)wk.dz[l];
                pl.x[l] = l == 0 ? (const T *)x0 : (const T *)wk.y[l];
                pl.N[l] = h->dims[l + 1]; pl.K[l] = h->dims[l];
                pl.kt[l] = (h->dims[l] + 1 + 15) / 16;
                pl.tile0[l] = t0;
                t0 += ((h->dims[l + 1] + 15) / 16) * pl.kt[l];
                pl.w_off[l] = h->w_off[l]; pl.b_off[l] = h->b_off[l];
            }
            pl.tile0[h->L] = t0;
            SmallAdamT<T> sa{};
            if constexpr (sizeof(T) == 4) { if (adam) sa = *adam; }
            hipLaunchKernelGGL(dw_small_all_k<T>, dim3((unsigned)((t0 + 3) / 4 + 1)), dim3(256), 0, s, pl, rows, grads, chunk_i > 0 ? 1 : 0, sa,
                               (const double *)h->lossp.p, nblk, 1.0 / c, (int64_t)np);
        }
        if (dw_small) {
        } else if (sp.ok) {
            if constexpr (sizeof(T) == 4)
                hipLaunchKernelGGL(reduce_layers_k, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, (const float *)slabs, sp.rp, grads,
                                   chunk_i > 0 ? 1 : 0);
        } else {
            hipLaunchKernelGGL(reduce_slabs_k<T>, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, slabs,
                               (int)nsplit, np, np, grads, chunk_i > 0 ? 1 : 0);
        }
    }
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

int generic_fwd_bwd(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features,
                    void *grads, hipStream_t s, const void *latent_grad) {
    if (h->esize == 8) return fwd_bwd_T<double>(h, x, x_dtype, n, features, grads, latent_grad, s);
    return fwd_bwd_T<float>(h, x, x_dtype, n, features, grads, latent_grad, s);
}

// bamd_train_step of a small float32 batch of a wide model: forward + loss + backward + weight gradients + Adam, the optimiser step inside
// the one weight-gradient launch (BAMD_ERR_UNSUPPORTED: not such a batch -- the caller runs fwd_bwd and the Adam kernel)
int generic_small_train_step(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads, void *params,
                             void *m, void *v, const bamd_adam &hp, double *loss_accum, hipStream_t s) {
    if (h->esize != 4) return BAMD_ERR_UNSUPPORTED;
    if (env_off("BALER_AMD_SMALL_ADAM")) return BAMD_ERR_UNSUPPORTED;
    SmallAdam sa{};
    sa.p = (float *)params; sa.pcopy = (float *)h->params.p; sa.m = (float *)m; sa.v = (float *)v;
    void *packed = nullptr;
    fused_scatter(h, &sa.sc_off, &sa.sc_idx, &packed);
    sa.packed = (float *)packed;
    sa.loss_accum = loss_accum;
    sa.b1 = hp.beta1; sa.b2 = hp.beta2; sa.eps = hp.eps;                    // the scalars of launch_adam (elementwise.hip)
    sa.step_size = hp.lr / (1.0 - pow(hp.beta1, (double)hp.step));
    sa.bc2_sqrt = sqrt(1.0 - pow(hp.beta2, (double)hp.step));
    sa.np = h->nparams;
    sa.on = 1;
    return fwd_bwd_T<float>(h, x, x_dtype, n, features, grads, nullptr, s, &sa);
}

template <typename T>
static int act_means_T(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features,
                       double *out, int max_nodes, hipStream_t s) {
    Work<T> wk;
    int rc = carve<T>(h, n, false, wk);
    if (rc) return rc;
    if (wk.chunk < n) { set_error("activation_means: batch larger than one workspace chunk"); return BAMD_ERR_UNSUPPORTED; }
    int nact = h->L - 2;
    hipLaunchKernelGGL(fill_nan_k, dim3((nact * max_nodes + 255) / 256), dim3(256), 0, s, out, nact * max_nodes);
    rc = stage_input<T>(h, x, x_dtype, 0, n, h->dims[0], features, wk.x0, s);
    if (rc) return rc;
    int r = 0;
    for (int l = 0; l < h->L; ++l) {
        launch_fwd_layer<T>(h, l, wk.y[l], wk.y[l + 1], n, s);
        if (h->has_act(l)) {
            if (h->dims[l + 1] > max_nodes) { set_error("activation_means: max_nodes too small"); return BAMD_ERR_INVALID; }
            hipLaunchKernelGGL(colmean_k<T>, dim3(h->dims[l + 1]), dim3(256), 0, s, wk.y[l + 1], n, h->dims[l + 1],
                               out + (int64_t)r * max_nodes);
            ++r;
        }
    }
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

int generic_activation_means(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features,
                             double *out, int max_nodes, hipStream_t s) {
    if (h->esize == 8) return act_means_T<double>(h, x, x_dtype, n, features, out, max_nodes, s);
    return act_means_T<float>(h, x, x_dtype, n, features, out, max_nodes, s);
}

}  // namespace bamd
