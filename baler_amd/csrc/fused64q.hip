// The 4-row fp64 chain (chain64q_kernel) in its own translation unit: 14 fully unrolled instantiations compile beside fused64.hip instead of
// behind it (the two files take ~3 minutes each; together they took 6).  Geometry: fused64_net.hpp.  Launched by Impl64::step (fused64.hip).
#include "fused64_net.hpp"

#include <cstdint>
#include <utility>

namespace bamd {
namespace {

// ---- the 4-row chain: the reference's own regime (fp64, batch_size = 512: models.py:128-136, CMS_project_v1_config.py:12) -------------
// chain64_kernel gives a 16-row block to a workgroup: a 512-row batch occupies 32 of the 256 CUs and every workgroup runs 2,110
// v_mfma_f64_16x16x4 of 64 cycles = 34k cycles per wave whatever the chip size (28.7 us of the step's 41).  v_mfma_f64_4x4x4_4b_f64 computes
// four independent 4 x 4 x 4 products per instruction at the same FLOP rate (tools/probe/mfma64_4x4_probe.hip: A lane 16 k + 4 b + i,
// B lane 16 k + 4 b + j, D lane 16 i + 4 b + j; 16.6 - 17.7 cycles per instruction on independent accumulators, 21.7 on one): with
// A = 16 output features (block b, row i: feature 4 b + i) x 4 contraction indices and B = the FOUR batch rows replicated over the
// blocks, a workgroup needs only four rows -- 128 workgroups carry the 512-row batch and a workgroup's chain is a quarter of the MFMA
// work: 606 instructions per wave.  The fp32 twin is lat4_chain_kernel (fused.hip); like there
//   * what bounds a workgroup is its WEIGHT STREAM (every workgroup reads every weight once, forward and transposed: 1 MB through one
//     CU's vector-memory path), so all four waves stream in every GEMM: a GEMM's 16-feature groups are dealt to the waves (group w, w + 4,
//     ..), and a GEMM of one or two groups (en4, de4 and their transposes) splits its contraction over the waves instead and adds the
//     partial sums through LDS in a fixed order;
//   * the fragments ([k / 8][group], 1 KiB = two MFMAs) flow through a register ring D fragments ahead of their use, across GEMM
//     boundaries; groups / contraction steps a wave does not own are requested through an out-of-range offset (zeros, no traffic);
//   * activations live in LDS as [4 rows][features] (X_l: the next GEMM's B operand and the backward masks; dZ_l: ping-pong), the B
//     operand of a fragment's two MFMAs is two ds_read_b64 that serve all of the wave's groups of that contraction step;
//   * results go to the SAME global images as chain64_kernel's ([16-row block][slot][16 rows]; this workgroup fills rows 4 q .. 4 q + 3 of
//     every slot), so dw64_kernel (weight-gradient tiles + Adam) is unchanged; the loss partial is per workgroup (4 per block).
#ifdef BAMD_Q4_TRACE   // debug build: shader-clock stamps of workgroup 0, wave 0 at every GEMM boundary (tools/q4_trace.py)
__device__ unsigned long long g_q4_trace[32];
#define Q4_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_q4_trace[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define Q4_T(i) do {} while (0)
#endif
#ifndef BAMD_Q4_RING
#define BAMD_Q4_RING 24
#endif
template <class N> struct Q4 {
    static constexpr int NG = 15, D = BAMD_Q4_RING, KB = 3;       // chain GEMMs; fragment ring depth; B operand reads run KB fragments ahead
    __host__ __device__ static constexpr int G(int g) { return N::q_groups(g); }
    __host__ __device__ static constexpr int KS8(int g) { return N::q_ks8(g); }
    __host__ __device__ static constexpr int P(int g) { return G(g) >= 3 ? 1 : (G(g) == 2 ? 2 : 4); }      // contraction parts
    __host__ __device__ static constexpr int NGM(int g) { return P(g) == 1 ? (G(g) + 3) / 4 : 1; }         // most groups of one wave
    __host__ __device__ static constexpr int KS8P(int g) { return (KS8(g) + P(g) - 1) / P(g); }             // contraction steps of one wave
    __host__ __device__ static constexpr int steps(int g) { return KS8P(g) * NGM(g); }                      // fragments of one wave
    __host__ __device__ static constexpr int pos(int g) { int s = 0; for (int j = 0; j < g; ++j) s += steps(j); return s; }
    static constexpr int total = pos(NG);
    __host__ __device__ static constexpr int gemm_at(int S) { int g = 0; for (int j = 0; j < NG; ++j) if (S >= pos(j)) g = j; return g; }
    // LDS images [4 rows][stride]: stride = 16 x groups rounded up to 32 m + 2 doubles (the four rows of a B read then sit in different
    // banks); X_0 .. X_7, two dZ buffers, the partial sums [part][group][64 lanes], the biases
    __host__ __device__ static constexpr int stride_for(int d) { return (16 * tiles(d) + 8 + 31) / 32 * 32 + 2; }
    __host__ __device__ static constexpr int xs(int l) { return stride_for(N::dim(l)); }
    __host__ __device__ static constexpr int xo(int l) { int s = 0; for (int j = 0; j < l; ++j) s += 4 * xs(j); return s; }
    static constexpr int zs = stride_for(200);
    __host__ __device__ static constexpr int zo(int i) { return xo(8) + i * 4 * zs; }
    static constexpr int po = zo(2);
    static constexpr int bo = po + 4 * 64;
    static constexpr int lds_doubles = bo + N::qb_off(N::L);
};
typedef double d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }

template <class N, int S_>
__device__ __forceinline__ void q_issue(d2 (&ring)[Q4<N>::D], __amdgpu_buffer_rsrc_t rs, int lane16, int wave) {
    using T = Q4<N>;
    if constexpr (S_ < T::total) {
        constexpr int g = T::gemm_at(S_), i = S_ - T::pos(g), k8i = i / T::NGM(g), gi = i % T::NGM(g), G = T::G(g), P = T::P(g);
        const int part = P > 1 ? wave / G : 0;                       // wave-uniform
        const int grp = P > 1 ? wave % G : wave + 4 * gi;
        const int k8 = part * T::KS8P(g) + k8i;
        const int vo = (k8 < T::KS8(g) && grp < G) ? lane16 : 0x7F000000;      // not this wave's: a zero fragment, no traffic
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, vo, (N::q_frag_off(g) + k8 * G + grp) * 1024, 0);
        ring[S_ % T::D] = __builtin_bit_cast(d2, v);
    }
}
template <class N, int... S_>
__device__ __forceinline__ void q_prologue(d2 (&ring)[Q4<N>::D], __amdgpu_buffer_rsrc_t rs, int lane16, int wave, std::integer_sequence<int, S_...>) {
    (q_issue<N, S_>(ring, rs, lane16, wave), ...);
    __builtin_amdgcn_sched_barrier(0);
}
template <class N, int g, int I>
__device__ __forceinline__ void q_step(double (&acc)[Q4<N>::NGM(g)][2], d2 (&xb)[Q4<N>::KB + 1], const double *brow, d2 (&ring)[Q4<N>::D],
                                       __amdgpu_buffer_rsrc_t rs, int lane16, int wave) {
    using T = Q4<N>;
    constexpr int NGM = T::NGM(g), k8i = I / NGM, gi = I % NGM, S0 = T::pos(g);
    if constexpr (gi == 0 && k8i + T::KB < T::KS8P(g)) xb[(k8i + T::KB) % (T::KB + 1)] = (d2){brow[8 * (k8i + T::KB)], brow[8 * (k8i + T::KB) + 4]};
    const d2 a = ring[(S0 + I) % T::D], b = xb[k8i % (T::KB + 1)];
    acc[gi][0] = mfma4(a[0], b[0], acc[gi][0]);      // two accumulators per group: a dependent 4x4x4 costs 21.7 cycles, not 17
    acc[gi][1] = mfma4(a[1], b[1], acc[gi][1]);
    q_issue<N, S0 + I + T::D>(ring, rs, lane16, wave);
    __builtin_amdgcn_sched_barrier(0);
}
template <class N, int g, int... I>
__device__ __forceinline__ void q_gemm(double (&out)[Q4<N>::NGM(g)], const double *brow /* this lane's B row at this wave's first step */,
                                       d2 (&ring)[Q4<N>::D], __amdgpu_buffer_rsrc_t rs, int lane16, int wave, std::integer_sequence<int, I...>) {
    using T = Q4<N>;
    double acc[T::NGM(g)][2];
#pragma unroll
    for (int i = 0; i < T::NGM(g); ++i) { acc[i][0] = 0.0; acc[i][1] = 0.0; }
    d2 xb[T::KB + 1];
#pragma unroll
    for (int k = 0; k < T::KB && k < T::KS8P(g); ++k) xb[k] = (d2){brow[8 * k], brow[8 * k + 4]};
    (q_step<N, g, I>(acc, xb, brow, ring, rs, lane16, wave), ...);
#pragma unroll
    for (int i = 0; i < T::NGM(g); ++i) out[i] = acc[i][0] + acc[i][1];
}

template <int F, int Z, bool RT = false>
__global__ void __launch_bounds__(256) chain64q_kernel(const double *__restrict__ qpacked /* the Q region of the packed buffer */,
                                                       const void *__restrict__ xin, int in_f64, int64_t n, const double *__restrict__ feats,
                                                       double *__restrict__ imgs, double *__restrict__ loss_part, int fr) {
    using N = Net64<F, Z>;
    using T = Q4<N>;
    static_assert(F % 16 != 0 && F <= 127, "input rows: up to two 64-slot halves x 4 rows per thread");
    __shared__ __attribute__((aligned(16))) double lds[T::lds_doubles];
    __shared__ double loss_lds[4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the four workgroups of a 16-row block share its image lines (32 of the 128 bytes of every slot each): workgroup ids equal mod 8,
    // i.e. (workgroups are dealt to the XCDs round-robin) the same XCD and L2 -- placement only, nothing depends on it
    const int wg = blockIdx.x, nwg = gridDim.x;
    int blk = wg >> 2, quad = wg & 3;
    if (nwg % 32 == 0) { blk = (wg >> 5) * 8 + (wg & 7); quad = (wg >> 3) & 3; }
    Q4_T(0);
    const int fw = RT ? fr : F;                                    // the table's real width (row stride, valid features, loss scale)
    double *img = imgs + (int64_t)blk * N::img_doubles + 4 * quad;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)qpacked, 0, N::q_frags() * 1024, 0x00020000);
    const int lane16 = lane * 16;
    // ---- request order = order of use: the 4 input rows (HBM), the biases, then the fragment ring
    const int tj = threadIdx.x & 3, tf = threadIdx.x >> 2;          // input / finalising thread: row tj, feature tf (+ 64 i)
    const int64_t trow = (int64_t)blk * 16 + 4 * quad + tj;
    const bool tvalid = trow < n;
    constexpr int kXH = (16 * tiles(F) + 63) / 64;                 // 64-slot halves of the input image (1 up to 63 columns, 2 up to 127)
    double xv[kXH], xmn[kXH], xrg[kXH];
#pragma unroll
    for (int hh = 0; hh < kXH; ++hh) {
        const int f = tf + 64 * hh;
        const int fc = f < fw ? f : 0;                              // padding slots read feature 0 (finite, never used)
        const int64_t at = (tvalid ? trow : 0) * fw + fc;
        xv[hh] = in_f64 ? ((const double *)xin)[at] : (double)((const float *)xin)[at];
        xmn[hh] = 0.0; xrg[hh] = 1.0;
        if (feats) { xmn[hh] = feats[fc]; xrg[hh] = feats[fw + fc]; }
    }
    // (the biases wait in registers until the ring is requested: written to LDS first, they held the ring's requests behind the rows'
    // HBM round trip -- 4,100 cycles in front of the first GEMM, tools/q4_trace.py)
    constexpr int kNB = N::qb_off(N::L), kNBI = (kNB + 255) / 256;
    double bv[kNBI];
#pragma unroll
    for (int i = 0; i < kNBI; ++i) {
        const int idx = threadIdx.x + 256 * i;
        bv[i] = qpacked[N::q_frags() * 128 + (idx < kNB ? idx : 0)];
    }
    d2 ring[T::D];
    q_prologue<N>(ring, rs, lane16, wave, std::make_integer_sequence<int, T::D>{});
    Q4_T(1);
#pragma unroll
    for (int hh = 0; hh < kXH; ++hh) {
        const int f = tf + 64 * hh;
        double v = feats ? (xv[hh] - xmn[hh]) / xrg[hh] : xv[hh];
        v = f < fw ? v : 0.0;
        if (f == F) v = 1.0;                                         // the ones slot (carries db) sits at the class width
        if (f < 16 * tiles(F)) lds[T::xo(0) + tj * T::xs(0) + f] = v;
        if (f < N::x_rows(0)) img[(N::x_off(0) + f) * 16 + tj] = v;
    }
#pragma unroll
    for (int i = 0; i < kNBI; ++i) {
        const int idx = threadIdx.x + 256 * i;
        if (idx < kNB) lds[T::bo + idx] = bv[i];
    }
    __syncthreads();
    Q4_T(2);
    // D-operand lane (i, b, j) = lane 16 i + 4 b + j holds feature 16 grp + 4 b + i of row j; B-operand lane (k, b, j): contraction index k of row j
    const int dj = lane & 3, dfo = ((lane >> 2) & 3) * 4 + (lane >> 4), bk = lane >> 4;
    const bool dvalid = (int64_t)blk * 16 + 4 * quad + dj < n;
    double lacc = 0.0;
    double *pbuf = lds + T::po;
    // One chain GEMM.  `fin(value without bias, feature, row, row is a real one)` finalises one output element: direct from the
    // accumulator layout for a GEMM whose groups are dealt to the waves, from the summed partials (thread (tf + 64 i, tj)) for a split one.
#define Q_GEMM(g, IN_OFF, IN_RS, FIN)                                                                                        \
    {                                                                                                                        \
        constexpr int G_ = T::G(g), P_ = T::P(g);                                                                            \
        const int part_ = P_ > 1 ? wave / G_ : 0;                                                                            \
        const double *brow = lds + (IN_OFF) + dj * (IN_RS) + bk + 8 * part_ * T::KS8P(g);                                    \
        double o[T::NGM(g)];                                                                                                 \
        q_gemm<N, g>(o, brow, ring, rs, lane16, wave, std::make_integer_sequence<int, T::steps(g)>{});                       \
        if constexpr (P_ == 1) {                                                                                             \
            _Pragma("unroll") for (int gi = 0; gi < T::NGM(g); ++gi) {                                                       \
                const int grp = wave + 4 * gi;                                                                               \
                if (grp < G_) FIN(o[gi], 16 * grp + dfo, dj, dvalid);                                                        \
            }                                                                                                                \
        } else {                                                                                                             \
            pbuf[(part_ * G_ + wave % G_) * 64 + lane] = o[0];                                                               \
            __syncthreads();                                                                                                 \
            if ((int)threadIdx.x < 64 * G_) {                                                                                \
                const int grp = threadIdx.x >> 6;                                                                            \
                double v = 0.0;                                                                                              \
                _Pragma("unroll") for (int p = 0; p < P_; ++p) v += pbuf[(p * G_ + grp) * 64 + lane];                        \
                FIN(v, 16 * grp + dfo, dj, dvalid);                                                                          \
            }                                                                                                                \
        }                                                                                                                    \
        __syncthreads();                                                                                                     \
    }
    // ---------------- forward: X_{l+1} = act(W_l X_l + b_l) ----------------
#define Q_FWD(l)                                                                                                             \
    {                                                                                                                        \
        auto fin = [&](double v, int f, int j, bool) {                                                                       \
            v += lds[T::bo + N::qb_off(l) + f];                                                                              \
            if (N::act(l)) v = v > 0.0 ? v : v * kSlope;                                                                     \
            if (f == N::dim((l) + 1)) v = 1.0;                                                                               \
            lds[T::xo((l) + 1) + j * T::xs((l) + 1) + f] = v;                                                                \
            img[(N::x_off((l) + 1) + f) * 16 + j] = v;                                                                       \
        };                                                                                                                   \
        Q_GEMM(l, T::xo(l), T::xs(l), fin)                                                                                   \
        Q4_T(3 + (l));                                                                                                       \
    }
    Q_FWD(0) Q_FWD(1) Q_FWD(2) Q_FWD(3) Q_FWD(4) Q_FWD(5) Q_FWD(6)
#undef Q_FWD
    {   // de4 (no activation) + loss + dL/drecon = 2 (r - x) / C (utils.py:195-199)
        auto fin = [&](double v, int f, int j, bool rowok) {
            v += lds[T::bo + N::qb_off(7) + f];
            const double d = v - lds[T::xo(0) + j * T::xs(0) + f];
            const bool live = rowok && f < fw;
            if (live) lacc += d * d;
            const double dz = live ? d * (2.0 / (double)fw) : 0.0;
            lds[T::zo(1) + j * T::zs + f] = dz;
            img[(N::z_off(7) + f) * 16 + j] = dz;
        };
        Q_GEMM(7, T::xo(7), T::xs(7), fin)
        Q4_T(10);
    }
    // ---------------- backward chain: GEMM 15 - l: dZ_{l-1} = (W_l^T dZ_l) . lrelu'(X_l); dZ_l in buffer l & 1 ----------------
#define Q_BWD(l)                                                                                                             \
    {                                                                                                                        \
        auto fin = [&](double v, int f, int j, bool) {                                                                       \
            if (N::act((l) - 1)) v = lds[T::xo(l) + j * T::xs(l) + f] > 0.0 ? v : v * kSlope;                                \
            lds[T::zo(((l) - 1) & 1) + j * T::zs + f] = v;                                                                   \
            img[(N::z_off((l) - 1) + f) * 16 + j] = v;                                                                       \
        };                                                                                                                   \
        Q_GEMM(15 - (l), T::zo((l) & 1), T::zs, fin)                                                                         \
        Q4_T(18 - (l));                                                                                                      \
    }
    Q_BWD(7) Q_BWD(6) Q_BWD(5) Q_BWD(4) Q_BWD(3) Q_BWD(2) Q_BWD(1)
#undef Q_BWD
#undef Q_GEMM
    // loss partial of these 4 rows: lanes of a wave, then waves 0..3 (fixed order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lacc += __shfl_down(lacc, off);
    if (lane == 0) loss_lds[wave] = lacc;
    __syncthreads();
    if (threadIdx.x == 0) loss_part[4 * blk + quad] = ((loss_lds[0] + loss_lds[1]) + loss_lds[2]) + loss_lds[3];
    Q4_T(18);
}

template <int F, int Z, bool RT>
int launch_q(unsigned grid, hipStream_t s, const double *qpacked, const void *x, int in_f64, int64_t rows, const double *feats, double *imgs,
             double *loss_part, int fr) {
    hipLaunchKernelGGL((chain64q_kernel<F, Z, RT>), dim3(grid), dim3(256), 0, s, qpacked, x, in_f64, rows, feats, imgs, loss_part, fr);
    return BAMD_OK;
}

}  // namespace

int fused64q_launch(int F, int Z, bool rt, unsigned grid, hipStream_t s, const double *qpacked, const void *x, int in_f64, int64_t rows,
                    const double *feats, double *imgs, double *loss_part, int fr) {
#define Q_CASE(F_, Z_, RT_) if (F == F_ && Z == Z_ && rt == RT_) return launch_q<F_, Z_, RT_>(grid, s, qpacked, x, in_f64, rows, feats, imgs, loss_part, fr);
    Q_CASE(24, 15, false) Q_CASE(24, 12, false) Q_CASE(24, 8, false) Q_CASE(24, 6, false) Q_CASE(24, 10, false)
    Q_CASE(24, 5, false) Q_CASE(24, 4, false) Q_CASE(24, 3, false) Q_CASE(24, 2, false)
    Q_CASE(31, 15, true) Q_CASE(47, 15, true) Q_CASE(63, 15, true) Q_CASE(31, 31, true) Q_CASE(63, 31, true)
    Q_CASE(79, 31, true) Q_CASE(95, 31, true) Q_CASE(111, 31, true) Q_CASE(127, 31, true)      // 64 .. 127 columns: Impl64Q (fused64.hip)
    Q_CASE(63, 63, true) Q_CASE(79, 63, true) Q_CASE(95, 63, true) Q_CASE(111, 63, true) Q_CASE(127, 63, true)      // a latent of 32 .. 63
#undef Q_CASE
    set_error("fp64 4-row chain: no instantiation for this shape");
    return BAMD_ERR_UNSUPPORTED;
}

}  // namespace bamd

#ifdef BAMD_Q4_TRACE
// debug builds only (tools/q4_trace.py): the stamps of the last chain64q_kernel launch.  Not part of the ABI: the shipped library does not export it.
extern "C" int bamd_debug_q4_trace(unsigned long long *dst, int count) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(bamd::g_q4_trace), sizeof(unsigned long long) * (count < 32 ? count : 32));
}
#endif
