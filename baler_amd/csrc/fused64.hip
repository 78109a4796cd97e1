// fp64 small-batch training step of AE(F, Z) on v_mfma_f64_16x16x4_f64 (BAMD_MODE_F64): the reference's own dtype
// (models.py:128-136) at the reference's batch size -- the mode that pins 500-step CLI runs at 1e-9.  Same two-launch
// structure as the fp32 small-batch step in fused.hip (a workgroup owns ONE 16-row block and its 4 waves split every
// layer's output tiles, swapping tiles through a double-buffered LDS exchange; then one workgroup per weight-gradient
// tile over the whole batch, optionally fused with Adam and the refresh of the packed weights), re-derived for the
// f64 MFMA, whose C/D map is NOT the f32 one:
//     f32 16x16x4:  D row = 4 (lane >> 4) + reg        f64 16x16x4:  D row = (lane >> 4) + 4 reg
// so register r of an output tile on lane group g holds feature 16 t + 4 r + g, and -- because the B operand of MFMA
// step r takes k = g from lane group g -- step r of the next layer consumes exactly features 16 q + 4 r .. + 3: the chain
// closes with NATURAL feature order (slot s = feature s; no g-major / r-major distinction between full and partial tiles).
// Batches above FusedState64::max_rows stay on the layer-wise kernels (generic.hip); the inference entry points run on
// infer64_kernel below (every wave its own 16 rows, register chain, no exchange).
// Three chains write the same images for the weight-gradient kernels: chain64q_kernel (FOUR rows per workgroup on v_mfma_f64_4x4x4: up
// to 1,536 rows, the reference's 512-row step), chain64_kernel (one workgroup per 16-row block, tiles exchanged through LDS) and
// chain64r_kernel (one wave per block, activations in registers: from 16,384 rows on).
#include "fused.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <utility>

#include "fused64_net.hpp"

namespace bamd {
namespace {

#ifdef BAMD_Q4_TRACE   // debug build: shader-clock stamps of one dw64_kernel workgroup (tools/q4_trace.py)
__device__ unsigned long long g_dw64_trace[8];
#endif
// The 15 chain GEMMs of one step as ONE fragment sequence per wave (forward layers 0..7, then the transposed fragments
// of layers 7..1); a register ring runs D fragments ahead of the MFMAs across layer boundaries (see fused.hip LatSeq).
template <class N, int W, int D_> struct Seq {
    static constexpr int D = D_, Wv = W, NG = 15;
    __host__ __device__ static constexpr int layer(int g) { return g < 8 ? g : 15 - g; }
    __host__ __device__ static constexpr int kd(int g) { return g < 8 ? N::dim(g) : N::dim(layer(g) + 1); }
    __host__ __device__ static constexpr int nt(int g) { return tiles(g < 8 ? N::dim(g + 1) : N::dim(layer(g))); }
    __host__ __device__ static constexpr int base(int g) { return (g < 8 ? N::wf_off(g) : N::wb_off(layer(g))) / 64; }
    __host__ __device__ static constexpr int nl(int g) { return (nt(g) + W - 1) / W; }
    __host__ __device__ static constexpr int nf(int g) { return tiles(kd(g)) * nl(g); }
    __host__ __device__ static constexpr int start(int g) { int s = 0; for (int j = 0; j < g; ++j) s += nf(j); return s; }
    static constexpr int total = start(NG);
    __host__ __device__ static constexpr int gemm_of(int G) { int g = 0; for (int j = 1; j < NG; ++j) if (G >= start(j)) g = j; return g; }
};
template <class SQ, int G>
__device__ __forceinline__ void seq_issue(d4 (&slot)[SQ::D], const WStream &ws, int wave) {
    if constexpr (G < SQ::total) {
        constexpr int g = SQ::gemm_of(G), f = G - SQ::start(g), NL = SQ::nl(g), q = f / NL, i = f % NL, NT = SQ::nt(g);
        int t = wave + SQ::Wv * i;
        t = t < NT ? t : NT - 1;
        slot[G % SQ::D] = frag_rt(ws, SQ::base(g) + q * NT + t);
    }
}
template <class SQ, int... G>
__device__ __forceinline__ void seq_prologue(d4 (&slot)[SQ::D], const WStream &ws, int wave, std::integer_sequence<int, G...>) {
    (seq_issue<SQ, G>(slot, ws, wave), ...);
    __builtin_amdgcn_sched_barrier(0);
}
template <class SQ, int g, int f>
__device__ __forceinline__ void seq_one(const d4 (&in)[tiles(SQ::kd(g))], d4 (&out)[SQ::nl(g)], d4 (&slot)[SQ::D], const WStream &ws,
                                        int wave) {
    constexpr int NL = SQ::nl(g), S0 = SQ::start(g), KD = SQ::kd(g), q = f / NL, i = f % NL, s = (S0 + f) % SQ::D;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (r < tile_steps(KD, q)) out[i] = mfma(slot[s][r], in[q][r], out[i]);
    seq_issue<SQ, S0 + f + SQ::D>(slot, ws, wave);
    __builtin_amdgcn_sched_barrier(0);
}
template <class SQ, int g, int... P>
__device__ __forceinline__ void seq_mm_impl(const d4 (&in)[tiles(SQ::kd(g))], d4 (&out)[SQ::nl(g)], d4 (&slot)[SQ::D], const WStream &ws,
                                            int wave, std::integer_sequence<int, P...>) {
    (seq_one<SQ, g, P>(in, out, slot, ws, wave), ...);
}
template <class SQ, int g>
__device__ __forceinline__ void seq_mm(const d4 (&in)[tiles(SQ::kd(g))], d4 (&out)[SQ::nl(g)], d4 (&slot)[SQ::D], const WStream &ws,
                                       int wave) {
    seq_mm_impl<SQ, g>(in, out, slot, ws, wave, std::make_integer_sequence<int, SQ::nf(g)>{});
}

template <int NT> __device__ __forceinline__ void collect(const d4 *xch, d4 (&all)[NT], int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t) all[t] = xch[t * 64 + lane];
}
template <int NL> __device__ __forceinline__ void lrelu_bwd(d4 (&d)[NL], const d4 (&y)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[i][r] = y[i][r] > 0.0 ? d[i][r] : d[i][r] * kSlope;
}

// own tiles -> LDS exchange buffer (C layout) and -> the global [slot][16 rows] image; slot of register (g, r) of tile t =
// 16 t + g + 4 r (natural feature order).  ONES: the first padding slot of dimension D is the ones column that carries db.
template <int D, bool ONES, int W>
__device__ __forceinline__ void publish(d4 *xch, double *img, int slot0, const d4 (&loc)[(tiles(D) + W - 1) / W], int lane, int wave) {
    constexpr int NT = tiles(D), NL = (NT + W - 1) / W;
    constexpr int T1 = tiles(D) - 1, V = D - 16 * T1;      // V = slot of the ones column inside tile T1 (D % 16 != 0)
    const int g = lane >> 4, col = lane & 15;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int t = wave + W * i;
        if (t < NT) {
            if (xch) xch[t * 64 + lane] = loc[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double v = loc[i][r];
                if (ONES && t == T1 && g + 4 * r == V) v = 1.0;
                img[(slot0 + 16 * t + g + 4 * r) * 16 + col] = v;
            }
        }
    }
}

template <int F, int Z, int W, bool RT = false>
__global__ void __launch_bounds__(64 * W) chain64_kernel(const d4 *packed, const void *__restrict__ xin, int in_f64, int64_t n,
                                                         const double *__restrict__ feats, double *__restrict__ imgs,
                                                         double *__restrict__ loss_part, int fr) {
    using N = Net64<F, Z>;
    constexpr int TF = tiles(F), TZ = tiles(Z);
    static_assert(TF <= W && TZ <= W && F % 16 != 0, "input / latent tiles");
    constexpr int kNB = N::bf_off(N::L) - N::bf_off(0);
    extern __shared__ __attribute__((aligned(32))) unsigned char lds_raw[];
    d4 *xchA = (d4 *)lds_raw, *xchB = xchA + 13 * 64, *bias_lds = xchB + 13 * 64;
    __shared__ double loss_lds[W];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    WStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)packed, 0, N::packed_d4() * 32, 0x00020000);
    ws.voff = lane * 32;
    // rows first (layer 0 waits for them), then the ring's first fragments queue up behind them
    const int64_t row = (int64_t)blockIdx.x * 16 + (lane & 15);
    const bool valid = row < n;
    const int fw = RT ? fr : F;                                    // the table's real width (row stride, valid features, loss scale)
    const int64_t rbase = (valid ? row : 0) * fw;
    d4 a0[TF];
#pragma unroll
    for (int t = 0; t < TF; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = creg_feature(F, t, g, r);
            const bool live = f >= 0 && (!RT || f < fr);
            const int fc = live ? f : 0;                          // padding slots read feature 0 (finite, meets zero weights)
            double v = in_f64 ? ((const double *)xin)[rbase + fc] : (double)((const float *)xin)[rbase + fc];
            if (feats) v = (v - feats[fc]) / feats[fw + fc];
            a0[t][r] = live ? v : 0.0;
        }
    using SQ = Seq<N, W, 12>;
    d4 ring[SQ::D];
    for (int i = threadIdx.x; i < kNB; i += 64 * W) bias_lds[i] = packed[N::bf_off(0) + i];
    seq_prologue<SQ>(ring, ws, wave, std::make_integer_sequence<int, SQ::D>{});
    __syncthreads();
    double *img = imgs + (int64_t)blockIdx.x * N::img_doubles;
#define BIAS64(loc, l, NTl)                                                                              \
    _Pragma("unroll") for (int i = 0; i < (NTl + W - 1) / W; ++i) {                                      \
        int t_ = wave + W * i; t_ = t_ < NTl ? t_ : NTl - 1;                                             \
        loc[i] = bias_lds[(N::bf_off(l) - N::bf_off(0)) + t_ * 4 + g];                                   \
    }
    constexpr int L13 = (13 + W - 1) / W, L7 = (7 + W - 1) / W, L4 = (4 + W - 1) / W, LF = (TF + W - 1) / W;
    static_assert(LF == 1, "one input / output tile per wave");
    // ---------------- forward ----------------
    {
        d4 own[LF];
        own[0] = a0[wave < TF ? wave : TF - 1];
        publish<F, true, W>(nullptr, img, N::x_off(0), own, lane, wave);
    }
    d4 s1[L13], s2[L7], s3[L4], s4[1], s5[L4], s6[L7], s7[L13], o8[LF];
    BIAS64(s1, 0, 13) seq_mm<SQ, 0>(a0, s1, ring, ws, wave); lrelu(s1);
    publish<200, true, W>(xchA, img, N::x_off(1), s1, lane, wave);
    __syncthreads();
    { d4 a1[13]; collect(xchA, a1, lane); BIAS64(s2, 1, 7) seq_mm<SQ, 1>(a1, s2, ring, ws, wave); lrelu(s2); }
    publish<100, true, W>(xchB, img, N::x_off(2), s2, lane, wave);
    __syncthreads();
    { d4 a2[7]; collect(xchB, a2, lane); BIAS64(s3, 2, 4) seq_mm<SQ, 2>(a2, s3, ring, ws, wave); lrelu(s3); }
    publish<50, true, W>(xchA, img, N::x_off(3), s3, lane, wave);
    __syncthreads();
    { d4 a3[4]; collect(xchA, a3, lane); BIAS64(s4, 3, TZ) seq_mm<SQ, 3>(a3, s4, ring, ws, wave); }          // en4: no activation
    publish<Z, true, W>(xchB, img, N::x_off(4), s4, lane, wave);
    __syncthreads();
    { d4 a4[TZ]; collect(xchB, a4, lane); BIAS64(s5, 4, 4) seq_mm<SQ, 4>(a4, s5, ring, ws, wave); lrelu(s5); }
    publish<50, true, W>(xchA, img, N::x_off(5), s5, lane, wave);
    __syncthreads();
    { d4 a5[4]; collect(xchA, a5, lane); BIAS64(s6, 5, 7) seq_mm<SQ, 5>(a5, s6, ring, ws, wave); lrelu(s6); }
    publish<100, true, W>(xchB, img, N::x_off(6), s6, lane, wave);
    __syncthreads();
    { d4 a6[7]; collect(xchB, a6, lane); BIAS64(s7, 6, 13) seq_mm<SQ, 6>(a6, s7, ring, ws, wave); lrelu(s7); }
    publish<200, true, W>(xchA, img, N::x_off(7), s7, lane, wave);
    __syncthreads();
    { d4 a7[13]; collect(xchA, a7, lane); BIAS64(o8, 7, TF) seq_mm<SQ, 7>(a7, o8, ring, ws, wave); }         // de4: no activation
#undef BIAS64
    // ---------------- loss, dL/drecon = 2 (r - x) / C (utils.py:195-199) ----------------
    double lacc = 0.0;
    {
        const int t = wave < TF ? wave : TF - 1;
        const d4 x0 = a0[t];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double d = o8[0][r] - x0[r];
            const bool live = valid && wave < TF && creg_feature(F, t, g, r) >= 0 && (!RT || creg_feature(F, t, g, r) < fr);
            if (live) lacc += d * d;
            o8[0][r] = live ? d * (2.0 / (double)fw) : 0.0;
        }
    }
    // ---------------- backward chain (input gradients), publishing dZ images ----------------
    publish<F, false, W>(xchB, img, N::z_off(7), o8, lane, wave);
    __syncthreads();
#define BWD64(GI, NIN, LOUT, SACT, XIN, XOUT, ZO, DOUT, MASK)                                            \
    {                                                                                                    \
        d4 din[NIN]; collect(XIN, din, lane);                                                            \
        d4 dx[LOUT];                                                                                     \
        _Pragma("unroll") for (int i = 0; i < LOUT; ++i) dx[i] = (d4){0.0, 0.0, 0.0, 0.0};               \
        seq_mm<SQ, GI>(din, dx, ring, ws, wave);                                                         \
        if (MASK) lrelu_bwd(dx, SACT);                                                                   \
        _Pragma("unroll") for (int i = 0; i < LOUT; ++i) SACT[i] = dx[i];                                \
    }                                                                                                    \
    publish<DOUT, false, W>(XOUT, img, ZO, SACT, lane, wave);                                            \
    __syncthreads();
    BWD64(8, TF, L13, s7, xchB, xchA, N::z_off(6), 200, true)
    BWD64(9, 13, L7, s6, xchA, xchB, N::z_off(5), 100, true)
    BWD64(10, 7, L4, s5, xchB, xchA, N::z_off(4), 50, true)
    BWD64(11, 4, 1, s4, xchA, xchB, N::z_off(3), Z, false)                 // dL/dz: en4 has no activation
    BWD64(12, TZ, L4, s3, xchB, xchA, N::z_off(2), 50, true)
    BWD64(13, 4, L7, s2, xchA, xchB, N::z_off(1), 100, true)
    BWD64(14, 7, L13, s1, xchB, (d4 *)nullptr, N::z_off(0), 200, true)
#undef BWD64
    // loss partial of this 16-row block: lanes of a wave, then waves 0..W-1 (fixed order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lacc += __shfl_down(lacc, off);
    if (lane == 0) loss_lds[wave] = lacc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < W; ++w) sum += loss_lds[w];
        loss_part[blockIdx.x] = sum;
    }
}

// ---- the training chain for LARGE fp64 batches: one WAVE per 16-row block, activations in registers ------------------------------
// chain64_kernel gives a block to a whole workgroup (4 waves split every layer's tiles and swap them through LDS, one barrier per
// layer): right for the 512-row step, but at 262,144 rows it spends 1.6 ms at 55 % MFMA busy -- 15 barriers per block, and every
// block's workgroup streams the whole model.  infer64_kernel shows what the fp64 MFMA does when a wave owns all tiles of its rows
// (85-89 % busy).  This is that formulation for training: Seq<N, 1, D> is the 15-GEMM fragment sequence of ONE wave (forward 0..7,
// then the transposed fragments of layers 7..1), a layer's output tiles are the next GEMM's B operand as they stand, nothing is
// exchanged, no barrier after the bias fragments are staged.  The images ([slot][16 rows], global) are written exactly as
// chain64_kernel writes them -- same values bit for bit: every output element still accumulates its k blocks in order -- so the
// weight-gradient kernels do not change.  The activations are not kept for the backward masks (392 registers): LeakyReLU records
// the SIGN of each value in a bit mask (4 bits per tile and lane: 12 registers for the six activated layers).
template <int NL> __device__ __forceinline__ void lrelu_rec(d4 (&a)[NL], unsigned long long &mask) {
    mask = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool pos = a[i][r] > 0.0;
            a[i][r] = pos ? a[i][r] : a[i][r] * kSlope;
            mask |= (unsigned long long)pos << (4 * i + r);
        }
}
template <int NL> __device__ __forceinline__ void lrelu_bwd_mask(d4 (&d)[NL], unsigned long long mask) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[i][r] = ((mask >> (4 * i + r)) & 1ull) ? d[i][r] : d[i][r] * kSlope;
}
template <int F, int Z, bool RT = false>
__global__ void __launch_bounds__(256) chain64r_kernel(const d4 *packed, const void *__restrict__ xin, int in_f64, int64_t n,
                                                       const double *__restrict__ feats, double *__restrict__ imgs,
                                                       double *__restrict__ loss_part, int nblk, int fr) {
    using N = Net64<F, Z>;
    constexpr int TF = tiles(F), TZ = tiles(Z);
    static_assert(TZ <= 2 && F % 16 != 0, "input / latent tiles");
    constexpr int kNB = N::bf_off(N::L) - N::bf_off(0);
    extern __shared__ __attribute__((aligned(32))) unsigned char lds_raw[];
    d4 *bias_lds = (d4 *)lds_raw;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    for (int i = threadIdx.x; i < kNB; i += 256) bias_lds[i] = packed[N::bf_off(0) + i];
    __syncthreads();
    const int blk = (int)blockIdx.x * 4 + wave;
    if (blk >= nblk) return;
    WStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)packed, 0, N::packed_d4() * 32, 0x00020000);
    ws.voff = lane * 32;
    const int64_t row = (int64_t)blk * 16 + (lane & 15);
    const bool valid = row < n;
    const int fw = RT ? fr : F;                                    // the table's real width (row stride, valid features, loss scale)
    const int64_t rbase = (valid ? row : 0) * fw;
    d4 a0[TF];
#pragma unroll
    for (int t = 0; t < TF; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = creg_feature(F, t, g, r);
            const bool live = f >= 0 && (!RT || f < fr);
            const int fc = live ? f : 0;                          // padding slots read feature 0 (finite, meets zero weights)
            double v = in_f64 ? ((const double *)xin)[rbase + fc] : (double)((const float *)xin)[rbase + fc];
            if (feats) v = (v - feats[fc]) / feats[fw + fc];
            a0[t][r] = live ? v : 0.0;
        }
    using SQ = Seq<N, 1, 8>;
    d4 ring[SQ::D];
    seq_prologue<SQ>(ring, ws, 0, std::make_integer_sequence<int, SQ::D>{});
    double *img = imgs + (int64_t)blk * N::img_doubles;
#define BIAS64R(loc, l, NTl)                                                                             \
    _Pragma("unroll") for (int i = 0; i < NTl; ++i) loc[i] = bias_lds[(N::bf_off(l) - N::bf_off(0)) + i * 4 + g];
    unsigned long long m1, m2, m3, m5, m6, m7;
    double lacc = 0.0;
    d4 o8[TF];
    publish<F, true, 1>(nullptr, img, N::x_off(0), a0, lane, 0);
    {   // ---------------- forward: a layer's tiles live until the next layer has consumed them ----------------
        d4 s7[13];
        {
            d4 s6[7];
            {
                d4 s5[4];
                {
                    d4 s4[TZ];
                    {
                        d4 s3[4];
                        {
                            d4 s2[7];
                            {
                                d4 s1[13];
                                BIAS64R(s1, 0, 13) seq_mm<SQ, 0>(a0, s1, ring, ws, 0); lrelu_rec(s1, m1);
                                publish<200, true, 1>(nullptr, img, N::x_off(1), s1, lane, 0);
                                BIAS64R(s2, 1, 7) seq_mm<SQ, 1>(s1, s2, ring, ws, 0); lrelu_rec(s2, m2);
                            }
                            publish<100, true, 1>(nullptr, img, N::x_off(2), s2, lane, 0);
                            BIAS64R(s3, 2, 4) seq_mm<SQ, 2>(s2, s3, ring, ws, 0); lrelu_rec(s3, m3);
                        }
                        publish<50, true, 1>(nullptr, img, N::x_off(3), s3, lane, 0);
                        BIAS64R(s4, 3, TZ) seq_mm<SQ, 3>(s3, s4, ring, ws, 0);                                   // en4: no activation
                    }
                    publish<Z, true, 1>(nullptr, img, N::x_off(4), s4, lane, 0);
                    BIAS64R(s5, 4, 4) seq_mm<SQ, 4>(s4, s5, ring, ws, 0); lrelu_rec(s5, m5);
                }
                publish<50, true, 1>(nullptr, img, N::x_off(5), s5, lane, 0);
                BIAS64R(s6, 5, 7) seq_mm<SQ, 5>(s5, s6, ring, ws, 0); lrelu_rec(s6, m6);
            }
            publish<100, true, 1>(nullptr, img, N::x_off(6), s6, lane, 0);
            BIAS64R(s7, 6, 13) seq_mm<SQ, 6>(s6, s7, ring, ws, 0); lrelu_rec(s7, m7);
        }
        publish<200, true, 1>(nullptr, img, N::x_off(7), s7, lane, 0);
        BIAS64R(o8, 7, TF) seq_mm<SQ, 7>(s7, o8, ring, ws, 0);                                                   // de4: no activation
    }
#undef BIAS64R
    // ---------------- loss, dL/drecon = 2 (r - x) / C (utils.py:195-199) ----------------
#pragma unroll
    for (int t = 0; t < TF; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double d = o8[t][r] - a0[t][r];
            const bool live = valid && creg_feature(F, t, g, r) >= 0 && (!RT || creg_feature(F, t, g, r) < fr);
            if (live) lacc += d * d;
            o8[t][r] = live ? d * (2.0 / (double)fw) : 0.0;
        }
    publish<F, false, 1>(nullptr, img, N::z_off(7), o8, lane, 0);
    // ---------------- backward chain (input gradients), publishing the dZ images ----------------
#define ZERO64(a, NTl) _Pragma("unroll") for (int i = 0; i < NTl; ++i) a[i] = (d4){0.0, 0.0, 0.0, 0.0};
    {
        d4 d1[13];
        {
            d4 d2[7];
            {
                d4 d3[4];
                {
                    d4 d4_[TZ];
                    {
                        d4 d5[4];
                        {
                            d4 d6[7];
                            {
                                d4 d7[13];
                                ZERO64(d7, 13) seq_mm<SQ, 8>(o8, d7, ring, ws, 0); lrelu_bwd_mask(d7, m7);
                                publish<200, false, 1>(nullptr, img, N::z_off(6), d7, lane, 0);
                                ZERO64(d6, 7) seq_mm<SQ, 9>(d7, d6, ring, ws, 0); lrelu_bwd_mask(d6, m6);
                            }
                            publish<100, false, 1>(nullptr, img, N::z_off(5), d6, lane, 0);
                            ZERO64(d5, 4) seq_mm<SQ, 10>(d6, d5, ring, ws, 0); lrelu_bwd_mask(d5, m5);
                        }
                        publish<50, false, 1>(nullptr, img, N::z_off(4), d5, lane, 0);
                        ZERO64(d4_, TZ) seq_mm<SQ, 11>(d5, d4_, ring, ws, 0);                                     // dL/dz: en4 has no activation
                    }
                    publish<Z, false, 1>(nullptr, img, N::z_off(3), d4_, lane, 0);
                    ZERO64(d3, 4) seq_mm<SQ, 12>(d4_, d3, ring, ws, 0); lrelu_bwd_mask(d3, m3);
                }
                publish<50, false, 1>(nullptr, img, N::z_off(2), d3, lane, 0);
                ZERO64(d2, 7) seq_mm<SQ, 13>(d3, d2, ring, ws, 0); lrelu_bwd_mask(d2, m2);
            }
            publish<100, false, 1>(nullptr, img, N::z_off(1), d2, lane, 0);
            ZERO64(d1, 13) seq_mm<SQ, 14>(d2, d1, ring, ws, 0); lrelu_bwd_mask(d1, m1);
        }
        publish<200, false, 1>(nullptr, img, N::z_off(0), d1, lane, 0);
    }
#undef ZERO64
    // loss partial of this 16-row block: the wave's lanes in a fixed order
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lacc += __shfl_down(lacc, off);
    if (lane == 0) loss_part[blk] = lacc;
}

struct Adam64 {
    double *params, *pcopy, *m, *v, *packed;
    const int *sc_off, *sc_idx;
    double *loss_accum;
    double b1, b2, eps, step_size, bc2_sqrt;
};

// one workgroup per weight-gradient tile: contracts dZ^T (tile nt of layer l) with [X | 1] (tile kt) over all 16-row blocks
// in a fixed order (wave w takes blocks w, w + 4, ..; then waves 0..3); optionally applies Adam to the parameters it owns.
enum { DW_WRITE = 0, DW_ADAM = 1 };
template <class N, int MODE>
__global__ void __launch_bounds__(256) dw64_kernel(const double *__restrict__ imgs, int nblk, const double *__restrict__ loss_part,
                                                   const int *__restrict__ inv_map, double *__restrict__ grads, Adam64 ad,
                                                   const double *__restrict__ part, int nsplit, int np, double inv_c, int nloss) {
    // nsplit == 0: the whole job.  nsplit > 0: the tiles' partial sums over `nsplit` block ranges are in `part` (dw64m_kernel); this
    // launch adds them in range order and finishes (store / Adam).
    constexpr int T = N::slab_off(N::L);       // (np, inv_c: the handle's real parameter count and 1 / columns -- class instantiations)
    constexpr int kPerXcd = (T + 1 + 7) / 8;
    __shared__ __attribute__((aligned(32))) d4 red[4 * 64];
    const int tile = (blockIdx.x & 7) * kPerXcd + (blockIdx.x >> 3);       // XCD c takes a contiguous tile range (see fused.hip)
    if (tile > T) return;
#ifdef BAMD_Q4_TRACE
#define DW_T(i) do { if (tile == 150 && threadIdx.x == 0) g_dw64_trace[(i) - 20] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DW_T(i) do {} while (0)
#endif
    DW_T(20);
    if (tile == T) {   // loss: fixed-order sum of the per-block partials, / C
        const double s = block_sum_fixed(loss_part, nloss, (double *)red);      // (one partial per block, four with the 4-row chain)
        if (threadIdx.x == 0) {
            const double gl = s * inv_c;
            if (grads) grads[np] = gl;
            if (MODE == DW_ADAM && ad.loss_accum) *ad.loss_accum += gl;
        }
        return;
    }
    const int p = inv_map[tile * 256 + threadIdx.x];
    int l = 0;
#pragma unroll
    for (int j = 1; j < N::L; ++j) if (tile >= N::slab_off(j)) l = j;
    int nt_count = tiles(N::dim(1)), soff = 0, xo = N::x_off(0), zo = N::z_off(0);
#pragma unroll
    for (int j = 1; j < N::L; ++j)
        if (l == j) { nt_count = tiles(N::dim(j + 1)); soff = N::slab_off(j); xo = N::x_off(j); zo = N::z_off(j); }
    const int idx = tile - soff, kt = idx / nt_count, nt = idx - kt * nt_count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
    // A[i][k = g] of step r = dZ[slot 16 nt + i][batch row 4 g + r]: four consecutive rows of one image row per lane
    const double *pz = imgs + ((zo + 16 * nt + i) * 16 + 4 * g);
    const double *px = imgs + ((xo + 16 * kt + i) * 16 + 4 * g);
    double pm = 0.0, pv = 0.0, pp = 0.0;
    int s0 = 0, s1 = 0;
    if (MODE == DW_ADAM && p >= 0) { pm = ad.m[p]; pv = ad.v[p]; pp = ad.params[p]; s0 = ad.sc_off[p]; s1 = ad.sc_off[p + 1]; }
    // the packed slots of the parameter (forward and transposed fragment of either chain: up to four) are looked up now, not behind the reduction
    int sc0 = -1, sc1 = -1, sc2 = -1, sc3 = -1;
    if (MODE == DW_ADAM && s1 > s0) sc0 = ad.sc_idx[s0];
    if (MODE == DW_ADAM && s1 > s0 + 1) sc1 = ad.sc_idx[s0 + 1];
    if (MODE == DW_ADAM && s1 > s0 + 2) sc2 = ad.sc_idx[s0 + 2];
    if (MODE == DW_ADAM && s1 > s0 + 3) sc3 = ad.sc_idx[s0 + 3];
    DW_T(21);
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
    // (requesting the first UB blocks' image slices BEFORE the parameter index and the optimiser state behind it was measured in round 6:
    // 512-row step 30.1 -> 29.9 us, 1,024 rows 35.6 -> 37.0: not kept; so was a per-thread table of the parameter's packed slots that cuts
    // the dependent index chain to one round trip behind the images: 30.3 -> 30.3 us.  tools/q4_trace.py: the image slices are what this
    // kernel waits for -- 16k - 19k of its 23k cycles at 512 rows, whatever is requested first)
    constexpr int UB = 8;                                            // blocks per wave in flight (a 512-row batch: all of a wave's blocks)
    for (int b0 = wave; b0 < (nsplit > 0 ? 0 : nblk); b0 += 4 * UB) {
        d4 a[UB], x[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int b = b0 + 4 * u;
            const bool ok = b < nblk;
            a[u] = ok ? *(const d4 *)(pz + (int64_t)b * N::img_doubles) : (d4){0.0, 0.0, 0.0, 0.0};
            x[u] = ok ? *(const d4 *)(px + (int64_t)b * N::img_doubles) : (d4){0.0, 0.0, 0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < UB; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = mfma(a[u][r], x[u][r], acc);
    }
    DW_T(22);
    red[wave * 64 + lane] = acc;
    __syncthreads();
    DW_T(23);
    // thread e = 4 lane' + r' owns D[n slot = g' + 4 r'][k slot = lane' & 15]  (f64 C/D map)
    const double *rf = (const double *)red;
    const int e = threadIdx.x;
    double gsum = ((rf[e] + rf[256 + e]) + rf[512 + e]) + rf[768 + e];
    if (nsplit > 0) {
        gsum = 0.0;
        const double *q = part + (int64_t)tile * nsplit * 256 + e;
        int k = 0;
        for (; k + 8 <= nsplit; k += 8) {      // eight range partials on their way at a time, added in range order
            double t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = q[(k + j) * 256];
#pragma unroll
            for (int j = 0; j < 8; ++j) gsum += t[j];
        }
        for (; k < nsplit; ++k) gsum += q[k * 256];
    }
    if (p < 0) return;
    if (grads) grads[p] = gsum;
    if (MODE == DW_ADAM) {   // elementwise.hip adam_k, on the parameters this tile owns
        double mi = pm, vi = pv;
        mi = mi + (gsum - mi) * (1.0 - ad.b1);
        vi = vi * ad.b2 + (1.0 - ad.b2) * gsum * gsum;
        const double denom = sqrt(vi) / ad.bc2_sqrt + ad.eps;
        const double pn = pp - ad.step_size * (mi / denom);
        ad.m[p] = mi;
        ad.v[p] = vi;
        ad.params[p] = pn;
        if (ad.pcopy) ad.pcopy[p] = pn;
        if (sc0 >= 0) ad.packed[sc0] = pn;
        if (sc1 >= 0) ad.packed[sc1] = pn;
        if (sc2 >= 0) ad.packed[sc2] = pn;
        if (sc3 >= 0) ad.packed[sc3] = pn;
        for (int k = s0 + 4; k < s1; ++k) ad.packed[ad.sc_idx[k]] = pn;
    }
    DW_T(24);
}
#undef DW_T

// Weight-gradient tiles in BLOCKS for large fp64 batches.  dw64_kernel's one-tile workgroups each read their two image slices of
// every 16-row block: at 262,144 rows that is 19.5 GB through L2 / fabric (3.6 of the step's 3.9 ms, MFMA busy 14 %).  Here a workgroup
// owns MN output x MK input tiles of ONE layer and walks a range of 16-row blocks (wave w: blocks w, w + 4, ..): per block a wave loads
// MN + MK image slices (32 bytes per lane each) for 4 MN MK MFMAs; the four waves' accumulators meet in LDS in wave order, the range
// partial goes to `part`, dw64_kernel(nsplit) adds the ranges in order and finishes.
//   * below 16,384 rows (Dwm64, dw64m_kernel): 2 x 4 blocks everywhere (52 blocks x 8 ranges fill the chip; tiles beyond a layer's edge
//     run on a clamped slice and are not stored: uniform code), the blocks of a layer on one XCD.
//   * from 16,384 rows on (Dwx64, dw64x_kernel): the kernel is bound by the image bytes it pulls from HBM (3.4 GB of images per 262,144
//     rows, every slice wanted by 2-13 tile blocks) and by padded tiles.  Blocks are EXACT: the smaller tile dimension of a layer stays
//     whole, the other one is dealt out in balanced pieces of <= 16 tiles per block (13 x 2 tiles -> 7 x 2 + 6 x 2, 7 x 13 -> six 7 x 2 +
//     one 7 x 1, 4 x 7 -> 4 x 4 + 4 x 3, ..: 24 blocks, 298 tile slots for 298 tiles; round 4's padded shapes had 368), and ONE XCD runs
//     all blocks of a block range side by side (workgroup b of the launch sits on XCD b & 7 and takes range 8 (j / 24) + XCD, block
//     j % 24 with j = b >> 3), one workgroup per CU, each wave with the slices of its next two blocks in flight (~330 registers).
//     Measured at 262,144 rows: the launch 1.30 -> 0.85 ms.  What the counters say about sharing: the XCD's L2 turns over
//     within about one block time (256 waves x 18 KB per block against 4 MB), so tile blocks that drift a block apart share nothing --
//     42 M lines missed per launch = 1.58 x the images with either mapping (loading the slices only one tile block wants non-temporal:
//     slower; one wave per tile block with a barrier per block: 1.12 x, and slower -- the launch is bound by latency, not by HBM bytes).
constexpr int kDw64Ahead = 2;      // dw64x_kernel: blocks whose slices are in flight ahead of a wave's MFMAs
struct DwShape { int mn, mk; };
template <class N> struct Dwm64 {
    __host__ __device__ static constexpr DwShape shape(int) { return DwShape{2, 4}; }
    __host__ __device__ static constexpr int mn(int l) { return (tiles(N::dim(l + 1)) + shape(l).mn - 1) / shape(l).mn; }
    __host__ __device__ static constexpr int mk(int l) { return (tiles(N::dim(l) + 1) + shape(l).mk - 1) / shape(l).mk; }
    __host__ __device__ static constexpr int off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += mn(j) * mk(j); return s; }
    static constexpr int total = off(N::L);
    static constexpr int per_xcd = (total + 7) / 8;
};
struct DwDeal { bool keep_n; int cnt, base, rem; };      // the dealt dimension in `cnt` pieces: `rem` of base + 1 tiles, then base tiles
template <class N> struct Dwx64 {
    __host__ __device__ static constexpr DwDeal deal(int l) {
        const int nt = tiles(N::dim(l + 1)), kt = tiles(N::dim(l) + 1);
        const bool keep_n = nt <= kt;
        const int a = keep_n ? nt : kt, b = keep_n ? kt : nt;
        const int mx = 16 / a > 0 ? 16 / a : 1, cnt = (b + mx - 1) / mx;
        return DwDeal{keep_n, cnt, b / cnt, b % cnt};
    }
    __host__ __device__ static constexpr int off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += deal(j).cnt; return s; }
    static constexpr int total = off(N::L);
};
// one tile block (compile-time shape MN x MK of layer l, first tiles n0 / k0; CLAMP: tiles beyond the layer's edge run on a clamped slice
// and are not stored): accumulate over block range `range` of `nsplit`, sum the four waves' accumulators through LDS in wave order, store
// the range partial
template <class N, int l, int MN, int MK, bool CLAMP>
__device__ __forceinline__ void dw64_tile_block(const double *__restrict__ imgs, int nblk, double *__restrict__ part, int nsplit_total, int accumulate,
                                                int n0, int k0, d4 *red, int range, int nsplit) {
    constexpr int NA = MN * MK, U = NA >= 12 ? 1 : 2, H0 = (NA + 1) / 2;
    constexpr int ntc = tiles(N::dim(l + 1)), ktc = tiles(N::dim(l) + 1), soff = N::slab_off(l), xo = N::x_off(l), zo = N::z_off(l);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
    const double *pz[MN], *px[MK];
    int nt[MN], kt[MK];
#pragma unroll
    for (int a = 0; a < MN; ++a) {
        nt[a] = n0 + a;
        const int c = !CLAMP || nt[a] < ntc ? nt[a] : ntc - 1;
        pz[a] = imgs + ((zo + 16 * c + i) * 16 + 4 * g);
    }
#pragma unroll
    for (int b = 0; b < MK; ++b) {
        kt[b] = k0 + b;
        const int c = !CLAMP || kt[b] < ktc ? kt[b] : ktc - 1;
        px[b] = imgs + ((xo + 16 * c + i) * 16 + 4 * g);
    }
    d4 acc[NA];
#pragma unroll
    for (int t = 0; t < NA; ++t) acc[t] = (d4){0.0, 0.0, 0.0, 0.0};
    const int per = (nblk + nsplit - 1) / nsplit, blo = range * per, bhi = blo + per < nblk ? blo + per : nblk;
    if constexpr (!CLAMP) {
        // the slices of a wave's next D blocks are on their way while it multiplies: D + 1 register sets, filled in turn (loads retire in
        // order, so waiting for a set leaves the D younger ones in flight; ~330 registers, one wave per SIMD); a set beyond the range is
        // loaded from the range's first block and not multiplied
        constexpr int D = kDw64Ahead;
        d4 za[D + 1][MN], xa[D + 1][MK];
        int b = blo + wave;
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const int bb = b + 4 * k;
            const int64_t o = (int64_t)(bb < bhi ? bb : blo) * N::img_doubles;
#pragma unroll
            for (int a = 0; a < MN; ++a) za[k][a] = *(const d4 *)(pz[a] + o);
#pragma unroll
            for (int c = 0; c < MK; ++c) xa[k][c] = *(const d4 *)(px[c] + o);
        }
        for (; b < bhi; b += 4 * (D + 1)) {
#pragma unroll
            for (int p = 0; p <= D; ++p) {
                const int bb = b + 4 * (p + D);
                const int64_t o = (int64_t)(bb < bhi ? bb : blo) * N::img_doubles;
#pragma unroll
                for (int a = 0; a < MN; ++a) za[(p + D) % (D + 1)][a] = *(const d4 *)(pz[a] + o);
#pragma unroll
                for (int c = 0; c < MK; ++c) xa[(p + D) % (D + 1)][c] = *(const d4 *)(px[c] + o);
                if (b + 4 * p < bhi) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < MK; ++c)
#pragma unroll
                            for (int a = 0; a < MN; ++a) acc[c * MN + a] = mfma(za[p][a][r], xa[p][c][r], acc[c * MN + a]);
                }
            }
        }
    } else
    for (int b0 = blo + wave; b0 < bhi; b0 += 4 * U) {
        d4 za[U][MN], xa[U][MK];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + 4 * u;
            const bool ok = b < bhi;
            const int64_t o = (int64_t)(ok ? b : blo) * N::img_doubles;
#pragma unroll
            for (int a = 0; a < MN; ++a) { za[u][a] = *(const d4 *)(pz[a] + o); if (!ok) za[u][a] = (d4){0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
            for (int c = 0; c < MK; ++c) xa[u][c] = *(const d4 *)(px[c] + o);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < MK; ++c)
#pragma unroll
                    for (int a = 0; a < MN; ++a) acc[c * MN + a] = mfma(za[u][a][r], xa[u][c][r], acc[c * MN + a]);
    }
    const double *rf = (const double *)red;
    const int e = threadIdx.x;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int cnt = h ? NA - H0 : H0;       // the accumulators pass through LDS in two halves (<= 8 tiles x 4 waves x 2 KB)
        if (h) __syncthreads();
#pragma unroll
        for (int t = 0; t < H0; ++t)
            if (t < cnt) red[(t * 4 + wave) * 64 + lane] = acc[h * H0 + t];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < H0; ++t) {
            if (t >= cnt) continue;
            const int ta = h * H0 + t, c = ta / MN, a = ta % MN;
            if (CLAMP && (nt[a] >= ntc || kt[c] >= ktc)) continue;
            const double *q = rf + t * 1024;
            const double gsum = ((q[e] + q[256 + e]) + q[512 + e]) + q[768 + e];
            const int tile = soff + kt[c] * ntc + nt[a];
            double *dst = part + ((int64_t)tile * nsplit_total + range) * 256 + e;
            *dst = accumulate ? *dst + gsum : gsum;      // chunks after the first add to the range's running sum (chunk order: one stream)
        }
    }
}
// Both launches cover block ranges 0 .. nsplit - 1 of the nsplit_total ranges the finishing launch adds up.  A batch beyond
// State64::chunk_rows runs chunk after chunk over the same image buffer, one launch per chunk: every chunk after the first ADDS its
// range partials to the first chunk's (`accumulate`; the launches are ordered on one stream, so the sum has a fixed order: chunk
// after chunk per range, then range after range in dw64_kernel) -- the partial buffer and the finishing launch do not grow with the batch
template <class N>
__global__ void __launch_bounds__(256) dw64m_kernel(const double *__restrict__ imgs, int nblk, double *__restrict__ part, int nsplit_total,
                                                    int accumulate) {
    using D = Dwm64<N>;
    __shared__ __attribute__((aligned(32))) d4 red[4 * 4 * 64];      // half of a block's accumulators from four waves: 32 KB
    const int mac = (blockIdx.x & 7) * D::per_xcd + (blockIdx.x >> 3);
    if (mac >= D::total) return;
    static_assert(N::L == 8, "one case per layer below");
#define BAMD_DW64M_CASE(l_)                                                                                                               \
    if (mac >= D::off(l_) && mac < D::off(l_ + 1)) {                                                                                      \
        const int idx = mac - D::off(l_), mkt = idx / D::mn(l_), mnt = idx - mkt * D::mn(l_);                                             \
        dw64_tile_block<N, l_, 2, 4, true>(imgs, nblk, part, nsplit_total, accumulate, 2 * mnt, 4 * mkt, red, (int)blockIdx.y, (int)gridDim.y); \
        return;                                                                                                                           \
    }
    BAMD_DW64M_CASE(0) BAMD_DW64M_CASE(1) BAMD_DW64M_CASE(2) BAMD_DW64M_CASE(3)
    BAMD_DW64M_CASE(4) BAMD_DW64M_CASE(5) BAMD_DW64M_CASE(6) BAMD_DW64M_CASE(7)
#undef BAMD_DW64M_CASE
}
// the blocks of layer l: the `rem` larger pieces, then the others (two shapes per layer at most)
template <class N, int l>
__device__ __forceinline__ void dw64x_layer(const double *__restrict__ imgs, int nblk, double *__restrict__ part, int nsplit_total, int accumulate,
                                            int idx, d4 *red, int range, int nsplit) {
    constexpr DwDeal d = Dwx64<N>::deal(l);
    constexpr int nt = tiles(N::dim(l + 1)), kt = tiles(N::dim(l) + 1);
    if constexpr (d.rem > 0) {
        if (idx < d.rem) {
            const int at = idx * (d.base + 1);
            if constexpr (d.keep_n) dw64_tile_block<N, l, nt, d.base + 1, false>(imgs, nblk, part, nsplit_total, accumulate, 0, at, red, range, nsplit);
            else dw64_tile_block<N, l, d.base + 1, kt, false>(imgs, nblk, part, nsplit_total, accumulate, at, 0, red, range, nsplit);
            return;
        }
    }
    const int at = d.rem * (d.base + 1) + (idx - d.rem) * d.base;
    if constexpr (d.keep_n) dw64_tile_block<N, l, nt, d.base, false>(imgs, nblk, part, nsplit_total, accumulate, 0, at, red, range, nsplit);
    else dw64_tile_block<N, l, d.base, kt, false>(imgs, nblk, part, nsplit_total, accumulate, at, 0, red, range, nsplit);
}
template <class N>
__global__ void __launch_bounds__(256) dw64x_kernel(const double *__restrict__ imgs, int nblk, double *__restrict__ part, int nsplit_total,
                                                    int accumulate, int nsplit) {
    using D = Dwx64<N>;
    __shared__ __attribute__((aligned(32))) d4 red[8 * 4 * 64];      // half of a block's accumulators from four waves: 64 KB
    const int j = (int)(blockIdx.x >> 3), range = (j / D::total) * 8 + (int)(blockIdx.x & 7), mac = j % D::total;
    if (range >= nsplit) return;
    static_assert(N::L == 8, "one case per layer below");
#define BAMD_DW64X_CASE(l_) \
    if (mac >= D::off(l_) && mac < D::off(l_ + 1)) { dw64x_layer<N, l_>(imgs, nblk, part, nsplit_total, accumulate, mac - D::off(l_), red, range, nsplit); return; }
    BAMD_DW64X_CASE(0) BAMD_DW64X_CASE(1) BAMD_DW64X_CASE(2) BAMD_DW64X_CASE(3)
    BAMD_DW64X_CASE(4) BAMD_DW64X_CASE(5) BAMD_DW64X_CASE(6) BAMD_DW64X_CASE(7)
#undef BAMD_DW64X_CASE
}

__global__ void __launch_bounds__(256) pack64_k(const double *__restrict__ params, const int *__restrict__ src, int count,
                                                double *__restrict__ packed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) packed[i] = src[i] >= 0 ? params[src[i]] : 0.0;
}

struct Ops64;
struct State64 {
    const Ops64 *ops = nullptr;
    DevBuf pack_src, inv_map, sc_off, sc_idx, packed, imgs, dwpart;
    int packed_doubles = 0;
    // The fused pair stays ahead of the layer-wise kernels at every size measured (us per step, fused / layer-wise: 16384 rows 227 / 847,
    // 65536 rows 850 / 2666).  Its images take 13 KB per row (3.5 GB at 262144 rows), so a larger batch runs CHUNK AFTER CHUNK over the
    // same image buffer: chain + tile blocks per chunk, the partial tiles of all chunks' block ranges added in order by ONE finishing
    // launch (round 4; before, batches above 262144 rows fell to the layer-wise kernels at ~0.20 of the fp64 peak).
    // BALER_AMD_LATENCY_ROWS (as for fp32) caps the rows the fused pair takes at all: larger batches then run layer by layer.
    int64_t chunk_rows = 262144;
    int64_t max_rows = INT64_MAX;
};
struct Ops64 {
    int (*setup)(bamd_handle *, State64 *);
    int (*step)(bamd_handle *, State64 *, const void *, int, int64_t, const double *, double *, const Adam64 *, hipStream_t);
    // kind (I_ENCODE / I_DECODE / I_FORWARD), input, its dtype, rows, features applied to the input, output (+ dtype),
    // un-normalisation of the output, int-column mask, loss sum (I_FORWARD)
    int (*infer)(bamd_handle *, State64 *, int, const void *, int, int64_t, const double *, void *, int, const double *, const uint8_t *,
                 double *, hipStream_t);
};
State64 *st64(bamd_handle *h) { return (State64 *)h->fused64_state; }

// The fused fp64 training pass of any instantiation: chunks over one image buffer, per chunk the chain launch (`chain`: which of the
// three chains, decided by the instantiation) and -- from 64 blocks on -- the tile-block launch over block ranges, then ONE finishing
// launch (the tiles themselves for a small batch; loss, range sums, store or Adam).  fw = the table's real width.
template <class N, class ChainFn>
int step64_common(bamd_handle *h, State64 *st, int64_t n, double *grads, const Adam64 *ad, hipStream_t s, int fw, ChainFn chain) {
    const int64_t nblk_all = (n + 15) / 16;
    if (nblk_all > (int64_t)1 << 30) { set_error("fp64 fused step: batch too large"); return BAMD_ERR_INVALID; }
    const int64_t chunk = st->chunk_rows & ~(int64_t)15;
    const int nchunk = (int)((n + chunk - 1) / chunk);
    const int nblk_max = (int)((std::min(n, chunk) + 15) / 16);
    int rc = st->imgs.ensure((size_t)N::img_doubles * sizeof(double) * (size_t)nblk_max);
    if (rc) return rc;
    rc = h->lossp.ensure(sizeof(double) * (size_t)(4 * nblk_all > 1024 ? 4 * nblk_all : 1024));      // (four partials per block with the 4-row chain)
    if (rc) return rc;
    const dim3 grid(8 * ((N::slab_off(N::L) + 1 + 7) / 8));
    // from 64 blocks (1,024 rows) on: 2 x 4 tile blocks over block ranges + the finishing launch (BALER_AMD_DW64_MACRO_BLKS, 0 = off).
    // Measured ms per bamd_fwd_bwd, blocks / one tile per workgroup: 512 rows 0.045 / 0.040, 1,024: 0.046 / 0.049, 2,048: 0.054 / 0.064,
    // 4,096: 0.068 / 0.094, 16,384: 0.227 / 0.341, 65,536: 0.85 / 1.32, 262,144: 3.22 / 5.51 (0.37 / 0.22 of the fp64 MFMA peak)
    const int macro_blks = (int)env_ll("BALER_AMD_DW64_MACRO_BLKS", 64);
    const bool macro = nchunk > 1 || (macro_blks > 0 && nblk_all >= macro_blks);
    // tile-block shape by batch size; block ranges per chunk (tile blocks x ranges = workgroups): from 1,024 blocks on the exact
    // per-layer blocks (24 of them, one workgroup per CU) over 32 ranges = three full rounds of the chip (BALER_AMD_DW64_RANGES;
    // measured at 262,144 rows 32 / 64 / 128 ranges: 2.11 / 2.15 / 2.21 ms per bamd_fwd_bwd); below that 2 x 4 blocks (52) over 8
    // ranges, at least 4 blocks per range.  Every chunk but the last is a full one
    const bool big = nblk_all >= 1024;
    const int big_ranges = (int)std::max<long long>(1, env_ll("BALER_AMD_DW64_RANGES", 32));
    // every range must own at least one block: the kernels prefetch block `range * ceil(blks / ranges)` unconditionally, which an
    // empty trailing range would read past the images (e.g. 128 ranges of 1,025 blocks: 9 blocks each, range 114 starts at 1,026)
    auto splits_of = [big, big_ranges](int64_t blks) {
        int64_t ns = std::min<int64_t>(big ? big_ranges : 8, std::max<int64_t>(1, blks / 4));
        while (ns > 1 && (ns - 1) * ((blks + ns - 1) / ns) >= blks) --ns;
        return (int)ns;
    };
    int nsplit = 0;
    if (macro) {
        // the first chunk is the largest (every chunk but the last is a full one): its range count is the buffer's; 32 ranges by
        // default, i.e. (tiles + 1) x 32 x 2 KB = 19.6 MB whatever the batch
        nsplit = splits_of((std::min(n, chunk) + 15) / 16);
        rc = st->dwpart.ensure((size_t)(N::slab_off(N::L) + 1) * nsplit * 256 * sizeof(double));
        if (rc) return rc;
    }
    bool quad = false;
    for (int k = 0; k < nchunk; ++k) {
        const int64_t r0 = k * chunk, rows = std::min(n - r0, chunk);
        const int nblk = (int)((rows + 15) / 16);
        const int per_blk = chain(k, r0, rows, nblk, nblk_all, nchunk);      // the chunk's chain launch; loss partials per 16-row block (1 or 4)
        if (per_blk < 0) return per_blk;
        quad = per_blk == 4;
        if (macro) {
            const int ns = splits_of(nblk);
            if (big)
                hipLaunchKernelGGL((dw64x_kernel<N>), dim3(8 * ((ns + 7) / 8) * Dwx64<N>::total), dim3(256), 0, s, (const double *)st->imgs.p, nblk,
                                   (double *)st->dwpart.p, nsplit, k > 0, ns);
            else
                hipLaunchKernelGGL((dw64m_kernel<N>), dim3(8 * Dwm64<N>::per_xcd, ns), dim3(256), 0, s, (const double *)st->imgs.p, nblk,
                                   (double *)st->dwpart.p, nsplit, k > 0);
        }
    }
    const double *part = macro ? (const double *)st->dwpart.p : nullptr;
    // the finishing launch: the loss partials of ALL blocks; with `part` the sum over all chunks' block ranges in order, else
    // (one small chunk) the tiles themselves over the images
    const int nblk_fin = (int)nblk_all, nloss = quad ? 4 * nblk_fin : nblk_fin;
    if (ad)
        hipLaunchKernelGGL((dw64_kernel<N, DW_ADAM>), grid, dim3(256), 0, s, (const double *)st->imgs.p, nblk_fin, (const double *)h->lossp.p,
                           (const int *)st->inv_map.p, grads, *ad, part, nsplit, (int)h->nparams, 1.0 / fw, nloss);
    else
        hipLaunchKernelGGL((dw64_kernel<N, DW_WRITE>), grid, dim3(256), 0, s, (const double *)st->imgs.p, nblk_fin, (const double *)h->lossp.p,
                           (const int *)st->inv_map.p, grads, Adam64{}, part, nsplit, (int)h->nparams, 1.0 / fw, nloss);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

// The maps of a handle: which packed slot (both fragment orders, biases) holds which canonical parameter, which thread of which
// weight-gradient tile owns which parameter, and the CSR scatter lists the Adam kernels refresh the packed copies through.  Geometry
// (tiles, fragment order, slot -> feature) from the instantiated Net64; which slots hold a parameter, and its canonical index, from
// the handle's REAL dimensions (identical for an exact instantiation).
template <class N>
int build_maps64(bamd_handle *h, State64 *st) {
    const int nparams = (int)h->nparams;
    auto dimr = [&](int i) { return h->dims[i]; };
    auto woff = [&](int l) { return (int)h->w_off[l]; };
    auto boff = [&](int l) { return (int)h->b_off[l]; };
    std::vector<int> src((size_t)N::packed_all_doubles(), -1);
    // the 4-row chain's copy (chain64q_kernel): fragment (k8, grp) of GEMM g, lane l, half h = A[16 grp + (l & 15)][8 k8 + 4 h + (l >> 4)]
    // with A = W_g (forward) or W_{15-g}^T (backward), then the biases in natural order
    {
        const size_t q0 = (size_t)N::packed_d4() * 4;
        for (int g = 0; g < 15; ++g) {
            const int l = N::q_layer(g), G = N::q_groups(g), Kr = dimr(l), NNr = dimr(l + 1);
            for (int k8 = 0; k8 < N::q_ks8(g); ++k8)
                for (int grp = 0; grp < G; ++grp)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int hh = 0; hh < 2; ++hh) {
                            const int f = 16 * grp + (lane & 15), k = 8 * k8 + 4 * hh + (lane >> 4);
                            const size_t o = q0 + ((size_t)(N::q_frag_off(g) + k8 * G + grp) * 64 + lane) * 2 + hh;
                            if (g < 8) { if (f < NNr && k < Kr) src[o] = woff(l) + f * Kr + k; }          // W_l[f][k]
                            else if (f < Kr && k < NNr) src[o] = woff(l) + k * Kr + f;                     // W_l[k][f]
                        }
        }
        for (int l = 0; l < N::L; ++l)
            for (int f = 0; f < dimr(l + 1); ++f) src[q0 + (size_t)N::q_frags() * 128 + N::qb_off(l) + f] = boff(l) + f;
    }
    for (int l = 0; l < N::L; ++l) {
        const int K = N::dim(l), NN = N::dim(l + 1), KT = tiles(K), NT = tiles(NN), Kr = dimr(l), NNr = dimr(l + 1);
        // forward fragment (q, t): lane (i, g) component r = W[16 t + i][16 q + 4 r + g]
        for (int q = 0; q < KT; ++q)
            for (int t = 0; t < NT; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int n = 16 * t + (lane & 15), k = creg_feature(K, q, lane >> 4, r);
                        if (n < NNr && k >= 0 && k < Kr) src[((size_t)N::wf_off(l) + (q * NT + t) * 64 + lane) * 4 + r] = woff(l) + n * Kr + k;
                    }
        // backward fragment (tq, tk), l >= 1: lane (i, g) component r = W[16 tq + 4 r + g][16 tk + i]
        for (int tq = 0; tq < NT && l >= 1; ++tq)
            for (int tk = 0; tk < KT; ++tk)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int n = creg_feature(NN, tq, lane >> 4, r), k = 16 * tk + (lane & 15);
                        if (n >= 0 && n < NNr && k < Kr) src[((size_t)N::wb_off(l) + (tq * KT + tk) * 64 + lane) * 4 + r] = woff(l) + n * Kr + k;
                    }
        // bias fragment (t, g) component r = b[16 t + 4 r + g]
        for (int t = 0; t < NT; ++t)
            for (int g = 0; g < 4; ++g)
                for (int r = 0; r < 4; ++r) {
                    const int n = creg_feature(NN, t, g, r);
                    if (n >= 0 && n < NNr) src[((size_t)N::bf_off(l) + t * 4 + g) * 4 + r] = boff(l) + n;
                }
    }
    // weight-gradient tile (kt, nt) of layer l: thread e = 4 lane + r holds dW[16 nt + g + 4 r][16 kt + (lane & 15)]; column K = db
    const int ntiles = N::slab_off(N::L);
    std::vector<int> inv((size_t)ntiles * 256, -1);
    for (int l = 0; l < N::L; ++l) {
        const int K = N::dim(l), NN = N::dim(l + 1), NT = tiles(NN), Kr = dimr(l), NNr = dimr(l + 1);
        for (int kt = 0; kt < tiles(K + 1); ++kt)
            for (int nt = 0; nt < NT; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        const int n = creg_feature(NN, nt, lane >> 4, r), kc = 16 * kt + (lane & 15);
                        if (n < 0 || n >= NNr) continue;
                        const size_t o = ((size_t)(N::slab_off(l) + kt * NT + nt) * 64 + lane) * 4 + r;
                        if (kc < K) { if (kc < Kr) inv[o] = woff(l) + n * Kr + kc; }
                        else if (kc == K) inv[o] = boff(l) + n;      // (the ones slot sits at the CLASS width)
                    }
    }
    {
        std::vector<char> seen(nparams, 0);
        for (int v : inv) if (v >= 0) seen[v]++;
        for (char c : seen) if (c != 1) { set_error("fp64 fused step: incomplete gradient map"); return BAMD_ERR_INVALID; }
    }
    std::vector<int> off((size_t)nparams + 1, 0), idx;
    for (int v : src) if (v >= 0) off[v + 1]++;
    for (int p = 0; p < nparams; ++p) off[p + 1] += off[p];
    idx.resize(off[nparams]);
    std::vector<int> cur(off.begin(), off.end() - 1);
    for (size_t i = 0; i < src.size(); ++i) if (src[i] >= 0) idx[cur[src[i]]++] = (int)i;
    st->packed_doubles = (int)src.size();
    int rc = st->pack_src.ensure(src.size() * sizeof(int));
    if (!rc) rc = st->inv_map.ensure(inv.size() * sizeof(int));
    if (!rc) rc = st->sc_off.ensure(off.size() * sizeof(int));
    if (!rc) rc = st->sc_idx.ensure(idx.size() * sizeof(int));
    if (!rc) rc = st->packed.ensure(src.size() * sizeof(double));
    if (rc) return rc;
    BAMD_HIP(hipMemcpy(st->pack_src.p, src.data(), src.size() * sizeof(int), hipMemcpyHostToDevice));
    BAMD_HIP(hipMemcpy(st->inv_map.p, inv.data(), inv.size() * sizeof(int), hipMemcpyHostToDevice));
    BAMD_HIP(hipMemcpy(st->sc_off.p, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
    BAMD_HIP(hipMemcpy(st->sc_idx.p, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    return BAMD_OK;
}

// RT: the instantiation serves every AE(f <= F, z <= Z) with the reference's hidden widths (as Impl<F, Z, true> in fused.hip): the
// geometry is the class's, the maps and the row I/O follow the handle's real dimensions
template <int F, int Z, bool RT = false> struct Impl64 {
    using N = Net64<F, Z>;
    static int fr(const bamd_handle *h) { return h->dims[0]; }
    static int zr(const bamd_handle *h) { return h->dims[4]; }
    static constexpr int kLds = (2 * 13 * 64 + (N::bf_off(N::L) - N::bf_off(0))) * 32;
    static constexpr int kLdsR = (N::bf_off(N::L) - N::bf_off(0)) * 32;       // chain64r_kernel: the bias fragments only
    static bool matches(const bamd_handle *h) {
        if (h->L != 8) return false;
        if (RT) {
            for (int i = 1; i <= 7; ++i)
                if (i != 4 && h->dims[i] != N::dim(i)) return false;
            return h->dims[0] == h->dims[8] && h->dims[0] >= 1 && h->dims[0] <= F && h->dims[4] >= 1 && h->dims[4] <= Z;
        }
        for (int i = 0; i <= 8; ++i)
            if (h->dims[i] != N::dim(i)) return false;
        return true;
    }
    static int setup(bamd_handle *h, State64 *st) {
        if (const int rc = build_maps64<N>(h, st)) return rc;
        BAMD_HIP(hipFuncSetAttribute((const void *)chain64_kernel<F, Z, 4, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
        return BAMD_OK;
    }
    static int step(bamd_handle *h, State64 *st, const void *x, int x_dtype, int64_t n, const double *features, double *grads,
                    const Adam64 *ad, hipStream_t s) {
        const size_t xes = x_dtype == BAMD_F64 ? 8 : 4;
        auto chain = [&](int k, int64_t r0, int64_t rows, int nblk, int64_t nblk_all, int nchunk) -> int {
            (void)k;
            int rc = BAMD_OK;
            bool quad = false;
            // one wave per 16-row block (register chain) once the blocks fill the chip's wave slots (1,024 blocks = one wave per SIMD:
            // 16,384 rows 0.233 -> 0.201 ms, 8,192 rows 0.123 -> 0.141); one workgroup per block below.  BALER_AMD_F64_REGCHAIN_BLKS
            // moves the switch (0: always, read per call: the parity tests run both kernels on the same batch)
            const char *re = getenv("BALER_AMD_F64_REGCHAIN_BLKS");
            const int64_t rmin = re ? atoll(re) : 1024;
            // up to BALER_AMD_F64_QCHAIN_BLKS blocks (default 96 = 1,536 rows; 0: never): FOUR rows per workgroup on v_mfma_f64_4x4x4
            // (chain64q_kernel): the reference's 512-row batch on 128 CUs instead of 32
            quad = nchunk == 1 && nblk_all < rmin && nblk_all <= env_ll("BALER_AMD_F64_QCHAIN_BLKS", 96);
            if (quad) {
                rc = fused64q_launch(F, Z, RT, 4u * (unsigned)nblk, s, (const double *)st->packed.p + (size_t)N::packed_d4() * 4, x,
                                     x_dtype == BAMD_F64, rows, features, (double *)st->imgs.p, (double *)h->lossp.p, fr(h));
                if (rc) return rc;
            } else if (nblk_all >= rmin)
                hipLaunchKernelGGL((chain64r_kernel<F, Z, RT>), dim3((nblk + 3) / 4), dim3(256), kLdsR, s, (const d4 *)st->packed.p,
                                   (const void *)((const char *)x + (size_t)r0 * fr(h) * xes), x_dtype == BAMD_F64, rows, features, (double *)st->imgs.p,
                                   (double *)h->lossp.p + r0 / 16, nblk, fr(h));
            else
                hipLaunchKernelGGL((chain64_kernel<F, Z, 4, RT>), dim3(nblk), dim3(256), kLds, s, (const d4 *)st->packed.p,
                                   (const void *)((const char *)x + (size_t)r0 * fr(h) * xes), x_dtype == BAMD_F64, rows, features, (double *)st->imgs.p,
                                   (double *)h->lossp.p + r0 / 16, fr(h));
            return quad ? 4 : 1;
        };
        return step64_common<N>(h, st, n, grads, ad, s, fr(h), chain);
    }
    static int infer(bamd_handle *h, State64 *st, int kind, const void *x, int x_dtype, int64_t n, const double *features, void *out,
                     int out_dtype, const double *renorm, const uint8_t *imask, double *loss_sum, hipStream_t s) {
        return fused64_infer_launch(F, Z, RT, h, (const double *)st->packed.p, kind, x, x_dtype, n, features, out, out_dtype, renorm, imask, loss_sum, s);
    }
    static const Ops64 *ops() {
        static const Ops64 o = {setup, step, infer};
        return &o;
    }
};

// 64 .. 127 columns (latent <= 63), and up to 63 columns with a latent of 32 .. 63, in fp64 -- chain64q_kernel (four rows per workgroup; its input rows are two
// feature slots per thread) + the common weight-gradient launches for EVERY training batch (chunks of 65,536 rows): the reference's batch_size = 512 in the
// reference's dtype for its wider tables (models.py:128-136 builds AE(n_features, z_dim) in float64 for any table) runs 3.8x, 4,096 rows 6x,
// 65,536 rows 2.2x faster than on the layer-wise kernels.  The exchange chain and the register chain give ONE input tile to a wave
// (<= 63 columns) and are never instantiated for these classes; encode / decode / forward + loss run on the register-chained inference
// kernel, which is generic in the tile counts (fused64j.hip).
template <int F, int FLO, int Z, int ZLO = 0> struct Impl64Q {
    using N = Net64<F, Z>;
    static bool matches(const bamd_handle *h) {
        if (h->L != 8) return false;
        for (int i = 1; i <= 7; ++i)
            if (i != 4 && h->dims[i] != N::dim(i)) return false;
        return h->dims[0] == h->dims[8] && h->dims[0] > FLO && h->dims[0] <= F && h->dims[4] > ZLO && h->dims[4] <= Z;
    }
    static int setup(bamd_handle *h, State64 *st) {
        return build_maps64<N>(h, st);
    }
    static int step(bamd_handle *h, State64 *st, const void *x, int x_dtype, int64_t n, const double *features, double *grads,
                    const Adam64 *ad, hipStream_t s) {
        // The common pass (chunks, tile blocks over block ranges from 1,024 rows on, one finishing launch) with the 4-row chain for EVERY
        // chunk: chunks of BALER_AMD_F64_QCHAIN_BLKS 16-row blocks (default here 4,096 = 65,536 rows, 15 KB of images per row; 0: the
        // layer-wise kernels).  Measured us per step, this path with one-tile weight gradients / layer-wise, AE(80, 16): 512 rows 34 / 134,
        // 1,536: 61 / 849, 4,096: 132 / 839, 16,384: 469 / 892, 65,536: 1,820 / 2,780 (profiles/r6_fp64_mid_width_step.txt)
        const int64_t lim = env_ll("BALER_AMD_F64_QCHAIN_BLKS", -1);
        if (lim == 0) return BAMD_ERR_UNSUPPORTED;                                     // the caller runs the layer-wise kernels
        st->chunk_rows = (lim > 0 ? lim : 4096) * 16;
        const size_t row_bytes = (size_t)h->dims[0] * (x_dtype == BAMD_F64 ? 8 : 4);
        auto chain = [&](int, int64_t r0, int64_t rows, int nblk, int64_t, int) -> int {
            const int rc = fused64q_launch(F, Z, true, 4u * (unsigned)nblk, s, (const double *)st->packed.p + (size_t)N::packed_d4() * 4,
                                           (const char *)x + (size_t)r0 * row_bytes, x_dtype == BAMD_F64, rows, features, (double *)st->imgs.p,
                                           (double *)h->lossp.p + 4 * (r0 / 16), h->dims[0]);
            return rc ? rc : 4;
        };
        return step64_common<N>(h, st, n, grads, ad, s, h->dims[0], chain);
    }
    // encode / decode / forward + loss: the register-chained inference kernel is generic in the tile counts (a wave owns every tile of its 16
    // rows); its instantiations for these classes live in fused64j.hip
    static int infer(bamd_handle *h, State64 *st, int kind, const void *x, int x_dtype, int64_t n, const double *features, void *out,
                     int out_dtype, const double *renorm, const uint8_t *imask, double *loss_sum, hipStream_t s) {
        return fused64_infer_launch(F, Z, true, h, (const double *)st->packed.p, kind, x, x_dtype, n, features, out, out_dtype, renorm, imask, loss_sum, s);
    }
    static const Ops64 *ops() {
        static const Ops64 o = {setup, step, infer};
        return &o;
    }
};

const Ops64 *find64(const bamd_handle *h) {
    if (h->mode != BAMD_MODE_F64) return nullptr;
    if (Impl64<24, 15>::matches(h)) return Impl64<24, 15>::ops();
    if (Impl64<24, 12>::matches(h)) return Impl64<24, 12>::ops();
    if (Impl64<24, 8>::matches(h)) return Impl64<24, 8>::ops();
    if (Impl64<24, 6>::matches(h)) return Impl64<24, 6>::ops();
    if (Impl64<24, 10>::matches(h)) return Impl64<24, 10>::ops();
    if (Impl64<24, 5>::matches(h)) return Impl64<24, 5>::ops();
    if (Impl64<24, 4>::matches(h)) return Impl64<24, 4>::ops();
    if (Impl64<24, 3>::matches(h)) return Impl64<24, 3>::ops();
    if (Impl64<24, 2>::matches(h)) return Impl64<24, 2>::ops();
    // any other narrow table: class instantiations with run-time widths (the exchange chain gives one input tile to a wave: up to
    // 63 columns; a latent of up to 31)
    if (Impl64<31, 15, true>::matches(h)) return Impl64<31, 15, true>::ops();
    if (Impl64<47, 15, true>::matches(h)) return Impl64<47, 15, true>::ops();
    if (Impl64<63, 15, true>::matches(h)) return Impl64<63, 15, true>::ops();
    if (Impl64<31, 31, true>::matches(h)) return Impl64<31, 31, true>::ops();
    if (Impl64<63, 31, true>::matches(h)) return Impl64<63, 31, true>::ops();
    // 64 .. 127 columns: the small-batch step only (Impl64Q)
    if (Impl64Q<79, 63, 31>::matches(h)) return Impl64Q<79, 63, 31>::ops();
    if (Impl64Q<95, 79, 31>::matches(h)) return Impl64Q<95, 79, 31>::ops();
    if (Impl64Q<111, 95, 31>::matches(h)) return Impl64Q<111, 95, 31>::ops();
    if (Impl64Q<127, 111, 31>::matches(h)) return Impl64Q<127, 111, 31>::ops();
    // ... and a latent of 32 .. 63 (up to 127 columns) the same way
    if (Impl64Q<63, 0, 63, 31>::matches(h)) return Impl64Q<63, 0, 63, 31>::ops();
    if (Impl64Q<79, 63, 63, 31>::matches(h)) return Impl64Q<79, 63, 63, 31>::ops();
    if (Impl64Q<95, 79, 63, 31>::matches(h)) return Impl64Q<95, 79, 63, 31>::ops();
    if (Impl64Q<111, 95, 63, 31>::matches(h)) return Impl64Q<111, 95, 63, 31>::ops();
    if (Impl64Q<127, 111, 63, 31>::matches(h)) return Impl64Q<127, 111, 63, 31>::ops();
    return nullptr;
}

}  // namespace

int fused64_setup(bamd_handle *h) {
    const Ops64 *ops = find64(h);
    if (!ops) return BAMD_OK;
    const char *env = getenv("BALER_AMD_FORCE_GENERIC");
    if (env && env[0] == '1') return BAMD_OK;
    State64 *st = new State64();
    st->ops = ops;
    if (const char *lr = getenv("BALER_AMD_LATENCY_ROWS")) st->max_rows = atoll(lr);
    if (const char *cr = getenv("BALER_AMD_F64_CHUNK_ROWS")) st->chunk_rows = std::max<int64_t>(16, atoll(cr));      // tests: several chunks at small sizes
    h->fused64_state = st;
    return ops->setup(h, st);
}

void fused64_teardown(bamd_handle *h) {
    State64 *st = st64(h);
    if (!st) return;
    st->pack_src.release(); st->inv_map.release(); st->sc_off.release(); st->sc_idx.release(); st->packed.release(); st->imgs.release(); st->dwpart.release();
    delete st;
    h->fused64_state = nullptr;
}

int fused64_pack(bamd_handle *h, hipStream_t s) {
    State64 *st = st64(h);
    if (!st) return BAMD_OK;
    hipLaunchKernelGGL(pack64_k, dim3((st->packed_doubles + 255) / 256), dim3(256), 0, s, (const double *)h->params.p,
                       (const int *)st->pack_src.p, st->packed_doubles, (double *)st->packed.p);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

void fused64_scatter(bamd_handle *h, const int **sc_off, const int **sc_idx, void **packed) {
    State64 *st = st64(h);
    if (!st) return;
    *sc_off = (const int *)st->sc_off.p;
    *sc_idx = (const int *)st->sc_idx.p;
    *packed = st->packed.p;
}

// fwd + loss + bwd of a small batch (returns BAMD_ERR_UNSUPPORTED when this handle / batch size has no fp64 fused path);
// with `hp`: also Adam and the refresh of the packed weights, in the weight-gradient kernel
int fused64_step(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads, void *params, void *m,
                 void *v, const bamd_adam *hp, double *loss_accum, hipStream_t s) {
    State64 *st = st64(h);
    if (!st || n > st->max_rows || n <= 0) return BAMD_ERR_UNSUPPORTED;
    if (!hp) return st->ops->step(h, st, x, x_dtype, n, features, (double *)grads, nullptr, s);
    Adam64 ad;
    ad.params = (double *)params; ad.pcopy = (double *)h->params.p; ad.m = (double *)m; ad.v = (double *)v;
    ad.packed = (double *)st->packed.p;
    ad.sc_off = (const int *)st->sc_off.p; ad.sc_idx = (const int *)st->sc_idx.p;
    ad.loss_accum = loss_accum;
    ad.b1 = hp->beta1; ad.b2 = hp->beta2; ad.eps = hp->eps;      // same scalars as launch_adam (elementwise.hip)
    ad.step_size = hp->lr / (1.0 - pow(hp->beta1, (double)hp->step));
    ad.bc2_sqrt = sqrt(1.0 - pow(hp->beta2, (double)hp->step));
    return st->ops->step(h, st, x, x_dtype, n, features, (double *)grads, &ad, s);
}


// encode / decode / forward + loss of an F64 handle on the register-chained kernel (BAMD_ERR_UNSUPPORTED: no fp64 fused path for this
// shape -> the caller falls back to the layer-wise kernels)
int fused64_infer(bamd_handle *h, int kind, const void *x, int x_dtype, int64_t n, const double *features, void *out, int out_dtype,
                  const double *renorm, const uint8_t *int_mask, double *loss_sum, hipStream_t s) {
    State64 *st = st64(h);
    static const bool on = !(getenv("BALER_AMD_F64_INFER") && getenv("BALER_AMD_F64_INFER")[0] == '0');
    if (!st || !on || n <= 0) return BAMD_ERR_UNSUPPORTED;
    return st->ops->infer(h, st, kind, x, x_dtype, n, features, out, out_dtype, renorm, int_mask, loss_sum, s);
}

}  // namespace bamd

#ifdef BAMD_Q4_TRACE
// debug builds only (tools/q4_trace.py): the stamps of one dw64_kernel workgroup.  Not part of the ABI: the shipped library does not export it.
extern "C" int bamd_debug_dw64_trace(unsigned long long *dst, int count) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(bamd::g_dw64_trace), sizeof(unsigned long long) * (count < 8 ? count : 8));
}
#endif
