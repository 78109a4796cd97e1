// fp64 register-chained inference (encode / decode / forward + loss), shared by fused64i.hip and fused64j.hip: each translation unit
// instantiates infer64_kernel for its own list of shapes.  Geometry and the low-level helpers: fused64_net.hpp.
#pragma once
#include "fused64_net.hpp"

#include <cstdint>
#include <cstdlib>
#include <utility>

namespace bamd {
namespace {

// ---- fp64 throughput inference: encode / decode / forward + loss at any row count ---------------------------------------------
// The reference computes in fp64 (models.py:128-136); until round 3 bamd_encode / bamd_decode / bamd_forward_loss of an F64 handle
// ran layer by layer (activations through HBM, LDS-tiled GEMMs).  Here every WAVE pushes its own 16 rows through the layers with
// the activations in registers (the transposed register chain above with W = 1: a wave owns every tile of its rows, so there is no
// exchange and no barrier), 4 waves per workgroup, persistent over row tiles.  A wave streams the half-model's fragments through
// a ring that wraps across row tiles: 2 KiB per 4 MFMAs of 64 cycles = 8 B/clk per wave, far below the per-wave load rate.
enum { I_ENCODE = 0, I_DECODE = 1, I_FORWARD = 2 };
template <class N, int G0, int NGM, int D_> struct ISeq {      // GEMMs G0 .. G0 + NGM - 1 (forward fragments), one wave = all tiles
    static constexpr int D = D_;
    __host__ __device__ static constexpr int kd(int g) { return N::dim(G0 + g); }
    __host__ __device__ static constexpr int nt(int g) { return tiles(N::dim(G0 + g + 1)); }
    __host__ __device__ static constexpr int base(int g) { return N::wf_off(G0 + g) / 64; }
    __host__ __device__ static constexpr int nf(int g) { return tiles(kd(g)) * nt(g); }
    __host__ __device__ static constexpr int start(int g) { int s = 0; for (int j = 0; j < g; ++j) s += nf(j); return s; }
    static constexpr int real = start(NGM);
    static constexpr int total = (real + D - 1) / D * D;          // padded so that fragment S always lives in ring slot S % D
    __host__ __device__ static constexpr int gemm_of(int S) { int g = 0; for (int j = 1; j < NGM; ++j) if (S >= start(j)) g = j; return g; }
};
template <class SQ, int S>
__device__ __forceinline__ void iseq_issue(d4 (&slot)[SQ::D], const WStream &ws) {
    constexpr int Sm = S % SQ::total;
    if constexpr (Sm < SQ::real) {
        constexpr int g = SQ::gemm_of(Sm), f = Sm - SQ::start(g), NT = SQ::nt(g);
        slot[S % SQ::D] = frag_rt(ws, SQ::base(g) + (f / NT) * NT + f % NT);
    }
}
template <class SQ, int... S>
__device__ __forceinline__ void iseq_span(d4 (&slot)[SQ::D], const WStream &ws, std::integer_sequence<int, S...>, int) {
    (iseq_issue<SQ, S>(slot, ws), ...);
}
template <class SQ, int S0, int... S>
__device__ __forceinline__ void iseq_tail(d4 (&slot)[SQ::D], const WStream &ws, std::integer_sequence<int, S...>) {
    (iseq_issue<SQ, S0 + S + SQ::D>(slot, ws), ...);
}
template <class SQ, int g, int f>
__device__ __forceinline__ void iseq_one(const d4 (&in)[tiles(SQ::kd(g))], d4 (&out)[SQ::nt(g)], d4 (&slot)[SQ::D], const WStream &ws) {
    constexpr int NT = SQ::nt(g), S0 = SQ::start(g), KD = SQ::kd(g), q = f / NT, i = f % NT, s = (S0 + f) % SQ::D;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (r < tile_steps(KD, q)) out[i] = mfma(slot[s][r], in[q][r], out[i]);
    iseq_issue<SQ, S0 + f + SQ::D>(slot, ws);
    __builtin_amdgcn_sched_barrier(0);
}
template <class SQ, int g, int... P>
__device__ __forceinline__ void iseq_mm_impl(const d4 (&in)[tiles(SQ::kd(g))], d4 (&out)[SQ::nt(g)], d4 (&slot)[SQ::D], const WStream &ws,
                                             std::integer_sequence<int, P...>) {
    (iseq_one<SQ, g, P>(in, out, slot, ws), ...);
}
// GEMM g: out (initialised with the bias) += W_l in, then the activation
template <class N, class SQ, int g, int G0>
__device__ __forceinline__ void ilayer(const d4 (&in)[tiles(SQ::kd(g))], d4 (&out)[SQ::nt(g)], d4 (&slot)[SQ::D], const WStream &ws,
                                       const d4 *bias_lds, int lg) {
#pragma unroll
    for (int t = 0; t < SQ::nt(g); ++t) out[t] = bias_lds[(N::bf_off(G0 + g) - N::bf_off(0)) + t * 4 + lg];
    iseq_mm_impl<SQ, g>(in, out, slot, ws, std::make_integer_sequence<int, SQ::nf(g)>{});
    if (N::act(G0 + g)) lrelu(out);
}
// rows of width D_ -> register tiles in the f64 accumulator layout (register r of tile t on lane group g = feature 16 t + 4 r + g)
// RT: the instantiation serves a CLASS of narrow tables (D_ = 16 T - 1 is the class width, `dr` the table's real width: row stride,
// valid features and the min / range pairs come from it; the class's slots beyond it are zeros that meet zero weights) -- as in fused.hip
template <int D_, bool RT = false>
__device__ __forceinline__ void load_rows64(d4 (&a)[tiles(D_)], const void *xin, int in_f64, int64_t row, bool valid, int lg,
                                            const double *__restrict__ feats, int dr = D_) {
    const int dw = RT ? dr : D_;
    const int64_t rbase = (valid ? row : 0) * dw;
#pragma unroll
    for (int t = 0; t < tiles(D_); ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = creg_feature(D_, t, lg, r);
            const bool live = f >= 0 && (!RT || f < dr);
            const int fc = live ? f : 0;                          // padding slots read feature 0 (finite, meets zero weights)
            double v = in_f64 ? ((const double *)xin)[rbase + fc] : (double)((const float *)xin)[rbase + fc];
            if (feats) v = (v - feats[fc]) / feats[dw + fc];
            a[t][r] = live ? v : 0.0;
        }
}
template <int D_, bool RT = false>
__device__ __forceinline__ void store_rows64(const d4 (&a)[tiles(D_)], void *out, int out_f64, int64_t row, bool valid, int lg,
                                             const double *__restrict__ renorm, const uint8_t *__restrict__ imask, int dr = D_) {
    if (!valid) return;
    const int dw = RT ? dr : D_;
#pragma unroll
    for (int t = 0; t < tiles(D_); ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = creg_feature(D_, t, lg, r);
            if (f < 0 || (RT && f >= dr)) continue;
            double v = a[t][r];
            if (renorm) {      // norm * range + min with two roundings, then the int-column truncation (elementwise.hip renormalize_k)
                v = __dadd_rn(__dmul_rn(v, renorm[dw + f]), renorm[f]);
                if (imask && imask[f]) v = trunc(v);
            }
            if (out_f64) ((double *)out)[row * dw + f] = v;
            else ((float *)out)[row * dw + f] = (float)v;
        }
}
template <int F, int Z, int KIND, bool RT = false>
__global__ void __launch_bounds__(256) infer64_kernel(const d4 *packed, const void *__restrict__ xin, int in_f64, int64_t n,
                                                      const double *__restrict__ feats, void *__restrict__ out, int out_f64,
                                                      const double *__restrict__ renorm, const uint8_t *__restrict__ imask,
                                                      double *__restrict__ loss_part, int fr, int zr) {
    using N = Net64<F, Z>;
    constexpr int G0 = KIND == I_DECODE ? 4 : 0, NGM = KIND == I_FORWARD ? 8 : 4;
    using SQ = ISeq<N, G0, NGM, 8>;
    constexpr int kNB = N::bf_off(N::L) - N::bf_off(0);
    extern __shared__ __attribute__((aligned(32))) unsigned char lds_raw[];
    d4 *bias_lds = (d4 *)lds_raw;
    __shared__ double red[256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lg = lane >> 4;
    for (int i = threadIdx.x; i < kNB; i += 256) bias_lds[i] = packed[N::bf_off(0) + i];
    WStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)packed, 0, N::packed_d4() * 32, 0x00020000);
    ws.voff = lane * 32;
    d4 ring[SQ::D];
    iseq_span<SQ>(ring, ws, std::make_integer_sequence<int, SQ::D>{}, 0);
    __syncthreads();
    double lacc = 0.0;
    const int64_t ntile = (n + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntile; tile += (int64_t)gridDim.x * 4) {
        asm volatile("" : "+v"(ws.voff));      // keep the fragment loads inside the loop (LICM would hoist the whole model)
        const int64_t row = tile * 16 + (lane & 15);
        const bool valid = row < n;
        if constexpr (KIND == I_DECODE) {
            d4 a4[tiles(Z)], s5[4], s6[7], s7[13], o8[tiles(F)];
            load_rows64<Z, RT>(a4, xin, in_f64, row, valid, lg, feats, zr);
            ilayer<N, SQ, 0, G0>(a4, s5, ring, ws, bias_lds, lg);
            ilayer<N, SQ, 1, G0>(s5, s6, ring, ws, bias_lds, lg);
            ilayer<N, SQ, 2, G0>(s6, s7, ring, ws, bias_lds, lg);
            ilayer<N, SQ, 3, G0>(s7, o8, ring, ws, bias_lds, lg);
            store_rows64<F, RT>(o8, out, out_f64, row, valid, lg, renorm, imask, fr);
        } else {
            d4 a0[tiles(F)], s1[13], s2[7], s3[4], s4[tiles(Z)];
            load_rows64<F, RT>(a0, xin, in_f64, row, valid, lg, feats, fr);
            ilayer<N, SQ, 0, G0>(a0, s1, ring, ws, bias_lds, lg);
            ilayer<N, SQ, 1, G0>(s1, s2, ring, ws, bias_lds, lg);
            ilayer<N, SQ, 2, G0>(s2, s3, ring, ws, bias_lds, lg);
            ilayer<N, SQ, 3, G0>(s3, s4, ring, ws, bias_lds, lg);
            if constexpr (KIND == I_ENCODE) {
                store_rows64<Z, RT>(s4, out, out_f64, row, valid, lg, nullptr, nullptr, zr);
            } else {
                d4 s5[4], s6[7], s7[13], o8[tiles(F)];
                ilayer<N, SQ, 4, G0>(s4, s5, ring, ws, bias_lds, lg);
                ilayer<N, SQ, 5, G0>(s5, s6, ring, ws, bias_lds, lg);
                ilayer<N, SQ, 6, G0>(s6, s7, ring, ws, bias_lds, lg);
                ilayer<N, SQ, 7, G0>(s7, o8, ring, ws, bias_lds, lg);
#pragma unroll
                for (int t = 0; t < tiles(F); ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double d = o8[t][r] - a0[t][r];
                        if (valid && creg_feature(F, t, lg, r) >= 0 && (!RT || creg_feature(F, t, lg, r) < fr)) lacc += d * d;
                    }
                if (out) store_rows64<F, RT>(o8, out, out_f64, row, valid, lg, nullptr, nullptr, fr);
            }
        }
        iseq_tail<SQ, SQ::real>(ring, ws, std::make_integer_sequence<int, SQ::total - SQ::real>{});      // step over the padding
    }
    if constexpr (KIND == I_FORWARD) {      // per-workgroup loss partial, fixed order
        red[threadIdx.x] = lacc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) loss_part[blockIdx.x] = red[0];
    }
}
__global__ void __launch_bounds__(256) sum_loss64_k(const double *__restrict__ part, int n, double scale, double *__restrict__ out) {
    __shared__ double sh[256];
    const double s = block_sum_fixed(part, n, sh);
    if (threadIdx.x == 0) *out = s * scale;
}

// The launch of one shape (what Impl64::infer was): persistent grid, two workgroups per CU, the loss partials summed by a second launch
template <int F, int Z, bool RT>
int infer64_run(bamd_handle *h, const double *packed, int kind, const void *x, int x_dtype, int64_t n, const double *features, void *out,
                int out_dtype, const double *renorm, const uint8_t *imask, double *loss_sum, hipStream_t s) {
    using N = Net64<F, Z>;
    const int fr = h->dims[0], zr = h->dims[4];
    const int64_t ngroup = (n + 63) / 64;
    static const int cap = getenv("BALER_AMD_F64_INFER_WGS") ? atoi(getenv("BALER_AMD_F64_INFER_WGS")) : 512;
    const int grid = (int)(ngroup < cap ? ngroup : cap);          // <= 256 registers: two workgroups per CU (two waves per SIMD), persistent
    constexpr int lds = (N::bf_off(N::L) - N::bf_off(0)) * 32;
    const int in64 = x_dtype == BAMD_F64, out64 = out_dtype == BAMD_F64;
    if (kind == I_FORWARD) {
        int rc = h->lossp.ensure(sizeof(double) * (size_t)(grid > 1024 ? grid : 1024));     // one partial per workgroup (BALER_AMD_F64_INFER_WGS may exceed 1024)
        if (rc) return rc;
    }
    if (kind == I_ENCODE)
        hipLaunchKernelGGL((infer64_kernel<F, Z, I_ENCODE, RT>), dim3(grid), dim3(256), lds, s, (const d4 *)packed, x, in64, n, features,
                           out, out64, renorm, imask, (double *)nullptr, fr, zr);
    else if (kind == I_DECODE)
        hipLaunchKernelGGL((infer64_kernel<F, Z, I_DECODE, RT>), dim3(grid), dim3(256), lds, s, (const d4 *)packed, x, in64, n, features,
                           out, out64, renorm, imask, (double *)nullptr, fr, zr);
    else {
        hipLaunchKernelGGL((infer64_kernel<F, Z, I_FORWARD, RT>), dim3(grid), dim3(256), lds, s, (const d4 *)packed, x, in64, n, features,
                           out, out64, renorm, imask, (double *)h->lossp.p, fr, zr);
        hipLaunchKernelGGL(sum_loss64_k, dim3(1), dim3(256), 0, s, (const double *)h->lossp.p, grid, 1.0 / fr, loss_sum);
    }
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

}  // namespace
}  // namespace bamd
