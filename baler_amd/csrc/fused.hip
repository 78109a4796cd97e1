// Fused register-chained kernels for the narrow dense autoencoder (CMS shape AE(24,15)), fp32 MFMA.
//
// Formulation (gfx950, v_mfma_f32_16x16x4_f32 = exact fp32):  every Linear layer is computed
// TRANSPOSED, Y^T[N x M] = W[N x K] . X^T[K x M], with the weights as the MFMA A operand and the batch
// rows on the MFMA column (lane & 15).  The accumulator (C/D) layout of a 16x16 tile is
//     column = lane & 15 (batch row m),  row = 4*(lane >> 4) + reg   (feature slot)
// and the B-operand layout of the NEXT layer's product is  B[k = lane >> 4][col = lane & 15]  per MFMA
// step, so register `reg` of an output tile IS the B operand of step `reg` of the next layer, with the
// k order permuted (k = 4g + reg on lane group g).  The weights are pre-packed ("fragment order") with
// the same permutation, so a whole wave pushes its 16 batch rows through all layers with activations
// living only in registers: no LDS, no cross-lane traffic, no barriers on the forward/backward chain.
// HBM sees x in and z / recon / gradients out.  Partial last tiles use an r-major slot order so the
// reduction runs over ceil(K/4) MFMA steps (K granularity 4, N granularity 16): 86 % of the issued MACs
// are algorithmic for AE(24,15).
//
// Training adds, per layer, the weight-gradient product  [dW | db] = dZ^T [X | 1]  reduced over the
// workgroup's 64 rows: the four waves write their dZ^T / X^T tiles into a [slot][row] LDS image, and the
// (n-tile, k-tile) outputs are dealt round-robin to the waves, accumulated into a PRIVATE per-workgroup
// slab in fragment order (plain 16-byte loads/stores, no atomics).  A second kernel sums the slabs in a
// fixed order into the canonical state-dict layout: bitwise reproducible.
#include "fused.hpp"

#include <cstdlib>
#include <type_traits>

namespace bamd {
namespace {

using v4 = float __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int tiles(int d) { return (d + 15) / 16; }
// MFMA steps needed for feature tile t of a dimension d (4 for full tiles, ceil(valid/4) for the last)
__host__ __device__ constexpr int tile_steps(int d, int t) {
    return d - 16 * t >= 16 ? 4 : (d - 16 * t + 3) / 4;
}
// feature held by slot (tile t, lane group g, register r); -1 = padding.  Full tiles: 16t + 4g + r
// (4 consecutive features per lane: vector I/O); partial last tile: 16t + 4r + g (fills registers first).
__host__ __device__ constexpr int slot_feature(int d, int t, int g, int r) {
    int v = d - 16 * t;
    if (v >= 16) return 16 * t + 4 * g + r;
    return 4 * r + g < v ? 16 * t + 4 * r + g : -1;
}

constexpr int kQS = 68;        // LDS row stride (floats) of the [slot][row] images: 64 rows + 4 pad
constexpr int kQRows = 208;    // 13 tiles
constexpr int kRowsPerWG = 64;

// ---- compile-time description of AE(F, Z): 8 layers F-200-100-50-Z-50-100-200-F -----------------------
template <int F, int Z> struct Net {
    static constexpr int L = 8;
    __host__ __device__ static constexpr int dim(int i) {
        return i == 0 ? F : i == 1 ? 200 : i == 2 ? 100 : i == 3 ? 50 : i == 4 ? Z : i == 5 ? 50 : i == 6 ? 100 : i == 7 ? 200 : F;
    }
    __host__ __device__ static constexpr bool act(int l) { return !(l == 3 || l == 7); }
    // packed buffer (float4 units): [Wf of all layers | Wb of all layers | bias frags of all layers]
    __host__ __device__ static constexpr int wcount(int l) { return tiles(dim(l)) * tiles(dim(l + 1)) * 64; }
    __host__ __device__ static constexpr int wf_off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += wcount(j); return s; }
    // transposed (backward) fragments, packed in consumption order: layer 7, 6, ..., 1 (layer 0 last, unused)
    __host__ __device__ static constexpr int wb_off(int l) { int s = wf_off(L); for (int j = L - 1; j > l; --j) s += wcount(j); return s; }
    __host__ __device__ static constexpr int bf_off(int l) { int s = 2 * wf_off(L); for (int j = 0; j < l; ++j) s += tiles(dim(j + 1)) * 4; return s; }
    __host__ __device__ static constexpr int packed_f4() { return bf_off(L); }
    // weight-gradient tiles of layer l: tiles(N) x tiles(K + 1) (the extra slot carries db)
    __host__ __device__ static constexpr int dw_tiles(int l) { return tiles(dim(l + 1)) * tiles(dim(l) + 1); }
    __host__ __device__ static constexpr int slab_off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += dw_tiles(j); return s; }
    __host__ __device__ static constexpr int slab_f4() { return (slab_off(L) + 1) * 64 + 4; }  // + dummy tile + loss slot
    // canonical (state-dict) offsets
    __host__ __device__ static constexpr int w_off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += dim(j + 1) * dim(j) + dim(j + 1); return s; }
    __host__ __device__ static constexpr int b_off(int l) { return w_off(l) + dim(l + 1) * dim(l); }
    __host__ __device__ static constexpr int nparams() { return w_off(L); }
};

__device__ __forceinline__ v4 mfma(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// Weight fragments are consumed as ONE linear stream per kernel iteration (forward layers 0..7, then
// the transposed fragments of layers 7..1 for the backward chain): fragment G of the stream is the
// 1-KiB block stream[G*64 + lane].  A ring of P fragments is kept in flight: right after fragment G
// has fed its MFMAs, its ring slot is refilled with fragment G + P (wrapping to the next iteration), so
// every L2 access has P*4 MFMAs (~2 us) to land.  sched_barrier(0) pins that order; hipcc inserts the
// counted s_waitcnt vmcnt(P-1) itself.
typedef const v4 __attribute__((address_space(1))) *gv4p;
constexpr int kRing = 8;
struct Ring { v4 slot[kRing]; };

// The stream is walked cyclically (iteration after iteration), so its length is padded to a multiple of
// the ring size: fragment G always lives in slot G % kRing.  The pad fragments are loaded (they alias
// whatever follows the stream in the packed buffer) but never multiplied.
__host__ __device__ constexpr int pad_total(int t) { return (t + kRing - 1) / kRing * kRing; }

template <int TOTAL>
__device__ __forceinline__ void ring_prime(Ring &ring, gv4p stream, int lane) {
    static_assert(TOTAL >= kRing, "stream shorter than the ring");
#pragma unroll
    for (int i = 0; i < kRing; ++i) ring.slot[i] = stream[i * 64 + lane];
}
// end of an iteration: step over the pad fragments, refilling their slots for the next iteration
template <int TOTAL>
__device__ __forceinline__ void ring_tail(Ring &ring, gv4p stream, int lane) {
#pragma unroll
    for (int G = TOTAL; G < pad_total(TOTAL); ++G)
        ring.slot[G % kRing] = stream[((G + kRing) % pad_total(TOTAL)) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
}

// out^T tiles += frags . in^T tiles;  KD = reduction dimension (size of `in`); this layer's fragments are
// stream fragments BASE .. BASE + tiles(KD)*NT - 1 in [q][t] order (component r = step r of k-tile q).
template <int KD, int NT, int BASE, int TOTAL>
__device__ __forceinline__ void chain_gemm(const v4 (&in)[tiles(KD)], v4 (&out)[NT], Ring &ring, gv4p stream, int lane) {
#pragma unroll
    for (int q = 0; q < tiles(KD); ++q) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            constexpr int dummy = 0;
            const int G = BASE + q * NT + t;
            const int sl = G % kRing;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < tile_steps(KD, q)) out[t] = mfma(ring.slot[sl][r], in[q][r], out[t]);
            ring.slot[sl] = stream[((G + kRing) % pad_total(TOTAL)) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
            (void)dummy;
        }
    }
}

template <int NT> __device__ __forceinline__ void init_bias(v4 (&out)[NT], const v4 *bias_lds, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t] = bias_lds[t * 4 + (lane >> 4)];
}
template <int NT> __device__ __forceinline__ void zero_tiles(v4 (&out)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t] = (v4){0.f, 0.f, 0.f, 0.f};
}
template <int NT> __device__ __forceinline__ void lrelu(v4 (&a)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[t][r] = a[t][r] > 0.f ? a[t][r] : a[t][r] * 0.01f;
}
// dZ = dY * lrelu'(pre) ; sign(pre) == sign(post-activation y)
template <int NT> __device__ __forceinline__ void lrelu_bwd(v4 (&d)[NT], const v4 (&y)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[t][r] = y[t][r] > 0.f ? d[t][r] : d[t][r] * 0.01f;
}

// one Linear layer of the forward chain.  S = stream description (frag base of every layer, total)
template <class N, class S, int l>
__device__ __forceinline__ void fwd_layer(const v4 (&in)[tiles(N::dim(l))], v4 (&out)[tiles(N::dim(l + 1))],
                                          Ring &ring, gv4p stream, const v4 *bias_lds, int lane) {
    init_bias(out, bias_lds + (N::bf_off(l) - N::bf_off(0)), lane);
    chain_gemm<N::dim(l), tiles(N::dim(l + 1)), S::fwd_base(l), S::total>(in, out, ring, stream, lane);
    if (N::act(l)) lrelu(out);
}
// dY_{l-1}^T = W_l^T dZ_l^T
template <class N, class S, int l>
__device__ __forceinline__ void bwd_layer(const v4 (&dz)[tiles(N::dim(l + 1))], v4 (&dx)[tiles(N::dim(l))],
                                          Ring &ring, gv4p stream, int lane) {
    zero_tiles(dx);
    chain_gemm<N::dim(l + 1), tiles(N::dim(l)), S::bwd_base(l), S::total>(dz, dx, ring, stream, lane);
}

// stream descriptions: which layers' fragments a kernel iteration walks, in order
template <class N> struct StreamEncode {   // forward fragments of layers 0..3
    static constexpr int fwd_base(int l) { return N::wf_off(l) / 64; }
    static constexpr int total = N::wf_off(4) / 64;
    static constexpr int start_f4 = 0;
};
template <class N> struct StreamDecode {   // forward fragments of layers 4..7
    static constexpr int fwd_base(int l) { return (N::wf_off(l) - N::wf_off(4)) / 64; }
    static constexpr int total = (N::wf_off(8) - N::wf_off(4)) / 64;
    static constexpr int start_f4 = N::wf_off(4);
};
template <class N> struct StreamForward {  // forward fragments of layers 0..7
    static constexpr int fwd_base(int l) { return N::wf_off(l) / 64; }
    static constexpr int total = N::wf_off(8) / 64;
    static constexpr int start_f4 = 0;
};
template <class N> struct StreamTrain {    // forward 0..7 then backward 7..1 (packed in that order)
    static constexpr int fwd_base(int l) { return N::wf_off(l) / 64; }
    static constexpr int bwd_base(int l) { return (N::wb_off(l)) / 64; }
    static constexpr int total = N::wb_off(0) / 64;   // layer 0 needs no input gradient
    static constexpr int start_f4 = 0;
};

// ---- row I/O in slot order ------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void load_rows(v4 (&a)[tiles(D)], const void *x, int is_f64, int64_t row, bool valid,
                                          int lane, const double *__restrict__ feats) {
    const int g = lane >> 4;
#pragma unroll
    for (int t = 0; t < tiles(D); ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = slot_feature(D, t, g, r);
            float v = 0.f;
            if (valid && f >= 0) {
                const int64_t i = row * D + f;
                if (feats) {
                    double d = is_f64 ? ((const double *)x)[i] : (double)((const float *)x)[i];
                    v = (float)((d - feats[f]) / feats[D + f]);   // (x - min)/(max - min) in float64
                } else {
                    v = is_f64 ? (float)((const double *)x)[i] : ((const float *)x)[i];
                }
            }
            a[t][r] = v;
        }
}

template <int D>
__device__ __forceinline__ void store_rows(const v4 (&a)[tiles(D)], void *out, int is_f64, int64_t row, bool valid,
                                           int lane, const double *__restrict__ renorm, const uint8_t *__restrict__ imask) {
    const int g = lane >> 4;
#pragma unroll
    for (int t = 0; t < tiles(D); ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = slot_feature(D, t, g, r);
            if (valid && f >= 0) {
                const int64_t i = row * D + f;
                if (renorm) {
                    // norm*range + min with two roundings (numpy), then trunc for "int" columns (baler.py:420-435)
                    double d = __dadd_rn(__dmul_rn((double)a[t][r], renorm[D + f]), renorm[f]);
                    if (imask && imask[f]) d = trunc(d);
                    if (is_f64) ((double *)out)[i] = d; else ((float *)out)[i] = (float)d;
                } else {
                    if (is_f64) ((double *)out)[i] = (double)a[t][r]; else ((float *)out)[i] = a[t][r];
                }
            }
        }
}

// ---- inference kernels: every wave streams 16-row tiles on its own ---------------------------------------
enum { K_ENCODE = 0, K_DECODE = 1, K_FORWARD = 2 };

template <class N>
__device__ __forceinline__ void stage_bias(v4 *bias_lds, const v4 *packed) {
    constexpr int nb = N::bf_off(N::L) - N::bf_off(0);
    for (int i = threadIdx.x; i < nb; i += blockDim.x) bias_lds[i] = packed[N::bf_off(0) + i];
    __syncthreads();
}

template <int F, int Z, int KIND>
__global__ void __launch_bounds__(256) infer_kernel(const v4 *packed, const void *__restrict__ xin, int in_f64,
                                                    int64_t n, const double *__restrict__ feats, void *__restrict__ out,
                                                    int out_f64, const uint8_t *__restrict__ imask,
                                                    double *__restrict__ loss_part) {
    using N = Net<F, Z>;
    using S = typename std::conditional<KIND == K_ENCODE, StreamEncode<N>,
                                        typename std::conditional<KIND == K_DECODE, StreamDecode<N>, StreamForward<N>>::type>::type;
    __shared__ __attribute__((aligned(16))) v4 bias_lds[N::bf_off(N::L) - N::bf_off(0)];
    __shared__ double sh[256];
    stage_bias<N>(bias_lds, packed);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ntile = (n + 15) / 16;
    gv4p stream = (gv4p)packed + S::start_f4;
    double lacc = 0.0;
    Ring ring;
    ring_prime<S::total>(ring, stream, lane);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntile; tile += (int64_t)gridDim.x * 4) {
        const int64_t row = tile * 16 + (lane & 15);
        const bool valid = row < n;
        // keep the weight loads INSIDE the loop: without this LICM hoists all of them (loop-invariant
        // addresses) and spills the whole model to scratch
        asm volatile("" : "+s"(stream));
        if (KIND == K_ENCODE || KIND == K_FORWARD) {
            v4 a0[tiles(F)], a1[13], a2[7], a3[4], a4[tiles(Z)];
            load_rows<F>(a0, xin, in_f64, row, valid, lane, feats);
            fwd_layer<N, S, 0>(a0, a1, ring, stream, bias_lds, lane);
            fwd_layer<N, S, 1>(a1, a2, ring, stream, bias_lds, lane);
            fwd_layer<N, S, 2>(a2, a3, ring, stream, bias_lds, lane);
            fwd_layer<N, S, 3>(a3, a4, ring, stream, bias_lds, lane);
            if (KIND == K_ENCODE) {
                store_rows<Z>(a4, out, out_f64, row, valid, lane, nullptr, nullptr);
            } else {
                v4 a5[4], a6[7], a7[13], a8[tiles(F)];
                fwd_layer<N, S, 4>(a4, a5, ring, stream, bias_lds, lane);
                fwd_layer<N, S, 5>(a5, a6, ring, stream, bias_lds, lane);
                fwd_layer<N, S, 6>(a6, a7, ring, stream, bias_lds, lane);
                fwd_layer<N, S, 7>(a7, a8, ring, stream, bias_lds, lane);
                if (out) store_rows<F>(a8, out, out_f64, row, valid, lane, nullptr, nullptr);
                if (valid) {
#pragma unroll
                    for (int t = 0; t < tiles(F); ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (slot_feature(F, t, lane >> 4, r) >= 0) {
                                double d = (double)a8[t][r] - (double)a0[t][r];
                                lacc += d * d;
                            }
                }
            }
        } else {
            v4 a4[tiles(Z)], a5[4], a6[7], a7[13], a8[tiles(F)];
            load_rows<Z>(a4, xin, in_f64, row, valid, lane, nullptr);
            fwd_layer<N, S, 4>(a4, a5, ring, stream, bias_lds, lane);
            fwd_layer<N, S, 5>(a5, a6, ring, stream, bias_lds, lane);
            fwd_layer<N, S, 6>(a6, a7, ring, stream, bias_lds, lane);
            fwd_layer<N, S, 7>(a7, a8, ring, stream, bias_lds, lane);
            store_rows<F>(a8, out, out_f64, row, valid, lane, feats, imask);
        }
        ring_tail<S::total>(ring, stream, lane);
    }
    if (KIND == K_FORWARD) {
        sh[threadIdx.x] = lacc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) loss_part[blockIdx.x] = sh[0];
    }
}

// ---- training kernel ----------------------------------------------------------------------------------
// LDS image helpers: rows = feature slots (16t + 4g + r), columns = the workgroup's 64 batch rows.
template <int NT>
__device__ __forceinline__ void q_write(float *__restrict__ q, const v4 (&a)[NT], int lane, int wave) {
    const int col = 16 * wave + (lane & 15), g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) q[(16 * t + 4 * g + r) * kQS + col] = a[t][r];
}
// X^T image with the ones row in the first padding slot of dimension D (carries db through the GEMM)
template <int D>
__device__ __forceinline__ void q_write_x(float *__restrict__ q, const v4 (&a)[tiles(D)], int lane, int wave) {
    static_assert(D % 16 != 0, "ones slot lives in the partial last tile");
    constexpr int T = tiles(D) - 1, V = D - 16 * T;      // partial tile, V valid slots; ones slot idx = V
    constexpr int R1 = V / 4, G1 = V % 4;
    const int col = 16 * wave + (lane & 15), g = lane >> 4;
#pragma unroll
    for (int t = 0; t < tiles(D); ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = a[t][r];
            if (t == T && r == R1 && g == G1) v = 1.0f;
            q[(16 * t + 4 * g + r) * kQS + col] = v;
        }
}

// [dW | db] tiles of layer l: D[n-slot][k-slot] = sum over the workgroup's 64 rows of dZ^T[n][m] X^T[k][m].
// Tiles are dealt round-robin to the 4 waves; every wave runs the same static schedule (the surplus
// tile of a short wave recomputes the last tile into a dummy slab tile), LDS fragment reads of tile i+1
// are issued before the MFMAs of tile i, and the old slab tile is added AFTER the MFMA chain so its load
// latency hides behind it.
template <class N, int l>
__device__ __forceinline__ void dw_phase(const float *__restrict__ qdz, const float *__restrict__ qx, v4 *__restrict__ slab,
                                         bool accumulate, int lane, int wave) {
    constexpr int NT = tiles(N::dim(l + 1)), KT = tiles(N::dim(l) + 1), TOT = NT * KT, T = (TOT + 3) / 4;
    constexpr int U = 4 * T, DEPTH = 3;            // (tile, 16-row group) steps; LDS fragment lookahead
    const int g = lane >> 4, i = lane & 15;
    v4 fa[DEPTH], fb[DEPTH];
    auto lds_frags = [&](int u, v4 &a, v4 &b) {
        int idx = wave + 4 * (u >> 2);
        idx = idx < TOT ? idx : TOT - 1;
        const int kt = idx / NT, nt = idx - kt * NT;
        a = *(const v4 *)(qdz + (16 * nt + i) * kQS + 4 * g + 16 * (u & 3));
        b = *(const v4 *)(qx + (16 * kt + i) * kQS + 4 * g + 16 * (u & 3));
    };
#pragma unroll
    for (int u = 0; u < DEPTH - 1; ++u) lds_frags(u, fa[u], fb[u]);
    v4 acc, old;
    v4 *dst = nullptr;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if ((u & 3) == 0) {
            const int idx = wave + 4 * (u >> 2);
            // dead surplus tile -> dummy tile at the end of the slab (index slab_off(L))
            dst = slab + (idx < TOT ? N::slab_off(l) + idx : N::slab_off(N::L)) * 64 + lane;
            old = *dst;
            acc = (v4){0.f, 0.f, 0.f, 0.f};
        }
        if (u + DEPTH - 1 < U) lds_frags(u + DEPTH - 1, fa[(u + DEPTH - 1) % DEPTH], fb[(u + DEPTH - 1) % DEPTH]);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma(fa[u % DEPTH][r], fb[u % DEPTH][r], acc);
        if ((u & 3) == 3) {
            if (accumulate) acc += old;
            *dst = acc;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <class N, int l>
__device__ __forceinline__ void layer_grads(const float *qdz, const float *qx, v4 *slab, bool accumulate, int lane, int wave) {
    // (callers run the dX chain between the image writes and this function's first barrier)
    __syncthreads();
    dw_phase<N, l>(qdz, qx, slab, accumulate, lane, wave);
    __syncthreads();
}

template <int F, int Z>
__global__ void __launch_bounds__(256) train_kernel(const v4 *packed, const void *__restrict__ xin, int in_f64,
                                                    int64_t n, const double *__restrict__ feats, v4 *__restrict__ slabs) {
    using N = Net<F, Z>;
    using S = StreamTrain<N>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *qx = lds, *qdz = lds + kQRows * kQS;
    v4 *bias_lds = (v4 *)(lds + 2 * kQRows * kQS);
    stage_bias<N>(bias_lds, packed);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4 *slab = slabs + (int64_t)blockIdx.x * N::slab_f4();
    const int64_t ngroups = (n + kRowsPerWG - 1) / kRowsPerWG;
    gv4p stream = (gv4p)packed;
    double lacc = 0.0;
    bool accumulate = false;
    Ring ring;
    ring_prime<S::total>(ring, stream, lane);
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t row = grp * kRowsPerWG + 16 * wave + (lane & 15);
        const bool valid = row < n;
        asm volatile("" : "+s"(stream));  // keep the weight loads inside the loop (see infer_kernel)
        // ---- forward, keeping every layer input (the stash) in registers
        v4 a0[tiles(F)], a1[13], a2[7], a3[4], a4[tiles(Z)], a5[4], a6[7], a7[13];
        v4 d8[tiles(F)];
        load_rows<F>(a0, xin, in_f64, row, valid, lane, feats);
        fwd_layer<N, S, 0>(a0, a1, ring, stream, bias_lds, lane);
        fwd_layer<N, S, 1>(a1, a2, ring, stream, bias_lds, lane);
        fwd_layer<N, S, 2>(a2, a3, ring, stream, bias_lds, lane);
        fwd_layer<N, S, 3>(a3, a4, ring, stream, bias_lds, lane);
        fwd_layer<N, S, 4>(a4, a5, ring, stream, bias_lds, lane);
        fwd_layer<N, S, 5>(a5, a6, ring, stream, bias_lds, lane);
        fwd_layer<N, S, 6>(a6, a7, ring, stream, bias_lds, lane);
        fwd_layer<N, S, 7>(a7, d8, ring, stream, bias_lds, lane);
        // ---- loss and dL/drecon = 2 (r - x)/C  (utils.py:195-199); invalid rows contribute nothing
#pragma unroll
        for (int t = 0; t < tiles(F); ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float d = d8[t][r] - a0[t][r];
                const bool live = valid && slot_feature(F, t, lane >> 4, r) >= 0;
                if (live) lacc += (double)d * (double)d;
                d8[t][r] = live ? d * (2.0f / (float)F) : 0.f;
            }
        // ---- backward: layer 8 .. 1.  Image writes, then the dX chain (registers only), then the
        //      workgroup-wide weight-gradient tiles between two barriers.
        v4 d7[13], d6[7], d5[4], d4[tiles(Z)], d3[4], d2[7], d1[13];
        q_write(qdz, d8, lane, wave); q_write_x<200>(qx, a7, lane, wave);
        bwd_layer<N, S, 7>(d8, d7, ring, stream, lane); lrelu_bwd(d7, a7);
        layer_grads<N, 7>(qdz, qx, slab, accumulate, lane, wave);

        q_write(qdz, d7, lane, wave); q_write_x<100>(qx, a6, lane, wave);
        bwd_layer<N, S, 6>(d7, d6, ring, stream, lane); lrelu_bwd(d6, a6);
        layer_grads<N, 6>(qdz, qx, slab, accumulate, lane, wave);

        q_write(qdz, d6, lane, wave); q_write_x<50>(qx, a5, lane, wave);
        bwd_layer<N, S, 5>(d6, d5, ring, stream, lane); lrelu_bwd(d5, a5);
        layer_grads<N, 5>(qdz, qx, slab, accumulate, lane, wave);

        q_write(qdz, d5, lane, wave); q_write_x<Z>(qx, a4, lane, wave);
        bwd_layer<N, S, 4>(d5, d4, ring, stream, lane);            // en4 has no activation
        layer_grads<N, 4>(qdz, qx, slab, accumulate, lane, wave);

        q_write(qdz, d4, lane, wave); q_write_x<50>(qx, a3, lane, wave);
        bwd_layer<N, S, 3>(d4, d3, ring, stream, lane); lrelu_bwd(d3, a3);
        layer_grads<N, 3>(qdz, qx, slab, accumulate, lane, wave);

        q_write(qdz, d3, lane, wave); q_write_x<100>(qx, a2, lane, wave);
        bwd_layer<N, S, 2>(d3, d2, ring, stream, lane); lrelu_bwd(d2, a2);
        layer_grads<N, 2>(qdz, qx, slab, accumulate, lane, wave);

        q_write(qdz, d2, lane, wave); q_write_x<200>(qx, a1, lane, wave);
        bwd_layer<N, S, 1>(d2, d1, ring, stream, lane); lrelu_bwd(d1, a1);
        layer_grads<N, 1>(qdz, qx, slab, accumulate, lane, wave);

        q_write(qdz, d1, lane, wave); q_write_x<F>(qx, a0, lane, wave);
        layer_grads<N, 0>(qdz, qx, slab, accumulate, lane, wave);
        ring_tail<S::total>(ring, stream, lane);
        accumulate = true;
    }
    // per-workgroup loss partial (fixed-order tree)
    __syncthreads();
    double *sh = (double *)lds;
    sh[threadIdx.x] = lacc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) *(double *)(slab + (N::slab_off(N::L) + 1) * 64) = sh[0];
}

// grads[p] = sum over workgroup slabs (fixed order) of slab[map[p]];  grads[np] = sum of loss partials / C
template <typename T>
__global__ void __launch_bounds__(256) reduce_slabs_map_k(const float *__restrict__ slabs, int nslab, int64_t slab_floats,
                                                          const int *__restrict__ map, int np, int loss_float_off,
                                                          double inv_c, T *__restrict__ grads) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < np) {
        const int src = map[p];
        float s = 0.f;
        for (int k = 0; k < nslab; ++k) s += slabs[(int64_t)k * slab_floats + src];
        grads[p] = (T)s;
    } else if (p == np) {
        double s = 0.0;
        for (int k = 0; k < nslab; ++k) s += *(const double *)(slabs + (int64_t)k * slab_floats + loss_float_off);
        grads[np] = (T)(s * inv_c);
    }
}

__global__ void __launch_bounds__(256) pack_k(const float *__restrict__ params, const int *__restrict__ src, int count,
                                              float *__restrict__ packed) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        int s = src[i];
        packed[i] = s >= 0 ? params[s] : 0.f;
    }
}

__global__ void sum_loss_k(const double *__restrict__ part, int n, double scale, double *__restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += part[i];
        *out = s * scale;
    }
}

// ---- host side ------------------------------------------------------------------------------------------
struct FusedState {
    DevBuf pack_src;   // int per packed float: canonical parameter index or -1
    DevBuf slab_map;   // int per canonical parameter: float offset inside a slab
    int packed_floats = 0;
    int nwg_max = 256;
};

template <int F, int Z>
static int build_maps(bamd_handle *h, FusedState *st) {
    using N = Net<F, Z>;
    std::vector<int> src((size_t)N::packed_f4() * 4, -1);
    std::vector<int> smap((size_t)N::nparams(), -1);
    for (int l = 0; l < N::L; ++l) {
        const int K = N::dim(l), NN = N::dim(l + 1), KT = tiles(K), NT = tiles(NN);
        // forward frags: [q][t][lane].comp[r] = W[n(t, i = lane & 15)][k(q, g = lane >> 4, r)]
        for (int q = 0; q < KT; ++q)
            for (int t = 0; t < NT; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        int i = lane & 15, g = lane >> 4;
                        int nf = slot_feature(NN, t, i / 4, i % 4), kf = slot_feature(K, q, g, r);
                        if (nf >= 0 && kf >= 0)
                            src[((size_t)N::wf_off(l) + (q * NT + t) * 64 + lane) * 4 + r] = N::w_off(l) + nf * K + kf;
                    }
        // backward frags: [tq][tk][lane].comp[r] = W[n(tq, g, r)][k(tk, i)]
        for (int tq = 0; tq < NT; ++tq)
            for (int tk = 0; tk < KT; ++tk)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        int i = lane & 15, g = lane >> 4;
                        int nf = slot_feature(NN, tq, g, r), kf = slot_feature(K, tk, i / 4, i % 4);
                        if (nf >= 0 && kf >= 0)
                            src[((size_t)N::wb_off(l) + (tq * KT + tk) * 64 + lane) * 4 + r] = N::w_off(l) + nf * K + kf;
                    }
        // bias frags: [t][g].comp[r] = b[n(t, g, r)]
        for (int t = 0; t < NT; ++t)
            for (int g = 0; g < 4; ++g)
                for (int r = 0; r < 4; ++r) {
                    int nf = slot_feature(NN, t, g, r);
                    if (nf >= 0) src[((size_t)N::bf_off(l) + t * 4 + g) * 4 + r] = N::b_off(l) + nf;
                }
        // slab map: tile idx = kt*NT + nt; lane (j = lane & 15 -> k slot row 16kt + j), reg r -> n slot row 4g + r
        const int KTp = tiles(K + 1);
        const int T1 = tiles(K) - 1, V = K - 16 * T1, ones_row = 16 * T1 + 4 * (V % 4) + (V / 4);
        for (int kt = 0; kt < KTp; ++kt)
            for (int nt = 0; nt < NT; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        int j = lane & 15, g = lane >> 4;
                        int nf = slot_feature(NN, nt, g, r);
                        if (nf < 0) continue;
                        int off = ((N::slab_off(l) + kt * NT + nt) * 64 + lane) * 4 + r;
                        int krow = 16 * kt + j;
                        if (krow == ones_row) smap[N::b_off(l) + nf] = off;
                        else if (kt < tiles(K)) {
                            int kf = slot_feature(K, kt, j / 4, j % 4);
                            if (kf >= 0) smap[N::w_off(l) + nf * K + kf] = off;
                        }
                    }
    }
    for (int v : smap)
        if (v < 0) { set_error("fused: incomplete slab map"); return BAMD_ERR_INVALID; }
    st->packed_floats = (int)src.size();
    int rc = st->pack_src.ensure(src.size() * sizeof(int));
    if (rc) return rc;
    rc = st->slab_map.ensure(smap.size() * sizeof(int));
    if (rc) return rc;
    BAMD_HIP(hipMemcpy(st->pack_src.p, src.data(), src.size() * sizeof(int), hipMemcpyHostToDevice));
    BAMD_HIP(hipMemcpy(st->slab_map.p, smap.data(), smap.size() * sizeof(int), hipMemcpyHostToDevice));
    rc = h->packed.ensure(src.size() * sizeof(float));
    return rc;
}

constexpr int kF = 24, kZ = 15;   // the instantiated shape: CMS example, compression_ratio 1.6
using CMS = Net<kF, kZ>;
constexpr int kTrainLds = 2 * kQRows * kQS * (int)sizeof(float) + (CMS::bf_off(CMS::L) - CMS::bf_off(0)) * 16;

static bool shape_is_cms(const bamd_handle *h) {
    if (h->L != 8 || h->mode != BAMD_MODE_F32) return false;
    for (int i = 0; i <= 8; ++i)
        if (h->dims[i] != CMS::dim(i)) return false;
    return true;
}

static FusedState *state_of(bamd_handle *h) { return (FusedState *)h->fused_state; }

}  // namespace

int fused_setup(bamd_handle *h) {
    h->fused_ok = false;
    if (!shape_is_cms(h)) return BAMD_OK;
    const char *env = getenv("BALER_AMD_FORCE_GENERIC");
    if (env && env[0] == '1') return BAMD_OK;
    FusedState *st = new FusedState();
    h->fused_state = st;
    int rc = build_maps<kF, kZ>(h, st);
    if (rc) return rc;
    BAMD_HIP(hipFuncSetAttribute((const void *)train_kernel<kF, kZ>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                 kTrainLds));
    h->fused_ok = true;
    return BAMD_OK;
}

void fused_teardown(bamd_handle *h) {
    FusedState *st = state_of(h);
    if (!st) return;
    st->pack_src.release();
    st->slab_map.release();
    delete st;
    h->fused_state = nullptr;
}

int fused_pack(bamd_handle *h, hipStream_t s) {
    if (!h->fused_ok) return BAMD_OK;
    FusedState *st = state_of(h);
    hipLaunchKernelGGL(pack_k, dim3((st->packed_floats + 255) / 256), dim3(256), 0, s, (const float *)h->params.p,
                       (const int *)st->pack_src.p, st->packed_floats, (float *)h->packed.p);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

static int infer_grid(int64_t n) {
    int64_t wg = ((n + 15) / 16 + 3) / 4;
    return (int)(wg < 1 ? 1 : (wg > 1024 ? 1024 : wg));
}

int fused_encode(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *z, int z_dtype,
                 hipStream_t s) {
    hipLaunchKernelGGL((infer_kernel<kF, kZ, K_ENCODE>), dim3(infer_grid(n)), dim3(256), 0, s, (const v4 *)h->packed.p, x,
                       x_dtype == BAMD_F64, n, features, z, z_dtype == BAMD_F64, (const uint8_t *)nullptr, (double *)nullptr);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

int fused_decode(bamd_handle *h, const void *z, int z_dtype, int64_t n, const double *features, const uint8_t *int_mask,
                 void *out, int out_dtype, hipStream_t s) {
    hipLaunchKernelGGL((infer_kernel<kF, kZ, K_DECODE>), dim3(infer_grid(n)), dim3(256), 0, s, (const v4 *)h->packed.p, z,
                       z_dtype == BAMD_F64, n, features, out, out_dtype == BAMD_F64, int_mask, (double *)nullptr);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

int fused_forward_loss(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *recon,
                       int recon_dtype, double *loss_sum, hipStream_t s) {
    int grid = infer_grid(n);
    int rc = h->lossp.ensure(sizeof(double) * 1024);
    if (rc) return rc;
    hipLaunchKernelGGL((infer_kernel<kF, kZ, K_FORWARD>), dim3(grid), dim3(256), 0, s, (const v4 *)h->packed.p, x,
                       x_dtype == BAMD_F64, n, features, recon, recon_dtype == BAMD_F64, (const uint8_t *)nullptr,
                       (double *)h->lossp.p);
    hipLaunchKernelGGL(sum_loss_k, dim3(1), dim3(64), 0, s, (const double *)h->lossp.p, grid, 1.0 / kF, loss_sum);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

int fused_fwd_bwd(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads,
                  hipStream_t s) {
    FusedState *st = state_of(h);
    int64_t ngroups = (n + kRowsPerWG - 1) / kRowsPerWG;
    int grid = (int)(ngroups < st->nwg_max ? ngroups : st->nwg_max);
    const size_t slab_bytes = (size_t)CMS::slab_f4() * 16;
    int rc = h->slabs.ensure(slab_bytes * (size_t)grid);
    if (rc) return rc;
    const int lds_bytes = kTrainLds;
    hipLaunchKernelGGL((train_kernel<kF, kZ>), dim3(grid), dim3(256), lds_bytes, s, (const v4 *)h->packed.p, x,
                       x_dtype == BAMD_F64, n, features, (v4 *)h->slabs.p);
    const int np = CMS::nparams();
    hipLaunchKernelGGL(reduce_slabs_map_k<float>, dim3((np + 1 + 255) / 256), dim3(256), 0, s, (const float *)h->slabs.p,
                       grid, (int64_t)CMS::slab_f4() * 4, (const int *)st->slab_map.p, np, (CMS::slab_off(CMS::L) + 1) * 64 * 4,
                       1.0 / kF, (float *)grads);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

}  // namespace bamd
