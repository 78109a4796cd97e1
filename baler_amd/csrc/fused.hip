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
// (n-tile, k-tile) outputs are dealt round-robin to the waves, which keep them in MFMA accumulators for
// the whole persistent loop.  At the end every workgroup stores its tiles once into a PRIVATE slab in
// fragment order (plain 16-byte stores, no atomics) and a small kernel sums the slabs in a fixed order
// into the canonical state-dict layout: bitwise reproducible.
#include "fused.hpp"

#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <utility>

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
constexpr int kRowsPerWG = 64;

// Which layer the training pair is cut at: the decoder-gradient kernel back-propagates layers 7..kSplit, the
// encoder-gradient kernel layers kSplit-1..0 (recomputing the forward of layers 0..kSplit-2).  4 = cut at the
// bottleneck (16-float hand-off per row, encoder forward recomputed); 2 = cut after en2 (112-float hand-off, only
// en1 recomputed, 481 registers in the first kernel): measured 3.59 ms vs 3.91 ms per 1M rows, so 2 is the default.
#ifndef BAMD_SPLIT
#define BAMD_SPLIT 2
#endif
constexpr int kSplit = BAMD_SPLIT;

// ---- compile-time description of AE(F, Z): 8 layers F-200-100-50-Z-50-100-200-F -----------------------
template <int F, int Z> struct Net {
    static constexpr int L = 8;
    __host__ __device__ static constexpr int dim(int i) {
        return i == 0 ? F : i == 1 ? 200 : i == 2 ? 100 : i == 3 ? 50 : i == 4 ? Z : i == 5 ? 50 : i == 6 ? 100 : i == 7 ? 200 : F;
    }
    __host__ __device__ static constexpr bool act(int l) { return !(l == 3 || l == 7); }
    // packed buffer (float4 units):
    //   [ Wf(0..7) | Wb(7,6,..,1) | bias frags | E: Wf(0..2) Wb(3,2,1) ]   (+ slack for the ring's pad reads)
    // Wf = forward fragments, Wb = transposed fragments for the input-gradient chain, packed in the order
    // the kernels consume them so that every kernel walks ONE linear stream.  Region E duplicates the
    // encoder's fragments for the encoder-gradient kernel (forward 0..2 then backward 3..1).
    __host__ __device__ static constexpr int wcount(int l) { return tiles(dim(l)) * tiles(dim(l + 1)) * 64; }
    __host__ __device__ static constexpr int wf_off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += wcount(j); return s; }
    __host__ __device__ static constexpr int wb_off(int l) { int s = wf_off(L); for (int j = L - 1; j > l; --j) s += wcount(j); return s; }
    __host__ __device__ static constexpr int bf_off(int l) { int s = wb_off(0); for (int j = 0; j < l; ++j) s += tiles(dim(j + 1)) * 4; return s; }
    __host__ __device__ static constexpr int e_off() { return (bf_off(L) + 63) / 64 * 64; }
    __host__ __device__ static constexpr int ef_off(int l) { return e_off() + wf_off(l); }                       // l = 0..2
    __host__ __device__ static constexpr int eb_off(int l) { int s = e_off() + wf_off(kSplit - 1); for (int j = kSplit - 1; j > l; --j) s += wcount(j); return s; }  // l = kSplit-1..1
    // region L4 (the 4-row small-batch chain, lat4_chain_kernel): per chain GEMM and 64-feature output group, fragments of
    // 64 output features x 4 contraction indices; then one bias fragment per forward layer and group (L4 below)
    __host__ __device__ static constexpr int l4_off() { return eb_off(0) + 16 * 64; }
    __host__ __device__ static constexpr int l4_gemm_k(int g) { return g < 8 ? dim(g) : dim(15 - g + 1); }     // contraction length
    __host__ __device__ static constexpr int l4_gemm_n(int g) { return g < 8 ? dim(g + 1) : dim(15 - g); }     // outputs
    __host__ __device__ static constexpr int l4_groups(int g) { return (l4_gemm_n(g) + 63) / 64; }
    __host__ __device__ static constexpr int l4_ks(int g) { return (l4_gemm_k(g) + 3) / 4; }
    // offsets in float4 units: GEMM g holds [k / 4][output feature] pieces of 4 contraction indices, features NOT padded
    __host__ __device__ static constexpr int l4_frag_off(int g) { int s = 0; for (int j = 0; j < g; ++j) s += l4_gemm_n(j) * l4_ks(j); return s; }
    __host__ __device__ static constexpr int l4_frags() { return F <= 32 ? l4_frag_off(15) : 0; }      // narrow-input models only
    __host__ __device__ static constexpr int packed_f4() { return l4_off() + l4_frags() + 16 * 64; }
    // weight-gradient tiles of layer l: tiles(N) x tiles(K + 1) (the extra slot carries db)
    __host__ __device__ static constexpr int dw_tiles(int l) { return tiles(dim(l + 1)) * tiles(dim(l) + 1); }
    __host__ __device__ static constexpr int slab_off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += dw_tiles(j); return s; }
    __host__ __device__ static constexpr int slab_f4() { return slab_off(L) * 64 + 4; }  // + loss slot (16 B)
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
// Loads go through a buffer resource: ONE per-lane offset VGPR (lane*16) serves every fragment, the
// fragment index is an SGPR offset (s_mov), so the ring costs no address arithmetic and no address
// registers (flat/global loads made hipcc rebuild a 64-bit address per load in registers that aliased
// in-flight ring slots, which forced s_waitcnt vmcnt(0) in the middle of the pipeline).
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;   // lane * 16
};
__device__ __forceinline__ WStream make_stream(const v4 *base, int bytes, int lane) {
    WStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, bytes, 0x00020000);
    ws.voff = lane * 16;
    return ws;
}
__device__ __forceinline__ v4 frag(const WStream &ws, int idx) {
    return __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, idx * 1024, 0));
}
__device__ __forceinline__ v4 frag_rt(const WStream &ws, int idx) {   // runtime (wave-uniform) fragment index
    return __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, idx * 1024, 0));
}
#ifndef BAMD_RING
#define BAMD_RING 8
#endif
constexpr int kRing = BAMD_RING;
#ifndef BAMD_CHAIN_WAYS
#define BAMD_CHAIN_WAYS 2
#endif
struct Ring { v4 slot[kRing]; };

// The stream is walked cyclically (iteration after iteration), so its length is padded to a multiple of
// the ring size: fragment G always lives in slot G % kRing.  The pad fragments are loaded (they alias
// whatever follows the stream in the packed buffer) but never multiplied.
__host__ __device__ constexpr int pad_total(int t) { return (t + kRing - 1) / kRing * kRing; }

template <int TOTAL>
__device__ __forceinline__ void ring_prime(Ring &ring, const WStream &ws) {
    static_assert(TOTAL >= kRing, "stream shorter than the ring");
#pragma unroll
    for (int i = 0; i < kRing; ++i) ring.slot[i] = frag(ws, i);
}
// end of an iteration: step over the pad fragments, refilling their slots for the next iteration
template <int TOTAL>
__device__ __forceinline__ void ring_tail(Ring &ring, const WStream &ws) {
#pragma unroll
    for (int G = TOTAL; G < pad_total(TOTAL); ++G)
        ring.slot[G % kRing] = frag(ws, (G + kRing) % pad_total(TOTAL));
    __builtin_amdgcn_sched_barrier(0);
}

// out^T tiles += frags . in^T tiles;  KD = reduction dimension (size of `in`); this layer's fragments are
// stream fragments BASE .. BASE + tiles(KD)*NT - 1 in [q][t] order (component r = step r of k-tile q).
template <int KD, int NT, int BASE, int TOTAL>
__device__ __forceinline__ void chain_gemm(const v4 (&in)[tiles(KD)], v4 (&out)[NT], Ring &ring, const WStream &ws) {
    // v_mfma_f32_16x16x4_f32 issues every 32 cycles but a DEPENDENT accumulate needs 40: consecutive
    // fragments (different output tiles) are walked in pairs with their MFMA steps interleaved, so each
    // wave alternates between two independent accumulators and the pipe stays back-to-back at one wave
    // per SIMD.
    constexpr int NF = tiles(KD) * NT;
    constexpr int W = BAMD_CHAIN_WAYS;   // fragments whose MFMA steps are interleaved (independent accumulators)
#pragma unroll
    for (int f = 0; f < NF; f += W) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < W; ++j)
                if (f + j < NF && r < tile_steps(KD, (f + j) / NT))
                    out[(f + j) % NT] = mfma(ring.slot[(BASE + f + j) % kRing][r], in[(f + j) / NT][r], out[(f + j) % NT]);
#pragma unroll
        for (int j = 0; j < W; ++j)
            if (f + j < NF)
                ring.slot[(BASE + f + j) % kRing] = frag(ws, (BASE + f + j + kRing) % pad_total(TOTAL));
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Two batch tiles per wave: every fragment feeds both tiles' accumulators (the two independent MFMA chains that
// chain_gemm gets from pairing fragments), so a 32-row pass issues HALF the fragment loads per MFMA of a 16-row one.
// Every non-MFMA instruction costs ~8 cycles of MFMA issue; used by the inference kernels, which have the registers.
template <int KD, int NT, int BASE, int TOTAL>
__device__ __forceinline__ void chain_gemm2(const v4 (&in0)[tiles(KD)], const v4 (&in1)[tiles(KD)], v4 (&out0)[NT], v4 (&out1)[NT],
                                            Ring &ring, const WStream &ws) {
    constexpr int NF = tiles(KD) * NT;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < tile_steps(KD, f / NT)) {
                out0[f % NT] = mfma(ring.slot[(BASE + f) % kRing][r], in0[f / NT][r], out0[f % NT]);
                out1[f % NT] = mfma(ring.slot[(BASE + f) % kRing][r], in1[f / NT][r], out1[f % NT]);
            }
        ring.slot[(BASE + f) % kRing] = frag(ws, (BASE + f + kRing) % pad_total(TOTAL));
        __builtin_amdgcn_sched_barrier(0);
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
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f k2 = (v2f){0.01f, 0.01f};
    asm volatile("" : "+v"(k2));   // a register pair, not a literal: with the literal hipcc scalarises back to v_mul_f32
    const v4 kk = (v4){k2[0], k2[1], k2[0], k2[1]};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        // x > 0 ? x : 0.01 x == max(x, 0.01 x).  The products of a tile are one vector multiply (two v_pk_mul_f32 instead of
        // four v_mul_f32: every VALU instruction costs MFMA issue slots), then ONE v_maximum3_f32 per register (gfx950's IEEE-2019
        // maximum: a NaN stays a NaN, as in torch's leaky_relu).  fmaxf() costs a canonicalising v_max in front of the v_max; the
        // med3(x, 0.01 x, FLT_MAX) of rounds 1-3 was one instruction too but turned a NaN into FLT_MAX (v_med3 falls back to
        // min3 on a NaN input) -- found by tests/test_gpu_edge.py
        v4 m = a[t] * kk;
        asm("" : "+v"(m));            // keep the product a vector: extract-of-fmul would be scalarised again
#pragma unroll
        for (int r = 0; r < 4; ++r) a[t][r] = __builtin_elementwise_maximum(a[t][r], m[r]);
    }
}
// dZ = dY * lrelu'(pre) ; sign(pre) == sign(post-activation y)
template <int NT> __device__ __forceinline__ void lrelu_bwd(v4 (&d)[NT], const v4 (&y)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        v4 sl;                                    // slope per register (v_cmp + v_cndmask), then one vector multiply
#pragma unroll
        for (int r = 0; r < 4; ++r) sl[r] = y[t][r] > 0.f ? 1.0f : 0.01f;
        d[t] = d[t] * sl;
    }
}

// one Linear layer of the forward chain.  S = stream description (frag base of every layer, total)
template <class N, class S, int l>
__device__ __forceinline__ void fwd_layer(const v4 (&in)[tiles(N::dim(l))], v4 (&out)[tiles(N::dim(l + 1))],
                                          Ring &ring, const WStream &ws, const v4 *bias_lds, int lane) {
    init_bias(out, bias_lds + (N::bf_off(l) - N::bf_off(0)), lane);
    chain_gemm<N::dim(l), tiles(N::dim(l + 1)), S::fwd_base(l), S::total>(in, out, ring, ws);
    if (N::act(l)) lrelu(out);
}
template <class N, class S, int l>
__device__ __forceinline__ void fwd_layer2(const v4 (&in0)[tiles(N::dim(l))], const v4 (&in1)[tiles(N::dim(l))],
                                           v4 (&out0)[tiles(N::dim(l + 1))], v4 (&out1)[tiles(N::dim(l + 1))], Ring &ring,
                                           const WStream &ws, const v4 *bias_lds, int lane) {
    init_bias(out0, bias_lds + (N::bf_off(l) - N::bf_off(0)), lane);
#pragma unroll
    for (int t = 0; t < tiles(N::dim(l + 1)); ++t) out1[t] = out0[t];
    chain_gemm2<N::dim(l), tiles(N::dim(l + 1)), S::fwd_base(l), S::total>(in0, in1, out0, out1, ring, ws);
    if (N::act(l)) { lrelu(out0); lrelu(out1); }
}
// dY_{l-1}^T = W_l^T dZ_l^T
template <class N, class S, int l>
__device__ __forceinline__ void bwd_layer(const v4 (&dz)[tiles(N::dim(l + 1))], v4 (&dx)[tiles(N::dim(l))],
                                          Ring &ring, const WStream &ws) {
    zero_tiles(dx);
    chain_gemm<N::dim(l + 1), tiles(N::dim(l)), S::bwd_base(l), S::total>(dz, dx, ring, ws);
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
template <class N> struct StreamTrainDec {  // decoder-gradient kernel: forward 0..7 then backward 7,6,5,4
    static constexpr int fwd_base(int l) { return N::wf_off(l) / 64; }
    static constexpr int bwd_base(int l) { return N::wb_off(l) / 64; }
    static constexpr int total = N::wb_off(kSplit - 1) / 64;
    static constexpr int start_f4 = 0;
};
template <class N> struct StreamTrainEnc {  // encoder-gradient kernel: forward 0..2 then backward 3,2,1 (region E)
    static constexpr int fwd_base(int l) { return (N::ef_off(l) - N::e_off()) / 64; }
    static constexpr int bwd_base(int l) { return (N::eb_off(l) - N::e_off()) / 64; }
    static constexpr int total = (N::eb_off(0) - N::e_off()) / 64;
    static constexpr int start_f4 = N::e_off();
};

template <class N> struct StreamWideEnc {   // wide models: forward fragments of layers 1..3 (layer 0 is streamed by its own loop)
    static constexpr int fwd_base(int l) { return (N::wf_off(l) - N::wf_off(1)) / 64; }
    static constexpr int total = (N::wf_off(4) - N::wf_off(1)) / 64;
    static constexpr int start_f4 = N::wf_off(1);
};
template <class N> struct StreamWideDec {   // wide models: forward fragments of layers 4..6 (layer 7 is streamed by its own loop)
    static constexpr int fwd_base(int l) { return (N::wf_off(l) - N::wf_off(4)) / 64; }
    static constexpr int total = (N::wf_off(7) - N::wf_off(4)) / 64;
    static constexpr int start_f4 = N::wf_off(4);
};

// ---- row I/O in slot order ------------------------------------------------------------------------------
// Row loads are split in two so that the training kernels can issue the NEXT row group's loads before a
// long MFMA phase and convert them after it (the HBM latency of the first touch of a row is ~2 us).
template <int D> struct RawRows { double v[tiles(D) * 4]; };

// RT = "run-time width": the kernel is instantiated for a CLASS of narrow tables -- D is the class width (16 T - 1: T tiles, the last
// one partial so that the ones slot of the weight-gradient images has a place), `dr` <= D the table's real column count, a kernel
// argument.  Slots map to features exactly as for a D-column table (slot_feature); a slot whose feature is >= dr reads feature 0 and
// is ZEROED, so that padding meets the zero weights / biases the pack map gives it with a zero on the data side too: the loss, dL/drecon
// and every weight-gradient product see exact zeros there, and nothing else in a kernel needs to know the real width.
template <int D, bool RT = false>
__device__ __forceinline__ void load_rows_issue(RawRows<D> &raw, const void *x, int is_f64, int64_t row, bool valid, int lane, int dr = D) {
    // Branch-free: lanes beyond the last row read row 0 and padding slots read feature 0 (always inside the table), so no
    // lane needs an exec-masked branch or a zero fill.  Their values are never used: padding slots meet zero weights and
    // unmapped gradient slots, rows beyond n get a zero loss gradient / are not stored.
    constexpr int NS = tiles(D) * 4;
    const int g = lane >> 4;
    const int64_t base = (valid ? row : 0) * (RT ? dr : D);
    // slot (t, r) holds a feature on SOME lane group iff the tile is full or 4r < its valid count (r-major partial tiles)
    auto used = [](int s) { return D - 16 * (s >> 2) >= 16 || 4 * (s & 3) < D - 16 * (s >> 2); };
    // ... and on EVERY lane group iff the tile is full or 4r + 3 < count: then its index needs no select
    auto every = [](int s) { return D - 16 * (s >> 2) >= 16 || 4 * (s & 3) + 3 < D - 16 * (s >> 2); };
    auto feat = [&](int s) {
        const int t = s >> 2, r = s & 3;
        int f;
        if (D - 16 * t >= 16) f = 16 * t + 4 * g + r;
        else {
            f = 16 * t + 4 * r + g;                       // r-major partial tile (slot_feature)
            f = every(s) ? f : (4 * r + g < D - 16 * t ? f : 0);
        }
        if (RT) f = f < dr ? f : 0;
        return f;
    };
    if (is_f64) {
#pragma unroll
        for (int s = 0; s < NS; ++s) raw.v[s] = used(s) ? ((const double *)x)[base + feat(s)] : 0.0;
    } else {
        float w[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) w[s] = used(s) ? ((const float *)x)[base + feat(s)] : 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) raw.v[s] = (double)w[s];
    }
}

template <int D, bool RT = false>
__device__ __forceinline__ void load_rows_finish(v4 (&a)[tiles(D)], RawRows<D> &raw, bool valid, int lane,
                                                 const double *__restrict__ feats, int dr = D) {
    constexpr int NS = tiles(D) * 4;
    const int g = lane >> 4;
    if (feats) {
        double mn[NS], rg[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int t = s >> 2, r = s & 3, left = D - 16 * t;
            int f = left >= 16 ? 16 * t + 4 * g + r : 16 * t + 4 * r + g;
            if (!(left >= 16 || 4 * r + 3 < left)) f = (4 * r + g < left) ? f : 0;     // padding on some lane group: clamp
            if (!(left >= 16 || 4 * r < left)) f = 0;                                  // padding everywhere
            if (RT) f = f < dr ? f : 0;
            mn[s] = feats[f];
            rg[s] = feats[(RT ? dr : D) + f];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) raw.v[s] = (raw.v[s] - mn[s]) / rg[s];   // (x - min)/(max - min) in float64
    }
    if (RT) {      // the class's slots beyond the table's width: exact zeros
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int f = slot_feature(D, s >> 2, g, s & 3);
            if (!(f >= 0 && f < dr)) raw.v[s] = 0.0;
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s >> 2][s & 3] = (float)raw.v[s];
}

template <int D, bool RT = false>
__device__ __forceinline__ void load_rows(v4 (&a)[tiles(D)], const void *x, int is_f64, int64_t row, bool valid,
                                          int lane, const double *__restrict__ feats, int dr = D) {
    RawRows<D> raw;
    load_rows_issue<D, RT>(raw, x, is_f64, row, valid, lane, dr);
    load_rows_finish<D, RT>(a, raw, valid, lane, feats, dr);
}

// Wide rows (D > 64, e.g. the 512-column table): tile by tile, 16/32-byte vector loads for full tiles
// (4 consecutive features per lane), no fp64 staging array.
template <int D>
__device__ __forceinline__ void load_rows_wide(v4 (&a)[tiles(D)], const void *x, int is_f64, int64_t row, bool valid,
                                               int lane, const double *__restrict__ feats) {
    const int g = lane >> 4;
#pragma unroll
    for (int t = 0; t < tiles(D); ++t) {
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        if (valid) {
            if (D - 16 * t >= 16) {
                const int64_t i = row * D + 16 * t + 4 * g;
                if (is_f64) {
                    const double2 lo = *(const double2 *)((const double *)x + i), hi = *(const double2 *)((const double *)x + i + 2);
                    v[0] = lo.x; v[1] = lo.y; v[2] = hi.x; v[3] = hi.y;
                } else {
                    const float4 w = *(const float4 *)((const float *)x + i);
                    v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = slot_feature(D, t, g, r);
                    if (f >= 0) v[r] = is_f64 ? ((const double *)x)[row * D + f] : (double)((const float *)x)[row * D + f];
                }
            }
            if (feats) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = slot_feature(D, t, g, r);
                    if (f >= 0) v[r] = (v[r] - feats[f]) / feats[D + f];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) a[t][r] = (float)v[r];
    }
}

template <int D, bool RT = false>
__device__ __forceinline__ void store_rows(const v4 (&a)[tiles(D)], void *out, int is_f64, int64_t row, bool valid,
                                           int lane, const double *__restrict__ renorm, const uint8_t *__restrict__ imask, int dr = D) {
    // ONE lane mask (valid row) around everything and the uniform dtype / renorm tests outside the element loops; a slot
    // needs its own lane test only if it is padding on SOME lane group (the last register of a partial r-major tile).
    const int g = lane >> 4;
    if (!valid) return;
    double v[tiles(D)][4];
#pragma unroll
    for (int t = 0; t < tiles(D); ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[t][r] = (double)a[t][r];
            const int f = slot_feature(D, t, g, r);
            if (renorm && f >= 0 && (!RT || f < dr)) {
                // norm*range + min with two roundings (numpy), then trunc for "int" columns (baler.py:420-435)
                v[t][r] = __dadd_rn(__dmul_rn(v[t][r], renorm[(RT ? dr : D) + f]), renorm[f]);
                if (imask && imask[f]) v[t][r] = trunc(v[t][r]);
            }
        }
    if (RT) {      // run-time width (a class instantiation): element by element, rows of dr values
#pragma unroll
        for (int t = 0; t < tiles(D); ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = slot_feature(D, t, g, r);
                if (f >= 0 && f < dr) {
                    const int64_t i = row * dr + f;
                    if (is_f64) ((double *)out)[i] = v[t][r]; else ((float *)out)[i] = (float)v[t][r];
                }
            }
        return;
    }
#pragma unroll
    for (int t = 0; t < tiles(D); ++t) {
        const int left = D - 16 * t;                       // features from this tile on
        if (left >= 16 && D % 4 == 0) {                    // full tile, aligned rows: the lane's 4 consecutive features as vectors
            const int64_t i = row * D + 16 * t + 4 * g;
            if (is_f64) {
                *(double2 *)((double *)out + i) = make_double2(v[t][0], v[t][1]);
                *(double2 *)((double *)out + i + 2) = make_double2(v[t][2], v[t][3]);
            } else {
                *(float4 *)((float *)out + i) = make_float4((float)v[t][0], (float)v[t][1], (float)v[t][2], (float)v[t][3]);
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool some = left >= 16 || 4 * r < left, all = left >= 16 || 4 * r + 3 < left;   // compile-time per (t, r)
            if (!some) continue;
            const int f = slot_feature(D, t, g, r);
            if (all || f >= 0) {
                const int64_t i = row * D + f;
                if (is_f64) ((double *)out)[i] = v[t][r]; else ((float *)out)[i] = (float)v[t][r];
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

template <int F, int Z, int KIND, bool RT = false>
__global__ void __launch_bounds__(256) infer_kernel(const v4 *packed, const void *__restrict__ xin, int in_f64,
                                                    int64_t n, const double *__restrict__ feats, void *__restrict__ out,
                                                    int out_f64, const uint8_t *__restrict__ imask,
                                                    double *__restrict__ loss_part, int fr, int zr) {
    using N = Net<F, Z>;
    using S = typename std::conditional<KIND == K_ENCODE, StreamEncode<N>,
                                        typename std::conditional<KIND == K_DECODE, StreamDecode<N>, StreamForward<N>>::type>::type;
    __shared__ __attribute__((aligned(16))) v4 bias_lds[N::bf_off(N::L) - N::bf_off(0)];
    __shared__ double sh[256];
    stage_bias<N>(bias_lds, packed);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t ntile = (n + 15) / 16;
    WStream ws = make_stream(packed + S::start_f4, (N::packed_f4() - S::start_f4) * 16, threadIdx.x & 63);
    double lacc = 0.0;
    Ring ring;
    ring_prime<S::total>(ring, ws);
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntile; tile += (int64_t)gridDim.x * 4) {
        const int64_t row = tile * 16 + (lane & 15);
        const bool valid = row < n;
        // keep the weight loads INSIDE the loop: without this LICM hoists all of them (loop-invariant
        // addresses) and spills the whole model to scratch
        asm volatile("" : "+v"(ws.voff));
        if (KIND == K_ENCODE || KIND == K_FORWARD) {
            v4 a0[tiles(F)], a1[13], a2[7], a3[4], a4[tiles(Z)];
            if (F > 64 && !RT) load_rows_wide<F>(a0, xin, in_f64, row, valid, lane, feats);
            else load_rows<F, RT>(a0, xin, in_f64, row, valid, lane, feats, fr);
            fwd_layer<N, S, 0>(a0, a1, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 1>(a1, a2, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 2>(a2, a3, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 3>(a3, a4, ring, ws, bias_lds, lane);
            if (KIND == K_ENCODE) {
                store_rows<Z, RT>(a4, out, out_f64, row, valid, lane, nullptr, nullptr, zr);
            } else {
                v4 a5[4], a6[7], a7[13], a8[tiles(F)];
                fwd_layer<N, S, 4>(a4, a5, ring, ws, bias_lds, lane);
                fwd_layer<N, S, 5>(a5, a6, ring, ws, bias_lds, lane);
                fwd_layer<N, S, 6>(a6, a7, ring, ws, bias_lds, lane);
                fwd_layer<N, S, 7>(a7, a8, ring, ws, bias_lds, lane);
                if (out) store_rows<F, RT>(a8, out, out_f64, row, valid, lane, nullptr, nullptr, fr);
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
            load_rows<Z, RT>(a4, xin, in_f64, row, valid, lane, nullptr, zr);
            fwd_layer<N, S, 4>(a4, a5, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 5>(a5, a6, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 6>(a6, a7, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 7>(a7, a8, ring, ws, bias_lds, lane);
            store_rows<F, RT>(a8, out, out_f64, row, valid, lane, feats, imask, fr);
        }
        ring_tail<S::total>(ring, ws);
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

// Encode / decode with TWO 16-row tiles per wave (see chain_gemm2): same results, half the fragment loads per MFMA.
template <int F, int Z, int KIND, bool RT = false>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) infer2_kernel(const v4 *packed, const void *__restrict__ xin, int in_f64, int64_t n,
                                                     const double *__restrict__ feats, void *__restrict__ out, int out_f64,
                                                     const uint8_t *__restrict__ imask, int fr, int zr) {
    using N = Net<F, Z>;
    using S = typename std::conditional<KIND == K_ENCODE, StreamEncode<N>, StreamDecode<N>>::type;
    static_assert(KIND == K_ENCODE || KIND == K_DECODE, "pair kernel: encode or decode");
    __shared__ __attribute__((aligned(16))) v4 bias_lds[N::bf_off(N::L) - N::bf_off(0)];
    stage_bias<N>(bias_lds, packed);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t npair = (n + 31) / 32;
    WStream ws = make_stream(packed + S::start_f4, (N::packed_f4() - S::start_f4) * 16, threadIdx.x & 63);
    Ring ring;
    ring_prime<S::total>(ring, ws);
    // The NEXT pass's rows are requested in front of this pass's last long layer (behind the register peak) and converted at
    // the top of the next pass: requested at the top of their own pass, the wave sat out one HBM round trip (~1.4 us of a 14-us pass)
    // behind the previous pass's stores before its first MFMA (BALER_AMD_INFER_PREFETCH=0 at build time: BAMD_INFER_NOPF).
    constexpr int DIN = KIND == K_ENCODE ? F : Z;
    RawRows<DIN> ra, rb;
    {
        const int64_t p0 = (int64_t)blockIdx.x * 4 + wave;
        const int64_t q0 = p0 * 32 + (lane & 15), q1 = q0 + 16;
        load_rows_issue<DIN, RT>(ra, xin, in_f64, q0, q0 < n, lane, (KIND == K_ENCODE ? fr : zr));
        load_rows_issue<DIN, RT>(rb, xin, in_f64, q1, q1 < n, lane, (KIND == K_ENCODE ? fr : zr));
    }
    for (int64_t pr = (int64_t)blockIdx.x * 4 + wave; pr < npair; pr += (int64_t)gridDim.x * 4) {
        const int64_t r0 = pr * 32 + (lane & 15), r1 = r0 + 16;
        const bool v0 = r0 < n, v1 = r1 < n;
        const int64_t pn = pr + (int64_t)gridDim.x * 4;
        const int64_t n0 = pn * 32 + (lane & 15), n1 = n0 + 16;
        asm volatile("" : "+v"(ws.voff));   // keep the weight loads inside the loop (see infer_kernel)
        if constexpr (KIND == K_ENCODE) {
            v4 a0[tiles(F)], b0[tiles(F)], a1[13], b1[13], a2[7], b2[7], a3[4], b3[4], a4[tiles(Z)], b4[tiles(Z)];
            load_rows_finish<F, RT>(a0, ra, v0, lane, feats, fr);
            load_rows_finish<F, RT>(b0, rb, v1, lane, feats, fr);
            fwd_layer2<N, S, 0>(a0, b0, a1, b1, ring, ws, bias_lds, lane);
            fwd_layer2<N, S, 1>(a1, b1, a2, b2, ring, ws, bias_lds, lane);
            __builtin_amdgcn_sched_barrier(0);
            load_rows_issue<F, RT>(ra, xin, in_f64, n0, n0 < n, lane, fr);      // (behind the widest layer: the raw rows do not add to the register peak)
            load_rows_issue<F, RT>(rb, xin, in_f64, n1, n1 < n, lane, fr);
            __builtin_amdgcn_sched_barrier(0);
            fwd_layer2<N, S, 2>(a2, b2, a3, b3, ring, ws, bias_lds, lane);
            fwd_layer2<N, S, 3>(a3, b3, a4, b4, ring, ws, bias_lds, lane);
            store_rows<Z, RT>(a4, out, out_f64, r0, v0, lane, nullptr, nullptr, zr);
            store_rows<Z, RT>(b4, out, out_f64, r1, v1, lane, nullptr, nullptr, zr);
        } else {
            v4 a4[tiles(Z)], b4[tiles(Z)], a5[4], b5[4], a6[7], b6[7], a7[13], b7[13], a8[tiles(F)], b8[tiles(F)];
            load_rows_finish<Z, RT>(a4, ra, v0, lane, nullptr, zr);
            load_rows_finish<Z, RT>(b4, rb, v1, lane, nullptr, zr);
            fwd_layer2<N, S, 4>(a4, b4, a5, b5, ring, ws, bias_lds, lane);
            __builtin_amdgcn_sched_barrier(0);
            load_rows_issue<Z, RT>(ra, xin, in_f64, n0, n0 < n, lane, zr);
            load_rows_issue<Z, RT>(rb, xin, in_f64, n1, n1 < n, lane, zr);
            __builtin_amdgcn_sched_barrier(0);
            fwd_layer2<N, S, 5>(a5, b5, a6, b6, ring, ws, bias_lds, lane);
            fwd_layer2<N, S, 6>(a6, b6, a7, b7, ring, ws, bias_lds, lane);
            fwd_layer2<N, S, 7>(a7, b7, a8, b8, ring, ws, bias_lds, lane);
            store_rows<F, RT>(a8, out, out_f64, r0, v0, lane, feats, imask, fr);
            store_rows<F, RT>(b8, out, out_f64, r1, v1, lane, feats, imask, fr);
        }
        ring_tail<S::total>(ring, ws);
    }
}

// ---- wide first / last layer (CFD_dense_AE(2500, 25): 2500 -> 200 -> ... -> 25 -> ... -> 200 -> 2500) ----------------------
// 95 % of the model's work is en1 and de4, whose 2500-wide side cannot sit in registers the way the chain's tiles do.  They
// are STREAMED: every wave owns 16 rows, keeps the 13 tiles of the 200-feature side in registers (the accumulators of en1,
// the B operand of de4) and walks the wide dimension 16 features at a time -- one 16-byte row segment per lane and 13
// 1-KiB weight fragments per 52 MFMAs (the register chain's ratio of one load per 4 MFMAs), fragments one chunk ahead in a
// ping-pong register buffer.  The six narrow layers in between are the ordinary register chain.  One launch for encode,
// one for decode; activations never touch HBM (the layer-wise path writes and re-reads 42 KB of them per row).
// FULL: the caller guarantees a full chunk (kc < F / 16).  The streamed loops only ever load full chunks and treat the
// remainder of the wide dimension as an epilogue: with the "full chunk?" test inside the loop every row load sat behind a branch
// whose other side is a different load sequence, and hipcc joins such paths with conservative waits.
// WRT ("run-time width"): the kernel serves a CLASS of wide models -- F is the class width (a multiple of 16: the geometry of the
// packed weights, where every chunk is a full one), `fr` <= F the model's real column count, a kernel argument: rows are fr values
// long, the loops over the wide dimension run tiles(fr) chunks, and the one chunk that reaches beyond fr reads / stores its
// features one by one (natural order: a full class tile), the rest zero -- they meet the zero weights the pack map gives them.
template <int F, bool FULL = false, bool WRT = false>
__device__ __forceinline__ v4 wide_x_chunk(const void *x, int in_f64, int64_t row, int kc, int g, int fr = F) {
    // features 16 kc + 4 g .. + 3 of `row` (register r = MFMA step r, k = 4 g + r: the packed weights' order for full tiles);
    // the partial last chunk is r-major (slot_feature): register 0 of lane group g = feature 16 kc + g, the rest padding
    v4 v = (v4){0.f, 0.f, 0.f, 0.f};
    if constexpr (WRT && !FULL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * kc + 4 * g + r;
            if (f < fr) v[r] = in_f64 ? (float)((const double *)x)[row * fr + f] : ((const float *)x)[row * fr + f];
        }
        return v;
    }
    if (FULL || 16 * kc + 16 <= F) {
        const int64_t i = row * (WRT ? fr : F) + 16 * kc + 4 * g;
        if (in_f64) {
            const double2 lo = *(const double2 *)((const double *)x + i), hi = *(const double2 *)((const double *)x + i + 2);
            v = (v4){(float)lo.x, (float)lo.y, (float)hi.x, (float)hi.y};
        } else {
            v = *(const v4 *)((const float *)x + i);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = slot_feature(F, kc, g, r);
            if (f >= 0 && r < tile_steps(F, kc)) v[r] = in_f64 ? (float)((const double *)x)[row * F + f] : ((const float *)x)[row * F + f];
        }
    }
    return v;
}

// store tile t of a wide row (C layout) as float / double
template <int F, bool FULL = false, bool WRT = false>
__device__ __forceinline__ void wide_store_tile(const v4 &o, void *out, int out_f64, int64_t row, int t, int g, int fr = F) {
    if constexpr (WRT && !FULL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = 16 * t + 4 * g + r;
            if (f < fr) {
                if (out_f64) ((double *)out)[row * fr + f] = (double)o[r]; else ((float *)out)[row * fr + f] = o[r];
            }
        }
        return;
    }
    if (FULL || 16 * t + 16 <= F) {      // full tile: the lane's 4 consecutive features as one vector store
        const int64_t i = row * (WRT ? fr : F) + 16 * t + 4 * g;
        if (out_f64) {
            *(double2 *)((double *)out + i) = make_double2((double)o[0], (double)o[1]);
            *(double2 *)((double *)out + i + 2) = make_double2((double)o[2], (double)o[3]);
        } else {
            *(v4 *)((float *)out + i) = o;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = slot_feature(F, t, g, r);
            if (f >= 0) {
                if (out_f64) ((double *)out)[row * F + f] = (double)o[r]; else ((float *)out)[row * F + f] = o[r];
            }
        }
    }
}

// Two 16-row tiles per wave: every fragment feeds both tiles' accumulators (half the fragment traffic per MFMA), and the 13
// fragments of a chunk are fetched in two halves (tiles 0..6 / 7..12) so that two row tiles cost the registers of one:
// 104 accumulators + 52 fragment + 32 x registers.  Each half multiplies 56 / 48 MFMAs while the other half loads.
template <int F>
__device__ __forceinline__ void wide_in_product2(v4 (&acc0)[13], v4 (&acc1)[13], const WStream &ww, const void *xin, int in_f64,
                                                 int64_t rrow0, int64_t rrow1, int g) {
    constexpr int KC = tiles(F);
    v4 wlo[7], whi[6];
    auto load_lo = [&](int kc) {
        kc = kc < KC ? kc : KC - 1;
#pragma unroll
        for (int t = 0; t < 7; ++t) wlo[t] = frag_rt(ww, kc * 13 + t);
    };
    auto load_hi = [&](int kc) {
#pragma unroll
        for (int t = 0; t < 6; ++t) whi[t] = frag_rt(ww, kc * 13 + 7 + t);
    };
    auto steps_of = [&](int kc) { return (F % 16 != 0 && kc == KC - 1) ? tile_steps(F, KC - 1) : 4; };
    auto mm_lo = [&](const v4 &x0, const v4 &x1, int steps) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < steps) {
#pragma unroll
                for (int t = 0; t < 7; ++t) {
                    acc0[t] = mfma(wlo[t][r], x0[r], acc0[t]);
                    acc1[t] = mfma(wlo[t][r], x1[r], acc1[t]);
                }
            }
    };
    auto mm_hi = [&](const v4 &x0, const v4 &x1, int steps) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < steps) {
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    acc0[7 + t] = mfma(whi[t][r], x0[r], acc0[7 + t]);
                    acc1[7 + t] = mfma(whi[t][r], x1[r], acc1[7 + t]);
                }
            }
    };
    constexpr int kXA = 4;
    v4 x0r[kXA], x1r[kXA];
    auto load_x0 = [&](int kc) { return wide_x_chunk<F>(xin, in_f64, rrow0, kc < KC ? kc : 0, g); };
    auto load_x1 = [&](int kc) { return wide_x_chunk<F>(xin, in_f64, rrow1, kc < KC ? kc : 0, g); };
    load_lo(0);
#pragma unroll
    for (int u = 0; u < kXA; ++u) { x0r[u] = load_x0(u); x1r[u] = load_x1(u); }
    auto chunk = [&](int kc, auto slot) {
        constexpr int SL = decltype(slot)::value;
        const int steps = steps_of(kc);
        load_hi(kc);
        mm_lo(x0r[SL], x1r[SL], steps);
        __builtin_amdgcn_sched_barrier(0);
        load_lo(kc + 1);
        mm_hi(x0r[SL], x1r[SL], steps);
        x0r[SL] = load_x0(kc + kXA);
        x1r[SL] = load_x1(kc + kXA);
        __builtin_amdgcn_sched_barrier(0);
    };
    int kc = 0;
    for (; kc + kXA <= KC; kc += kXA) {
        chunk(kc, std::integral_constant<int, 0>());
        chunk(kc + 1, std::integral_constant<int, 1>());
        chunk(kc + 2, std::integral_constant<int, 2>());
        chunk(kc + 3, std::integral_constant<int, 3>());
    }
    if (kc < KC) chunk(kc, std::integral_constant<int, 0>());
    if (kc + 1 < KC) chunk(kc + 1, std::integral_constant<int, 1>());
    if (kc + 2 < KC) chunk(kc + 2, std::integral_constant<int, 2>());
}

template <int F, int Z, bool IN64>       // IN64: the row type is a template parameter (no branch in front of the row loads)
__global__ void __launch_bounds__(256) wide_encode2_kernel(const v4 *packed, const void *__restrict__ xin, int64_t n,
                                                           void *__restrict__ out, int out_f64) {
    constexpr int in_f64 = IN64 ? 1 : 0;
    using N = Net<F, Z>;
    using S = StreamWideEnc<N>;
    __shared__ __attribute__((aligned(16))) v4 bias_lds[N::bf_off(N::L) - N::bf_off(0)];
    stage_bias<N>(bias_lds, packed);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
    const int64_t npair = (n + 31) / 32;
    WStream ws = make_stream(packed + S::start_f4, (N::packed_f4() - S::start_f4) * 16, lane);
    WStream ww = make_stream(packed + N::wf_off(0), N::wcount(0) * 16, lane);
    for (int64_t pr = (int64_t)blockIdx.x * 4 + wave; pr < npair; pr += (int64_t)gridDim.x * 4) {
        const int64_t r0 = pr * 32 + (lane & 15), r1 = r0 + 16;
        const bool v0 = r0 < n, v1 = r1 < n;
        asm volatile("" : "+v"(ws.voff), "+v"(ww.voff));
        v4 a1[13], b1[13];
        init_bias(a1, bias_lds, lane);
#pragma unroll
        for (int t = 0; t < 13; ++t) b1[t] = a1[t];
        wide_in_product2<F>(a1, b1, ww, xin, in_f64, v0 ? r0 : 0, v1 ? r1 : 0, g);
        lrelu(a1);
        lrelu(b1);
        Ring ring;
        ring_prime<S::total>(ring, ws);
        v4 a2[7], b2[7], a3[4], b3[4], a4[tiles(Z)], b4[tiles(Z)];
        fwd_layer2<N, S, 1>(a1, b1, a2, b2, ring, ws, bias_lds, lane);
        fwd_layer2<N, S, 2>(a2, b2, a3, b3, ring, ws, bias_lds, lane);
        fwd_layer2<N, S, 3>(a3, b3, a4, b4, ring, ws, bias_lds, lane);
        store_rows<Z>(a4, out, out_f64, r0, v0, lane, nullptr, nullptr);
        store_rows<Z>(b4, out, out_f64, r1, v1, lane, nullptr, nullptr);
    }
}

// ---- wide models in the bf16 mode: en1 / de4 on v_mfma_f32_16x16x32_bf16, the six narrow layers on the fp32 chain ------------
// 95 % of the work of CFD_dense_AE(2500, 25) is the two wide layers.  On the bf16 MFMA (16 cycles for 16 x 16 x 32) they stop
// being the bound: a frame is 10 KB of float32 input (encode) or output (decode), so the kernels are HBM-BOUND at
// 8 TB/s / 10 KB = 0.8 G frames/s (fp32 MFMA bound: 0.15 G).  Two 16-row tiles per wave share every 1-KiB bf16 fragment
// (16 outputs x 32 k; one tile per wave would need 250 B/clk/CU of fragments through a 64 B/clk L1).
//   encode: B operand = 8 consecutive floats of the row per lane (k = 32 c + 8 g + j: natural order, a full 128-byte line per
//           row and chunk), converted with four v_cvt_pk_bf16_f32; A = W0 fragments [chunk][tile][lane].
//   decode: the fp32 chain leaves a7 in C layout (tile q, register r = feature slot (q, g, r)); k slot (g, j) of chunk c is
//           DEFINED as (tile 2 c + j / 4, register j % 4), so a lane's 8 k values are its own registers of two tiles -- no
//           cross-lane movement; W7's fragments are packed in that k order.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v4 mfma_bf(bf8 a, bf8 b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ bf8 to_bf8(const v4 &lo, const v4 &hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const bf2 p0 = {(__bf16)lo[0], (__bf16)lo[1]}, p1 = {(__bf16)lo[2], (__bf16)lo[3]};     // pair by pair: v_cvt_pk_bf16_f32
    const bf2 p2 = {(__bf16)hi[0], (__bf16)hi[1]}, p3 = {(__bf16)hi[2], (__bf16)hi[3]};
    const u4v u = {__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1), __builtin_bit_cast(unsigned, p2),
                   __builtin_bit_cast(unsigned, p3)};
    return __builtin_bit_cast(bf8, u);
}
__device__ __forceinline__ bf8 frag_bf(const WStream &ws, int idx) {
    return __builtin_bit_cast(bf8, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, idx * 1024, 0));
}
struct XPair { v4 lo, hi; };          // 8 consecutive features of one row (k = 8 g .. 8 g + 7 of a 32-feature chunk)
// Encode.  All loads of a wave retire in order (vmcnt), so a weight fragment fetched "just ahead" would wait for every row chunk
// fetched "far ahead" before it: the first version (fragments half a chunk ahead in registers, rows three chunks ahead) spent one
// HBM round trip per chunk (1.9 us; 216 M frames/s).  Here EVERY load has the same lead: the four waves of a workgroup walk the
// chunks in lockstep and share the chunk's 13 fragments through a double-buffered LDS stage -- each wave fetches its 3-4
// fragments three chunks ahead (16 registers per chunk in flight), stores them one chunk ahead, one barrier per chunk -- and the
// rows are fetched three chunks ahead.  Fragment traffic through the L1 drops four times as well.
// the same through a buffer resource based at the workgroup's first row: one byte offset per lane (row and lane group) in a
// VGPR, the chunk offset in an SGPR -- no 64-bit address arithmetic per load
template <int F, bool FULL = false>
__device__ __forceinline__ XPair wide_x_chunk32_buf(__amdgpu_buffer_rsrc_t rs, int voff, int in_f64, int c, int g) {
    XPair p;
    if (FULL || 32 * c + 32 <= F) {
        if (in_f64) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            d2 q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = __builtin_bit_cast(d2, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * k, c * 256, 0));
            p.lo = (v4){(float)q[0][0], (float)q[0][1], (float)q[1][0], (float)q[1][1]};
            p.hi = (v4){(float)q[2][0], (float)q[2][1], (float)q[3][0], (float)q[3][1]};
        } else {
            p.lo = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, c * 128, 0));
            p.hi = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16, c * 128, 0));
        }
    } else {                      // the last, partial chunk: element by element, zero beyond the row
        const int k0 = 32 * c + 8 * g;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (k0 + j < F) {
                if (in_f64) v = (float)__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 8 * j, c * 256, 0));
                else v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff + 4 * j, c * 128, 0));
            }
            if (j < 4) p.lo[j] = v; else p.hi[j - 4] = v;
        }
    }
    return p;
}

// (IN64 is a template parameter: as a run-time flag every row load sat behind a branch and hipcc joined the paths with
// s_waitcnt vmcnt(0), i.e. no load stayed in flight across a chunk)
template <int F, int Z, bool IN64>
__global__ void __launch_bounds__(256) wide_bf16_encode_kernel(const v4 *packed, const v4 *w0b, const void *__restrict__ xin, int64_t n,
                                                               void *__restrict__ out, int out_f64) {
    constexpr int in_f64 = IN64 ? 1 : 0;
    using N = Net<F, Z>;
    using S = StreamWideEnc<N>;
    constexpr int KB = F / 32, KBT = (F + 31) / 32;      // full chunks (the loop) / all chunks (a remainder chunk is the epilogue)
    __shared__ __attribute__((aligned(16))) v4 bias_lds[N::bf_off(N::L) - N::bf_off(0)];
    __shared__ __attribute__((aligned(16))) v4 wst[2][13][64];       // the chunk's fragments, [slot][tile][lane]
    stage_bias<N>(bias_lds, packed);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    const int64_t ngroup = (n + 127) / 128;                          // 4 waves x 2 tiles x 16 rows
    WStream ws = make_stream(packed + S::start_f4, (N::packed_f4() - S::start_f4) * 16, lane);
    WStream ww = make_stream(w0b, KBT * 13 * 1024, lane);
    for (int64_t grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
        const int64_t r0 = (grp * 4 + wave) * 32 + (lane & 15), r1 = r0 + 16;
        const bool v0 = r0 < n, v1 = r1 < n;
        // rows beyond n read the group's first row (never stored)
        const int es = in_f64 ? 8 : 4;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)xin + (size_t)(grp * 128) * F * es), 0,
                                                                             0x7fffffff, 0x00020000);
        const int lr0 = wave * 32 + (lane & 15);
        const int xo0 = ((v0 ? lr0 : 0) * F + 8 * g) * es, xo1 = ((v1 ? lr0 + 16 : 0) * F + 8 * g) * es;
        asm volatile("" : "+v"(ws.voff), "+v"(ww.voff));
        v4 a1[13], b1[13];
        init_bias(a1, bias_lds, lane);
#pragma unroll
        for (int t = 0; t < 13; ++t) b1[t] = a1[t];
        {
            bf8 wq[2][4];                 // this wave's share (tiles wave, wave + 4, wave + 8, 12 for wave 0) of two chunks in flight
            XPair x0r[3], x1r[3];
            auto wload = [&](bf8 (&w)[4], int c) {
                c = c < KB ? c : KB - 1;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int t = wave + 4 * k;
                    w[k] = frag_bf(ww, c * 13 + (t < 13 ? t : 12));
                }
            };
            auto wstore = [&](const bf8 (&w)[4], int slot) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int t = wave + 4 * k;
                    if (t < 13) wst[slot][t][lane] = __builtin_bit_cast(v4, w[k]);
                }
            };
            auto lx0 = [&](int c) { return wide_x_chunk32_buf<F, true>(xrs, xo0, in_f64, c < KB ? c : 0, g); };
            auto lx1 = [&](int c) { return wide_x_chunk32_buf<F, true>(xrs, xo1, in_f64, c < KB ? c : 0, g); };
            // prologue: chunk 0 staged, chunks 1, 2 in flight; rows of chunks 0..2 in flight
            wload(wq[0], 0);
            wload(wq[1], 1);
#pragma unroll
            for (int u = 0; u < 3; ++u) { x0r[u] = lx0(u); x1r[u] = lx1(u); }
            __syncthreads();              // the previous group's last chunk has been read
            wstore(wq[0], 0);
            wload(wq[0], 2);
            auto iter = [&](int c, auto wsl, auto xsl) {
                constexpr int WS = decltype(wsl)::value, XS = decltype(xsl)::value;      // c % 2, c % 3
                __syncthreads();          // fragments of chunk c visible in stage slot WS; stage slot WS ^ 1 free
                wstore(wq[WS ^ 1], WS ^ 1);                                              // chunk c + 1 (fetched two chunks ago)
                const bf8 q0 = to_bf8(x0r[XS].lo, x0r[XS].hi), q1 = to_bf8(x1r[XS].lo, x1r[XS].hi);
                wload(wq[WS ^ 1], c + 3);
                x0r[XS] = lx0(c + 3);
                x1r[XS] = lx1(c + 3);
                // the chunk's fragments from the stage, four at a time (two register sets: 32 instead of 52 registers -- with
                // all 13 resident the kernel needed 304 registers, one wave per SIMD)
                bf8 wl[2][4];
                auto rd = [&](bf8 (&w)[4], int t0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) w[k] = __builtin_bit_cast(bf8, wst[WS][t0 + k < 13 ? t0 + k : 12][lane]);
                };
                auto mm = [&](const bf8 (&w)[4], int t0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (t0 + k < 13) { a1[t0 + k] = mfma_bf(w[k], q0, a1[t0 + k]); b1[t0 + k] = mfma_bf(w[k], q1, b1[t0 + k]); }
                };
                rd(wl[0], 0);
                rd(wl[1], 4);
                __builtin_amdgcn_sched_barrier(0);
                mm(wl[0], 0);
                rd(wl[0], 8);
                __builtin_amdgcn_sched_barrier(0);
                mm(wl[1], 4);
                rd(wl[1], 12);
                __builtin_amdgcn_sched_barrier(0);
                mm(wl[0], 8);
                mm(wl[1], 12);
                __builtin_amdgcn_sched_barrier(0);
            };
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
            int c = 0;
            for (; c + 6 <= KB; c += 6) {
                iter(c, I0(), I0()); iter(c + 1, I1(), I1()); iter(c + 2, I0(), I2());
                iter(c + 3, I1(), I0()); iter(c + 4, I0(), I1()); iter(c + 5, I1(), I2());
            }
            if (c < KB) iter(c, I0(), I0());
            if (c + 1 < KB) iter(c + 1, I1(), I1());
            if (c + 2 < KB) iter(c + 2, I0(), I2());
            if (c + 3 < KB) iter(c + 3, I1(), I0());
            if (c + 4 < KB) iter(c + 4, I0(), I1());
            if (F % 32 != 0) {            // the remaining F % 32 features: one partial chunk, fragments straight from L2
                const XPair p0 = wide_x_chunk32_buf<F>(xrs, xo0, in_f64, KB, g), p1 = wide_x_chunk32_buf<F>(xrs, xo1, in_f64, KB, g);
                const bf8 q0 = to_bf8(p0.lo, p0.hi), q1 = to_bf8(p1.lo, p1.hi);
#pragma unroll
                for (int t = 0; t < 13; ++t) {
                    const bf8 w = frag_bf(ww, KB * 13 + t);
                    a1[t] = mfma_bf(w, q0, a1[t]);
                    b1[t] = mfma_bf(w, q1, b1[t]);
                }
            }
        }
        // the narrow layers one row tile after the other (the two-tile chain needs 300 registers next to the other tile's
        // accumulators: one wave per SIMD for the whole kernel; they are 4 % of the work)
        lrelu(a1);
        lrelu(b1);
        {
            Ring ring;
            ring_prime<S::total>(ring, ws);
            v4 a2[7], a3[4], a4[tiles(Z)];
            fwd_layer<N, S, 1>(a1, a2, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 2>(a2, a3, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 3>(a3, a4, ring, ws, bias_lds, lane);
            store_rows<Z>(a4, out, out_f64, r0, v0, lane, nullptr, nullptr);
        }
        {
            Ring ring;
            ring_prime<S::total>(ring, ws);
            v4 b2[7], b3[4], b4[tiles(Z)];
            fwd_layer<N, S, 1>(b1, b2, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 2>(b2, b3, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 3>(b3, b4, ring, ws, bias_lds, lane);
            store_rows<Z>(b4, out, out_f64, r1, v1, lane, nullptr, nullptr);
        }
    }
}

// ---- the NARROW layers of a wide model on the bf16 MFMA, two row tiles at once ----------------------------------------------------
// In the bf16 encode / decode of the wide models the six narrow layers were left on the float32 chain "because they are 4 % of the
// work" -- but a float32 MFMA is 16x slower: per 32 rows 2 x 508 v_mfma_f32_16x16x4 = 32.5k cycles, 13.5 us in which a wave does not
// touch its rows (measured with timing-only builds of the encode kernel below: 4.18 -> 5.28 TB/s of rows without that tail).  Here a
// layer is NT output tiles x KBK k blocks of 32 inputs: a C-tile pair (2 c, 2 c + 1) of the previous layer is k block c of this one
// after v_cvt_pk_bf16_f32 (no lane movement, the trick of bf16.hip), the fragments (ImplWideBf16::setup: k slot (g, j) <-> the feature
// of slot (tile 2 c + j / 4, g, j % 4)) come straight from L2, one k block ahead, and feed both row tiles; accumulation and bias in
// float32.  69 / 70 MFMAs of 16 cycles per tile instead of 508 of 32.
template <int KD, int NT>
__device__ __forceinline__ void chain_bf16_pair(const v4 (&in0)[tiles(KD)], const v4 (&in1)[tiles(KD)], v4 (&out0)[NT], v4 (&out1)[NT],
                                                const WStream &wc, int base) {
    constexpr int KT = tiles(KD), KBK = (KT + 1) / 2;
    const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
    bf8 w[2][NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) w[0][t] = frag_bf(wc, base + t);
    // all B operands first: the float32 input tiles are dead from here on (the wide layer's 2 x 13 tiles would not fit next to the
    // fragment buffers otherwise)
    bf8 q0[KBK], q1[KBK];
#pragma unroll
    for (int c = 0; c < KBK; ++c) {
        q0[c] = to_bf8(in0[2 * c], 2 * c + 1 < KT ? in0[2 * c + 1 < KT ? 2 * c + 1 : 0] : zero);
        q1[c] = to_bf8(in1[2 * c], 2 * c + 1 < KT ? in1[2 * c + 1 < KT ? 2 * c + 1 : 0] : zero);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < KBK; ++c) {
        if (c + 1 < KBK) {
#pragma unroll
            for (int t = 0; t < NT; ++t) w[(c + 1) & 1][t] = frag_bf(wc, base + (c + 1) * NT + t);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) { out0[t] = mfma_bf(w[c & 1][t], q0[c], out0[t]); out1[t] = mfma_bf(w[c & 1][t], q1[c], out1[t]); }
        __builtin_amdgcn_sched_barrier(0);
    }
}
// one narrow layer l of the pair: bias (float32, from LDS) -> bf16 product -> LeakyReLU;  CB = first layer of the chain buffer `wc`
template <class N, int l, int CB>
__device__ __forceinline__ void fwd_layer_bf16_pair(const v4 (&in0)[tiles(N::dim(l))], const v4 (&in1)[tiles(N::dim(l))],
                                                    v4 (&out0)[tiles(N::dim(l + 1))], v4 (&out1)[tiles(N::dim(l + 1))], const WStream &wc,
                                                    const v4 *bias_l, int lane) {
    constexpr int base = [] { int b = 0; for (int j = CB; j < l; ++j) b += ((tiles(N::dim(j)) + 1) / 2) * tiles(N::dim(j + 1)); return b; }();
    init_bias(out0, bias_l, lane);
#pragma unroll
    for (int t = 0; t < tiles(N::dim(l + 1)); ++t) out1[t] = out0[t];
    chain_bf16_pair<N::dim(l), tiles(N::dim(l + 1))>(in0, in1, out0, out1, wc, base);
    if (N::act(l)) { lrelu(out0); lrelu(out1); }
}
template <class N, int CB> constexpr int chain_bf16_frags() {
    int b = 0;
    for (int j = CB; j < CB + 3; ++j) b += ((tiles(N::dim(j)) + 1) / 2) * tiles(N::dim(j + 1));
    return b;
}

// ---- bf16 encode of the wide models with the ROW STREAM DECOUPLED from the compute waves (float32 rows whose length is a multiple
// of 16 bytes: C4's 2500 and C5's 512 columns) -------------------------------------------------------------------------------------
// The kernel above is HBM-bound on paper (10 KB of float32 per C4 frame against 26 bf16 MFMAs per 4 KB) and ran at 0.40 of the HBM
// roof: its rows travel through REGISTERS, three chunks (12 KB) ahead per wave at 300 registers and one wave per SIMD, i.e. 48 KB in
// flight per CU -- under Little's law for 8 TB/s x 2-3 us.  Here two LOADER waves (waves 4, 5; 384-thread workgroups) do nothing but
// stream the workgroup's 128 rows into a 6-slot LDS ring, and the 13 weight fragments of every chunk into a 4-slot stage, with
// direct-to-LDS buffer loads (buffer_load_dwordx4 ... lds: no VGPRs, 3 chunks = 87 KB in flight per CU; a wave can have at most 63
// loads outstanding, hence two loaders with 45 each).  Their vector-memory queue holds NOTHING ELSE, so the lead is theirs to keep
// (loads of a wave retire in order: in the compute waves of the kernel above every fragment wait also waited for the rows
// requested before it); the compute waves issue no vector-memory instruction in the chunk loop at all -- operands A and B both
// come from LDS.  (The DMA is inline asm in the loader branch only; the waits there are hand-counted.)
//   What bounds it (timing-only ablation builds, profiles/r4_c4_bf16_encode_ablation.txt): the CU's vector-memory path.  The row
//   stream alone runs at 5.3 TB/s; the 13 KiB of L2-resident fragments per 16 KiB of rows cost it ~0.45 bytes each (-> 3.9-4.0);
//   who fetches them (compute waves through registers, loaders by DMA: 3.89 -> 3.95 TB/s) and how far ahead (lead 2 .. 7) do not
//   matter.  More rows per fragment byte needs 64 rows = 208 accumulator registers per wave, i.e. one wave per SIMD issuing its
//   own DMA: built (256-row groups, 32-KiB chunks) and measured at 3.65-3.78 TB/s -- half the fragment bytes, but the row stream
//   of four self-serving waves is slower (4.45 TB/s without any fragments) than that of dedicated loaders.  Not kept.
//   * ring slot = [compute wave][row tile][half][64 lanes x 16 B]: a 1-KiB block is one DMA instruction, whose lane (i, g) fetches
//     bytes 64 h + 16 g .. + 15 of the chunk of row i -- 64 contiguous bytes per row and instruction -- and it is read back with one
//     linear, conflict-free ds_read_b128 by the same lane of the compute wave: that lane then holds k slots (g, j) <-> features
//     16 (j >> 2) + 4 g + (j & 3) of the chunk, the order the fragments `w0p` are packed in (ImplWideBf16::setup);
//   * per chunk ONE workgroup barrier: a loader waits (counted vmcnt, asm) until its share of chunk c has landed, joins barrier c,
//     then issues chunk c + 3 into the ring slot of chunk c - 3 and the stage slot of chunk c - 1, which every compute wave
//     finished reading before it joined barrier c.  The loaders run ahead across row groups: no bubble at a group's start.
constexpr int kDmaRing = 6;             // row ring: 6 x 16 KiB
constexpr int kDmaStage = 4;            // fragment stage: 4 x 13 KiB (ring + stage + biases = 150 KiB of the 160)
constexpr int kDmaLead = 3;             // chunks in flight per loader wave: 3 x 15 loads (a wave counts at most 63)
constexpr int kDmaChunk = 16384;        // 128 rows x 32 float32 features
__device__ __forceinline__ void lds_dma_b128(unsigned lds_addr, int voff, __amdgpu_buffer_rsrc_t rs, int soff) {
    unsigned keep;      // M0 (the LDS base of a direct-to-LDS load) is not preserved by hipcc around asm: set and restore it here
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int F, int Z>
__global__ void __launch_bounds__(384) wide_bf16_encode_dma_kernel(const v4 *packed, const v4 *w0p, const v4 *wce, const float *__restrict__ xin, int64_t n,
                                                                   void *__restrict__ out, int out_f64) {
    using N = Net<F, Z>;
    using S = StreamWideEnc<N>;
    constexpr int KB = F / 32, KBT = (F + 31) / 32;
    static_assert((F * 4) % 16 == 0 && KB >= kDmaRing, "16-byte pieces of float32 rows; a row group fills the ring");
    extern __shared__ __attribute__((aligned(1024))) unsigned char dma_lds[];
    unsigned char *const ring_b = dma_lds;
    v4 (*const wst)[13][64] = (v4 (*)[13][64])(dma_lds + kDmaRing * kDmaChunk);
    v4 *const bias_lds = (v4 *)(dma_lds + kDmaRing * kDmaChunk + kDmaStage * 13 * 1024);
    constexpr int nb = N::bf_off(4) - N::bf_off(0);                      // biases of layers 0..3
    for (int i = threadIdx.x; i < nb; i += 384) bias_lds[i] = packed[N::bf_off(0) + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    const int64_t ngroup = (n + 127) / 128;
    if (wave >= 4) {
        // ---- loader: rows 64 L .. 64 L + 63 of every group of this workgroup, chunk after chunk, group after group ------------
        const int L = wave - 4;
        const unsigned ring0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)ring_b + (unsigned)L * 8192u;
        const unsigned stage0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)ring_b + (unsigned)(kDmaRing * kDmaChunk);
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)w0p, 0, KBT * 13 * 1024, 0x00020000);
        int64_t gi = blockIdx.x;            // issue cursor: group, chunk, ring slot, stage slot
        int ci = 0, pi = 0, si = 0;
        auto issue = [&]() {
            const int64_t gg = gi < ngroup ? gi : ngroup - 1;            // past the end: harmless re-reads keep the DMA count uniform
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(xin + (size_t)gg * 128 * F), 0, 0x7fffffff, 0x00020000);
            const int soff = ci * 128;
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // (compute wave 2 L + (q >> 1), row tile q & 1)
                const int rl = 64 * L + 16 * q + (lane & 15);
                const int voff = (gg * 128 + rl < n ? rl : 0) * (F * 4) + 16 * g;       // rows beyond n read the group's first row
                const unsigned dst = __builtin_amdgcn_readfirstlane(ring0 + (unsigned)pi * kDmaChunk + (unsigned)q * 2048u);
                lds_dma_b128(dst, voff, rs, soff);
                lds_dma_b128(dst + 1024u, voff, rs, soff + 64);
            }
            // the chunk's 13 weight fragments (L2-resident, 1 KiB each) into the stage: tiles 0..6 by loader 0, 7..12 by loader 1
            // (which fetches tile 12 twice: both loaders count 15 loads per chunk)
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const int t = 7 * L + k < 13 ? 7 * L + k : 12;
                const unsigned dst = __builtin_amdgcn_readfirstlane(stage0 + (unsigned)(si * 13 + t) * 1024u);
                lds_dma_b128(dst, lane * 16, wrs, (ci * 13 + t) * 1024);
            }
            if (++ci == KB) { ci = 0; gi += gridDim.x; }
            pi = pi + 1 == kDmaRing ? 0 : pi + 1;
            si = (si + 1) & (kDmaStage - 1);
        };
        for (int k = 0; k < kDmaLead; ++k) issue();
        for (int64_t grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
            for (int c = 0; c < KB; ++c) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "i"(15 * (kDmaLead - 1)) : "memory");    // all but the youngest chunks: chunk c is in LDS
                __builtin_amdgcn_s_barrier();                            // barrier c: the compute waves have finished chunk c - 1
                issue();                                                 // chunk c + 3: ring slot of chunk c - 3, stage slot of chunk c - 1
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // nothing may land in LDS after the workgroup has gone
        return;
    }
    // ---- compute waves: wave w owns rows 32 w .. 32 w + 31 of the group (two 16-row tiles) -------------------------------------
    WStream ww = make_stream(w0p, KBT * 13 * 1024, lane);
    WStream wc = make_stream(wce, chain_bf16_frags<N, 1>() * 1024, lane);
    int slot = 0, ws = 0;
    for (int64_t grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
        const int64_t r0 = (grp * 4 + wave) * 32 + (lane & 15), r1 = r0 + 16;
        const bool v0 = r0 < n, v1 = r1 < n;
        asm volatile("" : "+v"(wc.voff), "+v"(ww.voff));
        v4 a1[13], b1[13];
        init_bias(a1, bias_lds, lane);
#pragma unroll
        for (int t = 0; t < 13; ++t) b1[t] = a1[t];
        {
            for (int c = 0; c < KB; ++c) {
                __syncthreads();          // barrier c: rows of chunk c in ring slot `slot`, its fragments in stage slot `ws`
                const v4 *xs = (const v4 *)(ring_b + slot * kDmaChunk + wave * 4096) + lane;
                const v4 (*const wf)[64] = wst[ws];
                const v4 l0 = xs[0], h0 = xs[64], l1 = xs[128], h1 = xs[192];
                slot = slot + 1 == kDmaRing ? 0 : slot + 1;
                ws = (ws + 1) & (kDmaStage - 1);
                const bf8 q0 = to_bf8(l0, h0), q1 = to_bf8(l1, h1);
                bf8 wl[2][4];
                auto rd = [&](bf8 (&w)[4], int t0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) w[k] = __builtin_bit_cast(bf8, wf[t0 + k < 13 ? t0 + k : 12][lane]);
                };
                auto mm = [&](const bf8 (&w)[4], int t0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (t0 + k < 13) { a1[t0 + k] = mfma_bf(w[k], q0, a1[t0 + k]); b1[t0 + k] = mfma_bf(w[k], q1, b1[t0 + k]); }
                };
                rd(wl[0], 0);
                rd(wl[1], 4);
                __builtin_amdgcn_sched_barrier(0);
                mm(wl[0], 0);
                rd(wl[0], 8);
                __builtin_amdgcn_sched_barrier(0);
                mm(wl[1], 4);
                rd(wl[1], 12);
                __builtin_amdgcn_sched_barrier(0);
                mm(wl[0], 8);
                mm(wl[1], 12);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (F % 32 != 0) {            // the remaining F % 32 features: one partial chunk, rows and fragments straight from L2 (natural k order)
                const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)(xin + (size_t)(grp * 128) * F), 0, 0x7fffffff, 0x00020000);
                const int lr0 = wave * 32 + (lane & 15);
                const int xo0 = ((v0 ? lr0 : 0) * F + 8 * g) * 4, xo1 = ((v1 ? lr0 + 16 : 0) * F + 8 * g) * 4;
                const XPair p0 = wide_x_chunk32_buf<F>(xrs, xo0, 0, KB, g), p1 = wide_x_chunk32_buf<F>(xrs, xo1, 0, KB, g);
                const bf8 q0 = to_bf8(p0.lo, p0.hi), q1 = to_bf8(p1.lo, p1.hi);
#pragma unroll
                for (int t = 0; t < 13; ++t) {
                    const bf8 w = frag_bf(ww, KB * 13 + t);
                    a1[t] = mfma_bf(w, q0, a1[t]);
                    b1[t] = mfma_bf(w, q1, b1[t]);
                }
            }
        }
        lrelu(a1);
        lrelu(b1);
        {   // the narrow layers on the bf16 MFMA, both row tiles at once (chain_bf16_pair)
            v4 a2[7], b2[7], a3[4], b3[4], a4[tiles(Z)], b4[tiles(Z)];
            fwd_layer_bf16_pair<N, 1, 1>(a1, b1, a2, b2, wc, bias_lds + (N::bf_off(1) - N::bf_off(0)), lane);
            fwd_layer_bf16_pair<N, 2, 1>(a2, b2, a3, b3, wc, bias_lds + (N::bf_off(2) - N::bf_off(0)), lane);
            fwd_layer_bf16_pair<N, 3, 1>(a3, b3, a4, b4, wc, bias_lds + (N::bf_off(3) - N::bf_off(0)), lane);
            store_rows<Z>(a4, out, out_f64, r0, v0, lane, nullptr, nullptr);
            store_rows<Z>(b4, out, out_f64, r1, v1, lane, nullptr, nullptr);
        }
    }
}

// Decode.  Two things bound the round-3 kernel (200-236 M frames/s, 0.25-0.30 of HBM; 532 M without its stores): every wave fetched
// all 7 fragments of an output tile for itself, and -- what mattered -- those loads sat in the same in-order queue as the wave's
// stores: vector-memory operations of a wave retire in order (vmcnt counts loads AND stores), so every wait for a fragment also
// waited for the 1-KiB stores issued before it, i.e. for an HBM write acknowledgement (~4 us under load) every four tiles.  Here a
// FIFTH wave does all the fragment traffic: it streams the tiles' fragments (7 KiB each, L2-resident) into a 6-slot LDS stage with
// direct-to-LDS loads, five tiles ahead, one workgroup barrier per tile; the four compute waves read their A operands from the stage
// and their vector-memory queue holds NOTHING BUT STORES, which they never wait for.
constexpr int kDecSlots = 6;             // stage slots: 5 tiles of fragments (35 KiB) in flight; 3 left the loader latency-bound (4 slots: 0.35-0.39 of HBM at
                                         // 131,072 frames, 6 / 8 / 10 slots: 0.40-0.44)
template <int F, int Z, bool OUT64>
__global__ void __launch_bounds__(320) wide_bf16_decode_kernel(const v4 *packed, const v4 *w7b, const v4 *wcd, const void *__restrict__ zin, int in_f64,
                                                               int64_t n, void *__restrict__ out) {
    constexpr int out_f64 = OUT64 ? 1 : 0;
    using N = Net<F, Z>;
    constexpr int KT = tiles(F);                   // output tiles
    // per wave: kTG output tiles of its 32 rows, row-major, rows 16 bytes longer than the data (the C-layout writes of 16 rows
    // would otherwise all fall on the same banks) -- the transposing stage of the contiguous row segments below
    constexpr int kTG = 4, kTS = 16 * kTG + 4;     // 256-byte segments (512-byte ones measured the same and cost 68 KB of LDS)
    constexpr int nb = N::bf_off(N::L) - N::bf_off(4);                                     // biases of layers 4..7
    extern __shared__ __attribute__((aligned(1024))) unsigned char dec_lds[];
    v4 (*const wst)[7][64] = (v4 (*)[7][64])dec_lds;                                       // [kDecSlots][7][64]
    constexpr bool kAligned = !OUT64 && F % 4 == 0;                    // float32 rows of a multiple of 16 bytes: line-aligned window stores
    constexpr int kFullTiles = F / 16, kRS = 128 + 4;                  // ring: 128 columns per row (+ 16 bytes: bank spread of the C-layout writes)
    float (*const tstage)[32][kTS] = (float (*)[32][kTS])(dec_lds + kDecSlots * 7 * 1024);  // [4][32][kTS]   (!kAligned)
    float (*const tring)[32][kRS] = (float (*)[32][kRS])(dec_lds + kDecSlots * 7 * 1024);   // [4][32][kRS]   (kAligned)
    v4 *const bias_lds = (v4 *)(dec_lds + kDecSlots * 7 * 1024 + 4 * 32 * kRS * 4);
    for (int i = threadIdx.x; i < nb; i += 320) bias_lds[i] = packed[N::bf_off(4) + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    const int64_t ngroup = (n + 127) / 128;
    if (wave == 4) {
        // ---- loader: tile after tile, group after group: 7 direct-to-LDS loads per tile, two tiles in flight ----------------------
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)w7b, 0, KT * 7 * 1024, 0x00020000);
        const unsigned st0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)dec_lds;
        int ti = 0, si = 0;                        // issue cursor: tile, stage slot
        auto issue = [&]() {
#pragma unroll
            for (int c = 0; c < 7; ++c)
                lds_dma_b128(__builtin_amdgcn_readfirstlane(st0 + (unsigned)(si * 7 + c) * 1024u), lane * 16, rs, (ti * 7 + c) * 1024);
            if (++ti == KT) ti = 0;
            si = si + 1 == kDecSlots ? 0 : si + 1;
        };
        for (int k = 0; k < kDecSlots - 1; ++k) issue();
        for (int64_t grp = blockIdx.x; grp < ngroup; grp += gridDim.x)
            for (int t = 0; t < KT; ++t) {
                asm volatile("s_waitcnt vmcnt(%0)" :: "i"(7 * (kDecSlots - 2)) : "memory");      // all but the youngest tiles: tile t is in the stage
                __builtin_amdgcn_s_barrier();                            // barrier t
                issue();                                                 // tile t + 5 into the slot of tile t - 1
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    WStream wc = make_stream(wcd, chain_bf16_frags<N, 4>() * 1024, lane);
    const v4 *bias7 = bias_lds + (N::bf_off(7) - N::bf_off(4));
    int slot = 0;
    for (int64_t grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
        const int64_t pr = grp * 4 + wave;
        const int64_t r0 = pr * 32 + (lane & 15), r1 = r0 + 16;
        const bool v0 = r0 < n, v1 = r1 < n;
        asm volatile("" : "+v"(wc.voff));
        bf8 qa[7], qb[7];                          // a7 of both row tiles as bf16 B operands (k chunk c = tiles 2 c, 2 c + 1)
        {   // the narrow layers on the bf16 MFMA (chain_bf16_pair: 70 MFMAs of 16 cycles per tile instead of 508 of 32 on the float32 chain)
            v4 a4[tiles(Z)], b4[tiles(Z)], a5[4], b5[4], a6[7], b6[7], a7[13], b7[13];
            load_rows<Z>(a4, zin, in_f64, r0, v0, lane, nullptr);
            load_rows<Z>(b4, zin, in_f64, r1, v1, lane, nullptr);
            fwd_layer_bf16_pair<N, 4, 4>(a4, b4, a5, b5, wc, bias_lds, lane);
            fwd_layer_bf16_pair<N, 5, 4>(a5, b5, a6, b6, wc, bias_lds + (N::bf_off(5) - N::bf_off(4)), lane);
            fwd_layer_bf16_pair<N, 6, 4>(a6, b6, a7, b7, wc, bias_lds + (N::bf_off(6) - N::bf_off(4)), lane);
            const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                qa[c] = to_bf8(a7[2 * c], 2 * c + 1 < 13 ? a7[2 * c + 1 < 13 ? 2 * c + 1 : 12] : zero);
                qb[c] = to_bf8(b7[2 * c], 2 * c + 1 < 13 ? b7[2 * c + 1 < 13 ? 2 * c + 1 : 12] : zero);
            }
        }
        // Output: a lane holds 16 bytes of a row per tile, i.e. the wave wrote 32 rows x 64 bytes per tile -- half cache lines
        // that only the L2 could merge (180 M frames/s; non-temporal: 95 M).  Float32 output goes through
        // the stage instead: four tiles are collected per row, then every store instruction writes 4 rows x 256 contiguous bytes.
        auto tile = [&](int t, auto jj) {
            constexpr int J = decltype(jj)::value % kTG;           // t % kTG
            if (t >= KT) return;                                   // (the same in all waves)
            __builtin_amdgcn_s_barrier();                          // barrier t: the loader has tile t in stage slot `slot`
            bf8 w[7];
#pragma unroll
            for (int c = 0; c < 7; ++c) w[c] = __builtin_bit_cast(bf8, wst[slot][c][lane]);
            slot = slot + 1 == kDecSlots ? 0 : slot + 1;
            v4 o0 = bias7[t * 4 + g], o1 = o0;
#pragma unroll
            for (int c = 0; c < 7; ++c) { o0 = mfma_bf(w[c], qa[c], o0); o1 = mfma_bf(w[c], qb[c], o1); }
            if constexpr (kAligned) {
                // LINE-ALIGNED stores.  A C4 row is 10,000 bytes: rows start at every 16-byte phase of a 128-byte line, and a store
                // of row segments at fixed columns leaves every segment's first and last line partly written -- written back and
                // touched again by the next segment (tools/probe/hbm_store_pattern_probe: 3.56 TB/s for 256-byte segments of
                // 10,000-byte rows against 5.5 TB/s for rows of a multiple of 128 bytes).  So each row stores 256-byte windows that
                // start on ITS OWN line boundaries: column a_r + 64 k with a_r = (-row * F) mod 32 floats.  The last eight tiles
                // of the wave's 32 rows live in a ring (128 columns per row); window k is complete once tile 4 k + 5 is in.
                if (t < kFullTiles) {
                    *(v4 *)&tring[wave][lane & 15][16 * (t & 7) + 4 * g] = o0;
                    *(v4 *)&tring[wave][16 + (lane & 15)][16 * (t & 7) + 4 * g] = o1;
                } else {                                   // the partial last tile: element by element
                    if (v0) wide_store_tile<F>(o0, out, out_f64, r0, t, g);
                    if (v1) wide_store_tile<F>(o1, out, out_f64, r1, t, g);
                }
                const int64_t rb = pr * 32;
                const int p = lane & 15;
                auto window = [&](int col0 /* first column of the window of a row with a_r = 0 */, bool head, bool tail) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int rl = 4 * k + (lane >> 4);
                        const int64_t R = rb + rl;
                        const int ar = (32 - (int)((R * (F % 32)) & 31)) & 31;
                        int col = head ? 4 * p : col0 + ar + 4 * p;
                        const bool ok = R < n && (head ? 4 * p < ar : (!tail || col < 16 * kFullTiles));
                        col = ok ? col : 0;
                        const v4 v = *(const v4 *)&tring[wave][rl][col & 127];
                        if (ok) *(v4 *)((float *)out + R * F + col) = v;
                    }
                };
                if (J == 1 && t >= 5 && t < kFullTiles) {
                    if (t == 5) window(0, true, false);                   // the columns in front of the row's first line boundary
                    window(16 * (t - 5), false, false);
                }
                if (t == kFullTiles - 1) {                                // what is left behind the last complete window: <= 2 masked rounds
                    constexpr int tl = kFullTiles - 1 - ((kFullTiles - 1 - 5) % 4 + 4) % 4;      // the last t that stored a window (t % 4 == 1, t >= 5)
                    constexpr int c0 = kFullTiles > 5 ? 16 * (tl - 5) + 64 : 0;
                    if (kFullTiles <= 5) window(0, true, false);
                    window(c0, false, true);
                    window(c0 + 64, false, true);
                }
            } else if (!OUT64 && t + (kTG - 1 - J) < F / 16) {            // the whole group is made of full tiles
                *(v4 *)&tstage[wave][lane & 15][16 * J + 4 * g] = o0;
                *(v4 *)&tstage[wave][16 + (lane & 15)][16 * J + 4 * g] = o1;
                if (J == kTG - 1) {
                    const int64_t rb = pr * 32;
                    constexpr int LR = 4 * kTG;                     // lanes per row segment (16 bytes each)
#pragma unroll
                    for (int k = 0; k < 32 * LR / 64; ++k) {
                        const int rl = (64 / LR) * k + lane / LR;   // row of the wave's 32
                        const v4 v = *(const v4 *)&tstage[wave][rl][4 * (lane % LR)];
                        if (rb + rl < n) *(v4 *)((float *)out + (rb + rl) * F + 16 * (t - (kTG - 1)) + 4 * (lane % LR)) = v;
                    }
                }
            } else {
                if (v0) wide_store_tile<F>(o0, out, out_f64, r0, t, g);
                if (v1) wide_store_tile<F>(o1, out, out_f64, r1, t, g);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        for (int t0 = 0; t0 < KT; t0 += 4) {
            tile(t0, std::integral_constant<int, 0>());
            tile(t0 + 1, std::integral_constant<int, 1>());
            tile(t0 + 2, std::integral_constant<int, 2>());
            tile(t0 + 3, std::integral_constant<int, 3>());
        }
    }
}

// params (fp32) -> bf16 fragments through an index map (-1: zero)
__global__ void __launch_bounds__(256) pack_wide_bf16_k(const float *__restrict__ params, const int *__restrict__ src, int count,
                                                        __bf16 *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = (__bf16)(src[i] >= 0 ? params[src[i]] : 0.f);
}

// The streamed wide -> 200 product with the chunk's 13 fragments SHARED by the four waves of the workgroup through a
// double-buffered LDS stage (see wide_bf16_encode_kernel: every load of a wave then has the same lead -- fragments and rows three
// chunks ahead -- so the in-order vmcnt never makes an L2 hit wait for an HBM miss; fragment traffic through the L1 drops 4x;
// 32 + 32 fragment registers instead of 104).  All four waves must call it together (one barrier per chunk).
// [kc0, kc1) (kc1 < 0: all): a RANGE of the full chunks only -- the small-batch training pass splits the wide dimension of one row
// group over several workgroups (wide_small_in_kernel); `rem`: this call also takes the partial last chunk.
template <int F, bool IN64, int RT = 1, bool WRT = false>
__device__ __forceinline__ void wide_in_product_lds(v4 (&acc)[13], v4 (&acc1)[13], v4 (*wst)[13][64], const WStream &ww, const void *xin,
                                                    int64_t rrow, int64_t rrow1, int g, int lane, int wave, int fr = F, int kc0 = 0, int kc1 = -1,
                                                    bool rem = true) {
    const int KCA = (WRT ? fr : F) / 16;  // FULL chunks of the model: the loop; the remainder is the epilogue below
    const int KC = kc1 < 0 ? KCA : kc1;   // end of this call's range
    static_assert(F / 16 >= 3, "at least three full chunks");
    v4 wq[2][4], xr[3], xs[3];            // xs / acc1: the second row tile (RT == 2)
    auto wload = [&](v4 (&w)[4], int kc) {
        kc = kc < KC ? kc : KC - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = wave + 4 * k;
            w[k] = frag_rt(ww, kc * 13 + (t < 13 ? t : 12));
        }
    };
    auto wstore = [&](const v4 (&w)[4], int slot) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = wave + 4 * k;
            if (t < 13) wst[slot][t][lane] = w[k];
        }
    };
    auto lx = [&](int kc) { return wide_x_chunk<F, true, WRT>(xin, IN64 ? 1 : 0, rrow, kc < KC ? kc : 0, g, fr); };
    auto ly = [&](int kc) { return RT == 2 ? wide_x_chunk<F, true, WRT>(xin, IN64 ? 1 : 0, rrow1, kc < KC ? kc : 0, g, fr) : (v4){0.f, 0.f, 0.f, 0.f}; };
    wload(wq[0], kc0);
    wload(wq[1], kc0 + 1);
#pragma unroll
    for (int u = 0; u < 3; ++u) { xr[u] = lx(kc0 + u); xs[u] = ly(kc0 + u); }
    __syncthreads();                      // the previous row group's last chunk has been read
    wstore(wq[0], 0);
    wload(wq[0], kc0 + 2);
    auto iter = [&](int kc, auto wsl, auto xsl) {
        constexpr int WS = decltype(wsl)::value, XS = decltype(xsl)::value;      // (kc - kc0) % 2, (kc - kc0) % 3
        __syncthreads();                  // fragments of chunk kc visible in stage slot WS; slot WS ^ 1 free
        wstore(wq[WS ^ 1], WS ^ 1);       // chunk kc + 1 (fetched two chunks ago)
        const v4 xv = xr[XS], yv = xs[XS];
        wload(wq[WS ^ 1], kc + 3);
        xr[XS] = lx(kc + 3);
        xs[XS] = ly(kc + 3);
        v4 wl[2][4];
        auto rd = [&](v4 (&w)[4], int t0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = wst[WS][t0 + k < 13 ? t0 + k : 12][lane];
        };
        auto mm = [&](const v4 (&w)[4], int t0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (t0 + k < 13) {
                        acc[t0 + k] = mfma(w[k][r], xv[r], acc[t0 + k]);
                        if (RT == 2) acc1[t0 + k] = mfma(w[k][r], yv[r], acc1[t0 + k]);
                    }
            }
        };
        rd(wl[0], 0);
        rd(wl[1], 4);
        __builtin_amdgcn_sched_barrier(0);
        mm(wl[0], 0);
        rd(wl[0], 8);
        __builtin_amdgcn_sched_barrier(0);
        mm(wl[1], 4);
        rd(wl[1], 12);
        __builtin_amdgcn_sched_barrier(0);
        mm(wl[0], 8);
        mm(wl[1], 12);
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    int kc = kc0;
    for (; kc + 6 <= KC; kc += 6) {
        iter(kc, I0(), I0()); iter(kc + 1, I1(), I1()); iter(kc + 2, I0(), I2());
        iter(kc + 3, I1(), I0()); iter(kc + 4, I0(), I1()); iter(kc + 5, I1(), I2());
    }
    if (kc < KC) iter(kc, I0(), I0());
    if (kc + 1 < KC) iter(kc + 1, I1(), I1());
    if (kc + 2 < KC) iter(kc + 2, I0(), I2());
    if (kc + 3 < KC) iter(kc + 3, I1(), I0());
    if (kc + 4 < KC) iter(kc + 4, I0(), I1());
    if (rem && (WRT ? fr : F) % 16 != 0) {       // the remaining features: one partial chunk, fragments straight from L2
        const int KL = KCA;
        constexpr int ST = WRT ? 4 : tile_steps(F, F / 16);      // (a class chunk is a full tile: padding meets zero weights and a zero x)
        const v4 xv = wide_x_chunk<F, false, WRT>(xin, IN64 ? 1 : 0, rrow, KL, g, fr);
        const v4 yv = RT == 2 ? wide_x_chunk<F, false, WRT>(xin, IN64 ? 1 : 0, rrow1, KL, g, fr) : xv;
        v4 wt[13];
#pragma unroll
        for (int t = 0; t < 13; ++t) wt[t] = frag_rt(ww, KL * 13 + t);
#pragma unroll
        for (int r = 0; r < ST; ++r)
#pragma unroll
            for (int t = 0; t < 13; ++t) {
                acc[t] = mfma(wt[t][r], xv[r], acc[t]);
                if (RT == 2) acc1[t] = mfma(wt[t][r], yv[r], acc1[t]);
            }
    }
}

// The streamed 200 -> wide product (de4) the same way: the 13 fragments of an output tile shared through the LDS stage, fetched
// three tiles ahead; `pre(t, slot, full)` issues the caller's own loads for tile t (three tiles ahead as well, slot = t % 3),
// `emit(o, t, slot, full)` consumes output tile t; full = std::true_type in the main loop, where every tile touched is a full one
// (no "partial tile?" test in front of the loads and stores), std::false_type for the last tiles.  All four waves together; one
// barrier per tile.
// [t0, t1) (t1 < 0: all): a RANGE of the output tiles -- the small-batch training pass deals them to several workgroups per row group
// (wide_small_out_kernel).
template <int F, bool WRT = false, class Pre, class Emit>
__device__ __forceinline__ void wide_out_product_lds(const v4 (&a7)[13], v4 (*wst)[13][64], const WStream &ww, const v4 *bias7, int g,
                                                     int lane, int wave, Pre pre, Emit emit, int fr = F, int t0 = 0, int t1 = -1) {
    constexpr int KCS = tiles(F);                  // stride of the fragment array [k tile of the 200 side][output tile]
    const int KCA = WRT ? (fr + 15) / 16 : tiles(F), KFA = (WRT ? fr : F) / 16;      // all / full output tiles of the model
    const int KC = t1 < 0 || t1 > KCA ? KCA : t1, KF = KFA < KC ? KFA : KC;          // ... of this call's range
    static_assert(F / 16 >= 3, "at least three full tiles");
    v4 wq[2][4];
    auto wload = [&](v4 (&w)[4], int t) {
        t = t < KC ? t : KC - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = wave + 4 * k;
            w[k] = frag_rt(ww, (q < 13 ? q : 12) * KCS + t);
        }
    };
    auto wstore = [&](const v4 (&w)[4], int slot) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = wave + 4 * k;
            if (q < 13) wst[slot][q][lane] = w[k];
        }
    };
    wload(wq[0], t0);
    wload(wq[1], t0 + 1);
    if (t0 + 3 <= KF) {
        pre(t0, std::integral_constant<int, 0>(), std::true_type());
        pre(t0 + 1, std::integral_constant<int, 1>(), std::true_type());
        pre(t0 + 2, std::integral_constant<int, 2>(), std::true_type());
    } else {                                       // (a short range at the end of the row: the general path)
        pre(t0, std::integral_constant<int, 0>(), std::false_type());
        pre(t0 + 1, std::integral_constant<int, 1>(), std::false_type());
        pre(t0 + 2, std::integral_constant<int, 2>(), std::false_type());
    }
    __syncthreads();
    wstore(wq[0], 0);
    wload(wq[0], t0 + 2);
    auto iter = [&](int t, auto wsl, auto xsl, auto full) {
        constexpr int WS = decltype(wsl)::value;
        __syncthreads();
        wstore(wq[WS ^ 1], WS ^ 1);
        wload(wq[WS ^ 1], t + 3);
        v4 o0 = bias7[t * 4 + g], o1 = (v4){0.f, 0.f, 0.f, 0.f};
        v4 wl[2][4];
        auto rd = [&](v4 (&w)[4], int q0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = wst[WS][q0 + k < 13 ? q0 + k : 12][lane];
        };
        auto mm = [&](const v4 (&w)[4], int q0) {       // k tiles alternate between two accumulators (dependent MFMAs: 40 cycles)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (q0 + k < 13 && r < tile_steps(200, q0 + k)) {
                        if (k & 1) o1 = mfma(w[k][r], a7[q0 + k][r], o1); else o0 = mfma(w[k][r], a7[q0 + k][r], o0);
                    }
        };
        rd(wl[0], 0);
        rd(wl[1], 4);
        __builtin_amdgcn_sched_barrier(0);
        mm(wl[0], 0);
        rd(wl[0], 8);
        __builtin_amdgcn_sched_barrier(0);
        mm(wl[1], 4);
        rd(wl[1], 12);
        __builtin_amdgcn_sched_barrier(0);
        mm(wl[0], 8);
        mm(wl[1], 12);
        emit(o0 + o1, t, xsl, full);
        pre(t + 3, xsl, full);                          // into the slot just consumed
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    using TT = std::true_type;
    using FT = std::false_type;
    int t = t0;
    for (; t + 9 <= KF; t += 6) {                       // tiles t .. t + 5 and the prefetched t + 3 .. t + 8 are all full
        iter(t, I0(), I0(), TT()); iter(t + 1, I1(), I1(), TT()); iter(t + 2, I0(), I2(), TT());
        iter(t + 3, I1(), I0(), TT()); iter(t + 4, I0(), I1(), TT()); iter(t + 5, I1(), I2(), TT());
    }
    // the last <= 14 tiles (the partial one among them) on the general path; slots keep following t % 2 / t % 3
    auto tail = [&](int k, auto wsl, auto xsl) { if (t + k < KC) iter(t + k, wsl, xsl, FT()); };
    tail(0, I0(), I0()); tail(1, I1(), I1()); tail(2, I0(), I2()); tail(3, I1(), I0()); tail(4, I0(), I1()); tail(5, I1(), I2());
    tail(6, I0(), I0()); tail(7, I1(), I1()); tail(8, I0(), I2()); tail(9, I1(), I0()); tail(10, I0(), I1()); tail(11, I1(), I2());
    tail(12, I0(), I0()); tail(13, I1(), I1());
}

template <int F, int Z, bool IN64, bool WRT = false>
__global__ void __launch_bounds__(256) wide_encode_lds_kernel(const v4 *packed, const void *__restrict__ xin, int64_t n,
                                                              void *__restrict__ out, int out_f64, int fr = F, int zr = Z) {
    using N = Net<F, Z>;
    using S = StreamWideEnc<N>;
    __shared__ __attribute__((aligned(16))) v4 bias_lds[N::bf_off(N::L) - N::bf_off(0)];
    __shared__ __attribute__((aligned(16))) v4 wst[2][13][64];
    stage_bias<N>(bias_lds, packed);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    const int64_t ngroup = (n + 63) / 64;
    WStream ws = make_stream(packed + S::start_f4, (N::packed_f4() - S::start_f4) * 16, lane);
    WStream ww = make_stream(packed + N::wf_off(0), N::wcount(0) * 16, lane);
    for (int64_t grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
        const int64_t row = (grp * 4 + wave) * 16 + (lane & 15);
        const bool valid = row < n;
        asm volatile("" : "+v"(ws.voff), "+v"(ww.voff));
        v4 a1[13];
        init_bias(a1, bias_lds, lane);
        wide_in_product_lds<F, IN64, 1, WRT>(a1, a1, wst, ww, xin, valid ? row : 0, 0, g, lane, wave, fr);
        lrelu(a1);
        Ring ring;
        ring_prime<S::total>(ring, ws);
        v4 a2[7], a3[4], a4[tiles(Z)];
        fwd_layer<N, S, 1>(a1, a2, ring, ws, bias_lds, lane);
        fwd_layer<N, S, 2>(a2, a3, ring, ws, bias_lds, lane);
        fwd_layer<N, S, 3>(a3, a4, ring, ws, bias_lds, lane);
        store_rows<Z, WRT>(a4, out, out_f64, row, valid, lane, nullptr, nullptr, zr);
    }
}

template <int F, int Z, bool OUT64, bool WRT = false>
__global__ void __launch_bounds__(256) wide_decode_lds_kernel(const v4 *packed, const void *__restrict__ zin, int in_f64, int64_t n,
                                                              void *__restrict__ out, int fr = F, int zr = Z) {
    using N = Net<F, Z>;
    using S = StreamWideDec<N>;
    __shared__ __attribute__((aligned(16))) v4 bias_lds[N::bf_off(N::L) - N::bf_off(0)];
    __shared__ __attribute__((aligned(16))) v4 wst[2][13][64];
    stage_bias<N>(bias_lds, packed);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    const int64_t ngroup = (n + 63) / 64;
    WStream ws = make_stream(packed + S::start_f4, (N::packed_f4() - S::start_f4) * 16, lane);
    WStream ww = make_stream(packed + N::wf_off(7), N::wcount(7) * 16, lane);
    for (int64_t grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
        const int64_t row = (grp * 4 + wave) * 16 + (lane & 15);
        const bool valid = row < n;
        asm volatile("" : "+v"(ws.voff), "+v"(ww.voff));
        Ring ring;
        ring_prime<S::total>(ring, ws);
        v4 a4[tiles(Z)], a5[4], a6[7], a7[13];
        load_rows<Z, WRT>(a4, zin, in_f64, row, valid, lane, nullptr, zr);
        fwd_layer<N, S, 4>(a4, a5, ring, ws, bias_lds, lane);
        fwd_layer<N, S, 5>(a5, a6, ring, ws, bias_lds, lane);
        fwd_layer<N, S, 6>(a6, a7, ring, ws, bias_lds, lane);
        wide_out_product_lds<F, WRT>(a7, wst, ww, bias_lds + (N::bf_off(7) - N::bf_off(0)), g, lane, wave, [&](int, auto, auto) {},
                                [&](const v4 &o, int t, auto, auto full) {
                                    if (valid) wide_store_tile<F, decltype(full)::value, WRT>(o, out, OUT64 ? 1 : 0, row, t, g, fr);
                                }, fr);
    }
}

// ---- wide models: the row-local part of a training pass ---------------------------------------------------------------
// The weight gradients of a wide model are split-K GEMMs over the whole chunk (generic.hip); everything that is LOCAL to a row --
// the forward pass, the loss and its gradient, the input-gradient chain -- runs here, one launch each, with the activations and
// the pre-activation gradients written once as row-major float32 for those GEMMs (no layer-by-layer round trips).
template <class N> struct StreamWideMid {   // forward fragments of layers 1..6
    static constexpr int fwd_base(int l) { return (N::wf_off(l) - N::wf_off(1)) / 64; }
    static constexpr int total = (N::wf_off(7) - N::wf_off(1)) / 64;
    static constexpr int start_f4 = N::wf_off(1);
};
template <class N> struct StreamWideMidBwd {   // transposed fragments of layers 6..1 (packed in that order)
    static constexpr int bwd_base(int l) { return (N::wb_off(l) - N::wb_off(6)) / 64; }
    static constexpr int total = (N::wb_off(0) - N::wb_off(6)) / 64;
    static constexpr int start_f4 = N::wb_off(6);
};
// one activation / gradient row block in C layout from / to a row-major float32 matrix [rows][D]
template <int D, bool WRT = false>
__device__ __forceinline__ void load_act(v4 (&a)[tiles(D)], const float *__restrict__ y, int64_t rrow, int g, int dr = D) {
#pragma unroll
    for (int t = 0; t < tiles(D); ++t) {
        if constexpr (WRT) {      // a class width: rows of dr values, slots beyond dr are zero
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = slot_feature(D, t, g, r);
                a[t][r] = (f >= 0 && f < dr) ? y[rrow * dr + f] : 0.f;
            }
        } else
        if (D - 16 * t >= 16 && D % 4 == 0) {
            a[t] = *(const v4 *)(y + rrow * D + 16 * t + 4 * g);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = slot_feature(D, t, g, r);
                a[t][r] = f >= 0 ? y[rrow * D + f] : 0.f;
            }
        }
    }
}

// forward + loss + dL/drecon of 16 rows per wave: en1 streamed, layers 1..6 chained in registers, de4 streamed tile by tile against
// the x tile it reconstructs (re-read: the wave streamed that row block a few microseconds earlier).  y1..y7 = activations,
// dz8 = 2 (recon - x) / F, loss_part[workgroup] = sum of squared errors (double, fixed order).
// TRAIN = false is the validation pass (training.py:104-137): no activation stores, `dz8` (may be null) receives the
// reconstruction itself as float32 / float64.
// MID (the small-batch pass, see wide_small_in_kernel): en1's sums arrive as `nsplit` partial accumulator sets per wave in `part`
// ([split][row group][wave][tile][lane], added in split order), the launch ends with y7 -- de4, the loss and dz8 are wide_small_out_kernel's.
template <int F, int Z, bool TRAIN, bool WRT = false, bool MID = false>
__global__ void __launch_bounds__(256) wide_train_fwd_kernel(const v4 *packed, const float *__restrict__ x, int64_t n, float *__restrict__ y1,
                                                             float *__restrict__ y2, float *__restrict__ y3, float *__restrict__ y4,
                                                             float *__restrict__ y5, float *__restrict__ y6, float *__restrict__ y7,
                                                             void *__restrict__ dz8, int out_f64, double *__restrict__ loss_part,
                                                             int fr = F, int zr = Z, const v4 *__restrict__ part = nullptr, int nsplit = 0) {
    using N = Net<F, Z>;
    using S = StreamWideMid<N>;
    const int KC = WRT ? (fr + 15) / 16 : tiles(F);
    __shared__ __attribute__((aligned(16))) v4 bias_lds[N::bf_off(N::L) - N::bf_off(0)];
    __shared__ double sh[256];
    stage_bias<N>(bias_lds, packed);
    __shared__ __attribute__((aligned(16))) v4 wst[2][13][64];      // en1's fragments, shared by the four waves (wide_in_product_lds)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    const int64_t ngroup = (n + 63) / 64;
    WStream ws = make_stream(packed + S::start_f4, (N::packed_f4() - S::start_f4) * 16, lane);
    WStream w0 = make_stream(packed + N::wf_off(0), N::wcount(0) * 16, lane);
    WStream w7 = make_stream(packed + N::wf_off(7), N::wcount(7) * 16, lane);
    const float gscale = 2.0f / (float)(WRT ? fr : F);
    double lacc = 0.0;
    for (int64_t grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {      // workgroup-uniform trip count: the product has barriers
        const int64_t row = (grp * 4 + wave) * 16 + (lane & 15);
        const bool valid = row < n;
        const int64_t rrow = valid ? row : 0;
        asm volatile("" : "+v"(ws.voff), "+v"(w0.voff), "+v"(w7.voff));
        v4 a7[13];
        {
            v4 a1[13];
            init_bias(a1, bias_lds, lane);
            if constexpr (MID) {
                for (int sp = 0; sp < nsplit; ++sp) {
                    const v4 *pp = part + (((int64_t)sp * ngroup + grp) * 4 + wave) * (13 * 64) + lane;
#pragma unroll
                    for (int t = 0; t < 13; ++t) a1[t] += pp[t * 64];
                }
            } else {
                wide_in_product_lds<F, false, 1, WRT>(a1, a1, wst, w0, x, rrow, 0, g, lane, wave, fr);
            }
            lrelu(a1);
            if (TRAIN) store_rows<200>(a1, y1, 0, row, valid, lane, nullptr, nullptr);
            Ring ring;
            ring_prime<S::total>(ring, ws);
            v4 a2[7], a3[4], a4[tiles(Z)], a5[4], a6[7];
            fwd_layer<N, S, 1>(a1, a2, ring, ws, bias_lds, lane);
            if (TRAIN) store_rows<100>(a2, y2, 0, row, valid, lane, nullptr, nullptr);
            fwd_layer<N, S, 2>(a2, a3, ring, ws, bias_lds, lane);
            if (TRAIN) store_rows<50>(a3, y3, 0, row, valid, lane, nullptr, nullptr);
            fwd_layer<N, S, 3>(a3, a4, ring, ws, bias_lds, lane);
            if (TRAIN) store_rows<Z, WRT>(a4, y4, 0, row, valid, lane, nullptr, nullptr, zr);
            fwd_layer<N, S, 4>(a4, a5, ring, ws, bias_lds, lane);
            if (TRAIN) store_rows<50>(a5, y5, 0, row, valid, lane, nullptr, nullptr);
            fwd_layer<N, S, 5>(a5, a6, ring, ws, bias_lds, lane);
            if (TRAIN) store_rows<100>(a6, y6, 0, row, valid, lane, nullptr, nullptr);
            fwd_layer<N, S, 6>(a6, a7, ring, ws, bias_lds, lane);
            if (TRAIN || MID) store_rows<200>(a7, y7, 0, row, valid, lane, nullptr, nullptr);      // (MID: wide_small_out_kernel reads it)
        }
        if constexpr (MID) continue;
        // de4 + loss: the x tiles are re-read three tiles ahead of the tile being multiplied, like the fragments (HBM again:
        // 327 MB of rows do not stay in the 256-MB MALL between the two passes)
        v4 xr[3];
        wide_out_product_lds<F, WRT>(a7, wst, w7, bias_lds + (N::bf_off(7) - N::bf_off(0)), g, lane, wave,
            [&](int t, auto slot, auto full) {
                constexpr int SL = decltype(slot)::value;
                if (decltype(full)::value) xr[SL] = wide_x_chunk<F, true, WRT>(x, 0, rrow, t, g, fr);
                else xr[SL] = wide_x_chunk<F, false, WRT>(x, 0, rrow, t < KC ? t : 0, g, fr);
            },
            [&](const v4 &o, int t, auto slot, auto full) {
                constexpr int SL = decltype(slot)::value;
                constexpr bool FL = decltype(full)::value;
                const v4 d = o - xr[SL];     // padding slots: zero weights and bias against a zero x
                if (valid) {
                    lacc += (double)(d[0] * d[0] + d[1] * d[1]) + (double)(d[2] * d[2] + d[3] * d[3]);
                    if (TRAIN) wide_store_tile<F, FL, WRT>(d * gscale, dz8, 0, row, t, g, fr);
                    else if (dz8) wide_store_tile<F, FL, WRT>(o, dz8, out_f64, row, t, g, fr);
                }
            }, fr);
    }
    if constexpr (MID) return;
    sh[threadIdx.x] = lacc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_part[blockIdx.x] = sh[0];
}

// ---- wide models, SMALL training batches (the reference's own: CFD_project_still trains with batch_size = 60, exafel 1 .. 36, hurricane
// 85, CFD_project_animation 6000) ------------------------------------------------------------------------------------------------
// wide_train_fwd / bwd_kernel give 16 rows to a wave that walks the whole wide dimension: 8,164 MFMAs for en1 and as many for de4 and for
// de4's input-gradient product, whatever the batch -- 0.3 + 0.16 ms per pass for 1 .. 4,096 rows.  Here the three wide products of a row
// group are SPLIT over workgroups: en1 and de4's input-gradient product over the wide (contraction) dimension -- partial accumulator sets,
// added in split order by the MID launches above and below --, de4 over its output tiles (with the loss and dz8).  Five launches instead of
// two, the same row-major activations / gradients for the weight-gradient kernels, the same arithmetic per element except for the order in
// which en1's and the input-gradient product's chunks are added (a fixed order).
template <int F, bool WRT = false>
__global__ void __launch_bounds__(256) wide_small_in_kernel(const v4 *wfrag /* [wide chunk][tile of the 200 side] */, int wcount_f4,
                                                            const float *__restrict__ src, int64_t n, v4 *__restrict__ part, int cps, int fr) {
    __shared__ __attribute__((aligned(16))) v4 wst[2][13][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    const int64_t ngroup = (n + 63) / 64;
    const int sp = blockIdx.x;                             // split of the wide dimension: full chunks [sp cps, (sp + 1) cps)
    const int64_t grp = blockIdx.y;
    const int KCA = (WRT ? fr : F) / 16;
    const int kc0 = sp * cps, kc1 = kc0 + cps < KCA ? kc0 + cps : KCA;
    const WStream ww = make_stream(wfrag, wcount_f4 * 16, lane);
    const int64_t row = (grp * 4 + wave) * 16 + (lane & 15);
    const int64_t rrow = row < n ? row : 0;
    v4 acc[13];
    zero_tiles(acc);
    wide_in_product_lds<F, false, 1, WRT>(acc, acc, wst, ww, src, rrow, 0, g, lane, wave, fr, kc0, kc1, sp == (int)gridDim.x - 1);
    v4 *pp = part + (((int64_t)sp * ngroup + grp) * 4 + wave) * (13 * 64) + lane;
#pragma unroll
    for (int t = 0; t < 13; ++t) pp[t * 64] = acc[t];
}
// The same partial sets for FEW rows (up to 128: the reference's 1 .. 85-row batches): a workgroup takes ONE 16-row tile and split, its
// four waves every fourth chunk of the split's range (fragments straight from L2: nothing to share, every wave is on its own chunk) and
// add their accumulators through LDS in wave order; 4 x as many waves on the product: 12.7 -> ~6 us per launch at 60 rows.
template <int F, bool WRT = false>
__global__ void __launch_bounds__(256) wide_small_in16_kernel(const v4 *wfrag, int wcount_f4, const float *__restrict__ src, int64_t n,
                                                              v4 *__restrict__ part, int cps, int fr) {
    __shared__ __attribute__((aligned(16))) v4 red[3][13][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    const int64_t ngroup = (n + 63) / 64;
    const int sp = blockIdx.x;
    const int64_t rt = blockIdx.y;                         // 16-row tile
    const int KCA = (WRT ? fr : F) / 16;
    const int kc0 = sp * cps, kc1 = kc0 + cps < KCA ? kc0 + cps : KCA;
    const WStream ww = make_stream(wfrag, wcount_f4 * 16, lane);
    const int64_t row = rt * 16 + (lane & 15);
    const int64_t rrow = row < n ? row : 0;
    v4 acc[13];
    zero_tiles(acc);
    for (int kc = kc0 + wave; kc < kc1; kc += 4) {
        const v4 xv = wide_x_chunk<F, true, WRT>(src, 0, rrow, kc, g, fr);
        v4 wt[13];
#pragma unroll
        for (int t = 0; t < 13; ++t) wt[t] = frag_rt(ww, kc * 13 + t);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 13; ++t) acc[t] = mfma(wt[t][r], xv[r], acc[t]);
    }
    if (wave == 3 && sp == (int)gridDim.x - 1 && (WRT ? fr : F) % 16 != 0) {      // the partial last chunk (as wide_in_product_lds)
        constexpr int ST = WRT ? 4 : tile_steps(F, F / 16);
        const v4 xv = wide_x_chunk<F, false, WRT>(src, 0, rrow, KCA, g, fr);
        v4 wt[13];
#pragma unroll
        for (int t = 0; t < 13; ++t) wt[t] = frag_rt(ww, KCA * 13 + t);
#pragma unroll
        for (int r = 0; r < ST; ++r)
#pragma unroll
            for (int t = 0; t < 13; ++t) acc[t] = mfma(wt[t][r], xv[r], acc[t]);
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < 13; ++t) red[wave - 1][t][lane] = acc[t];
    }
    __syncthreads();
    if (wave == 0) {
        v4 *pp = part + ((int64_t)sp * ngroup * 4 + rt) * (13 * 64) + lane;
#pragma unroll
        for (int t = 0; t < 13; ++t) pp[t * 64] = ((acc[t] + red[0][t][lane]) + red[1][t][lane]) + red[2][t][lane];
    }
}
// de4 + loss + dz8 of output tiles [blockIdx.x tps, ..) of row group blockIdx.y (y7 from the MID forward launch)
// TRAIN = false (the validation pass): `dz8` (may be null) receives the reconstruction itself as float32 / float64
template <int F, int Z, bool WRT = false, bool TRAIN = true>
__global__ void __launch_bounds__(256) wide_small_out_kernel(const v4 *packed, const float *__restrict__ x, int64_t n, const float *__restrict__ y7,
                                                             void *__restrict__ dz8, double *__restrict__ loss_part, int tps, int fr, int out_f64 = 0) {
    using N = Net<F, Z>;
    const int KC = WRT ? (fr + 15) / 16 : tiles(F);
    __shared__ __attribute__((aligned(16))) v4 wst[2][13][64];
    __shared__ __attribute__((aligned(16))) v4 bias7[tiles(F) * 4];
    __shared__ double sh[256];
    for (int i = threadIdx.x; i < tiles(F) * 4; i += 256) bias7[i] = packed[N::bf_off(7) + i];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    const int64_t grp = blockIdx.y;
    const int t0 = blockIdx.x * tps, t1 = t0 + tps < KC ? t0 + tps : KC;
    const WStream w7 = make_stream(packed + N::wf_off(7), N::wcount(7) * 16, lane);
    const int64_t row = (grp * 4 + wave) * 16 + (lane & 15);
    const bool valid = row < n;
    const int64_t rrow = valid ? row : 0;
    const float gscale = 2.0f / (float)(WRT ? fr : F);
    double lacc = 0.0;
    v4 a7[13];
    load_act<200>(a7, y7, rrow, g);
    __syncthreads();                                       // the bias tile table
    v4 xr[3];
    wide_out_product_lds<F, WRT>(a7, wst, w7, bias7, g, lane, wave,
        [&](int t, auto slot, auto full) {
            constexpr int SL = decltype(slot)::value;
            if (decltype(full)::value) xr[SL] = wide_x_chunk<F, true, WRT>(x, 0, rrow, t, g, fr);
            else xr[SL] = wide_x_chunk<F, false, WRT>(x, 0, rrow, t < KC ? t : 0, g, fr);
        },
        [&](const v4 &o, int t, auto slot, auto full) {
            constexpr int SL = decltype(slot)::value;
            constexpr bool FL = decltype(full)::value;
            const v4 d = o - xr[SL];
            if (valid) {
                lacc += (double)(d[0] * d[0] + d[1] * d[1]) + (double)(d[2] * d[2] + d[3] * d[3]);
                if (TRAIN) wide_store_tile<F, FL, WRT>(d * gscale, dz8, 0, row, t, g, fr);
                else if (dz8) wide_store_tile<F, FL, WRT>(o, dz8, out_f64, row, t, g, fr);
            }
        }, fr, t0, t1);
    sh[threadIdx.x] = lacc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_part[blockIdx.y * gridDim.x + blockIdx.x] = sh[0];
}

// The same for FEW rows (<= in16_rows()): ONE 16-row tile per workgroup (blockIdx.y), its four waves deal the tiles of the range among
// themselves (wave w: t0 + w, t0 + w + 4, ..) -- a 64-row group of a 60-row batch did the full work on its padding tiles.  A wave
// fetches the 13 fragments of its output tile itself (no stage, no barrier); the products of a tile are added in wide_out_product_lds's
// order (k tiles in groups of four, even / odd k tiles on two accumulators).
template <int F, int Z, bool WRT = false, bool TRAIN = true>
__global__ void __launch_bounds__(256) wide_small_out16_kernel(const v4 *packed, const float *__restrict__ x, int64_t n, const float *__restrict__ y7,
                                                               void *__restrict__ dz8, double *__restrict__ loss_part, int tps, int fr, int out_f64 = 0) {
    using N = Net<F, Z>;
    constexpr int KCS = tiles(F);
    const int KC = WRT ? (fr + 15) / 16 : tiles(F), KF = (WRT ? fr : F) / 16;
    __shared__ __attribute__((aligned(16))) v4 bias7[tiles(F) * 4];
    __shared__ double sh[256];
    for (int i = threadIdx.x; i < tiles(F) * 4; i += 256) bias7[i] = packed[N::bf_off(7) + i];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    const int t0 = blockIdx.x * tps, t1 = t0 + tps < KC ? t0 + tps : KC;
    const WStream w7 = make_stream(packed + N::wf_off(7), N::wcount(7) * 16, lane);
    const int64_t row = (int64_t)blockIdx.y * 16 + (lane & 15);
    const bool valid = row < n;
    const int64_t rrow = valid ? row : 0;
    const float gscale = 2.0f / (float)(WRT ? fr : F);
    double lacc = 0.0;
    v4 a7[13];
    load_act<200>(a7, y7, rrow, g);
    __syncthreads();                                       // the bias tile table
    for (int t = t0 + wave; t < t1; t += 4) {
        v4 w[13];
#pragma unroll
        for (int q = 0; q < 13; ++q) w[q] = frag_rt(w7, q * KCS + t);
        const bool full = t < KF;
        const v4 xr = full ? wide_x_chunk<F, true, WRT>(x, 0, rrow, t, g, fr) : wide_x_chunk<F, false, WRT>(x, 0, rrow, t, g, fr);
        v4 o0 = bias7[t * 4 + g], o1 = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q0 = 0; q0 < 16; q0 += 4)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (q0 + k < 13 && r < tile_steps(200, q0 + k)) {
                        if (k & 1) o1 = mfma(w[q0 + k][r], a7[q0 + k][r], o1); else o0 = mfma(w[q0 + k][r], a7[q0 + k][r], o0);
                    }
        const v4 o = o0 + o1, d = o - xr;
        if (valid) {
            lacc += (double)(d[0] * d[0] + d[1] * d[1]) + (double)(d[2] * d[2] + d[3] * d[3]);
            if (TRAIN) {
                if (full) wide_store_tile<F, true, WRT>(d * gscale, dz8, 0, row, t, g, fr); else wide_store_tile<F, false, WRT>(d * gscale, dz8, 0, row, t, g, fr);
            } else if (dz8) {
                if (full) wide_store_tile<F, true, WRT>(o, dz8, out_f64, row, t, g, fr); else wide_store_tile<F, false, WRT>(o, dz8, out_f64, row, t, g, fr);
            }
        }
    }
    sh[threadIdx.x] = lacc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_part[blockIdx.y * gridDim.x + blockIdx.x] = sh[0];
}

// the input-gradient chain of 16 rows per wave: dZ_6 = (dZ_7 W_7) * lrelu'(y7) streamed over the wide dimension, then layers 6..1
// chained in registers; every dZ_l (dL/d pre-activation of layer l) is stored for the weight-gradient GEMMs.
// MID (small batches): d6 starts from the partial sums of wide_small_in_kernel (as the forward launch's MID mode)
template <int F, int Z, bool WRT = false, bool MID = false>
__global__ void __launch_bounds__(256) wide_train_bwd_kernel(const v4 *packed, const float *__restrict__ dz7, int64_t n,
                                                             const float *__restrict__ y1, const float *__restrict__ y2,
                                                             const float *__restrict__ y3, const float *__restrict__ y5,
                                                             const float *__restrict__ y6, const float *__restrict__ y7,
                                                             float *__restrict__ dz0, float *__restrict__ dz1, float *__restrict__ dz2,
                                                             float *__restrict__ dz3, float *__restrict__ dz4, float *__restrict__ dz5,
                                                             float *__restrict__ dz6, const float *__restrict__ dz_latent,
                                                             int fr = F, int zr = Z, const v4 *__restrict__ part = nullptr, int nsplit = 0) {
    using N = Net<F, Z>;
    using S = StreamWideMidBwd<N>;
    __shared__ __attribute__((aligned(16))) v4 wst[2][13][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    const int64_t ngroup = (n + 63) / 64;
    WStream ws = make_stream(packed + S::start_f4, (N::packed_f4() - S::start_f4) * 16, lane);
    WStream w7 = make_stream(packed + N::wb_off(7), N::wcount(7) * 16, lane);       // [wide chunk][tile of the 200 side]
    for (int64_t grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
        const int64_t row = (grp * 4 + wave) * 16 + (lane & 15);
        const bool valid = row < n;
        const int64_t rrow = valid ? row : 0;
        asm volatile("" : "+v"(ws.voff), "+v"(w7.voff));
        v4 d6[13];
        zero_tiles(d6);
        if constexpr (MID) {
            for (int sp = 0; sp < nsplit; ++sp) {
                const v4 *pp = part + (((int64_t)sp * ngroup + grp) * 4 + wave) * (13 * 64) + lane;
#pragma unroll
                for (int t = 0; t < 13; ++t) d6[t] += pp[t * 64];
            }
        } else {
            wide_in_product_lds<F, false, 1, WRT>(d6, d6, wst, w7, dz7, rrow, 0, g, lane, wave, fr);
        }
        Ring ring;
        ring_prime<S::total>(ring, ws);
        {
            v4 a[13];
            load_act<200>(a, y7, rrow, g);
            lrelu_bwd(d6, a);
        }
        store_rows<200>(d6, dz6, 0, row, valid, lane, nullptr, nullptr);
        v4 d5[7], d4[4], d3[tiles(Z)], d2[4], d1[7], d0[13];
        bwd_layer<N, S, 6>(d6, d5, ring, ws);
        { v4 a[7]; load_act<100>(a, y6, rrow, g); lrelu_bwd(d5, a); }
        store_rows<100>(d5, dz5, 0, row, valid, lane, nullptr, nullptr);
        bwd_layer<N, S, 5>(d5, d4, ring, ws);
        { v4 a[4]; load_act<50>(a, y5, rrow, g); lrelu_bwd(d4, a); }
        store_rows<50>(d4, dz4, 0, row, valid, lane, nullptr, nullptr);
        bwd_layer<N, S, 4>(d4, d3, ring, ws);                      // the latent layer has no activation
        if (dz_latent) {      // dL/dz of the caller's regulariser (bamd_fwd_bwd_latent: the sliced-Wasserstein term), added at the bottleneck
            v4 e[tiles(Z)];
            load_act<Z, WRT>(e, dz_latent, rrow, g, zr);
#pragma unroll
            for (int t = 0; t < tiles(Z); ++t) d3[t] += e[t];
        }
        store_rows<Z, WRT>(d3, dz3, 0, row, valid, lane, nullptr, nullptr, zr);
        bwd_layer<N, S, 3>(d3, d2, ring, ws);
        { v4 a[4]; load_act<50>(a, y3, rrow, g); lrelu_bwd(d2, a); }
        store_rows<50>(d2, dz2, 0, row, valid, lane, nullptr, nullptr);
        bwd_layer<N, S, 2>(d2, d1, ring, ws);
        { v4 a[7]; load_act<100>(a, y2, rrow, g); lrelu_bwd(d1, a); }
        store_rows<100>(d1, dz1, 0, row, valid, lane, nullptr, nullptr);
        bwd_layer<N, S, 1>(d1, d0, ring, ws);
        { v4 a[13]; load_act<200>(a, y1, rrow, g); lrelu_bwd(d0, a); }
        store_rows<200>(d0, dz0, 0, row, valid, lane, nullptr, nullptr);
    }
}

// ---- bf16 training of the wide models (BAMD_MODE_BF16): the two wide products of the forward pass (en1, de4) and de4's input-gradient
// product on v_mfma_f32_16x16x32_bf16, everything else -- the six narrow layers, the loss, the masks, and all weight-gradient products
// (which read the float32 activations / gradients these kernels store) -- in float32 as in wide_train_fwd/bwd_kernel.  95 % of the
// model's forward and input-gradient work is in those three products; at fp32 MFMA rates they made the row-local launches MFMA-bound
// (75 % busy), at bf16 rates the launches are bound by their 10-KB-per-frame row traffic.
//
// acc0 / acc1[13 tiles of the 200-feature side] += sum over the wide dimension of frag(chunk c, tile t) . x^T[chunk c] for the wave's two
// 16-row tiles: the streamed product of wide_bf16_encode_kernel as a function (fragments [32-feature chunk][tile], shared by the four waves
// through the double-buffered LDS stage `wst`, every load three chunks ahead; see that kernel).  Contains workgroup barriers.
// SRC16: the rows are stored as bfloat16 (the dz8 of a BF16 handle's training pass, F % 4 == 0): 16 bytes per lane and chunk as two
// 8-byte loads (a row is F x 2 bytes: 8-byte aligned), no conversion; xo0 / xo1 are then byte offsets of bfloat16 elements.
template <int F, bool SRC16 = false>
__device__ __forceinline__ void wide_in_product_bf16(v4 (&a1)[13], v4 (&b1)[13], v4 (&wst)[2][13][64], const WStream &ww,
                                                     __amdgpu_buffer_rsrc_t xrs, int xo0, int xo1, int wave, int lane, int g) {
    constexpr int KB = F / 32;
    bf8 wq[2][4];
    struct XR { XPair p; bf8 h; };
    XR x0r[3], x1r[3];
    auto ldx = [&](int xo, int c) {
        XR r;
        if constexpr (SRC16) {
            typedef unsigned u2_ __attribute__((ext_vector_type(2)));
            typedef unsigned u4_ __attribute__((ext_vector_type(4)));
            const u2_ lo = __builtin_bit_cast(u2_, __builtin_amdgcn_raw_buffer_load_b64(xrs, xo, c * 64, 0));
            const u2_ hi = __builtin_bit_cast(u2_, __builtin_amdgcn_raw_buffer_load_b64(xrs, xo + 8, c * 64, 0));
            r.h = __builtin_bit_cast(bf8, (u4_){lo[0], lo[1], hi[0], hi[1]});
        } else {
            r.p = wide_x_chunk32_buf<F, true>(xrs, xo, 0, c, g);
        }
        return r;
    };
    auto cvx = [&](const XR &r) {
        if constexpr (SRC16) return r.h;
        else return to_bf8(r.p.lo, r.p.hi);
    };
    auto wload = [&](bf8 (&w)[4], int c) {
        c = c < KB ? c : KB - 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = wave + 4 * k;
            w[k] = frag_bf(ww, c * 13 + (t < 13 ? t : 12));
        }
    };
    auto wstore = [&](const bf8 (&w)[4], int slot) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = wave + 4 * k;
            if (t < 13) wst[slot][t][lane] = __builtin_bit_cast(v4, w[k]);
        }
    };
    auto lx0 = [&](int c) { return ldx(xo0, c < KB ? c : 0); };
    auto lx1 = [&](int c) { return ldx(xo1, c < KB ? c : 0); };
    wload(wq[0], 0);
    wload(wq[1], 1);
#pragma unroll
    for (int u = 0; u < 3; ++u) { x0r[u] = lx0(u); x1r[u] = lx1(u); }
    __syncthreads();              // the previous user of the stage has read its last chunk
    wstore(wq[0], 0);
    wload(wq[0], 2);
    auto iter = [&](int c, auto wsl, auto xsl) {
        constexpr int WS = decltype(wsl)::value, XS = decltype(xsl)::value;      // c % 2, c % 3
        __syncthreads();
        wstore(wq[WS ^ 1], WS ^ 1);
        const bf8 q0 = cvx(x0r[XS]), q1 = cvx(x1r[XS]);
        wload(wq[WS ^ 1], c + 3);
        x0r[XS] = lx0(c + 3);
        x1r[XS] = lx1(c + 3);
        bf8 wl[2][4];
        auto rd = [&](bf8 (&w)[4], int t0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = __builtin_bit_cast(bf8, wst[WS][t0 + k < 13 ? t0 + k : 12][lane]);
        };
        auto mm = [&](const bf8 (&w)[4], int t0) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (t0 + k < 13) { a1[t0 + k] = mfma_bf(w[k], q0, a1[t0 + k]); b1[t0 + k] = mfma_bf(w[k], q1, b1[t0 + k]); }
        };
        rd(wl[0], 0);
        rd(wl[1], 4);
        __builtin_amdgcn_sched_barrier(0);
        mm(wl[0], 0);
        rd(wl[0], 8);
        __builtin_amdgcn_sched_barrier(0);
        mm(wl[1], 4);
        rd(wl[1], 12);
        __builtin_amdgcn_sched_barrier(0);
        mm(wl[0], 8);
        mm(wl[1], 12);
        __builtin_amdgcn_sched_barrier(0);
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
    int c = 0;
    for (; c + 6 <= KB; c += 6) {
        iter(c, I0(), I0()); iter(c + 1, I1(), I1()); iter(c + 2, I0(), I2());
        iter(c + 3, I1(), I0()); iter(c + 4, I0(), I1()); iter(c + 5, I1(), I2());
    }
    if (c < KB) iter(c, I0(), I0());
    if (c + 1 < KB) iter(c + 1, I1(), I1());
    if (c + 2 < KB) iter(c + 2, I0(), I2());
    if (c + 3 < KB) iter(c + 3, I1(), I0());
    if (c + 4 < KB) iter(c + 4, I0(), I1());
    if (F % 32 != 0) {            // the remaining F % 32 features: one partial chunk, fragments straight from L2
        bf8 q0, q1;
        if constexpr (SRC16) {      // pairs of elements (F is even: a pair never straddles the end of the row), zero beyond the row
            typedef unsigned u4_ __attribute__((ext_vector_type(4)));
            u4_ w0 = {0u, 0u, 0u, 0u}, w1 = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int jp = 0; jp < 4; ++jp)
                if (32 * KB + 8 * g + 2 * jp < F) {
                    w0[jp] = __builtin_bit_cast(unsigned, __builtin_amdgcn_raw_buffer_load_b32(xrs, xo0 + 4 * jp, KB * 64, 0));
                    w1[jp] = __builtin_bit_cast(unsigned, __builtin_amdgcn_raw_buffer_load_b32(xrs, xo1 + 4 * jp, KB * 64, 0));
                }
            q0 = __builtin_bit_cast(bf8, w0);
            q1 = __builtin_bit_cast(bf8, w1);
        } else {
            const XPair p0 = wide_x_chunk32_buf<F>(xrs, xo0, 0, KB, g), p1 = wide_x_chunk32_buf<F>(xrs, xo1, 0, KB, g);
            q0 = to_bf8(p0.lo, p0.hi);
            q1 = to_bf8(p1.lo, p1.hi);
        }
#pragma unroll
        for (int t = 0; t < 13; ++t) {
            const bf8 w = frag_bf(ww, KB * 13 + t);
            a1[t] = mfma_bf(w, q0, a1[t]);
            b1[t] = mfma_bf(w, q1, b1[t]);
        }
    }
}

// forward + loss + dL/drecon of 2 x 16 rows per wave (128 rows per workgroup pass): interface and results as wide_train_fwd_kernel<TRAIN = true>
// (y1..y7, dz8 = 2 (recon - x) / F in float32, per-workgroup loss partials), en1 and de4 on the bf16 MFMA.
// four values of a wide row as bfloat16 (DZ16: dz8 is stored as bfloat16 -- both of its readers, the input-gradient product and de4's
// weight gradient, round it to bfloat16 on load anyway: same numbers, half the bytes of its one write and two reads)
__device__ __forceinline__ unsigned long long pack4_bf16(const v4 &o) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 lo = {(__bf16)o[0], (__bf16)o[1]}, hi = {(__bf16)o[2], (__bf16)o[3]};
    return (unsigned long long)__builtin_bit_cast(unsigned, lo) | ((unsigned long long)__builtin_bit_cast(unsigned, hi) << 32);
}
template <int F, bool FULL = false>
__device__ __forceinline__ void wide_store_tile_bf16(const v4 &o, __bf16 *out, int64_t row, int t, int g) {
    if (FULL || 16 * t + 16 <= F) {
        *(unsigned long long *)(out + row * F + 16 * t + 4 * g) = pack4_bf16(o);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = slot_feature(F, t, g, r);
            if (f >= 0) out[row * F + f] = (__bf16)o[r];
        }
    }
}
template <int F, int Z, bool DZ16>
__global__ void __launch_bounds__(256) wide_bf16_train_fwd_kernel(const v4 *packed, const v4 *w0b, const v4 *w7b, const float *__restrict__ x,
                                                                  int64_t n, float *__restrict__ y1, float *__restrict__ y2,
                                                                  float *__restrict__ y3, float *__restrict__ y4, float *__restrict__ y5,
                                                                  float *__restrict__ y6, float *__restrict__ y7, void *__restrict__ dz8v,
                                                                  double *__restrict__ loss_part) {
    float *const dz8 = (float *)dz8v;
    __bf16 *const dz8h = (__bf16 *)dz8v;
    using N = Net<F, Z>;
    using S = StreamWideMid<N>;
    constexpr int KBT = (F + 31) / 32, KT = tiles(F), KTF = F / 16;
    __shared__ __attribute__((aligned(16))) v4 bias_lds[N::bf_off(N::L) - N::bf_off(0)];
    __shared__ __attribute__((aligned(16))) v4 wst[2][13][64];
    constexpr int kTG = 4, kTS = 16 * kTG + 4;
    __shared__ __attribute__((aligned(16))) float tstage[4][32][kTS];     // transposing stage of the dz8 stores (wide_bf16_decode_kernel)
    __shared__ double sh[256];
    stage_bias<N>(bias_lds, packed);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    const int64_t ngroup = (n + 127) / 128;
    WStream ws = make_stream(packed + S::start_f4, (N::packed_f4() - S::start_f4) * 16, lane);
    WStream w0 = make_stream(w0b, KBT * 13 * 1024, lane);
    WStream w7 = make_stream(w7b, KT * 7 * 1024, lane);
    const v4 *bias7 = bias_lds + (N::bf_off(7) - N::bf_off(0));
    const float gscale = 2.0f / F;
    double lacc = 0.0;
    for (int64_t grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
        const int64_t r0 = (grp * 4 + wave) * 32 + (lane & 15), r1 = r0 + 16;
        const bool v0 = r0 < n, v1 = r1 < n;
        const int64_t rr0 = v0 ? r0 : 0, rr1 = v1 ? r1 : 0;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)x + (size_t)(grp * 128) * F * 4), 0,
                                                                             0x7fffffff, 0x00020000);
        const int lr0 = wave * 32 + (lane & 15);
        const int xo0 = ((v0 ? lr0 : 0) * F + 8 * g) * 4, xo1 = ((v1 ? lr0 + 16 : 0) * F + 8 * g) * 4;
        asm volatile("" : "+v"(ws.voff), "+v"(w0.voff), "+v"(w7.voff));
        bf8 qa[7], qb[7];
        {
            v4 a1[13], b1[13];
            init_bias(a1, bias_lds, lane);
#pragma unroll
            for (int t = 0; t < 13; ++t) b1[t] = a1[t];
            wide_in_product_bf16<F>(a1, b1, wst, w0, xrs, xo0, xo1, wave, lane, g);
            lrelu(a1);
            lrelu(b1);
            store_rows<200>(a1, y1, 0, r0, v0, lane, nullptr, nullptr);
            store_rows<200>(b1, y1, 0, r1, v1, lane, nullptr, nullptr);
            const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
            auto narrow = [&](const v4 (&in)[13], bf8 (&q)[7], int64_t row, bool valid) {
                Ring ring;
                ring_prime<S::total>(ring, ws);
                v4 a2[7], a3[4], a4[tiles(Z)], a5[4], a6[7], a7[13];
                fwd_layer<N, S, 1>(in, a2, ring, ws, bias_lds, lane);
                store_rows<100>(a2, y2, 0, row, valid, lane, nullptr, nullptr);
                fwd_layer<N, S, 2>(a2, a3, ring, ws, bias_lds, lane);
                store_rows<50>(a3, y3, 0, row, valid, lane, nullptr, nullptr);
                fwd_layer<N, S, 3>(a3, a4, ring, ws, bias_lds, lane);
                store_rows<Z>(a4, y4, 0, row, valid, lane, nullptr, nullptr);
                fwd_layer<N, S, 4>(a4, a5, ring, ws, bias_lds, lane);
                store_rows<50>(a5, y5, 0, row, valid, lane, nullptr, nullptr);
                fwd_layer<N, S, 5>(a5, a6, ring, ws, bias_lds, lane);
                store_rows<100>(a6, y6, 0, row, valid, lane, nullptr, nullptr);
                fwd_layer<N, S, 6>(a6, a7, ring, ws, bias_lds, lane);
                store_rows<200>(a7, y7, 0, row, valid, lane, nullptr, nullptr);
#pragma unroll
                for (int c = 0; c < 7; ++c) q[c] = to_bf8(a7[2 * c], 2 * c + 1 < 13 ? a7[2 * c + 1 < 13 ? 2 * c + 1 : 12] : zero);
            };
            narrow(a1, qa, r0, v0);
            narrow(b1, qb, r1, v1);
        }
        // de4 tile by tile against the x tile it reconstructs; fragments and x tiles three output tiles ahead in four rotating buffers
        bf8 wr[4][7];
        v4 xa[4], xb[4];
        // (the loop covers the KTF full tiles; a partial last tile is an epilogue: a "full tile?" test in front of the x loads would put
        // every load of the loop behind a branch, see wide_x_chunk)
        auto load_w = [&](bf8 (&w)[7], v4 &x0t, v4 &x1t, int t) {
            t = t < KTF ? t : KTF - 1;
#pragma unroll
            for (int c = 0; c < 7; ++c) w[c] = frag_bf(w7, t * 7 + c);
            x0t = wide_x_chunk<F, true>(x, 0, rr0, t, g);
            x1t = wide_x_chunk<F, true>(x, 0, rr1, t, g);
        };
        auto tile_out = [&](const bf8 (&w)[7], const v4 &x0t, const v4 &x1t, int t, auto jj) {
            constexpr int J = decltype(jj)::value % kTG;
            if (t >= KTF) return;
            v4 o0 = bias7[t * 4 + g], o1 = o0;
#pragma unroll
            for (int c = 0; c < 7; ++c) { o0 = mfma_bf(w[c], qa[c], o0); o1 = mfma_bf(w[c], qb[c], o1); }
            const v4 d0 = o0 - x0t, d1 = o1 - x1t;
            if (v0) lacc += (double)(d0[0] * d0[0] + d0[1] * d0[1]) + (double)(d0[2] * d0[2] + d0[3] * d0[3]);
            if (v1) lacc += (double)(d1[0] * d1[0] + d1[1] * d1[1]) + (double)(d1[2] * d1[2] + d1[3] * d1[3]);
            const v4 e0 = d0 * gscale, e1 = d1 * gscale;
            if (t + (kTG - 1 - J) < KTF) {               // the whole group is made of full tiles: 4 rows x 256 contiguous bytes per store
                *(v4 *)&tstage[wave][lane & 15][16 * J + 4 * g] = e0;
                *(v4 *)&tstage[wave][16 + (lane & 15)][16 * J + 4 * g] = e1;
                if (J == kTG - 1) {
                    const int64_t rb = (grp * 4 + wave) * 32;
                    constexpr int LR = 4 * kTG;
#pragma unroll
                    for (int k = 0; k < 32 * LR / 64; ++k) {
                        const int rl = (64 / LR) * k + lane / LR;
                        const v4 v = *(const v4 *)&tstage[wave][rl][4 * (lane % LR)];
                        if (rb + rl < n) {
                            if constexpr (DZ16) *(unsigned long long *)(dz8h + (rb + rl) * F + 16 * (t - (kTG - 1)) + 4 * (lane % LR)) = pack4_bf16(v);
                            else *(v4 *)(dz8 + (rb + rl) * F + 16 * (t - (kTG - 1)) + 4 * (lane % LR)) = v;
                        }
                    }
                }
            } else if constexpr (DZ16) {
                if (v0) wide_store_tile_bf16<F, true>(e0, dz8h, r0, t, g);
                if (v1) wide_store_tile_bf16<F, true>(e1, dz8h, r1, t, g);
            } else {
                if (v0) wide_store_tile<F, true>(e0, dz8, 0, r0, t, g);
                if (v1) wide_store_tile<F, true>(e1, dz8, 0, r1, t, g);
            }
        };
        load_w(wr[0], xa[0], xb[0], 0);
        load_w(wr[1], xa[1], xb[1], 1);
        load_w(wr[2], xa[2], xb[2], 2);
        for (int t0 = 0; t0 < KTF; t0 += 8) {
            load_w(wr[3], xa[3], xb[3], t0 + 3);
            tile_out(wr[0], xa[0], xb[0], t0, std::integral_constant<int, 0>());
            __builtin_amdgcn_sched_barrier(0);
            load_w(wr[0], xa[0], xb[0], t0 + 4);
            tile_out(wr[1], xa[1], xb[1], t0 + 1, std::integral_constant<int, 1>());
            __builtin_amdgcn_sched_barrier(0);
            load_w(wr[1], xa[1], xb[1], t0 + 5);
            tile_out(wr[2], xa[2], xb[2], t0 + 2, std::integral_constant<int, 2>());
            __builtin_amdgcn_sched_barrier(0);
            load_w(wr[2], xa[2], xb[2], t0 + 6);
            tile_out(wr[3], xa[3], xb[3], t0 + 3, std::integral_constant<int, 3>());
            __builtin_amdgcn_sched_barrier(0);
            load_w(wr[3], xa[3], xb[3], t0 + 7);
            tile_out(wr[0], xa[0], xb[0], t0 + 4, std::integral_constant<int, 4>());
            __builtin_amdgcn_sched_barrier(0);
            load_w(wr[0], xa[0], xb[0], t0 + 8);
            tile_out(wr[1], xa[1], xb[1], t0 + 5, std::integral_constant<int, 5>());
            __builtin_amdgcn_sched_barrier(0);
            load_w(wr[1], xa[1], xb[1], t0 + 9);
            tile_out(wr[2], xa[2], xb[2], t0 + 6, std::integral_constant<int, 6>());
            __builtin_amdgcn_sched_barrier(0);
            load_w(wr[2], xa[2], xb[2], t0 + 10);
            tile_out(wr[3], xa[3], xb[3], t0 + 7, std::integral_constant<int, 7>());
            __builtin_amdgcn_sched_barrier(0);
        }
        if (F % 16 != 0) {      // the partial last tile (r-major slots; padding slots: zero weights and bias against a zero x)
            constexpr int t = KT - 1;
            v4 o0 = bias7[t * 4 + g], o1 = o0;
#pragma unroll
            for (int c = 0; c < 7; ++c) {
                const bf8 w = frag_bf(w7, t * 7 + c);
                o0 = mfma_bf(w, qa[c], o0);
                o1 = mfma_bf(w, qb[c], o1);
            }
            const v4 d0 = o0 - wide_x_chunk<F>(x, 0, rr0, t, g), d1 = o1 - wide_x_chunk<F>(x, 0, rr1, t, g);
            if (v0) {
                lacc += (double)(d0[0] * d0[0] + d0[1] * d0[1]) + (double)(d0[2] * d0[2] + d0[3] * d0[3]);
                if constexpr (DZ16) wide_store_tile_bf16<F>(d0 * gscale, dz8h, r0, t, g); else wide_store_tile<F>(d0 * gscale, dz8, 0, r0, t, g);
            }
            if (v1) {
                lacc += (double)(d1[0] * d1[0] + d1[1] * d1[1]) + (double)(d1[2] * d1[2] + d1[3] * d1[3]);
                if constexpr (DZ16) wide_store_tile_bf16<F>(d1 * gscale, dz8h, r1, t, g); else wide_store_tile<F>(d1 * gscale, dz8, 0, r1, t, g);
            }
        }
    }
    sh[threadIdx.x] = lacc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_part[blockIdx.x] = sh[0];
}

// the input-gradient chain of 2 x 16 rows per wave: dZ_6 = (dZ_7 W_7) * lrelu'(y7) with the wide product on the bf16 MFMA (fragments of
// W_7^T in en1's [chunk][tile] order), then layers 6..1 in float32 as in wide_train_bwd_kernel, one row tile after the other.
template <int F, int Z, bool DZ16>
__global__ void __launch_bounds__(256) wide_bf16_train_bwd_kernel(const v4 *packed, const v4 *w7tb, const void *__restrict__ dz7, int64_t n,
                                                                  const float *__restrict__ y1, const float *__restrict__ y2,
                                                                  const float *__restrict__ y3, const float *__restrict__ y5,
                                                                  const float *__restrict__ y6, const float *__restrict__ y7,
                                                                  float *__restrict__ dz0, float *__restrict__ dz1, float *__restrict__ dz2,
                                                                  float *__restrict__ dz3, float *__restrict__ dz4, float *__restrict__ dz5,
                                                                  float *__restrict__ dz6, const float *__restrict__ dz_latent) {
    using N = Net<F, Z>;
    using S = StreamWideMidBwd<N>;
    constexpr int KBT = (F + 31) / 32;
    __shared__ __attribute__((aligned(16))) v4 wst[2][13][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    const int64_t ngroup = (n + 127) / 128;
    WStream ws = make_stream(packed + S::start_f4, (N::packed_f4() - S::start_f4) * 16, lane);
    WStream w7 = make_stream(w7tb, KBT * 13 * 1024, lane);
    for (int64_t grp = blockIdx.x; grp < ngroup; grp += gridDim.x) {
        const int64_t r0 = (grp * 4 + wave) * 32 + (lane & 15), r1 = r0 + 16;
        const bool v0 = r0 < n, v1 = r1 < n;
        constexpr int es = DZ16 ? 2 : 4;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)dz7 + (size_t)(grp * 128) * F * es), 0,
                                                                             0x7fffffff, 0x00020000);
        const int lr0 = wave * 32 + (lane & 15);
        const int xo0 = ((v0 ? lr0 : 0) * F + 8 * g) * es, xo1 = ((v1 ? lr0 + 16 : 0) * F + 8 * g) * es;
        asm volatile("" : "+v"(ws.voff), "+v"(w7.voff));
        v4 d6a[13], d6b[13];
        zero_tiles(d6a);
        zero_tiles(d6b);
        wide_in_product_bf16<F, DZ16>(d6a, d6b, wst, w7, xrs, xo0, xo1, wave, lane, g);
        auto narrow = [&](v4 (&d6)[13], int64_t row, bool valid) {
            const int64_t rrow = valid ? row : 0;
            Ring ring;
            ring_prime<S::total>(ring, ws);
            {
                v4 a[13];
                load_act<200>(a, y7, rrow, g);
                lrelu_bwd(d6, a);
            }
            store_rows<200>(d6, dz6, 0, row, valid, lane, nullptr, nullptr);
            v4 d5[7], d4[4], d3[tiles(Z)], d2[4], d1[7], d0[13];
            bwd_layer<N, S, 6>(d6, d5, ring, ws);
            { v4 a[7]; load_act<100>(a, y6, rrow, g); lrelu_bwd(d5, a); }
            store_rows<100>(d5, dz5, 0, row, valid, lane, nullptr, nullptr);
            bwd_layer<N, S, 5>(d5, d4, ring, ws);
            { v4 a[4]; load_act<50>(a, y5, rrow, g); lrelu_bwd(d4, a); }
            store_rows<50>(d4, dz4, 0, row, valid, lane, nullptr, nullptr);
            bwd_layer<N, S, 4>(d4, d3, ring, ws);
            if (dz_latent) {
                v4 e[tiles(Z)];
                load_act<Z>(e, dz_latent, rrow, g);
#pragma unroll
                for (int t = 0; t < tiles(Z); ++t) d3[t] += e[t];
            }
            store_rows<Z>(d3, dz3, 0, row, valid, lane, nullptr, nullptr);
            bwd_layer<N, S, 3>(d3, d2, ring, ws);
            { v4 a[4]; load_act<50>(a, y3, rrow, g); lrelu_bwd(d2, a); }
            store_rows<50>(d2, dz2, 0, row, valid, lane, nullptr, nullptr);
            bwd_layer<N, S, 2>(d2, d1, ring, ws);
            { v4 a[7]; load_act<100>(a, y2, rrow, g); lrelu_bwd(d1, a); }
            store_rows<100>(d1, dz1, 0, row, valid, lane, nullptr, nullptr);
            bwd_layer<N, S, 1>(d1, d0, ring, ws);
            { v4 a[13]; load_act<200>(a, y1, rrow, g); lrelu_bwd(d0, a); }
            store_rows<200>(d0, dz0, 0, row, valid, lane, nullptr, nullptr);
        };
        narrow(d6a, r0, v0);
        narrow(d6b, r1, v1);
    }
}

// ---- training kernels ---------------------------------------------------------------------------------
// Activations + weight-gradient accumulators of the whole model do not fit one CU (register file 512 KB + LDS
// 160 KB), so training runs as TWO launches over the same rows, cut at layer kSplit.  The first
// ("decoder-gradient") kernel runs the whole forward, the loss and the backward chain of layers 7..kSplit,
// accumulating their weight gradients; it hands dL/d(pre-activation of layer kSplit-1) to the second
// ("encoder-gradient") kernel, which recomputes the forward of layers 0..kSplit-2, back-propagates layers
// kSplit-1..0 and accumulates their weight gradients.  Each kernel keeps its share of the weight-gradient tiles
// in MFMA accumulators for its WHOLE persistent loop: no partial-gradient traffic inside the loop; one slab store
// per workgroup at the end.  kSplit = 4 cuts at the bottleneck (16 floats per row handed off, encoder forward
// recomputed: +16 % MFMAs); kSplit = 2 (default) cuts after en2 (112 floats per row, only en1 recomputed: +2 %).
//
// LDS image helpers: rows = feature slots (16t + 4g + r), columns = the workgroup's 64 batch rows.
template <int NT>
__device__ __forceinline__ void q_write(float *__restrict__ q, const v4 (&a)[NT], int lane, int wave) {
    // One base address per register r (slot 4g + r of tile 0); tiles are 16 * kQS floats = 17 x 256 bytes apart, so
    // the stores of tiles t, t+1 pair into ds_write2st64_b32 with IMMEDIATE offsets: 4 address adds per image instead
    // of one per tile (ds_write2_b32 reaches only 1 KiB, which pairs registers r, r+1 and needs a new base every tile).
    typedef float __attribute__((address_space(3))) *lds_f;
    const int col = 16 * wave + (lane & 15), g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        lds_f p = (lds_f)q + ((4 * g + r) * kQS + col);
        asm volatile("" : "+v"(p));
#pragma unroll
        for (int t = 0; t < NT; ++t) p[t * 16 * kQS] = a[t][r];
    }
}
// X^T image with the ones row in the first padding slot of dimension D (carries db through the GEMM)
template <int D>
__device__ __forceinline__ void q_write_x(float *__restrict__ q, const v4 (&a)[tiles(D)], int lane, int wave) {
    static_assert(D % 16 != 0, "ones slot lives in the partial last tile");
    constexpr int T = tiles(D) - 1, V = D - 16 * T;      // partial tile, V valid slots; ones slot idx = V
    constexpr int R1 = V / 4, G1 = V % 4;
    typedef float __attribute__((address_space(3))) *lds_f;
    const int col = 16 * wave + (lane & 15), g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                        // see q_write: tiles t, t+1 pair with immediate offsets
        lds_f p = (lds_f)q + ((4 * g + r) * kQS + col);
        asm volatile("" : "+v"(p));
#pragma unroll
        for (int t = 0; t < tiles(D); ++t) {
            float v = a[t][r];
            if (t == T && r == R1 && g == G1) v = 1.0f;
            p[t * 16 * kQS] = v;
        }
    }
}

template <class N, int l> struct DW {
    static constexpr int NT = tiles(N::dim(l + 1)), KT = tiles(N::dim(l) + 1), TOT = NT * KT, T = (TOT + 3) / 4;
    static constexpr int rows_x = 16 * KT, rows_dz = 16 * NT;   // image rows
};

// [dW | db] tiles of layer l: D[n-slot][k-slot] += sum over the workgroup's 64 rows of dZ^T[n][m] X^T[k][m].
// Tiles are dealt round-robin to the 4 waves (tile idx = wave + 4*it lives in acc[it] for the whole
// kernel); every wave runs the same static schedule (the surplus tile of a short wave recomputes the last
// tile into an accumulator that is never stored).  LDS fragment reads run DEPTH-1 steps ahead of the MFMAs.
template <class N, int l>
__device__ __forceinline__ void dw_phase(const float *__restrict__ qdz, const float *__restrict__ qx,
                                         v4 (&acc)[DW<N, l>::T], int lane, int wave) {
    using D = DW<N, l>;
    // step u = (tile pair p, 16-row group s): the two tiles of a pair are independent accumulators whose
    // MFMAs alternate (see chain_gemm); LDS fragment reads run one step ahead of the MFMAs.
    constexpr int NP = (D::T + 1) / 2, U = 4 * NP;
    const int g = lane >> 4, i = lane & 15;
    // LDS addresses as 32-bit: per-lane part once per phase, wave-uniform tile part from the scalar unit, the 16-row group
    // as the instruction's immediate offset: one v_add per fragment pointer (generic pointers cost a 64-bit mad + shift each)
    typedef const float __attribute__((address_space(3))) *lds_cf;
    typedef const v4 __attribute__((address_space(3))) *lds_cv4;
    const lds_cf la = (lds_cf)qdz + (i * kQS + 4 * g), lb = (lds_cf)qx + (i * kQS + 4 * g);
    const lds_cf la_w = la + wave * (16 * kQS);          // tile "wave" of the dZ image
    v4 fa[2][2], fb[2][2];   // [buffer][tile of the pair]
    auto lds_frags = [&](int u, v4 (&a)[2], v4 (&b)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int it = 2 * (u >> 2) + h;
            if (it < D::T) {
                // tile index = wave + 4 it -> (kt, nt) = divmod(.., NT).  When 4 it .. 4 it + 3 stay inside one row of the
                // tile grid, kt does not depend on the wave and nt = (4 it) % NT + wave: both addresses are a per-phase
                // register plus an IMMEDIATE offset (no per-tile address arithmetic at all); otherwise compute them.
                const int c = (4 * it) % D::NT, k0 = (4 * it) / D::NT;
                if (c + 3 < D::NT && 4 * it + 3 < D::TOT) {
                    a[h] = *(lds_cv4)(la_w + c * (16 * kQS) + 16 * (u & 3));
                    b[h] = *(lds_cv4)(lb + k0 * (16 * kQS) + 16 * (u & 3));
                } else {
                    int idx = wave + 4 * it;
                    idx = idx < D::TOT ? idx : D::TOT - 1;
                    const int kt = idx / D::NT, nt = idx - kt * D::NT;
                    a[h] = *(lds_cv4)(la + nt * (16 * kQS) + 16 * (u & 3));
                    b[h] = *(lds_cv4)(lb + kt * (16 * kQS) + 16 * (u & 3));
                }
            }
        }
    };
    lds_frags(0, fa[0], fb[0]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (u + 1 < U) lds_frags(u + 1, fa[(u + 1) & 1], fb[(u + 1) & 1]);
        // pin the reads of step u + 1 in front of the MFMAs of step u: left free, hipcc sank the second tile's reads to one MFMA
        // before their first use (it reuses the registers of the buffer in use).  Measured neutral (298 M rows/s either way:
        // one MFMA covers most of an LDS round trip here); kept because the schedule no longer depends on the allocator's mood
        __builtin_amdgcn_sched_barrier(0);
        const int it0 = 2 * (u >> 2), it1 = it0 + 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[it0] = mfma(fa[u & 1][0][r], fb[u & 1][0][r], acc[it0]);
            if (it1 < D::T) acc[it1] = mfma(fa[u & 1][1][r], fb[u & 1][1][r], acc[it1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <class N, int l>
__device__ __forceinline__ void dw_flush(v4 *__restrict__ slab, const v4 (&acc)[DW<N, l>::T], int lane, int wave) {
    // partial-gradient buffer is TILE-major: [tile][workgroup][64 lanes] (`slab` already points at this workgroup's column):
    // the reduction then streams gridDim.x KiB contiguously per tile instead of striding 298 KiB between workgroups
#pragma unroll
    for (int it = 0; it < DW<N, l>::T; ++it) {
        const int idx = wave + 4 * it;
        if (idx < DW<N, l>::TOT) slab[(int64_t)(N::slab_off(l) + idx) * gridDim.x * 64 + lane] = acc[it];
    }
}

// Image buffers alternate between two LDS regions (A: 240 slot rows, B: 320 slot rows) so that the next
// layer's image writes never touch what a slower wave is still reading: ONE barrier per layer.
constexpr int kImgB = 320;
template <class N> constexpr int img_a_rows();
// the bias fragments sit in LDS behind the images when both fit the 160 KB; the 63-column class reads them from the packed copy (L2)
template <class N> constexpr bool train_bias_in_lds() {
    return (size_t)(img_a_rows<N>() + kImgB) * kQS * sizeof(float) + (size_t)(N::bf_off(8) - N::bf_off(0)) * 16 <= 160 * 1024;
}
template <class N> constexpr int img_a_rows() {     // 240 up to 31 columns, 256 for the 47-column class (its [X_7 | dZ_7] and [X_0 | dZ_0] images)
    int m = 240;
    const int need[] = {DW<N, 7>::rows_x + DW<N, 7>::rows_dz, DW<N, 5>::rows_x + DW<N, 5>::rows_dz, DW<N, 2>::rows_x + DW<N, 2>::rows_dz,
                        DW<N, 0>::rows_x + DW<N, 0>::rows_dz};
    for (int v : need) m = v > m ? v : m;
    return m;
}

template <int F, int Z, bool RT = false>
__global__ void __launch_bounds__(256) train_dec_kernel(const v4 *packed, const void *__restrict__ xin, int in_f64,
                                                        int64_t n, const double *__restrict__ feats, v4 *__restrict__ slabs,
                                                        v4 *__restrict__ dz_out, int fr, int zr) {
    using N = Net<F, Z>;
    using S = StreamTrainDec<N>;
    constexpr int kImgA = img_a_rows<N>();
    static_assert(DW<N, 7>::rows_x + DW<N, 7>::rows_dz <= kImgA && DW<N, 6>::rows_x + DW<N, 6>::rows_dz <= kImgB &&
                  DW<N, 5>::rows_x + DW<N, 5>::rows_dz <= kImgA && DW<N, 4>::rows_x + DW<N, 4>::rows_dz <= kImgB, "image buffers");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *imgA = lds, *imgB = lds + kImgA * kQS;
    const v4 *bias_lds = packed + N::bf_off(0);
    if constexpr (train_bias_in_lds<N>()) {
        v4 *stage = (v4 *)(lds + (kImgA + kImgB) * kQS);
        stage_bias<N>(stage, packed);
        bias_lds = stage;
    }
    int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v4 *slab = slabs + (int64_t)blockIdx.x * 64;   // column of this workgroup in the [tile][workgroup][lane] buffer
    const int64_t ngroups = (n + kRowsPerWG - 1) / kRowsPerWG;
    WStream ws = make_stream(packed + S::start_f4, (N::packed_f4() - S::start_f4) * 16, threadIdx.x & 63);
    double lacc = 0.0;
    v4 g7[DW<N, 7>::T], g6[DW<N, 6>::T], g5[DW<N, 5>::T], g4[DW<N, 4>::T];
    zero_tiles(g7); zero_tiles(g6); zero_tiles(g5); zero_tiles(g4);
#if BAMD_SPLIT == 2
    v4 g3[DW<N, 3>::T], g2[DW<N, 2>::T];
    zero_tiles(g3); zero_tiles(g2);
#endif
    Ring ring;
    ring_prime<S::total>(ring, ws);
    v4 a0n[tiles(F)];   // next row group's input, loaded one iteration ahead (software pipeline)
    {
        const int64_t row0 = (int64_t)blockIdx.x * kRowsPerWG + 16 * wave + (lane & 15);
        load_rows<F, RT>(a0n, xin, in_f64, row0, row0 < n, lane, feats, fr);
    }
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        // keep the weight loads AND the per-tile LDS address arithmetic inside the loop: both are loop
        // invariant, and LICM would hoist hundreds of registers' worth of them (-> scratch spills)
        asm volatile("" : "+v"(ws.voff), "+s"(wave), "+v"(lane));
        const int64_t row = grp * kRowsPerWG + 16 * wave + (lane & 15);
        const bool valid = row < n;
        const int64_t row_next = row + (int64_t)gridDim.x * kRowsPerWG;
        const bool valid_next = row_next < n;
        RawRows<F> xraw;
        v4 a4[tiles(Z)], a5[4], a6[7], a7[13], d8[tiles(F)];
#if BAMD_SPLIT == 2
        v4 a2[7], a3[4];
#endif
        {
#if BAMD_SPLIT == 2
            v4 a0[tiles(F)], a1[13];
#else
            v4 a0[tiles(F)], a1[13], a2[7], a3[4];
#endif
#pragma unroll
            for (int t = 0; t < tiles(F); ++t) a0[t] = a0n[t];
            fwd_layer<N, S, 0>(a0, a1, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 1>(a1, a2, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 2>(a2, a3, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 3>(a3, a4, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 4>(a4, a5, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 5>(a5, a6, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 6>(a6, a7, ring, ws, bias_lds, lane);
            fwd_layer<N, S, 7>(a7, d8, ring, ws, bias_lds, lane);
            // loss and dL/drecon = 2 (r - x)/C  (utils.py:195-199); invalid rows contribute nothing
#pragma unroll
            for (int t = 0; t < tiles(F); ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float d = d8[t][r] - a0[t][r];
                    const bool live = valid && slot_feature(F, t, lane >> 4, r) >= 0;
                    if (live) lacc += (double)d * (double)d;
                    d8[t][r] = live ? d * (2.0f / (float)(RT ? fr : F)) : 0.f;
                }
        }
        // decoder backward: per layer, image writes -> dX chain (registers only) -> barrier -> dW tiles
        v4 d7[13], d6[7], d5[4], d4[tiles(Z)];
        q_write_x<200>(imgA, a7, lane, wave); q_write(imgA + DW<N, 7>::rows_x * kQS, d8, lane, wave);
        bwd_layer<N, S, 7>(d8, d7, ring, ws); lrelu_bwd(d7, a7);
        __syncthreads();
        dw_phase<N, 7>(imgA + DW<N, 7>::rows_x * kQS, imgA, g7, lane, wave);

        q_write_x<100>(imgB, a6, lane, wave); q_write(imgB + DW<N, 6>::rows_x * kQS, d7, lane, wave);
        bwd_layer<N, S, 6>(d7, d6, ring, ws); lrelu_bwd(d6, a6);
        __syncthreads();
        load_rows_issue<F, RT>(xraw, xin, in_f64, row_next, valid_next, lane, fr);   // lands during the longest dW phase
        dw_phase<N, 6>(imgB + DW<N, 6>::rows_x * kQS, imgB, g6, lane, wave);
        load_rows_finish<F, RT>(a0n, xraw, valid_next, lane, feats, fr);

        q_write_x<50>(imgA, a5, lane, wave); q_write(imgA + DW<N, 5>::rows_x * kQS, d6, lane, wave);
        bwd_layer<N, S, 5>(d6, d5, ring, ws); lrelu_bwd(d5, a5);
        __syncthreads();
        dw_phase<N, 5>(imgA + DW<N, 5>::rows_x * kQS, imgA, g5, lane, wave);

        q_write_x<Z>(imgB, a4, lane, wave); q_write(imgB + DW<N, 4>::rows_x * kQS, d5, lane, wave);
        bwd_layer<N, S, 4>(d5, d4, ring, ws);            // en4 has no activation: dL/dz
#if BAMD_SPLIT == 4
        if (valid) dz_out[row * 4 + (lane >> 4)] = d4[0];           // 16 slots per row, slot order
#endif
        __syncthreads();
        dw_phase<N, 4>(imgB + DW<N, 4>::rows_x * kQS, imgB, g4, lane, wave);
#if BAMD_SPLIT == 2
        v4 d3[4], d2[7];
        q_write_x<50>(imgA, a3, lane, wave); q_write(imgA + DW<N, 3>::rows_x * kQS, d4, lane, wave);
        bwd_layer<N, S, 3>(d4, d3, ring, ws); lrelu_bwd(d3, a3);
        __syncthreads();
        dw_phase<N, 3>(imgA + DW<N, 3>::rows_x * kQS, imgA, g3, lane, wave);

        q_write_x<100>(imgB, a2, lane, wave); q_write(imgB + DW<N, 2>::rows_x * kQS, d3, lane, wave);
        bwd_layer<N, S, 2>(d3, d2, ring, ws); lrelu_bwd(d2, a2);
        // dZ_1 hand-off, [16-row tile][t][lane]: 1 KiB contiguous per store.  Rows beyond n store exact zeros (their dL/drecon
        // was zeroed above), so the second kernel loads the record without a validity select.
#pragma unroll
        for (int t = 0; t < 7; ++t) dz_out[((row >> 4) * 7 + t) * 64 + lane] = d2[t];
        __syncthreads();
        dw_phase<N, 2>(imgB + DW<N, 2>::rows_x * kQS, imgB, g2, lane, wave);
#endif
        ring_tail<S::total>(ring, ws);
    }
    dw_flush<N, 7>(slab, g7, lane, wave); dw_flush<N, 6>(slab, g6, lane, wave);
    dw_flush<N, 5>(slab, g5, lane, wave); dw_flush<N, 4>(slab, g4, lane, wave);
#if BAMD_SPLIT == 2
    dw_flush<N, 3>(slab, g3, lane, wave); dw_flush<N, 2>(slab, g2, lane, wave);
#endif
    // per-workgroup loss partial (fixed-order tree)
    __syncthreads();
    double *sh = (double *)lds;
    sh[threadIdx.x] = lacc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) ((double *)(slabs + (int64_t)N::slab_off(N::L) * gridDim.x * 64))[blockIdx.x] = sh[0];   // loss partials after the tiles
}

#if BAMD_SPLIT == 2
// Encoder-gradient kernel, cut after en2: recomputes only en1's forward, receives dZ_1 (7 tiles per row).
template <int F, int Z, bool RT = false>
__global__ void __launch_bounds__(256) train_enc_kernel(const v4 *packed, const void *__restrict__ xin, int in_f64,
                                                        int64_t n, const double *__restrict__ feats, v4 *__restrict__ slabs,
                                                        const v4 *__restrict__ dz_in, int fr, int zr) {
    using N = Net<F, Z>;
    using S = StreamTrainEnc<N>;
    constexpr int kImgA = img_a_rows<N>();
    static_assert(DW<N, 1>::rows_x + DW<N, 1>::rows_dz <= kImgB && DW<N, 0>::rows_x + DW<N, 0>::rows_dz <= kImgA, "image buffers");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *imgA = lds, *imgB = lds + kImgA * kQS;
    const v4 *bias_lds = packed + N::bf_off(0);
    if constexpr (train_bias_in_lds<N>()) {
        v4 *stage = (v4 *)(lds + (kImgA + kImgB) * kQS);
        stage_bias<N>(stage, packed);
        bias_lds = stage;
    }
    int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v4 *slab = slabs + (int64_t)blockIdx.x * 64;   // column of this workgroup in the [tile][workgroup][lane] buffer
    const int64_t ngroups = (n + kRowsPerWG - 1) / kRowsPerWG;
    WStream ws = make_stream(packed + S::start_f4, (N::packed_f4() - S::start_f4) * 16, threadIdx.x & 63);
    v4 g1[DW<N, 1>::T], g0[DW<N, 0>::T];
    zero_tiles(g1); zero_tiles(g0);
    Ring ring;
    ring_prime<S::total>(ring, ws);
    v4 a0n[tiles(F)], d2n[7];   // next row group's inputs, loaded one iteration ahead (software pipeline)
    {
        const int64_t row0 = (int64_t)blockIdx.x * kRowsPerWG + 16 * wave + (lane & 15);
        load_rows<F, RT>(a0n, xin, in_f64, row0, row0 < n, lane, feats, fr);
#pragma unroll
        for (int t = 0; t < 7; ++t) d2n[t] = dz_in[((row0 >> 4) * 7 + t) * 64 + lane];
    }
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        asm volatile("" : "+v"(ws.voff), "+s"(wave), "+v"(lane));   // see train_dec_kernel
        const int64_t row = grp * kRowsPerWG + 16 * wave + (lane & 15);
        const int64_t row_next = row + (int64_t)gridDim.x * kRowsPerWG;
        const bool valid_next = row_next < n;
        RawRows<F> xraw;
        v4 a0[tiles(F)], a1[13], d2[7], d1[13];
#pragma unroll
        for (int t = 0; t < tiles(F); ++t) a0[t] = a0n[t];
#pragma unroll
        for (int t = 0; t < 7; ++t) d2[t] = d2n[t];
        fwd_layer<N, S, 0>(a0, a1, ring, ws, bias_lds, lane);

        q_write_x<200>(imgB, a1, lane, wave); q_write(imgB + DW<N, 1>::rows_x * kQS, d2, lane, wave);
        bwd_layer<N, S, 1>(d2, d1, ring, ws); lrelu_bwd(d1, a1);
        __syncthreads();
        load_rows_issue<F, RT>(xraw, xin, in_f64, row_next, valid_next, lane, fr);   // lands during the long dW phase
#pragma unroll
        for (int t = 0; t < 7; ++t) d2n[t] = dz_in[((row_next >> 4) * 7 + t) * 64 + lane];   // one round past the end stays inside the buffer
        dw_phase<N, 1>(imgB + DW<N, 1>::rows_x * kQS, imgB, g1, lane, wave);
        load_rows_finish<F, RT>(a0n, xraw, valid_next, lane, feats, fr);

        q_write_x<F>(imgA, a0, lane, wave); q_write(imgA + DW<N, 0>::rows_x * kQS, d1, lane, wave);
        __syncthreads();
        dw_phase<N, 0>(imgA + DW<N, 0>::rows_x * kQS, imgA, g0, lane, wave);
        ring_tail<S::total>(ring, ws);
    }
    dw_flush<N, 1>(slab, g1, lane, wave); dw_flush<N, 0>(slab, g0, lane, wave);
}
#else
template <int F, int Z, bool RT = false>
__global__ void __launch_bounds__(256) train_enc_kernel(const v4 *packed, const void *__restrict__ xin, int in_f64,
                                                        int64_t n, const double *__restrict__ feats, v4 *__restrict__ slabs,
                                                        const v4 *__restrict__ dz_in, int fr, int zr) {
    using N = Net<F, Z>;
    using S = StreamTrainEnc<N>;
    static_assert(Z <= 16, "dL/dz hand-off is one tile per row");
    constexpr int kImgA = img_a_rows<N>();
    static_assert(DW<N, 3>::rows_x + DW<N, 3>::rows_dz <= kImgB && DW<N, 2>::rows_x + DW<N, 2>::rows_dz <= kImgA &&
                  DW<N, 1>::rows_x + DW<N, 1>::rows_dz <= kImgB && DW<N, 0>::rows_x + DW<N, 0>::rows_dz <= kImgA, "image buffers");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *imgA = lds, *imgB = lds + kImgA * kQS;   // layers 3,1 -> B ; layers 2,0 -> A : one barrier per layer
    const v4 *bias_lds = packed + N::bf_off(0);
    if constexpr (train_bias_in_lds<N>()) {
        v4 *stage = (v4 *)(lds + (kImgA + kImgB) * kQS);
        stage_bias<N>(stage, packed);
        bias_lds = stage;
    }
    int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v4 *slab = slabs + (int64_t)blockIdx.x * 64;   // column of this workgroup in the [tile][workgroup][lane] buffer
    const int64_t ngroups = (n + kRowsPerWG - 1) / kRowsPerWG;
    WStream ws = make_stream(packed + S::start_f4, (N::packed_f4() - S::start_f4) * 16, threadIdx.x & 63);
    v4 g3[DW<N, 3>::T], g2[DW<N, 2>::T], g1[DW<N, 1>::T], g0[DW<N, 0>::T];
    zero_tiles(g3); zero_tiles(g2); zero_tiles(g1); zero_tiles(g0);
    Ring ring;
    ring_prime<S::total>(ring, ws);
    v4 a0n[tiles(F)], d4n;   // next row group's inputs, loaded one iteration ahead (software pipeline)
    {
        const int64_t row0 = (int64_t)blockIdx.x * kRowsPerWG + 16 * wave + (lane & 15);
        load_rows<F, RT>(a0n, xin, in_f64, row0, row0 < n, lane, feats, fr);
        d4n = row0 < n ? dz_in[row0 * 4 + (lane >> 4)] : (v4){0.f, 0.f, 0.f, 0.f};
    }
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        asm volatile("" : "+v"(ws.voff), "+s"(wave), "+v"(lane));   // see train_dec_kernel
        const int64_t row = grp * kRowsPerWG + 16 * wave + (lane & 15);
        const int64_t row_next = row + (int64_t)gridDim.x * kRowsPerWG;
        const bool valid_next = row_next < n;
        RawRows<F> xraw;
        v4 a0[tiles(F)], a1[13], a2[7], a3[4], d4[tiles(Z)];
#pragma unroll
        for (int t = 0; t < tiles(F); ++t) a0[t] = a0n[t];
        d4[0] = d4n;
        fwd_layer<N, S, 0>(a0, a1, ring, ws, bias_lds, lane);
        fwd_layer<N, S, 1>(a1, a2, ring, ws, bias_lds, lane);
        fwd_layer<N, S, 2>(a2, a3, ring, ws, bias_lds, lane);
        // (z itself is not needed again: en4's weight gradient uses a3 and dL/dz)
        v4 d3[4], d2[7], d1[13];
        q_write_x<50>(imgB, a3, lane, wave); q_write(imgB + DW<N, 3>::rows_x * kQS, d4, lane, wave);
        bwd_layer<N, S, 3>(d4, d3, ring, ws); lrelu_bwd(d3, a3);
        __syncthreads();
        dw_phase<N, 3>(imgB + DW<N, 3>::rows_x * kQS, imgB, g3, lane, wave);

        q_write_x<100>(imgA, a2, lane, wave); q_write(imgA + DW<N, 2>::rows_x * kQS, d3, lane, wave);
        bwd_layer<N, S, 2>(d3, d2, ring, ws); lrelu_bwd(d2, a2);
        __syncthreads();
        dw_phase<N, 2>(imgA + DW<N, 2>::rows_x * kQS, imgA, g2, lane, wave);

        q_write_x<200>(imgB, a1, lane, wave); q_write(imgB + DW<N, 1>::rows_x * kQS, d2, lane, wave);
        bwd_layer<N, S, 1>(d2, d1, ring, ws); lrelu_bwd(d1, a1);
        __syncthreads();
        load_rows_issue<F, RT>(xraw, xin, in_f64, row_next, valid_next, lane, fr);   // lands during the longest dW phase
        v4 dzraw = valid_next ? dz_in[row_next * 4 + (lane >> 4)] : (v4){0.f, 0.f, 0.f, 0.f};
        dw_phase<N, 1>(imgB + DW<N, 1>::rows_x * kQS, imgB, g1, lane, wave);
        load_rows_finish<F, RT>(a0n, xraw, valid_next, lane, feats, fr);
        d4n = dzraw;

        q_write_x<F>(imgA, a0, lane, wave); q_write(imgA + DW<N, 0>::rows_x * kQS, d1, lane, wave);
        __syncthreads();
        dw_phase<N, 0>(imgA + DW<N, 0>::rows_x * kQS, imgA, g0, lane, wave);
        ring_tail<S::total>(ring, ws);
    }
    dw_flush<N, 3>(slab, g3, lane, wave); dw_flush<N, 2>(slab, g2, lane, wave);
    dw_flush<N, 1>(slab, g1, lane, wave); dw_flush<N, 0>(slab, g0, lane, wave);
}

#endif

// ---- small-batch kernels (the reference's batch_size = 512 regime; used up to FusedState::latency_max_rows) -------
// The throughput kernels give a whole 16-row chain to ONE wave: a 512-row batch occupies 32 waves for ~55 us
// whatever the chip size.  Here the step is two launches:
//   lat2_chain_kernel: a workgroup owns ONE 16-row block and its W waves split every layer's OUTPUT tiles (tile t ->
//     wave t mod W); after each layer the waves swap their tiles through a double-buffered LDS exchange (one barrier
//     per layer) so that each wave again holds the full input of the next layer.  Per-layer critical path =
//     ceil(NT/W) tiles instead of NT.  Waves with fewer tiles recompute their last tile (identical instruction
//     streams, no divergence).  The X^T / dZ^T images of all layers go to global memory ([16-row block][slot][16
//     rows], 104 KiB per block).
//   lat2_dw_kernel: one workgroup per weight-gradient TILE (298 of them: the whole chip, not 32 CUs), contracting
//     over ALL rows of the batch (fixed order: wave w takes blocks w, w+4, ..; then waves 0..3), optionally fused
//     with the Adam update of exactly those 256 parameters and the refresh of their packed copies.
// History (profiles/README.md): the first version kept the images in LDS, formed every [dW | db] tile per workgroup
// with 4-step MFMAs, wrote a 247-KiB slab per workgroup and reduced the slabs in a second kernel, with an 8-deep
// fragment ring: 28 + 6 + 5 us per 512-row step against 17 + 6 us now.
template <class N> struct Lat {
    __host__ __device__ static constexpr int x_rows(int l) { return 16 * tiles(N::dim(l) + 1); }
    __host__ __device__ static constexpr int z_rows(int l) { return 16 * tiles(N::dim(l + 1)); }
    __host__ __device__ static constexpr int x_off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += x_rows(j); return s; }
    __host__ __device__ static constexpr int z_off(int l) { int s = x_off(N::L); for (int j = 0; j < l; ++j) s += z_rows(j); return s; }
    static constexpr int xch_f4 = 13 * 64;                       // one exchange buffer: up to 13 tiles
};


template <int NT> __device__ __forceinline__ void lat_collect(const v4 *xch, v4 (&all)[NT], int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t) all[t] = xch[t * 64 + lane];
}

constexpr int kImgStride = 16;   // floats per slot in the global images (= rows per block)

#ifdef BAMD_LAT_TRACE   // debug build: shader-clock stamps of workgroup 0 at every layer boundary (tools/lat_trace.py)
__device__ unsigned long long g_lat_trace[64];
#define LAT_T(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_lat_trace[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define LAT_T(i) do {} while (0)
#endif

// The 15 chain GEMMs of one training step as ONE fragment sequence per wave (forward layers 0..7, then the
// transposed fragments of layers 7..1): a register ring of D fragments runs D fragments AHEAD of the MFMAs across
// layer boundaries, one refill per consumed fragment, so the loads are spread between the MFMAs.  (Requesting whole
// layers at once stalls the wave: a 1-KiB fragment load occupies the CU's 64 B/clk L1 path for 16 cycles and a
// wave issues in order, so 28 loads x 4 waves in a row = 1800 cycles before the first MFMA of the layer.)
template <class N, int W, int D_> struct LatSeq {
    static constexpr int D = D_, Wv = W, NG = 15;
    __host__ __device__ static constexpr int layer(int g) { return g < 8 ? g : 15 - g; }              // 0..7, 7..1
    __host__ __device__ static constexpr int kd(int g) { return g < 8 ? N::dim(g) : N::dim(layer(g) + 1); }
    __host__ __device__ static constexpr int nt(int g) { return tiles(g < 8 ? N::dim(g + 1) : N::dim(layer(g))); }
    __host__ __device__ static constexpr int base(int g) { return (g < 8 ? N::wf_off(g) : N::wb_off(layer(g))) / 64; }
    __host__ __device__ static constexpr int nl(int g) { return (nt(g) + W - 1) / W; }
    __host__ __device__ static constexpr int nf(int g) { return tiles(kd(g)) * nl(g); }
    __host__ __device__ static constexpr int start(int g) { int s = 0; for (int j = 0; j < g; ++j) s += nf(j); return s; }
    static constexpr int total = start(NG);
    __host__ __device__ static constexpr int gemm_of(int G) { int g = 0; for (int j = 1; j < NG; ++j) if (G >= start(j)) g = j; return g; }
};
// (every index below is a template constant: with loop variables hipcc left the ring in scratch memory)
template <class SQ, int G>
__device__ __forceinline__ void seq_issue(v4 (&slot)[SQ::D], const WStream &ws, int wave) {
    if constexpr (G < SQ::total) {
        constexpr int g = SQ::gemm_of(G), f = G - SQ::start(g), NL = SQ::nl(g), q = f / NL, i = f % NL, NT = SQ::nt(g);
        int t = wave + SQ::Wv * i;
        t = t < NT ? t : NT - 1;
        slot[G % SQ::D] = frag_rt(ws, SQ::base(g) + q * NT + t);
    }
}
template <class SQ, int... G>
__device__ __forceinline__ void seq_prologue(v4 (&slot)[SQ::D], const WStream &ws, int wave, std::integer_sequence<int, G...>) {
    (seq_issue<SQ, G>(slot, ws, wave), ...);
    __builtin_amdgcn_sched_barrier(0);
}
// fragments f and f+1 of GEMM g feed two interleaved accumulators; each consumed fragment is replaced by the one D ahead
template <class SQ, int g, int f>
__device__ __forceinline__ void seq_pair(const v4 (&in)[tiles(SQ::kd(g))], v4 (&out)[SQ::nl(g)], v4 (&slot)[SQ::D],
                                         const WStream &ws, int wave) {
    constexpr int NL = SQ::nl(g), NF = SQ::nf(g), S0 = SQ::start(g), KD = SQ::kd(g);
    constexpr bool two = f + 1 < NF;
    constexpr int q0 = f / NL, i0 = f % NL, q1 = two ? (f + 1) / NL : q0, i1 = two ? (f + 1) % NL : i0;
    constexpr int s0 = (S0 + f) % SQ::D, s1 = (S0 + f + (two ? 1 : 0)) % SQ::D;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r < tile_steps(KD, q0)) out[i0] = mfma(slot[s0][r], in[q0][r], out[i0]);
        if (two && r < tile_steps(KD, q1)) out[i1] = mfma(slot[s1][r], in[q1][r], out[i1]);
    }
    seq_issue<SQ, S0 + f + SQ::D>(slot, ws, wave);
    if constexpr (two) seq_issue<SQ, S0 + f + 1 + SQ::D>(slot, ws, wave);
    __builtin_amdgcn_sched_barrier(0);
}
template <class SQ, int g, int... P>
__device__ __forceinline__ void seq_mm_impl(const v4 (&in)[tiles(SQ::kd(g))], v4 (&out)[SQ::nl(g)], v4 (&slot)[SQ::D],
                                            const WStream &ws, int wave, std::integer_sequence<int, P...>) {
    (seq_pair<SQ, g, 2 * P>(in, out, slot, ws, wave), ...);
}
// GEMM g of the sequence: out[i] += frags(q, tile wave + W i) . in[q]
template <class SQ, int g>
__device__ __forceinline__ void seq_mm(const v4 (&in)[tiles(SQ::kd(g))], v4 (&out)[SQ::nl(g)], v4 (&slot)[SQ::D], const WStream &ws,
                                       int wave) {
    seq_mm_impl<SQ, g>(in, out, slot, ws, wave, std::make_integer_sequence<int, (SQ::nf(g) + 1) / 2>{});
}
// own tiles -> LDS exchange buffer (C layout) and -> the global [slot][16 rows] image through a buffer resource
// (per-lane offset once, slot offsets as SGPR/immediate: no address arithmetic per store).  Measured: 16-byte stores
// of C-layout tiles save 0.3 us here but cost lat2_dw_kernel 1.2 us of transposing 4-byte loads.
template <int D, bool ONES, int W>
__device__ __forceinline__ void lat2_publish(v4 *xch, __amdgpu_buffer_rsrc_t irs, int slot0, const v4 (&loc)[(tiles(D) + W - 1) / W],
                                             int lane, int wave) {
    constexpr int NT = tiles(D), NL = (NT + W - 1) / W;
    constexpr int T1 = tiles(D) - 1, V = D - 16 * T1, R1 = V / 4, G1 = V % 4;
    const int g = lane >> 4, col = lane & 15;
    const int voff = (4 * g * kImgStride + col) * 4;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int t = wave + W * i;
        if (t < NT) {
            if (xch) xch[t * 64 + lane] = loc[i];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = loc[i][r];
                if (ONES && t == T1 && r == R1 && g == G1) v = 1.0f;      // the ones slot that carries db
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), irs, voff + r * kImgStride * 4,
                                                      (slot0 + 16 * t) * kImgStride * 4, 0);
            }
        }
    }
}

template <int F, int Z, int W, bool RT = false>
__global__ void __launch_bounds__(64 * W) lat2_chain_kernel(const v4 *packed, const void *__restrict__ xin, int in_f64, int64_t n,
                                                            const double *__restrict__ feats, float *__restrict__ imgs,
                                                            double *__restrict__ loss_part, int fr, int zr) {
    using N = Net<F, Z>;
    using LT = Lat<N>;
    constexpr int kImgFloats = LT::z_off(N::L) * kImgStride;
    __shared__ __attribute__((aligned(16))) v4 xch_lds[2 * LT::xch_f4 + (N::bf_off(N::L) - N::bf_off(0))];
    __shared__ double loss_lds[W];
    v4 *xchA = xch_lds, *xchB = xchA + LT::xch_f4, *bias_lds = xchB + LT::xch_f4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), g = lane >> 4;
    WStream ws = make_stream(packed, N::packed_f4() * 16, lane);
    constexpr int TF = tiles(F), TZ = tiles(Z);
    static_assert(TF <= 8 && TZ <= 2, "input / latent tiles");
    LAT_T(0);
    // the rows first (layer 0 waits for them), then the ring's first D fragments queue up behind them
    const int64_t row = (int64_t)blockIdx.x * 16 + (lane & 15);
    const bool valid = row < n;
    RawRows<F> xraw;
    load_rows_issue<F, RT>(xraw, xin, in_f64, row, valid, lane, fr);
    using SQ = LatSeq<N, W, (W == 4 ? 32 : 24)>;
    v4 ring[SQ::D];
    constexpr int kNB = N::bf_off(N::L) - N::bf_off(0), kBV = (kNB + 64 * W - 1) / (64 * W);
    v4 bv[kBV];                                   // bias fragments: requested with the rows, not after them
#pragma unroll
    for (int k = 0; k < kBV; ++k) {
        const int idx = (int)threadIdx.x + k * 64 * W;
        bv[k] = idx < kNB ? packed[N::bf_off(0) + idx] : (v4){0.f, 0.f, 0.f, 0.f};
    }
    seq_prologue<SQ>(ring, ws, wave, std::make_integer_sequence<int, SQ::D>{});
    v4 a0[TF];
    load_rows_finish<F, RT>(a0, xraw, valid, lane, feats, fr);
#pragma unroll
    for (int k = 0; k < kBV; ++k) {
        const int idx = (int)threadIdx.x + k * 64 * W;
        if (idx < kNB) bias_lds[idx] = bv[k];
    }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void *)(imgs + (int64_t)blockIdx.x * kImgFloats), 0,
                                                                         kImgFloats * 4, 0x00020000);
#define LAT2_BIAS(loc, l, NTl)                                                                           \
    _Pragma("unroll") for (int i = 0; i < (NTl + W - 1) / W; ++i) {                                      \
        int t_ = wave + W * i; t_ = t_ < NTl ? t_ : NTl - 1;                                             \
        loc[i] = bias_lds[(N::bf_off(l) - N::bf_off(0)) + t_ * 4 + g];                                   \
    }
    LAT_T(1);
    constexpr int L13 = (13 + W - 1) / W, L7 = (7 + W - 1) / W, L4 = (4 + W - 1) / W, LF = (TF + W - 1) / W;
    // ---------------- forward ----------------
    {   // X image of layer 0 (= the input): every wave holds all of a0, wave t publishes tile t
        v4 own[LF];
#pragma unroll
        for (int i = 0; i < LF; ++i) own[i] = a0[(wave + W * i) < TF ? (wave + W * i) : TF - 1];
        lat2_publish<F, true, W>(nullptr, irs, LT::x_off(0), own, lane, wave);
    }
    v4 s1[L13], s2[L7], s3[L4], s4[1], s5[L4], s6[L7], s7[L13], o8[LF];
    LAT2_BIAS(s1, 0, 13) seq_mm<SQ, 0>(a0, s1, ring, ws, wave); lrelu(s1);
    lat2_publish<200, true, W>(xchA, irs, LT::x_off(1), s1, lane, wave);
    __syncthreads();
    LAT_T(2);
    {
        v4 a1[13]; lat_collect(xchA, a1, lane);
        LAT2_BIAS(s2, 1, 7) seq_mm<SQ, 1>(a1, s2, ring, ws, wave); lrelu(s2);
    }
    lat2_publish<100, true, W>(xchB, irs, LT::x_off(2), s2, lane, wave);
    __syncthreads();
    LAT_T(3);
    {
        v4 a2[7]; lat_collect(xchB, a2, lane);
        LAT2_BIAS(s3, 2, 4) seq_mm<SQ, 2>(a2, s3, ring, ws, wave); lrelu(s3);
    }
    lat2_publish<50, true, W>(xchA, irs, LT::x_off(3), s3, lane, wave);
    __syncthreads();
    LAT_T(4);
    {
        v4 a3[4]; lat_collect(xchA, a3, lane);
        LAT2_BIAS(s4, 3, TZ) seq_mm<SQ, 3>(a3, s4, ring, ws, wave);                       // en4: no activation
    }
    lat2_publish<Z, true, W>(xchB, irs, LT::x_off(4), s4, lane, wave);
    __syncthreads();
    LAT_T(5);
    {
        v4 a4[TZ]; lat_collect(xchB, a4, lane);
        LAT2_BIAS(s5, 4, 4) seq_mm<SQ, 4>(a4, s5, ring, ws, wave); lrelu(s5);
    }
    lat2_publish<50, true, W>(xchA, irs, LT::x_off(5), s5, lane, wave);
    __syncthreads();
    LAT_T(6);
    {
        v4 a5[4]; lat_collect(xchA, a5, lane);
        LAT2_BIAS(s6, 5, 7) seq_mm<SQ, 5>(a5, s6, ring, ws, wave); lrelu(s6);
    }
    lat2_publish<100, true, W>(xchB, irs, LT::x_off(6), s6, lane, wave);
    __syncthreads();
    LAT_T(7);
    {
        v4 a6[7]; lat_collect(xchB, a6, lane);
        LAT2_BIAS(s7, 6, 13) seq_mm<SQ, 6>(a6, s7, ring, ws, wave); lrelu(s7);
    }
    lat2_publish<200, true, W>(xchA, irs, LT::x_off(7), s7, lane, wave);
    __syncthreads();
    LAT_T(8);
    {
        v4 a7[13]; lat_collect(xchA, a7, lane);
        LAT2_BIAS(o8, 7, TF) seq_mm<SQ, 7>(a7, o8, ring, ws, wave);                       // de4: no activation
    }
    // ---------------- loss, dL/drecon ----------------
    double lacc = 0.0;
#pragma unroll
    for (int i = 0; i < LF; ++i) {                // this wave's recon tiles wave, wave + W, ... (one up to 63 columns, two in the 79-column class)
        const int t = wave + W * i, tc = t < TF ? t : TF - 1;
        v4 x0 = a0[0];
#pragma unroll
        for (int u = 1; u < TF; ++u) x0 = tc == u ? a0[u] : x0;       // (register select: a run-time index would put a0 in scratch)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float d = o8[i][r] - x0[r];
            bool live = false;
#pragma unroll
            for (int u = 0; u < TF; ++u) live = live || (tc == u && slot_feature(F, u, g, r) >= 0);
            live = live && valid && t < TF;
            if (live) lacc += (double)d * (double)d;
            o8[i][r] = live ? d * (2.0f / (float)(RT ? fr : F)) : 0.f;
        }
    }
    // ---------------- backward chain (input gradients), publishing dZ images ----------------
    lat2_publish<F, false, W>(xchB, irs, LT::z_off(7), o8, lane, wave);
    __syncthreads();
    LAT_T(9);
    {
        v4 d8[TF]; lat_collect(xchB, d8, lane);
        v4 dx[L13]; zero_tiles(dx);
        seq_mm<SQ, 8>(d8, dx, ring, ws, wave); lrelu_bwd(dx, s7);
#pragma unroll
        for (int i = 0; i < L13; ++i) s7[i] = dx[i];          // s7 now holds dZ_6 (own tiles)
    }
    lat2_publish<200, false, W>(xchA, irs, LT::z_off(6), s7, lane, wave);
    __syncthreads();
    LAT_T(10);
    {
        v4 d7[13]; lat_collect(xchA, d7, lane);
        v4 dx[L7]; zero_tiles(dx);
        seq_mm<SQ, 9>(d7, dx, ring, ws, wave); lrelu_bwd(dx, s6);
#pragma unroll
        for (int i = 0; i < L7; ++i) s6[i] = dx[i];
    }
    lat2_publish<100, false, W>(xchB, irs, LT::z_off(5), s6, lane, wave);
    __syncthreads();
    LAT_T(11);
    {
        v4 d6[7]; lat_collect(xchB, d6, lane);
        v4 dx[L4]; zero_tiles(dx);
        seq_mm<SQ, 10>(d6, dx, ring, ws, wave); lrelu_bwd(dx, s5);
#pragma unroll
        for (int i = 0; i < L4; ++i) s5[i] = dx[i];
    }
    lat2_publish<50, false, W>(xchA, irs, LT::z_off(4), s5, lane, wave);
    __syncthreads();
    LAT_T(12);
    {
        v4 d5[4]; lat_collect(xchA, d5, lane);
        v4 dx[1]; zero_tiles(dx);
        seq_mm<SQ, 11>(d5, dx, ring, ws, wave);                                            // dL/dz: en4 has no activation
        s4[0] = dx[0];
    }
    lat2_publish<Z, false, W>(xchB, irs, LT::z_off(3), s4, lane, wave);
    __syncthreads();
    LAT_T(13);
    {
        v4 d4[TZ]; lat_collect(xchB, d4, lane);
        v4 dx[L4]; zero_tiles(dx);
        seq_mm<SQ, 12>(d4, dx, ring, ws, wave); lrelu_bwd(dx, s3);
#pragma unroll
        for (int i = 0; i < L4; ++i) s3[i] = dx[i];
    }
    lat2_publish<50, false, W>(xchA, irs, LT::z_off(2), s3, lane, wave);
    __syncthreads();
    LAT_T(14);
    {
        v4 d3[4]; lat_collect(xchA, d3, lane);
        v4 dx[L7]; zero_tiles(dx);
        seq_mm<SQ, 13>(d3, dx, ring, ws, wave); lrelu_bwd(dx, s2);
#pragma unroll
        for (int i = 0; i < L7; ++i) s2[i] = dx[i];
    }
    lat2_publish<100, false, W>(xchB, irs, LT::z_off(1), s2, lane, wave);
    __syncthreads();
    LAT_T(15);
    {
        v4 d2[7]; lat_collect(xchB, d2, lane);
        v4 dx[L13]; zero_tiles(dx);
        seq_mm<SQ, 14>(d2, dx, ring, ws, wave); lrelu_bwd(dx, s1);
#pragma unroll
        for (int i = 0; i < L13; ++i) s1[i] = dx[i];
    }
    lat2_publish<200, false, W>(nullptr, irs, LT::z_off(0), s1, lane, wave);
#undef LAT2_BIAS
    // loss partial of this 16-row block: lanes of a wave, then waves 0..W-1 (fixed order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lacc += __shfl_down(lacc, off);
    if (lane == 0) loss_lds[wave] = lacc;
    __syncthreads();
    LAT_T(16);
    if (threadIdx.x == 0) {
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < W; ++w) sum += loss_lds[w];
        loss_part[blockIdx.x] = sum;
    }
}

// ---- 4-row small-batch chain ---------------------------------------------------------------------------------------------
// lat2_chain_kernel gives a 16-row block (the smallest 16x16x4 MFMA batch tile) to one CU: the reference's 512-row batch
// occupies 32 of 256 CUs, and the 15 chain GEMMs of a block are strictly sequential: 634 MFMAs x 32 cycles per wave = 9.7 us
// whatever the chip size.  v_mfma_f32_4x4x1_16B_f32 computes 16 independent 4 x 4 outer products per instruction at the same
// 64 FLOP/clk (tools/probe/mfma4_probe.hip: lane 4 b + i holds A_b[i], lane 4 b + j holds B_b[j], lane 4 b + j register i
// receives D_b[i][j]; 10 cycles per instruction on independent accumulators, 14 on one): with A = 64 output features (lane =
// feature), B = the 4 batch rows replicated over the 16 blocks and one contraction index per instruction, a workgroup needs
// only FOUR rows -- 128 CUs carry the 512-row batch and a workgroup's chain is a quarter of the MFMA work.
//   * what bounds it is the WEIGHT STREAM, not the MFMAs: every workgroup reads all weights once, and one wave takes in at most
//     ~18 B/clk of vector-memory loads (measured: 58 cycles per 1-KiB fragment and wave whatever the ring depth; the CU's L1
//     path gives 64 B/clk).  So ALL FOUR waves stream in EVERY GEMM: the 64-feature output groups of a GEMM (200 outputs = 4,
//     100 = 2, <= 64 = 1) times P = 4 / groups contiguous K ranges; P > 1 costs a reduction through LDS and a second barrier
//     (the first version gave a group's whole K to one wave: its 1- and 2-group GEMMs ran at 18 / 36 B/clk, chain 11.7 us);
//   * the fragments are stored [k / 4][feature] WITHOUT padding the features to 64: lanes beyond the last feature (and whole
//     steps beyond a wave's K range) load through an out-of-range offset -- zeros, no traffic: 482 instead of 634 KB per workgroup;
//   * activations live in LDS as [4 rows][features] float32 (X_l for the masks of the backward pass, dZ_l); the B operand of
//     four consecutive contraction steps is ONE ds_read_b128 (all 16 blocks read the same row: a broadcast);
//   * outputs go to the SAME global images as lat2_chain_kernel's ([16-row block][slot][16 rows]; this workgroup fills rows
//     4 q .. 4 q + 3 of every slot), so lat2_dw_kernel (weight-gradient tiles + Adam) is unchanged.
#ifndef BAMD_L4_RING
#define BAMD_L4_RING 16
#endif
template <class N> struct L4 {
    static constexpr int NG = 15, D = BAMD_L4_RING;                  // chain GEMMs; ring depth (fragments = 4 MFMAs each)
    __host__ __device__ static constexpr int layer(int g) { return g < 8 ? g : 15 - g; }
    __host__ __device__ static constexpr int groups(int g) { return N::l4_groups(g); }
    __host__ __device__ static constexpr int parts(int g) { return 4 / groups(g); }
    __host__ __device__ static constexpr int ks(int g) { return N::l4_ks(g); }
    __host__ __device__ static constexpr int ksp(int g) { return (ks(g) + parts(g) - 1) / parts(g); }             // steps per wave
    __host__ __device__ static constexpr int pos(int g) { int s = 0; for (int j = 0; j < g; ++j) s += ksp(j); return s; }
    static constexpr int total = pos(NG);
    __host__ __device__ static constexpr int gemm_at(int S) { int g = 0; for (int j = 0; j < NG; ++j) if (S >= pos(j)) g = j; return g; }
    // LDS images, float offsets; row strides 64 x groups + 4 (the four rows of a B read then sit in four different bank groups)
    __host__ __device__ static constexpr int xs(int l) { return (l == 0 ? 32 : 64 * groups(l - 1)) + 4; }          // X_l, l = 0..7
    __host__ __device__ static constexpr int zs(int l) { return (l == 7 ? 64 : 64 * groups(14 - l)) + 4; }         // dZ_l, l = 0..7
    __host__ __device__ static constexpr int xo(int l) { int s = 0; for (int j = 0; j < l; ++j) s += 4 * xs(j); return s; }
    __host__ __device__ static constexpr int zo(int l) { int s = xo(8); for (int j = 0; j < l; ++j) s += 4 * zs(j); return s; }
    static constexpr int ps = 132;                                   // partial sums: [part][row][<= 128 features + 4]
    static constexpr int po = zo(8);
    static constexpr int bo = po + 4 * 4 * ps;                       // biases of the 8 layers, 256 slots each (zeros beyond the layer's width)
    static constexpr int lds_floats = bo + 8 * 256;
    static_assert(groups(0) == 4 && groups(1) == 2 && groups(6) == 4 && groups(14) == 4, "K is split only for GEMMs of 1 or 2 groups");
};
__device__ __forceinline__ v4 mfma4(float a, float b, v4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

template <class N, int S>
__device__ __forceinline__ void l4_issue(v4 (&ring)[L4<N>::D], const WStream &ws, const int (&voff)[15], const int (&k0)[15]) {
    using T = L4<N>;
    if constexpr (S < T::total) {
        constexpr int g = T::gemm_at(S), i = S - T::pos(g);
        const int k4 = k0[g] + i;                                                   // wave-uniform
        const int vo = k4 < T::ks(g) ? voff[g] : 0x7F000000;                        // beyond this GEMM's K: a zero fragment
        ring[S % T::D] = __builtin_bit_cast(v4, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, vo, (N::l4_frag_off(g) + k4 * N::l4_gemm_n(g)) * 16, 0));
    }
}
template <class N, int... S>
__device__ __forceinline__ void l4_prologue(v4 (&ring)[L4<N>::D], const WStream &ws, const int (&voff)[15], const int (&k0)[15], std::integer_sequence<int, S...>) {
    (l4_issue<N, S>(ring, ws, voff, k0), ...);
    __builtin_amdgcn_sched_barrier(0);
}
constexpr int kL4B = 4;        // B operand reads run this many steps (of 4 MFMAs) ahead
template <class N, int g, int I>
__device__ __forceinline__ void l4_step(v4 (&acc)[4], v4 (&xb)[kL4B + 1], const float *brow, v4 (&ring)[L4<N>::D], const WStream &ws,
                                        const int (&voff)[15], const int (&k0)[15]) {
    using T = L4<N>;
    constexpr int S0 = T::pos(g), KSP = T::ksp(g);
    if constexpr (I + kL4B < KSP) xb[(I + kL4B) % (kL4B + 1)] = *(const v4 *)(brow + 4 * (I + kL4B));
    const v4 a = ring[(S0 + I) % T::D], b = xb[I % (kL4B + 1)];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = mfma4(a[r], b[r], acc[r]);     // four accumulators: a dependent 4x4x1 costs 14 cycles, not 10
    l4_issue<N, S0 + I + T::D>(ring, ws, voff, k0);
    __builtin_amdgcn_sched_barrier(0);
}
template <class N, int g, int... I>
__device__ __forceinline__ v4 l4_gemm(v4 init, const float *brow /* B row of this lane at this wave's first step */, v4 (&ring)[L4<N>::D],
                                      const WStream &ws, const int (&voff)[15], const int (&k0)[15], std::integer_sequence<int, I...>) {
    v4 acc[4] = {init, (v4){0.f, 0.f, 0.f, 0.f}, (v4){0.f, 0.f, 0.f, 0.f}, (v4){0.f, 0.f, 0.f, 0.f}};
    v4 xb[kL4B + 1];
#pragma unroll
    for (int k = 0; k < kL4B && k < L4<N>::ksp(g); ++k) xb[k] = *(const v4 *)(brow + 4 * k);
    (l4_step<N, g, I>(acc, xb, brow, ring, ws, voff, k0), ...);
    return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
// image slot (x 64 bytes) of feature f of an image of width D, or out of range: full tiles 16 t + i; the partial last tile is
// r-major (slot_feature): feature index i within the tile sits in slot 4 (i & 3) + (i >> 2)
template <int D> __device__ __forceinline__ int l4_slot_bytes(int f) {
    constexpr int T1 = tiles(D) - 1;
    static_assert(D % 16 != 0, "the last image tile is a partial one (the ones slot lives there)");
    const int t = f >> 4, i = f & 15;
    const int s = t < T1 ? f : 16 * t + 4 * (i & 3) + (i >> 2);
    return t <= T1 ? s * 64 : 0x7F000000;
}
// The 64 outputs of one group held in the accumulator layout (lane (b, j): features 64 grp + 4 b .. + 3 of row j) -> LDS image
// (row stride `rs`) and global image of width D at slot offset `slot0` (ONES: the slot after the last feature holds 1.0: db).
template <int D, bool ONES>
__device__ __forceinline__ void l4_publish(v4 val, float *lds_img, int rs, __amdgpu_buffer_rsrc_t irs, int slot0, int rowbytes, int grp, int lane) {
    const int b = lane >> 2, j = lane & 3, f0 = 64 * grp + 4 * b;
    if (ONES) {
#pragma unroll
        for (int r = 0; r < 4; ++r) val[r] = f0 + r == D ? 1.0f : val[r];
    }
    *(v4 *)(lds_img + j * rs + f0) = val;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float vr = val[r];      // (bit_cast of the vector element itself stored element 0 four times: hipcc, ROCm 7.2)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vr), irs, l4_slot_bytes<D>(f0 + r) + rowbytes, slot0 * 64, 0);
    }
}

template <int F, int Z, bool RT = false>
__global__ void __launch_bounds__(256) lat4_chain_kernel(const v4 *__restrict__ l4, const float *__restrict__ params, const void *__restrict__ xin,
                                                         int in_f64, int64_t n, const double *__restrict__ feats, float *__restrict__ imgs,
                                                         double *__restrict__ loss_part, int fr, int zr) {
    using N = Net<F, Z>;
    using T = L4<N>;
    using LT = Lat<N>;
    constexpr int kImgFloats = LT::z_off(N::L) * kImgStride;
    // every feature slot a GEMM reads (4 x steps <= 64 x groups) is written by the producing epilogue: no zero fill needed
    __shared__ __attribute__((aligned(16))) float lds[T::lds_floats];
    __shared__ double loss_lds[4];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    LAT_T(46);
    const int b = lane >> 2, j = lane & 3;
    // the four workgroups of a 16-row block share its image lines (16 of the 64 bytes of every slot each): give them workgroup
    // ids that are equal mod 8, i.e. (as workgroups are dealt to the XCDs round-robin) the same XCD and L2 -- placement only,
    // nothing depends on it (with ids 4 blk + q the lines were merged in memory and lat2_dw_kernel took 9.2 instead of 7.8 us)
    const int wg = blockIdx.x, nwg = gridDim.x;
    int blk = wg >> 2, quad = wg & 3;
    if (nwg % 32 == 0) { blk = (wg >> 5) * 8 + (wg & 7); quad = (wg >> 3) & 3; }
    const int64_t row = (int64_t)blk * 16 + 4 * quad + j;
    const bool valid = row < n;
    const int rowbytes = (4 * quad + j) * 4;
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void *)(imgs + (int64_t)blk * kImgFloats), 0, kImgFloats * 4, 0x00020000);
    WStream ws = make_stream(l4, N::l4_frags() * 16, lane);
    // this wave's group / K part per GEMM
    int voff[15], k0[15], grp[15], part[15];
#pragma unroll
    for (int g = 0; g < 15; ++g) {
        const int G = T::groups(g);
        grp[g] = wave & (G - 1);
        part[g] = wave / G;
        k0[g] = part[g] * T::ksp(g);
        const int f = 64 * grp[g] + lane;
        voff[g] = f < N::l4_gemm_n(g) ? f * 16 : 0x7F000000;
    }
    // Request order = order of use (loads of a wave retire in order): the 4 input rows (HBM), the biases, then the fragment ring.
    // (Before: ring, then 24 individually predicated bias loads with 64-bit address arithmetic, then the rows -- 2,460 + 1,000 cycles
    // in front of the first GEMM, tools/lat_trace.py.)
    // the 4 input rows: lane (b, j), b < F / 4, of wave 0 reads features 4 b .. 4 b + 3 of row j (rows beyond n: row 0, never used)
    static_assert(F % 4 == 0 && F <= 32, "input rows as 4-feature pieces of one 32-slot row");
    double xd[4] = {0.0, 0.0, 0.0, 0.0}, xmn[4] = {0.0, 0.0, 0.0, 0.0}, xrg[4] = {1.0, 1.0, 1.0, 1.0};
    if (wave == 0) {
        const int fb0 = 4 * b < F ? 4 * b : 0;
        const int64_t base = (valid ? row : 0) * F + fb0;
        if (in_f64) {
#pragma unroll
            for (int r = 0; r < 4; ++r) xd[r] = ((const double *)xin)[base + r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) xd[r] = (double)((const float *)xin)[base + r];
        }
        if (feats) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { xmn[r] = feats[fb0 + r]; xrg[r] = feats[F + fb0 + r]; }
        }
    }
    // biases -> LDS, 256 slots per layer: thread t brings bias t of every layer through a buffer resource (beyond the layer's width: an
    // out-of-range offset = 0.0, no branch)
    {
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void *)params, 0, N::w_off(N::L) * 4, 0x00020000);
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const int vo = (int)threadIdx.x < N::dim(l + 1) ? (int)threadIdx.x * 4 : 0x7F000000;
            lds[T::bo + 256 * l + threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(prs, vo, N::b_off(l) * 4, 0));
        }
    }
    v4 ring[T::D];
    l4_prologue<N>(ring, ws, voff, k0, std::make_integer_sequence<int, T::D>{});
    const int tf = threadIdx.x >> 2, tj = threadIdx.x & 3;         // finalising thread: feature tf (+ 64), row tj
    double lacc = 0.0;
    LAT_T(47);
    if (wave == 0) {
        v4 x0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double d = xd[r];
            if (feats) d = (d - xmn[r]) / xrg[r];
            x0[r] = 4 * b < F ? (float)d : 0.f;
        }
        if (4 * b < 32) l4_publish<F, true>(x0, lds + T::xo(0), T::xs(0), irs, LT::x_off(0), rowbytes, 0, lane);
    }
    __syncthreads();
    // biases: accumulator layout for the GEMMs without a K split (layers 0 and 6), per finalising thread for the others
    const v4 bias0 = *(const v4 *)(lds + T::bo + 64 * wave + 4 * b), bias6 = *(const v4 *)(lds + T::bo + 256 * 6 + 64 * wave + 4 * b);
    float fb[8][2];
#pragma unroll
    for (int l = 0; l < 8; ++l)
#pragma unroll
        for (int i = 0; i < 2; ++i) fb[l][i] = (T::parts(l) > 1 && i < T::groups(l)) ? lds[T::bo + 256 * l + tf + 64 * i] : 0.f;
    LAT_T(48);
    float *pbuf = lds + T::po;
    // One chain GEMM.  FWD: layer l = g, input X_l, output X_{l+1} = act(W x + b).  !FWD: layer l = 15 - g, input dZ_l, output
    // dZ_{l-1} = (W^T dZ_l) . lrelu'(X_l).  `fin(value, feature, i)` finalises one value of a K-split GEMM (thread (tf + 64 i, tj)).
#define L4_GEMM(g, IN_OFF, IN_RS, DIRECT, FINAL)                                                                             \
    {                                                                                                                        \
        const float *brow = lds + (IN_OFF) + j * (IN_RS) + 4 * k0[g];                                                        \
        v4 init = (v4){0.f, 0.f, 0.f, 0.f};                                                                                  \
        if ((g) == 0) init = bias0;                                                                                          \
        if ((g) == 6) init = bias6;                                                                                          \
        v4 o = l4_gemm<N, g>(init, brow, ring, ws, voff, k0, std::make_integer_sequence<int, T::ksp(g)>{});                  \
        if constexpr (T::parts(g) == 1) {                                                                                    \
            DIRECT(o);                                                                                                       \
        } else {                                                                                                             \
            *(v4 *)(pbuf + (part[g] * 4 + j) * T::ps + 64 * grp[g] + 4 * b) = o;                                            \
            __syncthreads();                                                                                                 \
            _Pragma("unroll") for (int i = 0; i < T::groups(g); ++i) {                                                       \
                const int f = tf + 64 * i;                                                                                   \
                float v = 0.f;                                                                                               \
                _Pragma("unroll") for (int p = 0; p < T::parts(g); ++p) v += pbuf[(p * 4 + tj) * T::ps + f];                 \
                FINAL(v, f, i);                                                                                              \
            }                                                                                                                \
        }                                                                                                                    \
        __syncthreads();                                                                                                     \
    }
    const int trow = (4 * quad + tj) * 4;          // the finalising thread's row offset in an image slot
    // store one finalised value: LDS image + global image (4 consecutive threads = the 4 rows of a feature = 16 contiguous bytes)
    auto store1 = [&](auto dtag, bool ones, float v, int f, int ldso, int rs, int slot0) {
        constexpr int D_ = decltype(dtag)::value;
        if (ones && f == D_) v = 1.0f;
        lds[ldso + tj * rs + f] = v;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), irs, l4_slot_bytes<D_>(f) + trow, slot0 * 64, 0);
    };
    // ---------------- forward ----------------
#define L4_FWD(l)                                                                                                            \
    {                                                                                                                        \
        auto direct = [&](v4 o) {                                                                                            \
            if (N::act(l)) { _Pragma("unroll") for (int r = 0; r < 4; ++r) o[r] = o[r] > 0.f ? o[r] : 0.01f * o[r]; }       \
            l4_publish<N::dim((l) + 1), true>(o, lds + T::xo((l) + 1), T::xs((l) + 1), irs, LT::x_off((l) + 1), rowbytes, grp[l], lane); \
        };                                                                                                                   \
        auto fin = [&](float v, int f, int i) {                                                                              \
            v += fb[l][i];                                                                                                   \
            if (N::act(l)) v = v > 0.f ? v : 0.01f * v;                                                                      \
            store1(std::integral_constant<int, N::dim((l) + 1)>{}, true, v, f, T::xo((l) + 1), T::xs((l) + 1), LT::x_off((l) + 1)); \
        };                                                                                                                   \
        L4_GEMM(l, T::xo(l), T::xs(l), direct, fin)                                                                          \
        LAT_T(49 + (l));                                                                                                     \
    }
    L4_FWD(0) L4_FWD(1) L4_FWD(2) L4_FWD(3) L4_FWD(4) L4_FWD(5) L4_FWD(6)
#undef L4_FWD
    {   // layer 7 (one group, K split over the four waves) + loss + dL/drecon
        auto direct = [&](v4) {};
        auto fin = [&](float v, int f, int i) {
            v += fb[7][i];
            const float d = v - lds[T::xo(0) + tj * T::xs(0) + (f < 32 ? f : 0)];
            const bool live = ((int64_t)blk * 16 + 4 * quad + tj < n) && f < F;
            if (live) lacc += (double)d * (double)d;
            v = live ? d * (2.0f / (float)(RT ? fr : F)) : 0.f;
            store1(std::integral_constant<int, F>{}, false, v, f, T::zo(7), T::zs(7), LT::z_off(7));
        };
        L4_GEMM(7, T::xo(7), T::xs(7), direct, fin)
        LAT_T(56);
    }
    // ---------------- backward chain: GEMM 15 - l: dZ_{l-1} = (W_l^T dZ_l) . lrelu'(X_l) ----------------
#define L4_BWD(l)                                                                                                            \
    {                                                                                                                        \
        auto direct = [&](v4 o) {                                                                                            \
            if (N::act((l) - 1)) {                                                                                           \
                const v4 y = *(const v4 *)(lds + T::xo(l) + j * T::xs(l) + 64 * grp[15 - (l)] + 4 * b);                      \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) o[r] = y[r] > 0.f ? o[r] : 0.01f * o[r];                       \
            }                                                                                                                \
            l4_publish<N::dim(l), false>(o, lds + T::zo((l) - 1), T::zs((l) - 1), irs, LT::z_off((l) - 1), rowbytes, grp[15 - (l)], lane); \
        };                                                                                                                   \
        auto fin = [&](float v, int f, int) {                                                                                \
            if (N::act((l) - 1)) v = lds[T::xo(l) + tj * T::xs(l) + f] > 0.f ? v : 0.01f * v;                                \
            store1(std::integral_constant<int, N::dim(l)>{}, false, v, f, T::zo((l) - 1), T::zs((l) - 1), LT::z_off((l) - 1)); \
        };                                                                                                                   \
        L4_GEMM(15 - (l), T::zo(l), T::zs(l), direct, fin)                                                                   \
        LAT_T(64 - (l));                                                                                                     \
    }
    L4_BWD(7) L4_BWD(6) L4_BWD(5) L4_BWD(4) L4_BWD(3) L4_BWD(2) L4_BWD(1)
#undef L4_BWD
#undef L4_GEMM
    // loss partial of these 4 rows: lanes of a wave, then waves 0..3 (fixed order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lacc += __shfl_down(lacc, off);
    if (lane == 0) loss_lds[wave] = lacc;
    __syncthreads();
    if (threadIdx.x == 0) loss_part[4 * blk + quad] = ((loss_lds[0] + loss_lds[1]) + loss_lds[2]) + loss_lds[3];
}

struct AdamArgs {   // scalars of one Adam step (launch_adam's), and where the state lives
    float *params, *pcopy, *m, *v, *packed;
    const int *sc_off, *sc_idx;
    double *loss_accum;
    double b1, b2, eps, step_size, bc2_sqrt;
};

// grid = 8 x ceil((dW tiles + 1) / 8).  A block contracts dZ^T (tile nt of layer l) with X (tile kt) over all 16-row
// blocks.  Workgroups are dealt to the 8 XCDs round-robin and every XCD has its own L2, so XCD c takes the CONTIGUOUS
// tile range [c * per, (c + 1) * per): neighbouring tiles share their X / dZ image slices, each slice then crosses the
// fabric into (about) one L2 instead of eight.  The map entry and the optimiser state of the thread's parameter are
// requested BEFORE the image loop, so the dependent global round trips overlap instead of queueing up.
enum { DW_WRITE = 0, DW_ADAM = 1, DW_ACCUM = 2 };   // grads = g | Adam on g (+ grads = g) | grads += g
template <class N, int MODE>
__global__ void __launch_bounds__(256) lat2_dw_kernel(const float *__restrict__ imgs, int nblk, const double *__restrict__ loss_part, int nloss,
                                                      const int *__restrict__ inv_map, float *__restrict__ grads, AdamArgs ad,
                                                      float *__restrict__ part, int nsplit, int phase, int np, double inv_c) {
    // (np = index of the loss slot = the model's parameter count, inv_c = 1 / columns: arguments, since an instantiation for a CLASS of
    // narrow tables serves models of different real widths)
    // phase 0: the whole job in one launch.  From 512 blocks (8192 rows) on a tile's blocks are cut into `nsplit` ranges
    // (blockIdx.y): phase 1 leaves one partial tile per range in `part`, phase 2 (a second launch, one workgroup per tile again)
    // adds them in range order and finishes (Adam / store / accumulate).  Measured us per step, split / one launch: 8192 rows
    // 64.8 / 69.6, 12288 rows 90.5 / 99.8; at 4096 rows 43.8 / 44.2 and below that the second launch costs more than it saves:
    // there the kernel is bound by the 153 MB the 298 tiles read from the images (each slice is shared by 7-13 tiles), not by latency.
    using LT = Lat<N>;
    constexpr int kImgFloats = LT::z_off(N::L) * kImgStride, T = N::slab_off(N::L);
    constexpr int kPerXcd = (T + 1 + 7) / 8;
    __shared__ __attribute__((aligned(16))) v4 red[4 * 64];
    const int tile = (blockIdx.x & 7) * kPerXcd + (blockIdx.x >> 3);
    if (tile > T) return;
    if (tile == T && phase == 1) return;
    if (tile == T) {   // loss: fixed-order sum of the per-block partials, / C
        // 256 strided sums, then a fixed tree (ONE thread adding 32 .. 128 partials one dependent L2 round trip after the other
        // was the longest workgroup of this kernel: 6.5 us of its 6.5-9 us)
        double *lred = (double *)red;
        double s = 0.0;
        for (int k = threadIdx.x; k < nloss; k += 256) s += loss_part[k];
        lred[threadIdx.x] = s;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) lred[threadIdx.x] += lred[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            const float gl = (float)(lred[0] * inv_c);
            if (grads) grads[np] = MODE == DW_ACCUM ? grads[np] + gl : gl;
            if (MODE == DW_ADAM && ad.loss_accum) *ad.loss_accum += (double)gl;
        }
        return;
    }
#ifdef BAMD_LAT_TRACE
#define DW_T(i) do { if ((blockIdx.x == 0 || blockIdx.x == 150) && threadIdx.x == 0) g_lat_trace[32 + (blockIdx.x ? 8 : 0) + i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DW_T(i) do {} while (0)
#endif
    DW_T(0);
    const int p = inv_map[tile * 256 + threadIdx.x];
    int l = 0;
#pragma unroll
    for (int j = 1; j < N::L; ++j) if (tile >= N::slab_off(j)) l = j;
    int nt_count = tiles(N::dim(1)), soff = 0, xo = LT::x_off(0), zo = LT::z_off(0);
#pragma unroll
    for (int j = 1; j < N::L; ++j)
        if (l == j) { nt_count = tiles(N::dim(j + 1)); soff = N::slab_off(j); xo = LT::x_off(j); zo = LT::z_off(j); }
    const int idx = tile - soff, kt = idx / nt_count, nt = idx - kt * nt_count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, i = lane & 15;
    const float *pz = imgs + ((zo + 16 * nt + i) * kImgStride + 4 * g);
    const float *px = imgs + ((xo + 16 * kt + i) * kImgStride + 4 * g);
    float pm = 0.f, pv = 0.f, pp = 0.f;
    int s0 = 0, s1 = 0;
    if (MODE == DW_ADAM && p >= 0) { pm = ad.m[p]; pv = ad.v[p]; pp = ad.params[p]; s0 = ad.sc_off[p]; s1 = ad.sc_off[p + 1]; }
    DW_T(1);
    v4 acc = (v4){0.f, 0.f, 0.f, 0.f};
    const int per = phase == 1 ? (nblk + nsplit - 1) / nsplit : nblk;
    const int blo = phase == 1 ? (int)blockIdx.y * per : 0;
    const int bhi = phase == 2 ? 0 : (blo + per < nblk ? blo + per : nblk);
    for (int b0 = blo + wave; b0 < bhi; b0 += 32) {   // 8 blocks per wave in flight
        v4 a[8], x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int b = b0 + 4 * u;
            const bool ok = b < bhi;
            a[u] = ok ? *(const v4 *)(pz + (int64_t)b * kImgFloats) : (v4){0.f, 0.f, 0.f, 0.f};
            x[u] = ok ? *(const v4 *)(px + (int64_t)b * kImgFloats) : (v4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = mfma(a[u][r], x[u][r], acc);
    }
    DW_T(2);
    red[wave * 64 + lane] = acc;
    int sidx[4] = {-1, -1, -1, -1};                                  // packed copies of this parameter (forward, transposed, region E)
    if (MODE == DW_ADAM && p >= 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (s0 + k < s1) sidx[k] = ad.sc_idx[s0 + k];
    }
    __syncthreads();
    DW_T(3);
    const float *rf = (const float *)red;
    const int e = threadIdx.x;
    float gsum = ((rf[e] + rf[256 + e]) + rf[512 + e]) + rf[768 + e];
    if (phase == 1) { part[((int64_t)tile * nsplit + blockIdx.y) * 256 + e] = gsum; return; }
    if (phase == 2) {
        gsum = 0.f;
        for (int k = 0; k < nsplit; ++k) gsum += part[((int64_t)tile * nsplit + k) * 256 + e];
    }
    if (p < 0) return;
    if (grads) grads[p] = MODE == DW_ACCUM ? grads[p] + gsum : gsum;
    if (MODE == DW_ADAM) {   // elementwise.hip adam_k, on the 256 parameters this tile owns
        const double gi = (double)gsum;
        double mi = (double)pm, vi = (double)pv;
        mi = mi + (gi - mi) * (1.0 - ad.b1);
        vi = vi * ad.b2 + (1.0 - ad.b2) * gi * gi;
        const double denom = sqrt(vi) / ad.bc2_sqrt + ad.eps;
        const float pn = (float)((double)pp - ad.step_size * (mi / denom));
        ad.m[p] = (float)mi;
        ad.v[p] = (float)vi;
        ad.params[p] = pn;
        if (ad.pcopy) ad.pcopy[p] = pn;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (sidx[k] >= 0) ad.packed[sidx[k]] = pn;
        for (int k = s0 + 4; k < s1; ++k) ad.packed[ad.sc_idx[k]] = pn;
    }
    DW_T(4);
}

// Reduction of the per-workgroup partial gradients ([tile][workgroup][64 lanes] float4, see dw_flush).
template <typename T>
__global__ void __launch_bounds__(256) reduce_slabs_k(const v4 *__restrict__ slabs, int nslab, int ntiles, const int *__restrict__ inv_map,
                                                      int np, double inv_c, T *__restrict__ grads) {
    // one workgroup per tile: lane l of wave w sums float4 l of the tile over the w-th quarter of the workgroups' slabs, in
    // workgroup order; the four partial sums are added in wave order (fixed => bitwise reproducible) and scattered to their
    // canonical (state-dict) positions through the inverse map (-1 = padding).  (One wave per tile, 299 waves on the whole
    // chip, took 24 us for 76 MB.)  Block ntiles: grads[np] = sum of loss partials / C.
    __shared__ v4 part[4][64];
    const int tile = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (tile == ntiles) {
        const double l = block_sum_fixed((const double *)(slabs + (int64_t)ntiles * nslab * 64), nslab, (double *)part);
        if (threadIdx.x == 0) grads[np] = (T)(l * inv_c);
        return;
    }
    const int q = (nslab + 3) / 4, k0 = wave * q, k1 = k0 + q < nslab ? k0 + q : nslab;
    const v4 *src = slabs + (int64_t)tile * nslab * 64 + lane;
    v4 s = (v4){0.f, 0.f, 0.f, 0.f};
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        v4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(k + u) * 64];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; k < k1; ++k) s += src[k * 64];
    part[wave][lane] = s;
    __syncthreads();
    if (wave != 0) return;
    s = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int p = inv_map[(tile * 64 + lane) * 4 + c];
        if (p >= 0) grads[p] = (T)s[c];
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

__global__ void __launch_bounds__(256) sum_loss_k(const double *__restrict__ part, int n, double scale, double *__restrict__ out) {
    __shared__ double sh[256];
    const double s = block_sum_fixed(part, n, sh);
    if (threadIdx.x == 0) *out = s * scale;
}

// the same for thousands of partials: 256 strided fixed-order sums, then a fixed tree
__global__ void __launch_bounds__(256) sum_loss_wide_k(const double *__restrict__ part, int n, double scale, double *__restrict__ out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0] * scale;
}

// ---- host side ------------------------------------------------------------------------------------------
struct FusedOps;
struct FusedState {
    const FusedOps *ops = nullptr;
    DevBuf pack_src;   // int per packed float: canonical parameter index or -1
    DevBuf slab_map;   // int per slab float: canonical parameter index or -1 (padding)
    DevBuf dz;         // dL/dz hand-off between the two training kernels: 16 floats per row
    DevBuf sc_off, sc_idx;   // CSR parameter -> packed float positions (fused Adam + pack)
    int packed_floats = 0;
    int nwg_max = 256;
    // <= this many rows: small-batch kernels (BALER_AMD_LATENCY_ROWS overrides).  Measured us/step small-batch vs
    // throughput pair: 1024 rows 27 / 74, 4096 44 / 88, 8192 76 / 101, 16384 133 / 131
    int64_t latency_max_rows = 12288;
    // <= this many rows the chain runs on 4-row workgroups (lat4_chain_kernel: 128 instead of 32 CUs carry a 512-row batch);
    // BALER_AMD_LAT4_ROWS overrides, 0 = off.  Measured us per bamd_train_step, 4-row / 16-row chain: 64 rows 16.1 / 20.9,
    // 256 16.6 / 21.9, 512 18.0 / 23.4, 1024 21.7 / 27.0, 2048 33.1 / 34.1, 4096 53.6 / 44.1 (every workgroup streams all
    // weights: 4x the L2 traffic of the 16-row chain)
    int64_t lat4_max_rows = 2048;
    DevBuf wb_src[6], wb[6];           // wide models in the bf16 mode: index maps and bf16 fragments of W0 / W7 / W7^T (training) / W0 in the DMA encode's
                                       // k order / the narrow encoder layers 1..3 / the narrow decoder layers 4..6 (chain_bf16_pair)
    int wb_count[6] = {0, 0, 0, 0, 0, 0};
    bool wb_stale = false;             // the bf16 fragments lag the parameters (re-rounded before the next encode / decode)
    bool packed_stale = false;         // handles with a second state (64..127 columns): an optimiser step refreshed the small-batch state's fragments only
    bool dz16 = false;                 // this pass stores dL/drecon as bfloat16 (set per pass by the layer-wise driver: fused_wide_set_dz16)
    DevBuf imgs;                       // X^T / dZ^T images of the small-batch path: 104 KiB per 16-row block
    DevBuf dwpart;                     // partial weight-gradient tiles of the small-batch path when a tile's blocks are split over workgroups
    DevBuf wpart;                      // wide models, small training batches: partial sums of the split wide products (wide_small_in_kernel)
    bool tail_split = true;            // short remainder of the persistent loop on the small-batch kernels (BALER_AMD_TAIL_SPLIT=0: off)
};

template <int F, int Z, bool TRAIN>
static int build_maps(bamd_handle *h, FusedState *st) {
    // Geometry (tiles, fragment order, slot -> feature) from the instantiated Net<F, Z>; which slots hold a parameter, and the
    // parameter's canonical index, from the HANDLE's real dimensions: identical for an exact instantiation, and for an instantiation
    // that serves a class of narrow tables (Impl<F, Z, true>) the class's slots beyond the real width map to nothing (-1 -> a zero
    // weight / bias in the packed copy, no entry in the gradient map).
    using N = Net<F, Z>;
    const int nparams_r = (int)h->nparams;
    auto dim_r = [&](int i) { return h->dims[i]; };
    auto w_off_r = [&](int l) { return (int)h->w_off[l]; };
    auto b_off_r = [&](int l) { return (int)h->b_off[l]; };
    std::vector<int> src((size_t)N::packed_f4() * 4, -1);
    std::vector<int> smap((size_t)nparams_r, -1);
    for (int l = 0; l < N::L; ++l) {
        const int K = N::dim(l), NN = N::dim(l + 1), KT = tiles(K), NT = tiles(NN);
        const int Kr = dim_r(l), NNr = dim_r(l + 1);
        // forward frags: [q][t][lane].comp[r] = W[n(t, i = lane & 15)][k(q, g = lane >> 4, r)]
        for (int q = 0; q < KT; ++q)
            for (int t = 0; t < NT; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        int i = lane & 15, g = lane >> 4;
                        int nf = slot_feature(NN, t, i / 4, i % 4), kf = slot_feature(K, q, g, r);
                        if (nf >= 0 && kf >= 0 && nf < NNr && kf < Kr)
                            src[((size_t)N::wf_off(l) + (q * NT + t) * 64 + lane) * 4 + r] = w_off_r(l) + nf * Kr + kf;
                    }
        // backward frags (layers 1..7): [tq][tk][lane].comp[r] = W[n(tq, g, r)][k(tk, i)]
        for (int tq = 0; tq < NT && l >= 1; ++tq)
            for (int tk = 0; tk < KT; ++tk)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        int i = lane & 15, g = lane >> 4;
                        int nf = slot_feature(NN, tq, g, r), kf = slot_feature(K, tk, i / 4, i % 4);
                        if (nf >= 0 && kf >= 0 && nf < NNr && kf < Kr)
                            src[((size_t)N::wb_off(l) + (tq * KT + tk) * 64 + lane) * 4 + r] = w_off_r(l) + nf * Kr + kf;
                    }
        // bias frags: [t][g].comp[r] = b[n(t, g, r)]
        for (int t = 0; t < NT; ++t)
            for (int g = 0; g < 4; ++g)
                for (int r = 0; r < 4; ++r) {
                    int nf = slot_feature(NN, t, g, r);
                    if (nf >= 0 && nf < NNr) src[((size_t)N::bf_off(l) + t * 4 + g) * 4 + r] = b_off_r(l) + nf;
                }
        if (!TRAIN) continue;
        // slab map: tile idx = kt*NT + nt; lane (j = lane & 15 -> k slot row 16kt + j), reg r -> n slot row 4g + r
        const int KTp = tiles(K + 1);
        const int T1 = tiles(K) - 1, V = K - 16 * T1, ones_row = 16 * T1 + 4 * (V % 4) + (V / 4);
        for (int kt = 0; kt < KTp; ++kt)
            for (int nt = 0; nt < NT; ++nt)
                for (int lane = 0; lane < 64; ++lane)
                    for (int r = 0; r < 4; ++r) {
                        int j = lane & 15, g = lane >> 4;
                        int nf = slot_feature(NN, nt, g, r);
                        if (nf < 0 || nf >= NNr) continue;
                        int off = ((N::slab_off(l) + kt * NT + nt) * 64 + lane) * 4 + r;
                        int krow = 16 * kt + j;
                        if (krow == ones_row) smap[b_off_r(l) + nf] = off;
                        else if (kt < tiles(K)) {
                            int kf = slot_feature(K, kt, j / 4, j % 4);
                            if (kf >= 0 && kf < Kr) smap[w_off_r(l) + nf * Kr + kf] = off;
                        }
                    }
    }
    // region E: copies of Wf(0..2) and Wb(3,2,1) in the encoder-gradient kernel's consumption order
    for (int l = 0; l < kSplit - 1; ++l)
        for (int i = 0; i < N::wcount(l) * 4; ++i) src[(size_t)N::ef_off(l) * 4 + i] = src[(size_t)N::wf_off(l) * 4 + i];
    for (int l = kSplit - 1; l >= 1; --l)
        for (int i = 0; i < N::wcount(l) * 4; ++i) src[(size_t)N::eb_off(l) * 4 + i] = src[(size_t)N::wb_off(l) * 4 + i];
    bool exact = true;
    for (int i = 0; i <= N::L; ++i) exact = exact && h->dims[i] == N::dim(i);
    if (TRAIN && N::l4_frags() > 0 && exact) {      // (the 4-row chain reads canonical biases at compile-time offsets: exact instantiations only)
        // region L4: GEMM g, step k4, output feature o: component r = W[o][4 k4 + r] (forward: layer g) or its transpose
        // W_l[4 k4 + r][o] (input-gradient product of layer l = 15 - g)
        for (int g = 0; g < 15; ++g) {
            const int l = g < 8 ? g : 15 - g, K = dim_r(l), NN = dim_r(l + 1);      // (region L4 serves exact instantiations only)
            const int NO = N::l4_gemm_n(g);
            for (int k4 = 0; k4 < N::l4_ks(g); ++k4)
                for (int o = 0; o < NO; ++o)
                    for (int r = 0; r < 4; ++r) {
                        const int c = 4 * k4 + r;
                        const size_t at = ((size_t)N::l4_off() + N::l4_frag_off(g) + (size_t)k4 * NO + o) * 4 + r;
                        if (g < 8) { if (c < K) src[at] = w_off_r(l) + o * K + c; }
                        else if (c < NN) src[at] = w_off_r(l) + c * K + o;
                    }
        }
    }
    if (TRAIN) {
        for (int v : smap)
            if (v < 0) { set_error("fused: incomplete slab map"); return BAMD_ERR_INVALID; }
        std::vector<int> inv((size_t)N::slab_off(N::L) * 64 * 4, -1);   // slab float -> canonical parameter
        for (int p = 0; p < nparams_r; ++p) inv[smap[p]] = p;
        smap.swap(inv);
    }
    {   // inverse of the pack map: for every parameter the list of packed positions that hold a copy of it
        std::vector<int> off((size_t)nparams_r + 1, 0), idx;
        for (int v : src) if (v >= 0) off[v + 1]++;
        for (int p = 0; p < nparams_r; ++p) off[p + 1] += off[p];
        idx.resize(off[nparams_r]);
        std::vector<int> cur(off.begin(), off.end() - 1);
        for (size_t i = 0; i < src.size(); ++i) if (src[i] >= 0) idx[cur[src[i]]++] = (int)i;
        int rc2 = st->sc_off.ensure(off.size() * sizeof(int));
        if (rc2) return rc2;
        rc2 = st->sc_idx.ensure(idx.size() * sizeof(int));
        if (rc2) return rc2;
        BAMD_HIP(hipMemcpy(st->sc_off.p, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
        BAMD_HIP(hipMemcpy(st->sc_idx.p, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    }
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

constexpr int kBiasF4 = Net<24, 15>::bf_off(8) - Net<24, 15>::bf_off(0);   // 51 tiles x 4 lane groups for every Z <= 16

// entry points of one instantiated shape
struct FusedOps {
    int (*setup)(bamd_handle *, FusedState *);
    int (*encode)(bamd_handle *, const void *, int, int64_t, const double *, void *, int, hipStream_t);
    int (*decode)(bamd_handle *, const void *, int, int64_t, const double *, const uint8_t *, void *, int, hipStream_t);
    int (*forward_loss)(bamd_handle *, const void *, int, int64_t, const double *, void *, int, double *, hipStream_t);
    int (*fwd_bwd)(bamd_handle *, const void *, int, int64_t, const double *, void *, hipStream_t);
    // fwd + bwd + Adam + pack in two launches (small batches only; returns BAMD_ERR_UNSUPPORTED when n is too large)
    int (*train_step)(bamd_handle *, const void *, int, int64_t, const double *, void *, const AdamArgs &, hipStream_t);
    // wide models: the row-local parts of a layer-wise training pass (see wide_train_fwd_kernel / wide_train_bwd_kernel)
    int (*wide_fwd)(bamd_handle *, const float *, int64_t, float *const *, float *, double *, int *, hipStream_t);
    int (*wide_bwd)(bamd_handle *, int64_t, float *const *, float *const *, const float *, hipStream_t);
    int (*pack_extra)(bamd_handle *, FusedState *, hipStream_t);    // further packed copies of the parameters (bf16 fragments)
    bool throughput_training = true;      // false: large-batch training of this shape runs on generic.hip (bamd_path_of: FUSED_INFER)
    bool (*wide_small)(const bamd_handle *, int64_t) = nullptr;     // wide models: this batch size takes the split launches
};

static FusedState *state_of(bamd_handle *h) { return (FusedState *)h->fused_state; }

static constexpr bool infer_pair() { return true; }   // two 16-row tiles per wave in encode / decode (+5.5 % / +7 % over one tile)
static int infer_grid(int64_t n) {
    int64_t wg = ((n + 15) / 16 + 3) / 4;
    return (int)(wg < 1 ? 1 : (wg > 1024 ? 1024 : wg));
}

// RT = true: the instantiation serves a CLASS of narrow tables -- every AE(f, z) with f <= F, z <= Z and the
// reference's hidden widths (models.py:122-139 builds AE(n_features, z_dim) for ANY column count, baler.py:117-123 derives any latent):
// F = 16 T - 1 is the class width, the real widths are kernel arguments (load_rows / store_rows, RT), the pack / gradient maps leave
// the class's extra slots empty (build_maps).  Same kernels, same tile counts as the 24-column instantiation for 16..31 columns.
template <int F, int Z, bool RT = false> struct Impl {
    using N = Net<F, Z>;
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
    static constexpr int train_lds = (img_a_rows<N>() + kImgB) * kQS * (int)sizeof(float) + (train_bias_in_lds<N>() ? (N::bf_off(8) - N::bf_off(0)) * 16 : 0);      // images (+ bias fragments)
    static int fr(const bamd_handle *h) { return h->dims[0]; }
    static int zr(const bamd_handle *h) { return h->dims[4]; }
    static int setup(bamd_handle *h, FusedState *st) {
        static_assert(RT || N::bf_off(8) - N::bf_off(0) == kBiasF4, "bias fragment count");
        static_assert(train_lds <= 160 * 1024, "LDS of the training pair");
        int rc = build_maps<F, Z, true>(h, st);
        if (rc) return rc;
        BAMD_HIP(hipFuncSetAttribute((const void *)train_dec_kernel<F, Z, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, train_lds));
        BAMD_HIP(hipFuncSetAttribute((const void *)train_enc_kernel<F, Z, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, train_lds));
        return BAMD_OK;
    }
    static int encode(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *z, int z_dtype,
                      hipStream_t s) {
        constexpr int extra_lds = 0;
        if (F <= 64 && infer_pair()) {
            hipLaunchKernelGGL((infer2_kernel<F <= 64 ? F : 24, Z, K_ENCODE, RT>), dim3(infer_grid((n + 1) / 2)), dim3(256), 0, s,
                               (const v4 *)h->packed.p, x, x_dtype == BAMD_F64, n, features, z, z_dtype == BAMD_F64,
                               (const uint8_t *)nullptr, fr(h), zr(h));
            BAMD_HIP(hipGetLastError());
            return BAMD_OK;
        }
        hipLaunchKernelGGL((infer_kernel<F, Z, K_ENCODE, RT>), dim3(infer_grid(n)), dim3(256), extra_lds, s, (const v4 *)h->packed.p, x,
                           x_dtype == BAMD_F64, n, features, z, z_dtype == BAMD_F64, (const uint8_t *)nullptr, (double *)nullptr, fr(h), zr(h));
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static int decode(bamd_handle *h, const void *z, int z_dtype, int64_t n, const double *features, const uint8_t *int_mask,
                      void *out, int out_dtype, hipStream_t s) {
        if (F <= 64 && infer_pair()) {
            hipLaunchKernelGGL((infer2_kernel<F <= 64 ? F : 24, Z, K_DECODE, RT>), dim3(infer_grid((n + 1) / 2)), dim3(256), 0, s,
                               (const v4 *)h->packed.p, z, z_dtype == BAMD_F64, n, features, out, out_dtype == BAMD_F64, int_mask, fr(h), zr(h));
            BAMD_HIP(hipGetLastError());
            return BAMD_OK;
        }
        hipLaunchKernelGGL((infer_kernel<F, Z, K_DECODE, RT>), dim3(infer_grid(n)), dim3(256), 0, s, (const v4 *)h->packed.p, z,
                           z_dtype == BAMD_F64, n, features, out, out_dtype == BAMD_F64, int_mask, (double *)nullptr, fr(h), zr(h));
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static int forward_loss(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *recon,
                            int recon_dtype, double *loss_sum, hipStream_t s) {
        int grid = infer_grid(n);
        int rc = h->lossp.ensure(sizeof(double) * 1024);
        if (rc) return rc;
        hipLaunchKernelGGL((infer_kernel<F, Z, K_FORWARD, RT>), dim3(grid), dim3(256), 0, s, (const v4 *)h->packed.p, x,
                           x_dtype == BAMD_F64, n, features, recon, recon_dtype == BAMD_F64, (const uint8_t *)nullptr,
                           (double *)h->lossp.p, fr(h), zr(h));
        hipLaunchKernelGGL(sum_loss_k, dim3(1), dim3(256), 0, s, (const double *)h->lossp.p, grid, 1.0 / fr(h), loss_sum);
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static int fwd_bwd(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads,
                       hipStream_t s) {
        FusedState *st = state_of(h);
        const int np = (int)h->nparams;
        if (n <= st->latency_max_rows) return small_batch(h, x, x_dtype, n, features, grads, nullptr, s);
        int64_t ngroups = (n + kRowsPerWG - 1) / kRowsPerWG;
        int grid = (int)(ngroups < st->nwg_max ? ngroups : st->nwg_max);
        // Persistent-loop quantisation: with 15,625 row groups on 256 workgroups the 62nd iteration runs on 9 of them.
        // A short remainder (<= 32 groups) goes to the small-batch kernels instead, accumulated into the same gradient:
        // ~23 us against one ~56-us iteration of the pair.
        int64_t tail_rows = 0;
        if (ngroups > grid && ngroups % grid != 0 && ngroups % grid <= 32 && st->tail_split) {
            const int64_t main_rows = (ngroups - ngroups % grid) * kRowsPerWG;
            tail_rows = n - main_rows;
            n = main_rows;
            ngroups -= ngroups % grid;
        }
        int rc = h->slabs.ensure((size_t)N::slab_f4() * 16 * (size_t)grid);
        if (rc) return rc;
        // whole row groups + one round of prefetch overrun (the second kernel loads the next group's record unconditionally)
        rc = st->dz.ensure((size_t)(ngroups + grid) * kRowsPerWG * (kSplit == 2 ? 7 * 64 : 64));
        if (rc) return rc;
        hipLaunchKernelGGL((train_dec_kernel<F, Z, RT>), dim3(grid), dim3(256), train_lds, s, (const v4 *)h->packed.p, x,
                           x_dtype == BAMD_F64, n, features, (v4 *)h->slabs.p, (v4 *)st->dz.p, fr(h), zr(h));
        hipLaunchKernelGGL((train_enc_kernel<F, Z, RT>), dim3(grid), dim3(256), train_lds, s, (const v4 *)h->packed.p, x,
                           x_dtype == BAMD_F64, n, features, (v4 *)h->slabs.p, (const v4 *)st->dz.p, fr(h), zr(h));
        hipLaunchKernelGGL(reduce_slabs_k<float>, dim3(N::slab_off(N::L) + 1), dim3(256), 0, s, (const v4 *)h->slabs.p, grid,
                           N::slab_off(N::L), (const int *)st->slab_map.p, np, 1.0 / fr(h), (float *)grads);
        BAMD_HIP(hipGetLastError());
        if (tail_rows > 0) {
            const char *xt = (const char *)x + (size_t)n * fr(h) * (x_dtype == BAMD_F64 ? 8 : 4);
            return small_batch(h, xt, x_dtype, tail_rows, features, grads, nullptr, s, true);
        }
        return BAMD_OK;
    }
    // small batch: chain kernel (one workgroup per 16-row block) + one workgroup per weight-gradient tile; with
    // `ad` the second kernel also applies Adam and refreshes the packed weights (grads may then be null)
    static int small_batch(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads,
                           const AdamArgs *ad, hipStream_t s, bool accumulate = false) {
        FusedState *st = state_of(h);
        const int nblk = (int)((n + 15) / 16);
        constexpr size_t img_bytes = (size_t)Lat<N>::z_off(N::L) * kImgStride * sizeof(float);
        int rc = st->imgs.ensure(img_bytes * (size_t)nblk);
        if (rc) return rc;
        rc = h->lossp.ensure(sizeof(double) * (size_t)(4 * nblk > 1024 ? 4 * nblk : 1024));
        if (rc) return rc;
        int nloss = nblk;
        bool four_row = false;
        if constexpr (!RT) {      // (the 4-row chain reads canonical biases at compile-time offsets: exact instantiations only)
            if (n <= st->lat4_max_rows) {
                // one workgroup per FOUR rows: a 512-row batch on 128 CUs (v_mfma_f32_4x4x1_16B_f32: 64 output features x 4 rows per
                // instruction); same images, so the weight-gradient kernel below does not change
                four_row = true;
                nloss = 4 * nblk;
                hipLaunchKernelGGL((lat4_chain_kernel<F, Z, RT>), dim3(4 * nblk), dim3(256), 0, s, (const v4 *)h->packed.p + N::l4_off(),
                                   (const float *)h->params.p, x, x_dtype == BAMD_F64, n, features, (float *)st->imgs.p, (double *)h->lossp.p,
                                   fr(h), zr(h));
            }
        }
        if (!four_row)
            // 4 waves per workgroup: a CU has four MFMA units, 8 waves (2 per SIMD) measured no faster (23.4 vs 23.3 us per step)
            hipLaunchKernelGGL((lat2_chain_kernel<F, Z, 4, RT>), dim3(nblk), dim3(256), 0, s, (const v4 *)h->packed.p, x,
                               x_dtype == BAMD_F64, n, features, (float *)st->imgs.p, (double *)h->lossp.p, fr(h), zr(h));
        const dim3 grid(8 * ((N::slab_off(N::L) + 1 + 7) / 8));
        // tiles x block ranges from 8192 rows on (see lat2_dw_kernel)
        const int nsplit = nblk >= 512 ? 4 : 1;
        float *part = nullptr;
        if (nsplit > 1) {
            rc = st->dwpart.ensure((size_t)(N::slab_off(N::L) + 1) * nsplit * 256 * sizeof(float));
            if (rc) return rc;
            part = (float *)st->dwpart.p;
        }
        auto launch = [&](auto mode, const AdamArgs &aa) {
            constexpr int M = decltype(mode)::value;
            if (nsplit > 1) {
                hipLaunchKernelGGL((lat2_dw_kernel<N, M>), dim3(grid.x, nsplit), dim3(256), 0, s, (const float *)st->imgs.p, nblk,
                                   (const double *)h->lossp.p, nloss, (const int *)st->slab_map.p, (float *)grads, aa, part, nsplit, 1, (int)h->nparams, 1.0 / fr(h));
                hipLaunchKernelGGL((lat2_dw_kernel<N, M>), grid, dim3(256), 0, s, (const float *)st->imgs.p, nblk,
                                   (const double *)h->lossp.p, nloss, (const int *)st->slab_map.p, (float *)grads, aa, part, nsplit, 2, (int)h->nparams, 1.0 / fr(h));
            } else
                hipLaunchKernelGGL((lat2_dw_kernel<N, M>), grid, dim3(256), 0, s, (const float *)st->imgs.p, nblk,
                                   (const double *)h->lossp.p, nloss, (const int *)st->slab_map.p, (float *)grads, aa, part, 1, 0, (int)h->nparams, 1.0 / fr(h));
        };
        if (ad) launch(std::integral_constant<int, DW_ADAM>{}, *ad);
        else if (accumulate) launch(std::integral_constant<int, DW_ACCUM>{}, AdamArgs{});
        else launch(std::integral_constant<int, DW_WRITE>{}, AdamArgs{});
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static int train_step(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads,
                          const AdamArgs &ad, hipStream_t s) {
        FusedState *st = state_of(h);
        if (n > st->latency_max_rows) return BAMD_ERR_UNSUPPORTED;
        return small_batch(h, x, x_dtype, n, features, grads, &ad, s);
    }
    static const FusedOps *ops() {
        static const FusedOps o = {setup, encode, decode, forward_loss, fwd_bwd, train_step};
        return &o;
    }
};

// Classes beyond what the throughput training pair's LDS images hold (64..79 columns: 165 KB): encode / decode / forward + loss
// are the register-chained kernels above (they have no images), training steps of up to 12288 rows run on the small-batch kernels
// (their images live in global memory) -- the reference's 512-row steps -- and larger batches on the layer-wise kernels
// (generic_fwd_bwd below; bamd_train_step falls through on BAMD_ERR_UNSUPPORTED).  bamd_path_of() = BAMD_PATH_FUSED_INFER.
template <int F, int Z, bool SMALL = true> struct ImplInferClass {
    using B = Impl<F, Z, true>;
    static_assert(F % 16 == 15 && Z % 16 == 15 && (F <= 127 || !SMALL), "class widths are 16 T - 1; the small-batch chain takes up to 8 input tiles");
    static bool matches(const bamd_handle *h) { return B::matches(h); }
    static int setup(bamd_handle *h, FusedState *st) { return build_maps<F, Z, SMALL>(h, st); }
    static int fwd_bwd(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads, hipStream_t s) {
        if constexpr (SMALL) {
            if (n <= state_of(h)->latency_max_rows) return B::small_batch(h, x, x_dtype, n, features, grads, nullptr, s);
            // Larger batches: the same two kernels chunk after chunk over one image buffer, every chunk after the first ADDING to the
            // gradient and the loss (lat2_dw_kernel<DW_ACCUM>; one stream: a fixed order).  Measured against the layer-wise pass at
            // 1M rows (tools/bench_mid_width_train.py): AE(80, 16) 78 -> 101 M rows/s, AE(64, 16) 85 -> 105, AE(127, 31) 76 -> 95.
            // BALER_AMD_CLASS_CHUNK_ROWS sets the chunk (tests: several chunks at small sizes); 0 = the layer-wise pass.
            const char *ce = getenv("BALER_AMD_CLASS_CHUNK_ROWS");      // (read per call: a training pass is milliseconds; tests toggle it)
            const int64_t chunk = ce ? atoll(ce) : 65536;
            if (chunk >= 16) {
                const size_t row_bytes = (size_t)B::fr(h) * (x_dtype == BAMD_F64 ? 8 : 4);
                const int64_t ck = chunk & ~(int64_t)15;
                for (int64_t r0 = 0; r0 < n; r0 += ck) {
                    const int rc = B::small_batch(h, (const char *)x + (size_t)r0 * row_bytes, x_dtype, n - r0 < ck ? n - r0 : ck, features, grads,
                                                  nullptr, s, r0 > 0);
                    if (rc) return rc;
                }
                return BAMD_OK;
            }
        }
        return generic_fwd_bwd(h, x, x_dtype, n, features, grads, s);
    }
    static int train_step(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads,
                          const AdamArgs &ad, hipStream_t s) {
        if constexpr (SMALL)
            if (n <= state_of(h)->latency_max_rows) return B::small_batch(h, x, x_dtype, n, features, grads, &ad, s);
        return BAMD_ERR_UNSUPPORTED;
    }
    static const FusedOps *ops() {
        static const FusedOps o = {setup, B::encode, B::decode, B::forward_loss, fwd_bwd, train_step, nullptr, nullptr, nullptr, /*throughput_training=*/false};
        return &o;
    }
};

// Wide models (CFD_dense_AE(2500, 25), BASELINE.json configs[3]): encode and decode as ONE launch each (wide_encode_lds_kernel / wide_encode2_kernel, wide_decode_lds_kernel);
// training = the two row-local launches wide_fwd / wide_bwd inside generic.hip's layer-wise pass (the weight gradient of a
// 2500 x 200 layer is 2 MB of accumulators per workgroup: it has to be a split-K product, generic.hip's dw_wide_k); forward_loss
// layer-wise.  Normalise-on-load / un-normalise-on-store go through a float32 staging buffer
// (the per-feature min / range of 2500 features do not fit next to the chain's registers).
// WRT = true: a CLASS instantiation with run-time widths (models.py:192-209 builds CFD_dense_AE(n_features, z_dim) for ANY flattened
// field, baler.py:117-123 derives any latent): F = the class width (a multiple of 16; geometry of the packed weights), Z = the class
// latent; the handle's real column count (48 .. F) and latent (<= Z) are kernel arguments (wide_x_chunk).  The one-tile kernels
// with LDS-shared fragments serve every size of such a handle.
template <int F, int Z, bool WRT = false> struct ImplWide {
    using N = Net<F, Z>;
    static constexpr int64_t kChunkRows = 1 << 18;     // staging chunk: 2.6 GB of float32 rows
    static int Fr(const bamd_handle *h) { return WRT ? h->dims[0] : F; }
    static int Zr(const bamd_handle *h) { return WRT ? h->dims[4] : Z; }
    static bool matches(const bamd_handle *h) {
        if (h->L != 8) return false;
        if (WRT) {
            for (int i = 1; i <= 7; ++i)
                if (i != 4 && h->dims[i] != N::dim(i)) return false;
            return h->dims[0] == h->dims[8] && h->dims[0] >= 48 && h->dims[0] <= F && h->dims[4] <= Z;
        }
        for (int i = 0; i <= 8; ++i)
            if (h->dims[i] != N::dim(i)) return false;
        return true;
    }
    static_assert(!WRT || F % 16 == 0, "a wide class width is a multiple of 16 (every class chunk a full tile)");
    static int setup(bamd_handle *h, FusedState *st) { return build_maps<F, Z, false>(h, st); }
    static int grid_for(int64_t n) {
        const int64_t wg = ((n + 15) / 16 + 3) / 4;
        return (int)(wg < 1 ? 1 : (wg > 2048 ? 2048 : wg));
    }
    static int encode(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *z, int z_dtype,
                      hipStream_t s) {
        const size_t xes = x_dtype == BAMD_F64 ? 8 : 4, zes = z_dtype == BAMD_F64 ? 8 : 4;
        const int fr = Fr(h), zr = Zr(h);
        for (int64_t r0 = 0; r0 < n; r0 += kChunkRows) {
            const int64_t rows = n - r0 < kChunkRows ? n - r0 : kChunkRows;
            const void *src = (const char *)x + (size_t)r0 * fr * xes;
            int src_f64 = x_dtype == BAMD_F64;
            if (features) {
                int rc = h->work.ensure((size_t)rows * fr * sizeof(float));
                if (rc) return rc;
                rc = launch_normalize(src, x_dtype, rows, fr, features, h->work.p, BAMD_F32, s);
                if (rc) return rc;
                src = h->work.p;
                src_f64 = 0;
            }
            // two row tiles per wave once that still leaves two workgroups per CU (measured: 512-column model, 524288 rows
            // 384 -> 419 M rows/s; C4 131072 frames 87.6 -> 91.8 M, but 32768 frames 96 -> 72 M: half the chip's wave slots empty);
            // BALER_AMD_WIDE2=0 / 1 forces one / two tiles
            const char *e2 = getenv("BALER_AMD_WIDE2");
            const bool two = !WRT && (e2 ? e2[0] == '1' : rows >= 32 * 4 * 512);
            void *zo = (void *)((char *)z + (size_t)r0 * zr * zes);
            const int z64 = z_dtype == BAMD_F64;
            if constexpr (!WRT) {
                if (two && src_f64)
                    hipLaunchKernelGGL((wide_encode2_kernel<F, Z, true>), dim3(grid_for((rows + 1) / 2)), dim3(256), 0, s, (const v4 *)h->packed.p, src, rows, zo, z64);
                else if (two)
                    hipLaunchKernelGGL((wide_encode2_kernel<F, Z, false>), dim3(grid_for((rows + 1) / 2)), dim3(256), 0, s, (const v4 *)h->packed.p, src, rows, zo, z64);
            }
            if (!two) {
                // fragments shared through LDS (C4, 32768 frames: 92.5 -> 98.5 M rows/s against per-wave fragments)
                const int64_t ng = (rows + 63) / 64;
                const dim3 gl((unsigned)(ng > 2048 ? 2048 : ng));
                if (src_f64) hipLaunchKernelGGL((wide_encode_lds_kernel<F, Z, true, WRT>), gl, dim3(256), 0, s, (const v4 *)h->packed.p, src, rows, zo, z64, fr, zr);
                else hipLaunchKernelGGL((wide_encode_lds_kernel<F, Z, false, WRT>), gl, dim3(256), 0, s, (const v4 *)h->packed.p, src, rows, zo, z64, fr, zr);
            }
        }
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static int decode(bamd_handle *h, const void *z, int z_dtype, int64_t n, const double *features, const uint8_t *int_mask,
                      void *out, int out_dtype, hipStream_t s) {
        const size_t zes = z_dtype == BAMD_F64 ? 8 : 4, oes = out_dtype == BAMD_F64 ? 8 : 4;
        if (features && out_dtype != BAMD_F64) { set_error("decode with features needs a float64 output"); return BAMD_ERR_INVALID; }
        const int fr = Fr(h), zr = Zr(h);
        for (int64_t r0 = 0; r0 < n; r0 += kChunkRows) {
            const int64_t rows = n - r0 < kChunkRows ? n - r0 : kChunkRows;
            void *dst = (char *)out + (size_t)r0 * fr * oes;
            void *kout = dst;
            int kout_f64 = out_dtype == BAMD_F64;
            if (features) {
                int rc = h->work.ensure((size_t)rows * fr * sizeof(float));
                if (rc) return rc;
                kout = h->work.p;
                kout_f64 = 0;
            }
            const void *zi = (const void *)((const char *)z + (size_t)r0 * zr * zes);
            if (kout_f64)
                hipLaunchKernelGGL((wide_decode_lds_kernel<F, Z, true, WRT>), dim3(grid_for(rows)), dim3(256), 0, s, (const v4 *)h->packed.p, zi,
                                   z_dtype == BAMD_F64, rows, kout, fr, zr);
            else
                hipLaunchKernelGGL((wide_decode_lds_kernel<F, Z, false, WRT>), dim3(grid_for(rows)), dim3(256), 0, s, (const v4 *)h->packed.p, zi,
                                   z_dtype == BAMD_F64, rows, kout, fr, zr);
            if (features) {
                int rc = launch_renormalize(kout, BAMD_F32, rows, fr, features, int_mask, (double *)dst, s);
                if (rc) return rc;
            }
        }
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    // y[l] = activations entering layer l (row-major float32, y[0] unused), dz[l] = dL/d(pre-activation of layer l)
    // Small training batches (up to BALER_AMD_WIDE_SMALL_ROWS rows, default 8192): how many workgroups share the wide dimension of one
    // 64-row group (wide_small_in_kernel / wide_small_out_kernel); 1 = the one-launch kernels.  At least three chunks / tiles per
    // workgroup, at most 16 splits (the MID launches add the partial sums of every split).
    static int small_splits(const bamd_handle *h, int64_t rows, int units, int *per) {
        const int64_t lim = env_ll("BALER_AMD_WIDE_SMALL_ROWS", 8192);
        *per = units;
        if (rows > lim) return 1;
        const int64_t ngroup = (rows + 63) / 64;
        // the split count that fills the chip's 256 workgroup slots in the fewest rounds of the shortest workgroups (each split also costs
        // the MID launches one more partial set to add): measured on C4, 6,000 rows: 4 splits 497 us per step, 8 splits 460; 2,048 rows: 12
        // splits 302, 8 splits 288
        int best = 1;
        double best_cost = 1e30;
        for (int sp = 1; sp <= 16; ++sp) {
            const int p = (units + sp - 1) / sp;
            if (sp > 1 && p < 3) break;
            const double cost = (double)((sp * ngroup + 255) / 256) * (3.0 + 0.9 * p) + 0.3 * sp;
            if (cost < best_cost) { best_cost = cost; best = sp; }
        }
        if (best < 2) return 1;
        const int p = (units + best - 1) / best;
        *per = p;
        return (units + p - 1) / p;
    }
    static int64_t in16_rows() {      // BALER_AMD_WIDE_IN16_ROWS: most rows whose contraction-split products take one 16-row tile per workgroup
        return env_ll("BALER_AMD_WIDE_IN16_ROWS", 512);
    }
    static int64_t out16_rows() {     // BALER_AMD_WIDE_OUT16_ROWS: most rows whose tile-split product (de4) takes one 16-row tile per workgroup
        return env_ll("BALER_AMD_WIDE_OUT16_ROWS", 512);
    }
    static bool small_pass(const bamd_handle *h, int64_t rows) {      // this batch runs on the split launches
        int per = 0;
        return Fr(h) / 16 >= 3 && small_splits(h, rows, Fr(h) / 16, &per) > 1 && small_splits(h, rows, (Fr(h) + 15) / 16, &per) > 1;
    }
    static int wide_fwd(bamd_handle *h, const float *x, int64_t rows, float *const *y, float *dz_last, double *loss_part, int *nblk,
                        hipStream_t s) {
        const int grid = grid_for(rows);
        int cps = 0, tps = 0;
        const int fr = Fr(h), ngroup = (int)((rows + 63) / 64);
        const int s_in = small_splits(h, rows, fr / 16, &cps), s_out = small_splits(h, rows, (fr + 15) / 16, &tps);
        if (s_in > 1 && s_out > 1 && fr / 16 >= 3) {
            FusedState *st = state_of(h);
            int rc = st->wpart.ensure((size_t)s_in * ngroup * 4 * 13 * 64 * sizeof(v4));
            if (rc) return rc;
            if (rows <= in16_rows())
                hipLaunchKernelGGL((wide_small_in16_kernel<F, WRT>), dim3(s_in, (unsigned)(4 * ngroup)), dim3(256), 0, s, (const v4 *)h->packed.p + N::wf_off(0),
                                   N::wcount(0), x, rows, (v4 *)st->wpart.p, cps, fr);
            else
                hipLaunchKernelGGL((wide_small_in_kernel<F, WRT>), dim3(s_in, ngroup), dim3(256), 0, s, (const v4 *)h->packed.p + N::wf_off(0), N::wcount(0), x, rows,
                                   (v4 *)st->wpart.p, cps, fr);
            hipLaunchKernelGGL((wide_train_fwd_kernel<F, Z, true, WRT, true>), dim3(grid), dim3(256), 0, s, (const v4 *)h->packed.p, x, rows, y[1], y[2],
                               y[3], y[4], y[5], y[6], y[7], (void *)dz_last, 0, loss_part, fr, Zr(h), (const v4 *)st->wpart.p, s_in);
            const bool out16 = rows <= out16_rows();
            if (out16)
                hipLaunchKernelGGL((wide_small_out16_kernel<F, Z, WRT>), dim3(s_out, (unsigned)(4 * ngroup)), dim3(256), 0, s, (const v4 *)h->packed.p, x, rows,
                                   (const float *)y[7], (void *)dz_last, loss_part, tps, fr, 0);
            else
                hipLaunchKernelGGL((wide_small_out_kernel<F, Z, WRT>), dim3(s_out, ngroup), dim3(256), 0, s, (const v4 *)h->packed.p, x, rows, (const float *)y[7],
                                   (void *)dz_last, loss_part, tps, fr, 0);
            *nblk = s_out * ngroup * (out16 ? 4 : 1);
            BAMD_HIP(hipGetLastError());
            return BAMD_OK;
        }
        hipLaunchKernelGGL((wide_train_fwd_kernel<F, Z, true, WRT>), dim3(grid), dim3(256), 0, s, (const v4 *)h->packed.p, x, rows, y[1], y[2], y[3],
                           y[4], y[5], y[6], y[7], (void *)dz_last, 0, loss_part, Fr(h), Zr(h));
        *nblk = grid;
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    // validation pass: sum of squared errors / F into *loss_sum, optional reconstruction (float32 staging for non-float32 rows and
    // normalise-on-load, as in encode)
    static int forward_loss(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *recon, int recon_dtype,
                            double *loss_sum, hipStream_t s) {
        const size_t xes = x_dtype == BAMD_F64 ? 8 : 4, oes = recon_dtype == BAMD_F64 ? 8 : 4;
        const int fr = Fr(h), zr = Zr(h);
        const int64_t chunk = 1 << 16;
        int rc = h->lossp.ensure(sizeof(double) * 4096 * ((n + chunk - 1) / chunk > 0 ? (n + chunk - 1) / chunk : 1));
        if (rc) return rc;
        int nblk = 0;
        for (int64_t r0 = 0; r0 < n; r0 += chunk) {
            const int64_t rows = n - r0 < chunk ? n - r0 : chunk;
            const void *src = (const char *)x + (size_t)r0 * fr * xes;
            if (features || x_dtype != BAMD_F32) {
                rc = h->work.ensure((size_t)rows * fr * sizeof(float));
                if (rc) return rc;
                rc = features ? launch_normalize(src, x_dtype, rows, fr, features, h->work.p, BAMD_F32, s)
                              : launch_convert(src, x_dtype, h->work.p, BAMD_F32, rows * fr, s);
                if (rc) return rc;
                src = h->work.p;
            }
            const int grid = grid_for(rows);
            void *rdst = recon ? (void *)((char *)recon + (size_t)r0 * fr * oes) : nullptr;
            if (small_pass(h, rows)) {      // validation batches of the reference's sizes (60 rows ...): the split launches (see wide_fwd)
                FusedState *st = state_of(h);
                int cps = 0, tps = 0;
                const int ngroup = (int)((rows + 63) / 64);
                const int s_in = small_splits(h, rows, fr / 16, &cps), s_out = small_splits(h, rows, (fr + 15) / 16, &tps);
                rc = st->wpart.ensure((size_t)s_in * ngroup * 4 * 13 * 64 * sizeof(v4) + (size_t)ngroup * 64 * 200 * sizeof(float));
                if (rc) return rc;
                float *y7 = (float *)((char *)st->wpart.p + (size_t)s_in * ngroup * 4 * 13 * 64 * sizeof(v4));
                if (rows <= in16_rows())
                    hipLaunchKernelGGL((wide_small_in16_kernel<F, WRT>), dim3(s_in, (unsigned)(4 * ngroup)), dim3(256), 0, s,
                                       (const v4 *)h->packed.p + N::wf_off(0), N::wcount(0), (const float *)src, rows, (v4 *)st->wpart.p, cps, fr);
                else
                    hipLaunchKernelGGL((wide_small_in_kernel<F, WRT>), dim3(s_in, ngroup), dim3(256), 0, s, (const v4 *)h->packed.p + N::wf_off(0), N::wcount(0),
                                       (const float *)src, rows, (v4 *)st->wpart.p, cps, fr);
                hipLaunchKernelGGL((wide_train_fwd_kernel<F, Z, false, WRT, true>), dim3(grid), dim3(256), 0, s, (const v4 *)h->packed.p, (const float *)src,
                                   rows, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, y7,
                                   (void *)nullptr, 0, (double *)nullptr, fr, zr, (const v4 *)st->wpart.p, s_in);
                const bool out16 = rows <= out16_rows();
                if (out16)
                    hipLaunchKernelGGL((wide_small_out16_kernel<F, Z, WRT, false>), dim3(s_out, (unsigned)(4 * ngroup)), dim3(256), 0, s, (const v4 *)h->packed.p,
                                       (const float *)src, rows, (const float *)y7, rdst, (double *)h->lossp.p + nblk, tps, fr, recon_dtype == BAMD_F64 ? 1 : 0);
                else
                    hipLaunchKernelGGL((wide_small_out_kernel<F, Z, WRT, false>), dim3(s_out, ngroup), dim3(256), 0, s, (const v4 *)h->packed.p, (const float *)src,
                                       rows, (const float *)y7, rdst, (double *)h->lossp.p + nblk, tps, fr, recon_dtype == BAMD_F64 ? 1 : 0);
                nblk += s_out * ngroup * (out16 ? 4 : 1);
                continue;
            }
            hipLaunchKernelGGL((wide_train_fwd_kernel<F, Z, false, WRT>), dim3(grid), dim3(256), 0, s, (const v4 *)h->packed.p, (const float *)src,
                               rows, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr,
                               (float *)nullptr, (float *)nullptr, rdst,
                               recon_dtype == BAMD_F64, (double *)h->lossp.p + nblk, fr, zr);
            nblk += grid;
        }
        hipLaunchKernelGGL(sum_loss_wide_k, dim3(1), dim3(256), 0, s, (const double *)h->lossp.p, nblk, 1.0 / fr, loss_sum);
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static int wide_bwd(bamd_handle *h, int64_t rows, float *const *y, float *const *dz, const float *dz_latent, hipStream_t s) {
        int cps = 0;
        const int fr = Fr(h), ngroup = (int)((rows + 63) / 64);
        const int s_in = small_splits(h, rows, fr / 16, &cps);
        if (s_in > 1 && fr / 16 >= 3) {      // small batches: de4's input-gradient product split over the wide dimension (see wide_fwd)
            FusedState *st = state_of(h);
            int rc = st->wpart.ensure((size_t)s_in * ngroup * 4 * 13 * 64 * sizeof(v4));
            if (rc) return rc;
            if (rows <= in16_rows())
                hipLaunchKernelGGL((wide_small_in16_kernel<F, WRT>), dim3(s_in, (unsigned)(4 * ngroup)), dim3(256), 0, s, (const v4 *)h->packed.p + N::wb_off(7),
                                   N::wcount(7), (const float *)dz[7], rows, (v4 *)st->wpart.p, cps, fr);
            else
                hipLaunchKernelGGL((wide_small_in_kernel<F, WRT>), dim3(s_in, ngroup), dim3(256), 0, s, (const v4 *)h->packed.p + N::wb_off(7), N::wcount(7),
                                   (const float *)dz[7], rows, (v4 *)st->wpart.p, cps, fr);
            hipLaunchKernelGGL((wide_train_bwd_kernel<F, Z, WRT, true>), dim3(grid_for(rows)), dim3(256), 0, s, (const v4 *)h->packed.p,
                               (const float *)dz[7], rows, (const float *)y[1], (const float *)y[2], (const float *)y[3], (const float *)y[5],
                               (const float *)y[6], (const float *)y[7], dz[0], dz[1], dz[2], dz[3], dz[4], dz[5], dz[6], dz_latent, fr, Zr(h),
                               (const v4 *)st->wpart.p, s_in);
            BAMD_HIP(hipGetLastError());
            return BAMD_OK;
        }
        hipLaunchKernelGGL((wide_train_bwd_kernel<F, Z, WRT>), dim3(grid_for(rows)), dim3(256), 0, s, (const v4 *)h->packed.p,
                           (const float *)dz[7], rows, (const float *)y[1], (const float *)y[2], (const float *)y[3], (const float *)y[5],
                           (const float *)y[6], (const float *)y[7], dz[0], dz[1], dz[2], dz[3], dz[4], dz[5], dz[6], dz_latent, Fr(h), Zr(h));
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static const FusedOps *ops() {
        static const FusedOps o = {setup, encode, decode, forward_loss, nullptr, nullptr, wide_fwd, wide_bwd, nullptr, true, small_pass};
        return &o;
    }
};

// The same models on a BAMD_MODE_BF16 handle: encode / decode on the bf16 kernels above (HBM-bound), training and validation on the
// fp32 wide-layer kernels (fp32 master weights, as in the 24-column bf16 mode).
template <int F, int Z> struct ImplWideBf16 {
    using N = Net<F, Z>;
    using W = ImplWide<F, Z>;
    static constexpr int KB = (F + 31) / 32, KT = tiles(F);
    // float32 rows of a multiple of 16 bytes take the encode kernel with the decoupled row stream (wide_bf16_encode_dma_kernel)
    static constexpr bool kDma = (F * 4) % 16 == 0 && F / 32 >= kDmaRing;
    static constexpr size_t dec_lds_bytes() { return (size_t)kDecSlots * 7 * 1024 + 4 * 32 * (128 + 4) * 4 + (size_t)(N::bf_off(N::L) - N::bf_off(4)) * 16; }
    static int dec_grid(const FusedState *st, int64_t rows) {      // persistent: the loader streams tile after tile across row groups
        const int64_t g = (rows + 127) / 128;
        return (int)(g < 1 ? 1 : (g > 2 * st->nwg_max ? 2 * st->nwg_max : g));
    }
    static constexpr size_t dma_lds_bytes() { return (size_t)kDmaRing * kDmaChunk + kDmaStage * 13 * 1024 + (size_t)(N::bf_off(4) - N::bf_off(0)) * 16; }
    static int setup(bamd_handle *h, FusedState *st) {
        int rc = build_maps<F, Z, false>(h, st);
        if (rc) return rc;
        std::vector<int> s0((size_t)KB * 13 * 64 * 8, -1), s7((size_t)KT * 7 * 64 * 8, -1);
        for (int c = 0; c < KB; ++c)
            for (int t = 0; t < 13; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int i = lane & 15, g = lane >> 4;
                        const int nf = slot_feature(200, t, i / 4, i % 4), kf = 32 * c + 8 * g + j;
                        if (nf >= 0 && kf < F) s0[(((size_t)c * 13 + t) * 64 + lane) * 8 + j] = N::w_off(0) + nf * F + kf;
                    }
        for (int t = 0; t < KT; ++t)
            for (int c = 0; c < 7; ++c)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int i = lane & 15, g = lane >> 4, q = 2 * c + j / 4;
                        const int nf = slot_feature(F, t, i / 4, i % 4), kf = q < 13 ? slot_feature(200, q, g, j % 4) : -1;
                        if (nf >= 0 && kf >= 0) s7[(((size_t)t * 7 + c) * 64 + lane) * 8 + j] = N::w_off(7) + nf * 200 + kf;
                    }
        // W7^T for the input-gradient product of training, in en1's [chunk of 32 wide features][tile of the 200 side] order
        std::vector<int> s7t((size_t)KB * 13 * 64 * 8, -1);
        for (int c = 0; c < KB; ++c)
            for (int t = 0; t < 13; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int i = lane & 15, g = lane >> 4;
                        const int nf = slot_feature(200, t, i / 4, i % 4), kf = 32 * c + 8 * g + j;
                        if (nf >= 0 && kf < F) s7t[(((size_t)c * 13 + t) * 64 + lane) * 8 + j] = N::w_off(7) + kf * 200 + nf;
                    }
        // W0 once more for the DMA encode kernel: full chunks in ITS k order -- k slot (g, j) <-> feature 16 (j >> 2) + 4 g + (j & 3) of the
        // chunk, what a lane holds after reading bytes 16 g .. + 15 of each 64-byte half of the row's chunk; a partial last chunk natural
        std::vector<int> s0p((size_t)KB * 13 * 64 * 8, -1);
        for (int c = 0; c < KB; ++c)
            for (int t = 0; t < 13; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int i = lane & 15, g = lane >> 4;
                        const int nf = slot_feature(200, t, i / 4, i % 4);
                        const int kf = 32 * c + (c < F / 32 ? 16 * (j >> 2) + 4 * g + (j & 3) : 8 * g + j);
                        if (nf >= 0 && kf < F) s0p[(((size_t)c * 13 + t) * 64 + lane) * 8 + j] = N::w_off(0) + nf * F + kf;
                    }
        // the narrow layers for chain_bf16_pair: layer l as [k block c][output tile t] fragments, lane (i, g) element j =
        // W_l[slot (t, i / 4, i % 4) of the outputs][slot (tile 2 c + j / 4, g, j % 4) of the inputs]
        auto chain_map = [](int l0) {
            std::vector<int> m;
            for (int l = l0; l < l0 + 3; ++l) {
                const int K = N::dim(l), NN = N::dim(l + 1), kbk = (tiles(K) + 1) / 2, nt = tiles(NN);
                const size_t off = m.size();
                m.resize(off + (size_t)kbk * nt * 64 * 8, -1);
                for (int c = 0; c < kbk; ++c)
                    for (int t = 0; t < nt; ++t)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 8; ++j) {
                                const int i = lane & 15, g = lane >> 4, q = 2 * c + j / 4;
                                const int nf = slot_feature(NN, t, i / 4, i % 4), kf = q < tiles(K) ? slot_feature(K, q, g, j % 4) : -1;
                                if (nf >= 0 && kf >= 0) m[off + (((size_t)c * nt + t) * 64 + lane) * 8 + j] = N::w_off(l) + nf * K + kf;
                            }
            }
            return m;
        };
        const std::vector<int> sce = chain_map(1), scd = chain_map(4);
        const std::vector<int> *srcs[6] = {&s0, &s7, &s7t, &s0p, &sce, &scd};
        if constexpr (kDma)
            BAMD_HIP(hipFuncSetAttribute((const void *)wide_bf16_encode_dma_kernel<F, Z>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dma_lds_bytes()));
        BAMD_HIP(hipFuncSetAttribute((const void *)wide_bf16_decode_kernel<F, Z, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dec_lds_bytes()));
        BAMD_HIP(hipFuncSetAttribute((const void *)wide_bf16_decode_kernel<F, Z, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dec_lds_bytes()));
        for (int k = 0; k < 6; ++k) {
            st->wb_count[k] = (int)srcs[k]->size();
            rc = st->wb_src[k].ensure(srcs[k]->size() * sizeof(int));
            if (rc) return rc;
            rc = st->wb[k].ensure(srcs[k]->size() * sizeof(__bf16) + 4096);
            if (rc) return rc;
            BAMD_HIP(hipMemcpy(st->wb_src[k].p, srcs[k]->data(), srcs[k]->size() * sizeof(int), hipMemcpyHostToDevice));
        }
        return BAMD_OK;
    }
    static int pack_extra(bamd_handle *h, FusedState *st, hipStream_t s) {
        for (int k = 0; k < 6; ++k)
            hipLaunchKernelGGL(pack_wide_bf16_k, dim3((st->wb_count[k] + 255) / 256), dim3(256), 0, s, (const float *)h->params.p,
                               (const int *)st->wb_src[k].p, st->wb_count[k], (__bf16 *)st->wb[k].p);
        st->wb_stale = false;
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static int grid_for(int64_t n) {
        const int64_t wg = ((n + 31) / 32 + 3) / 4;
        return (int)(wg < 1 ? 1 : (wg > 2048 ? 2048 : wg));
    }
    static int encode(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *z, int z_dtype,
                      hipStream_t s) {
        FusedState *st = state_of(h);
        if (st->wb_stale) { int rc = pack_extra(h, st, s); if (rc) return rc; }
        const size_t xes = x_dtype == BAMD_F64 ? 8 : 4, zes = z_dtype == BAMD_F64 ? 8 : 4;
        for (int64_t r0 = 0; r0 < n; r0 += W::kChunkRows) {
            const int64_t rows = n - r0 < W::kChunkRows ? n - r0 : W::kChunkRows;
            const void *src = (const char *)x + (size_t)r0 * F * xes;
            int src_f64 = x_dtype == BAMD_F64;
            if (features) {
                int rc = h->work.ensure((size_t)rows * F * sizeof(float));
                if (rc) return rc;
                rc = launch_normalize(src, x_dtype, rows, F, features, h->work.p, BAMD_F32, s);
                if (rc) return rc;
                src = h->work.p;
                src_f64 = 0;
            }
            const int64_t ngroup = (rows + 127) / 128;
            const dim3 grid((unsigned)(ngroup > 2048 ? 2048 : ngroup));
            void *zo = (void *)((char *)z + (size_t)r0 * Z * zes);
            if constexpr (kDma) {
                if (!src_f64) {      // persistent: one workgroup (4 compute + 2 loader waves, 150 KiB of LDS) per CU
                    hipLaunchKernelGGL((wide_bf16_encode_dma_kernel<F, Z>), dim3((unsigned)(ngroup > st->nwg_max ? st->nwg_max : ngroup)), dim3(384),
                                       dma_lds_bytes(), s, (const v4 *)h->packed.p, (const v4 *)st->wb[3].p, (const v4 *)st->wb[4].p, (const float *)src, rows, zo,
                                       z_dtype == BAMD_F64);
                    continue;
                }
            }
            if (src_f64)
                hipLaunchKernelGGL((wide_bf16_encode_kernel<F, Z, true>), grid, dim3(256), 0, s, (const v4 *)h->packed.p,
                                   (const v4 *)st->wb[0].p, src, rows, zo, z_dtype == BAMD_F64);
            else
                hipLaunchKernelGGL((wide_bf16_encode_kernel<F, Z, false>), grid, dim3(256), 0, s, (const v4 *)h->packed.p,
                                   (const v4 *)st->wb[0].p, src, rows, zo, z_dtype == BAMD_F64);
        }
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static int decode(bamd_handle *h, const void *z, int z_dtype, int64_t n, const double *features, const uint8_t *int_mask,
                      void *out, int out_dtype, hipStream_t s) {
        FusedState *st = state_of(h);
        if (st->wb_stale) { int rc = pack_extra(h, st, s); if (rc) return rc; }
        const size_t zes = z_dtype == BAMD_F64 ? 8 : 4, oes = out_dtype == BAMD_F64 ? 8 : 4;
        if (features && out_dtype != BAMD_F64) { set_error("decode with features needs a float64 output"); return BAMD_ERR_INVALID; }
        for (int64_t r0 = 0; r0 < n; r0 += W::kChunkRows) {
            const int64_t rows = n - r0 < W::kChunkRows ? n - r0 : W::kChunkRows;
            void *dst = (char *)out + (size_t)r0 * F * oes;
            void *kout = dst;
            int kout_f64 = out_dtype == BAMD_F64;
            if (features) {
                int rc = h->work.ensure((size_t)rows * F * sizeof(float));
                if (rc) return rc;
                kout = h->work.p;
                kout_f64 = 0;
            }
            if (kout_f64)
                hipLaunchKernelGGL((wide_bf16_decode_kernel<F, Z, true>), dim3(dec_grid(st, rows)), dim3(320), dec_lds_bytes(), s, (const v4 *)h->packed.p,
                                   (const v4 *)st->wb[1].p, (const v4 *)st->wb[5].p, (const void *)((const char *)z + (size_t)r0 * Z * zes), z_dtype == BAMD_F64, rows, kout);
            else
                hipLaunchKernelGGL((wide_bf16_decode_kernel<F, Z, false>), dim3(dec_grid(st, rows)), dim3(320), dec_lds_bytes(), s, (const v4 *)h->packed.p,
                                   (const v4 *)st->wb[1].p, (const v4 *)st->wb[5].p, (const void *)((const char *)z + (size_t)r0 * Z * zes), z_dtype == BAMD_F64, rows, kout);
            if (features) {
                int rc = launch_renormalize(kout, BAMD_F32, rows, F, features, int_mask, (double *)dst, s);
                if (rc) return rc;
            }
        }
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    // training: the row-local launches with en1 / de4 / de4's input-gradient product on the bf16 MFMA (BALER_AMD_BF16_WIDE_TRAIN=0: the
    // float32 launches, as before round 3); the weight-gradient products stay in float32 on what these launches store
    static bool bf16_train_on() {      // read per call (tests flip it)
        const char *e = getenv("BALER_AMD_BF16_WIDE_TRAIN");
        return !(e && e[0] == '0');
    }
    static int wide_fwd(bamd_handle *h, const float *x, int64_t rows, float *const *y, float *dz_last, double *loss_part, int *nblk,
                        hipStream_t s) {
        // (small batches: the float32 split launches -- the bf16 launches give 32 rows to a wave that walks the whole wide dimension)
        if (!bf16_train_on() || W::small_pass(h, rows)) return W::wide_fwd(h, x, rows, y, dz_last, loss_part, nblk, s);
        FusedState *st = state_of(h);
        if (st->wb_stale) { int rc = pack_extra(h, st, s); if (rc) return rc; }
        const int grid = grid_for(rows);
        if constexpr (F % 4 == 0) {
            if (st->dz16) {
                hipLaunchKernelGGL((wide_bf16_train_fwd_kernel<F, Z, true>), dim3(grid), dim3(256), 0, s, (const v4 *)h->packed.p,
                                   (const v4 *)st->wb[0].p, (const v4 *)st->wb[1].p, x, rows, y[1], y[2], y[3], y[4], y[5], y[6], y[7],
                                   (void *)dz_last, loss_part);
                *nblk = grid;
                BAMD_HIP(hipGetLastError());
                return BAMD_OK;
            }
        }
        hipLaunchKernelGGL((wide_bf16_train_fwd_kernel<F, Z, false>), dim3(grid), dim3(256), 0, s, (const v4 *)h->packed.p, (const v4 *)st->wb[0].p,
                           (const v4 *)st->wb[1].p, x, rows, y[1], y[2], y[3], y[4], y[5], y[6], y[7], (void *)dz_last, loss_part);
        *nblk = grid;
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static int wide_bwd(bamd_handle *h, int64_t rows, float *const *y, float *const *dz, const float *dz_latent, hipStream_t s) {
        if (!bf16_train_on() || W::small_pass(h, rows)) return W::wide_bwd(h, rows, y, dz, dz_latent, s);
        FusedState *st = state_of(h);
        if (st->wb_stale) { int rc = pack_extra(h, st, s); if (rc) return rc; }
        if constexpr (F % 4 == 0) {
            if (st->dz16) {
                hipLaunchKernelGGL((wide_bf16_train_bwd_kernel<F, Z, true>), dim3(grid_for(rows)), dim3(256), 0, s, (const v4 *)h->packed.p,
                                   (const v4 *)st->wb[2].p, (const void *)dz[7], rows, (const float *)y[1], (const float *)y[2],
                                   (const float *)y[3], (const float *)y[5], (const float *)y[6], (const float *)y[7], dz[0], dz[1], dz[2],
                                   dz[3], dz[4], dz[5], dz[6], dz_latent);
                BAMD_HIP(hipGetLastError());
                return BAMD_OK;
            }
        }
        hipLaunchKernelGGL((wide_bf16_train_bwd_kernel<F, Z, false>), dim3(grid_for(rows)), dim3(256), 0, s, (const v4 *)h->packed.p,
                           (const v4 *)st->wb[2].p, (const void *)dz[7], rows, (const float *)y[1], (const float *)y[2], (const float *)y[3],
                           (const float *)y[5], (const float *)y[6], (const float *)y[7], dz[0], dz[1], dz[2], dz[3], dz[4], dz[5], dz[6],
                           dz_latent);
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static const FusedOps *ops() {
        static const FusedOps o = {setup, encode, decode, W::forward_loss, nullptr, nullptr, wide_fwd, wide_bwd, pack_extra, true, W::small_pass};
        return &o;
    }
};

// Instantiated shapes.  The CMS 24-column model at the latent sizes its compression-ratio knob produces (latent = ceil(24 / ratio),
// baler.py:117-123): 1.6 -> 15, 2 -> 12, 2.4 -> 10, 3 -> 8, 4 -> 6, 5 -> 5, 6 -> 4, 8 -> 3, 12 -> 2; the wide-layer models the
// reference ships configs for: CFD_dense_AE(2500, 25) (CFD_project_animation), CFD_dense_AE(625, 7) (exafel1 / exafel2: 25 x 25
// blocks at ratio 100, exafel1_config.py:14-15,33) and the 512-column encoder of BASELINE configs[4].  Anything else runs on
// generic.hip, and bamd_create says so once (bamd_path_of() = BAMD_PATH_GENERIC).
#define BAMD_AE24(Z_) if (Impl<24, Z_>::matches(h)) return Impl<24, Z_>::ops();
#define BAMD_AE24_ALL BAMD_AE24(15) BAMD_AE24(12) BAMD_AE24(10) BAMD_AE24(8) BAMD_AE24(6) BAMD_AE24(5) BAMD_AE24(4) BAMD_AE24(3) BAMD_AE24(2)
// BALER_AMD_WIDE_CLASS: 0 = no run-time-width wide class (such shapes run layer by layer); "force" = the class also for the shapes
// that have an exact instantiation (tests: class vs exact on the same model)
static bool wide_class_on() { const char *e = getenv("BALER_AMD_WIDE_CLASS"); return !(e && e[0] == '0'); }
static bool wide_class_forced() { const char *e = getenv("BALER_AMD_WIDE_CLASS"); return e && e[0] == 'f'; }
// BALER_AMD_MID_HYBRID=0: 64..127-column tables on the small-batch class alone (large batches chunked on its kernels), as before round 5
static bool mid_width_hybrid() {
    const char *e = getenv("BALER_AMD_MID_HYBRID");
    return wide_class_on() && !(e && e[0] == '0');
}
static const FusedOps *find_small_ops(const bamd_handle *h) {
    if (h->mode != BAMD_MODE_F32) return nullptr;
    if (ImplInferClass<79, 31, true>::matches(h)) return ImplInferClass<79, 31, true>::ops();
    if (ImplInferClass<95, 31, true>::matches(h)) return ImplInferClass<95, 31, true>::ops();
    if (ImplInferClass<111, 31, true>::matches(h)) return ImplInferClass<111, 31, true>::ops();
    if (ImplInferClass<127, 31, true>::matches(h)) return ImplInferClass<127, 31, true>::ops();
    return nullptr;
}
static const FusedOps *find_ops(const bamd_handle *h) {
    if (h->mode == BAMD_MODE_BF16) {
        if (ImplWide<2500, 25>::matches(h)) return ImplWideBf16<2500, 25>::ops();
        if (ImplWide<625, 7>::matches(h)) return ImplWideBf16<625, 7>::ops();
        if (ImplWide<512, 6>::matches(h)) return ImplWideBf16<512, 6>::ops();
        // the 24-column model's bf16 kernels live in bf16.hip / bf16_train.hip; the fp32 kernels here serve its SMALL batches
        // (api.hip: the bf16 training kernels need ~3000 rows to beat the fp32 small-batch step)
        BAMD_AE24_ALL
        return nullptr;
    }
    if (h->mode != BAMD_MODE_F32) return nullptr;
    BAMD_AE24_ALL
    // any other narrow table: the class instantiations (run-time widths; Impl<F, Z, true>): up to 63 columns with a latent of up to
    // 31 on every kernel, 64..79 columns on the one-tile inference and the small-batch kernels (ImplInferClass).  80 columns and
    // more, or a latent above 31, run on generic.hip -- DESIGN.md section 8
    if (Impl<31, 15, true>::matches(h)) return Impl<31, 15, true>::ops();
    if (Impl<47, 15, true>::matches(h)) return Impl<47, 15, true>::ops();
    if (Impl<31, 31, true>::matches(h)) return Impl<31, 31, true>::ops();
    if (Impl<47, 31, true>::matches(h)) return Impl<47, 31, true>::ops();
    if (Impl<63, 15, true>::matches(h)) return Impl<63, 15, true>::ops();
    if (Impl<63, 31, true>::matches(h)) return Impl<63, 31, true>::ops();
    if (!wide_class_forced()) {
        if (ImplWide<512, 6>::matches(h)) return ImplWide<512, 6>::ops();
        if (ImplWide<2500, 25>::matches(h)) return ImplWide<2500, 25>::ops();
        if (ImplWide<625, 7>::matches(h)) return ImplWide<625, 7>::ops();
    }
    // 64..127 columns: one-tile inference kernels and small-batch training (two reconstruction tiles per wave); large batches layer-wise
    // 64..127 columns: the small-batch class (one-tile inference kernels + small-batch training, two reconstruction tiles per wave).  With
    // the wide class on it is only the SECOND state of such a handle (find_small_ops): encode / decode / forward and large training
    // batches run on the wide class -- measured at 1M rows (tools/bench_mid_width_wide.py) fwd_bwd AE(80,16) 100.8 -> 142.3 M rows/s,
    // AE(64,16) 104.8 -> 156.8, AE(100,1) 100.8 -> 147.7, AE(127,31) 94.7 -> 137.6, encode 0.954 -> 0.855 ms -- and the class kernels keep
    // the small batches (512-row step 29.7 us against the wide class's 160)
    if (!mid_width_hybrid()) {
        if (const FusedOps *o = find_small_ops(h)) return o;
    }
    // any other wide model with the reference's hidden widths (CFD_dense_AE(n_features, z_dim), models.py:192-209): class instantiations
    // of the wide-layer kernels with run-time widths: up to 4096 columns, a latent of up to 15 / 31 / 63
    if (wide_class_on()) {
        if (ImplWide<4096, 15, true>::matches(h)) return ImplWide<4096, 15, true>::ops();
        if (ImplWide<4096, 31, true>::matches(h)) return ImplWide<4096, 31, true>::ops();
        if (ImplWide<4096, 63, true>::matches(h)) return ImplWide<4096, 63, true>::ops();
    }
    return nullptr;
}
#undef BAMD_AE24
#undef BAMD_AE24_ALL

}  // namespace

#ifdef BAMD_DEBUG
// debug builds only (make HIPFLAGS+=-DBAMD_DEBUG; tools/check_lat4_imgs.py): copy the small-batch images of the last step to a
// device buffer.  Not part of the ABI: the shipped library does not export it.
extern "C" int bamd_debug_copy_imgs(bamd_handle *h, void *dst, size_t bytes) {
    if (!h || !dst) return BAMD_ERR_INVALID;
    FusedState *st = (FusedState *)h->fused_state;
    if (!st || !st->imgs.p || bytes > st->imgs.bytes) return BAMD_ERR_INVALID;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(h->device) != hipSuccess) return BAMD_ERR_HIP;
    const int rc = (int)hipMemcpy(dst, st->imgs.p, bytes, hipMemcpyDeviceToDevice);
    (void)hipSetDevice(prev);
    return rc;
}
#endif

// A handle of a 64..127-column table has TWO states: fused_state = the wide class (inference, large training batches), fused_small =
// the small-batch class with its own packed fragments in h->packed_small.  SmallScope swaps the pair in for the duration of one host call,
// so that the class's code sees an ordinary handle; the optimiser keeps the SMALL state's fragments current (the 512-row regime is the
// latency-critical one) and the wide state's are re-packed before its next use (one pack_k launch of ~100 KB).
struct SmallScope {
    bamd_handle *h;
    explicit SmallScope(bamd_handle *h_) : h(h_) { swap(); }
    ~SmallScope() { swap(); }
    void swap() { std::swap(h->fused_state, h->fused_small); std::swap(h->packed, h->packed_small); }
    SmallScope(const SmallScope &) = delete;
    SmallScope &operator=(const SmallScope &) = delete;
};
static FusedState *small_of(bamd_handle *h) { return (FusedState *)h->fused_small; }
static bool small_takes(bamd_handle *h, int64_t n) { return h->fused_small && n <= small_of(h)->latency_max_rows; }

static void state_env(FusedState *st) {
    if (const char *lr = getenv("BALER_AMD_LATENCY_ROWS")) st->latency_max_rows = atoll(lr);
    if (const char *l4 = getenv("BALER_AMD_LAT4_ROWS")) st->lat4_max_rows = atoll(l4);
    if (const char *ts = getenv("BALER_AMD_TAIL_SPLIT")) st->tail_split = ts[0] != '0';
}
int fused_setup(bamd_handle *h) {
    h->fused_ok = false;
    const FusedOps *ops = find_ops(h);
    if (!ops) return BAMD_OK;
    const char *env = getenv("BALER_AMD_FORCE_GENERIC");
    if (env && env[0] == '1') return BAMD_OK;
    FusedState *st = new FusedState();
    st->ops = ops;
    state_env(st);
    h->fused_state = st;
    int rc = ops->setup(h, st);
    if (rc) return rc;
    if (ops->wide_fwd && h->dims[0] <= 127 && mid_width_hybrid()) {
        if (const FusedOps *so = find_small_ops(h)) {
            FusedState *ss = new FusedState();
            ss->ops = so;
            state_env(ss);
            h->fused_small = ss;
            SmallScope sc(h);
            rc = so->setup(h, ss);
            if (rc) return rc;
        }
    }
    h->fused_ok = true;
    return BAMD_OK;
}

static void release_state(FusedState *st) {
    st->pack_src.release();
    st->slab_map.release();
    st->dz.release();
    st->imgs.release();
    st->dwpart.release();
    st->wpart.release();
    for (int k = 0; k < 6; ++k) { st->wb_src[k].release(); st->wb[k].release(); }
    st->sc_off.release();
    st->sc_idx.release();
    delete st;
}
void fused_teardown(bamd_handle *h) {
    if (FusedState *ss = small_of(h)) {
        release_state(ss);
        h->packed_small.release();
        h->fused_small = nullptr;
    }
    FusedState *st = state_of(h);
    if (!st) return;
    release_state(st);
    h->fused_state = nullptr;
}

void fused_scatter(bamd_handle *h, const int **sc_off, const int **sc_idx, void **packed) {
    *sc_off = nullptr; *sc_idx = nullptr; *packed = nullptr;
    if (!h->fused_ok) return;
    if (FusedState *ss = small_of(h)) {      // two states: Adam refreshes the small-batch state's fragments, the wide state's go stale
        *sc_off = (const int *)ss->sc_off.p;
        *sc_idx = (const int *)ss->sc_idx.p;
        *packed = h->packed_small.p;
        return;
    }
    FusedState *st = state_of(h);
    *sc_off = (const int *)st->sc_off.p;
    *sc_idx = (const int *)st->sc_idx.p;
    *packed = h->packed.p;
}

static int pack_state(bamd_handle *h, hipStream_t s) {
    FusedState *st = state_of(h);
    hipLaunchKernelGGL(pack_k, dim3((st->packed_floats + 255) / 256), dim3(256), 0, s, (const float *)h->params.p,
                       (const int *)st->pack_src.p, st->packed_floats, (float *)h->packed.p);
    BAMD_HIP(hipGetLastError());
    st->packed_stale = false;
    if (st->ops->pack_extra) return st->ops->pack_extra(h, st, s);
    return BAMD_OK;
}
int fused_pack(bamd_handle *h, hipStream_t s) {
    if (!h->fused_ok) return BAMD_OK;
    if (h->fused_small) {
        SmallScope sc(h);
        const int rc = pack_state(h, s);
        if (rc) return rc;
    }
    return pack_state(h, s);
}
// before a call on the FIRST state of a two-state handle: its fragments if an optimiser step left them behind
static int refresh_primary(bamd_handle *h, hipStream_t s) {
    if (h->fused_small && state_of(h)->packed_stale) return pack_state(h, s);
    return BAMD_OK;
}

bool fused_trains(const bamd_handle *h) {   // false: large-batch training of this handle runs layer by layer
    if (!h->fused_ok) return false;
    return ((const FusedState *)h->fused_state)->ops->throughput_training;
}
int64_t fused_latency_rows(const bamd_handle *h) {   // the handle's small-batch limit (default 12288; BALER_AMD_LATENCY_ROWS)
    if (h->fused_ok && h->fused_small) return ((const FusedState *)h->fused_small)->latency_max_rows;
    return h->fused_ok ? ((const FusedState *)h->fused_state)->latency_max_rows : 0;
}
bool fused_has_bf16_kernels(const bamd_handle *h) {
    return ImplWide<2500, 25>::matches(h) || ImplWide<625, 7>::matches(h) || ImplWide<512, 6>::matches(h);
}
bool fused_serves_bf16_inference(const bamd_handle *h) {   // wide models in the bf16 mode: encode / decode live in fused.hip
    return h->fused_ok && ((const FusedState *)h->fused_state)->ops->pack_extra != nullptr;
}

void fused_params_changed(bamd_handle *h) {   // after an optimiser step: further packed copies are refreshed on demand
    if (h->fused_ok && state_of(h)->ops->pack_extra) state_of(h)->wb_stale = true;
    if (h->fused_ok && h->fused_small) state_of(h)->packed_stale = true;
}

static bool wide_train_on() {   // BALER_AMD_WIDE_TRAIN=0: every layer of a wide model's training pass on the layer-wise kernels
    const char *e = getenv("BALER_AMD_WIDE_TRAIN");      // read per call (a training pass is milliseconds): tests toggle it
    return !(e && e[0] == '0');
}
bool fused_wide_train(const bamd_handle *h) {
    return h->fused_ok && ((const FusedState *)h->fused_state)->ops->wide_fwd && wide_train_on();
}
int fused_wide_train_forward(bamd_handle *h, const float *x, int64_t rows, float *const *y, float *dz_last, double *loss_part, int *nblk,
                             hipStream_t s) {
    if (!fused_wide_train(h)) return BAMD_ERR_UNSUPPORTED;
    if (const int rc = refresh_primary(h, s)) return rc;
    return state_of(h)->ops->wide_fwd(h, x, rows, y, dz_last, loss_part, nblk, s);
}
// BF16 handles of a wide model whose last layer's weight gradient runs on dw_wide_bf16_k: dL/drecon (the largest array of the pass)
// is stored as bfloat16 by the forward launch and read as such by the backward launch and that kernel
// this batch of a wide model runs on the split (small-batch) float32 launches -- also on a BF16 handle, whose weight-gradient kernels
// then read float32 activations / gradients
bool fused_wide_small(const bamd_handle *h, int64_t rows) {
    if (!fused_wide_train(h)) return false;
    const FusedOps *o = ((const FusedState *)h->fused_state)->ops;
    return o->wide_small && o->wide_small(h, rows);
}
void fused_wide_set_dz16(bamd_handle *h, bool on) {
    if (h->fused_ok && h->fused_state) ((FusedState *)h->fused_state)->dz16 = on;
}
int fused_wide_train_backward(bamd_handle *h, int64_t rows, float *const *y, float *const *dz, const float *dz_latent, hipStream_t s) {
    if (!fused_wide_train(h)) return BAMD_ERR_UNSUPPORTED;
    return state_of(h)->ops->wide_bwd(h, rows, y, dz, dz_latent, s);
}

int fused_encode(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *z, int z_dtype,
                 hipStream_t s) {
    if (const int rc = refresh_primary(h, s)) return rc;
    return state_of(h)->ops->encode(h, x, x_dtype, n, features, z, z_dtype, s);
}
int fused_decode(bamd_handle *h, const void *z, int z_dtype, int64_t n, const double *features, const uint8_t *int_mask,
                 void *out, int out_dtype, hipStream_t s) {
    if (!state_of(h)->ops->decode)
        return generic_forward(h, z, z_dtype, n, nullptr, h->L / 2, h->L, out, out_dtype, features, int_mask, s);
    if (const int rc = refresh_primary(h, s)) return rc;
    return state_of(h)->ops->decode(h, z, z_dtype, n, features, int_mask, out, out_dtype, s);
}
int fused_forward_loss(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *recon,
                       int recon_dtype, double *loss_sum, hipStream_t s) {
    if (!state_of(h)->ops->forward_loss) return generic_forward_loss(h, x, x_dtype, n, features, recon, recon_dtype, loss_sum, s);
    if (const int rc = refresh_primary(h, s)) return rc;
    return state_of(h)->ops->forward_loss(h, x, x_dtype, n, features, recon, recon_dtype, loss_sum, s);
}
int fused_fwd_bwd(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads,
                  hipStream_t s) {
    if (small_takes(h, n)) {     // (the small state's fragments are always current: fused_scatter)
        SmallScope sc(h);
        return state_of(h)->ops->fwd_bwd(h, x, x_dtype, n, features, grads, s);
    }
    if (!state_of(h)->ops->fwd_bwd) return generic_fwd_bwd(h, x, x_dtype, n, features, grads, s);
    if (const int rc = refresh_primary(h, s)) return rc;
    return state_of(h)->ops->fwd_bwd(h, x, x_dtype, n, features, grads, s);
}
static int train_step_on(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads,
                         void *params, void *m, void *v, const bamd_adam &hp, double *loss_accum, hipStream_t s) {
    if (!state_of(h)->ops->train_step) return BAMD_ERR_UNSUPPORTED;
    FusedState *st = state_of(h);
    AdamArgs ad;
    ad.params = (float *)params; ad.pcopy = (float *)h->params.p; ad.m = (float *)m; ad.v = (float *)v;
    ad.packed = (float *)h->packed.p;
    ad.sc_off = (const int *)st->sc_off.p; ad.sc_idx = (const int *)st->sc_idx.p;
    ad.loss_accum = loss_accum;
    ad.b1 = hp.beta1; ad.b2 = hp.beta2; ad.eps = hp.eps;      // same scalars as launch_adam (elementwise.hip)
    ad.step_size = hp.lr / (1.0 - pow(hp.beta1, (double)hp.step));
    ad.bc2_sqrt = sqrt(1.0 - pow(hp.beta2, (double)hp.step));
    return st->ops->train_step(h, x, x_dtype, n, features, grads, ad, s);
}

int fused_train_step(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads,
                     void *params, void *m, void *v, const bamd_adam &hp, double *loss_accum, hipStream_t s) {
    if (!h->fused_ok) return BAMD_ERR_UNSUPPORTED;
    if (small_takes(h, n)) {
        state_of(h)->packed_stale = true;         // the step's fused Adam refreshes the small state's fragments only
        SmallScope sc(h);
        return train_step_on(h, x, x_dtype, n, features, grads, params, m, v, hp, loss_accum, s);
    }
    return train_step_on(h, x, x_dtype, n, features, grads, params, m, v, hp, loss_accum, s);
}

#ifdef BAMD_LAT_TRACE
extern "C" int bamd_debug_lat_trace(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lat_trace), sizeof(unsigned long long) * (n < 64 ? n : 64));
}
#endif

}  // namespace bamd
