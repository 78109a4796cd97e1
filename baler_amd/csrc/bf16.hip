// bf16 inference mode (BAMD_MODE_BF16): encode / decode / forward_loss of AE(F, Z) on v_mfma_f32_16x16x32_bf16.
//
// A THROUGHPUT mode, not the parity mode: weights and layer inputs are rounded to bfloat16 (8 significant
// bits), accumulation is fp32; outputs follow the fp64 reference to ~3e-3 relative (measured, tests), against
// 1e-7 for the fp32 MFMA mode.  Training calls on a bf16 handle run on the fp32 layer-wise kernels.
//
// Same transposed register chain as fused.hip (Y^T = W X^T, batch rows on the MFMA column, a layer's C tiles
// ARE the next layer's B operand), re-derived for the 16x16x32 shape:
//   * one MFMA contracts 32 input features; lane (j = lane & 15, g = lane >> 4) supplies 8 consecutive k slots.
//     Two output tiles (2p, 2p+1) of a layer give lane (j, g) the 8 values {16(2p) + 4g + r, 16(2p+1) + 4g + r},
//     which after LeakyReLU and ONE v_cvt_pk_bf16_f32 per pair of values are k-block p of the next layer with
//     slot (g, e) -> feature 32p + 16(e >> 2) + 4g + (e & 3); the packed weights carry that permutation.
//   * a bf16 MFMA takes 16 cycles and eats a 1-KiB A fragment: 256 B/clk/CU if every MFMA fetched its own, 4x
//     the L1 rate and 2x the LDS rate.  So (a) the whole half-model (80 / 84 fragments) is staged ONCE per
//     workgroup into LDS, and (b) every fragment read feeds kMB = 4 batch tiles (64 rows per wave per pass):
//     8 waves x 1 KiB per 4 MFMAs = 32 B/clk/CU of LDS reads.
//   * 8 waves per workgroup (2 per SIMD, <= 256 registers) share the LDS copy; the VALU work of one wave
//     (LeakyReLU + conversions: ~2 instructions per value) overlaps the MFMAs of the other.
// Encode at fp64 I/O moves 312 B/row: at these MFMA rates the kernel is HBM-bound, not MFMA-bound.
#include "bf16.hpp"

#include <cstdlib>

namespace bamd {
namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
using v4 = float __attribute__((ext_vector_type(4)));
#ifndef BAMD_BF16_KMB
#define BAMD_BF16_KMB 4
#endif
#ifndef BAMD_BF16_WAVES
#define BAMD_BF16_WAVES 8
#endif
constexpr int kMB = BAMD_BF16_KMB;        // 16-row batch tiles per wave per pass
constexpr int kWaves = BAMD_BF16_WAVES;   // waves per workgroup
constexpr int kRowsPerPass = 16 * kMB;

template <int F, int Z> struct BNet {
    static constexpr int L = 8;
    __host__ __device__ static constexpr int dim(int i) {
        return i == 0 ? F : i == 1 ? 200 : i == 2 ? 100 : i == 3 ? 50 : i == 4 ? Z : i == 5 ? 50 : i == 6 ? 100 : i == 7 ? 200 : F;
    }
    __host__ __device__ static constexpr bool act(int l) { return !(l == 3 || l == 7); }
    __host__ __device__ static constexpr int kb(int l) { return (dim(l) + 31) / 32; }       // 32-wide k blocks
    __host__ __device__ static constexpr int nt(int l) { return (dim(l + 1) + 15) / 16; }   // 16-wide output tiles
    __host__ __device__ static constexpr int frags(int l) { return kb(l) * nt(l); }
    // fragment / bias offsets inside a half (encoder = layers 0..3, decoder = 4..7)
    __host__ __device__ static constexpr int f_off(int l) { int s = 0; for (int j = (l < 4 ? 0 : 4); j < l; ++j) s += frags(j); return s; }
    __host__ __device__ static constexpr int b_off(int l) { int s = 0; for (int j = (l < 4 ? 0 : 4); j < l; ++j) s += nt(j) * 4; return s; }
    __host__ __device__ static constexpr int half_frags(int h) { return f_off(4 * h + 3) + frags(4 * h + 3); }
    __host__ __device__ static constexpr int half_bias(int h) { return b_off(4 * h + 3) + nt(4 * h + 3) * 4; }
    // input feature behind k slot (block q, lane group g, element e) of layer l; -1 = zero padding.
    // Layers fed from memory (0 and 4) use 8 consecutive features per lane; the others the C-tile pairing.
    __host__ __device__ static constexpr int in_feature(int l, int q, int g, int e) {
        int f = (l == 0 || l == 4) ? 32 * q + 8 * g + e : 32 * q + 16 * (e >> 2) + 4 * g + (e & 3);
        return f < dim(l) ? f : -1;
    }
    __host__ __device__ static constexpr int w_off_c(int l) { int s = 0; for (int j = 0; j < l; ++j) s += dim(j + 1) * dim(j) + dim(j + 1); return s; }
    __host__ __device__ static constexpr int b_off_c(int l) { return w_off_c(l) + dim(l + 1) * dim(l); }
    __host__ __device__ static constexpr int nparams() { return w_off_c(L); }
    static_assert(F <= 32 && Z <= 32, "first layers read one 32-feature block from memory");
    static constexpr size_t lds_bytes(int h) { return (size_t)half_frags(h) * 1024 + (size_t)half_bias(h) * 16 + 4 * 32 * sizeof(double); }
};

__device__ __forceinline__ v4 mfma_bf16(bf8 a, bf8 b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// LeakyReLU as max(x, 0.01 x) with FOUR v_mul_f32, not two v_pk_mul_f32: a packed fp32 multiply does not overlap the bf16 MFMA of its
// own wave at all (tools/probe/valu_beside_mfma_probe.hip: + 17 cycles for the first one in an MFMA slot, a plain VALU instruction
// + 0.5); measured on this kernel: float32 rows 11.7 -> 12.5 G rows/s encode, float64 rows unchanged (bound by their conversions)
__device__ __forceinline__ void lrelu4(v4 &a) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = __builtin_elementwise_maximum(a[r], a[r] * 0.01f);
}
__device__ __forceinline__ bf8 pack8(const v4 &lo, const v4 &hi) {
    bf8 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) { o[r] = (__bf16)lo[r]; o[4 + r] = (__bf16)hi[r]; }
    return o;
}

// One Linear (+ LeakyReLU) layer for kMB batch tiles.  w: this layer's fragments in LDS ([q][t], 64 lanes x 16 B
// each, already offset by the lane); bias: [t][g] float4.  Output tiles are produced in pairs = k blocks of the next layer.
template <int KB, int NT, bool ACT>
__device__ __forceinline__ void blayer(const bf8 (&in)[KB][kMB], bf8 (&out)[(NT + 1) / 2][kMB], const bf8 *w, const v4 *bias, int g) {
#pragma unroll
    for (int p = 0; p < (NT + 1) / 2; ++p) {
        const bool has1 = 2 * p + 1 < NT;
        v4 acc0[kMB], acc1[kMB];
        const v4 b0 = bias[(2 * p) * 4 + g];
        const v4 b1 = has1 ? bias[(2 * p + 1) * 4 + g] : (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mb = 0; mb < kMB; ++mb) { acc0[mb] = b0; acc1[mb] = b1; }
        // fragment reads one step ahead of their MFMAs (hipcc placed every ds_read_b128 right in front of its use: a full LDS
        // round trip before each group of four MFMAs, SQ_WAIT_ANY 45-51 %); sched_barrier pins the order
        bf8 a0 = w[(2 * p) * 64], a1 = has1 ? w[(2 * p + 1) * 64] : a0;
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            bf8 n0 = a0, n1 = a1;
            if (q + 1 < KB) {
                n0 = w[((q + 1) * NT + 2 * p) * 64];
                n1 = has1 ? w[((q + 1) * NT + 2 * p + 1) * 64] : n0;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < kMB; ++mb) {
                acc0[mb] = mfma_bf16(a0, in[q][mb], acc0[mb]);
                if (has1) acc1[mb] = mfma_bf16(a1, in[q][mb], acc1[mb]);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = n0;
            a1 = n1;
        }
#pragma unroll
        for (int mb = 0; mb < kMB; ++mb) {
            if (ACT) { lrelu4(acc0[mb]); if (has1) lrelu4(acc1[mb]); }
            out[p][mb] = pack8(acc0[mb], acc1[mb]);      // without a second tile acc1 = 0: zero k slots
        }
    }
}
// Two activated layers back to back with the roles swapped: ALL output tiles of the second layer are accumulators
// (NT1 x kMB tiles) and every k block of its input is consumed as soon as the first layer has produced it.  For
// en1 -> en2 this replaces the 200-feature activation (112 registers) + a second set of accumulators by 7 x 4
// accumulator tiles: the encoder fits 256 registers without scratch.
template <int KB, int NT, int NT1>
__device__ __forceinline__ void blayer_pair(const bf8 (&in)[KB][kMB], bf8 (&out)[(NT1 + 1) / 2][kMB], const bf8 *w, const v4 *bias,
                                            const bf8 *w1, const v4 *bias1, int g) {
    v4 acc[NT1][kMB];
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
        const v4 b = bias1[t * 4 + g];
#pragma unroll
        for (int mb = 0; mb < kMB; ++mb) acc[t][mb] = b;
    }
#pragma unroll
    for (int p = 0; p < (NT + 1) / 2; ++p) {
        const bool has1 = 2 * p + 1 < NT;
        v4 acc0[kMB], acc1[kMB];
        const v4 b0 = bias[(2 * p) * 4 + g];
        const v4 b1 = has1 ? bias[(2 * p + 1) * 4 + g] : (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mb = 0; mb < kMB; ++mb) { acc0[mb] = b0; acc1[mb] = b1; }
        // fragment reads one step ahead of their MFMAs (hipcc placed every ds_read_b128 right in front of its use: a full LDS
        // round trip before each group of four MFMAs, SQ_WAIT_ANY 45-51 %); sched_barrier pins the order
        bf8 a0 = w[(2 * p) * 64], a1 = has1 ? w[(2 * p + 1) * 64] : a0;
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            bf8 n0 = a0, n1 = a1;
            if (q + 1 < KB) {
                n0 = w[((q + 1) * NT + 2 * p) * 64];
                n1 = has1 ? w[((q + 1) * NT + 2 * p + 1) * 64] : n0;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < kMB; ++mb) {
                acc0[mb] = mfma_bf16(a0, in[q][mb], acc0[mb]);
                if (has1) acc1[mb] = mfma_bf16(a1, in[q][mb], acc1[mb]);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = n0;
            a1 = n1;
        }
        bf8 blk[kMB];
#pragma unroll
        for (int mb = 0; mb < kMB; ++mb) {
            lrelu4(acc0[mb]);
            if (has1) lrelu4(acc1[mb]);
            blk[mb] = pack8(acc0[mb], acc1[mb]);
        }
        {
            bf8 a = w1[(p * NT1) * 64];
#pragma unroll
            for (int t = 0; t < NT1; ++t) {
                bf8 nx = a;
                if (t + 1 < NT1) nx = w1[(p * NT1 + t + 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mb = 0; mb < kMB; ++mb) acc[t][mb] = mfma_bf16(a, blk[mb], acc[t][mb]);
                __builtin_amdgcn_sched_barrier(0);
                a = nx;
            }
        }
    }
#pragma unroll
    for (int p = 0; p < (NT1 + 1) / 2; ++p)
#pragma unroll
        for (int mb = 0; mb < kMB; ++mb) {
            lrelu4(acc[2 * p][mb]);
            if (2 * p + 1 < NT1) {
                lrelu4(acc[2 * p + 1][mb]);
                out[p][mb] = pack8(acc[2 * p][mb], acc[2 * p + 1][mb]);
            } else {
                out[p][mb] = pack8(acc[2 * p][mb], (v4){0.f, 0.f, 0.f, 0.f});
            }
        }
}

// Layer l (+ LeakyReLU) fused with the LAST layer of the half: every k block of the last layer's input is consumed
// as soon as its tile pair is finished, so the widest activation (200 features x 64 rows = 112 registers) is never
// materialised (decode kept 99 registers in scratch without this).
template <int KB, int NT, int NT2>
__device__ __forceinline__ void blayer_then_last(const bf8 (&in)[KB][kMB], v4 (&y)[NT2][kMB], const bf8 *w, const v4 *bias,
                                                 const bf8 *w2, const v4 *bias2, int g) {
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        const v4 b = bias2[t * 4 + g];
#pragma unroll
        for (int mb = 0; mb < kMB; ++mb) y[t][mb] = b;
    }
#pragma unroll
    for (int p = 0; p < (NT + 1) / 2; ++p) {
        const bool has1 = 2 * p + 1 < NT;
        v4 acc0[kMB], acc1[kMB];
        const v4 b0 = bias[(2 * p) * 4 + g];
        const v4 b1 = has1 ? bias[(2 * p + 1) * 4 + g] : (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mb = 0; mb < kMB; ++mb) { acc0[mb] = b0; acc1[mb] = b1; }
        // fragment reads one step ahead of their MFMAs (hipcc placed every ds_read_b128 right in front of its use: a full LDS
        // round trip before each group of four MFMAs, SQ_WAIT_ANY 45-51 %); sched_barrier pins the order
        bf8 a0 = w[(2 * p) * 64], a1 = has1 ? w[(2 * p + 1) * 64] : a0;
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            bf8 n0 = a0, n1 = a1;
            if (q + 1 < KB) {
                n0 = w[((q + 1) * NT + 2 * p) * 64];
                n1 = has1 ? w[((q + 1) * NT + 2 * p + 1) * 64] : n0;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mb = 0; mb < kMB; ++mb) {
                acc0[mb] = mfma_bf16(a0, in[q][mb], acc0[mb]);
                if (has1) acc1[mb] = mfma_bf16(a1, in[q][mb], acc1[mb]);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = n0;
            a1 = n1;
        }
        bf8 blk[kMB];
#pragma unroll
        for (int mb = 0; mb < kMB; ++mb) {
            lrelu4(acc0[mb]);
            if (has1) lrelu4(acc1[mb]);
            blk[mb] = pack8(acc0[mb], acc1[mb]);
        }
        {
            bf8 a = w2[(p * NT2) * 64];
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                bf8 nx = a;
                if (t + 1 < NT2) nx = w2[(p * NT2 + t + 1) * 64];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int mb = 0; mb < kMB; ++mb) y[t][mb] = mfma_bf16(a, blk[mb], y[t][mb]);
                __builtin_amdgcn_sched_barrier(0);
                a = nx;
            }
        }
    }
}

// rows -> first k block: lane (j, g) reads features 8g .. 8g+7 of its row (64 contiguous bytes in fp64).  Branch-free:
// lanes beyond the last row read row 0 and lane groups / elements beyond D read feature 0 -- always inside the table, and
// never used (those k slots meet zero weights; rows beyond n are not stored and carry no loss).
// In two halves, so that the NEXT pass's rows can be requested while this pass computes: `issue` only loads (raw
// values stay in registers: 8 or 16 per batch tile), `finish` normalises, rounds and packs.  IN64 is a template parameter (a
// run-time dtype branch in front of the loads makes hipcc join the paths with conservative waits).
template <int D, bool IN64> struct RawBlock { typename std::conditional<IN64, double, float>::type v[8]; };
template <int D, bool IN64>
__device__ __forceinline__ void load_block_issue(RawBlock<D, IN64> &raw, const void *x, int64_t row, bool valid, int g) {
    using T = typename std::conditional<IN64, double, float>::type;
    const int64_t base = (valid ? row : 0) * D;
    if (D % 8 == 0) {
        const int f0 = 8 * g < D ? 8 * g : 0;
        if (IN64) {
            const double2 *p = (const double2 *)((const double *)x + base + f0);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double2 t = p[e]; raw.v[2 * e] = (T)t.x; raw.v[2 * e + 1] = (T)t.y; }
        } else {
            const float4 *p = (const float4 *)((const float *)x + base + f0);
            const float4 t0 = p[0], t1 = p[1];
            raw.v[0] = (T)t0.x; raw.v[1] = (T)t0.y; raw.v[2] = (T)t0.z; raw.v[3] = (T)t0.w;
            raw.v[4] = (T)t1.x; raw.v[5] = (T)t1.y; raw.v[6] = (T)t1.z; raw.v[7] = (T)t1.w;
        }
    } else {   // narrow, unaligned rows (the latent codes): element loads; lane groups that hold no feature re-read feature 0
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int f = 8 * g + e < D ? 8 * g + e : 0;
            raw.v[e] = ((const T *)x)[base + f];
        }
    }
}
template <int D, bool IN64>
__device__ __forceinline__ bf8 load_block_finish(const RawBlock<D, IN64> &raw, int g, const double *feats_lds) {
    float v[8];
    const int f0 = 8 * g < D ? 8 * g : 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        double d = (double)raw.v[e];
        const int f = D % 8 == 0 ? f0 + e : (8 * g + e < D ? 8 * g + e : 0);
        if (feats_lds) d = (d - feats_lds[f]) / feats_lds[32 + f];
        v[e] = (D % 8 == 0 || 8 * g + e < D) ? (float)d : 0.f;
    }
    bf8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
    return o;
}

// DEC = false: z = encode(x).  DEC = true: out = decode(z) (+ un-normalise / int truncation), and with xref the
// squared-error partial of out against (normalised) xref rows (forward_loss).
template <int F, int Z, bool DEC, bool IN64>
__global__ void __launch_bounds__(64 * kWaves) bf16_infer_kernel(const uint4 *__restrict__ wfrags, const v4 *__restrict__ bias_g,
                                                                 const void *__restrict__ xin, int64_t n,
                                                                 const double *__restrict__ feats, void *__restrict__ out, int out_f64,
                                                                 const uint8_t *__restrict__ imask, const void *__restrict__ xref,
                                                                 int xref_f64, const double *__restrict__ xref_feats,
                                                                 double *__restrict__ loss_part) {
    using N = BNet<F, Z>;
    constexpr int H = DEC ? 1 : 0;
    constexpr int NFR = N::half_frags(H), NB = N::half_bias(H);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint4 *wl = (uint4 *)lds_raw;
    v4 *bias = (v4 *)(wl + NFR * 64);
    double *fl = (double *)(bias + NB);                 // [0..31] min, [32..63] range applied to rows READ here; [64..127]: to rows written
    __shared__ double red[64 * kWaves];
    for (int i = threadIdx.x; i < NFR * 64; i += 64 * kWaves) wl[i] = wfrags[i];
    for (int i = threadIdx.x; i < NB; i += 64 * kWaves) bias[i] = bias_g[i];
    const double *fsrc = DEC ? xref_feats : feats;     // features applied to rows READ by this kernel
    if (threadIdx.x < 128) {
        const int f = threadIdx.x & 31, which = (threadIdx.x >> 5) & 1;
        const double *src = threadIdx.x < 64 ? fsrc : (DEC ? feats : nullptr);
        fl[threadIdx.x] = (src && f < F) ? src[which * F + f] : (which ? 1.0 : 0.0);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
    const int64_t npass = (n + kRowsPerPass - 1) / kRowsPerPass;
    double lacc = 0.0;
    int lane_off = lane;
    // The rows of the NEXT pass are requested while this pass computes (they used to be loaded at the top of the pass and used at
    // once: SQ_WAIT_ANY 45-51 %); requested BEFORE this pass's stores, so the in-order vmcnt makes the next pass wait for its
    // loads only, not for the stores to drain.
    constexpr int DIN = DEC ? Z : F;
    RawBlock<DIN, IN64> raw[kMB];
    const int64_t pass0 = (int64_t)blockIdx.x * kWaves + wave, pstride = (int64_t)gridDim.x * kWaves;
    auto prefetch = [&](int64_t pass) {
#pragma unroll
        for (int mb = 0; mb < kMB; ++mb) {
            const int64_t r = pass * kRowsPerPass + 16 * mb + j;
            load_block_issue<DIN, IN64>(raw[mb], xin, r, pass < npass && r < n, g);
        }
    };
    if (pass0 < npass) prefetch(pass0);
    for (int64_t pass = pass0; pass < npass; pass += pstride) {
        asm volatile("" : "+v"(lane_off));              // keep the LDS fragment reads inside the loop (LICM would spill the model)
        const bf8 *w = (const bf8 *)wl + lane_off;
        int64_t row[kMB];
        bool valid[kMB];
#pragma unroll
        for (int mb = 0; mb < kMB; ++mb) { row[mb] = pass * kRowsPerPass + 16 * mb + j; valid[mb] = row[mb] < n; }
        if (!DEC) {
            bf8 a0[1][kMB], a2[N::kb(2)][kMB];
#pragma unroll
            for (int mb = 0; mb < kMB; ++mb) a0[0][mb] = load_block_finish<DIN, IN64>(raw[mb], g, feats ? fl : nullptr);
            if (!IN64) prefetch(pass + pstride);          // float64 rows (16 registers per batch tile): after the widest layer pair
            blayer_pair<N::kb(0), N::nt(0), N::nt(1)>(a0, a2, w + N::f_off(0) * 64, bias + N::b_off(0), w + N::f_off(1) * 64,
                                                      bias + N::b_off(1), g);
            if (IN64) prefetch(pass + pstride);
            v4 z[N::nt(3)][kMB];
            blayer_then_last<N::kb(2), N::nt(2), N::nt(3)>(a2, z, w + N::f_off(2) * 64, bias + N::b_off(2), w + N::f_off(3) * 64,
                                                           bias + N::b_off(3), g);
            // one lane mask per batch tile, the output dtype test outside the element loops, a per-element feature test
            // only for registers that are padding on some lane group
#pragma unroll
            for (int mb = 0; mb < kMB; ++mb) {
                if (!valid[mb]) continue;
#pragma unroll
                for (int t = 0; t < N::nt(3); ++t) {
                    const int64_t i0 = row[mb] * Z + 16 * t + 4 * g;
                    if (out_f64) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (16 * t + 12 + r < Z || 16 * t + 4 * g + r < Z) ((double *)out)[i0 + r] = (double)z[t][mb][r];
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (16 * t + 12 + r < Z || 16 * t + 4 * g + r < Z) ((float *)out)[i0 + r] = z[t][mb][r];
                    }
                }
            }
        } else {
            bf8 a4[1][kMB], a5[N::kb(5)][kMB], a6[N::kb(6)][kMB];
#pragma unroll
            for (int mb = 0; mb < kMB; ++mb) a4[0][mb] = load_block_finish<DIN, IN64>(raw[mb], g, nullptr);
            if (!IN64) prefetch(pass + pstride);
            blayer<N::kb(4), N::nt(4), true>(a4, a5, w + N::f_off(4) * 64, bias + N::b_off(4), g);
            blayer<N::kb(5), N::nt(5), true>(a5, a6, w + N::f_off(5) * 64, bias + N::b_off(5), g);
            v4 y[N::nt(7)][kMB];
            blayer_then_last<N::kb(6), N::nt(6), N::nt(7)>(a6, y, w + N::f_off(6) * 64, bias + N::b_off(6), w + N::f_off(7) * 64,
                                                           bias + N::b_off(7), g);
            if (IN64) prefetch(pass + pstride);           // still before this pass's stores
#pragma unroll
            for (int mb = 0; mb < kMB; ++mb)
#pragma unroll
                for (int t = 0; t < N::nt(7); ++t) {
                    const int f0 = 16 * t + 4 * g;                   // this lane's 4 consecutive output features of tile t
                    if (!valid[mb] || f0 >= F) continue;
                    const int64_t i0 = row[mb] * F + f0;
                    double d[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        d[r] = (double)y[t][mb][r];
                        if (feats && f0 + r < F) {   // norm*range + min with two roundings (numpy), trunc for "int" columns (baler.py:420-435)
                            d[r] = __dadd_rn(__dmul_rn(d[r], fl[96 + f0 + r]), fl[64 + f0 + r]);
                            if (imask && imask[f0 + r]) d[r] = trunc(d[r]);
                        }
                    }
                    if (out) {
                        if (F % 4 == 0) {                            // 32 / 16 contiguous, aligned bytes per lane: vector stores
                            if (out_f64) {
                                *(double2 *)((double *)out + i0) = make_double2(d[0], d[1]);
                                *(double2 *)((double *)out + i0 + 2) = make_double2(d[2], d[3]);
                            } else {
                                *(float4 *)((float *)out + i0) = make_float4((float)d[0], (float)d[1], (float)d[2], (float)d[3]);
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (f0 + r < F) {
                                    if (out_f64) ((double *)out)[i0 + r] = d[r]; else ((float *)out)[i0 + r] = (float)d[r];
                                }
                        }
                    }
                    if (xref) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (f0 + r < F) {
                                double xv = xref_f64 ? ((const double *)xref)[i0 + r] : (double)((const float *)xref)[i0 + r];
                                if (xref_feats) xv = (xv - fl[f0 + r]) / fl[32 + f0 + r];
                                const double e = (double)y[t][mb][r] - (double)(float)xv;
                                lacc += e * e;
                            }
                    }
                }
        }
    }
    if (DEC && loss_part) {
        red[threadIdx.x] = lacc;
        __syncthreads();
        for (int st = 32 * kWaves; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) loss_part[blockIdx.x] = red[0];
    }
}

// params (fp32, state-dict order) -> bf16 fragments / fp32 bias fragments through precomputed index maps
__global__ void __launch_bounds__(256) pack_bf16_k(const float *__restrict__ params, const int *__restrict__ src, int count,
                                                   __bf16 *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = (__bf16)(src[i] >= 0 ? params[src[i]] : 0.f);
}
__global__ void __launch_bounds__(256) pack_bias_k(const float *__restrict__ params, const int *__restrict__ src, int count,
                                                   float *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) dst[i] = src[i] >= 0 ? params[src[i]] : 0.f;
}
__global__ void __launch_bounds__(256) sum_loss_bf16_k(const double *__restrict__ part, int n, double scale, double *__restrict__ out) {
    __shared__ double sh[256];
    const double s = block_sum_fixed(part, n, sh);
    if (threadIdx.x == 0) *out = s * scale;
}

struct Bf16Ops;
struct Bf16State {
    const Bf16Ops *ops = nullptr;
    DevBuf wsrc[2], bsrc[2];     // index maps (encoder half, decoder half)
    DevBuf w[2], b[2];           // packed bf16 fragments / fp32 bias fragments
    DevBuf zscratch;             // latent codes between the two launches of forward_loss
    int wcount[2] = {0, 0}, bcount[2] = {0, 0};
    int grid = 256;
};
struct Bf16Ops {
    int (*setup)(bamd_handle *, Bf16State *);
    int (*run)(bamd_handle *, Bf16State *, bool dec, const void *xin, int in_f64, int64_t n, const double *feats, void *out,
               int out_f64, const uint8_t *imask, const void *xref, int xref_f64, const double *xref_feats, double *loss_part,
               hipStream_t s);
    int z_dim, n_features;
};
Bf16State *bstate(bamd_handle *h) { return (Bf16State *)h->bf16_state; }

template <int F, int Z> struct BImpl {
    using N = BNet<F, Z>;
    static bool matches(const bamd_handle *h) {
        if (h->L != 8) return false;
        for (int i = 0; i <= 8; ++i)
            if (h->dims[i] != N::dim(i)) return false;
        return true;
    }
    static int setup(bamd_handle *h, Bf16State *st) {
        for (int hf = 0; hf < 2; ++hf) {
            std::vector<int> wsrc((size_t)N::half_frags(hf) * 512, -1), bsrc((size_t)N::half_bias(hf) * 4, -1);
            for (int l = 4 * hf; l < 4 * hf + 4; ++l) {
                const int K = N::dim(l), NN = N::dim(l + 1);
                for (int q = 0; q < N::kb(l); ++q)
                    for (int t = 0; t < N::nt(l); ++t)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int i = lane & 15, g = lane >> 4;
                                const int nf = 16 * t + i, kf = N::in_feature(l, q, g, e);
                                if (nf < NN && kf >= 0)
                                    wsrc[((size_t)(N::f_off(l) + q * N::nt(l) + t) * 64 + lane) * 8 + e] = N::w_off_c(l) + nf * K + kf;
                            }
                for (int t = 0; t < N::nt(l); ++t)
                    for (int g = 0; g < 4; ++g)
                        for (int r = 0; r < 4; ++r) {
                            const int nf = 16 * t + 4 * g + r;
                            if (nf < NN) bsrc[((size_t)N::b_off(l) + t * 4 + g) * 4 + r] = N::b_off_c(l) + nf;
                        }
            }
            st->wcount[hf] = (int)wsrc.size();
            st->bcount[hf] = (int)bsrc.size();
            int rc = st->wsrc[hf].ensure(wsrc.size() * sizeof(int));
            if (rc) return rc;
            rc = st->bsrc[hf].ensure(bsrc.size() * sizeof(int));
            if (rc) return rc;
            rc = st->w[hf].ensure(wsrc.size() * sizeof(__bf16));
            if (rc) return rc;
            rc = st->b[hf].ensure(bsrc.size() * sizeof(float));
            if (rc) return rc;
            BAMD_HIP(hipMemcpy(st->wsrc[hf].p, wsrc.data(), wsrc.size() * sizeof(int), hipMemcpyHostToDevice));
            BAMD_HIP(hipMemcpy(st->bsrc[hf].p, bsrc.data(), bsrc.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        BAMD_HIP(hipFuncSetAttribute((const void *)bf16_infer_kernel<F, Z, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)N::lds_bytes(0)));
        BAMD_HIP(hipFuncSetAttribute((const void *)bf16_infer_kernel<F, Z, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)N::lds_bytes(0)));
        BAMD_HIP(hipFuncSetAttribute((const void *)bf16_infer_kernel<F, Z, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)N::lds_bytes(1)));
        BAMD_HIP(hipFuncSetAttribute((const void *)bf16_infer_kernel<F, Z, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)N::lds_bytes(1)));
        return BAMD_OK;
    }
    static int run(bamd_handle *h, Bf16State *st, bool dec, const void *xin, int in_f64, int64_t n, const double *feats, void *out,
                   int out_f64, const uint8_t *imask, const void *xref, int xref_f64, const double *xref_feats, double *loss_part,
                   hipStream_t s) {
        const int64_t npass = (n + kRowsPerPass - 1) / kRowsPerPass;
        int64_t wg = (npass + kWaves - 1) / kWaves;
        const int grid = (int)(wg < 1 ? 1 : (wg > st->grid ? st->grid : wg));
        auto go = [&](auto decv, auto inv) {
            constexpr bool D_ = decltype(decv)::value, I_ = decltype(inv)::value;
            hipLaunchKernelGGL((bf16_infer_kernel<F, Z, D_, I_>), dim3(grid), dim3(64 * kWaves), N::lds_bytes(D_ ? 1 : 0), s,
                               (const uint4 *)st->w[D_ ? 1 : 0].p, (const v4 *)st->b[D_ ? 1 : 0].p, xin, n, feats, out, out_f64, imask, xref,
                               xref_f64, xref_feats, loss_part);
        };
        if (dec && in_f64) go(std::true_type(), std::true_type());
        else if (dec) go(std::true_type(), std::false_type());
        else if (in_f64) go(std::false_type(), std::true_type());
        else go(std::false_type(), std::false_type());
        BAMD_HIP(hipGetLastError());
        return grid;
    }
    static const Bf16Ops *ops() {
        static const Bf16Ops o = {setup, run, Z, F};
        return &o;
    }
};

const Bf16Ops *find_bf16(const bamd_handle *h) {
    if (BImpl<24, 15>::matches(h)) return BImpl<24, 15>::ops();
    if (BImpl<24, 12>::matches(h)) return BImpl<24, 12>::ops();
    if (BImpl<24, 8>::matches(h)) return BImpl<24, 8>::ops();
    if (BImpl<24, 6>::matches(h)) return BImpl<24, 6>::ops();
    // the other latent sizes of the compression-ratio knob (see fused.hip find_ops)
    if (BImpl<24, 10>::matches(h)) return BImpl<24, 10>::ops();
    if (BImpl<24, 5>::matches(h)) return BImpl<24, 5>::ops();
    if (BImpl<24, 4>::matches(h)) return BImpl<24, 4>::ops();
    if (BImpl<24, 3>::matches(h)) return BImpl<24, 3>::ops();
    if (BImpl<24, 2>::matches(h)) return BImpl<24, 2>::ops();
    return nullptr;
}

}  // namespace

bool bf16_has_kernels(const bamd_handle *h) { return find_bf16(h) != nullptr; }

int bf16_setup(bamd_handle *h) {
    const Bf16Ops *ops = find_bf16(h);
    if (!ops) {
        set_error("BAMD_MODE_BF16 is instantiated for the 24-column AE (latent 15/12/8/6) and the wide models (2500-25, 512-6) only; use BAMD_MODE_F32");
        return BAMD_ERR_UNSUPPORTED;
    }
    Bf16State *st = new Bf16State();
    st->ops = ops;
    h->bf16_state = st;
    return ops->setup(h, st);
}

void bf16_teardown(bamd_handle *h) {
    Bf16State *st = bstate(h);
    if (!st) return;
    for (int i = 0; i < 2; ++i) { st->wsrc[i].release(); st->bsrc[i].release(); st->w[i].release(); st->b[i].release(); }
    st->zscratch.release();
    delete st;
    h->bf16_state = nullptr;
}

int bf16_pack(bamd_handle *h, hipStream_t s) {
    Bf16State *st = bstate(h);
    for (int hf = 0; hf < 2; ++hf) {
        hipLaunchKernelGGL(pack_bf16_k, dim3((st->wcount[hf] + 255) / 256), dim3(256), 0, s, (const float *)h->params.p,
                           (const int *)st->wsrc[hf].p, st->wcount[hf], (__bf16 *)st->w[hf].p);
        hipLaunchKernelGGL(pack_bias_k, dim3((st->bcount[hf] + 255) / 256), dim3(256), 0, s, (const float *)h->params.p,
                           (const int *)st->bsrc[hf].p, st->bcount[hf], (float *)st->b[hf].p);
    }
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

int bf16_encode(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *z, int z_dtype, hipStream_t s) {
    Bf16State *st = bstate(h);
    int rc = st->ops->run(h, st, false, x, x_dtype == BAMD_F64, n, features, z, z_dtype == BAMD_F64, nullptr, nullptr, 0, nullptr,
                          nullptr, s);
    return rc < 0 ? rc : BAMD_OK;
}

int bf16_decode(bamd_handle *h, const void *z, int z_dtype, int64_t n, const double *features, const uint8_t *int_mask, void *out,
                int out_dtype, hipStream_t s) {
    Bf16State *st = bstate(h);
    int rc = st->ops->run(h, st, true, z, z_dtype == BAMD_F64, n, features, out, out_dtype == BAMD_F64, int_mask, nullptr, 0, nullptr,
                          nullptr, s);
    return rc < 0 ? rc : BAMD_OK;
}

int bf16_forward_loss(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *recon, int recon_dtype,
                      double *loss_sum, hipStream_t s) {
    Bf16State *st = bstate(h);
    int rc = st->zscratch.ensure((size_t)n * st->ops->z_dim * sizeof(float));
    if (rc) return rc;
    rc = h->lossp.ensure(sizeof(double) * 1024);
    if (rc) return rc;
    rc = st->ops->run(h, st, false, x, x_dtype == BAMD_F64, n, features, st->zscratch.p, 0, nullptr, nullptr, 0, nullptr, nullptr, s);
    if (rc < 0) return rc;
    const int grid = st->ops->run(h, st, true, st->zscratch.p, 0, n, nullptr, recon, recon_dtype == BAMD_F64, nullptr, x,
                                  x_dtype == BAMD_F64, features, (double *)h->lossp.p, s);
    if (grid < 0) return grid;
    hipLaunchKernelGGL(sum_loss_bf16_k, dim3(1), dim3(256), 0, s, (const double *)h->lossp.p, grid, 1.0 / st->ops->n_features, loss_sum);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

}  // namespace bamd
