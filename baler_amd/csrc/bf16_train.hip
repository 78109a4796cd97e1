// bf16 MFMA training of AE(F, Z) (BAMD_MODE_BF16): forward + sum-of-squares loss + backward on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation; the caller keeps fp32 master weights and fp32 Adam state
// (bamd_adam_step), the library re-rounds its bf16 fragment copies after every step.
//
// Why this is not the fp32 kernels with another MFMA: a bf16 MFMA is 16 cycles and eats a 1-KiB weight fragment.
// The fp32 path gives every wave its own 16 rows and lets it stream ALL weights (one fragment per 128 MFMA
// cycles); at bf16 rates that is 64 B/clk per wave, 4x what a CU's L1 path delivers.  So here the WORKGROUP, not
// the wave, owns a batch tile (64 rows) and the waves split every layer's OUTPUT tiles: a weight fragment is
// loaded by one wave and feeds four MFMAs (the four 16-row tiles), i.e. the weights cross the L1 once per 64 rows.
// Activations therefore live in LDS, not in registers:
//   * image i = [64 batch rows][feature slots] in bf16, one per layer input X_i (2-bit XOR swizzle of the 16-byte
//     chunks, row strides of 64 x odd bytes: every read shape below is bank-conflict free, tools/probe/lds_conflicts.py);
//   * a layer's B operand (32 input features of one row per lane) is ONE ds_read_b128 of the row-major image;
//   * the epilogue (LeakyReLU; v_cvt_pk_bf16_f32) writes the C tile back with one ds_write_b64 per lane and tile;
//   * the weight-gradient product [dW | db] = dZ^T [X | 1] contracts over the BATCH index, i.e. needs both images
//     transposed: gfx950's ds_read_b64_tr_b16 delivers exactly that (4 rows x 16 columns per 16 lanes), so the same
//     row-major images serve the chain (row reads) and the weight gradients (transposed reads) -- no second copy,
//     no transposing stores.  The first padding slot of every image is a ONES column: it carries db through the
//     weight-gradient product, and the forward product reads the bias through it (the packed weights hold b in input
//     column K, and a 1 at [padding output K'][column K] so that every layer regenerates the next layer's ones column);
//   * dZ_{l-1} is written into the image of the MIRROR layer (same shape, dead at that time: see TNet::zoff), so the epilogue of
//     layer l needs no barrier against the waves that still read X_l for its weight-gradient tiles: one barrier per backward
//     layer, 132 KB of images for the whole network at 64 rows, no second set of buffers.
// [dW | db] tiles stay in MFMA accumulators for the whole persistent loop (as in fused.hip) and are reduced over
// workgroups in a fixed order: bitwise reproducible.  All 298 tiles x 256 floats do not fit one CU's registers
// next to the chain, so training is two launches over the same rows.  The cut sits BELOW layer 1: PART 0 = forward, loss,
// the whole input-gradient chain down to dZ_1 and the weight gradients of layers 7..2 (181 tiles); PART 1 = layer 0's
// forward recomputed (one k block), dZ_1 handed over as 224 B per row, layers 1..0 (117 tiles).  (Round 2 cut at the
// bottleneck -- 32 B per row -- and PART 1 recomputed the forward of layers 0..2: a third of its time.)
// The four NARROW layers in the middle (100 -> 50 -> Z -> 50 -> 100, forward and backward) run as a per-wave REGISTER chain
// ("M-split": every wave computes ALL output tiles of ITS 16 rows, a C tile pair is the next layer's B operand after
// v_cvt_pk_bf16_f32, the packed weights carry the k permutation): no barrier between them -- as N-split phases each of them
// cost 500-1300 cycles for 32-256 cycles of MFMAs (barrier, LDS round trip, lock step of the four waves).
#include "bf16.hpp"

#include <cmath>
#include <cstdlib>
#include <utility>

namespace bamd {
namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
using v4 = float __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

constexpr int kRows = 64;      // batch rows per workgroup iteration (four 16-row MFMA tiles)

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

template <int F, int Z> struct TNet {
    static constexpr int L = 8;
    __host__ __device__ static constexpr int dim(int i) {
        return i == 0 ? F : i == 1 ? 200 : i == 2 ? 100 : i == 3 ? 50 : i == 4 ? Z : i == 5 ? 50 : i == 6 ? 100 : i == 7 ? 200 : F;
    }
    __host__ __device__ static constexpr bool act(int l) { return !(l == 3 || l == 7); }
    // forward product of layer l: k blocks of 32 input features x tiles of 16 output features
    __host__ __device__ static constexpr int kb(int l) { return cdiv(dim(l), 32); }
    __host__ __device__ static constexpr int nt(int l) { return cdiv(dim(l + 1), 16); }
    // input-gradient product of layer l (dX_l = W_l^T dZ_l, l >= 1): k blocks over OUTPUT features, tiles over inputs
    __host__ __device__ static constexpr int kbb(int l) { return cdiv(dim(l + 1), 32); }
    __host__ __device__ static constexpr int ntb(int l) { return cdiv(dim(l), 16); }
    // weight-gradient tiles of layer l: nt(l) x kt(l)  (the extra input slot carries db)
    __host__ __device__ static constexpr int kt(int l) { return cdiv(dim(l) + 1, 16); }
    // fragment stream (1-KiB units): forward layers 0..7 as [q][t], then backward layers 7..1 as [q][t]
    __host__ __device__ static constexpr int ffo(int l) { int s = 0; for (int j = 0; j < l; ++j) s += kb(j) * nt(j); return s; }
    __host__ __device__ static constexpr int bfo(int l) { int s = ffo(L); for (int j = L - 1; j > l; --j) s += kbb(j) * ntb(j); return s; }
    __host__ __device__ static constexpr int nfrag() { return bfo(1) + kbb(1) * ntb(1); }
    // LDS images: i = 0..7 holds X_i (the input of layer i) and later dZ_{i-1}; i = 8 holds dZ_7
    __host__ __device__ static constexpr int iblocks(int i) { return i < L ? kb(i) : cdiv(F, 32); }
    __host__ __device__ static constexpr int istride(int i) { int b = iblocks(i); return 64 * (b % 2 ? b : b + 1); }   // bytes, 64 x odd
    __host__ __device__ static constexpr int ioff(int i) { int s = 0; for (int j = 0; j < i; ++j) s += kRows * istride(j); return s; }
    __host__ __device__ static constexpr int img_bytes() { return ioff(L + 1); }
    // Where dZ_l lives -- OUT OF PLACE, in an image region that is dead by then, so that the epilogue of layer l + 1 may run while
    // other waves still read X_{l+1} for that layer's weight-gradient tiles: ONE barrier per backward layer, no second set of
    // buffers.  A region is reused with the stride of the image whose SHAPE dZ_l has (X_{l+1}'s), which is never larger.
    //   PART 0:  dZ_7 -> image 8;  dZ_6 -> region 1 (X_1 is dead after layer 1's forward: its weight gradient is PART 1's);
    //            dZ_5 -> region 7 (X_7 dead after layer 7);  dZ_4 -> region 6 (X_6 dead after layer 6);  dZ_3 -> region 0;
    //            dZ_2 -> region 1 (dZ_6 dead after layer 6);  dZ_1 -> global memory (the hand-off)
    //   PART 1:  dZ_1 -> region 2 (loaded from the hand-off);  dZ_0 -> region 7
    __host__ __device__ static constexpr int zimg_of(int l) { return l == 7 ? 8 : l == 6 ? 1 : l == 5 ? 7 : l == 4 ? 6 : l == 3 ? 0 : l == 2 ? 1 : l == 1 ? 2 : 7; }
    __host__ __device__ static constexpr int zoff(int l) { return ioff(zimg_of(l)); }
    // M-split (per-wave register chain) products: forward layers 2..5, input-gradient products of layers 5..2.  The first of
    // each run reads its B operand from the image (natural k order); the others take it from the previous product's packed
    // C tiles: k slot (g, e) <-> feature 32 q + 16 (e >> 2) + 4 g + (e & 3) (TImpl::setup packs the fragments that way).
    __host__ __device__ static constexpr bool mf(int l) { return l >= 2 && l <= 5; }
    __host__ __device__ static constexpr bool mb(int l) { return l >= 2 && l <= 5; }
    __host__ __device__ static constexpr bool regfed_f(int l) { return l >= 3 && l <= 5; }
    __host__ __device__ static constexpr bool regfed_b(int l) { return l >= 2 && l <= 4; }
    static constexpr int hand_tiles = 7;                 // dZ_1: cdiv(100, 16) tiles = 224 B per row
    // weight-gradient tiles in the partial-gradient buffer
    __host__ __device__ static constexpr int dwt(int l) { return nt(l) * kt(l); }
    __host__ __device__ static constexpr int slab_off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += dwt(j); return s; }
    // ownership of layer l's tiles: by output tile (nt = wave + 4 i, every kt) or by input tile (kt = wave + 4 i)
    __host__ __device__ static constexpr bool by_nt(int l) { return cdiv(nt(l), 4) * kt(l) <= cdiv(kt(l), 4) * nt(l); }
    __host__ __device__ static constexpr int dwn(int l) { return by_nt(l) ? cdiv(nt(l), 4) * kt(l) : cdiv(kt(l), 4) * nt(l); }
    // canonical (state-dict) offsets
    __host__ __device__ static constexpr int w_off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += dim(j + 1) * dim(j) + dim(j + 1); return s; }
    __host__ __device__ static constexpr int b_off(int l) { return w_off(l) + dim(l + 1) * dim(l); }
    __host__ __device__ static constexpr int nparams() { return w_off(L); }
    static_assert(F % 8 == 0 && F < 32 && Z < 16, "input rows are read as 8-feature chunks of one 32-slot block; the latent is one tile");
};

// Which layers a launch covers.  PART 0: forward 0..7, loss, backward 7..2 (input-gradient chain down to dZ_1).
// PART 1: forward 0, backward 1..0.
// dZ_1 only is handed over (row-major, 224 B per row); PART 1 reads the rows again and recomputes layer 0.  Handing X_0 and X_1 over as
// well (704 B per row, tile-major) was measured and rejected in round 3: identical results, 0.811 vs 0.774 ms per 1M rows -- PART 1 is
// bound by the record it streams one iteration ahead, not by the layer-0 work it would drop.
// PART 1 requests the NEXT iteration's rows and hand-off records right after the last fragment wait of its input-gradient product
// (layer 1), not at the top of the iteration.  Loads of a wave retire in order: requested at the top, these HBM fetches (~2 us)
// stood in front of every fragment requested after them, and the product's third k block -- 3,000 cycles later -- waited for them.
// From the late place the next wait on a younger load is a whole epilogue + two weight-gradient phases + the rows phase away.
template <int PART> struct Part {
    static constexpr int fwd_end = PART == 1 ? 1 : 8;        // forward layers [0, fwd_end)
    static constexpr int bwd_hi = PART == 1 ? 1 : 7;         // backward layers bwd_hi .. bwd_lo
    static constexpr int bwd_lo = PART == 0 ? 2 : 0;
    __host__ __device__ static constexpr bool has(int l) { return l <= bwd_hi && l >= bwd_lo; }
};

// ---- the weight-fragment schedule of one iteration ------------------------------------------------------------
// A STEP is one k block of one chain product; every wave loads at most 4 fragments per step, two steps ahead of the
// MFMAs, into a ring of 3 step buffers.  The step count is padded to a multiple of 3 so that the ring wraps across
// persistent iterations (padding steps load nothing).
constexpr int kRD = 3;      // step buffers of the fragment ring (fragments run kRD - 1 steps ahead; 2 / 3 / 5 measured: 3)
struct StepInfo { int bwd, l, q, valid, msplit; };
// steps of a product: N-split = one per k block (<= 4 fragments per wave); M-split = its kb x nt fragments, in MFMA order
// [k block][tile], four per step (every wave loads all of them)
template <class N, int PART> struct Sched {
    using P = Part<PART>;
    __host__ __device__ static constexpr int chain_lo() { return P::bwd_lo < 1 ? 1 : P::bwd_lo; }   // layer 0 has no input gradient
    __host__ __device__ static constexpr int nf(int l) { return N::mf(l) ? cdiv(N::kb(l) * N::nt(l), 4) : N::kb(l); }
    __host__ __device__ static constexpr int nb(int l) { return N::mb(l) ? cdiv(N::kbb(l) * N::ntb(l), 4) : N::kbb(l); }
    __host__ __device__ static constexpr int fstep(int l) { int s = 0; for (int j = 0; j < l; ++j) s += nf(j); return s; }
    __host__ __device__ static constexpr int bstep(int l) { int s = fstep(P::fwd_end); for (int j = P::bwd_hi; j > l; --j) s += nb(j); return s; }
    static constexpr int real = bstep(chain_lo()) + nb(chain_lo());
    static constexpr int total = cdiv(real, kRD) * kRD;
    __host__ __device__ static constexpr StepInfo info(int s) {
        s %= total;
        if (s >= real) return {0, 0, 0, 0, 0};
        for (int l = 0; l < P::fwd_end; ++l)
            if (s < fstep(l) + nf(l)) return {0, l, s - fstep(l), 1, N::mf(l) ? 1 : 0};
        for (int l = P::bwd_hi; l >= chain_lo(); --l)
            if (s < bstep(l) + nb(l)) return {1, l, s - bstep(l), 1, N::mb(l) ? 1 : 0};
        return {0, 0, 0, 0, 0};
    }
};

// How the NT output tiles of a chain product are split over the 4 waves.  N-split slot i: tile wave + 4 i for ALL four
// row tiles (the fragment is loaded by one wave only); M-split tile k: every wave computes it for ITS row tile (fragment
// loaded by all four).  NT = 13 -> 3 N-split + tile 12 M-split; 7 -> 2 N-split slots (wave 3's second is empty);
// 4 -> 1; 2 and 1 -> M-split.
template <int NT> struct Split {
    static constexpr int R = NT % 4;
    static constexpr int NS = NT < 4 ? 0 : (R == 1 ? NT / 4 : cdiv(NT, 4));
    static constexpr int MS = NT < 4 ? NT : (R == 1 ? 1 : 0);
    static constexpr bool ragged = NT >= 4 && R != 1 && R != 0;   // the last N-split slot does not exist on every wave
    static constexpr int NF = NS + MS;                        // fragments per step and wave
    static constexpr int m0 = 4 * NS;                         // first M-split tile
    static_assert(NF <= 4, "ring step buffers hold 4 fragments");
};

struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;   // lane * 16
};
__device__ __forceinline__ bf8 frag_rt(const WStream &ws, int idx) {
    return __builtin_bit_cast(bf8, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, idx * 1024, 0));
}
struct Ring { bf8 buf[kRD][4]; };

template <class N, int PART, int STEP>
__device__ __forceinline__ void issue(Ring &ring, const WStream &ws, int wave) {
    constexpr StepInfo si = Sched<N, PART>::info(STEP);
    if constexpr (si.valid && si.msplit) {
        constexpr int cnt = si.bwd ? N::kbb(si.l) * N::ntb(si.l) : N::kb(si.l) * N::nt(si.l);
        constexpr int base = (si.bwd ? N::bfo(si.l) : N::ffo(si.l)) + 4 * si.q;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * si.q + k < cnt) ring.buf[STEP % kRD][k] = frag_rt(ws, base + k);
    } else if constexpr (si.valid) {
        constexpr int NT = si.bwd ? N::ntb(si.l) : N::nt(si.l);
        using SP = Split<NT>;
        constexpr int base = (si.bwd ? N::bfo(si.l) : N::ffo(si.l)) + si.q * NT;
#pragma unroll
        for (int k = 0; k < SP::NF; ++k) {
            int t = k < SP::NS ? wave + 4 * k : SP::m0 + (k - SP::NS);
            if (SP::ragged && k == SP::NS - 1) t = t < NT ? t : NT - 1;      // empty slot: load a valid fragment, unused
            ring.buf[STEP % kRD][k] = frag_rt(ws, base + t);
        }
    }
}

__device__ __forceinline__ v4 mfma(bf8 a, bf8 b, v4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// ---- LDS addressing ---------------------------------------------------------------------------------------------
// byte offset of 16-byte chunk c of row r in an image of row stride S: r S + ((c ^ sigma(r)) << 4), sigma(r) = (r >> 1) & 3.
// Every access below is "lane part + compile-time part"; the lane parts for one stride are computed once:
//   row    : B operand / input rows: lane (j, g) -> chunk g (+ 4 q) of row j (+ 16 m)
//   wr[par]: C tile t of parity par: lane (j, g) -> its 8 bytes (features 16 t + 4 g ..) of row j (+ 16 m)
//   tr[par]: transposed read of tile t of parity par: lane 4 q' + p of group g -> row 4 g + q' (+ 16 h + 32 kh), 8 bytes p
typedef unsigned char __attribute__((address_space(3))) *lds_p;
struct Lay {
    int row, wr0, wr1, tr0, tr1;
    // parity select as a conditional move (indexing a member array with a run-time value would put the struct in scratch)
    __device__ __forceinline__ int wr(int par) const { return par ? wr1 : wr0; }
    __device__ __forceinline__ int tr(int par) const { return par ? tr1 : tr0; }
};
template <int S> __device__ __forceinline__ Lay make_lay(int lane) {
    const int j = lane & 15, g = lane >> 4;
    const int sg = (j >> 1) & 3;
    Lay a;
    a.row = j * S + ((g ^ sg) << 4);
    a.wr0 = j * S + ((((sg & 2)) | ((g >> 1) ^ (sg & 1))) << 4) + 8 * (g & 1);
    a.wr1 = j * S + ((((2 ^ (sg & 2))) | ((g >> 1) ^ (sg & 1))) << 4) + 8 * (g & 1);
    const int rr = 4 * g + ((lane & 15) >> 2), p = lane & 3, st = (rr >> 1) & 3;
    a.tr0 = rr * S + ((((st & 2)) | ((p >> 1) ^ (st & 1))) << 4) + 8 * (p & 1);
    a.tr1 = rr * S + ((((2 ^ (st & 2))) | ((p >> 1) ^ (st & 1))) << 4) + 8 * (p & 1);
    return a;
}
// the four strides that occur (64 x {1, 3, 5, 7} bytes)
struct Lays { Lay s1, s3, s5, s7; };
template <int S> __device__ __forceinline__ const Lay &lay_of(const Lays &ls) {
    static_assert(S == 64 || S == 192 || S == 320 || S == 448, "image stride");
    if constexpr (S == 64) return ls.s1;
    else if constexpr (S == 192) return ls.s3;
    else if constexpr (S == 320) return ls.s5;
    else return ls.s7;
}

__device__ __forceinline__ bf8 lds_b128(lds_p p) { return *(const bf8 __attribute__((address_space(3))) *)p; }
__device__ __forceinline__ u2 lds_b64(lds_p p) { return *(const u2 __attribute__((address_space(3))) *)p; }
__device__ __forceinline__ void lds_w64(lds_p p, u2 v) { *(u2 __attribute__((address_space(3))) *)p = v; }
__device__ __forceinline__ s4 lds_tr(lds_p p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)p);
}
// operand of a weight-gradient MFMA: 8 batch rows per lane (k slot (g, 4 h + q) <-> row 16 h + 4 g + q of the 32-row half)
template <int S> __device__ __forceinline__ bf8 tr_operand(lds_p base, int kh) {
    const s4 lo = lds_tr(base + (32 * kh) * S), hi = lds_tr(base + (32 * kh + 16) * S);
    typedef short s8 __attribute__((ext_vector_type(8)));
    const s8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf8, v);
}

__device__ __forceinline__ void lrelu4(v4 &a) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f k2 = (v2f){0.01f, 0.01f};
    asm("" : "+v"(k2));                       // register pair, vector product: v_pk_mul_f32 (see fused.hip lrelu)
    v4 m = a * (v4){k2[0], k2[1], k2[0], k2[1]};
    asm("" : "+v"(m));
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = __builtin_elementwise_maximum(a[r], m[r]);
}
// two v_cvt_pk_bf16_f32 (built pair by pair: a 4-element bf16 vector makes hipcc convert elements 2, 3 one by one + v_perm)
__device__ __forceinline__ u2 pack4(const v4 &a) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 lo = {(__bf16)a[0], (__bf16)a[1]}, hi = {(__bf16)a[2], (__bf16)a[3]};
    return (u2){__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
}
// dZ = bf16(d * lrelu'(pre)), sign(pre) == sign(post); `y` = the 4 post-activation bf16 values of the same elements.
// Both candidates are rounded (d and 0.01 d, one v_pk_mul per pair) and the halves are picked by a sign mask of y
// (v_pk_ashrrev_i16) with one v_bfi_b32 per pair: 2.5 VALU instructions per value including the conversion
// (compare + select on the fp32 values: 3; mask arithmetic on the slope bits: 5.5).
__device__ __forceinline__ u2 lrelu_bwd_pack4(const v4 &d, u2 y) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f k2 = (v2f){0.01f, 0.01f};
    asm("" : "+v"(k2));
    v4 m = d * (v4){k2[0], k2[1], k2[0], k2[1]};
    asm("" : "+v"(m));
    const u2 p1 = pack4(d), p2 = pack4(m);
    unsigned sh = 0x000F000Fu;      // shift count per half (an inline constant would reach the low half only)
    u2 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // 0xFFFF where the activation is negative.  Through asm: written as a shift of a 2 x i16 vector, hipcc (ROCm 7.2)
        // uses the mask of y[0] for BOTH dwords (it drops the load of y[1]); the operands are plain VALU / LDS-load results
        unsigned mask;
        asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(mask) : "v"(sh), "v"(y[h]));
        o[h] = (p2[h] & mask) | (p1[h] & ~mask);                              // one v_bitop3_b32
    }
    return o;
}

// ---- one chain product: NT output tiles over the 4 waves, KB k blocks, B operand from image IN ---------------------
// acc tiles: an[i][m] = tile wave + 4 i, row tile m;  am[k] = tile m0 + k, row tile `wave`
template <int NT> struct ChainAcc {
    using SP = Split<NT>;
    v4 an[SP::NS > 0 ? SP::NS : 1][4];
    v4 am[SP::MS > 0 ? SP::MS : 1];
};

constexpr int kBD = 1;      // k blocks the B operand reads run ahead of the MFMAs (LDS latency under load: 150-200 cycles)
// B operands of one k block: [0..3] the four row tiles (N-split slots), [4] this wave's own row tile (M-split tiles)
template <int NT, int SIN>
__device__ __forceinline__ void chain_load_b(bf8 (&dst)[5], lds_p in_row, int wave, int q) {
    using SP = Split<NT>;
    if (SP::NS > 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m) dst[m] = lds_b128(in_row + 16 * m * SIN + 64 * q);
    }
    // M-split tiles use this wave's OWN row tile: read once more (a wave-uniform address) rather than selected from the four
    // with 12 v_cndmask -- and read AHEAD like the others (read at its use, every k block paid one LDS round trip)
    if (SP::MS > 0) dst[4] = lds_b128(in_row + 16 * wave * SIN + 64 * q);
}
// k block Q of the product (every index a template constant: the ring and the B buffers stay in registers)
template <class N, int PART, int STEP0, int KB, int NT, int SIN, int Q>
__device__ __forceinline__ void chain_step(ChainAcc<NT> &acc, bf8 (&b)[kBD + 1][5], lds_p in_row, Ring &ring, const WStream &ws, int wave,
                                           bool last_ok) {
    using SP = Split<NT>;
    issue<N, PART, STEP0 + Q + kRD - 1>(ring, ws, wave);
    if (Q + kBD < KB) chain_load_b<NT, SIN>(b[(Q + kBD) % (kBD + 1)], in_row, wave, Q + kBD);
    const bf8 (&bq)[5] = b[Q % (kBD + 1)];
    // k block 0 starts from a literal zero C operand (no accumulator initialisation; the bias arrives through the
    // ones slot of the input image, whose weight column holds it)
    const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < SP::NS; ++k) {
        if (SP::ragged && k == SP::NS - 1 && !last_ok) continue;
#pragma unroll
        for (int m = 0; m < 4; ++m) acc.an[k][m] = mfma(ring.buf[(STEP0 + Q) % kRD][k], bq[m], Q == 0 ? zero : acc.an[k][m]);
    }
#pragma unroll
    for (int k = 0; k < SP::MS; ++k) acc.am[k] = mfma(ring.buf[(STEP0 + Q) % kRD][SP::NS + k], bq[4], Q == 0 ? zero : acc.am[k]);
    __builtin_amdgcn_sched_barrier(0);
}
template <class N, int PART, int STEP0, int KB, int NT, int Q>
__device__ __forceinline__ void monly_step(ChainAcc<NT> &acc, const bf8 (&b)[KB], Ring &ring, const WStream &ws, int wave) {
    using SP = Split<NT>;
    issue<N, PART, STEP0 + Q + kRD - 1>(ring, ws, wave);
    const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < SP::MS; ++k) acc.am[k] = mfma(ring.buf[(STEP0 + Q) % kRD][k], b[Q], Q == 0 ? zero : acc.am[k]);
    __builtin_amdgcn_sched_barrier(0);
}
template <class N, int PART, int STEP0, int KB, int NT, int SIN, int... Q>
__device__ __forceinline__ void chain_mm_impl(ChainAcc<NT> &acc, lds_p in_row, Ring &ring, const WStream &ws, int wave,
                                              std::integer_sequence<int, Q...>) {
    using SP = Split<NT>;
    const bool last_ok = !SP::ragged || wave + 4 * (SP::NS - 1) < NT;      // wave-uniform
    if constexpr (SP::NS == 0) {
        // M-split only (layer 7: two tiles, 7 k blocks of 2 MFMAs): one k block ahead = 32 MFMA cycles, every step waited for
        // its LDS round trip -- read all k blocks of this wave's row tile first (KB x 4 registers)
        bf8 ball[KB];
#pragma unroll
        for (int q = 0; q < KB; ++q) ball[q] = lds_b128(in_row + 16 * wave * SIN + 64 * q);
        (monly_step<N, PART, STEP0, KB, NT, Q>(acc, ball, ring, ws, wave), ...);
        return;
    }
    bf8 b[kBD + 1][5];
#pragma unroll
    for (int q = 0; q < kBD && q < KB; ++q) chain_load_b<NT, SIN>(b[q], in_row, wave, q);
    (chain_step<N, PART, STEP0, KB, NT, SIN, Q>(acc, b, in_row, ring, ws, wave, last_ok), ...);
}
template <class N, int PART, int STEP0, int KB, int NT, int SIN>
__device__ __forceinline__ void chain_mm(ChainAcc<NT> &acc, lds_p in_row /* image base + lane row part */, Ring &ring,
                                         const WStream &ws, int wave) {
    chain_mm_impl<N, PART, STEP0, KB, NT, SIN>(acc, in_row, ring, ws, wave, std::make_integer_sequence<int, KB>{});
}

// epilogue of a chain product into image OUT (row stride SOUT): FWD: [LeakyReLU] -> bf16 -> store;  !FWD: [mask with the
// sign of what the image holds at the same place] -> bf16 -> store in place.  `fn(tile, row tile, value)` visits every tile.
template <int NT, int SOUT, class Fn>
__device__ __forceinline__ void acc_visit(ChainAcc<NT> &acc, lds_p img, const Lay &lay, int wave, Fn fn) {
    using SP = Split<NT>;
    // N-split slot k: tile t = wave + 4 k: parity of t = parity of wave, chunk pair 2 t -> 32 (t & ~1) bytes
    const lds_p wn = img + lay.wr(wave & 1) + 32 * (wave & ~1);
#pragma unroll
    for (int k = 0; k < SP::NS; ++k) {
        if (SP::ragged && k == SP::NS - 1 && wave + 4 * k >= NT) continue;
#pragma unroll
        for (int m = 0; m < 4; ++m) fn(acc.an[k][m], wn + 128 * k + 16 * m * SOUT);
    }
    const lds_p wm0 = img + 16 * wave * SOUT;
#pragma unroll
    for (int k = 0; k < SP::MS; ++k) {
        const int t = SP::m0 + k;                 // compile-time after unrolling
        fn(acc.am[k], wm0 + lay.wr(t & 1) + 32 * (t & ~1));
    }
}

// the same, `fn(tile, address, output tile index, row tile index)`
template <int NT, int SOUT, class Fn>
__device__ __forceinline__ void acc_visit_idx(ChainAcc<NT> &acc, lds_p img, const Lay &lay, int wave, Fn fn) {
    using SP = Split<NT>;
    const lds_p wn = img + lay.wr(wave & 1) + 32 * (wave & ~1);
#pragma unroll
    for (int k = 0; k < SP::NS; ++k) {
        if (SP::ragged && k == SP::NS - 1 && wave + 4 * k >= NT) continue;
#pragma unroll
        for (int m = 0; m < 4; ++m) fn(acc.an[k][m], wn + 128 * k + 16 * m * SOUT, wave + 4 * k, m);
    }
    const lds_p wm0 = img + 16 * wave * SOUT;
#pragma unroll
    for (int k = 0; k < SP::MS; ++k) {
        const int t = SP::m0 + k;
        fn(acc.am[k], wm0 + lay.wr(t & 1) + 32 * (t & ~1), t, wave);
    }
}

// The backward epilogue reads, per tile, the 8 bytes the image holds at the place it is about to write (the sign mask of the
// activation).  Same visits as acc_visit, with those reads issued one group of four tiles AHEAD of the arithmetic that needs
// them (read-then-use per tile left one LDS round trip per tile exposed): `pre(addr)` reads, `fn(tile, addr, value)` consumes.
template <int NT, int SOUT, class Pre, class Fn>
__device__ __forceinline__ void acc_visit_pre(ChainAcc<NT> &acc, lds_p img, const Lay &lay, int wave, Pre pre, Fn fn) {
    using SP = Split<NT>;
    const lds_p wn = img + lay.wr(wave & 1) + 32 * (wave & ~1);
    const lds_p wm0 = img + 16 * wave * SOUT;
    u2 ym[SP::MS > 0 ? SP::MS : 1];
#pragma unroll
    for (int k = 0; k < SP::MS; ++k) {
        const int t = SP::m0 + k;
        ym[k] = pre(wm0 + lay.wr(t & 1) + 32 * (t & ~1));
    }
    u2 yc[4], yn[4];
    if (SP::NS > 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m) yc[m] = pre(wn + 16 * m * SOUT);
    }
#pragma unroll
    for (int k = 0; k < SP::NS; ++k) {
        if (k + 1 < SP::NS) {
#pragma unroll
            for (int m = 0; m < 4; ++m) yn[m] = pre(wn + 128 * (k + 1) + 16 * m * SOUT);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(SP::ragged && k == SP::NS - 1 && wave + 4 * k >= NT)) {
#pragma unroll
            for (int m = 0; m < 4; ++m) fn(acc.an[k][m], wn + 128 * k + 16 * m * SOUT, yc[m]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) yc[m] = yn[m];
    }
#pragma unroll
    for (int k = 0; k < SP::MS; ++k) {
        const int t = SP::m0 + k;
        fn(acc.am[k], wm0 + lay.wr(t & 1) + 32 * (t & ~1), ym[k]);
    }
}

// ---- weight-gradient tiles of layer l --------------------------------------------------------------------------------
// A operand: dZ_l^T (image ZI, stride SZ), B operand: [X_l | 1] (image XI, stride SX), both by transposed reads;
// contraction over the 64 rows = 2 MFMAs per tile.  Tiles owned by this wave: see TNet::by_nt.
// The wave OWNS the tiles {wave + 4 i} of one side (output tiles when by_nt, input tiles otherwise) and streams over all tiles
// of the other side.  Its own operands (NO x 2 halves) are read once and stay in registers; every streamed operand feeds the
// MFMAs of all NO owned tiles (the first version re-read the streamed side per owned tile: twice to four times the LDS traffic),
// and the streamed reads run kDWD tiles ahead of their MFMAs: one tile ahead = 64 MFMA cycles left every step waiting for an
// LDS round trip of 150-200 cycles (13 steps of dW_6 took 2,280 cycles for 832 cycles of MFMAs).
// An owned slot that does not exist on this wave (13 tiles over 4 waves) is computed on a clamped address and never flushed:
// uniform code, and that wave would wait at the barrier anyway.
constexpr int kDWD = 2;
template <class N, int l> struct DwGeo {
    static constexpr int NT = N::nt(l), KT = N::kt(l);
    static constexpr bool BYN = N::by_nt(l);
    static constexpr int NO = BYN ? cdiv(NT, 4) : cdiv(KT, 4);      // owned slots
    static constexpr int OWN = BYN ? NT : KT;                       // tiles on the owned side
    static constexpr int NS = BYN ? KT : NT;                        // streamed tiles
};
template <class N, int l, int SZ, int SX, int S>
__device__ __forceinline__ void dw_step(v4 (&acc)[N::dwn(l)], const bf8 (&own)[DwGeo<N, l>::NO][2], bf8 (&ring)[kDWD + 1][2], lds_p sbase0,
                                        lds_p sbase1) {
    using G = DwGeo<N, l>;
    constexpr int SS = G::BYN ? SX : SZ;          // stride of the streamed image
    if constexpr (S + kDWD < G::NS) {
        constexpr int t = S + kDWD;
        const lds_p sb = ((t & 1) ? sbase1 : sbase0) + 32 * (t & ~1);
        ring[t % (kDWD + 1)][0] = tr_operand<SS>(sb, 0);
        ring[t % (kDWD + 1)][1] = tr_operand<SS>(sb, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const bf8 (&st)[2] = ring[S % (kDWD + 1)];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < G::NO; ++i) {
            v4 &c = acc[i * G::NS + S];
            c = G::BYN ? mfma(own[i][h], st[h], c) : mfma(st[h], own[i][h], c);      // A = dZ^T tile, B = [X | 1] tile
        }
    __builtin_amdgcn_sched_barrier(0);
}
template <class N, int l, int SZ, int SX, int... S>
__device__ __forceinline__ void dw_phase_impl(v4 (&acc)[N::dwn(l)], lds_p zimg, lds_p ximg, const Lay &lz, const Lay &lx, int wave,
                                              std::integer_sequence<int, S...>) {
    using G = DwGeo<N, l>;
    constexpr int SO = G::BYN ? SZ : SX, SS = G::BYN ? SX : SZ;
    const lds_p oimg = G::BYN ? zimg : ximg, simg = G::BYN ? ximg : zimg;
    const Lay &lo = G::BYN ? lz : lx, &lst = G::BYN ? lx : lz;
    const lds_p ob = oimg + lo.tr(wave & 1) + 32 * (wave & ~1);      // owned tile wave + 4 i -> + 128 i
    bf8 own[G::NO][2];
#pragma unroll
    for (int i = 0; i < G::NO; ++i) {
        const int ii = (i == 0 || wave + 4 * i < G::OWN) ? i : i - 1;      // wave-uniform clamp
        own[i][0] = tr_operand<SO>(ob + 128 * ii, 0);
        own[i][1] = tr_operand<SO>(ob + 128 * ii, 1);
    }
    const lds_p sb0 = simg + lst.tr0, sb1 = simg + lst.tr1;
    bf8 ring[kDWD + 1][2];
#pragma unroll
    for (int t = 0; t < kDWD && t < G::NS; ++t) {
        const lds_p sb = ((t & 1) ? sb1 : sb0) + 32 * (t & ~1);
        ring[t][0] = tr_operand<SS>(sb, 0);
        ring[t][1] = tr_operand<SS>(sb, 1);
    }
    (dw_step<N, l, SZ, SX, S>(acc, own, ring, sb0, sb1), ...);
}
template <class N, int l, int SZ, int SX>
__device__ __forceinline__ void dw_phase(v4 (&acc)[N::dwn(l)], lds_p zimg, lds_p ximg, const Lay &lz, const Lay &lx, int wave) {
    static_assert(N::dwn(l) == DwGeo<N, l>::NO * DwGeo<N, l>::NS, "accumulator count");
    dw_phase_impl<N, l, SZ, SX>(acc, zimg, ximg, lz, lx, wave, std::make_integer_sequence<int, DwGeo<N, l>::NS>{});
}

template <class N, int l>
__device__ __forceinline__ void dw_flush(v4 *__restrict__ slab, const v4 (&acc)[N::dwn(l)], int lane, int wave) {
    constexpr int NT = N::nt(l), KT = N::kt(l);
    // partial-gradient buffer is TILE-major: [tile][workgroup][64 lanes]; `slab` points at this workgroup's column
    if constexpr (N::by_nt(l)) {
#pragma unroll
        for (int i = 0; i < cdiv(NT, 4); ++i)
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const int t = wave + 4 * i;
                if (t < NT) slab[(int64_t)(N::slab_off(l) + k * NT + t) * gridDim.x * 64 + lane] = acc[i * KT + k];
            }
    } else {
#pragma unroll
        for (int i = 0; i < cdiv(KT, 4); ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int k = wave + 4 * i;
                if (k < KT) slab[(int64_t)(N::slab_off(l) + k * NT + t) * gridDim.x * 64 + lane] = acc[i * NT + t];
            }
    }
}
template <int NA> __device__ __forceinline__ void zero_acc(v4 (&a)[NA]) {
#pragma unroll
    for (int i = 0; i < NA; ++i) a[i] = (v4){0.f, 0.f, 0.f, 0.f};
}

// ---- input rows ----------------------------------------------------------------------------------------------------
// lane (j, g), g < F/8, reads features 8 g .. 8 g + 7 of row 16 wave + j (branch-free: rows beyond n read row 0)
template <int F> struct RawX { double d[8]; };
template <int F>
__device__ __forceinline__ void x_issue(RawX<F> &raw, const void *x, int is_f64, int64_t row, int64_t n, int g) {
    const int64_t base = (row < n ? row : 0) * F + (8 * g < F ? 8 * g : 0);
    if (is_f64) {
        const double2 *p = (const double2 *)((const double *)x + base);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const double2 t = p[e]; raw.d[2 * e] = t.x; raw.d[2 * e + 1] = t.y; }
    } else {
        const float4 *p = (const float4 *)((const float *)x + base);
        const float4 t0 = p[0], t1 = p[1];
        raw.d[0] = t0.x; raw.d[1] = t0.y; raw.d[2] = t0.z; raw.d[3] = t0.w; raw.d[4] = t1.x; raw.d[5] = t1.y; raw.d[6] = t1.z; raw.d[7] = t1.w;
    }
}

#ifdef BAMD_BF16_TRACE   // debug build: shader-clock stamps of workgroup 0, wave 0 at every phase boundary (tools/bf16_trace.py)
__device__ unsigned long long g_bf16_trace[2][4][128];
// stamps go to LDS (a global store per stamp sat in front of every later counted vmcnt wait: in-order retirement) and are
// copied out once after the loop
#define BT(i) do { if ((threadIdx.x & 63) == 0) bt_lds[(threadIdx.x >> 6) * 128 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BT(i) do {} while (0)
#endif

// ---- M-split products: this wave computes ALL NT output tiles of ITS 16 rows; the KB B operands are in registers ----------
// fragment f = q NT + t of the product's stream feeds MFMA (k block q, tile t); four fragments per ring step
template <class N, int PART, int STEP0, int KB, int NT, int F_>
__device__ __forceinline__ void mstep(v4 (&acc)[NT], const bf8 (&b)[KB], Ring &ring, const WStream &ws, int wave) {
    constexpr int q = F_ / NT, t = F_ % NT, st = STEP0 + F_ / 4, slot = F_ % 4;
    if constexpr (slot == 0) issue<N, PART, st + kRD - 1>(ring, ws, wave);
    const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
    acc[t] = mfma(ring.buf[st % kRD][slot], b[q], q == 0 ? zero : acc[t]);
    if constexpr (slot == 3) __builtin_amdgcn_sched_barrier(0);
}
template <class N, int PART, int STEP0, int KB, int NT, int... F_>
__device__ __forceinline__ void mchain_impl(v4 (&acc)[NT], const bf8 (&b)[KB], Ring &ring, const WStream &ws, int wave,
                                            std::integer_sequence<int, F_...>) {
    (mstep<N, PART, STEP0, KB, NT, F_>(acc, b, ring, ws, wave), ...);
    __builtin_amdgcn_sched_barrier(0);
}
template <class N, int PART, int STEP0, int KB, int NT>
__device__ __forceinline__ void mchain(v4 (&acc)[NT], const bf8 (&b)[KB], Ring &ring, const WStream &ws, int wave) {
    mchain_impl<N, PART, STEP0, KB, NT>(acc, b, ring, ws, wave, std::make_integer_sequence<int, KB * NT>{});
}
// B operands of the next product from this product's packed C tiles: k block q = tiles 2 q, 2 q + 1 (missing tile: zeros)
template <int KB, int NT>
__device__ __forceinline__ void regfeed(bf8 (&b)[KB], const u2 (&pk)[NT]) {
    typedef unsigned u4_ __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int q = 0; q < KB; ++q) {
        const u2 lo = 2 * q < NT ? pk[2 * q] : (u2){0u, 0u}, hi = 2 * q + 1 < NT ? pk[2 * q + 1] : (u2){0u, 0u};
        b[q] = __builtin_bit_cast(bf8, (u4_){lo[0], lo[1], hi[0], hi[1]});
    }
}
// B operands from image rows of THIS wave's row tile (`own` = image base + lane row part + 16 wave S)
template <int KB>
__device__ __forceinline__ void imgfeed(bf8 (&b)[KB], lds_p own) {
#pragma unroll
    for (int q = 0; q < KB; ++q) b[q] = lds_b128(own + 64 * q);
}

template <int F, int Z, int PART>
__global__ void __launch_bounds__(256) bf16_train_kernel(const uint4 *__restrict__ wfrags, const void *__restrict__ xin, int in_f64, int64_t n,
                                                         const double *__restrict__ feats, v4 *__restrict__ slabs,
                                                         u2 *__restrict__ dz, int loss_tile) {
    using N = TNet<F, Z>;
    using P = Part<PART>;
    using SC = Sched<N, PART>;
    extern __shared__ __attribute__((aligned(256))) unsigned char lds_raw[];
    const lds_p img = (lds_p)lds_raw;
    float *xf = (float *)(lds_raw + N::img_bytes());          // fp32 copy of the normalised input rows: [64][32]
    double *fl = (double *)(xf + kRows * 32);                 // [0..31] min, [32..63] range
#ifdef BAMD_BF16_TRACE
    unsigned long long *bt_lds = (unsigned long long *)(fl + 64 + 256);
#endif
    for (int i = threadIdx.x; i < N::img_bytes() / 16; i += 256) ((uint4 *)lds_raw)[i] = make_uint4(0, 0, 0, 0);   // finite padding slots
    if (threadIdx.x < 64) {
        const int f = threadIdx.x & 31, which = threadIdx.x >> 5;
        fl[threadIdx.x] = (feats && f < F) ? feats[which * F + f] : (which ? 1.0 : 0.0);
    }
    __syncthreads();
    int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ngroups = (n + kRows - 1) / kRows;
    WStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)wfrags, 0, N::nfrag() * 1024, 0x00020000);
    ws.voff = lane * 16;
    v4 *slab = slabs + (int64_t)blockIdx.x * 64;

    v4 g7[P::has(7) ? N::dwn(7) : 1], g6[P::has(6) ? N::dwn(6) : 1], g5[P::has(5) ? N::dwn(5) : 1], g4[P::has(4) ? N::dwn(4) : 1];
    v4 g3[P::has(3) ? N::dwn(3) : 1], g2[P::has(2) ? N::dwn(2) : 1], g1[P::has(1) ? N::dwn(1) : 1], g0[P::has(0) ? N::dwn(0) : 1];
    zero_acc(g7); zero_acc(g6); zero_acc(g5); zero_acc(g4); zero_acc(g3); zero_acc(g2); zero_acc(g1); zero_acc(g0);
    double lacc = 0.0;
    Ring ring;
    issue<N, PART, 0>(ring, ws, wave);
    if constexpr (kRD > 2) issue<N, PART, 1>(ring, ws, wave);
    if constexpr (kRD > 3) issue<N, PART, 2>(ring, ws, wave);
    if constexpr (kRD > 4) issue<N, PART, 3>(ring, ws, wave);
    if constexpr (kRD > 5) issue<N, PART, 4>(ring, ws, wave);
    static_assert(kRD >= 2 && kRD <= 6, "ring depth");
    RawX<F> xraw;
    x_issue<F>(xraw, xin, in_f64, (int64_t)blockIdx.x * kRows + 16 * wave + (lane & 15), n, lane >> 4);
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    constexpr int HB = N::hand_tiles * 32;                   // hand-off bytes per row
    static_assert(N::iblocks(2) * 64 >= HB, "hand-off row fits the image row");
    const int hr = threadIdx.x >> 2, hp = threadIdx.x & 3;
    u4v hand[PART == 1 ? 4 : 1];
    auto hand_load = [&](int64_t g_) {
        if constexpr (PART == 1) {
            const u4v *src = (const u4v *)((const unsigned char *)dz + (g_ * kRows + hr) * HB) + hp;
#pragma unroll
            for (int i = 0; i < 4; ++i) hand[i] = src[(16 * (hp + 4 * i) < HB) ? 4 * i : 0];
        }
    };
    hand_load(blockIdx.x);

    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        // keep the weight loads and the LDS address arithmetic inside the loop (LICM would hoist hundreds of registers)
        asm volatile("" : "+v"(ws.voff), "+s"(wave), "+v"(lane));
        BT(0);
        const int j = lane & 15, g = lane >> 4;
        Lays ls;
        ls.s1 = make_lay<64>(lane); ls.s3 = make_lay<192>(lane); ls.s5 = make_lay<320>(lane); ls.s7 = make_lay<448>(lane);
        const int64_t row = grp * kRows + 16 * wave + j;
        const bool valid = row < n;
        // ---- input rows -> image 0 (bf16, with the ones slot) and the fp32 copy the loss uses -------------------------
        {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int f = (8 * g < F ? 8 * g : 0) + e;
                double d = xraw.d[e];
                if (feats) d = (d - fl[f]) / fl[32 + f];
                v[e] = 8 * g < F ? (float)d : (8 * g == F && e == 0 ? 1.0f : 0.f);     // slot F = the ones column
            }
            bf8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
            *(bf8 __attribute__((address_space(3))) *)(img + N::ioff(0) + lay_of<N::istride(0)>(ls).row + 16 * wave * N::istride(0)) = o;
            if constexpr (PART == 0) {
                float *xr = xf + (16 * wave + j) * 32 + 8 * g;
                *(float4 *)xr = make_float4(v[0], v[1], v[2], v[3]);
                *(float4 *)(xr + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if constexpr (PART == 0)
                x_issue<F>(xraw, xin, in_f64, row + (int64_t)gridDim.x * kRows, n, g);       // next iteration's rows, a whole iteration ahead
        }
        if constexpr (PART == 1) {
            // dZ_1 of these rows from the first launch (row-major, 224 B per row) -> region 2: thread (r, p) moves chunks
            // p, p + 4, p + 8 (and p + 12 for p < 2) of row r to their swizzled places; requested a whole iteration ahead, like the rows
            const lds_p dst = img + N::zoff(1) + hr * N::istride(2) + ((hp ^ ((hr >> 1) & 3)) << 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (16 * (hp + 4 * i) < HB) *(u4v __attribute__((address_space(3))) *)(dst + 64 * i) = hand[i];
        }
        __syncthreads();
        BT(1);

        // ---- forward, N-split layers ------------------------------------------------------------------------------------
#define BAMD_FWD(l)                                                                                                          \
        {                                                                                                                    \
            ChainAcc<N::nt(l)> acc;                                                                                          \
            chain_mm<N, PART, SC::fstep(l), N::kb(l), N::nt(l), N::istride(l)>(                                              \
                acc, img + N::ioff(l) + lay_of<N::istride(l)>(ls).row, ring, ws, wave);                                      \
            BT(40 + 2 * (l));                                                                                                \
            acc_visit<N::nt(l), N::istride(l + 1)>(acc, img + N::ioff(l + 1), lay_of<N::istride(l + 1)>(ls), wave,           \
                                                   [&](v4 &a, lds_p dst) { if (N::act(l)) lrelu4(a); lds_w64(dst, pack4(a)); }); \
            BT(41 + 2 * (l));                                                                                                \
            __syncthreads();                                                                                                 \
            BT(2 + (l));                                                                                                     \
        }
        // ---- forward, M-split layer: B operands `bin` -> packed output tiles `pk` (also stored into image l + 1, own rows) ---
#define BAMD_FWD_M(l, bin, pk)                                                                                               \
        u2 pk[N::nt(l)];                                                                                                     \
        {                                                                                                                    \
            v4 acc[N::nt(l)];                                                                                                \
            mchain<N, PART, SC::fstep(l), N::kb(l), N::nt(l)>(acc, bin, ring, ws, wave);                                     \
            const Lay &lo = lay_of<N::istride((l) + 1)>(ls);                                                                 \
            const lds_p ob = img + N::ioff((l) + 1) + 16 * wave * N::istride((l) + 1);                                       \
            _Pragma("unroll") for (int t = 0; t < N::nt(l); ++t) {                                                           \
                if (N::act(l)) lrelu4(acc[t]);                                                                               \
                pk[t] = pack4(acc[t]);                                                                                       \
                lds_w64(ob + lo.wr(t & 1) + 32 * (t & ~1), pk[t]);                                                           \
            }                                                                                                                \
        }
        if constexpr (P::fwd_end >= 1) {
            BAMD_FWD(0)
        }
        if constexpr (P::fwd_end == 8) {
            BAMD_FWD(1)
            {
                static_assert(N::mf(2) && N::mf(3) && N::mf(4) && N::mf(5) && !N::regfed_f(2), "register chain 2..5");
                bf8 b2[N::kb(2)];
                imgfeed<N::kb(2)>(b2, img + N::ioff(2) + lay_of<N::istride(2)>(ls).row + 16 * wave * N::istride(2));
                BAMD_FWD_M(2, b2, p3)
                bf8 b3[N::kb(3)];
                regfeed<N::kb(3), N::nt(2)>(b3, p3);
                BAMD_FWD_M(3, b3, p4)
                bf8 b4[N::kb(4)];
                regfeed<N::kb(4), N::nt(3)>(b4, p4);
                BAMD_FWD_M(4, b4, p5)
                bf8 b5[N::kb(5)];
                regfeed<N::kb(5), N::nt(4)>(b5, p5);
                BAMD_FWD_M(5, b5, p6)
                (void)p6;
                BT(54);
                __syncthreads();
                BT(55);
            }
            BAMD_FWD(6)
            // layer 7 + loss: NT = 2 -> this wave holds both output tiles of ITS 16 rows
            ChainAcc<N::nt(7)> acc;
            static_assert(N::nt(7) < 4, "loss epilogue expects the M-split form");
            chain_mm<N, PART, SC::fstep(7), N::kb(7), N::nt(7), N::istride(7)>(acc, img + N::ioff(7) + lay_of<N::istride(7)>(ls).row,
                                                                             ring, ws, wave);
            const Lay &l8 = lay_of<N::istride(8)>(ls);
#pragma unroll
            for (int t = 0; t < N::nt(7); ++t) {
                const float4 xv = *(const float4 *)(xf + (16 * wave + j) * 32 + 16 * t + 4 * g);
                const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
                v4 d;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = acc.am[t][r] - xs[r];
                    const bool live = valid && 16 * t + 4 * g + r < F;
                    if (live) lacc += (double)e * (double)e;
                    d[r] = live ? e * (2.0f / (float)F) : 0.f;              // dL/drecon = 2 (r - x) / C  (utils.py:195-199)
                }
                lds_w64(img + N::ioff(8) + 16 * wave * N::istride(8) + l8.wr(t & 1) + 32 * (t & ~1), pack4(d));
            }
            __syncthreads();
            BT(9);
        }

        // ---- backward, N-split: per layer  [input-gradient MFMAs] [mask + store dZ_{l-1} into its own region] [weight-gradient tiles]  barrier
#define BAMD_BWD(l, G)                                                                                                       \
        if constexpr (P::has(l)) {                                                                                           \
            constexpr int SZ = N::istride((l) + 1);   /* dZ_l has the shape of X_{l+1} */                                    \
            const lds_p zimg = img + N::zoff(l);                                                                             \
            if constexpr ((l) >= 1) {                                                                                        \
                constexpr int NTB = N::ntb(l), SO = N::istride(l);                                                           \
                ChainAcc<NTB> acc;                                                                                           \
                chain_mm<N, PART, SC::bstep(l), N::kbb(l), NTB, SZ>(acc, zimg + lay_of<SZ>(ls).row, ring, ws, wave);         \
                BT(70 + 2 * (l));                                                                                            \
                if constexpr (PART == 1 && (l) == 1) {                                                                         \
                    x_issue<F>(xraw, xin, in_f64, row + (int64_t)gridDim.x * kRows, n, g);                                   \
                    hand_load(grp + gridDim.x < ngroups ? grp + gridDim.x : ngroups - 1);                                    \
                }                                                                                                            \
                constexpr int DELTA = N::zoff((l) - 1) - N::ioff(l);   /* same stride, same lane offsets: a constant shift */ \
                if constexpr (N::act((l) - 1))                                                                           \
                    acc_visit_pre<NTB, SO>(acc, img + N::ioff(l), lay_of<SO>(ls), wave,                                  \
                                           [&](lds_p src) { return lds_b64(src); },                                      \
                                           [&](v4 &a, lds_p src, u2 y) { lds_w64(src + DELTA, lrelu_bwd_pack4(a, y)); }); \
                else                                                                                                     \
                    acc_visit<NTB, SO>(acc, img + N::ioff(l), lay_of<SO>(ls), wave,                                      \
                                       [&](v4 &a, lds_p src) { lds_w64(src + DELTA, pack4(a)); });                       \
                BT(20 + 2 * (l));                                                                                        \
                dw_phase<N, l, SZ, SO>(G, zimg, img + N::ioff(l), lay_of<SZ>(ls), lay_of<SO>(ls), wave);                 \
            } else {                                                                                                         \
                BT(20 + 2 * (l));                                                                                            \
                dw_phase<N, l, SZ, N::istride(l)>(G, zimg, img + N::ioff(l), lay_of<SZ>(ls), lay_of<N::istride(l)>(ls), wave); \
            }                                                                                                                \
            BT(71 + 2 * (l));                                                                                                \
            __syncthreads();                                                                                                 \
            BT(21 + 2 * (l));                                                                                                \
        }
        // ---- backward, M-split product of layer l: B operands `bin` (dZ_l) -> packed dZ_{l-1} tiles `pk`, stored into their region
        //      (own rows), masked with the sign of X_l where layer l - 1 has an activation
#define BAMD_BWD_M(l, bin, pk)                                                                                               \
        u2 pk[N::ntb(l)];                                                                                                    \
        {                                                                                                                    \
            const Lay &lo = lay_of<N::istride(l)>(ls);                                                                       \
            const lds_p xb = img + N::ioff(l) + 16 * wave * N::istride(l);                                                   \
            constexpr int DELTA = N::zoff((l) - 1) - N::ioff(l);                                                             \
            u2 y[N::ntb(l)];                                                                                                 \
            if constexpr (N::act((l) - 1)) {                                                                                 \
                _Pragma("unroll") for (int t = 0; t < N::ntb(l); ++t) y[t] = lds_b64(xb + lo.wr(t & 1) + 32 * (t & ~1));     \
            }                                                                                                                \
            v4 acc[N::ntb(l)];                                                                                               \
            mchain<N, PART, SC::bstep(l), N::kbb(l), N::ntb(l)>(acc, bin, ring, ws, wave);                                   \
            _Pragma("unroll") for (int t = 0; t < N::ntb(l); ++t) {                                                          \
                if constexpr (N::act((l) - 1)) pk[t] = lrelu_bwd_pack4(acc[t], y[t]);                                        \
                else pk[t] = pack4(acc[t]);                                                                                  \
                if constexpr ((l) - 1 >= P::bwd_lo) lds_w64(xb + DELTA + lo.wr(t & 1) + 32 * (t & ~1), pk[t]);               \
            }                                                                                                                \
        }
        BAMD_BWD(7, g7) BAMD_BWD(6, g6)
        if constexpr (PART == 0) {
            static_assert(N::mb(5) && N::mb(4) && N::mb(3) && N::mb(2) && !N::regfed_b(5), "register chain 5..2");
            bf8 c5[N::kbb(5)];
            imgfeed<N::kbb(5)>(c5, img + N::zoff(5) + lay_of<N::istride(6)>(ls).row + 16 * wave * N::istride(6));
            BAMD_BWD_M(5, c5, q4)
            bf8 c4[N::kbb(4)];
            regfeed<N::kbb(4), N::ntb(5)>(c4, q4);
            BAMD_BWD_M(4, c4, q3)
            bf8 c3[N::kbb(3)];
            regfeed<N::kbb(3), N::ntb(4)>(c3, q3);
            BAMD_BWD_M(3, c3, q2)
            bf8 c2[N::kbb(2)];
            regfeed<N::kbb(2), N::ntb(3)>(c2, q2);
            BAMD_BWD_M(2, c2, q1)
            // hand-off to the second launch: dZ_1, row-major, 8 bytes (features 16 t + 4 g ..) per lane and tile; rows beyond n
            // carry zeros (their dL/drecon is zero)
            static_assert(N::ntb(2) == N::hand_tiles, "hand-off width");
#pragma unroll
            for (int t = 0; t < N::ntb(2); ++t) {
                dz[row * (4 * N::hand_tiles) + 4 * t + g] = q1[t];
            }
            BT(56);
            __syncthreads();
            BT(57);
            // weight-gradient tiles of the four narrow layers (images complete for all 64 rows now)
#define BAMD_DW(l, G) dw_phase<N, l, N::istride((l) + 1), N::istride(l)>(G, img + N::zoff(l), img + N::ioff(l), lay_of<N::istride((l) + 1)>(ls), \
                                                                        lay_of<N::istride(l)>(ls), wave);
            BAMD_DW(5, g5) BAMD_DW(4, g4) BAMD_DW(3, g3) BAMD_DW(2, g2)
#undef BAMD_DW
            BT(58);
            __syncthreads();     // the next iteration's rows overwrite region 0 (dZ_3), its layer 0 region 1 (dZ_2)
            BT(59);
        } else {
            BAMD_BWD(1, g1) BAMD_BWD(0, g0)
        }
#undef BAMD_FWD
#undef BAMD_FWD_M
#undef BAMD_BWD
#undef BAMD_BWD_M
        // step over the padding steps and prime the ring for the next iteration (step total + j is issued by real step
        // total + j - (kRD - 1) when that one exists)
        if constexpr (SC::total + 0 - (kRD - 1) >= SC::real) issue<N, PART, SC::total + 0>(ring, ws, wave);
        if constexpr (kRD > 2 && SC::total + 1 - (kRD - 1) >= SC::real) issue<N, PART, SC::total + 1>(ring, ws, wave);
        if constexpr (kRD > 3 && SC::total + 2 - (kRD - 1) >= SC::real) issue<N, PART, SC::total + 2>(ring, ws, wave);
        if constexpr (kRD > 4 && SC::total + 3 - (kRD - 1) >= SC::real) issue<N, PART, SC::total + 3>(ring, ws, wave);
        if constexpr (kRD > 5 && SC::total + 4 - (kRD - 1) >= SC::real) issue<N, PART, SC::total + 4>(ring, ws, wave);
    }
#ifdef BAMD_BF16_TRACE
    __syncthreads();
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < 512; i += 256) g_bf16_trace[PART][i >> 7][i & 127] = bt_lds[i];
#endif
    if constexpr (P::has(7)) dw_flush<N, 7>(slab, g7, lane, wave);
    if constexpr (P::has(6)) dw_flush<N, 6>(slab, g6, lane, wave);
    if constexpr (P::has(5)) dw_flush<N, 5>(slab, g5, lane, wave);
    if constexpr (P::has(4)) dw_flush<N, 4>(slab, g4, lane, wave);
    if constexpr (P::has(3)) dw_flush<N, 3>(slab, g3, lane, wave);
    if constexpr (P::has(2)) dw_flush<N, 2>(slab, g2, lane, wave);
    if constexpr (P::has(1)) dw_flush<N, 1>(slab, g1, lane, wave);
    if constexpr (P::has(0)) dw_flush<N, 0>(slab, g0, lane, wave);
    if constexpr (P::fwd_end == 8) {   // per-workgroup loss partial (fixed-order tree), stored after the tiles
        __syncthreads();
        double *sh = (double *)lds_raw;
        sh[threadIdx.x] = lacc;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) ((double *)(slabs + (int64_t)loss_tile * gridDim.x * 64))[blockIdx.x] = sh[0];
    }
}

// Fixed-order reduction of the per-workgroup partial gradients ([tile][workgroup][64 lanes] float4) into the canonical
// (state-dict) layout; block `ntiles`: grads[np] = sum of the loss partials / C.  One wave per tile (see fused.hip).
__global__ void __launch_bounds__(256) reduce_tiles_k(const v4 *__restrict__ slabs, int nslab, int ntiles, const int *__restrict__ inv_map,
                                                      int np, double inv_c, float *__restrict__ grads) {
    // one workgroup per tile: wave w sums the w-th quarter of the workgroups' slabs in order, the four partial sums are added
    // in wave order (fixed => bitwise reproducible)
    __shared__ v4 part[4][64];
    const int tile = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (tile == ntiles) {
        const double l = block_sum_fixed((const double *)(slabs + (int64_t)ntiles * nslab * 64), nslab, (double *)part);
        if (threadIdx.x == 0) grads[np] = (float)(l * inv_c);
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
        if (p >= 0) grads[p] = s[c];
    }
}

// params (fp32, state-dict order) -> bf16 weight fragments through an index map (-1: zero, -2: one)
__global__ void __launch_bounds__(256) pack_train_k(const float *__restrict__ params, const int *__restrict__ src, int count,
                                                    __bf16 *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int s = src[i];
    dst[i] = (__bf16)(s >= 0 ? params[s] : (s == -2 ? 1.0f : 0.f));
}

struct TrainOps;
struct TrainState {
    const TrainOps *ops = nullptr;
    DevBuf src, w, inv, dz;
    int wcount = 0, ntiles = 0, nparams = 0, n_features = 0;
    int nwg_max = 256;
};
struct TrainOps {
    int (*setup)(bamd_handle *, TrainState *);
    int (*fwd_bwd)(bamd_handle *, TrainState *, const void *, int, int64_t, const double *, float *, hipStream_t);
    int (*pack)(bamd_handle *, TrainState *, hipStream_t);
};
TrainState *tstate(bamd_handle *h) { return (TrainState *)h->bf16_train_state; }

template <int F, int Z> struct TImpl {
    using N = TNet<F, Z>;
#ifdef BAMD_BF16_TRACE
    static constexpr size_t lds_bytes() { return (size_t)N::img_bytes() + kRows * 32 * 4 + 64 * 8 + 2048 + 4096; }
#else
    static constexpr size_t lds_bytes() { return (size_t)N::img_bytes() + kRows * 32 * 4 + 64 * 8 + 2048; }
#endif
    static_assert(lds_bytes() <= 160 * 1024, "LDS images exceed one CU");
    static bool matches(const bamd_handle *h) {
        if (h->L != 8) return false;
        for (int i = 0; i <= 8; ++i)
            if (h->dims[i] != N::dim(i)) return false;
        return true;
    }
    // gradient map (accumulator tile slot -> canonical parameter), tile count, parameter count
    static int setup_maps(bamd_handle *h, TrainState *st) {
        // accumulator tile (kt, nt) of layer l, lane (j, g), register r = dW[16 nt + 4 g + r][16 kt + j]; column K = db
        const int ntiles = N::slab_off(N::L);
        std::vector<int> inv((size_t)ntiles * 256, -1);
        for (int l = 0; l < N::L; ++l) {
            const int K = N::dim(l), NN = N::dim(l + 1);
            for (int k = 0; k < N::kt(l); ++k)
                for (int t = 0; t < N::nt(l); ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int r = 0; r < 4; ++r) {
                            const int n = 16 * t + 4 * (lane >> 4) + r, kc = 16 * k + (lane & 15);
                            if (n >= NN) continue;
                            const size_t o = ((size_t)(N::slab_off(l) + k * N::nt(l) + t) * 64 + lane) * 4 + r;
                            if (kc < K) inv[o] = N::w_off(l) + n * K + kc;
                            else if (kc == K) inv[o] = N::b_off(l) + n;
                        }
        }
        {   // every parameter must be produced exactly once
            std::vector<char> seen(N::nparams(), 0);
            for (int v : inv) if (v >= 0) seen[v]++;
            for (char c : seen) if (c != 1) { set_error("bf16 training: incomplete gradient map"); return BAMD_ERR_INVALID; }
        }
        st->ntiles = ntiles; st->nparams = N::nparams(); st->n_features = F;
        int rc = st->inv.ensure(inv.size() * sizeof(int));
        if (rc) return rc;
        BAMD_HIP(hipMemcpy(st->inv.p, inv.data(), inv.size() * sizeof(int), hipMemcpyHostToDevice));
        return BAMD_OK;
    }
    static int setup(bamd_handle *h, TrainState *st) {
        const size_t wcount = (size_t)N::nfrag() * 512;
        std::vector<int> src(wcount, -1);
        for (int l = 0; l < N::L; ++l) {
            const int K = N::dim(l), NN = N::dim(l + 1);
            // forward fragment (q, t): lane (i, g) element e = [W | b | .][16 t + i][32 q + 8 g + e]: input column K (the ones
            // slot of the image) holds the bias, and padding output K' = NN has a 1 there: it becomes the next image's ones slot
            for (int q = 0; q < N::kb(l); ++q)
                for (int t = 0; t < N::nt(l); ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int g = lane >> 4;
                            const int n = 16 * t + (lane & 15), k = N::regfed_f(l) ? 32 * q + 16 * (e >> 2) + 4 * g + (e & 3) : 32 * q + 8 * g + e;
                            int v = -1;
                            if (n < NN && k < K) v = N::w_off(l) + n * K + k;
                            else if (n < NN && k == K) v = N::b_off(l) + n;
                            else if (n == NN && k == K) v = -2;
                            src[((size_t)(N::ffo(l) + q * N::nt(l) + t) * 64 + lane) * 8 + e] = v;
                        }
            // backward fragment (q, t), l >= 1: lane (i, g) element e = W[32 q + 8 g + e][16 t + i]
            for (int q = 0; q < N::kbb(l) && l >= 1; ++q)
                for (int t = 0; t < N::ntb(l); ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int g = lane >> 4;
                            const int n = N::regfed_b(l) ? 32 * q + 16 * (e >> 2) + 4 * g + (e & 3) : 32 * q + 8 * g + e, k = 16 * t + (lane & 15);
                            if (n < NN && k < K) src[((size_t)(N::bfo(l) + q * N::ntb(l) + t) * 64 + lane) * 8 + e] = N::w_off(l) + n * K + k;
                        }
            static_assert(N::dim(0) % 32 && N::dim(1) % 32 && N::dim(2) % 32 && N::dim(3) % 32 && N::dim(4) % 32,
                          "the ones slot must lie inside the last k block of every layer input");
            static_assert(N::dim(1) % 16 && N::dim(2) % 16 && N::dim(3) % 16 && N::dim(4) % 16,
                          "every layer output needs a padding slot for the ones column");
        }
        int rc = setup_maps(h, st);
        if (rc) return rc;
        st->wcount = (int)wcount;
        rc = st->src.ensure(src.size() * sizeof(int));
        if (rc) return rc;
        rc = st->w.ensure(wcount * sizeof(__bf16) + 4096);
        if (rc) return rc;
        BAMD_HIP(hipMemcpy(st->src.p, src.data(), src.size() * sizeof(int), hipMemcpyHostToDevice));
        BAMD_HIP(hipFuncSetAttribute((const void *)bf16_train_kernel<F, Z, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes()));
        BAMD_HIP(hipFuncSetAttribute((const void *)bf16_train_kernel<F, Z, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes()));
        return BAMD_OK;
    }
    static int fwd_bwd(bamd_handle *h, TrainState *st, const void *x, int x_dtype, int64_t n, const double *features, float *grads,
                       hipStream_t s) {
        const int64_t ngroups = (n + kRows - 1) / kRows;
        const int grid = (int)(ngroups < st->nwg_max ? ngroups : st->nwg_max);
        // tiles + one double per workgroup for the loss
        int rc = h->slabs.ensure(((size_t)st->ntiles * 1024 + 16) * (size_t)grid);
        if (rc) return rc;
        rc = st->dz.ensure((size_t)ngroups * kRows * N::hand_tiles * 32);
        if (rc) return rc;
        hipLaunchKernelGGL((bf16_train_kernel<F, Z, 0>), dim3(grid), dim3(256), lds_bytes(), s, (const uint4 *)st->w.p, x, x_dtype == BAMD_F64, n, features, (v4 *)h->slabs.p, (u2 *)st->dz.p, st->ntiles);
        hipLaunchKernelGGL((bf16_train_kernel<F, Z, 1>), dim3(grid), dim3(256), lds_bytes(), s, (const uint4 *)st->w.p, x, x_dtype == BAMD_F64, n, features, (v4 *)h->slabs.p, (u2 *)st->dz.p, st->ntiles);
        hipLaunchKernelGGL(reduce_tiles_k, dim3(st->ntiles + 1), dim3(256), 0, s, (const v4 *)h->slabs.p, grid, st->ntiles,
                           (const int *)st->inv.p, st->nparams, 1.0 / F, grads);
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static int pack(bamd_handle *h, TrainState *st, hipStream_t s) {
        hipLaunchKernelGGL(pack_train_k, dim3((st->wcount + 255) / 256), dim3(256), 0, s, (const float *)h->params.p,
                           (const int *)st->src.p, st->wcount, (__bf16 *)st->w.p);
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static const TrainOps *ops() {
        static const TrainOps o = {setup, fwd_bwd, pack};
        return &o;
    }
};

// (Round 5 built two more versions of the pair -- a per-wave register chain with an LDS fragment ring, 0.833 ms per 1M rows, and four
// launches at two waves per SIMD, 0.984 ms -- against 0.744 ms for the kernels above; both were correct, both lost, and both were
// removed in round 6: `git show 4981a23:baler_amd/csrc/bf16_train.hip`, profiles/r5_bf16_regchain_kernel_stats.csv,
// profiles/r5_bf16_quad_kernel_stats.csv, DESIGN_HISTORY.md section 4.6.)
template <int F, int Z> const TrainOps *pick_train(const bamd_handle *h) {
    return TImpl<F, Z>::matches(h) ? TImpl<F, Z>::ops() : nullptr;
}
const TrainOps *find_train(const bamd_handle *h) {
    const TrainOps *o = nullptr;
    if ((o = pick_train<24, 15>(h))) return o;
    if ((o = pick_train<24, 12>(h))) return o;
    if ((o = pick_train<24, 8>(h))) return o;
    if ((o = pick_train<24, 6>(h))) return o;
    if ((o = pick_train<24, 10>(h))) return o;
    if ((o = pick_train<24, 5>(h))) return o;
    if ((o = pick_train<24, 4>(h))) return o;
    if ((o = pick_train<24, 3>(h))) return o;
    if ((o = pick_train<24, 2>(h))) return o;
    return nullptr;
}

}  // namespace

#ifdef BAMD_BF16_TRACE
extern "C" int bamd_debug_bf16_trace(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bf16_trace), sizeof(unsigned long long) * 2 * 4 * 128);
}
#endif

int bf16_train_setup(bamd_handle *h) {
    const TrainOps *ops = find_train(h);
    if (!ops) return BAMD_OK;                  // no bf16 training kernels for this shape: training calls use the fp32 layer-wise path
    const char *env = getenv("BALER_AMD_BF16_TRAIN");
    if (env && env[0] == '0') return BAMD_OK;
    TrainState *st = new TrainState();
    st->ops = ops;
    h->bf16_train_state = st;
    return ops->setup(h, st);
}

void bf16_train_teardown(bamd_handle *h) {
    TrainState *st = tstate(h);
    if (!st) return;
    st->src.release(); st->w.release(); st->inv.release(); st->dz.release();
    delete st;
    h->bf16_train_state = nullptr;
}

bool bf16_train_ok(const bamd_handle *h) { return h->bf16_train_state != nullptr; }

int bf16_train_pack(bamd_handle *h, hipStream_t s) {
    TrainState *st = tstate(h);
    if (!st) return BAMD_OK;
    return st->ops->pack(h, st, s);
}

int bf16_fwd_bwd(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads, hipStream_t s) {
    TrainState *st = tstate(h);
    return st->ops->fwd_bwd(h, st, x, x_dtype, n, features, (float *)grads, s);
}

}  // namespace bamd
