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

// =====================================================================================================================
// Round 5 -- the training pair re-written: a per-wave REGISTER CHAIN through ALL layers, weight fragments through an LDS ring.
//
// What bounded the kernels above: four lock-stepped waves exchanging every wide layer through LDS -- a
// barrier, an LDS round trip and a serial VALU epilogue per layer on ONE in-order wave per SIMD (MFMA busy 27 %).  The narrow
// layers already ran as a per-wave register chain ("M-split": a wave computes ALL output tiles of ITS 16 rows, a packed C tile
// pair IS the next product's B operand); what kept the wide layers from it is the fragment stream: one 1-KiB fragment per
// 16-cycle MFMA and wave = 64 B/clk per wave, four times what a CU's vector-memory path delivers.  Here every fragment is
// fetched ONCE per workgroup, by direct-to-LDS loads (buffer_load ... lds: no registers, asynchronous), into a ring of kR
// slots of kG KiB that all four waves read (ds_read_b128, kPF reads ahead of the MFMAs) -- the L1 path carries 1 KiB per kG / 4
// ... per 4 MFMAs of the CU, the LDS one KiB per MFMA and wave.  The chain itself then has NO barrier and no LDS round trip between
// layers; the ring is kept in step by ONE workgroup barrier per kG MFMAs (all four waves run the same instruction stream on
// different rows, so they arrive together): before barrier s a wave waits (counted vmcnt, asm) for ITS quarter of slot s + 1,
// after it slot s + 1 is complete and slot s - 1 is free, and the wave requests its quarter of slot s + 3 into that place.
// Images are written (own rows, no barrier) only for what the weight-gradient phases and the LeakyReLU masks read back.
// The cut between the two launches moves to the BOTTLENECK: PART 0 = forward 0..7, loss, input-gradient products 7..4, weight
// gradients of the decoder (7..4: 149 tiles); PART 1 = forward 0..2 recomputed, dZ_3 handed over (ONE tile: 32 B per row instead
// of 224), input-gradient products 3..1, weight gradients of the encoder (3..0: 149 tiles).  Each launch then keeps only its own
// half of the images (108 / 100 KB instead of 132), which is what makes room for the 48-KB ring.
constexpr int kG = 12;        // fragments (KiB) per ring slot = MFMAs per wave between two ring barriers; a multiple of 4
constexpr int kR = 4;         // ring slots (a power of two): one being read, one complete, one in flight, one being requested
constexpr int kPF = 12;       // fragment reads ahead of the MFMAs (4 registers each); <= kG
static_assert(kG % 4 == 0 && (kR & (kR - 1)) == 0 && kPF <= kG, "ring geometry");

template <int PART> struct Cut {
    static constexpr int fwd_end = PART == 0 ? 8 : 3;        // forward layers [0, fwd_end)
    static constexpr int bwd_hi = PART == 0 ? 7 : 3;         // weight gradients of layers bwd_hi .. bwd_lo
    static constexpr int bwd_lo = PART == 0 ? 4 : 0;
    static constexpr int chain_lo = PART == 0 ? 4 : 1;       // input-gradient products of layers bwd_hi .. chain_lo
    __host__ __device__ static constexpr bool has(int l) { return l <= bwd_hi && l >= bwd_lo; }
};
// the launch's fragment stream in consumption order: forward products [q][t] of layers 0 .. fwd_end - 1, then the input-gradient
// products [q][t] of layers bwd_hi .. chain_lo
template <class N, int PART> struct Stream2 {
    using C = Cut<PART>;
    __host__ __device__ static constexpr int fo_f(int l) { int s = 0; for (int j = 0; j < l; ++j) s += N::kb(j) * N::nt(j); return s; }
    __host__ __device__ static constexpr int fo_b(int l) { int s = fo_f(C::fwd_end); for (int j = C::bwd_hi; j > l; --j) s += N::kbb(j) * N::ntb(j); return s; }
    static constexpr int nfrag = fo_b(C::chain_lo) + N::kbb(C::chain_lo) * N::ntb(C::chain_lo);
    static constexpr int nslot = cdiv(nfrag, kG);
    static_assert(nslot >= kR, "a launch's stream fills the ring");
};
// LDS image regions of a launch ([64 rows][stride] each; strides and swizzle as above).  dZ_l has the shape of X_{l+1}.
//   PART 0:  [X_7 | dZ_7] | dZ_6 | X_6 | X_5 | X_4 ;  dZ_5 and dZ_4 -> the bracket (dead after dW_7): 96 KB
//   PART 1:  X_1 | X_0 | [X_3 | dZ_3 | pad] | X_2 | dZ_2 | dZ_0 ;  dZ_1 -> the bracket (dead after dW_3): 112 KB
// Together with the 48-KB ring that is PART 1's LDS to the last byte (the per-feature min / range of normalise-on-load sit in registers).
template <class N, int PART> struct Plan2 {
    __host__ __device__ static constexpr int S(int i) { return N::istride(i); }
    __host__ __device__ static constexpr int xoff(int l) {
        if (PART == 0) return 64 * (l == 7 ? 0 : l == 6 ? 2 * S(7) + S(8) : l == 5 ? 2 * S(7) + S(8) + S(6) : 2 * S(7) + S(8) + S(6) + S(5));
        return 64 * (l == 1 ? 0 : l == 0 ? S(1) : l == 3 ? S(1) + S(0) : S(1) + S(0) + S(2));
    }
    __host__ __device__ static constexpr int zoff(int l) {
        if (PART == 0) return 64 * (l == 7 ? S(7) : l == 6 ? S(7) + S(8) : l == 5 ? 0 : S(6));
        return l == 3 ? xoff(3) + 64 * S(3) : l == 1 ? xoff(3) : l == 2 ? xoff(2) + 64 * S(2) : xoff(2) + 64 * (S(2) + S(3));
    }
    static constexpr int img_bytes = PART == 0 ? xoff(4) + 64 * S(4) : zoff(0) + 64 * S(1);
    static_assert(S(6) + S(5) <= S(7) + S(8), "dZ_5 | dZ_4 fit the region of X_7 | dZ_7");
    static_assert(S(3) + S(4) <= S(2), "X_3 | dZ_3 fit the region dZ_1 takes over");
    static constexpr int ring_off = (img_bytes + 1023) & ~1023;
    static constexpr int lds_bytes = ring_off + kR * kG * 1024;
    static_assert(lds_bytes <= 160 * 1024, "images + ring exceed one CU's LDS");
};

// one direct-to-LDS load: 64 lanes x 16 bytes = one fragment.  M0 (the LDS base) is not preserved by hipcc around asm: set and restored here
__device__ __forceinline__ void dma_1k(unsigned lds_addr, int voff, __amdgpu_buffer_rsrc_t rs, int soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
struct Ring2 {
    __amdgpu_buffer_rsrc_t rs;      // the launch's fragment stream
    unsigned lds0;                  // LDS byte address of the ring
    unsigned rot;                   // (slots requested before this iteration) mod kR
    int lane16, wave;
    bool req;                       // this wave requests fragments (the pair: every wave; the quad launches: waves 0 .. 3)
    lds_p rd[kR];                   // read base of the slot at stream position i mod kR of THIS iteration (+ lane * 16)
};
// this wave's quarter of slot NS of the stream into ring position `pos`
template <class ST, int NS> __device__ __forceinline__ void ring_request(const Ring2 &rg, unsigned pos) {
    if (!rg.req) return;            // (wave-uniform)
#pragma unroll
    for (int k = 0; k < kG / 4; ++k) {
        int fi = NS * kG + rg.wave * (kG / 4) + k;
        fi = fi < ST::nfrag ? fi : ST::nfrag - 1;                          // the tail of the last slot: a valid fragment, never read
        const unsigned dst = rg.lds0 + pos * (kG * 1024u) + (unsigned)(rg.wave * (kG / 4) + k) * 1024u;
        dma_1k(__builtin_amdgcn_readfirstlane(dst), rg.lane16, rg.rs, __builtin_amdgcn_readfirstlane(fi * 1024));
    }
}
// ring barrier S (in front of the first MFMA of slot S): afterwards slot S + 1 is complete and slot S - 1 free
template <class ST, int S> __device__ __forceinline__ void ring_barrier(const Ring2 &rg) {
    // all but my youngest requests (slot S + 2): my quarter of slot S + 1 has landed.  A bare s_barrier, not __syncthreads(): that one
    // drains lgkmcnt(0), i.e. the kPF fragment reads in flight, at every ring barrier; nothing this barrier orders needs it (the slot
    // whose place is requested next was consumed by MFMAs that have issued; image traffic has its own barriers A .. E).  The asm
    // statements keep hipcc from moving LDS reads across it.
    if (rg.req) asm volatile("s_waitcnt vmcnt(%0)" :: "i"(kG / 4) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ring_request<ST, (S + kR - 1) % ST::nslot>(rg, (rg.rot + S + kR - 1) & (kR - 1));
}
template <int GI> __device__ __forceinline__ bf8 ring_read(const Ring2 &rg) {
    return lds_b128(rg.rd[(GI / kG) & (kR - 1)] + (GI % kG) * 1024);
}
// LeakyReLU / its derivative for the register-chain pair: SCALAR v_mul_f32.  A v_pk_mul_f32 next to v_mfma_f32_16x16x32_bf16 cannot hide
// behind the MFMA at all (tools/probe/valu_beside_mfma_probe.hip, one wave per SIMD: MFMA slot 16.5 cycles, + one v_pk_mul_f32 = 33.3,
// + two independent v_mul_f32 = 17.3): the packed form that wins in the serial epilogues of the first pair costs 17 cycles per
// instruction here.
__device__ __forceinline__ void lrelu4s(v4 &a) {
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = __builtin_elementwise_maximum(a[r], a[r] * 0.01f);
}
__device__ __forceinline__ u2 lrelu_bwd_pack4s(const v4 &d, u2 y) {
    v4 m;
#pragma unroll
    for (int r = 0; r < 4; ++r) m[r] = d[r] * 0.01f;
    const u2 p1 = pack4(d), p2 = pack4(m);
    unsigned sh = 0x000F000Fu;
    u2 o;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        unsigned mask;
        asm("v_pk_ashrrev_i16 %0, %1, %2" : "=v"(mask) : "v"(sh), "v"(y[h]));      // (through asm: see lrelu_bwd_pack4)
        o[h] = (p2[h] & mask) | (p1[h] & ~mask);
    }
    return o;
}
// ---- a product of the register chain, software-pipelined -------------------------------------------------------------------------
// Measured on gfx950, ONE wave per SIMD (tools/probe/valu_beside_mfma_probe.hip): a v_mfma_f32_16x16x32_bf16 slot is 16.5 cycles; up
// to two INDEPENDENT VALU instructions issue beside it for free (17.3), further ones cost 4.4 cycles each, a DEPENDENT one ~8, a
// v_pk_mul_f32 17 (it does not overlap the MFMA at all), a ds_write_b64 ~20.  An epilogue placed BEHIND its product (the first
// version of this pair, and the first pair's N-split phases) therefore ADDS its ~10 VALU per tile to the MFMA time: 75 cycles per tile,
// more than the whole product of a narrow layer.  Here the output tiles of a product are computed GROUP BY GROUP (kGS tiles, all k
// blocks of a group before the next group: fragment order [group][k block][tile]), so that a group's tiles are final while the next
// group's MFMAs issue, and the scheduler is told to deal the finished group's epilogue between those MFMAs (sched_group_barrier:
// 1 MFMA, VPM VALU, LDS reads / writes).  The LAST group of a product is finished beside the first MFMAs of the next product
// (`carry`), or in front of a workgroup barrier where one separates the two.
constexpr int kGS = 4;        // output tiles per group
template <int V> using IC = std::integral_constant<int, V>;
template <int NT> struct Grp {
    static constexpr int NG = cdiv(NT, kGS);
    __host__ __device__ static constexpr int size(int g) { return g < NG - 1 ? kGS : NT - kGS * (NG - 1); }
    // fragment (group g, k block q, tile j of the group) of a product with KB k blocks: KB kGS g + q size(g) + j
    __host__ __device__ static constexpr int frag(int KB, int g, int q, int j) { return KB * kGS * g + q * size(g) + j; }
};
template <class ST, int GI, int VPM, bool FIRST>
__device__ __forceinline__ void mfma_one(v4 &acc, const bf8 &b, const Ring2 &rg, bf8 (&fr)[kPF]) {
    if constexpr (GI % kG == 0) {
        __builtin_amdgcn_sched_barrier(0);
        ring_barrier<ST, GI / kG>(rg);
        __builtin_amdgcn_sched_barrier(0);
    }
    const v4 zero = (v4){0.f, 0.f, 0.f, 0.f};
    acc = mfma(fr[GI % kPF], b, FIRST ? zero : acc);
    if constexpr (GI + kPF < ST::nfrag) fr[GI % kPF] = ring_read<GI + kPF>(rg);
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);          // this MFMA
    __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);        // its share of the VALU work of the region
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);          // its fragment read (+ a mask read)
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);          // an image store, if one is ready
}
template <class ST, int GI0, int KB, int NT, int VPM, int TG, int Q, int... J>
__device__ __forceinline__ void pstepq(v4 (&acc)[NT], const bf8 (&b)[KB], const Ring2 &rg, bf8 (&fr)[kPF], std::integer_sequence<int, J...>) {
    (mfma_one<ST, GI0 + Grp<NT>::frag(KB, TG, Q, J), VPM, Q == 0>(acc[kGS * TG + J], b[Q], rg, fr), ...);
}
template <class ST, int GI0, int KB, int NT, int VPM, int TG, int... Q>
__device__ __forceinline__ void pallq(v4 (&acc)[NT], const bf8 (&b)[KB], const Ring2 &rg, bf8 (&fr)[kPF], std::integer_sequence<int, Q...>) {
    (pstepq<ST, GI0, KB, NT, VPM, TG, Q>(acc, b, rg, fr, std::make_integer_sequence<int, Grp<NT>::size(TG)>{}), ...);
}
template <int NT, int G_, class Fn> __device__ __forceinline__ void for_group(Fn &fn) {      // fn(tile) for the tiles of group G_
    auto f = [&](auto jc) { if constexpr (decltype(jc)::value < Grp<NT>::size(G_)) fn(IC<kGS * G_ + decltype(jc)::value>{}); };
    f(IC<0>{}); f(IC<1>{}); f(IC<2>{}); f(IC<3>{});
    static_assert(kGS == 4, "group size");
}
// region TG of a product: request what the epilogue of group TG will read (pre), finish group TG - 1 (VT VALU per tile), MFMAs of group TG
template <class ST, int GI0, int KB, int NT, int VT, int TG, int CV, class Fin, class Pre>
__device__ __forceinline__ void pregion(v4 (&acc)[NT], const bf8 (&b)[KB], Fin &fin, Pre &pre, const Ring2 &rg, bf8 (&fr)[kPF]) {
    using G = Grp<NT>;
    constexpr int SZ = G::size(TG);
    constexpr int work = TG > 0 ? VT * G::size(TG > 0 ? TG - 1 : 0) : CV;         // VALU instructions to hide in this region
    constexpr int VPM = (work + KB * SZ - 1) / (KB * SZ) > 0 ? (work + KB * SZ - 1) / (KB * SZ) : 1;
    for_group<NT, TG>(pre);
    if constexpr (TG > 0) for_group<NT, (TG > 0 ? TG - 1 : 0)>(fin);
    pallq<ST, GI0, KB, NT, VPM, TG>(acc, b, rg, fr, std::make_integer_sequence<int, KB>{});
    __builtin_amdgcn_sched_barrier(0);
}
template <class ST, int GI0, int KB, int NT, int VT, int CV, class Fin, class Pre, int... TG>
__device__ __forceinline__ void pregions(v4 (&acc)[NT], const bf8 (&b)[KB], Fin &fin, Pre &pre, const Ring2 &rg, bf8 (&fr)[kPF],
                                         std::integer_sequence<int, TG...>) {
    (pregion<ST, GI0, KB, NT, VT, TG, CV>(acc, b, fin, pre, rg, fr), ...);
}
// acc[NT] = product over KB k blocks of the B operands made of the previous product's packed tiles pkp[NTP]; `carry` finishes what
// is still unfinished of pkp (CV VALU instructions, hidden beside the first group's MFMAs); fin(tile) finishes a tile of THIS product
// (all groups but the last: the caller's next carry does that one); pre(tile) requests what fin(tile) will need from LDS
template <class ST, int GI0, int KB, int NT, int NTP, int VT, int CV, class Fin, class Pre, class Carry>
__device__ __forceinline__ void mprod(v4 (&acc)[NT], const u2 (&pkp)[NTP], Fin &fin, Pre &pre, Carry &carry, const Ring2 &rg, bf8 (&fr)[kPF]) {
    carry();
    bf8 b[KB];
    regfeed<KB, NTP>(b, pkp);
    pregions<ST, GI0, KB, NT, VT, CV>(acc, b, fin, pre, rg, fr, std::make_integer_sequence<int, Grp<NT>::NG>{});
}
// the tiles of the LAST group of a product with NT tiles: what the next carry (or the code in front of a barrier) finishes
template <int NT, class Fin> __device__ __forceinline__ void finish_last(Fin &fin) { for_group<NT, Grp<NT>::NG - 1>(fin); }

// input rows in C-TILE layout: lane (j, g) holds features 16 t + 4 g + r (t = 0, 1; r = 0..3) of row j of its wave's 16 rows --
// the registers of two C tiles, i.e. (after the conversion) the B operand of layer 0 in the register chain's k order AND the
// values the loss compares the reconstruction with: no LDS copy of x
struct RawX2 { double d[8]; };
template <int F>
__device__ __forceinline__ void x_issue2(RawX2 &raw, const void *x, int is_f64, int64_t row, int64_t n, int g) {
    const int64_t r = row < n ? row : 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int f0 = 16 * t + 4 * g;
        const int64_t base = r * F + (f0 < F ? f0 : 0);
        if (is_f64) {
            const double2 *p = (const double2 *)((const double *)x + base);
            const double2 a = p[0], b = p[1];
            raw.d[4 * t] = a.x; raw.d[4 * t + 1] = a.y; raw.d[4 * t + 2] = b.x; raw.d[4 * t + 3] = b.y;
        } else {
            const float4 a = *(const float4 *)((const float *)x + base);
            raw.d[4 * t] = a.x; raw.d[4 * t + 1] = a.y; raw.d[4 * t + 2] = a.z; raw.d[4 * t + 3] = a.w;
        }
    }
}

template <int F, int Z, int PART>
__global__ void __launch_bounds__(256) bf16_train2_kernel(const uint4 *__restrict__ wfrags, const void *__restrict__ xin, int in_f64, int64_t n,
                                                          const double *__restrict__ feats, v4 *__restrict__ slabs,
                                                          u2 *__restrict__ dz, int loss_tile) {
    using N = TNet<F, Z>;
    using C = Cut<PART>;
    using ST = Stream2<N, PART>;
    using PL = Plan2<N, PART>;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    const lds_p img = (lds_p)lds_raw;
#ifdef BAMD_BF16_TRACE      // PART 0: stamps in LDS behind the ring (copied out at the end); PART 1 has no LDS left: straight to global memory
    unsigned long long *bt_lds = PART == 0 ? (unsigned long long *)(lds_raw + PL::lds_bytes) : &g_bf16_trace[1][0][0];
#define BT2(i) do { if ((threadIdx.x & 63) == 0 && (PART == 0 || blockIdx.x == 0)) bt_lds[(threadIdx.x >> 6) * (PART == 0 ? 64 : 128) + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BT2(i) do {} while (0)
#endif
    for (int i = threadIdx.x; i < PL::img_bytes / 16; i += 256) ((uint4 *)lds_raw)[i] = make_uint4(0, 0, 0, 0);   // finite padding slots
    int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ngroups = (n + kRows - 1) / kRows;
    // min / range of this lane's eight features (normalise-on-load), in registers: the launches' LDS is images + ring to the last byte
    double fmn[8], frg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int f = 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);
        fmn[e] = (feats && f < F) ? feats[f] : 0.0;
        frg[e] = (feats && f < F) ? feats[F + f] : 1.0;
    }
    Ring2 rg;
    rg.rs = __builtin_amdgcn_make_buffer_rsrc((void *)wfrags, 0, ST::nfrag * 1024, 0x00020000);
    rg.lds0 = (unsigned)(size_t)(img + PL::ring_off);
    rg.rot = 0;
    rg.lane16 = lane * 16;
    rg.wave = wave;
    rg.req = true;
    v4 *slab = slabs + (int64_t)blockIdx.x * 64;

    v4 g7[C::has(7) ? N::dwn(7) : 1], g6[C::has(6) ? N::dwn(6) : 1], g5[C::has(5) ? N::dwn(5) : 1], g4[C::has(4) ? N::dwn(4) : 1];
    v4 g3[C::has(3) ? N::dwn(3) : 1], g2[C::has(2) ? N::dwn(2) : 1], g1[C::has(1) ? N::dwn(1) : 1], g0[C::has(0) ? N::dwn(0) : 1];
    zero_acc(g7); zero_acc(g6); zero_acc(g5); zero_acc(g4); zero_acc(g3); zero_acc(g2); zero_acc(g1); zero_acc(g0);
    double lacc = 0.0;
    // the ring's first slots; slot 0 (and the images' zeros) must be there before the first fragment reads
    static_assert(kR == 4, "prologue requests slots 0 .. 2");
    ring_request<ST, 0>(rg, 0);
    ring_request<ST, 1 % ST::nslot>(rg, 1);
    ring_request<ST, 2 % ST::nslot>(rg, 2);
    RawX2 xraw;
    x_issue2<F>(xraw, xin, in_f64, (int64_t)blockIdx.x * kRows + 16 * wave + (lane & 15), n, lane >> 4);
    u2 hand = (u2){0u, 0u};
    if constexpr (PART == 1) hand = dz[((int64_t)blockIdx.x * kRows + 16 * wave + (lane & 15)) * 4 + (lane >> 4)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        // keep the LDS address arithmetic inside the loop (LICM would hoist hundreds of registers)
        asm volatile("" : "+v"(rg.lane16), "+s"(wave), "+v"(lane));
        BT2(0);
        const int j = lane & 15, g = lane >> 4;
        Lays ls;
        ls.s1 = make_lay<64>(lane); ls.s3 = make_lay<192>(lane); ls.s5 = make_lay<320>(lane); ls.s7 = make_lay<448>(lane);
#pragma unroll
        for (int i = 0; i < kR; ++i) rg.rd[i] = img + PL::ring_off + ((rg.rot + i) & (kR - 1)) * (kG * 1024) + 16 * lane;
        const int64_t row = grp * kRows + 16 * wave + j;
        const bool valid = row < n;
        bf8 fr[kPF];
#pragma unroll
        for (int i = 0; i < kPF; ++i) fr[i] = lds_b128(rg.rd[0] + i * 1024);       // slot 0: complete since the last ring barrier of the previous iteration
        // ---- input rows: normalise, fp32 values in C-tile layout (kept for the loss), bf16 B operand of layer 0 ------------------
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int f = 16 * (e >> 2) + 4 * g + (e & 3);
            double d = xraw.d[e];
            if (feats) d = (d - fmn[e]) / frg[e];
            v[e] = f < F ? (float)d : (f == F ? 1.0f : 0.f);                     // slot F = the ones column
        }
        if constexpr (C::has(0)) {
            const Lay &l0 = lay_of<N::istride(0)>(ls);
            const lds_p ob = img + PL::xoff(0) + 16 * wave * N::istride(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) lds_w64(ob + l0.wr(t & 1) + 32 * (t & ~1), pack4((v4){v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]}));
        }
        BT2(1);

        // epilogue of forward tile t of layer l into pkv[t]: [LeakyReLU] -> bf16 (-> image l + 1, own rows, when this launch reads it back)
#define BAMD3_FIN_F(l, accv, pkv)                                                                                            \
        [&](auto tc) {                                                                                                       \
            constexpr int t = decltype(tc)::value;                                                                           \
            v4 a = accv[t];                                                                                                  \
            if (N::act(l)) lrelu4s(a);                                                                                       \
            pkv[t] = pack4(a);                                                                                               \
            if constexpr (C::has((l) + 1))                                                                                   \
                lds_w64(img + PL::xoff((l) + 1) + 16 * wave * N::istride((l) + 1) + lay_of<N::istride((l) + 1)>(ls).wr(t & 1) + 32 * (t & ~1), pkv[t]); \
        }
        // epilogue of tile t of the input-gradient product of layer l (= dZ_{l-1}) into pkv[t]: mask with the sign of X_l (own rows,
        // requested a region ahead into yv[t]) where layer l - 1 has an activation -> bf16 (-> dZ_{l-1}'s region when this launch
        // computes that weight gradient)
#define BAMD3_FIN_B(l, accv, yv, pkv)                                                                                        \
        [&](auto tc) {                                                                                                       \
            constexpr int t = decltype(tc)::value;                                                                           \
            if constexpr (N::act((l) - 1)) pkv[t] = lrelu_bwd_pack4s(accv[t], yv[t]);                                        \
            else pkv[t] = pack4(accv[t]);                                                                                    \
            if constexpr ((l) - 1 >= C::bwd_lo)                                                                              \
                lds_w64(img + PL::zoff((l) - 1 >= C::bwd_lo ? (l) - 1 : C::bwd_lo) + 16 * wave * N::istride(l) + lay_of<N::istride(l)>(ls).wr(t & 1) + 32 * (t & ~1), pkv[t]); \
        }
        // sign mask of tile t of X_l (own rows) for the input-gradient product of layer l
#define BAMD3_PRE_B(l, yv)                                                                                                   \
        [&](auto tc) {                                                                                                       \
            constexpr int t = decltype(tc)::value;                                                                           \
            if constexpr (N::act((l) - 1))                                                                                   \
                yv[t] = lds_b64(img + PL::xoff(l) + 16 * wave * N::istride(l) + lay_of<N::istride(l)>(ls).wr(t & 1) + 32 * (t & ~1)); \
        }
#define BAMD2_DW(l, G) dw_phase<N, l, N::istride((l) + 1), N::istride(l)>(G, img + PL::zoff(l), img + PL::xoff(l), lay_of<N::istride((l) + 1)>(ls), \
                                                                         lay_of<N::istride(l)>(ls), wave);
        constexpr int VF = 10, VB = 12, VN = 2;          // VALU instructions per tile: forward epilogue, masked backward epilogue, conversion only
        auto nopre = [&](auto) {};
        auto nocarry = [&]() {};
        u2 pk0[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) pk0[t] = pack4((v4){v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]});
        // forward 0 .. 2 (both launches)
        v4 a1[N::nt(0)], a2[N::nt(1)], a3[N::nt(2)];
        u2 p1[N::nt(0)], p2[N::nt(1)], p3[N::nt(2)];
        auto fin0 = BAMD3_FIN_F(0, a1, p1);
        auto fin1 = BAMD3_FIN_F(1, a2, p2);
        auto fin2 = BAMD3_FIN_F(2, a3, p3);
        auto carry1 = [&]() { finish_last<N::nt(0)>(fin0); };
        auto carry2 = [&]() { finish_last<N::nt(1)>(fin1); };
        mprod<ST, ST::fo_f(0), N::kb(0), N::nt(0), 2, VF, 0>(a1, pk0, fin0, nopre, nocarry, rg, fr);
        BT2(2);
        mprod<ST, ST::fo_f(1), N::kb(1), N::nt(1), N::nt(0), VF, VF * Grp<N::nt(0)>::size(Grp<N::nt(0)>::NG - 1)>(a2, p1, fin1, nopre, carry1, rg, fr);
        BT2(3);
        mprod<ST, ST::fo_f(2), N::kb(2), N::nt(2), N::nt(1), VF, VF * Grp<N::nt(1)>::size(Grp<N::nt(1)>::NG - 1)>(a3, p2, fin2, nopre, carry2, rg, fr);
        BT2(4);
        if constexpr (PART == 0) {
            static_assert(N::nt(7) == 2 && N::nt(3) == 1 && N::ntb(4) == 1, "the reconstruction is two tiles, the latent one");
            v4 a4[N::nt(3)], a5[N::nt(4)], a6[N::nt(5)], a7[N::nt(6)], rec[2];
            u2 p4[N::nt(3)], p5[N::nt(4)], p6[N::nt(5)], p7[N::nt(6)], d7[2];
            auto fin3 = BAMD3_FIN_F(3, a4, p4);
            auto fin4 = BAMD3_FIN_F(4, a5, p5);
            auto fin5 = BAMD3_FIN_F(5, a6, p6);
            auto fin6 = BAMD3_FIN_F(6, a7, p7);
            // loss: both output tiles of this wave's 16 rows against the fp32 input values it kept; dL/drecon = 2 (r - x) / C (utils.py:195-199)
            auto fin7 = [&](auto tc) {
                constexpr int t = decltype(tc)::value;
                v4 d;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = rec[t][r] - v[4 * t + r];
                    const bool live = valid && 16 * t + 4 * g + r < F;
                    if (live) lacc += (double)e * (double)e;
                    d[r] = live ? e * (2.0f / (float)F) : 0.f;
                }
                d7[t] = pack4(d);
                lds_w64(img + PL::zoff(7) + 16 * wave * N::istride(8) + lay_of<N::istride(8)>(ls).wr(t & 1) + 32 * (t & ~1), d7[t]);
            };
            auto carry3 = [&]() { finish_last<N::nt(2)>(fin2); };
            auto carry4 = [&]() { finish_last<N::nt(3)>(fin3); };
            auto carry5 = [&]() { finish_last<N::nt(4)>(fin4); };
            auto carry6 = [&]() { finish_last<N::nt(5)>(fin5); };
            auto carry7 = [&]() { finish_last<N::nt(6)>(fin6); };
#define BAMD3_LASTV(nt_, vt) ((vt) * Grp<nt_>::size(Grp<nt_>::NG - 1))
            mprod<ST, ST::fo_f(3), N::kb(3), N::nt(3), N::nt(2), VN, BAMD3_LASTV(N::nt(2), VF)>(a4, p3, fin3, nopre, carry3, rg, fr);
            BT2(5);
            mprod<ST, ST::fo_f(4), N::kb(4), N::nt(4), N::nt(3), VF, BAMD3_LASTV(N::nt(3), VN)>(a5, p4, fin4, nopre, carry4, rg, fr);
            BT2(6);
            mprod<ST, ST::fo_f(5), N::kb(5), N::nt(5), N::nt(4), VF, BAMD3_LASTV(N::nt(4), VF)>(a6, p5, fin5, nopre, carry5, rg, fr);
            BT2(7);
            mprod<ST, ST::fo_f(6), N::kb(6), N::nt(6), N::nt(5), VF, BAMD3_LASTV(N::nt(5), VF)>(a7, p6, fin6, nopre, carry6, rg, fr);
            BT2(8);
            mprod<ST, ST::fo_f(7), N::kb(7), 2, N::nt(6), 30, BAMD3_LASTV(N::nt(6), VF)>(rec, p7, fin7, nopre, carry7, rg, fr);
            BT2(9);
            finish_last<2>(fin7);
            BT2(10);
            __syncthreads();                                                    // A: dZ_7 and X_4 .. X_7 of all 64 rows
            BT2(11);
            // input-gradient products 7 .. 4: dZ_6 .. dZ_3
            v4 e6[N::ntb(7)], e5[N::ntb(6)], e4[N::ntb(5)], e3[1];
            u2 q6[N::ntb(7)], q5[N::ntb(6)], q4[N::ntb(5)], q3[1], y7[N::ntb(7)], y6[N::ntb(6)], y5[N::ntb(5)], yz[1];
            auto finb7 = BAMD3_FIN_B(7, e6, y7, q6);
            auto finb6 = BAMD3_FIN_B(6, e5, y6, q5);
            auto finb5 = BAMD3_FIN_B(5, e4, y5, q4);
            auto finb4 = BAMD3_FIN_B(4, e3, yz, q3);
            auto pre7 = BAMD3_PRE_B(7, y7);
            auto pre6 = BAMD3_PRE_B(6, y6);
            auto pre5 = BAMD3_PRE_B(5, y5);
            auto carryb6 = [&]() { finish_last<N::ntb(7)>(finb7); };
            auto carryb5 = [&]() { finish_last<N::ntb(6)>(finb6); };
            auto carryb4 = [&]() { finish_last<N::ntb(5)>(finb5); };
            mprod<ST, ST::fo_b(7), N::kbb(7), N::ntb(7), 2, VB, 0>(e6, d7, finb7, pre7, nocarry, rg, fr);
            BT2(12);
            BAMD2_DW(7, g7)
            BT2(13);
            __syncthreads();                                                    // B: X_7 | dZ_7 dead (dZ_5, dZ_4 go there)
            BT2(14);
            mprod<ST, ST::fo_b(6), N::kbb(6), N::ntb(6), N::ntb(7), VB, BAMD3_LASTV(N::ntb(7), VB)>(e5, q6, finb6, pre6, carryb6, rg, fr);
            BT2(15);
            mprod<ST, ST::fo_b(5), N::kbb(5), N::ntb(5), N::ntb(6), VB, BAMD3_LASTV(N::ntb(6), VB)>(e4, q5, finb5, pre5, carryb5, rg, fr);
            BT2(16);
            mprod<ST, ST::fo_b(4), N::kbb(4), 1, N::ntb(5), VN, BAMD3_LASTV(N::ntb(5), VB)>(e3, q4, finb4, nopre, carryb4, rg, fr);
            finish_last<1>(finb4);
            // hand-off to the second launch: dZ_3 (ONE tile), 8 bytes per lane: [row][g]; rows beyond n carry zeros
            dz[row * 4 + g] = q3[0];
            // the NEXT iteration's rows, requested behind the last ring wait of this one: the weight-gradient phases below give the
            // HBM fetch its time (loads of a wave retire in order: requested at the top it would stand in front of every ring wait)
            x_issue2<F>(xraw, xin, in_f64, row + (int64_t)gridDim.x * kRows, n, g);
            BT2(17);
            __syncthreads();                                                    // D: dZ_6, dZ_5, dZ_4 of all 64 rows
            BT2(18);
            BAMD2_DW(6, g6)
            BT2(19);
            BAMD2_DW(5, g5)
            BT2(20);
            BAMD2_DW(4, g4)
            BT2(21);
            __syncthreads();                                                    // E: the next forward overwrites X_4 .. X_7
            BT2(22);
        } else {
            static_assert(N::ntb(4) == 1 && N::nt(3) == 1, "the latent is one tile");
            finish_last<N::nt(2)>(fin2);                                        // X_3 (a single group: nothing of it is finished yet)
            static_assert(Grp<N::nt(2)>::NG == 1, "forward 2 is one group");
            // dZ_3 of these rows from the first launch -> its image (own rows) and the B operand of the first input-gradient product
            lds_w64(img + PL::zoff(3) + 16 * wave * N::istride(4) + lay_of<N::istride(4)>(ls).wr(0), hand);
            u2 q3[1] = {hand};
            BT2(5);
            __syncthreads();                                                    // A: X_0 .. X_3 and dZ_3 of all 64 rows
            BT2(6);
            v4 e2[N::ntb(3)], e1[N::ntb(2)], e0[N::ntb(1)];
            u2 q2[N::ntb(3)], q1[N::ntb(2)], q0[N::ntb(1)], y3[N::ntb(3)], y2[N::ntb(2)], y1[N::ntb(1)];
            auto finb3 = BAMD3_FIN_B(3, e2, y3, q2);
            auto finb2 = BAMD3_FIN_B(2, e1, y2, q1);
            auto finb1 = BAMD3_FIN_B(1, e0, y1, q0);
            auto pre3 = BAMD3_PRE_B(3, y3);
            auto pre2 = BAMD3_PRE_B(2, y2);
            auto pre1 = BAMD3_PRE_B(1, y1);
            auto carryb2 = [&]() { finish_last<N::ntb(3)>(finb3); };
            auto carryb1 = [&]() { finish_last<N::ntb(2)>(finb2); };
#define BAMD3_LASTV(nt_, vt) ((vt) * Grp<nt_>::size(Grp<nt_>::NG - 1))
            mprod<ST, ST::fo_b(3), N::kbb(3), N::ntb(3), 1, VB, 0>(e2, q3, finb3, pre3, nocarry, rg, fr);
            BT2(7);
            BAMD2_DW(3, g3)
            BT2(8);
            __syncthreads();                                                    // B: X_3 | dZ_3 dead (dZ_1 goes there)
            BT2(9);
            mprod<ST, ST::fo_b(2), N::kbb(2), N::ntb(2), N::ntb(3), VB, BAMD3_LASTV(N::ntb(3), VB)>(e1, q2, finb2, pre2, carryb2, rg, fr);
            BT2(10);
            mprod<ST, ST::fo_b(1), N::kbb(1), N::ntb(1), N::ntb(2), VB, BAMD3_LASTV(N::ntb(2), VB)>(e0, q1, finb1, pre1, carryb1, rg, fr);
            finish_last<N::ntb(1)>(finb1);
            (void)q0;
            x_issue2<F>(xraw, xin, in_f64, row + (int64_t)gridDim.x * kRows, n, g);
            {
                const int64_t nr = row + (int64_t)gridDim.x * kRows;
                hand = dz[(nr < ngroups * kRows ? nr : row) * 4 + g];
            }
            BT2(11);
            __syncthreads();                                                    // D: dZ_2, dZ_1, dZ_0 of all 64 rows
            BT2(12);
            BAMD2_DW(2, g2)
            BT2(13);
            BAMD2_DW(1, g1)
            BT2(14);
            BAMD2_DW(0, g0)
            BT2(15);
            __syncthreads();                                                    // E
            BT2(16);
        }
#undef BAMD3_LASTV
#undef BAMD3_FIN_F
#undef BAMD3_FIN_B
#undef BAMD3_PRE_B
#undef BAMD2_DW
        rg.rot = (rg.rot + ST::nslot) & (kR - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // nothing may land in LDS after the workgroup has gone
#ifdef BAMD_BF16_TRACE
    __syncthreads();
    if (PART == 0 && blockIdx.x == 0 && threadIdx.x < 256) g_bf16_trace[0][threadIdx.x >> 6][threadIdx.x & 63] = bt_lds[threadIdx.x];
#endif
#undef BT2
    if constexpr (C::has(7)) dw_flush<N, 7>(slab, g7, lane, wave);
    if constexpr (C::has(6)) dw_flush<N, 6>(slab, g6, lane, wave);
    if constexpr (C::has(5)) dw_flush<N, 5>(slab, g5, lane, wave);
    if constexpr (C::has(4)) dw_flush<N, 4>(slab, g4, lane, wave);
    if constexpr (C::has(3)) dw_flush<N, 3>(slab, g3, lane, wave);
    if constexpr (C::has(2)) dw_flush<N, 2>(slab, g2, lane, wave);
    if constexpr (C::has(1)) dw_flush<N, 1>(slab, g1, lane, wave);
    if constexpr (C::has(0)) dw_flush<N, 0>(slab, g0, lane, wave);
    if constexpr (PART == 0) {   // per-workgroup loss partial (fixed-order tree), stored after the tiles
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

// =====================================================================================================================
// Round 5, second rewrite ("quad"): FOUR launches of TWO weight-gradient layers each, EIGHT waves per workgroup (two per SIMD).
//
// The pair above shows what ONE in-order wave per SIMD cannot overlap (a KiB of fragment from LDS, ~2.6 VALU instructions and the MFMA
// itself per MFMA slot); a second wave per SIMD can, but 128 rows of the pair's images do not fit the LDS and its resident
// weight-gradient tiles not the 256 registers a wave then has.  Cutting the backward pass into four launches {7, 6} {5, 4} {3, 2} {1, 0}
// makes both fit: a launch keeps X_hi, dZ_hi and X_lo only (dZ_lo is written IN PLACE over X_hi once dW_hi is done: the mask is read
// from the very slot the gradient goes to) -- 832 / 576 / 576 / 832 B per row = 104 KB for 128 rows next to the 48-KB ring -- and
// 117 / 32 / 32 / 117 tiles over eight waves (<= 17 accumulator tiles per wave).  Every launch recomputes the forward chain up to its
// layers (8 / 5 / 3 / 1 products) from the rows and takes dZ_hi from the previous launch (224 / 32 / 224 B per row): 489 instead of
// 327 chain MFMAs per 16 rows and four passes over the rows -- the price of the second wave.
// Per iteration (128 rows, a wave = 16 rows through the whole chain in registers, as in the pair):
//   rows -> forward products (X_hi, X_lo -> images, own rows) -> dZ_hi -> image (loss, or the hand-off record)
//   barrier A -> dW_hi -> barrier B -> product hi (dZ_lo over X_hi, own rows) -> product lo (dZ_{lo-1} -> hand-off record)
//   -> next rows / record requested -> barrier D -> dW_lo -> barrier E.
// The ring is the pair's (kG fragments per slot, one workgroup barrier per slot), requested by waves 0..3 (a quarter each).
constexpr int kRows3 = 128;
template <int PART> struct Cut3 {
    static constexpr int fwd_end = PART == 0 ? 8 : PART == 1 ? 5 : PART == 2 ? 3 : 1;     // forward layers [0, fwd_end)
    static constexpr int bwd_hi = 7 - 2 * PART, bwd_lo = 6 - 2 * PART;                    // weight gradients of these two layers
    static constexpr int chain_lo = PART == 3 ? 1 : bwd_lo;                               // input-gradient products of layers bwd_hi .. chain_lo
    __host__ __device__ static constexpr bool has(int l) { return l == bwd_hi || l == bwd_lo; }
};
template <class N, int PART> struct Stream3 {
    using C = Cut3<PART>;
    __host__ __device__ static constexpr int fo_f(int l) { int s = 0; for (int j = 0; j < l; ++j) s += N::kb(j) * N::nt(j); return s; }
    __host__ __device__ static constexpr int fo_b(int l) { int s = fo_f(C::fwd_end); for (int j = C::bwd_hi; j > l; --j) s += N::kbb(j) * N::ntb(j); return s; }
    static constexpr int nfrag = fo_b(C::chain_lo) + N::kbb(C::chain_lo) * N::ntb(C::chain_lo);
    static constexpr int nslot = cdiv(nfrag, kG);
    static_assert(nslot >= kR, "a launch's stream fills the ring");
};
// LDS regions of a launch, 128 rows each: [X_hi, later dZ_lo] | dZ_hi | X_lo (dZ_l has the shape, hence the stride, of X_{l+1})
template <class N, int PART> struct Plan3 {
    using C = Cut3<PART>;
    __host__ __device__ static constexpr int S(int i) { return N::istride(i); }
    __host__ __device__ static constexpr int xoff(int l) { return l == C::bwd_hi ? 0 : kRows3 * (S(C::bwd_hi) + S(C::bwd_hi + 1)); }
    __host__ __device__ static constexpr int zoff(int l) { return l == C::bwd_hi ? kRows3 * S(C::bwd_hi) : 0; }
    static constexpr int img_bytes = kRows3 * (S(C::bwd_hi) + S(C::bwd_hi + 1) + S(C::bwd_lo));
    static constexpr int ring_off = (img_bytes + 1023) & ~1023;
    static constexpr int lds_bytes = ring_off + kR * kG * 1024;
    static_assert(S(C::bwd_lo + 1) == S(C::bwd_hi), "dZ_lo takes X_hi's place");
    static_assert(lds_bytes <= 160 * 1024, "images + ring exceed one CU's LDS");
};
// weight-gradient tiles of layer l over EIGHT waves and 128 rows (four 32-row contractions per tile): as DwGeo / dw_phase above
template <class N, int l> struct DwGeo8 {
    static constexpr int NT = N::nt(l), KT = N::kt(l);
    static constexpr bool BYN = cdiv(NT, 8) * KT <= cdiv(KT, 8) * NT;
    static constexpr int NO = BYN ? cdiv(NT, 8) : cdiv(KT, 8);      // owned slots
    static constexpr int OWN = BYN ? NT : KT;                       // tiles on the owned side
    static constexpr int NS = BYN ? KT : NT;                        // streamed tiles
    static constexpr int NACC = NO * NS;
};
constexpr int kDWD8 = 1;
template <class N, int l> using Acc8 = v4[DwGeo8<N, l>::NACC];
template <class N, int l> using Own8 = bf8[DwGeo8<N, l>::NO][4];
template <class N, int l, int SZ, int SX, int S>
__device__ __forceinline__ void dw8_step(Acc8<N, l> &acc, const Own8<N, l> &own, bf8 (&ring)[kDWD8 + 1][4],
                                         lds_p sbase0, lds_p sbase1) {
    using G = DwGeo8<N, l>;
    constexpr int SS = G::BYN ? SX : SZ;          // stride of the streamed image
    if constexpr (S + kDWD8 < G::NS) {
        constexpr int t = S + kDWD8;
        const lds_p sb = ((t & 1) ? sbase1 : sbase0) + 32 * (t & ~1);
#pragma unroll
        for (int h = 0; h < 4; ++h) ring[t % (kDWD8 + 1)][h] = tr_operand<SS>(sb, h);
    }
    __builtin_amdgcn_sched_barrier(0);
    const bf8 (&st)[4] = ring[S % (kDWD8 + 1)];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int i = 0; i < G::NO; ++i) {
            v4 &c = acc[i * G::NS + S];
            c = G::BYN ? mfma(own[i][h], st[h], c) : mfma(st[h], own[i][h], c);      // A = dZ^T tile, B = [X | 1] tile
        }
    __builtin_amdgcn_sched_barrier(0);
}
template <class N, int l, int SZ, int SX, int... S>
__device__ __forceinline__ void dw8_phase_impl(Acc8<N, l> &acc, lds_p zimg, lds_p ximg, const Lay &lz, const Lay &lx, int wave,
                                               std::integer_sequence<int, S...>) {
    using G = DwGeo8<N, l>;
    constexpr int SO = G::BYN ? SZ : SX, SS = G::BYN ? SX : SZ;
    const lds_p oimg = G::BYN ? zimg : ximg, simg = G::BYN ? ximg : zimg;
    const Lay &lo = G::BYN ? lz : lx, &lst = G::BYN ? lx : lz;
    bf8 own[G::NO][4];
#pragma unroll
    for (int i = 0; i < G::NO; ++i) {
        int t = wave + 8 * i;                                        // owned tile; a slot this wave does not have computes on the last
        t = t < G::OWN ? t : G::OWN - 1;                             // tile (wave-uniform, never flushed)
        const lds_p ob = oimg + lo.tr(t & 1) + 32 * (t & ~1);
#pragma unroll
        for (int h = 0; h < 4; ++h) own[i][h] = tr_operand<SO>(ob, h);
    }
    const lds_p sb0 = simg + lst.tr0, sb1 = simg + lst.tr1;
    bf8 ring[kDWD8 + 1][4];
#pragma unroll
    for (int t = 0; t < kDWD8 && t < G::NS; ++t) {
        const lds_p sb = ((t & 1) ? sb1 : sb0) + 32 * (t & ~1);
#pragma unroll
        for (int h = 0; h < 4; ++h) ring[t][h] = tr_operand<SS>(sb, h);
    }
    (dw8_step<N, l, SZ, SX, S>(acc, own, ring, sb0, sb1), ...);
}
template <class N, int l, int SZ, int SX>
__device__ __forceinline__ void dw8_phase(Acc8<N, l> &acc, lds_p zimg, lds_p ximg, const Lay &lz, const Lay &lx, int wave) {
    dw8_phase_impl<N, l, SZ, SX>(acc, zimg, ximg, lz, lx, wave, std::make_integer_sequence<int, DwGeo8<N, l>::NS>{});
}
template <class N, int l>
__device__ __forceinline__ void dw8_flush(v4 *__restrict__ slab, const Acc8<N, l> &acc, int lane, int wave) {
    using G = DwGeo8<N, l>;
    constexpr int NT = N::nt(l), KT = N::kt(l);
#pragma unroll
    for (int i = 0; i < G::NO; ++i)
#pragma unroll
        for (int s_ = 0; s_ < G::NS; ++s_) {
            const int o = wave + 8 * i;
            const int t = G::BYN ? o : s_, k = G::BYN ? s_ : o;
            if (o < G::OWN) slab[(int64_t)(N::slab_off(l) + k * NT + t) * gridDim.x * 64 + lane] = acc[i * G::NS + s_];
        }
    (void)KT;
}

template <int F, int Z, int PART>
__global__ void __launch_bounds__(512) bf16_train3_kernel(const uint4 *__restrict__ wfrags, const void *__restrict__ xin, int in_f64, int64_t n,
                                                          const double *__restrict__ feats, v4 *__restrict__ slabs,
                                                          const u2 *__restrict__ dz_in, u2 *__restrict__ dz_out, int loss_tile) {
    using N = TNet<F, Z>;
    using C = Cut3<PART>;
    using ST = Stream3<N, PART>;
    using PL = Plan3<N, PART>;
    constexpr int HI = C::bwd_hi, LO = C::bwd_lo;
    constexpr int NTH = PART == 0 ? 1 : N::nt(HI);                 // tiles of the hand-off record this launch READS (dZ_hi; launch 0: none)
    constexpr int NTO = PART == 3 ? 1 : N::ntb(LO);                // tiles of the record it WRITES (dZ_{lo-1}; launch 3: none)
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds_raw[];
    const lds_p img = (lds_p)lds_raw;
    for (int i = threadIdx.x; i < PL::img_bytes / 16; i += 512) ((uint4 *)lds_raw)[i] = make_uint4(0, 0, 0, 0);   // finite padding slots
    int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t ngroups = (n + kRows3 - 1) / kRows3;
    // min / range of this lane's eight features (normalise-on-load): a small LDS table behind the ring would do as well; registers for now
    double fmn[8], frg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int f = 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);
        fmn[e] = (feats && f < F) ? feats[f] : 0.0;
        frg[e] = (feats && f < F) ? feats[F + f] : 1.0;
    }
    Ring2 rg;
    rg.rs = __builtin_amdgcn_make_buffer_rsrc((void *)wfrags, 0, ST::nfrag * 1024, 0x00020000);
    rg.lds0 = (unsigned)(size_t)(img + PL::ring_off);
    rg.rot = 0;
    rg.lane16 = lane * 16;
    rg.wave = wave;
    rg.req = wave < 4;
    v4 *slab = slabs + (int64_t)blockIdx.x * 64;
    v4 ghi[DwGeo8<N, HI>::NACC], glo[DwGeo8<N, LO>::NACC];
    zero_acc(ghi); zero_acc(glo);
    double lacc = 0.0;
    static_assert(kR == 4, "prologue requests slots 0 .. 2");
    ring_request<ST, 0>(rg, 0);
    ring_request<ST, 1 % ST::nslot>(rg, 1);
    ring_request<ST, 2 % ST::nslot>(rg, 2);
    RawX2 xraw;
    x_issue2<F>(xraw, xin, in_f64, (int64_t)blockIdx.x * kRows3 + 16 * wave + (lane & 15), n, lane >> 4);
    u2 hand[NTH];
#pragma unroll
    for (int t = 0; t < NTH; ++t) hand[t] = (u2){0u, 0u};
    if constexpr (PART > 0) {
        const int64_t r0 = (int64_t)blockIdx.x * kRows3 + 16 * wave + (lane & 15);
#pragma unroll
        for (int t = 0; t < NTH; ++t) hand[t] = dz_in[(r0 * NTH + t) * 4 + (lane >> 4)];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        asm volatile("" : "+v"(rg.lane16), "+s"(wave), "+v"(lane));      // (keeps the LDS address arithmetic inside the loop)
        const int j = lane & 15, g = lane >> 4;
        Lays ls;
        ls.s1 = make_lay<64>(lane); ls.s3 = make_lay<192>(lane); ls.s5 = make_lay<320>(lane); ls.s7 = make_lay<448>(lane);
#pragma unroll
        for (int i = 0; i < kR; ++i) rg.rd[i] = img + PL::ring_off + ((rg.rot + i) & (kR - 1)) * (kG * 1024) + 16 * lane;
        const int64_t row = grp * kRows3 + 16 * wave + j;
        const bool valid = row < n;
        bf8 fr[kPF];
#pragma unroll
        for (int i = 0; i < kPF; ++i) fr[i] = lds_b128(rg.rd[0] + i * 1024);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int f = 16 * (e >> 2) + 4 * g + (e & 3);
            double d = xraw.d[e];
            if (feats) d = (d - fmn[e]) / frg[e];
            v[e] = f < F ? (float)d : (f == F ? 1.0f : 0.f);                     // slot F = the ones column
        }
        if constexpr (C::has(0)) {
            const Lay &l0 = lay_of<N::istride(0)>(ls);
            const lds_p ob = img + PL::xoff(0) + 16 * wave * N::istride(0);
#pragma unroll
            for (int t = 0; t < 2; ++t) lds_w64(ob + l0.wr(t & 1) + 32 * (t & ~1), pack4((v4){v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]}));
        }
#define BAMD3_FIN_F(l, accv, pkv)                                                                                            \
        [&](auto tc) {                                                                                                       \
            constexpr int t = decltype(tc)::value;                                                                           \
            v4 a = accv[t];                                                                                                  \
            if (N::act(l)) lrelu4s(a);                                                                                       \
            pkv[t] = pack4(a);                                                                                               \
            if constexpr (C::has((l) + 1))                                                                                   \
                lds_w64(img + PL::xoff((l) + 1) + 16 * wave * N::istride((l) + 1) + lay_of<N::istride((l) + 1)>(ls).wr(t & 1) + 32 * (t & ~1), pkv[t]); \
        }
        // epilogue of tile t of the input-gradient product of layer l (= dZ_{l-1}): mask with the sign of X_l (own rows, read a region
        // ahead into yv[t] -- for l = hi from the very slot the result is written to) -> bf16 (-> image when this launch has dW_{l-1})
#define BAMD3_FIN_B(l, accv, yv, pkv)                                                                                        \
        [&](auto tc) {                                                                                                       \
            constexpr int t = decltype(tc)::value;                                                                           \
            if constexpr (N::act((l) - 1)) pkv[t] = lrelu_bwd_pack4s(accv[t], yv[t]);                                        \
            else pkv[t] = pack4(accv[t]);                                                                                    \
            if constexpr (C::has((l) - 1))                                                                                   \
                lds_w64(img + PL::zoff((l) - 1) + 16 * wave * N::istride(l) + lay_of<N::istride(l)>(ls).wr(t & 1) + 32 * (t & ~1), pkv[t]); \
        }
#define BAMD3_PRE_B(l, yv)                                                                                                   \
        [&](auto tc) {                                                                                                       \
            constexpr int t = decltype(tc)::value;                                                                           \
            if constexpr (N::act((l) - 1))                                                                                   \
                yv[t] = lds_b64(img + PL::xoff(l) + 16 * wave * N::istride(l) + lay_of<N::istride(l)>(ls).wr(t & 1) + 32 * (t & ~1)); \
        }
#define BAMD3_LASTV(nt_, vt) ((vt) * Grp<nt_>::size(Grp<nt_>::NG - 1))
        constexpr int VF = 10, VB = 12, VN = 2;
        auto nopre = [&](auto) {};
        auto nocarry = [&]() {};
        u2 pk0[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) pk0[t] = pack4((v4){v[4 * t], v[4 * t + 1], v[4 * t + 2], v[4 * t + 3]});
        // ---- forward products 0 .. fwd_end - 1 (every launch from the rows); dzh = this launch's dZ_hi as packed tiles ----------------
        u2 dzh[N::nt(HI)];
        {
            v4 a1[N::nt(0)];
            u2 p1[N::nt(0)];
            auto fin0 = BAMD3_FIN_F(0, a1, p1);
            mprod<ST, ST::fo_f(0), N::kb(0), N::nt(0), 2, VF, 0>(a1, pk0, fin0, nopre, nocarry, rg, fr);
            if constexpr (C::fwd_end == 1) {
                finish_last<N::nt(0)>(fin0);
            } else {
                v4 a2[N::nt(1)], a3[N::nt(2)];
                u2 p2[N::nt(1)], p3[N::nt(2)];
                auto fin1 = BAMD3_FIN_F(1, a2, p2);
                auto fin2 = BAMD3_FIN_F(2, a3, p3);
                auto carry1 = [&]() { finish_last<N::nt(0)>(fin0); };
                auto carry2 = [&]() { finish_last<N::nt(1)>(fin1); };
                mprod<ST, ST::fo_f(1), N::kb(1), N::nt(1), N::nt(0), VF, BAMD3_LASTV(N::nt(0), VF)>(a2, p1, fin1, nopre, carry1, rg, fr);
                mprod<ST, ST::fo_f(2), N::kb(2), N::nt(2), N::nt(1), VF, BAMD3_LASTV(N::nt(1), VF)>(a3, p2, fin2, nopre, carry2, rg, fr);
                if constexpr (C::fwd_end == 3) {
                    finish_last<N::nt(2)>(fin2);
                } else {
                    v4 a4[N::nt(3)], a5[N::nt(4)];
                    u2 p4[N::nt(3)], p5[N::nt(4)];
                    auto fin3 = BAMD3_FIN_F(3, a4, p4);
                    auto fin4 = BAMD3_FIN_F(4, a5, p5);
                    auto carry3 = [&]() { finish_last<N::nt(2)>(fin2); };
                    auto carry4 = [&]() { finish_last<N::nt(3)>(fin3); };
                    mprod<ST, ST::fo_f(3), N::kb(3), N::nt(3), N::nt(2), VN, BAMD3_LASTV(N::nt(2), VF)>(a4, p3, fin3, nopre, carry3, rg, fr);
                    mprod<ST, ST::fo_f(4), N::kb(4), N::nt(4), N::nt(3), VF, BAMD3_LASTV(N::nt(3), VN)>(a5, p4, fin4, nopre, carry4, rg, fr);
                    if constexpr (C::fwd_end == 5) {
                        finish_last<N::nt(4)>(fin4);
                    } else {
                        static_assert(C::fwd_end == 8 && N::nt(7) == 2, "launch 0: the whole forward pass; the reconstruction is two tiles");
                        v4 a6[N::nt(5)], a7[N::nt(6)], rec[2];
                        u2 p6[N::nt(5)], p7[N::nt(6)];
                        auto fin5 = BAMD3_FIN_F(5, a6, p6);
                        auto fin6 = BAMD3_FIN_F(6, a7, p7);
                        // loss: both output tiles of this wave's 16 rows against the fp32 input values it kept; dL/drecon = 2 (r - x) / C
                        auto fin7 = [&](auto tc) {
                            constexpr int t = decltype(tc)::value;
                            v4 d;
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float e = rec[t][r] - v[4 * t + r];
                                const bool live = valid && 16 * t + 4 * g + r < F;
                                if (live) lacc += (double)e * (double)e;
                                d[r] = live ? e * (2.0f / (float)F) : 0.f;
                            }
                            dzh[t] = pack4(d);
                        };
                        auto carry5 = [&]() { finish_last<N::nt(4)>(fin4); };
                        auto carry6 = [&]() { finish_last<N::nt(5)>(fin5); };
                        auto carry7 = [&]() { finish_last<N::nt(6)>(fin6); };
                        mprod<ST, ST::fo_f(5), N::kb(5), N::nt(5), N::nt(4), VF, BAMD3_LASTV(N::nt(4), VF)>(a6, p5, fin5, nopre, carry5, rg, fr);
                        mprod<ST, ST::fo_f(6), N::kb(6), N::nt(6), N::nt(5), VF, BAMD3_LASTV(N::nt(5), VF)>(a7, p6, fin6, nopre, carry6, rg, fr);
                        mprod<ST, ST::fo_f(7), N::kb(7), 2, N::nt(6), 30, BAMD3_LASTV(N::nt(6), VF)>(rec, p7, fin7, nopre, carry7, rg, fr);
                        finish_last<2>(fin7);
                    }
                }
            }
        }
        if constexpr (PART > 0) {
#pragma unroll
            for (int t = 0; t < NTH; ++t) dzh[t] = hand[t];
        }
        // dZ_hi of these rows -> its image (own rows)
#pragma unroll
        for (int t = 0; t < N::nt(HI); ++t)
            lds_w64(img + PL::zoff(HI) + 16 * wave * N::istride(HI + 1) + lay_of<N::istride(HI + 1)>(ls).wr(t & 1) + 32 * (t & ~1), dzh[t]);
        __syncthreads();                                                    // A: X_hi, X_lo and dZ_hi of all 128 rows
        dw8_phase<N, HI, N::istride(HI + 1), N::istride(HI)>(ghi, img + PL::zoff(HI), img + PL::xoff(HI), lay_of<N::istride(HI + 1)>(ls),
                                                              lay_of<N::istride(HI)>(ls), wave);
        __syncthreads();                                                    // B: X_hi is dead (dZ_lo goes there)
        // ---- input-gradient products hi (-> dZ_lo, in place over X_hi) and lo (-> the hand-off record; the last launch has none) ------
        {
            v4 eh[N::ntb(HI)];
            u2 qh[N::ntb(HI)], yh[N::ntb(HI)];
            auto finbh = BAMD3_FIN_B(HI, eh, yh, qh);
            auto preh = BAMD3_PRE_B(HI, yh);
            mprod<ST, ST::fo_b(HI), N::kbb(HI), N::ntb(HI), N::nt(HI), VB, 0>(eh, dzh, finbh, preh, nocarry, rg, fr);
            if constexpr (PART == 3) {
                finish_last<N::ntb(HI)>(finbh);
            } else {
                v4 el[N::ntb(LO)];
                u2 ql[N::ntb(LO)], yl[N::ntb(LO)];
                auto finbl = BAMD3_FIN_B(LO, el, yl, ql);
                auto prel = BAMD3_PRE_B(LO, yl);
                auto carryl = [&]() { finish_last<N::ntb(HI)>(finbh); };
                mprod<ST, ST::fo_b(LO), N::kbb(LO), N::ntb(LO), N::ntb(HI), N::act(LO - 1) ? VB : VN, BAMD3_LASTV(N::ntb(HI), VB)>(el, qh, finbl, prel, carryl, rg, fr);
                finish_last<N::ntb(LO)>(finbl);
                // hand-off to the next launch: dZ_{lo-1}, [row][tile][g], 8 bytes per lane and tile; rows beyond n carry zeros
#pragma unroll
                for (int t = 0; t < NTO; ++t) dz_out[(row * NTO + t) * 4 + g] = ql[t];
            }
        }
        // the NEXT iteration's rows (and record), requested behind the last ring wait of this one
        {
            const int64_t nr = row + (int64_t)gridDim.x * kRows3;
            x_issue2<F>(xraw, xin, in_f64, nr, n, g);
            if constexpr (PART > 0) {
                const int64_t hr = nr < ngroups * kRows3 ? nr : row;
#pragma unroll
                for (int t = 0; t < NTH; ++t) hand[t] = dz_in[(hr * NTH + t) * 4 + g];
            }
        }
        __syncthreads();                                                    // D: dZ_lo of all 128 rows
        dw8_phase<N, LO, N::istride(LO + 1), N::istride(LO)>(glo, img + PL::zoff(LO), img + PL::xoff(LO), lay_of<N::istride(LO + 1)>(ls),
                                                              lay_of<N::istride(LO)>(ls), wave);
        __syncthreads();                                                    // E: the next forward pass overwrites the images
#undef BAMD3_LASTV
#undef BAMD3_FIN_F
#undef BAMD3_FIN_B
#undef BAMD3_PRE_B
        rg.rot = (rg.rot + ST::nslot) & (kR - 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // nothing may land in LDS after the workgroup has gone
    dw8_flush<N, HI>(slab, ghi, lane, wave);
    dw8_flush<N, LO>(slab, glo, lane, wave);
    if constexpr (PART == 0) {   // per-workgroup loss partial (fixed-order tree), stored after the tiles
        __syncthreads();
        double *sh = (double *)lds_raw;
        sh[threadIdx.x] = lacc;
        __syncthreads();
        for (int st = 256; st > 0; st >>= 1) {
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
    DevBuf src2[2], w2[2];          // the register-chain pair: one fragment stream per launch
    DevBuf src3[4], w3[4], dz3[3];  // the quad launches: fragment streams and the three hand-off records (dZ_5, dZ_3, dZ_1)
    int wcount3[4] = {0, 0, 0, 0};
    int wcount = 0, ntiles = 0, nparams = 0, n_features = 0;
    int wcount2[2] = {0, 0};
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

// ---- host side of the round-5 pair -------------------------------------------------------------------------------------------
template <int F, int Z> struct TImpl2 {
    using N = TNet<F, Z>;
    static bool matches(const bamd_handle *h) { return TImpl<F, Z>::matches(h); }
    static constexpr int lds2(int part) {
#ifdef BAMD_BF16_TRACE
        return part ? Plan2<N, 1>::lds_bytes : Plan2<N, 0>::lds_bytes + 2048;        // + the stamps (PART 1 stamps go to global memory)
#else
        return part ? Plan2<N, 1>::lds_bytes : Plan2<N, 0>::lds_bytes;
#endif
    }
    // fragment source map of one launch's stream.  EVERY product is fed from packed C tiles (or, layer 0, from rows loaded in that
    // layout): k slot (g, e) of k block q <-> feature 32 q + 16 (e >> 2) + 4 g + (e & 3)
    template <int PART> static void stream_map(std::vector<int> &src) {
        using C = Cut<PART>;
        using ST = Stream2<N, PART>;
        src.assign((size_t)ST::nfrag * 512, -1);
        auto kperm = [](int q, int g, int e) { return 32 * q + 16 * (e >> 2) + 4 * g + (e & 3); };
        // position of fragment (k block q, tile t) in its product's stream: tiles in groups of kGS, [group][k block][tile of the group]
        auto gfrag = [](int KB, int NT, int q, int t) {
            const int NG = (NT + kGS - 1) / kGS, gidx = t / kGS, size = gidx < NG - 1 ? kGS : NT - kGS * (NG - 1);
            return KB * kGS * gidx + q * size + (t - kGS * gidx);
        };
        for (int l = 0; l < C::fwd_end; ++l) {
            const int K = N::dim(l), NN = N::dim(l + 1);
            // forward fragment (q, t): lane (i, g) element e = [W | b | .][16 t + i][k]: input column K (the ones slot) holds the
            // bias, and padding output NN has a 1 there: it becomes the next layer's ones slot
            for (int q = 0; q < N::kb(l); ++q)
                for (int t = 0; t < N::nt(l); ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int n = 16 * t + (lane & 15), k = kperm(q, lane >> 4, e);
                            int v = -1;
                            if (n < NN && k < K) v = N::w_off(l) + n * K + k;
                            else if (n < NN && k == K) v = N::b_off(l) + n;
                            else if (n == NN && k == K) v = -2;
                            src[((size_t)(ST::fo_f(l) + gfrag(N::kb(l), N::nt(l), q, t)) * 64 + lane) * 8 + e] = v;
                        }
        }
        for (int l = C::bwd_hi; l >= C::chain_lo; --l) {
            const int K = N::dim(l), NN = N::dim(l + 1);
            // input-gradient fragment (q, t): lane (i, g) element e = W[n][16 t + i], n = the permuted k slot (an output feature)
            for (int q = 0; q < N::kbb(l); ++q)
                for (int t = 0; t < N::ntb(l); ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int n = kperm(q, lane >> 4, e), k = 16 * t + (lane & 15);
                            if (n < NN && k < K) src[((size_t)(ST::fo_b(l) + gfrag(N::kbb(l), N::ntb(l), q, t)) * 64 + lane) * 8 + e] = N::w_off(l) + n * K + k;
                        }
        }
    }
    static int setup(bamd_handle *h, TrainState *st) {
        // the gradient map, the tile count and the checks are the first version's (same weight-gradient phases, same tile layout)
        int rc = TImpl<F, Z>::setup_maps(h, st);
        if (rc) return rc;
        std::vector<int> src;
        stream_map<0>(src);
        st->wcount2[0] = (int)src.size();
        rc = st->src2[0].ensure(src.size() * sizeof(int));
        if (!rc) rc = st->w2[0].ensure(src.size() * sizeof(__bf16) + 4096);
        if (rc) return rc;
        BAMD_HIP(hipMemcpy(st->src2[0].p, src.data(), src.size() * sizeof(int), hipMemcpyHostToDevice));
        stream_map<1>(src);
        st->wcount2[1] = (int)src.size();
        rc = st->src2[1].ensure(src.size() * sizeof(int));
        if (!rc) rc = st->w2[1].ensure(src.size() * sizeof(__bf16) + 4096);
        if (rc) return rc;
        BAMD_HIP(hipMemcpy(st->src2[1].p, src.data(), src.size() * sizeof(int), hipMemcpyHostToDevice));
        BAMD_HIP(hipFuncSetAttribute((const void *)bf16_train2_kernel<F, Z, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2(0)));
        BAMD_HIP(hipFuncSetAttribute((const void *)bf16_train2_kernel<F, Z, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2(1)));
        return BAMD_OK;
    }
    static int pack(bamd_handle *h, TrainState *st, hipStream_t s) {
        for (int p = 0; p < 2; ++p)
            hipLaunchKernelGGL(pack_train_k, dim3((st->wcount2[p] + 255) / 256), dim3(256), 0, s, (const float *)h->params.p,
                               (const int *)st->src2[p].p, st->wcount2[p], (__bf16 *)st->w2[p].p);
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static int fwd_bwd(bamd_handle *h, TrainState *st, const void *x, int x_dtype, int64_t n, const double *features, float *grads,
                       hipStream_t s) {
        const int64_t ngroups = (n + kRows - 1) / kRows;
        const int grid = (int)(ngroups < st->nwg_max ? ngroups : st->nwg_max);
        int rc = h->slabs.ensure(((size_t)st->ntiles * 1024 + 16) * (size_t)grid);      // tiles + one double per workgroup for the loss
        if (rc) return rc;
        rc = st->dz.ensure((size_t)ngroups * kRows * 32);                               // dZ_3: one tile = 32 bytes per row
        if (rc) return rc;
        hipLaunchKernelGGL((bf16_train2_kernel<F, Z, 0>), dim3(grid), dim3(256), (lds2(0)), s, (const uint4 *)st->w2[0].p, x,
                           x_dtype == BAMD_F64, n, features, (v4 *)h->slabs.p, (u2 *)st->dz.p, st->ntiles);
        hipLaunchKernelGGL((bf16_train2_kernel<F, Z, 1>), dim3(grid), dim3(256), (lds2(1)), s, (const uint4 *)st->w2[1].p, x,
                           x_dtype == BAMD_F64, n, features, (v4 *)h->slabs.p, (u2 *)st->dz.p, st->ntiles);
        hipLaunchKernelGGL(reduce_tiles_k, dim3(st->ntiles + 1), dim3(256), 0, s, (const v4 *)h->slabs.p, grid, st->ntiles,
                           (const int *)st->inv.p, st->nparams, 1.0 / F, grads);
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static const TrainOps *ops() {
        static const TrainOps o = {setup, fwd_bwd, pack};
        return &o;
    }
};


// ---- host side of the quad launches -----------------------------------------------------------------------------------------
template <int F, int Z> struct TImpl3 {
    using N = TNet<F, Z>;
    static bool matches(const bamd_handle *h) { return TImpl<F, Z>::matches(h); }
    // fragment source map of one launch's stream (see TImpl2::stream_map: every product is fed from packed C tiles)
    template <int PART> static void stream_map(std::vector<int> &src) {
        using C = Cut3<PART>;
        using ST = Stream3<N, PART>;
        src.assign((size_t)ST::nfrag * 512, -1);
        auto kperm = [](int q, int g, int e) { return 32 * q + 16 * (e >> 2) + 4 * g + (e & 3); };
        auto gfrag = [](int KB, int NT, int q, int t) {
            const int NG = (NT + kGS - 1) / kGS, gidx = t / kGS, size = gidx < NG - 1 ? kGS : NT - kGS * (NG - 1);
            return KB * kGS * gidx + q * size + (t - kGS * gidx);
        };
        for (int l = 0; l < C::fwd_end; ++l) {
            const int K = N::dim(l), NN = N::dim(l + 1);
            for (int q = 0; q < N::kb(l); ++q)
                for (int t = 0; t < N::nt(l); ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int n = 16 * t + (lane & 15), k = kperm(q, lane >> 4, e);
                            int v = -1;
                            if (n < NN && k < K) v = N::w_off(l) + n * K + k;
                            else if (n < NN && k == K) v = N::b_off(l) + n;
                            else if (n == NN && k == K) v = -2;
                            src[((size_t)(ST::fo_f(l) + gfrag(N::kb(l), N::nt(l), q, t)) * 64 + lane) * 8 + e] = v;
                        }
        }
        for (int l = C::bwd_hi; l >= C::chain_lo; --l) {
            const int K = N::dim(l), NN = N::dim(l + 1);
            for (int q = 0; q < N::kbb(l); ++q)
                for (int t = 0; t < N::ntb(l); ++t)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int n = kperm(q, lane >> 4, e), k = 16 * t + (lane & 15);
                            if (n < NN && k < K) src[((size_t)(ST::fo_b(l) + gfrag(N::kbb(l), N::ntb(l), q, t)) * 64 + lane) * 8 + e] = N::w_off(l) + n * K + k;
                        }
        }
    }
    template <int PART> static int setup_part(TrainState *st) {
        std::vector<int> src;
        stream_map<PART>(src);
        st->wcount3[PART] = (int)src.size();
        int rc = st->src3[PART].ensure(src.size() * sizeof(int));
        if (!rc) rc = st->w3[PART].ensure(src.size() * sizeof(__bf16) + 4096);
        if (rc) return rc;
        BAMD_HIP(hipMemcpy(st->src3[PART].p, src.data(), src.size() * sizeof(int), hipMemcpyHostToDevice));
        BAMD_HIP(hipFuncSetAttribute((const void *)bf16_train3_kernel<F, Z, PART>, hipFuncAttributeMaxDynamicSharedMemorySize, Plan3<N, PART>::lds_bytes));
        return BAMD_OK;
    }
    static int setup(bamd_handle *h, TrainState *st) {
        int rc = TImpl<F, Z>::setup_maps(h, st);
        if (!rc) rc = setup_part<0>(st);
        if (!rc) rc = setup_part<1>(st);
        if (!rc) rc = setup_part<2>(st);
        if (!rc) rc = setup_part<3>(st);
        return rc;
    }
    static int pack(bamd_handle *h, TrainState *st, hipStream_t s) {
        for (int p = 0; p < 4; ++p)
            hipLaunchKernelGGL(pack_train_k, dim3((st->wcount3[p] + 255) / 256), dim3(256), 0, s, (const float *)h->params.p,
                               (const int *)st->src3[p].p, st->wcount3[p], (__bf16 *)st->w3[p].p);
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static int fwd_bwd(bamd_handle *h, TrainState *st, const void *x, int x_dtype, int64_t n, const double *features, float *grads,
                       hipStream_t s) {
        const int64_t ngroups = (n + kRows3 - 1) / kRows3;
        const int grid = (int)(ngroups < st->nwg_max ? ngroups : st->nwg_max);
        int rc = h->slabs.ensure(((size_t)st->ntiles * 1024 + 16) * (size_t)grid);      // tiles + one double per workgroup for the loss
        // hand-off records: dZ_5 (7 tiles), dZ_3 (1), dZ_1 (7): 32 bytes per row and tile
        if (!rc) rc = st->dz3[0].ensure((size_t)ngroups * kRows3 * 32 * N::ntb(6));
        if (!rc) rc = st->dz3[1].ensure((size_t)ngroups * kRows3 * 32 * N::ntb(4));
        if (!rc) rc = st->dz3[2].ensure((size_t)ngroups * kRows3 * 32 * N::ntb(2));
        if (rc) return rc;
        const int f64 = x_dtype == BAMD_F64;
        hipLaunchKernelGGL((bf16_train3_kernel<F, Z, 0>), dim3(grid), dim3(512), (Plan3<N, 0>::lds_bytes), s, (const uint4 *)st->w3[0].p, x, f64, n,
                           features, (v4 *)h->slabs.p, (const u2 *)nullptr, (u2 *)st->dz3[0].p, st->ntiles);
        hipLaunchKernelGGL((bf16_train3_kernel<F, Z, 1>), dim3(grid), dim3(512), (Plan3<N, 1>::lds_bytes), s, (const uint4 *)st->w3[1].p, x, f64, n,
                           features, (v4 *)h->slabs.p, (const u2 *)st->dz3[0].p, (u2 *)st->dz3[1].p, st->ntiles);
        hipLaunchKernelGGL((bf16_train3_kernel<F, Z, 2>), dim3(grid), dim3(512), (Plan3<N, 2>::lds_bytes), s, (const uint4 *)st->w3[2].p, x, f64, n,
                           features, (v4 *)h->slabs.p, (const u2 *)st->dz3[1].p, (u2 *)st->dz3[2].p, st->ntiles);
        hipLaunchKernelGGL((bf16_train3_kernel<F, Z, 3>), dim3(grid), dim3(512), (Plan3<N, 3>::lds_bytes), s, (const uint4 *)st->w3[3].p, x, f64, n,
                           features, (v4 *)h->slabs.p, (const u2 *)st->dz3[2].p, (u2 *)nullptr, st->ntiles);
        hipLaunchKernelGGL(reduce_tiles_k, dim3(st->ntiles + 1), dim3(256), 0, s, (const v4 *)h->slabs.p, grid, st->ntiles,
                           (const int *)st->inv.p, st->nparams, 1.0 / F, grads);
        BAMD_HIP(hipGetLastError());
        return BAMD_OK;
    }
    static const TrainOps *ops() {
        static const TrainOps o = {setup, fwd_bwd, pack};
        return &o;
    }
};

// BALER_AMD_BF16_TRAIN_V2=1: the round-5 register-chain pair instead of the N-split pair; =3: the quad launches (four launches, eight
// waves per workgroup) -- DESIGN.md section 4.6
static int train_version() {
    const char *e = getenv("BALER_AMD_BF16_TRAIN_V2");
    return e && e[0] == '1' ? 2 : (e && e[0] == '3' ? 3 : 1);
}
template <int F, int Z> const TrainOps *pick_train(const bamd_handle *h) {
    if (!TImpl<F, Z>::matches(h)) return nullptr;
    const int v = train_version();
    if (v == 3) { if constexpr (Z == 15) return TImpl3<F, Z>::ops(); }      // (built for the benchmarked shape first)
    return v == 2 ? TImpl2<F, Z>::ops() : TImpl<F, Z>::ops();
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
    for (int p = 0; p < 2; ++p) { st->src2[p].release(); st->w2[p].release(); }
    for (int p = 0; p < 4; ++p) { st->src3[p].release(); st->w3[p].release(); }
    for (int p = 0; p < 3; ++p) st->dz3[p].release();
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
