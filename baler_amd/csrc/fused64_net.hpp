// Geometry of the fp64 fused kernels, shared by fused64.hip (16x16x4 chains, weight-gradient tiles, inference) and fused64q.hip (the
// 4-row chain): tile helpers and the compile-time description of AE(F, Z) in both fragment orders.  Every translation unit gets its own
// copy (anonymous namespace): nothing here has linkage.
#pragma once
#include "fused.hpp"

namespace bamd {
namespace {

using d4 = double __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int tiles(int d) { return (d + 15) / 16; }
__host__ __device__ constexpr int tile_steps(int d, int t) { return d - 16 * t >= 16 ? 4 : (d - 16 * t + 3) / 4; }
// feature held by register r of tile t on lane group g (-1 = padding)
__host__ __device__ constexpr int creg_feature(int d, int t, int g, int r) { return 16 * t + 4 * r + g < d ? 16 * t + 4 * r + g : -1; }

template <int F, int Z> struct Net64 {
    static constexpr int L = 8;
    __host__ __device__ static constexpr int dim(int i) {
        return i == 0 ? F : i == 1 ? 200 : i == 2 ? 100 : i == 3 ? 50 : i == 4 ? Z : i == 5 ? 50 : i == 6 ? 100 : i == 7 ? 200 : F;
    }
    __host__ __device__ static constexpr bool act(int l) { return !(l == 3 || l == 7); }
    // packed buffer (d4 units = 32 bytes; a fragment = 64 lanes x d4 = 2 KiB): [ Wf(0..7) | Wb(7..1) | bias fragments ]
    __host__ __device__ static constexpr int wcount(int l) { return tiles(dim(l)) * tiles(dim(l + 1)) * 64; }
    __host__ __device__ static constexpr int wf_off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += wcount(j); return s; }
    __host__ __device__ static constexpr int wb_off(int l) { int s = wf_off(L); for (int j = L - 1; j > l; --j) s += wcount(j); return s; }
    __host__ __device__ static constexpr int bf_off(int l) { int s = wb_off(0); for (int j = 0; j < l; ++j) s += tiles(dim(j + 1)) * 4; return s; }
    __host__ __device__ static constexpr int packed_d4() { return bf_off(L) + 64; }
    __host__ __device__ static constexpr int dw_tiles(int l) { return tiles(dim(l + 1)) * tiles(dim(l) + 1); }
    __host__ __device__ static constexpr int slab_off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += dw_tiles(j); return s; }
    __host__ __device__ static constexpr int w_off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += dim(j + 1) * dim(j) + dim(j + 1); return s; }
    __host__ __device__ static constexpr int b_off(int l) { return w_off(l) + dim(l + 1) * dim(l); }
    __host__ __device__ static constexpr int nparams() { return w_off(L); }
    // global images of the chain: [16-row block][slot][16 rows]; X_l has 16 tiles(dim(l) + 1) slots (with the ones slot), dZ_l 16 tiles(dim(l+1))
    __host__ __device__ static constexpr int x_rows(int l) { return 16 * tiles(dim(l) + 1); }
    __host__ __device__ static constexpr int z_rows(int l) { return 16 * tiles(dim(l + 1)); }
    __host__ __device__ static constexpr int x_off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += x_rows(j); return s; }
    __host__ __device__ static constexpr int z_off(int l) { int s = x_off(L); for (int j = 0; j < l; ++j) s += z_rows(j); return s; }
    static constexpr int img_doubles = z_off(L) * 16;
    // ---- the 4-row chain (chain64q_kernel, v_mfma_f64_4x4x4_4b_f64): a second copy of the weights behind the 16x16x4 fragments.  GEMM g =
    // forward layer g (g < 8) or the transposed layer 15 - g (g = 8 .. 14); a fragment = 1 KiB = A operand of TWO MFMAs: lane l holds
    // A[16 grp + (l & 15)][8 k8 + (l >> 4)] and A[..][8 k8 + 4 + (l >> 4)]; fragment (k8, grp) of GEMM g sits at q_frag_off(g) + k8 G + grp.
    // Behind the fragments: the biases of the 8 layers in natural order, each padded to whole 16-feature groups.
    __host__ __device__ static constexpr int q_layer(int g) { return g < 8 ? g : 15 - g; }
    __host__ __device__ static constexpr int q_nout(int g) { return g < 8 ? dim(g + 1) : dim(15 - g); }
    __host__ __device__ static constexpr int q_kdim(int g) { return g < 8 ? dim(g) : dim(16 - g); }
    __host__ __device__ static constexpr int q_groups(int g) { return tiles(q_nout(g)); }
    __host__ __device__ static constexpr int q_ks8(int g) { return (q_kdim(g) + 7) / 8; }
    __host__ __device__ static constexpr int q_frag_off(int g) { int s = 0; for (int j = 0; j < g; ++j) s += q_groups(j) * q_ks8(j); return s; }
    __host__ __device__ static constexpr int q_frags() { return q_frag_off(15); }
    __host__ __device__ static constexpr int qb_off(int l) { int s = 0; for (int j = 0; j < l; ++j) s += 16 * tiles(dim(j + 1)); return s; }
    __host__ __device__ static constexpr int q_doubles() { return q_frags() * 128 + qb_off(L); }
    __host__ __device__ static constexpr int packed_all_doubles() { return packed_d4() * 4 + q_doubles(); }
};


__device__ __forceinline__ d4 mfma(double a, double b, d4 c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

struct WStream {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;   // lane * 32
};
// fragment idx of the stream: two 16-byte halves per lane
__device__ __forceinline__ d4 frag_rt(const WStream &ws, int idx) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    typedef double d2 __attribute__((ext_vector_type(2)));
    const u4 lo = __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff, idx * 2048, 0);
    const u4 hi = __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff + 16, idx * 2048, 0);
    const d2 a = __builtin_bit_cast(d2, lo), b = __builtin_bit_cast(d2, hi);
    return (d4){a[0], a[1], b[0], b[1]};
}

template <int NL> __device__ __forceinline__ void lrelu(d4 (&a)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) a[i][r] = a[i][r] > 0.0 ? a[i][r] : a[i][r] * kSlope;
}}  // namespace

// fused64i.hip / fused64j.hip: infer64_kernel<F, Z, KIND, RT> (encode / decode / forward + loss of an fp64 handle; kind: 0 / 1 / 2) for the
// shapes of find64 -- the exact 24-column latents and the classes up to 63 columns in fused64i.hip, the 64 .. 127-column / latent <= 63
// classes in fused64j.hip (two translation units: 69 fully unrolled kernels would take six minutes in one); BAMD_ERR_UNSUPPORTED for others
int fused64_infer_launch(int F, int Z, bool rt, bamd_handle *h, const double *packed, int kind, const void *x, int x_dtype, int64_t n,
                         const double *features, void *out, int out_dtype, const double *renorm, const uint8_t *imask, double *loss_sum,
                         hipStream_t s);
int fused64j_infer_launch(int F, int Z, bool rt, bamd_handle *h, const double *packed, int kind, const void *x, int x_dtype, int64_t n,
                          const double *features, void *out, int out_dtype, const double *renorm, const uint8_t *imask, double *loss_sum,
                          hipStream_t s);
// fused64q.hip: chain64q_kernel<F, Z, RT> for the shapes fused64.hip instantiates; BAMD_ERR_UNSUPPORTED for any other (F, Z, RT)
int fused64q_launch(int F, int Z, bool rt, unsigned grid, hipStream_t s, const double *qpacked, const void *x, int in_f64, int64_t rows,
                    const double *feats, double *imgs, double *loss_part, int fr);
}  // namespace bamd
