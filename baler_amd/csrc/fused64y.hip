// Weight-gradient tiles of a large fp64 batch with every tile owned by ONE WAVE and the image slices staged through LDS (dw64y_kernel),
// in its own translation unit: compiled WITHOUT -amdgpu-mfma-vgpr-form (fused64.hip has it) so that the 160 accumulator registers of a wave
// live in AccVGPRs and nothing else competes for them.  Geometry: fused64_net.hpp.  Launched by step64_common (fused64.hip).
#include "fused64_net.hpp"

#include <cstdint>

namespace bamd {
namespace {

// dw64x_kernel's workgroups own <= 16 tiles each and every wave holds all of them: a slice of the X image of layer 6 is wanted by seven
// workgroups, 4.27 GB move per 262,144 rows for 3.34 GB of images, and the launch sits at 75 % of the HBM rate it can reach.  Here a
// workgroup owns a quarter of the 298 tiles for every 16-row block of its block range and each of its waves owns 15 - 20 of them (one
// or two rectangles of the tile grids): four workgroup types x R ranges, sixteen wave programs.  The image slices a workgroup needs
// (22 or 36 of the 102, 2 KiB each) are copied global -> LDS by direct-to-LDS loads one block ahead (two buffers, one barrier per
// block), and every wave reads its MFMA operands from there: a slice comes from HBM once per workgroup that wants it (116 slice streams
// for 102 slices), no wave keeps slices in registers, no cross-wave reduction -- the owner stores its range partial.
//      tile grids (N tiles x K tiles): L0 13 x 2, L1 7 x 13, L2 4 x 7, L3 1 x 4, L4 4 x 1, L5 7 x 4, L6 13 x 7, L7 2 x 13 = 298 tiles
//      type 0: L6 rows 0-9, L3            (74 tiles, 22 slices)        type 2: L6 rows 10-12, L0, L2     (75 tiles, 36 slices)
//      type 1: L1 columns 0-9, L4         (74 tiles, 22 slices)        type 3: L1 columns 10-12, L7, L5  (75 tiles, 36 slices)
template <int L_, int N0_, int MN_, int K0_, int MK_> struct DwRect { static constexpr int l = L_, n0 = N0_, mn = MN_, k0 = K0_, mk = MK_; };
using DwNone = DwRect<0, 0, 0, 0, 0>;
struct DwGroup { int isx, l, first, cnt; };      // `cnt` slices of the X (isx) or dZ image of layer l from tile `first`
constexpr int kDwyGroups = 6, kDwyMaxSlices = 36;
__host__ __device__ constexpr DwGroup dwy_group(int type, int gi) {
    constexpr DwGroup g[4][kDwyGroups] = {
        {{0, 6, 0, 10}, {1, 6, 0, 7}, {0, 3, 0, 1}, {1, 3, 0, 4}, {0, 0, 0, 0}, {0, 0, 0, 0}},
        {{0, 1, 0, 7}, {1, 1, 0, 10}, {0, 4, 0, 4}, {1, 4, 0, 1}, {0, 0, 0, 0}, {0, 0, 0, 0}},
        {{0, 6, 10, 3}, {1, 6, 0, 7}, {0, 0, 0, 13}, {1, 0, 0, 2}, {0, 2, 0, 4}, {1, 2, 0, 7}},
        {{0, 1, 0, 7}, {1, 1, 10, 3}, {0, 7, 0, 2}, {1, 7, 0, 13}, {0, 5, 0, 7}, {1, 5, 0, 4}}};
    return g[type][gi];
}
__host__ __device__ constexpr int dwy_nslices(int type) { int s = 0; for (int gi = 0; gi < kDwyGroups; ++gi) s += dwy_group(type, gi).cnt; return s; }
// LDS position of slice `idx` of the X / dZ image of layer l in a workgroup of `type` (-1: the type does not stage it)
__host__ __device__ constexpr int dwy_lds_index(int type, int isx, int l, int idx) {
    int at = 0;
    for (int gi = 0; gi < kDwyGroups; ++gi) {
        const DwGroup g = dwy_group(type, gi);
        if (g.cnt > 0 && g.isx == isx && g.l == l && idx >= g.first && idx < g.first + g.cnt) return at + idx - g.first;
        at += g.cnt;
    }
    return -1;
}
template <class N> struct Dwy64 {
    // the partition is written for these tile grids (24 .. 31 columns, a latent of up to 15)
    static constexpr bool ok = tiles(N::dim(1)) == 13 && tiles(N::dim(0) + 1) == 2 && tiles(N::dim(2)) == 7 && tiles(N::dim(1) + 1) == 13 &&
                               tiles(N::dim(3)) == 4 && tiles(N::dim(2) + 1) == 7 && tiles(N::dim(4)) == 1 && tiles(N::dim(3) + 1) == 4 &&
                               tiles(N::dim(5)) == 4 && tiles(N::dim(4) + 1) == 1 && tiles(N::dim(6)) == 7 && tiles(N::dim(5) + 1) == 4 &&
                               tiles(N::dim(7)) == 13 && tiles(N::dim(6) + 1) == 7 && tiles(N::dim(8)) == 2 && tiles(N::dim(7) + 1) == 13;
    // byte offset of LDS slice j of a `type` workgroup inside a 16-row block of the images
    __host__ __device__ static constexpr int src_bytes(int type, int j) {
        int at = 0;
        for (int gi = 0; gi < kDwyGroups; ++gi) {
            const DwGroup g = dwy_group(type, gi);
            if (j < at + g.cnt) return ((g.isx ? N::x_off(g.l) : N::z_off(g.l)) + 16 * (g.first + j - at)) * 128;
            at += g.cnt;
        }
        return 0;
    }
};
constexpr size_t kDwyLdsBytes = 2 * (size_t)kDwyMaxSlices * 2048;      // two buffers of up to 36 slices: 144 KiB

__device__ __forceinline__ void dwy_dma_b128(unsigned lds_addr, int voff, __amdgpu_buffer_rsrc_t rs, int soff) {
    unsigned keep;      // M0 (the LDS base of a direct-to-LDS load) is not preserved by hipcc around asm: set and restore it here
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// the workgroup's slices of block b into LDS buffer `buf`: 2 KiB per slice = two 1-KiB loads, dealt to the four waves in turn
template <class N, int TYPE>
__device__ __forceinline__ void dwy_stage(const double *__restrict__ imgs, int b, unsigned buf, int wave, int lane) {
    constexpr int S = dwy_nslices(TYPE);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(imgs + (int64_t)b * N::img_doubles), 0, N::img_doubles * 8, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2 * S; ++j) {
        if ((j & 3) != wave) continue;
        dwy_dma_b128(__builtin_amdgcn_readfirstlane(buf + (unsigned)j * 1024u), lane * 16, rs, Dwy64<N>::src_bytes(TYPE, j >> 1) + (j & 1) * 1024);
    }
}
// rectangle R of one block from LDS: lane (i, g) multiplies rows 4 g + r (r = 0..3: one MFMA each) of slot i of its slices
template <class N, int TYPE, class R>
__device__ __forceinline__ void dwy_mma(d4 (&acc)[R::mn * R::mk > 0 ? R::mn * R::mk : 1], const double *buf, int lane) {
    if constexpr (R::mn > 0) {
        typedef const double __attribute__((address_space(3))) *lds_cd;
        typedef double __attribute__((ext_vector_type(2))) d2;
        typedef const d2 __attribute__((address_space(3))) *lds_cd2;
        const lds_cd base = (lds_cd)buf + ((lane & 15) * 16 + 4 * (lane >> 4));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            d2 za[R::mn], xa[R::mk];
#pragma unroll
            for (int a = 0; a < R::mn; ++a) {
                constexpr int dummy = 0; (void)dummy;
                za[a] = *(lds_cd2)(base + dwy_lds_index(TYPE, 0, R::l, R::n0 + a) * 256 + 2 * h);
            }
#pragma unroll
            for (int c = 0; c < R::mk; ++c) xa[c] = *(lds_cd2)(base + dwy_lds_index(TYPE, 1, R::l, R::k0 + c) * 256 + 2 * h);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int c = 0; c < R::mk; ++c)
#pragma unroll
                    for (int a = 0; a < R::mn; ++a) acc[c * R::mn + a] = mfma(za[a][q], xa[c][q], acc[c * R::mn + a]);
        }
    }
}
template <class N, class R>
__device__ __forceinline__ void dwy_store(const d4 (&acc)[R::mn * R::mk > 0 ? R::mn * R::mk : 1], double *__restrict__ part, int nsplit_total,
                                          int accumulate, int range, int lane) {
    if constexpr (R::mn > 0) {
        constexpr int ntc = tiles(N::dim(R::l + 1));
#pragma unroll
        for (int c = 0; c < R::mk; ++c)
#pragma unroll
            for (int a = 0; a < R::mn; ++a) {
                const int tile = N::slab_off(R::l) + (R::k0 + c) * ntc + (R::n0 + a);
                d4 *dst = (d4 *)(part + ((int64_t)tile * nsplit_total + range) * 256) + lane;      // element 4 lane + r: dw64_tile_block's layout
                d4 v = acc[c * R::mn + a];
                if (accumulate) { const d4 o = *dst; v = (d4){o[0] + v[0], o[1] + v[1], o[2] + v[2], o[3] + v[3]}; }
                *dst = v;
            }
    }
}
// one wave program: rectangles R1 (+ R2) of every block blo .. bhi - 1 (the same trip count for the four waves of a workgroup)
template <class N, int TYPE, class R1, class R2>
__device__ __forceinline__ void dw64y_wave(const double *__restrict__ imgs, int blo, int bhi, double *__restrict__ part, int nsplit_total,
                                           int accumulate, int range, unsigned char *lds, int wave) {
    static_assert(R1::mn > 0 && dwy_lds_index(TYPE, 0, R1::l, R1::n0) >= 0 && dwy_lds_index(TYPE, 0, R1::l, R1::n0 + R1::mn - 1) >= 0 &&
                  dwy_lds_index(TYPE, 1, R1::l, R1::k0) >= 0 && dwy_lds_index(TYPE, 1, R1::l, R1::k0 + R1::mk - 1) >= 0, "R1's slices are staged");
    static_assert(R2::mn == 0 || (dwy_lds_index(TYPE, 0, R2::l, R2::n0) >= 0 && dwy_lds_index(TYPE, 0, R2::l, R2::n0 + R2::mn - 1) >= 0 &&
                                  dwy_lds_index(TYPE, 1, R2::l, R2::k0) >= 0 && dwy_lds_index(TYPE, 1, R2::l, R2::k0 + R2::mk - 1) >= 0), "R2's slices are staged");
    constexpr int NA1 = R1::mn * R1::mk, NA2 = R2::mn * R2::mk;
    const int lane = threadIdx.x & 63;
    d4 acc1[NA1], acc2[NA2 > 0 ? NA2 : 1];
    // zeros that come OUT OF AN MFMA: the loop-carried accumulators then are AccVGPR values on every path and stay there (a plain zero
    // initialiser makes hipcc carry them in VGPRs and copy all 160 registers into AccVGPRs and back around every block)
    double zin = 0.0;
    asm volatile("" : "+v"(zin));
#pragma unroll
    for (int t = 0; t < NA1; ++t) acc1[t] = mfma(zin, zin, (d4){0.0, 0.0, 0.0, 0.0});
#pragma unroll
    for (int t = 0; t < (NA2 > 0 ? NA2 : 1); ++t) acc2[t] = mfma(zin, zin, (d4){0.0, 0.0, 0.0, 0.0});
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    constexpr unsigned kBuf = (unsigned)kDwyMaxSlices * 2048u;
    dwy_stage<N, TYPE>(imgs, blo, lds0, wave, lane);
    for (int b = blo; b < bhi; ++b) {
        const unsigned cur = (unsigned)((b - blo) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of block b has landed ...
        __syncthreads();                                       // ... and so has everybody's; block b - 1's buffer is free
        if (b + 1 < bhi) dwy_stage<N, TYPE>(imgs, b + 1, lds0 + (cur ^ 1u) * kBuf, wave, lane);
        const double *buf = (const double *)(lds + cur * kBuf);
        dwy_mma<N, TYPE, R1>(acc1, buf, lane);
        dwy_mma<N, TYPE, R2>(acc2, buf, lane);
    }
    dwy_store<N, R1>(acc1, part, nsplit_total, accumulate, range, lane);
    dwy_store<N, R2>(acc2, part, nsplit_total, accumulate, range, lane);
}
template <class N>
__global__ void __launch_bounds__(256) dw64y_kernel(const double *__restrict__ imgs, int nblk, double *__restrict__ part, int nsplit_total,
                                                    int accumulate, int nsplit) {
    if constexpr (Dwy64<N>::ok) {
        extern __shared__ __attribute__((aligned(1024))) unsigned char dwy_lds[];
        const int type = (int)blockIdx.x & 3, range = (int)blockIdx.x >> 2;
        if (range >= nsplit) return;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int per = (nblk + nsplit - 1) / nsplit, blo = range * per, bhi = blo + per < nblk ? blo + per : nblk;
#define BAMD_DWY(T_, W_, R1_, R2_) \
        if (type == T_ && wave == W_) { dw64y_wave<N, T_, R1_, R2_>(imgs, blo, bhi, part, nsplit_total, accumulate, range, dwy_lds, W_); return; }
#define BAMD_R(...) DwRect<__VA_ARGS__>
        BAMD_DWY(0, 0, BAMD_R(6, 0, 5, 0, 4), DwNone)
        BAMD_DWY(0, 1, BAMD_R(6, 0, 5, 4, 3), BAMD_R(3, 0, 1, 0, 4))
        BAMD_DWY(0, 2, BAMD_R(6, 5, 5, 0, 4), DwNone)
        BAMD_DWY(0, 3, BAMD_R(6, 5, 5, 4, 3), DwNone)
        BAMD_DWY(1, 0, BAMD_R(1, 0, 4, 0, 5), DwNone)
        BAMD_DWY(1, 1, BAMD_R(1, 4, 3, 0, 5), BAMD_R(4, 0, 4, 0, 1))
        BAMD_DWY(1, 2, BAMD_R(1, 0, 4, 5, 5), DwNone)
        BAMD_DWY(1, 3, BAMD_R(1, 4, 3, 5, 5), DwNone)
        BAMD_DWY(2, 0, BAMD_R(6, 10, 3, 0, 6), DwNone)
        BAMD_DWY(2, 1, BAMD_R(6, 10, 3, 6, 1), BAMD_R(0, 0, 8, 0, 2))
        BAMD_DWY(2, 2, BAMD_R(0, 8, 5, 0, 2), BAMD_R(2, 0, 4, 0, 2))
        BAMD_DWY(2, 3, BAMD_R(2, 0, 4, 2, 5), DwNone)
        BAMD_DWY(3, 0, BAMD_R(1, 0, 6, 10, 3), DwNone)
        BAMD_DWY(3, 1, BAMD_R(1, 6, 1, 10, 3), BAMD_R(7, 0, 2, 0, 8))
        BAMD_DWY(3, 2, BAMD_R(7, 0, 2, 8, 5), BAMD_R(5, 0, 2, 0, 4))
        BAMD_DWY(3, 3, BAMD_R(5, 2, 5, 0, 4), DwNone)
#undef BAMD_R
#undef BAMD_DWY
    }
}

// ---- the same ownership with the slices in REGISTERS (dw64w_kernel): three workgroup types x R ranges, twelve wave programs of 24 - 26
// tiles; a wave loads its own <= 15 slices of the next block while it multiplies this one (two sets: 240 VGPRs beside 208 AccVGPRs); the
// four waves of a type-A / type-B workgroup want the same seven X slices of layer 6 / dZ slices of layer 1 and a barrier per block keeps
// them within one block of each other, so those come from HBM once (measured: 3.68 GB per 262,144 rows against dw64x_kernel's 4.27).
//      A: L6 rows 0-2 + (row 12, k 0-3) | L6 rows 3-5 + (row 12, k 4-6) | L6 rows 6-8 + L3 | L6 rows 9-11 + L4          25 24 25 25
//      B: L1 k 0-2 + (k 12, rows 0-3)   | L1 k 3-5 + (k 12, rows 4-6)   | L1 k 6-8 + (L2, k 6) | L1 k 9-11 + (L5, row 6)  25 24 25 25
//      C: L0 | L7 | L2 k 0-5 | L5 rows 0-5                                                                               26 26 24 24
typedef double __attribute__((ext_vector_type(2))) dw2;
// HALF a block of a rectangle's slices: rows 4 g + 2 h, 4 g + 2 h + 1 of slot i (16 bytes per lane) = the operands of two of the block's
// four MFMAs per tile.  Four half-sets rotate: one is multiplied while three are in flight (9.6k cycles of MFMA work ahead instead of the
// 6.4k of two whole-block sets, in the same 240 VGPRs).
template <class N, class R>
__device__ __forceinline__ void dww_load(dw2 (&za)[R::mn > 0 ? R::mn : 1], dw2 (&xa)[R::mk > 0 ? R::mk : 1], const double *__restrict__ base) {
    if constexpr (R::mn > 0) {
#pragma unroll
        for (int a = 0; a < R::mn; ++a) za[a] = *(const dw2 *)(base + (N::z_off(R::l) + 16 * (R::n0 + a)) * 16);
#pragma unroll
        for (int c = 0; c < R::mk; ++c) xa[c] = *(const dw2 *)(base + (N::x_off(R::l) + 16 * (R::k0 + c)) * 16);
    }
}
template <class R>
__device__ __forceinline__ void dww_mma(d4 (&acc)[R::mn * R::mk > 0 ? R::mn * R::mk : 1], const dw2 (&za)[R::mn > 0 ? R::mn : 1],
                                        const dw2 (&xa)[R::mk > 0 ? R::mk : 1]) {
    if constexpr (R::mn > 0) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int c = 0; c < R::mk; ++c)
#pragma unroll
                for (int a = 0; a < R::mn; ++a) acc[c * R::mn + a] = mfma(za[a][r], xa[c][r], acc[c * R::mn + a]);
    }
}
template <class N, class R1, class R2, bool SYNC>
__device__ __forceinline__ void dw64w_wave(const double *__restrict__ imgs, int blo, int bhi, double *__restrict__ part, int nsplit_total,
                                           int accumulate, int range) {
    constexpr int NA1 = R1::mn * R1::mk, NA2 = R2::mn * R2::mk;
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
    const double *lbase = imgs + (i * 16 + 4 * g);
    d4 acc1[NA1], acc2[NA2 > 0 ? NA2 : 1];
    double zin = 0.0;      // zeros out of an MFMA: see dw64y_wave
    asm volatile("" : "+v"(zin));
#pragma unroll
    for (int t = 0; t < NA1; ++t) acc1[t] = mfma(zin, zin, (d4){0.0, 0.0, 0.0, 0.0});
#pragma unroll
    for (int t = 0; t < (NA2 > 0 ? NA2 : 1); ++t) acc2[t] = mfma(zin, zin, (d4){0.0, 0.0, 0.0, 0.0});
    // NS half-sets rotate through 240 VGPRs: six for a wave of <= 10 slices (2.5 blocks ahead), four up to 15 slices (1.5 blocks)
    constexpr int SL = R1::mn + R1::mk + R2::mn + R2::mk, NS = SL <= 10 ? 6 : 4;
    static_assert(SL <= 15, "two whole blocks of slices in 240 VGPRs");
    dw2 z1[NS][R1::mn], x1[NS][R1::mk], z2[NS][R2::mn > 0 ? R2::mn : 1], x2[NS][R2::mk > 0 ? R2::mk : 1];
    const int hlo = 2 * blo, hhi = 2 * bhi;
    auto load = [&](int set, int hb) {      // half-block hb = 2 b + h; beyond the range: the first one again, not multiplied
        const int q = hb < hhi ? hb : hlo;
        const double *base = lbase + (int64_t)(q >> 1) * N::img_doubles + 2 * (q & 1);
        dww_load<N, R1>(z1[set], x1[set], base);
        dww_load<N, R2>(z2[set], x2[set], base);
    };
#pragma unroll
    for (int k = 0; k < NS - 1; ++k) load(k, hlo + k);
    int hb = hlo;
    // NS / 2 whole blocks per trip; sched_barrier: the loads are ISSUED before the MFMAs of the half in hand (left free, hipcc sinks
    // them to their first use and the wave waits a memory round trip per block)
    for (; hb + NS - 1 < hhi; hb += NS) {
#pragma unroll
        for (int p = 0; p < NS; ++p) {
            if (SYNC && (p & 1) == 0) __syncthreads();
            load((p + NS - 1) % NS, hb + p + NS - 1);
            __builtin_amdgcn_sched_barrier(0);
            dww_mma<R1>(acc1, z1[p], x1[p]);
            dww_mma<R2>(acc2, z2[p], x2[p]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int q = 0; q < NS - 2; q += 2)      // the remaining whole blocks (fewer than NS / 2): their halves sit in sets q, q + 1
        if (hb + q < hhi) {
            if (SYNC) __syncthreads();
            dww_mma<R1>(acc1, z1[q], x1[q]);
            dww_mma<R2>(acc2, z2[q], x2[q]);
            dww_mma<R1>(acc1, z1[q + 1], x1[q + 1]);
            dww_mma<R2>(acc2, z2[q + 1], x2[q + 1]);
        }
    dwy_store<N, R1>(acc1, part, nsplit_total, accumulate, range, lane);
    dwy_store<N, R2>(acc2, part, nsplit_total, accumulate, range, lane);
}
template <class N>
__global__ void __launch_bounds__(256) dw64w_kernel(const double *__restrict__ imgs, int nblk, double *__restrict__ part, int nsplit_total,
                                                    int accumulate, int nsplit) {
    if constexpr (Dwy64<N>::ok) {
        const int type = (int)blockIdx.x & 3, range = (int)blockIdx.x >> 2;
        if (range >= nsplit) return;
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int per = (nblk + nsplit - 1) / nsplit, blo = range * per, bhi = blo + per < nblk ? blo + per : nblk;
#define BAMD_DWW(T_, W_, R1_, R2_, S_) \
        if (type == T_ && wave == W_) { dw64w_wave<N, R1_, R2_, S_>(imgs, blo, bhi, part, nsplit_total, accumulate, range); return; }
#define BAMD_R(...) DwRect<__VA_ARGS__>
        BAMD_DWW(0, 0, BAMD_R(6, 0, 3, 0, 7), DwNone, true)
        BAMD_DWW(0, 1, BAMD_R(6, 3, 3, 0, 7), DwNone, true)
        BAMD_DWW(0, 2, BAMD_R(6, 6, 3, 0, 7), DwNone, true)
        BAMD_DWW(0, 3, BAMD_R(6, 9, 3, 0, 7), DwNone, true)
        BAMD_DWW(1, 0, BAMD_R(1, 0, 7, 0, 3), DwNone, true)
        BAMD_DWW(1, 1, BAMD_R(1, 0, 7, 3, 3), DwNone, true)
        BAMD_DWW(1, 2, BAMD_R(1, 0, 7, 6, 3), DwNone, true)
        BAMD_DWW(1, 3, BAMD_R(1, 0, 7, 9, 3), DwNone, true)
        BAMD_DWW(2, 0, BAMD_R(0, 0, 7, 0, 2), BAMD_R(3, 0, 1, 0, 4), false)
        BAMD_DWW(2, 1, BAMD_R(0, 7, 6, 0, 2), BAMD_R(4, 0, 4, 0, 1), false)
        BAMD_DWW(2, 2, BAMD_R(2, 0, 4, 0, 4), DwNone, false)
        BAMD_DWW(2, 3, BAMD_R(2, 0, 4, 4, 3), BAMD_R(6, 12, 1, 0, 7), false)
        BAMD_DWW(3, 0, BAMD_R(7, 0, 2, 0, 7), DwNone, false)
        BAMD_DWW(3, 1, BAMD_R(7, 0, 2, 7, 6), DwNone, false)
        BAMD_DWW(3, 2, BAMD_R(5, 0, 4, 0, 4), DwNone, false)
        BAMD_DWW(3, 3, BAMD_R(5, 4, 3, 0, 4), BAMD_R(1, 0, 7, 12, 1), false)
#undef BAMD_R
#undef BAMD_DWW
    }
}

template <int F, int Z>
int dwy_go(int ns, hipStream_t s, const double *imgs, int nblk, double *part, int nsplit_total, int accumulate) {
    static const hipError_t attr = hipFuncSetAttribute((const void *)dw64y_kernel<Net64<F, Z>>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)kDwyLdsBytes);
    BAMD_HIP(attr);
    if (env_ll("BALER_AMD_DW64Y_LDS", 0) == 0)
        hipLaunchKernelGGL((dw64w_kernel<Net64<F, Z>>), dim3(4 * ns), dim3(256), 0, s, imgs, nblk, part, nsplit_total, accumulate, ns);
    else
    hipLaunchKernelGGL((dw64y_kernel<Net64<F, Z>>), dim3(4 * ns), dim3(256), kDwyLdsBytes, s, imgs, nblk, part, nsplit_total, accumulate, ns);
    BAMD_HIP(hipGetLastError());
    return BAMD_OK;
}

}  // namespace

// the shapes instantiated: the 24-column model at every latent of the compression-ratio knob (fused64.hip find64)
#ifdef BAMD_F64_QUICK
#define BAMD_DWY_SHAPES(X) X(24, 15)
#else
#define BAMD_DWY_SHAPES(X) X(24, 15) X(24, 12) X(24, 8) X(24, 6) X(24, 10) X(24, 5) X(24, 4) X(24, 3) X(24, 2)
#endif
bool fused64y_has(int F, int Z) {
#define X(F_, Z_) if (F == F_ && Z == Z_) return Dwy64<Net64<F_, Z_>>::ok;
    BAMD_DWY_SHAPES(X)
#undef X
    return false;
}
int fused64y_launch(int F, int Z, int ns, hipStream_t s, const double *imgs, int nblk, double *part, int nsplit_total, int accumulate) {
#define X(F_, Z_) if (F == F_ && Z == Z_) return dwy_go<F_, Z_>(ns, s, imgs, nblk, part, nsplit_total, accumulate);
    BAMD_DWY_SHAPES(X)
#undef X
    set_error("dw64y_kernel is not instantiated for this shape");
    return BAMD_ERR_UNSUPPORTED;
}

}  // namespace bamd
