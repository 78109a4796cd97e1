// Internal declarations shared by the translation units of libbaler_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/baler_amd.h"

namespace bamd {

constexpr double kSlope = 0.01;  // F.leaky_relu default negative_slope (models.py:142-150); cast to T at use

void set_error(const std::string &msg);

#define BAMD_HIP(call)                                                                      \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            bamd::set_error(std::string(#call) + ": " + hipGetErrorString(e_));             \
            return BAMD_ERR_HIP;                                                            \
        }                                                                                   \
    } while (0)

#define BAMD_REQUIRE(cond, msg)                                                             \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            bamd::set_error(std::string(__func__) + ": " + (msg));                          \
            return BAMD_ERR_INVALID;                                                        \
        }                                                                                   \
    } while (0)

// Tuning knobs are read on EVERY call (a getenv is nothing beside a launch; tests and A/B runs toggle them inside one process)
inline long long env_ll(const char *name, long long dflt) {
    const char *e = getenv(name);
    return e && e[0] ? atoll(e) : dflt;
}
inline bool env_off(const char *name) {      // "NAME=0" switches a default-on path off
    const char *e = getenv(name);
    return e && e[0] == '0';
}

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need);  // grow-only; returns bamd_status
    void release();
};

}  // namespace bamd

struct bamd_handle {
    int L = 0;
    int mode = 0;
    int device = 0;
    std::vector<int> dims;          // L+1
    std::vector<int64_t> w_off;     // offset of W_l in the flat vector
    std::vector<int64_t> b_off;     // offset of b_l
    int64_t nparams = 0;
    int sum_dims = 0;               // sum of dims[1..L]
    int max_dim = 0;
    size_t esize = 4;               // sizeof parameter/compute scalar (4 or 8)

    bamd::DevBuf params;            // flat copy of the parameters in the compute type
    bamd::DevBuf packed;            // MFMA-fragment-packed weights for the fused kernels
    bamd::DevBuf work;              // activation workspace (generic path)
    bamd::DevBuf slabs;             // per-workgroup partial gradients
    bamd::DevBuf lossp;             // partial loss sums (double)
    bamd::DevBuf gscratch;          // gradient buffer of bamd_train_step() when the caller passes none
    bool params_loaded = false;
    bool fused_ok = false;          // shape is served by the fused register-chained kernels
    void *fused_state = nullptr;    // index maps of the fused path (fused.hip)
    void *fused_small = nullptr;    // 64..127-column tables: second fused state (small-batch class kernels) beside the wide class in fused_state
    bamd::DevBuf packed_small;      // ... and its fragment-packed weights (fused.hip: SmallScope swaps both in for a small-batch call)
    void *fused64_state = nullptr;  // maps + packed fp64 weights of the fp64 small-batch step (fused64.hip)
    void *bf16_state = nullptr;     // packed bf16 weights + maps of the bf16 inference mode (bf16.hip)
    void *bf16_train_state = nullptr;   // packed bf16 weights + maps of the bf16 training kernels (bf16_train.hip)
    bool bf16_infer_stale = false;  // the inference fragments lag h->params (re-packed lazily by the next inference call)
    bool bf16_train_stale = false;  // the bf16 TRAINING fragments lag h->params (re-packed by the next bf16 training launch)
    void *comm = nullptr;           // ncclComm_t of data-parallel training (comm.hip); null: single process
    bool comm_owned = false;        // created by bamd_comm_init (destroyed with the handle) vs attached by the caller
    int comm_world = 0;

    bool has_act(int l) const { return !(l == L / 2 - 1 || l == L - 1); }
};

namespace bamd {

// ---- elementwise.hip ----------------------------------------------------------------------------
DevBuf &scratch_for(int purpose, hipStream_t s);   // grow-only scratch of the handle-free kernels, per (purpose, device, stream)
int launch_minmax(const void *x, int dtype, int64_t n, int c, double *features, hipStream_t s, bool raw = false);
int launch_normalize(const void *x, int dtype, int64_t n, int c, const double *features, void *out,
                     int out_dtype, hipStream_t s);
int launch_renormalize(const void *x, int dtype, int64_t n, int c, const double *features,
                       const uint8_t *int_mask, double *out, hipStream_t s);
int launch_convert(const void *src, int src_dtype, void *dst, int dst_dtype, int64_t count,
                   hipStream_t s);
int launch_emd_rows(const void *x, const void *recon, int dtype, int64_t n, int c, double *out,
                    hipStream_t s);
int launch_swd(const void *z, const void *prior, const void *proj, int dtype, int64_t n, int d, int ns, double reg_weight,
               double *loss_out, void *dz_out, hipStream_t s);
int launch_error_deltas(const void *x, const void *recon, int dtype, int64_t n, double bound, uint8_t *flags, void *deltas,
                        hipStream_t s);
int launch_apply_deltas(void *out, int dtype, int n_cols, const int64_t *rows, const int32_t *cols, const void *deltas,
                        int64_t count, hipStream_t s);
int launch_adam(void *params, void *params_copy, const void *grads, void *m, void *v, int64_t np,
                size_t esize, const bamd_adam &hp, double *loss_accum, const int *sc_off, const int *sc_idx,
                void *packed, hipStream_t s);

// ---- comm.hip (RCCL resolved at run time) -----------------------------------------------------------
int comm_allreduce_sum(bamd_handle *h, void *buf, int dtype, int64_t count, hipStream_t s);
void comm_teardown(bamd_handle *h);

// ---- generic.hip (layer-by-layer MFMA path, any dims) -------------------------------------------
int generic_forward(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features,
                    int l0, int l1, void *out, int out_dtype, const double *renorm,
                    const uint8_t *int_mask, hipStream_t s);
int generic_forward_loss(bamd_handle *h, const void *x, int x_dtype, int64_t n,
                         const double *features, void *recon, int recon_dtype, double *loss_sum,
                         hipStream_t s);
int generic_fwd_bwd(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features,
                    void *grads, hipStream_t s,
                    const void *latent_grad = nullptr);
int generic_small_train_step(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads, void *params,
                             void *m, void *v, const bamd_adam &hp, double *loss_accum, hipStream_t s);   // latent_grad: (n, z_dim) of the compute type, added to dL/dz
int generic_activation_means(bamd_handle *h, const void *x, int x_dtype, int64_t n,
                             const double *features, double *out, int max_nodes, hipStream_t s);


// Sum of n doubles by ONE 256-thread workgroup in a fixed order (bitwise reproducible): 256 strided partial sums, then a fixed
// tree through `sh` (256 doubles of LDS).  The result is valid in thread 0.  (A single thread adding the partials one after the
// other -- what every loss reduction here did first -- is a chain of dependent L2 round trips: 6.5 us for 32 partials.)
#if defined(__HIPCC__)
__device__ __forceinline__ double block_sum_fixed(const double *__restrict__ part, int n, double *sh) {
    double s = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) s += part[k];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    return sh[0];
}
#endif
}  // namespace bamd
