// C ABI entry points of libbaler_amd.so (see include/baler_amd.h).  gfx950 only.
#include <cstdio>
#include <cstring>
#include <string>

#include "bamd_internal.hpp"
#include "fused.hpp"
#include "bf16.hpp"

namespace bamd {

static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }

int DevBuf::ensure(size_t need) {
    if (need <= bytes) return BAMD_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    // round up so that slowly growing batches do not reallocate every call
    size_t want = (need + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        set_error(std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e));
        p = nullptr;
        return BAMD_ERR_ALLOC;
    }
    bytes = want;
    return BAMD_OK;
}

void DevBuf::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
}

}  // namespace bamd

using namespace bamd;

namespace {
// The handle's buffers live on its device: make it current for the duration of a call and give the caller's current
// device back afterwards (a single process that drives several GPUs must not find its device changed under it).
struct DeviceGuard {
    int prev = -1, rc = hipSuccess;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) rc = (int)hipSetDevice(dev); else prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

extern "C" {

int bamd_abi_version(void) { return BAMD_ABI_VERSION; }

const char *bamd_last_error(void) { return g_err.c_str(); }

int bamd_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
        return BAMD_ERR_NO_DEVICE;
    }
    return n;
}

int bamd_create(const int *dims, int n_layers, int mode, int device, bamd_handle **out) {
    BAMD_REQUIRE(dims && out, "null argument");
    BAMD_REQUIRE(n_layers >= 2 && n_layers % 2 == 0, "n_layers must be even and >= 2");
    BAMD_REQUIRE(mode == BAMD_MODE_F32 || mode == BAMD_MODE_F64 || mode == BAMD_MODE_BF16, "unknown mode");
    for (int l = 0; l <= n_layers; ++l) BAMD_REQUIRE(dims[l] > 0, "layer widths must be positive");
    int ndev = bamd_device_count();
    if (ndev <= 0) {
        if (ndev == 0) set_error("no HIP device visible");
        return BAMD_ERR_NO_DEVICE;
    }
    BAMD_REQUIRE(device >= 0 && device < ndev, "device ordinal out of range");
    DeviceGuard guard(device);
    BAMD_REQUIRE(guard.rc == hipSuccess, "cannot select the device");
    hipDeviceProp_t prop;
    BAMD_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error(std::string("libbaler_amd is built for gfx950 only; device is ") + prop.gcnArchName);
        return BAMD_ERR_NO_DEVICE;
    }
    bamd_handle *h = new bamd_handle();
    h->L = n_layers;
    h->mode = mode;
    h->device = device;
    h->dims.assign(dims, dims + n_layers + 1);
    h->esize = mode == BAMD_MODE_F64 ? 8 : 4;
    int64_t off = 0;
    for (int l = 0; l < n_layers; ++l) {
        h->w_off.push_back(off);
        off += (int64_t)dims[l + 1] * dims[l];
        h->b_off.push_back(off);
        off += dims[l + 1];
        h->sum_dims += dims[l + 1];
    }
    h->nparams = off;
    for (int l = 0; l <= n_layers; ++l) h->max_dim = dims[l] > h->max_dim ? dims[l] : h->max_dim;
    // BAMD_MODE_BF16 of a shape without bf16 kernels (any AE / CFD_dense_AE(n_features, z_dim) the reference builds, models.py:122-139,
    // 192-209, other than the instantiated ones): the handle computes in float32 on whatever serves the shape there (run-time-width
    // classes, layer-wise kernels) and says so; bamd_mode_of() then reports BAMD_MODE_F32.  A slower path, not an error.
    bool demoted = false;
    if (mode == BAMD_MODE_BF16 && !bf16_has_kernels(h) && !fused_has_bf16_kernels(h)) {
        h->mode = BAMD_MODE_F32;
        demoted = true;
    }
    int rc = h->params.ensure((size_t)(h->nparams + 1) * h->esize);
    if (rc) { delete h; return rc; }
    rc = fused_setup(h);
    if (rc) { bamd_destroy(h); return rc; }
    rc = fused64_setup(h);
    if (rc) { bamd_destroy(h); return rc; }
    if (h->mode == BAMD_MODE_BF16 && !fused_serves_bf16_inference(h)) {   // (wide models in the bf16 mode are served by fused.hip)
        rc = bf16_setup(h);
        if (rc) { bamd_destroy(h); return rc; }
        rc = bf16_train_setup(h);
        if (rc) { bamd_destroy(h); return rc; }
    }
    *out = h;
    const int path = bamd_path_of(h);
    if (demoted) {
        const char *q = getenv("BALER_AMD_QUIET");
        if (!(q && q[0] == '1')) {
            std::string d;
            for (int l = 0; l <= n_layers; ++l) d += (l ? "-" : "") + std::to_string(dims[l]);
            fprintf(stderr, "[baler_amd] model %s: BAMD_MODE_BF16 has kernels for the 24-column AE and the 2500-25 / 625-7 / 512-6 wide models only; "
                            "this handle computes in float32 (%s)\n", d.c_str(),
                    path == BAMD_PATH_GENERIC ? "layer-wise kernels" : "fused run-time-width kernels");
        }
    }
    if (path == BAMD_PATH_GENERIC || path == BAMD_PATH_FUSED_INFER) {
        const char *q = getenv("BALER_AMD_QUIET");
        if (!(q && q[0] == '1')) {
            std::string d;
            for (int l = 0; l <= n_layers; ++l) d += (l ? "-" : "") + std::to_string(dims[l]);
            const std::string infer_only = "throughput training kernels (encode / decode / validation and training steps of up to " +
                                           std::to_string((long long)fused_latency_rows(h)) + " rows are fused)";
            fprintf(stderr, "[baler_amd] model %s (%s) has no fused %s: %s run layer by layer (generic.hip, activations through HBM)\n",
                    d.c_str(), h->mode == BAMD_MODE_F64 ? "fp64" : h->mode == BAMD_MODE_BF16 ? "bf16" : "fp32",
                    path == BAMD_PATH_GENERIC ? "kernel instantiation" : infer_only.c_str(),
                    path == BAMD_PATH_GENERIC ? "encode / decode / training" : "larger training batches");
        }
    }
    return BAMD_OK;
}

int bamd_path_of(const bamd_handle *h) {
    BAMD_REQUIRE(h, "null handle");
    if (h->mode == BAMD_MODE_BF16 && h->bf16_state) return BAMD_PATH_BF16;
    if (h->mode == BAMD_MODE_F64) return h->fused64_state ? BAMD_PATH_FUSED : BAMD_PATH_GENERIC;
    if (!h->fused_ok) return BAMD_PATH_GENERIC;
    return fused_trains(h) ? BAMD_PATH_FUSED : BAMD_PATH_FUSED_INFER;
}

void bamd_destroy(bamd_handle *h) {
    if (!h) return;
    DeviceGuard guard(h->device);
    fused_teardown(h);
    fused64_teardown(h);
    bf16_teardown(h);
    bf16_train_teardown(h);
    comm_teardown(h);
    h->params.release();
    h->packed.release();
    h->work.release();
    h->slabs.release();
    h->lossp.release();
    h->gscratch.release();
    delete h;
}

int64_t bamd_param_count(const bamd_handle *h) { return h ? h->nparams : 0; }
int bamd_mode_of(const bamd_handle *h) { return h ? h->mode : BAMD_ERR_INVALID; }

int bamd_load_params(bamd_handle *h, const void *params, int dtype, void *stream) {
    BAMD_REQUIRE(h && params, "null argument");
    BAMD_REQUIRE(dtype == BAMD_F32 || dtype == BAMD_F64, "bad dtype");
    DeviceGuard guard(h->device);
    BAMD_REQUIRE(guard.rc == hipSuccess, "cannot select the handle's device");
    hipStream_t s = (hipStream_t)stream;
    int rc = launch_convert(params, dtype, h->params.p, h->esize == 8 ? BAMD_F64 : BAMD_F32, h->nparams, s);
    if (rc) return rc;
    h->params_loaded = true;
    if (h->mode == BAMD_MODE_BF16 && h->bf16_state) {
        rc = bf16_pack(h, s);
        h->bf16_infer_stale = false;
        h->bf16_train_stale = false;
        if (!rc) rc = bf16_train_pack(h, s);
        return rc ? rc : fused_pack(h, s);           // fp32 fragments of the small-batch kernels (no-op without them)
    }
    if (h->mode == BAMD_MODE_F64) return fused64_pack(h, s);
    return fused_pack(h, s);
}

int bamd_minmax(const void *x, int dtype, int64_t n_rows, int n_cols, double *features, void *stream) {
    return launch_minmax(x, dtype, n_rows, n_cols, features, (hipStream_t)stream);
}

int bamd_col_minmax(const void *x, int dtype, int64_t n_rows, int n_cols, double *minmax, void *stream) {
    return launch_minmax(x, dtype, n_rows, n_cols, minmax, (hipStream_t)stream, true);
}

int bamd_normalize(const void *x, int dtype, int64_t n_rows, int n_cols, const double *features, void *out,
                   int out_dtype, void *stream) {
    return launch_normalize(x, dtype, n_rows, n_cols, features, out, out_dtype, (hipStream_t)stream);
}

int bamd_renormalize(const void *x, int dtype, int64_t n_rows, int n_cols, const double *features,
                     const uint8_t *int_mask, double *out, void *stream) {
    return launch_renormalize(x, dtype, n_rows, n_cols, features, int_mask, out, (hipStream_t)stream);
}

// BF16 handles of the 24-column model train SMALL batches on the fp32 small-batch kernels: measured us per bamd_train_step,
// fp32 / bf16 kernels: 512 rows 23 / 34, 2048 rows 34 / 37, 8192 rows 72 / 51, 32768 rows 161 / 80 -- the bf16 pair needs ~3000
// rows to win (its workgroups own 64 rows each: a 512-row batch occupies 8 CUs).  BALER_AMD_BF16_SMALL_ROWS overrides (0: never).
static int64_t bf16_small_rows() {
    const char *e = getenv("BALER_AMD_BF16_SMALL_ROWS");       // read per call: tests toggle it
    return e ? atoll(e) : 3072;
}
static bool bf16_kernels_train(const bamd_handle *h, int64_t n_rows) {
    return h->mode == BAMD_MODE_BF16 && bf16_train_ok(h) && !(h->fused_ok && n_rows <= bf16_small_rows());
}
static int bf16_train_sync(bamd_handle *h, hipStream_t s) {   // the bf16 training fragments are re-rounded on demand
    if (!h->bf16_train_stale) return BAMD_OK;
    h->bf16_train_stale = false;
    return bf16_train_pack(h, s);
}

// bf16 handles re-round the INFERENCE fragments lazily: a training step refreshes only what the next step reads
static int bf16_sync(bamd_handle *h, hipStream_t s) {
    if (h->mode != BAMD_MODE_BF16 || !h->bf16_state || !h->bf16_infer_stale) return BAMD_OK;
    h->bf16_infer_stale = false;
    return bf16_pack(h, s);
}

// Every path that has just written h->params / h->packed (the Adam kernel, or Adam inside a weight-gradient launch) ends here: lazily
// refreshed float32 copies are stale, and a BF16 handle's bf16 fragments are re-rounded (on demand where the handle has both sets).
static int params_stepped(bamd_handle *h, hipStream_t s) {
    fused_params_changed(h);
    if (h->mode == BAMD_MODE_BF16 && h->bf16_state) {
        if (bf16_train_ok(h)) { h->bf16_infer_stale = true; h->bf16_train_stale = true; }
        else return bf16_pack(h, s);
    }
    return BAMD_OK;
}

#define BAMD_CHECK_MODEL(h)                                                        \
    BAMD_REQUIRE(h, "null handle");                                                \
    BAMD_REQUIRE((h)->params_loaded, "bamd_load_params() has not been called");    \
    DeviceGuard guard_((h)->device);                                               \
    BAMD_REQUIRE(guard_.rc == hipSuccess, "cannot select the handle's device");

int bamd_encode(bamd_handle *h, const void *x, int x_dtype, int64_t n_rows, const double *features, void *z,
                int z_dtype, void *stream) {
    BAMD_CHECK_MODEL(h);
    BAMD_REQUIRE(n_rows >= 0 && ((x && z) || n_rows == 0), "bad arguments");
    if (n_rows == 0) return BAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = bf16_sync(h, s)) return rc;
    if (h->mode == BAMD_MODE_BF16 && h->bf16_state) return bf16_encode(h, x, x_dtype, n_rows, features, z, z_dtype, s);
    if (h->fused_ok) return fused_encode(h, x, x_dtype, n_rows, features, z, z_dtype, s);
    if (h->mode == BAMD_MODE_F64) {
        const int rc = fused64_infer(h, 0, x, x_dtype, n_rows, features, z, z_dtype, nullptr, nullptr, nullptr, s);
        if (rc != BAMD_ERR_UNSUPPORTED) return rc;
    }
    return generic_forward(h, x, x_dtype, n_rows, features, 0, h->L / 2, z, z_dtype, nullptr, nullptr, s);
}

int bamd_decode(bamd_handle *h, const void *z, int z_dtype, int64_t n_rows, const double *features,
                const uint8_t *int_mask, void *out, int out_dtype, void *stream) {
    BAMD_CHECK_MODEL(h);
    BAMD_REQUIRE(n_rows >= 0 && ((z && out) || n_rows == 0), "bad arguments");
    if (n_rows == 0) return BAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    if (int rc = bf16_sync(h, s)) return rc;
    if (h->mode == BAMD_MODE_BF16 && h->bf16_state) return bf16_decode(h, z, z_dtype, n_rows, features, int_mask, out, out_dtype, s);
    if (h->fused_ok) return fused_decode(h, z, z_dtype, n_rows, features, int_mask, out, out_dtype, s);
    if (h->mode == BAMD_MODE_F64 && (!features || out_dtype == BAMD_F64)) {      // (un-normalised output is float64, as renormalize_k's)
        const int rc = fused64_infer(h, 1, z, z_dtype, n_rows, nullptr, out, out_dtype, features, int_mask, nullptr, s);
        if (rc != BAMD_ERR_UNSUPPORTED) return rc;
    }
    return generic_forward(h, z, z_dtype, n_rows, nullptr, h->L / 2, h->L, out, out_dtype, features, int_mask, s);
}

int bamd_forward_loss(bamd_handle *h, const void *x, int x_dtype, int64_t n_rows, const double *features,
                      void *recon, int recon_dtype, double *loss_sum, void *stream) {
    BAMD_CHECK_MODEL(h);
    BAMD_REQUIRE(x && loss_sum && n_rows > 0, "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (int rc = bf16_sync(h, s)) return rc;
    if (h->mode == BAMD_MODE_BF16 && h->bf16_state) return bf16_forward_loss(h, x, x_dtype, n_rows, features, recon, recon_dtype, loss_sum, s);
    if (h->fused_ok) return fused_forward_loss(h, x, x_dtype, n_rows, features, recon, recon_dtype, loss_sum, s);
    if (h->mode == BAMD_MODE_F64) {
        const int rc = fused64_infer(h, 2, x, x_dtype, n_rows, features, recon, recon_dtype, nullptr, nullptr, loss_sum, s);
        if (rc != BAMD_ERR_UNSUPPORTED) return rc;
    }
    return generic_forward_loss(h, x, x_dtype, n_rows, features, recon, recon_dtype, loss_sum, s);
}

int bamd_fwd_bwd(bamd_handle *h, const void *x, int x_dtype, int64_t n_rows, const double *features,
                 void *grads, void *stream) {
    BAMD_CHECK_MODEL(h);
    BAMD_REQUIRE(grads && n_rows >= 0 && (x || n_rows == 0), "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (n_rows == 0) {  // an empty shard of a global batch contributes a zero gradient and zero loss
        BAMD_HIP(hipMemsetAsync(grads, 0, (size_t)(h->nparams + 1) * h->esize, s));
        return BAMD_OK;
    }
    if (bf16_kernels_train(h, n_rows)) {
        if (int rc = bf16_train_sync(h, s)) return rc;
        return bf16_fwd_bwd(h, x, x_dtype, n_rows, features, grads, s);
    }
    if (h->fused_ok) return fused_fwd_bwd(h, x, x_dtype, n_rows, features, grads, s);
    if (h->mode == BAMD_MODE_F64) {   // small batches: fp64 chain + weight-gradient tiles; otherwise the layer-wise kernels
        int rc = fused64_step(h, x, x_dtype, n_rows, features, grads, nullptr, nullptr, nullptr, nullptr, nullptr, s);
        if (rc != BAMD_ERR_UNSUPPORTED) return rc;
    }
    return generic_fwd_bwd(h, x, x_dtype, n_rows, features, grads, s);
}

int bamd_fwd_bwd_latent(bamd_handle *h, const void *x, int x_dtype, int64_t n_rows, const double *features,
                        const void *latent_grad, void *grads, void *stream) {
    BAMD_CHECK_MODEL(h);
    BAMD_REQUIRE(grads && x && n_rows > 0, "bad arguments");
    if (!latent_grad) return bamd_fwd_bwd(h, x, x_dtype, n_rows, features, grads, stream);
    // the regulariser's gradient enters between the decoder's and the encoder's backward products: layer-wise path
    return generic_fwd_bwd(h, x, x_dtype, n_rows, features, grads, (hipStream_t)stream, latent_grad);
}

int bamd_swd(const void *z, const void *prior, const void *proj, int dtype, int64_t n_rows, int z_dim, int n_proj,
             double reg_weight, double *loss_out, void *dz_out, void *stream) {
    BAMD_REQUIRE(dtype == BAMD_F32 || dtype == BAMD_F64, "bad dtype");
    return launch_swd(z, prior, proj, dtype, n_rows, z_dim, n_proj, reg_weight, loss_out, dz_out, (hipStream_t)stream);
}

int bamd_adam_step(bamd_handle *h, void *params, const void *grads, void *m, void *v, const bamd_adam *hp,
                   double *loss_accum, void *stream) {
    BAMD_CHECK_MODEL(h);
    BAMD_REQUIRE(params && grads && m && v && hp, "null argument");
    BAMD_REQUIRE(hp->step >= 1, "step must be >= 1");
    hipStream_t s = (hipStream_t)stream;
    const int *sc_off = nullptr, *sc_idx = nullptr;
    void *packed = nullptr;
    fused_scatter(h, &sc_off, &sc_idx, &packed);   // Adam also refreshes the packed weight copy (one launch)
    if (h->mode == BAMD_MODE_F64) fused64_scatter(h, &sc_off, &sc_idx, &packed);
    int rc = launch_adam(params, h->params.p, grads, m, v, h->nparams, h->esize, *hp, loss_accum, sc_off, sc_idx, packed, s);
    return rc == BAMD_OK ? params_stepped(h, s) : rc;
}

int bamd_train_step(bamd_handle *h, const void *x, int x_dtype, int64_t n_rows, const double *features, void *params,
                    void *grads, void *m, void *v, const bamd_adam *hp, double *loss_accum, void *stream) {
    BAMD_CHECK_MODEL(h);
    BAMD_REQUIRE(params && m && v && hp && n_rows >= 0 && (x || n_rows == 0), "bad arguments");
    BAMD_REQUIRE(hp->step >= 1, "step must be >= 1");
    hipStream_t s = (hipStream_t)stream;
    if (h->comm) {      // data parallel: this rank's rows -> [grads | loss] summed over the ranks -> the replicated Adam step
        if (!grads) {
            int rc = h->gscratch.ensure((size_t)(h->nparams + 1) * h->esize);
            if (rc) return rc;
            grads = h->gscratch.p;
        }
        int rc = bamd_fwd_bwd(h, x, x_dtype, n_rows, features, grads, stream);
        if (rc) return rc;
        rc = comm_allreduce_sum(h, grads, h->esize == 8 ? BAMD_F64 : BAMD_F32, h->nparams + 1, s);
        if (rc) return rc;
        return bamd_adam_step(h, params, grads, m, v, hp, loss_accum, stream);
    }
    if (n_rows > 0 && !bf16_kernels_train(h, n_rows)) {
        int rc = fused_train_step(h, x, x_dtype, n_rows, features, grads, params, m, v, *hp, loss_accum, s);
        if (rc != BAMD_ERR_UNSUPPORTED) {
            if (rc == BAMD_OK && h->mode == BAMD_MODE_BF16) rc = params_stepped(h, s);
            return rc;
        }
        if (h->mode == BAMD_MODE_F64) {
            rc = fused64_step(h, x, x_dtype, n_rows, features, grads, params, m, v, hp, loss_accum, s);
            if (rc != BAMD_ERR_UNSUPPORTED) return rc;
        }
    }
    if (!grads) {
        int rc = h->gscratch.ensure((size_t)(h->nparams + 1) * h->esize);
        if (rc) return rc;
        grads = h->gscratch.p;
    }
    // small batches on the layer-wise / wide launches: Adam inside the weight-gradient launch.  Only for models whose training runs on
    // generic.hip anyway (no fused state, or the wide launches): a fused narrow handle that declined above (BALER_AMD_LATENCY_ROWS below
    // the batch) keeps its throughput pair, as README says of that knob.
    if (n_rows > 0 && h->mode != BAMD_MODE_F64 && !bf16_kernels_train(h, n_rows) && (!h->fused_ok || fused_wide_train(h))) {
        int rc = generic_small_train_step(h, x, x_dtype, n_rows, features, grads, params, m, v, *hp, loss_accum, s);
        if (rc != BAMD_ERR_UNSUPPORTED) {
            if (rc == BAMD_OK) rc = params_stepped(h, s);
            return rc;
        }
    }
    int rc = bamd_fwd_bwd(h, x, x_dtype, n_rows, features, grads, stream);
    if (rc) return rc;
    return bamd_adam_step(h, params, grads, m, v, hp, loss_accum, stream);
}

int bamd_train_epoch(bamd_handle *h, const void *x, int x_dtype, int64_t n_rows, int64_t batch_size, const double *features, void *params,
                     void *grads, void *m, void *v, const bamd_adam *hp, double *loss_accum, int64_t *steps_out, void *stream) {
    BAMD_REQUIRE(h && hp, "null argument");
    BAMD_REQUIRE(batch_size > 0 && n_rows >= 0 && (x || n_rows == 0), "bad arguments");
    BAMD_REQUIRE(x_dtype == BAMD_F32 || x_dtype == BAMD_F64, "bad dtype");
    const size_t row_bytes = (size_t)h->dims[0] * (x_dtype == BAMD_F64 ? 8 : 4);
    bamd_adam step_hp = *hp;
    int64_t steps = 0;
    for (int64_t r0 = 0; r0 < n_rows; r0 += batch_size, ++steps) {
        const int64_t rows = n_rows - r0 < batch_size ? n_rows - r0 : batch_size;
        step_hp.step = hp->step + steps;
        const int rc = bamd_train_step(h, (const char *)x + (size_t)r0 * row_bytes, x_dtype, rows, features, params, grads, m, v, &step_hp,
                                       loss_accum, stream);
        if (rc) return rc;
    }
    if (steps_out) *steps_out = steps;
    return BAMD_OK;
}

int bamd_train_epoch_dp(bamd_handle *h, const void *x, int x_dtype, const int64_t *batch_rows, int64_t n_batches, const double *features,
                        void *params, void *grads, void *m, void *v, const bamd_adam *hp, double *loss_accum, void *stream) {
    BAMD_REQUIRE(h && hp, "null argument");
    BAMD_REQUIRE(n_batches >= 0 && (batch_rows || n_batches == 0), "bad arguments");
    BAMD_REQUIRE(x_dtype == BAMD_F32 || x_dtype == BAMD_F64, "bad dtype");
    const size_t row_bytes = (size_t)h->dims[0] * (x_dtype == BAMD_F64 ? 8 : 4);
    bamd_adam step_hp = *hp;
    int64_t r0 = 0;
    for (int64_t b = 0; b < n_batches; ++b) {
        const int64_t rows = batch_rows[b];
        BAMD_REQUIRE(rows >= 0 && (x || rows == 0), "bad batch_rows entry");
        step_hp.step = hp->step + b;
        const int rc = bamd_train_step(h, (const char *)x + (size_t)r0 * row_bytes, x_dtype, rows, features, params, grads, m, v, &step_hp,
                                       loss_accum, stream);
        if (rc) return rc;
        r0 += rows;
    }
    return BAMD_OK;
}

int bamd_emd_rows(const void *x, const void *recon, int dtype, int64_t n_rows, int n_cols, double *out,
                  void *stream) {
    return launch_emd_rows(x, recon, dtype, n_rows, n_cols, out, (hipStream_t)stream);
}

int bamd_error_deltas(const void *x, const void *recon, int dtype, int64_t n_elems, double bound, uint8_t *flags,
                      uint16_t *deltas, void *stream) {
    BAMD_REQUIRE(dtype == BAMD_F32 || dtype == BAMD_F64, "bad dtype");
    return launch_error_deltas(x, recon, dtype, n_elems, bound, flags, deltas, (hipStream_t)stream);
}

int bamd_apply_deltas(void *out, int dtype, int n_cols, const int64_t *rows, const int32_t *cols, const uint16_t *deltas,
                      int64_t count, void *stream) {
    BAMD_REQUIRE(dtype == BAMD_F32 || dtype == BAMD_F64, "bad dtype");
    return launch_apply_deltas(out, dtype, n_cols, rows, cols, deltas, count, (hipStream_t)stream);
}

int bamd_activation_means(bamd_handle *h, const void *x, int x_dtype, int64_t n_rows, const double *features,
                          double *out, int max_nodes, void *stream) {
    BAMD_CHECK_MODEL(h);
    BAMD_REQUIRE(x && out && n_rows > 0, "bad arguments");
    return generic_activation_means(h, x, x_dtype, n_rows, features, out, max_nodes, (hipStream_t)stream);
}

}  // extern "C"
