/*
 * baler_amd.h -- C ABI of libbaler_amd.so, the MI355X (gfx950) hot path of Baler's dense
 * autoencoder: train (forward + sum-of-squares loss + backward + Adam), encode (compress) and
 * decode (decompress), plus the per-column min-max (un)normalisation either side of it.
 *
 * The reference (baler-collaboration/baler @ 2024_10_08) has NO native/FFI boundary for this path:
 * the path sits behind Python call surfaces.  Each entry point below names
 * the reference interface it replaces (file:line, relative to the reference checkout); the Python
 * binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - every function returns 0 on success or a negative bamd_status; no exceptions cross the ABI;
 *    bamd_last_error() returns a message for the calling thread's last failure;
 *  - the CALLER owns every device buffer passed in (dataset, params, grads, Adam m/v, outputs);
 *    the library borrows the pointers for the duration of the call and owns only its handle
 *    (layer descriptor, MFMA-fragment-packed weight copy, partial-gradient slabs, activation
 *    workspace);
 *  - all work is enqueued asynchronously on the hipStream_t passed as `stream` (void*; NULL =
 *    the default stream); nothing synchronises the host;
 *  - one handle per (model, device); a handle is not thread-safe, and it is SINGLE-STREAM: the library re-packs its
 *    weight fragments lazily, on the stream of the first call that needs them after an optimiser step, so calls on one
 *    handle must be issued on one stream (or the caller orders its streams with events around every call);
 *  - parameter vectors are FLAT in Baler's state-dict order: for each layer l, W_l[out][in]
 *    row-major then b_l[out] (models.py:128-136; the key order of model.pt);
 *  - rows are independent: multi-GPU = one process and one handle per GPU, rows sharded by the
 *    caller; the only exchange is a SUM all-reduce of the flat gradient buffer between
 *    bamd_fwd_bwd() and bamd_adam_step() (done by the caller with RCCL).
 */
#ifndef BALER_AMD_H
#define BALER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BAMD_ABI_VERSION 1

typedef struct bamd_handle bamd_handle;

typedef enum bamd_status {
    BAMD_OK = 0,
    BAMD_ERR_INVALID = -1,     /* bad argument (null pointer, unsupported dtype/shape ...) */
    BAMD_ERR_NO_DEVICE = -2,   /* no HIP device / wrong architecture */
    BAMD_ERR_HIP = -3,         /* a HIP runtime call failed; see bamd_last_error() */
    BAMD_ERR_ALLOC = -4,       /* device allocation failed */
    BAMD_ERR_UNSUPPORTED = -5  /* feature not available in this compute mode */
} bamd_status;

/* element type of a caller buffer */
typedef enum bamd_dtype { BAMD_F32 = 0, BAMD_F64 = 1 } bamd_dtype;

/* arithmetic the Linear layers run in.  F32 = v_mfma_f32_16x16x4_f32 (exact fp32, the parity mode:
 * outputs within 1e-5 rel. of the fp64 reference).  F64 = v_mfma_f64_16x16x4_f64 (long-horizon
 * training parity).  BF16 = v_mfma_f32_16x16x32_bf16 with fp32 accumulation: a THROUGHPUT mode with its own 2e-2 bar.
 * Inference (bamd_encode / bamd_decode / bamd_forward_loss): weights rounded from the fp32 master copy and kept in LDS,
 * 4-5x the F32 rate (measured 2e-3 encode, 6e-3 decode).  Training (bamd_fwd_bwd / bamd_train_step): bf16 weights,
 * activations and gradients through the chain, fp32 accumulation and fp32 partial gradients; params / m / v / grads stay
 * FLOAT (the caller's fp32 master copy and Adam state); bamd_adam_step re-rounds the bf16 fragments.  3.3x the F32 training
 * rate; one-step gradients within ~5e-3 rel-L2 of the fp64 reference; batches of <= 3072 rows run on the F32 small-batch
 * kernels instead (exact fp32, and faster there: 18 vs 34 us per 512-row step).  Available for the 24-column AE (every fused latent
 * size) and for the wide models CFD_dense_AE(2500, 25), CFD_dense_AE(625, 7) and the 512-column model: their two wide layers on
 * the bf16 MFMA in bamd_encode / bamd_decode (HBM-bound) and in the training pass (forward of en1 / de4, de4's input-gradient
 * product AND the weight gradients of those two wide layers: operands rounded to bfloat16, float32 accumulation and float32
 * partial sums; only the six narrow layers, the loss, the masks and the narrow layers' weight gradients stay fp32 on the
 * activations the pass stores -- dL/drecon as bfloat16 where both of its readers take it that way: one-step gradients within
 * ~1e-3 rel-L2, which is the rounding of those wide operands; BALER_AMD_BF16_WIDE_TRAIN=0: the F32 launches); validation of such a handle runs
 * in fp32; any OTHER shape asked for in BAMD_MODE_BF16 is created as a float32 handle (run-time-width fused classes or the layer-wise
 * kernels, whatever serves the shape in BAMD_MODE_F32) with a notice on stderr -- bamd_mode_of() then returns BAMD_MODE_F32.  bamd_activation_means of a BF16 handle runs on the fp32
 * layer-wise kernels. */
typedef enum bamd_mode { BAMD_MODE_F32 = 0, BAMD_MODE_F64 = 1, BAMD_MODE_BF16 = 2 } bamd_mode;

/* Adam hyper-parameters of one step (torch.optim.Adam defaults are beta1=.9 beta2=.999 eps=1e-8). */
typedef struct bamd_adam {
    int64_t step;   /* t AFTER the increment (first step = 1) */
    double lr;      /* current learning rate (host-side ReduceLROnPlateau feeds this) */
    double beta1, beta2, eps;
} bamd_adam;

int bamd_abi_version(void);
const char *bamd_last_error(void);
/* number of visible HIP devices, or a negative bamd_status */
int bamd_device_count(void);

/* ---- model handle ------------------------------------------------------------------------------
 * Replaces: models.AE.__init__ / models.CFD_dense_AE.__init__ (models.py:122-139, 192-209) and
 * data_processing.initialise_model/load_model (data_processing.py:76-110).
 * dims has n_layers+1 entries (n_features, 200, 100, 50, z, 50, 100, 200, n_features for the
 * reference topologies; any even n_layers >= 2 is accepted).  LeakyReLU(0.01) follows every layer
 * except the last encoder layer and the last decoder layer (models.py:141-152). */
int bamd_create(const int *dims, int n_layers, int mode, int device, bamd_handle **out);
void bamd_destroy(bamd_handle *h);

/* Which kernels serve this handle's throughput calls (bamd_encode / bamd_decode / bamd_fwd_bwd at large batches).  The fused
 * register-chained / wide-layer kernels are template instantiations.  EXACT instantiations for the shapes the reference ships configs
 * for: AE(24, z) for z in {15, 12, 10, 8, 6, 5, 4, 3, 2} (models.py:116-183 at the compression ratios of baler.py:117-123),
 * CFD_dense_AE(2500, 25), CFD_dense_AE(625, 7) (exafel1_config.py:14-15,33: 25 x 25 blocks) and the 512-column model; an F64 handle of
 * the 24-column AE has fused fp64 kernels for inference and for training steps of any size.  Every OTHER model with the reference's
 * hidden widths (200-100-50; models.py:122-139, 192-209 build them for any n_features / z_dim) is served by a CLASS instantiation with
 * run-time widths:
 *   - up to 63 columns, latent <= 31: every kernel (BAMD_PATH_FUSED);
 *   - 64 .. 127 columns, latent <= 31: a handle with TWO sets of packed weights -- the small-batch class for training steps up to the
 *     small-batch limit (default 12288 rows; BALER_AMD_LATENCY_ROWS at bamd_create; reference batch_size = 512) and the wide class
 *     below for encode / decode / forward + loss and larger training batches (BAMD_PATH_FUSED).  BALER_AMD_MID_HYBRID=0: the
 *     small-batch class alone, larger batches chunk after chunk on its kernels (BAMD_PATH_FUSED_INFER);
 *   - 48 .. 4096 columns, latent <= 63, that no class above takes: the wide-layer kernels with the column count and the latent as
 *     kernel arguments -- encode / decode / forward + loss fused, a training pass = two fused row-local launches + the layer-wise
 *     weight-gradient kernels, as for the exact wide shapes (BAMD_PATH_FUSED; BALER_AMD_WIDE_CLASS=0 switches the class off);
 *   - F64 handles: class instantiations of the fp64 kernels for up to 63 columns with a latent of up to 31.
 * Any other shape (other hidden widths, more than 4096 columns, a latent above 63) runs on the layer-wise kernels (activations
 * through HBM, 1.5-2.5x slower): bamd_create prints one line to stderr for such a handle unless BALER_AMD_QUIET=1.  There is no model
 * object in the reference to query (models.py builds nn.Linear layers of any width); this call exists so that callers and tests can tell. */
typedef enum bamd_path {
    BAMD_PATH_GENERIC = 0,   /* generic.hip: LDS-tiled MFMA GEMM per layer */
    BAMD_PATH_FUSED = 1,     /* fused.hip: register chain (24-column AE) or streamed wide layers + chain */
    BAMD_PATH_BF16 = 2,      /* bf16.hip / bf16_train.hip (24-column AE, BAMD_MODE_BF16); its small batches use the fused fp32 step */
    BAMD_PATH_FUSED_INFER = 3 /* 64..127 columns with BALER_AMD_MID_HYBRID=0 or BALER_AMD_WIDE_CLASS=0: fused.hip for encode / decode / forward + loss and
                              * the small-batch training kernels (larger batches chunk after chunk on the same kernels) */
} bamd_path;
int bamd_path_of(const bamd_handle *h);   /* a bamd_path, or BAMD_ERR_INVALID for a null handle */
int64_t bamd_param_count(const bamd_handle *h);
int bamd_mode_of(const bamd_handle *h);    /* the mode the handle COMPUTES in (BAMD_MODE_BF16 asked of a shape without bf16 kernels: BAMD_MODE_F32) */

/* (Re)build the handle's MFMA-fragment-packed weight copy from the caller's flat parameter vector
 * (device pointer; dtype F32 or F64).  Call after loading a checkpoint or changing params outside
 * bamd_adam_step().  Replaces: model.load_state_dict (data_processing.py:105-110). */
int bamd_load_params(bamd_handle *h, const void *params, int dtype, void *stream);

/* ---- normalisation -----------------------------------------------------------------------------
 * Replaces: data_processing.find_minmax (data_processing.py:113-130).  features = [min ; max-min],
 * (2, n_cols) float64, device memory. */
int bamd_minmax(const void *x, int dtype, int64_t n_rows, int n_cols, double *features,
                void *stream);
/* The same reduction as bamd_minmax() with the raw extrema as output: minmax = [min ; max], (2, n_cols) float64.
 * For row-sharded tables (one process per GPU): every rank reduces its own rows, the caller combines the ranks
 * with one MIN and one MAX all-reduce of n_cols doubles each and forms range = max - min once -- bit-identical to
 * data_processing.find_minmax (data_processing.py:113-130) over the whole table. */
int bamd_col_minmax(const void *x, int dtype, int64_t n_rows, int n_cols, double *minmax, void *stream);
/* Replaces: helper.normalize -> data_processing.normalize (helper.py:261-274,
 * data_processing.py:133-153): out = (x - min)/(max - min) per column, evaluated in float64 and
 * rounded to out_dtype. */
int bamd_normalize(const void *x, int dtype, int64_t n_rows, int n_cols, const double *features,
                   void *out, int out_dtype, void *stream);
/* Replaces: data_processing.renormalize_func (data_processing.py:188-203) + the per-column
 * astype cast at baler.py:426-435: out = x*range + min (float64), then truncation toward zero for
 * columns with int_mask[c] != 0 (int_mask may be NULL; it is a DEVICE pointer of n_cols bytes). */
int bamd_renormalize(const void *x, int dtype, int64_t n_rows, int n_cols, const double *features,
                     const uint8_t *int_mask, double *out, void *stream);

/* ---- inference ---------------------------------------------------------------------------------
 * Replaces: AE.encode (models.py:141-145) as driven by helper.compress's loop (helper.py:583-611).
 * x: (n_rows, n_features) row-major, x_dtype.  If features != NULL the rows are min-max normalised
 * on load with features = [min ; range] (device, float64) -- the fused form of helper.py:500-504.
 * z: (n_rows, z_dim) row-major, z_dtype. */
int bamd_encode(bamd_handle *h, const void *x, int x_dtype, int64_t n_rows, const double *features,
                void *z, int z_dtype, void *stream);
/* Replaces: AE.decode (models.py:147-152) as driven by helper.decompress (helper.py:700-723), with
 * the optional un-normalise + int-column truncation epilogue of baler.py:420-435 fused in when
 * features != NULL (int_mask may still be NULL). */
int bamd_decode(bamd_handle *h, const void *z, int z_dtype, int64_t n_rows, const double *features,
                const uint8_t *int_mask, void *out, int out_dtype, void *stream);
/* Replaces: AE.forward (models.py:154-156) + utils.mse_sum_loss_l1(validate=True)
 * (utils.py:195-211) as used by training.validate (training.py:104-137).
 * recon may be NULL.  *loss_sum (device, float64) is OVERWRITTEN with sum((recon-x)^2)/n_features
 * of this batch. */
int bamd_forward_loss(bamd_handle *h, const void *x, int x_dtype, int64_t n_rows,
                      const double *features, void *recon, int recon_dtype, double *loss_sum,
                      void *stream);

/* ---- training ----------------------------------------------------------------------------------
 * Replaces: one iteration of training.fit's loop body up to loss.backward() (training.py:64-92):
 * zero_grad, forward, loss = sum((r-x)^2)/n_features, backward.
 * grads: device buffer of bamd_param_count()+1 elements of the handle's parameter type (float for
 * MODE_F32/BF16, double for MODE_F64); it is OVERWRITTEN with the gradient of this batch in
 * state-dict order, and element [param_count] receives the batch loss.  Summation order is fixed
 * (no float atomics): results are bitwise reproducible run to run.
 * Data-parallel use: each rank passes its slice of the global batch, then SUM-all-reduces the
 * whole buffer (the loss is a row sum, so the global-batch gradient is the sum of shard gradients). */
int bamd_fwd_bwd(bamd_handle *h, const void *x, int x_dtype, int64_t n_rows, const double *features,
                 void *grads, void *stream);
/* bamd_fwd_bwd() with an extra term dL/dz injected at the bottleneck: latent_grad is (n_rows, z_dim) of the
 * handle's parameter type and is ADDED to the gradient that reaches the encoder output.  Replaces:
 * loss.backward() of training.py:73-92 when config.custom_loss_function == "loss_function_swae" (the
 * loss then has a term that depends on z = model.encode(x) only).  grads[param_count] still receives the
 * reconstruction loss sum((r-x)^2)/n_cols alone.  Wide-layer models (CFD_dense_AE shapes with a fused path) run it on the fused
 * row-local launches, the latent term added at the bottleneck of the input-gradient chain; other shapes on the layer-wise kernels. */
int bamd_fwd_bwd_latent(bamd_handle *h, const void *x, int x_dtype, int64_t n_rows,
                        const double *features, const void *latent_grad, void *grads, void *stream);
/* Replaces: utils.compute_swd (utils.py:58-77) and its backward: for each of n_proj unit vectors
 * proj[s] (n_proj, z_dim) the projections of z and of prior (both (n_rows, z_dim)) are sorted across
 * the batch; *loss_out = reg_weight * mean_{s,k} (sort(z.P_s)_k - sort(prior.P_s)_k)^2 (float64), and
 * dz_out (n_rows, z_dim, `dtype`) = d loss / d z.  The random draws are the caller's (the reference
 * takes them from torch's global generator, utils.py:59,79-91).  2 <= n_rows <= 1048576 (up to 4096 rows one workgroup per
 * projection sorts in LDS; larger batches, e.g. CFD_project_animation's 6000, take passes through global memory). */
int bamd_swd(const void *z, const void *prior, const void *proj, int dtype, int64_t n_rows, int z_dim,
             int n_proj, double reg_weight, double *loss_out, void *dz_out, void *stream);
/* Replaces: torch.optim.Adam.step + zero_grad (training.py:68,95,266) on the flat buffers, and
 * refreshes the handle's packed weight copy.  params/m/v/grads: device, handle parameter type.
 * If loss_accum != NULL (device, float64): *loss_accum += grads[param_count]  (the running_loss of
 * training.py:97 without the per-step host sync). */
int bamd_adam_step(bamd_handle *h, void *params, const void *grads, void *m, void *v,
                   const bamd_adam *hp, double *loss_accum, void *stream);
/* Replaces: the whole body of the training.fit batch loop (training.py:64-97: zero_grad, forward,
 * loss, backward, optimizer.step, running_loss += loss) in ONE call; exactly bamd_fwd_bwd() followed
 * by bamd_adam_step() on the same arguments, for single-process training (no all-reduce between the
 * two).  Small batches (the reference's batch_size = 512 regime) run as two launches: the layer chain,
 * then one workgroup per weight-gradient tile that also applies Adam to the parameters it owns.
 * grads may be NULL (the gradient is then not materialised); otherwise as in bamd_fwd_bwd(). */
int bamd_train_step(bamd_handle *h, const void *x, int x_dtype, int64_t n_rows, const double *features,
                    void *params, void *grads, void *m, void *v, const bamd_adam *hp,
                    double *loss_accum, void *stream);

/* Replaces: ONE EPOCH of training.fit's batch loop (training.py:64-97) -- `for idx, inputs in enumerate(train_dl)`: sequential
 * batches of `batch_size` rows of the resident table, no shuffling, the partial last batch kept (training.py:237-263) -- in ONE
 * call: the loop runs inside the library, every batch exactly as bamd_train_step() (same kernels, same arguments, hp->step for the
 * first batch and +1 for every following one, one learning rate for the epoch as the reference's per-epoch scheduler gives it),
 * enqueued back to back on `stream` with no host synchronisation and no host round trip per step.  Bit-identical to the loop of
 * bamd_train_step() calls it replaces.  x: (n_rows, n_features) row-major, x_dtype.  *loss_accum += every batch's loss
 * (running_loss of training.py:97); grads (may be NULL) holds the LAST batch's gradient and loss afterwards (the "Training Loss"
 * training.py:100 prints).  *steps_out (host, may be NULL) receives the number of optimiser steps taken = ceil(n_rows / batch_size).
 * Single-process form (uniform batch size); the data-parallel epoch is bamd_train_epoch_dp() below. */
int bamd_train_epoch(bamd_handle *h, const void *x, int x_dtype, int64_t n_rows, int64_t batch_size, const double *features,
                     void *params, void *grads, void *m, void *v, const bamd_adam *hp, double *loss_accum, int64_t *steps_out,
                     void *stream);

/* ---- data-parallel training inside the library ("bmi_allreduce_init/... or pass ncclComm_t"; 8(e)) ----------
 * The reference is single-process; its batch loop (training.py:64-97) becomes data parallel by summing the [grads | loss] vector over
 * the ranks between loss.backward() (:92) and optimizer.step() (:93).  With a communicator attached to the handle, bamd_train_step()
 * and bamd_train_epoch_dp() run that sequence -- bamd_fwd_bwd on this rank's rows, ONE ncclAllReduce(ncclSum) of nparams + 1 elements
 * in place on `grads`, bamd_adam_step -- inside the library on the caller's stream: one host call per step / per epoch instead of three
 * Python -> C -> RCCL round trips per step.  RCCL is resolved at run time (dlopen of the librccl the process already has, else
 * librccl.so.1): the library has no link-time dependency on it, and handles without a communicator never touch it.
 *
 * bamd_comm_unique_id: rank 0 fills `id128` (128 bytes, ncclGetUniqueId) and hands it to the other ranks by any means (the Python
 *   host broadcasts it over its torch.distributed group).
 * bamd_comm_init: every rank, collectively: ncclCommInitRank(world, id, rank) on the handle's device; the handle owns the communicator
 *   (bamd_destroy / bamd_comm_release destroy it).  world = 1 is valid (RCCL on one GPU: the sum over one rank is the identity).
 * bamd_comm_attach: use an EXISTING ncclComm_t (`comm`, e.g. the caller's own) instead; the caller keeps ownership.
 * bamd_comm_release: detach (and destroy an owned communicator); the handle is single-process again.
 * bamd_allreduce_sum: the collective alone, in place on `buf` (count elements of dtype BAMD_F32 / BAMD_F64) -- for callers that keep
 *   the three-call sequence (e.g. the sliced-Wasserstein step) and for timing it. */
int bamd_comm_unique_id(void *id128);
int bamd_comm_init(bamd_handle *h, const void *id128, int rank, int world);
int bamd_comm_attach(bamd_handle *h, void *comm, int world);
int bamd_comm_release(bamd_handle *h);
int bamd_comm_world(const bamd_handle *h);      /* ranks of the attached communicator; 0 without one */
int bamd_allreduce_sum(bamd_handle *h, void *buf, int dtype, int64_t count, void *stream);
/* One epoch of the data-parallel batch loop in ONE call: `x` holds THIS rank's rows of every global batch back to back (the
 * block-cyclic shard of SURVEY 8(e)), batch_rows[i] (host array, n_batches entries, each >= 0: an empty slice contributes a zero
 * gradient) = this rank's rows of global batch i.  Every batch runs fwd_bwd -> all-reduce -> Adam as described above (hp->step for the
 * first, +1 per batch); without a communicator it is the single-process loop with explicit batch sizes (each batch exactly
 * bamd_train_step).  *loss_accum += every GLOBAL batch's loss; grads (may be NULL) holds the last batch's summed gradient and loss. */
int bamd_train_epoch_dp(bamd_handle *h, const void *x, int x_dtype, const int64_t *batch_rows, int64_t n_batches,
                        const double *features, void *params, void *grads, void *m, void *v, const bamd_adam *hp,
                        double *loss_accum, void *stream);

/* ---- diagnostics -------------------------------------------------------------------------------
 * Replaces: the EMD term of utils.mse_loss_emd_l1 (utils.py:112-119): sum over rows of the 1-D
 * Wasserstein distance between the row's column values of x and recon.  *out (device, float64)
 * is overwritten.  Forward-only metric; n_cols <= 64. */
int bamd_emd_rows(const void *x, const void *recon, int dtype, int64_t n_rows, int n_cols,
                  double *out, void *stream);
/* ---- error-bounded deltas side channel (config.save_error_bounded_deltas) -------------------------
 * Replaces: helper.save_error_bounded_requirement (helper.py:442-470) for a whole table at once.
 * x, recon: n_elems values of `dtype` (the normalised input and decode(encode(x)), row-major).
 * flags[i] = |(recon[i]-x[i])/x[i]*100| > bound with numpy's rules (+-inf -> 0, NaN never exceeds);
 * deltas[i] = IEEE binary16 bits of float16(recon[i]) - float16(x[i]) (np.subtract(dtype=float16)).
 * The caller compacts flags/deltas into the per-batch (row, col) lists of helper.py:589-606. */
int bamd_error_deltas(const void *x, const void *recon, int dtype, int64_t n_elems, double bound,
                      uint8_t *flags, uint16_t *deltas, void *stream);
/* Replaces: the delta loop of helper.decompress (helper.py:708-718): for i < count,
 * out[rows[i]][cols[i]] -= float16 deltas[i]; out is (n_rows, n_cols) of `dtype`, row-major, decoder
 * output BEFORE un-normalisation.  rows/cols/deltas are device arrays; (row, col) pairs are unique. */
int bamd_apply_deltas(void *out, int dtype, int n_cols, const int64_t *rows, const int32_t *cols,
                      const uint16_t *deltas, int64_t count, void *stream);
/* Replaces: activation extraction (models.py:160-183, diagnostics.py:10-47): mean over the batch of
 * leaky_relu(pre-activation) for every activated layer; out is (n_layers-2, max_nodes) float64
 * device memory, padded with NaN. */
int bamd_activation_means(bamd_handle *h, const void *x, int x_dtype, int64_t n_rows,
                          const double *features, double *out, int max_nodes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BALER_AMD_H */
