// Fused register-chained kernels for narrow autoencoders (fused.hip).
#pragma once
#include "bamd_internal.hpp"

namespace bamd {
int fused_setup(bamd_handle *h);                 // decides h->fused_ok, allocates the packed weights
int fused_pack(bamd_handle *h, hipStream_t s);   // h->params -> h->packed (no-op when !fused_ok)
void fused_teardown(bamd_handle *h);
bool fused_trains(const bamd_handle *h);                   // false: inference-only class, training runs on generic.hip
int64_t fused_latency_rows(const bamd_handle *h);          // rows up to which training steps run on the small-batch kernels (0: no fused path)
bool fused_serves_bf16_inference(const bamd_handle *h);   // BF16 handle of a wide model (no bf16.hip state)
bool fused_has_bf16_kernels(const bamd_handle *h);        // the shape is one of the wide models with bf16 kernels in fused.hip (before setup)
void fused_params_changed(bamd_handle *h);       // an optimiser step changed h->params / h->packed: lazily refreshed copies are stale
// scatter lists (CSR over parameters) into h->packed for the fused Adam+pack kernel; all null when !fused_ok
void fused_scatter(bamd_handle *h, const int **sc_off, const int **sc_idx, void **packed);
int fused_encode(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *z,
                 int z_dtype, hipStream_t s);
int fused_decode(bamd_handle *h, const void *z, int z_dtype, int64_t n, const double *features,
                 const uint8_t *int_mask, void *out, int out_dtype, hipStream_t s);
int fused_forward_loss(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features,
                       void *recon, int recon_dtype, double *loss_sum, hipStream_t s);
int fused_fwd_bwd(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads,
                  hipStream_t s);
// fwd + loss + bwd + Adam (+ re-pack) of one small batch in two launches; BAMD_ERR_UNSUPPORTED when this handle /
// batch size has no such path (the caller then runs fused_fwd_bwd / generic_fwd_bwd followed by launch_adam)
int fused_train_step(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads,
                     void *params, void *m, void *v, const bamd_adam &hp, double *loss_accum, hipStream_t s);
// Wide models (CFD_dense_AE(2500, 25), the 512-column model) in the layer-wise training pass of generic.hip: the row-local work in
// two launches.  forward: x -> activations y[1..7] (row-major float32), dz_last = 2 (recon - x) / F, one loss partial per
// workgroup (*nblk of them); backward: dz[7] -> dz[6..0] (dL/d pre-activation of every layer).  The weight gradients stay GEMMs.
bool fused_wide_train(const bamd_handle *h);
int fused_wide_train_forward(bamd_handle *h, const float *x, int64_t rows, float *const *y, float *dz_last, double *loss_part, int *nblk,
                             hipStream_t s);
void fused_wide_set_dz16(bamd_handle *h, bool on);
bool fused_wide_small(const bamd_handle *h, int64_t rows);      // this batch runs on the split float32 launches (also on a BF16 handle)
int fused_wide_train_backward(bamd_handle *h, int64_t rows, float *const *y, float *const *dz, const float *dz_latent, hipStream_t s);
// fp64 small-batch step (fused64.hip): chain + weight-gradient tiles on v_mfma_f64_16x16x4_f64 for BAMD_MODE_F64 handles
int fused64_setup(bamd_handle *h);               // leaves h->fused64_state null for shapes without an instantiation
void fused64_teardown(bamd_handle *h);
int fused64_pack(bamd_handle *h, hipStream_t s); // h->params (fp64) -> fragment-packed fp64 weights
void fused64_scatter(bamd_handle *h, const int **sc_off, const int **sc_idx, void **packed);   // for the fused Adam + pack kernel
// fwd + loss + bwd (hp == nullptr) or the whole training step (hp != nullptr) of a small batch; BAMD_ERR_UNSUPPORTED when this
// handle / batch size has no such path (the caller then runs the layer-wise kernels)
// kind: 0 = encode, 1 = decode, 2 = forward + loss; BAMD_ERR_UNSUPPORTED when the shape has no fp64 fused kernels
int fused64_infer(bamd_handle *h, int kind, const void *x, int x_dtype, int64_t n, const double *features, void *out, int out_dtype,
                  const double *renorm, const uint8_t *int_mask, double *loss_sum, hipStream_t s);
int fused64_step(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads, void *params, void *m,
                 void *v, const bamd_adam *hp, double *loss_accum, hipStream_t s);
}  // namespace bamd
