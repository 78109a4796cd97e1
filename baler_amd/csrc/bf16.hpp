// bf16 inference mode (bf16.hip): encode / decode / forward_loss on v_mfma_f32_16x16x32_bf16 with LDS-resident weights.
#pragma once
#include "bamd_internal.hpp"

namespace bamd {
bool bf16_has_kernels(const bamd_handle *h);     // this shape has a bf16 inference instantiation in bf16.hip
int bf16_setup(bamd_handle *h);                  // BAMD_ERR_UNSUPPORTED for shapes without an instantiation
int bf16_pack(bamd_handle *h, hipStream_t s);    // h->params (fp32) -> bf16 fragments + fp32 bias fragments
void bf16_teardown(bamd_handle *h);
int bf16_encode(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *z, int z_dtype,
                hipStream_t s);
int bf16_decode(bamd_handle *h, const void *z, int z_dtype, int64_t n, const double *features, const uint8_t *int_mask,
                void *out, int out_dtype, hipStream_t s);
int bf16_forward_loss(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *recon,
                      int recon_dtype, double *loss_sum, hipStream_t s);
// bf16 training (bf16_train.hip): fwd + loss + bwd on bf16 MFMA for the shapes it is instantiated for
int bf16_train_setup(bamd_handle *h);            // leaves h->bf16_train_state null when the shape has no instantiation
void bf16_train_teardown(bamd_handle *h);
bool bf16_train_ok(const bamd_handle *h);
int bf16_train_pack(bamd_handle *h, hipStream_t s);   // h->params (fp32) -> the training kernels' bf16 fragments
int bf16_fwd_bwd(bamd_handle *h, const void *x, int x_dtype, int64_t n, const double *features, void *grads, hipStream_t s);
}  // namespace bamd
