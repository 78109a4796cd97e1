#include "fused.hpp"

namespace bamd {
int fused_setup(bamd_handle *h) { h->fused_ok = false; return BAMD_OK; }
int fused_pack(bamd_handle *, hipStream_t) { return BAMD_OK; }
int fused_encode(bamd_handle *, const void *, int, int64_t, const double *, void *, int, hipStream_t) { return BAMD_ERR_UNSUPPORTED; }
int fused_decode(bamd_handle *, const void *, int, int64_t, const double *, const uint8_t *, void *, int, hipStream_t) { return BAMD_ERR_UNSUPPORTED; }
int fused_forward_loss(bamd_handle *, const void *, int, int64_t, const double *, void *, int, double *, hipStream_t) { return BAMD_ERR_UNSUPPORTED; }
int fused_fwd_bwd(bamd_handle *, const void *, int, int64_t, const double *, void *, hipStream_t) { return BAMD_ERR_UNSUPPORTED; }
}  // namespace bamd
