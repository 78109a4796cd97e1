// fp64 inference kernels of the 64 .. 127-column classes and of latents 32 .. 63 (Impl64Q in fused64.hip; see fused64_infer.hpp).
#include "fused64_infer.hpp"

namespace bamd {

int fused64j_infer_launch(int F, int Z, bool rt, bamd_handle *h, const double *packed, int kind, const void *x, int x_dtype, int64_t n,
                          const double *features, void *out, int out_dtype, const double *renorm, const uint8_t *imask, double *loss_sum,
                          hipStream_t s) {
#define I_CASE(F_, Z_, RT_) if (F == F_ && Z == Z_ && rt == RT_) return infer64_run<F_, Z_, RT_>(h, packed, kind, x, x_dtype, n, features, out, out_dtype, renorm, imask, loss_sum, s);
    I_CASE(79, 31, true) I_CASE(95, 31, true) I_CASE(111, 31, true) I_CASE(127, 31, true)
    I_CASE(63, 63, true) I_CASE(79, 63, true) I_CASE(95, 63, true) I_CASE(111, 63, true) I_CASE(127, 63, true)
#undef I_CASE
    set_error("fp64 fused inference: no instantiation for this shape");
    return BAMD_ERR_UNSUPPORTED;
}

}  // namespace bamd
