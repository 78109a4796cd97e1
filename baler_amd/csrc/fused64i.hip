// fp64 inference kernels of the exact 24-column latents and of the classes up to 63 columns (see fused64_infer.hpp).
#include "fused64_infer.hpp"

namespace bamd {

int fused64_infer_launch(int F, int Z, bool rt, bamd_handle *h, const double *packed, int kind, const void *x, int x_dtype, int64_t n,
                         const double *features, void *out, int out_dtype, const double *renorm, const uint8_t *imask, double *loss_sum,
                         hipStream_t s) {
#define I_CASE(F_, Z_, RT_) if (F == F_ && Z == Z_ && rt == RT_) return infer64_run<F_, Z_, RT_>(h, packed, kind, x, x_dtype, n, features, out, out_dtype, renorm, imask, loss_sum, s);
    I_CASE(24, 15, false) I_CASE(24, 12, false) I_CASE(24, 8, false) I_CASE(24, 6, false) I_CASE(24, 10, false)
    I_CASE(24, 5, false) I_CASE(24, 4, false) I_CASE(24, 3, false) I_CASE(24, 2, false)
    I_CASE(31, 15, true) I_CASE(47, 15, true) I_CASE(63, 15, true) I_CASE(31, 31, true) I_CASE(63, 31, true)
#undef I_CASE
    return fused64j_infer_launch(F, Z, rt, h, packed, kind, x, x_dtype, n, features, out, out_dtype, renorm, imask, loss_sum, s);
}

}  // namespace bamd
