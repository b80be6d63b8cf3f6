// split-float16 (float32-grade) instantiation of the NeRF MLP backward kernels (nerf_train_kernel.h).
#include "nerf_train_kernel.h"

namespace evd {

int run_nerf_backward_f16x3(const BwdPlan& b, hipStream_t st) { return run_nerf_backward<EVD_PREC_F16X3>(b, st); }

}  // namespace evd
