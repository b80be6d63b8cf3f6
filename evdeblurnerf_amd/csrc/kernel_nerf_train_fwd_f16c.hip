// Training forward of the 8 x 256 NeRF in the compensated float16 mode (nerf_mlp_c_kernel.h, TRAIN variant): the inference arithmetic +
// the activation store of the single-product float16 mode, whose dgrad / wgrad kernels run the backward on it.
// (Compiled with a raised pragma-unroll threshold, build.py.)
#include "nerf_mlp_c_kernel.h"

namespace evd {

int launch_nerf_train_fwd_f16c(const MlpParams& p, hipStream_t st) { return launch_nerf_c<256, 8, 4, false, true>(p, st); }

}  // namespace evd
