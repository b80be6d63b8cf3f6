// f16 training variant of the pipelined NeRF MLP kernel: the same instruction stream as kernel_nerf_mlp_pipe_f16.hip plus
// one 16-byte store per lane for every completed activation fragment (nerf_mlp.h: namespace astore).
#include "nerf_mlp_kernel.h"

namespace evd {

int launch_nerf_train_fwd_f16(const MlpParams& p, hipStream_t st) { return launch_pipe_mlp<EVD_PREC_F16, 256, 8, 4, 1, 512, false, true>(p, st); }

}  // namespace evd
