// split-float16 (float32-grade) training variant of the pipelined NeRF MLP kernel: the instruction stream of
// kernel_nerf_mlp_pipe_f16x3.hip plus the stores of every completed activation fragment as a (hi, lo) pair (nerf_mlp.h astore,
// 2 KiB slots).  The mode that gives the reference's float32 autograd (run_nerf.py:593-601) back to ~1e-5.
#include "nerf_mlp_kernel.h"

namespace evd {

int launch_nerf_train_fwd_f16x3(const MlpParams& p, hipStream_t st) { return launch_pipe_mlp<EVD_PREC_F16X3, 256, 8, 4, 1, 256, false, true>(p, st); }

// EVD_PREC_F16M: this arithmetic in front of the single-product float16 backward -- the float16 mode's store (hi halves, bit masks)
int launch_nerf_train_fwd_f16x3_hi(const MlpParams& p, hipStream_t st) { return launch_pipe_mlp<EVD_PREC_F16X3, 256, 8, 4, 1, 256, false, true, true>(p, st); }

}  // namespace evd
