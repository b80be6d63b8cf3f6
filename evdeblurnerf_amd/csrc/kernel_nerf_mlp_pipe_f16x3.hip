// split-float16 (three MFMA products, float32-grade) instantiations of the software-pipelined NeRF MLP kernel
// (nerf_mlp_kernel.h): one wavefront of 32 samples per SIMD (its B fragments take 8 registers per k-step).
#include "nerf_mlp_kernel.h"

namespace evd {

int launch_nerf_pipe_f16x3(bool feat, const MlpParams& p, hipStream_t st) {
    return feat ? launch_pipe_mlp<EVD_PREC_F16X3, 256, 8, 4, 1, 256, true>(p, st) : launch_pipe_mlp<EVD_PREC_F16X3, 256, 8, 4, 1, 256, false>(p, st);
}

}  // namespace evd
