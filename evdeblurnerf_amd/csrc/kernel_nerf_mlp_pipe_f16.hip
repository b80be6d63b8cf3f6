// f16 instantiations of the software-pipelined NeRF MLP kernel (nerf_mlp_kernel.h): the reference network
// (netdepth 8, netwidth 256, skips [4]), two wavefronts of 32 samples per SIMD.  (One wavefront of 64 samples per SIMD,
// <.., 2, 256, ..>, halves the LDS reads and runs at the same speed: the kernel is clock-limited)
#include "nerf_mlp_kernel.h"

namespace evd {

int launch_nerf_pipe_f16(bool feat, const MlpParams& p, hipStream_t st) {
    return feat ? launch_pipe_mlp<EVD_PREC_F16, 256, 8, 4, 1, 512, true>(p, st) : launch_pipe_mlp<EVD_PREC_F16, 256, 8, 4, 1, 512, false>(p, st);
}

}  // namespace evd
