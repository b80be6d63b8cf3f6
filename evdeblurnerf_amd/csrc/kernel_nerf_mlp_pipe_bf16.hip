// bf16 instantiations of the software-pipelined NeRF MLP kernel (nerf_mlp_kernel.h): the reference network
// (netdepth 8, netwidth 256, skips [4]), two wavefronts of 32 samples per SIMD.
#include "nerf_mlp_kernel.h"

namespace evd {

int launch_nerf_pipe_bf16(bool feat, const MlpParams& p, hipStream_t st) {
    return feat ? launch_pipe_mlp<EVD_PREC_BF16, 256, 8, 4, 1, 512, true>(p, st) : launch_pipe_mlp<EVD_PREC_BF16, 256, 8, 4, 1, 512, false>(p, st);
}

}  // namespace evd
