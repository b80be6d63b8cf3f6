// split-float16 instantiations of the software-pipelined PDRF fine-level network (voxel_mlp_kernel.h).
#include "voxel_mlp_kernel.h"

namespace evd {

int launch_voxel_pipe_f16x3(bool feat, const VoxMlpParams& p, hipStream_t st) {
    return feat ? launch_voxel_pipe<EVD_PREC_F16X3, true>(p, st) : launch_voxel_pipe<EVD_PREC_F16X3, false>(p, st);
}

}  // namespace evd
