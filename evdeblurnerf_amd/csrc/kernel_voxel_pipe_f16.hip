// f16 instantiations of the software-pipelined PDRF fine-level network (voxel_mlp_kernel.h).
#include "voxel_mlp_kernel.h"

namespace evd {

int launch_voxel_pipe_f16(bool feat, const VoxMlpParams& p, hipStream_t st) {
    return feat ? launch_voxel_pipe<EVD_PREC_F16, true>(p, st) : launch_voxel_pipe<EVD_PREC_F16, false>(p, st);
}

}  // namespace evd
