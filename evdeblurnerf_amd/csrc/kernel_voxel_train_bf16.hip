// bf16 instantiations of the PDRF training kernels, both levels (voxel_mlp_kernel.h TRAIN variant, voxel_train_kernel.h).
#include "voxel_train_kernel.h"

namespace evd {

int launch_voxel_train_fwd_bf16(int HD, const VoxMlpParams& p, hipStream_t st) {
    if (HD == 256) return p.feature ? launch_voxel_train_fwd<EVD_PREC_BF16, 256, 128, 64, true>(p, st) : launch_voxel_train_fwd<EVD_PREC_BF16, 256, 128, 64>(p, st);
    if (p.feature) return fail(EVD_E_INVALID, "evd_voxel_mlp_train: the feature output is built for the fine level (geo 128)");
    return launch_voxel_train_fwd<EVD_PREC_BF16, 64, 15, 32>(p, st);
}

int run_voxel_backward_bf16(int HD, const VoxBwdPlan& b, hipStream_t st) {
    return HD == 256 ? run_voxel_backward<EVD_PREC_BF16, 256, 128, 64>(b, st) : run_voxel_backward<EVD_PREC_BF16, 64, 15, 32>(b, st);
}

int launch_voxel_coarse_pipe_bf16(const VoxMlpParams& p, hipStream_t st) { return launch_voxel_resident_level<EVD_PREC_BF16, 64, 15, 32>(p, st); }

}  // namespace evd
