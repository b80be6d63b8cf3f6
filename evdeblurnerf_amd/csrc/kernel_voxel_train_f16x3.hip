// split-float16 (float32-grade) instantiations of the PDRF training kernels, both levels (voxel_mlp_kernel.h TRAIN variant,
// voxel_train_kernel.h): (hi, lo) fragments in 2 KiB slots, 3-MFMA products in the forward, the dgrad and the wgrad kernels.
#include "voxel_train_kernel.h"

namespace evd {

int launch_voxel_train_fwd_f16x3(int HD, const VoxMlpParams& p, hipStream_t st) {
    if (HD == 256) return p.feature ? launch_voxel_train_fwd<EVD_PREC_F16X3, 256, 128, 64, true>(p, st) : launch_voxel_train_fwd<EVD_PREC_F16X3, 256, 128, 64>(p, st);
    if (p.feature) return fail(EVD_E_INVALID, "evd_voxel_mlp_train: the feature output is built for the fine level (geo 128)");
    return launch_voxel_train_fwd<EVD_PREC_F16X3, 64, 15, 32>(p, st);
}

// training forward of EVD_PREC_F16C on a level without a compensated-float16 kernel (the 64-wide coarse level): this arithmetic, the
// float16 mode's store (hi halves + bit masks from the float32-grade pre-activations)
int launch_voxel_train_fwd_f16x3_hi(int HD, const VoxMlpParams& p, hipStream_t st) {
    if (p.feature) return fail(EVD_E_INVALID, "evd_voxel_mlp_train: the feature output of EVD_PREC_F16C training stays in the store (fragments)");
    return HD == 256 ? launch_voxel_train_fwd<EVD_PREC_F16X3, 256, 128, 64, false, true>(p, st) : launch_voxel_train_fwd<EVD_PREC_F16X3, 64, 15, 32, false, true>(p, st);
}

int run_voxel_backward_f16x3(int HD, const VoxBwdPlan& b, hipStream_t st) {
    return HD == 256 ? run_voxel_backward<EVD_PREC_F16X3, 256, 128, 64>(b, st) : run_voxel_backward<EVD_PREC_F16X3, 64, 15, 32>(b, st);
}

int launch_voxel_coarse_pipe_f16x3(const VoxMlpParams& p, hipStream_t st) { return launch_voxel_resident_level<EVD_PREC_F16X3, 64, 15, 32>(p, st); }

}  // namespace evd
