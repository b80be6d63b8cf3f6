// Training forward of the PDRF fine-level network in the compensated float16 mode (voxel_mlp_c_kernel.h, TRAIN variant): the inference
// arithmetic + the activation store of the single-product float16 mode, whose dgrad / wgrad kernels run the level's backward on it.
// (Compiled with a raised pragma-unroll threshold, build.py: the straight-line stream of the TRAIN variant is past hipcc's default.)
#include "voxel_mlp_c_kernel.h"

namespace evd {

int launch_voxel_train_fwd_f16c(const VoxMlpParams& p, hipStream_t st) { return launch_voxel_c<256, 128, 64, true>(p, st); }

}  // namespace evd
