// Compensated float16 mode (EVD_PREC_F16C) of the PDRF fine-level network (voxel_mlp_c_kernel.h): hidden 256, geo 128, 64 feature
// channels in -- the level every shipped configuration uses; one wavefront of 32 samples per SIMD.
#include "voxel_mlp_c_kernel.h"

namespace evd {

int voxel_mlp_c_chunks(int HD, int G, int FT) { return voxel_c_built(HD, G, FT) ? VoxNetC<256, 128, 64>::NCH : 0; }

int launch_voxel_pipe_f16c(const VoxMlpParams& p, hipStream_t st) { return launch_voxel_c<256, 128, 64>(p, st); }

}  // namespace evd
