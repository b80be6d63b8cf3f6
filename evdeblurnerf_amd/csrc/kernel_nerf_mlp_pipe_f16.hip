// f16 instantiations of the software-pipelined NeRF MLP kernel (nerf_mlp_kernel.h): the reference network
// (netdepth 8, netwidth 256, skips [4]).  Variant 0: two wavefronts of 32 samples per SIMD; variant 1
// (EVD_MLP_VARIANT=1): one wavefront of 64 samples per SIMD -- every A fragment read from LDS feeds two MFMAs.
#include <cstdlib>

#include "nerf_mlp_kernel.h"

namespace evd {

int launch_nerf_pipe_f16(bool feat, const MlpParams& p, hipStream_t st) {
    static const int variant = [] { const char* e = getenv("EVD_MLP_VARIANT"); return e ? atoi(e) : 0; }();
    if (variant == 1 && !feat) return launch_pipe_mlp<EVD_PREC_F16, 256, 8, 4, 2, 256, false>(p, st);
    return feat ? launch_pipe_mlp<EVD_PREC_F16, 256, 8, 4, 1, 512, true>(p, st) : launch_pipe_mlp<EVD_PREC_F16, 256, 8, 4, 1, 512, false>(p, st);
}

}  // namespace evd
