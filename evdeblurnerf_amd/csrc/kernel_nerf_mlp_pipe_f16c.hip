// Compensated float16 mode (EVD_PREC_F16C: one float16 product + two block-scaled fp6 products of the operands' rounding residuals)
// of the software-pipelined NeRF MLP kernel (nerf_mlp_c_kernel.h): the reference network (netdepth 8, netwidth 256, skips [4]),
// one wavefront of 32 samples per SIMD.
#include "nerf_mlp_c_kernel.h"

namespace evd {

int nerf_mlp_c_chunks(int W, int D, int skip) { return nerf_c_built(W, D, skip) ? nerf_c_chunks<256, 8, 4>() : 0; }

int nerf_mlp_c_dispatch(int W, int D, int skip, const MlpParams& p, hipStream_t st) {
    if (nerf_c_built(W, D, skip)) return p.fuse ? launch_nerf_c<256, 8, 4, true>(p, st) : launch_nerf_c<256, 8, 4, false>(p, st);
    return fail(EVD_E_INVALID, "evd_nerf_mlp: EVD_PREC_F16C is built for netdepth 8, netwidth 256, skips [4] only");
}

}  // namespace evd
