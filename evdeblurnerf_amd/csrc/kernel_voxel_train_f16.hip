// f16 instantiations of the PDRF training kernels, both levels (voxel_mlp_kernel.h TRAIN variant, voxel_train_kernel.h).
#include "voxel_train_kernel.h"

namespace evd {

int launch_voxel_train_fwd_f16(int HD, const VoxMlpParams& p, hipStream_t st) {
    if (HD == 256) return p.feature ? launch_voxel_train_fwd<EVD_PREC_F16, 256, 128, 64, true>(p, st) : launch_voxel_train_fwd<EVD_PREC_F16, 256, 128, 64>(p, st);
    if (p.feature) return fail(EVD_E_INVALID, "evd_voxel_mlp_train: the feature output is built for the fine level (geo 128)");
    return launch_voxel_train_fwd<EVD_PREC_F16, 64, 15, 32>(p, st);
}

int run_voxel_backward_f16(int HD, const VoxBwdPlan& b, hipStream_t st) {
    return HD == 256 ? run_voxel_backward<EVD_PREC_F16, 256, 128, 64>(b, st) : run_voxel_backward<EVD_PREC_F16, 64, 15, 32>(b, st);
}

int voxel_store_geo_slot(int HD) { return HD == 256 ? VStore<256, 128, 64>::GEO : VStore<64, 15, 32>::GEO; }
long voxel_store_tile_bytes_prec(int HD, int prec) { return HD == 256 ? VStore<256, 128, 64>::tile_bytes(prec) : VStore<64, 15, 32>::tile_bytes(prec); }
long voxel_store_tile_bytes(int HD) { return HD == 256 ? VStore<256, 128, 64>::TILE_BYTES : VStore<64, 15, 32>::TILE_BYTES; }

int launch_voxel_coarse_pipe_f16(const VoxMlpParams& p, hipStream_t st) { return launch_voxel_resident_level<EVD_PREC_F16, 64, 15, 32>(p, st); }

}  // namespace evd

#ifdef EVD_WD_STAMP     // developer build (tools/dev/stamp_wgrad_dgrad.sh): the phase stamps of the last k_wgrad_dgrad launch of a kind
extern "C" int evd_debug_wd_stamps(float* host_out, int kind) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(evd::g_wd_stamp), 2048 * 8 * sizeof(float), (size_t)kind * 2048 * 8 * sizeof(float), hipMemcpyDeviceToHost);
}
#endif
