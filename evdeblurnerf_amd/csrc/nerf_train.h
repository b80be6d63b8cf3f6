// Launchers of the NeRF-mode training kernels (kernel_nerf_train_*.hip), called by evd_train_api.hip.
#pragma once

#include "evd_common.h"
#include "nerf_mlp.h"

namespace evd {

int launch_nerf_train_fwd_f16(const MlpParams& p, hipStream_t st);
int launch_nerf_train_fwd_bf16(const MlpParams& p, hipStream_t st);

}  // namespace evd
