// bf16 instantiation of the AWP sample-feature embedding kernels (awp_embed_kernel.h).
#include "awp_embed_kernel.h"

namespace evd {

int launch_awp_embed_bf16(bool train, const AwpFwdParams& p, hipStream_t st) { return launch_awp_embed<EVD_PREC_BF16>(train, p, st); }
int run_awp_backward_bf16(const AwpBwdPlan& b, hipStream_t st) { return run_awp_backward<EVD_PREC_BF16>(b, st); }

}  // namespace evd
