// C-ABI entry points of the NeRF-mode training path: forward that keeps the activations, backward of the fused MLP
// (reference: the autograd graph of networks/nerf.py:46-72 NeRF.mlpforward, driven by run_nerf.py:1032-1036 loss.backward()).
#include "evd_common.h"
#include "nerf_mlp.h"
#include "nerf_net.h"
#include "nerf_train.h"

using namespace evd;

namespace evd {
constexpr int TRAIN_WG_SAMPLES = 256;       // samples per workgroup of the training kernels (8 wavefronts x 32)
static long train_tiles(long nsamp) { return cdiv(nsamp, (long)TRAIN_WG_SAMPLES) * (TRAIN_WG_SAMPLES / 32); }
static bool train_built(const evd_nerf* n, int prec) { return (prec == EVD_PREC_F16 || prec == EVD_PREC_BF16) && n->pipe_chunks[prec] > 0; }
}  // namespace evd

extern "C" {

size_t evd_nerf_train_store_bytes(long nsamp) { return nsamp < 0 ? 0 : (size_t)train_tiles(nsamp) * astore::TILE_BYTES; }

int evd_nerf_mlp_train(const evd_nerf* net, int precision, const float* ray_batch, const float* z, long R, int S, float* raw,
                       void* store, size_t store_bytes, void* stream) {
    EVD_REQUIRE(net && ray_batch && z && raw && store, "evd_nerf_mlp_train: null argument");
    EVD_REQUIRE(precision >= 0 && precision < EVD_NUM_PREC && train_built(net, precision),
                "evd_nerf_mlp_train: the training path is built for precision f16 / bf16 on the netdepth 8, netwidth 256, skips [4] network");
    EVD_REQUIRE(R >= 0 && S >= 1, "evd_nerf_mlp_train: bad shape R=%ld S=%d", R, S);
    if (R == 0) return EVD_OK;
    const long nsamp = R * (long)S;
    if (store_bytes < evd_nerf_train_store_bytes(nsamp))
        return fail(EVD_E_WORKSPACE, "evd_nerf_mlp_train: store %zu < %zu bytes", store_bytes, evd_nerf_train_store_bytes(nsamp));
    MlpParams p;
    p.wstream = (const char*)net->pipe[precision].p;
    p.bias = (const float*)net->bias.p;
    p.ray_batch = ray_batch; p.z = z; p.nsamp = nsamp; p.S = S; p.ncol = 11;
    p.D = net->D; p.skip = net->skip; p.nchunks = net->pipe_chunks[precision]; p.nbias = (int)(net->bias.bytes / sizeof(float));
    p.raw = raw; p.feature = nullptr; p.feature_kind = 0; p.act = (char*)store;
    return precision == EVD_PREC_F16 ? launch_nerf_train_fwd_f16(p, as_stream(stream)) : launch_nerf_train_fwd_bf16(p, as_stream(stream));
}

}  // extern "C"
