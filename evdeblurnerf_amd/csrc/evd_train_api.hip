// C-ABI entry points of the NeRF-mode training path: forward that keeps the activations, backward of the fused MLP
// (reference: the autograd graph of networks/nerf.py:46-72 NeRF.mlpforward, driven by run_nerf.py:593-601 loss.backward()).
#include "evd_common.h"
#include "nerf_mlp.h"
#include "nerf_net.h"
#include "nerf_train.h"

#include <cstdint>
#include <cstdlib>

using namespace evd;

namespace evd {
constexpr int TRAIN_WG_SAMPLES = 256;       // samples per workgroup of the training kernels (8 wavefronts x 32)
static long train_tiles(long nsamp) { return cdiv(nsamp, (long)TRAIN_WG_SAMPLES) * (TRAIN_WG_SAMPLES / 32); }
// the mixed modes (include/evdnerf.h): EVD_PREC_F16C = compensated forward, EVD_PREC_F16M = split-float16 forward; both on the
// float16 mode's store and backward
static bool train_built(const evd_nerf* n, int prec) {
    if (prec == EVD_PREC_F16C) return n->pipe_chunks[EVD_PREC_F16] > 0 && n->pipe_chunks[EVD_PREC_F16C] > 0;
    if (prec == EVD_PREC_F16M) return n->pipe_chunks[EVD_PREC_F16] > 0 && n->pipe_chunks[EVD_PREC_F16X3] > 0;
    return prec >= 0 && prec < EVD_NUM_PREC && is_train_prec(prec) && n->pipe_chunks[prec] > 0;
}
static int store_prec(int prec) { return (prec == EVD_PREC_F16C || prec == EVD_PREC_F16M) ? EVD_PREC_F16 : prec; }
}  // namespace evd


extern "C" {

size_t evd_nerf_train_store_bytes(long nsamp) { return nsamp < 0 ? 0 : (size_t)train_tiles(nsamp) * astore::TILE_BYTES; }
size_t evd_nerf_train_store_bytes_prec(int precision, long nsamp) {
    const int sp = store_prec(precision);
    return (nsamp < 0 || sp < 0 || sp >= EVD_NUM_PREC || !is_train_prec(sp)) ? 0 : (size_t)train_tiles(nsamp) * astore::tile_bytes(sp);
}

int evd_nerf_mlp_train(const evd_nerf* net, int precision, const float* ray_batch, const float* z, long R, int S, float* raw,
                       void* store, size_t store_bytes, void* stream) {
    EVD_REQUIRE(net && ray_batch && z && raw && store, "evd_nerf_mlp_train: null argument");
    EVD_REQUIRE(train_built(net, precision),
                "evd_nerf_mlp_train: the training path is built for precision f16 / bf16 / f16x3 / f16c / f16m on the netdepth 8, netwidth 256, skips [4] network");
    EVD_REQUIRE(R >= 0 && S >= 1, "evd_nerf_mlp_train: bad shape R=%ld S=%d", R, S);
    if (R == 0) return EVD_OK;
    const long nsamp = R * (long)S;
    if (store_bytes < evd_nerf_train_store_bytes_prec(precision, nsamp))
        return fail(EVD_E_WORKSPACE, "evd_nerf_mlp_train: store %zu < %zu bytes", store_bytes, evd_nerf_train_store_bytes_prec(precision, nsamp));
    MlpParams p{};
    const int fwd = precision == EVD_PREC_F16M ? EVD_PREC_F16X3 : precision;      // the forward's stream
    p.wstream = (const char*)(fwd == EVD_PREC_F16C ? net->pipe_c.data.p : net->pipe[fwd].data.p);
    p.bias = (const float*)net->bias.p;
    p.ray_batch = ray_batch; p.z = z; p.nsamp = nsamp; p.S = S; p.ncol = 11;
    p.D = net->D; p.skip = net->skip; p.nchunks = net->pipe_chunks[fwd]; p.nbias = (int)(net->bias.bytes / sizeof(float));
    p.raw = raw; p.feature = nullptr; p.feature_kind = 0; p.act = (char*)store;
    if (precision == EVD_PREC_F16C) {
        p.wscale = (const unsigned*)net->pipe_c.scales.p;
        p.pe_l = PE_L; p.pe_lv = PE_LV;
        return launch_nerf_train_fwd_f16c(p, as_stream(stream));
    }
    if (precision == EVD_PREC_F16M) return launch_nerf_train_fwd_f16x3_hi(p, as_stream(stream));
    return precision == EVD_PREC_F16 ? launch_nerf_train_fwd_f16(p, as_stream(stream))
           : precision == EVD_PREC_BF16 ? launch_nerf_train_fwd_bf16(p, as_stream(stream)) : launch_nerf_train_fwd_f16x3(p, as_stream(stream));
}

// scratch of the backward: the wgrad partial sums of the largest block (8 x 9 accumulator tiles per wavefront group) + the loss-scale word
static const int WGRAD_BLOCKS = 256;
size_t evd_nerf_backward_workspace_bytes(void) { return (size_t)WGRAD_BLOCKS * 8 * 9 * 4096 + 512; }

int evd_nerf_mlp_backward(const evd_nerf* net, int precision, const float* d_raw, long R, int S, void* store, size_t store_bytes,
                          const evd_nerf_grads* grads, const float* pts, const float* viewdirs, int vd_stride, float* d_pts, float* d_dirs,
                          void* workspace, size_t workspace_bytes, void* stream) {
    EVD_REQUIRE((!d_pts || pts) && (!d_dirs || viewdirs), "evd_nerf_mlp_backward: d_pts / d_dirs need the forward's pts / viewdirs");
    EVD_REQUIRE(net && d_raw && store && grads && workspace, "evd_nerf_mlp_backward: null argument");
    EVD_REQUIRE(train_built(net, precision),
                "evd_nerf_mlp_backward: the training path is built for precision f16 / bf16 / f16x3 / f16c / f16m on the netdepth 8, netwidth 256, skips [4] network");
    EVD_REQUIRE(R >= 0 && S >= 1, "evd_nerf_mlp_backward: bad shape R=%ld S=%d", R, S);
    if (R == 0) return EVD_OK;
    const long nsamp = R * (long)S;
    if (store_bytes < evd_nerf_train_store_bytes_prec(precision, nsamp))
        return fail(EVD_E_WORKSPACE, "evd_nerf_mlp_backward: store %zu < %zu bytes", store_bytes, evd_nerf_train_store_bytes_prec(precision, nsamp));
    if (workspace_bytes < evd_nerf_backward_workspace_bytes())
        return fail(EVD_E_WORKSPACE, "evd_nerf_mlp_backward: workspace %zu < %zu bytes", workspace_bytes, evd_nerf_backward_workspace_bytes());
    precision = store_prec(precision);          // the mixed modes: the float16 mode's store, W^T streams and kernels
    BwdPlan b;
    int rc0;
    b.d_raw = d_raw; b.nsamp = nsamp; b.tiles = train_tiles(nsamp); b.store = (char*)store;
    for (int k = 0; k < EVD_BWD_NSTREAMS; ++k) b.wt[k] = (const char*)net->bwd[precision][k].data.p;
    b.maps = (const int*)net->wmaps.p;
    char* w = (char*)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    b.maxbits = (unsigned*)w;
    b.partial = (float*)(w + 256);
    b.wgrad_blocks = WGRAD_BLOCKS; b.skip = net->skip;
    // the wgrad launches run on the handle's side stream, forked from / joined to the caller's stream with events (stream order
    // as seen by the caller is unchanged): measured 3.41 -> 3.05 ms at 2^19 samples with 192 persistent wgrad workgroups
    // (256: 3.14, 128: 3.52).  EVD_BWD_OVERLAP=0 keeps everything on the caller's stream; =N sets the workgroup count (at most
    // WGRAD_BLOCKS, what the workspace is sized for).
    b.side = nullptr; b.ev = nullptr;
    if (const int nb = bwd_overlap_blocks(WGRAD_BLOCKS)) {
        if ((rc0 = net->side.get(&b.side, &b.ev))) return rc0;
        b.wgrad_blocks = nb;
    }
    b.pts = pts; b.viewdirs = viewdirs; b.vd_stride = vd_stride; b.S = S; b.d_pts = d_pts; b.d_dirs = d_dirs;
    for (int l = 0; l < EVD_MAX_LAYERS; ++l) { b.grads.pts_w[l] = l < net->D ? grads->pts_w[l] : nullptr; b.grads.pts_b[l] = l < net->D ? grads->pts_b[l] : nullptr; }
    b.grads.views_w = grads->views_w; b.grads.views_b = grads->views_b; b.grads.feature_w = grads->feature_w; b.grads.feature_b = grads->feature_b;
    b.grads.alpha_w = grads->alpha_w; b.grads.alpha_b = grads->alpha_b; b.grads.rgb_w = grads->rgb_w; b.grads.rgb_b = grads->rgb_b;
    b.accumulate = grads->accumulate ? 1 : 0;
    return precision == EVD_PREC_F16 ? run_nerf_backward_f16(b, as_stream(stream))
           : precision == EVD_PREC_BF16 ? run_nerf_backward_bf16(b, as_stream(stream)) : run_nerf_backward_f16x3(b, as_stream(stream));
}

}  // extern "C"
